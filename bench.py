#!/usr/bin/env python
"""bench.py -- ALM iterations per second of the DOTs-SOCP hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload sphere10k|knot|torus100k|plane20]

One "step" is one full pass of the solver's main loop (reference solver_socp.py:656-823): steps 1-3
(Laplacian solve to the parity tolerance, cone projection, (q, lambda) + multiplier update) plus
whatever KKT evaluation / penalty update the lazy schedule puts on that iteration.  All state is
resident in HBM before the timed region starts.  Rank 0 prints ONE JSON line.

For N > 1 (launched by torch.distributed.run, one rank per GPU) the Laplacian solve is sharded over the
T+1 time modes with one all-gather per iteration; see dots-socp_amd/distributed.py.  `value` is whole-job iterations/s (the iterations are
collective: every rank advances the same ALM iteration).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured copy)

WORKLOADS = {
    # BASELINE.json configs[1]: "sphere mesh ~10k vertices, ntime=31, 1xMI355X, congestion=0.0"
    "sphere10k": dict(example="sphere", kw=dict(level=5), n_time=31, congestion=0.0, tol=1e-3),
    # configs[0] stand-in: knots_5-like tube, ntime=31
    "knot": dict(example="knot", kw={}, n_time=31, congestion=0.0, tol=1e-3),
    # configs[3]: ~100k-vertex torus, ntime=31
    "torus100k": dict(example="torus", kw=dict(nu=400, nv=250), n_time=31, congestion=0.0, tol=1e-3),
    # configs[2]: knots mesh, ntime=63, congestion 0.1 (time pitch 64)
    "knot63": dict(example="knot", kw={}, n_time=63, congestion=0.1, tol=1e-3),
    # configs[4] stand-in: 360 x 180 torus, V = 64 800, ntime=127 (time pitch 128), tol 1e-5
    "torus65k_T127": dict(example="torus", kw=dict(nu=360, nv=180), n_time=127, congestion=0.0, tol=1e-5),
    # the survey's analytic case
    "plane20": dict(example="plane", kw=dict(n=20), n_time=31, congestion=0.0, tol=1e-3),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=150)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--workload", default="sphere10k", choices=sorted(WORKLOADS))
    ap.add_argument("--lap-solver", default="modal_direct", choices=["modal_direct", "modal_pcg", "spacetime_pcg"])
    ap.add_argument("--preconditioner", default="multigrid", choices=["multigrid", "jacobi"])
    ap.add_argument("--cg-tol", type=float, default=None)
    ap.add_argument("--mg-coarsest", type=int, default=None, help="rows of the dense coarsest multigrid level (default: solver default)")
    ap.add_argument("--nd-leaf", type=int, default=None, help="leaf size of the nested dissection (default: solver default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-time-to-tol", action="store_true")
    ap.add_argument("--no-reorder", action="store_true", help="keep the generator's vertex numbering (A/B of the RCM renumbering)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="CPU budget of the cpu_baseline sample")
    return ap.parse_args()


def cpu_baseline(geom, n_time, congestion, budget_s):
    """Time the CPU oracle (numpy + SuperLU restatement of the reference) on the same workload.

    Bounded sample: operator setup is excluded (as for the GPU), then as many ALM iterations as
    fit in ~budget_s seconds (at least 2)."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("dots_oracle", os.path.join(ROOT, "oracle", "dots_oracle.py"))
    O = importlib.util.module_from_spec(spec)
    sys.modules["dots_oracle"] = O
    spec.loader.exec_module(O)
    import contextlib

    try:
        from threadpoolctl import threadpool_limits

        limit = threadpool_limits(limits=1)
    except Exception:  # pragma: no cover
        limit = contextlib.nullcontext()
    with limit:
        t0 = time.perf_counter()
        s = O.OracleSolver(n_time, geom, congestion=congestion)
        s.scale_z(2.0)
        setup = time.perf_counter() - t0
        s.iterate()   # warm caches
        n, t0 = 0, time.perf_counter()
        while True:
            s.iterate()
            n += 1
            el = time.perf_counter() - t0
            if (el > budget_s and n >= 2) or n >= 400:
                break
    return {
        "value": n / el, "unit": "ALM iterations/s", "cores": 1, "kind": "port",
        "sample": f"{n} ALM iterations of the numpy/SuperLU oracle after setup ({setup:.1f} s setup excluded), "
                  f"1 BLAS thread, host {os.cpu_count()} logical CPUs",
    }


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world

    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists for the product path)")
    # one rank per GPU; DOTS_DIST_BACKEND=gloo lets several ranks share one GPU (rehearsal on a 1-GPU box)
    backend = os.environ.get("DOTS_DIST_BACKEND", "nccl")
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)

    from dots_socp_amd import meshes
    from dots_socp_amd.socp.solver_socp import AlmSolver, DEFAULT_CG_TOL

    wl = WORKLOADS[args.workload]
    geom, _scale = meshes.example(wl["example"], **wl["kw"])
    n_time, congestion, tol = wl["n_time"], wl["congestion"], wl["tol"]
    cg_tol = args.cg_tol if args.cg_tol is not None else DEFAULT_CG_TOL
    mg_kw = {} if args.mg_coarsest is None else {"mg_coarsest": args.mg_coarsest}
    if args.nd_leaf is not None:
        mg_kw["nd_leaf"] = args.nd_leaf
    V, F = geom["vertices"].shape[0], geom["triangles"].shape[0]

    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
        from dots_socp_amd.distributed import ShardedAlmSolver, TorchComm

        if args.lap_solver == "spacetime_pcg":
            raise SystemExit("--gpus N>1 shards the time modes: use --lap-solver modal_direct or modal_pcg")
        alm = ShardedAlmSolver(n_time, geom, comm=TorchComm(), congestion=congestion, nit=args.warmup + args.steps + 8,
                               tol=1e-30, cg_tol=cg_tol, device=local_rank, preconditioner=args.preconditioner, lap_solver=args.lap_solver,
                               time_limit=float("inf"), reorder=not args.no_reorder, **mg_kw)
    else:
        alm = AlmSolver(n_time, geom, congestion=congestion, nit=args.warmup + args.steps + 8, tol=1e-30,
                        lap_solver=args.lap_solver, cg_tol=cg_tol, device=local_rank, preconditioner=args.preconditioner,
                        time_limit=float("inf"), reorder=not args.no_reorder, **mg_kw)

    def barrier():
        alm.dev.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        alm.iterate()
    barrier()
    cg0 = alm.cg_total
    t0 = time.perf_counter()
    for _ in range(args.steps):
        alm.iterate()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    cg_per_it = (alm.cg_total - cg0) / max(args.steps, 1)
    steps_time = dict(alm.run_history.steps_time)

    # ---- roofline of the dominant kernel(s), measured with hipEvents on the context's own stream in the
    # state the timed region left behind
    if getattr(alm, "front_summary", None):
        fs = alm.front_summary
        ms_solve, bytes_solve = alm.dev.bench_kernel(which=3, reps=100)
        achieved = bytes_solve / (ms_solve * 1e-3) / 1e9
        roofline = {
            "bound": "hbm", "kernel": "k_front_fwd + k_front_bwd (the two triangular sweeps of the multifrontal factor: "
                                      f"{2 * fs['levels']} launches per solve, one per tree height and sweep)",
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
            "ms_per_solve": ms_solve, "launches_per_solve": 2 * fs["levels"], "ms_per_launch": ms_solve / (2 * fs["levels"]),
            "algorithmic_bytes_per_solve": bytes_solve, "factor": fs,
        }
    else:
        ms_apply, bytes_apply = alm.dev.bench_kernel(which=0, reps=200)
        ms_update, bytes_update = alm.dev.bench_kernel(which=1, reps=200)
        achieved = bytes_apply / (ms_apply * 1e-3) / 1e9
        roofline = {
            "bound": "hbm", "kernel": "k_cg_apply (fused direction update + K p + p.Kp partials)",
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": None,
            "ms_per_launch": ms_apply, "algorithmic_bytes_per_launch": bytes_apply,
            "second_kernel": {"kernel": "k_cg_update", "ms_per_launch": ms_update,
                              "achieved": bytes_update / (ms_update * 1e-3) / 1e9, "algorithmic_bytes_per_launch": bytes_update},
            "working_set_note": "CG working set fits the 256 MiB Infinity Cache at this size" if V * (n_time + 1) * 8 * 6 < 256e6 else "",
        }
    if getattr(alm, "mg_summary", None):
        ms_vc, bytes_vc = alm.dev.bench_kernel(which=2, reps=100)
        roofline["multigrid_vcycle"] = {
            "ms_per_cycle": ms_vc, "finest_level_algorithmic_bytes": bytes_vc,
            "achieved_finest_only": bytes_vc / (ms_vc * 1e-3) / 1e9, "hierarchy": alm.mg_summary,
        }
    # SURVEY.md section 8(d)'s per-iteration figure: 2 S (every state array read and written once) + the right-hand
    # side + the solve, over the measured step time
    S_bytes = 8 * ((n_time + 1) * V + 7 * n_time * V + 6 * (n_time + 1) * F + 36 * n_time * F)
    rhs_bytes = 8 * (3 * n_time * V + 6 * (n_time + 1) * F + 2 * (n_time + 1) * V)
    if "algorithmic_bytes_per_solve" in roofline:
        solve_bytes = roofline["algorithmic_bytes_per_solve"]
    else:
        solve_bytes = cg_per_it * (roofline["algorithmic_bytes_per_launch"] + roofline["second_kernel"]["algorithmic_bytes_per_launch"])
    it_bytes = 2 * S_bytes + rhs_bytes + solve_bytes
    roofline["whole_iteration"] = {
        "algorithmic_bytes": it_bytes, "achieved": it_bytes / (elapsed / args.steps) / 1e9, "unit": "GB/s",
        "frac": it_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS,
        "note": "2*S + rhs + solve bytes over the measured step (SURVEY.md 8d); the driver moves less than 2*S on iterations that do not store z_mid",
    }
    # one streaming launch of known size: calibrates the FETCH_SIZE / WRITE_SIZE counters when this command runs
    # under rocprofv3 --pmc (profiles/tools/pmc_summary.py); a single launch, outside the timed region
    _, calib_bytes = alm.dev.bench_kernel(which=4, reps=1)
    roofline["pmc_calibration_bytes_each_way"] = calib_bytes
    traffic_file = os.path.join(ROOT, "profiles", f"traffic_{args.workload}.json")
    if world == 1 and args.lap_solver == "modal_direct" and os.path.exists(traffic_file):
        with open(traffic_file) as fh:
            tr = json.load(fh)
        roofline["traffic"] = tr["bytes_per_solve"]
        roofline["traffic_source"] = tr["source"]
    dev_bytes = alm.dev.device_bytes()
    alm.close()

    extra = {}
    if rank == 0 and world == 1 and not args.no_time_to_tol:
        t1 = time.perf_counter()
        solver = AlmSolver(n_time, geom, congestion=congestion, nit=20000, tol=tol, lap_solver=args.lap_solver, cg_tol=cg_tol,
                           device=local_rank, preconditioner=args.preconditioner, reorder=not args.no_reorder, **mg_kw)
        setup_s = time.perf_counter() - t1
        t1 = time.perf_counter()
        while not solver.iterate():
            pass
        _, hist = solver.finalize(download=False)
        solve_s = time.perf_counter() - t1
        solver.close()
        extra["time_to_tol"] = {
            "tol": tol, "seconds": solve_s, "setup_seconds": setup_s, "iterations": int(hist.kkt_iteration[-1]) + 1,
            "transport_cost": float(hist.history["Transportation cost"][-1]),
            "max_kkt": float(max(hist.kkt_errors[-1])), "pcg_iterations": hist.solver_stats["cg_iterations"],
        }

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return
    line = {
        "metric": "ALM iterations/s", "value": args.steps / elapsed, "unit": "iterations/s",
        "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {
            "workload": f"{args.workload}: {wl['example']} mesh V={V} F={F}, ntime={n_time}, congestion={congestion}, "
                        f"lap_solver={args.lap_solver}{'' if world == 1 else ' (mode-sharded)'}"
                        + ("" if args.lap_solver == "modal_direct" else f", cg_tol={cg_tol:g}, preconditioner={args.preconditioner}"),
            "unknowns": V * (n_time + 1), "state_bytes": 8 * ((n_time + 1) * V + 7 * n_time * V + 6 * (n_time + 1) * F + 36 * n_time * F),
            "device_bytes": dev_bytes, "pcg_iterations_per_step": cg_per_it,
            "step_seconds": steps_time,
        },
        "roofline": roofline,
    }
    line.update(extra)
    if world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(geom, n_time, congestion, args.cpu_seconds)
    else:
        line["cpu_baseline"] = None
    print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
