#!/usr/bin/env python
"""bench.py -- ALM iterations per second of the DOTs-SOCP hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload knot|sphere10k|knot63|torus100k|torus65k_T127|torus500k|plane20]

One "step" is one full pass of the solver's main loop (reference solver_socp.py:656-823): steps 1-3 (Laplacian solve,
cone projection, (q, lambda) + multiplier update) plus whatever KKT evaluation / penalty update the reference's lazy
schedule puts on that iteration.  All state is resident in HBM before the timed region starts.  Rank 0 prints ONE
JSON line.

* ``value`` / ``ms_per_step``: the workload BASELINE.json's metric is quoted on (``knot``: the knots_5 stand-in, T=31),
  W untimed + exactly K timed iterations as the driver asks.
* ``steady_state``: the same solver run on to iteration 100, then 200 iterations timed (independent of K / W: the first
  25 iterations are dense in penalty updates and KKT checks, admm_tools.py:43-48).
* ``configs``: all five BASELINE.json configurations (N = 1): steady-state it/s, ms/step, time to tol, roofline.
* ``roofline``: the dominant kernels (the two triangular sweeps of the direct solve), hipEvent-timed on the context's
  own stream; ``whole_iteration`` counts the bytes the implemented iteration moves (no z_mid traffic on the
  iterations that do not materialise it).
* ``cpu_baseline``: the numpy/SuperLU oracle on the same workload on this host's cores (bounded sample).

For N > 1 the iteration is sharded over time slabs (dots_socp_amd/distributed.py), one rank per GPU over RCCL.  Started as
``python bench.py --gpus N`` the script launches its N ranks itself (torch.distributed.run, before anything touches the GPU) and
relays rank 0's line and the ranks' exit code; started by torch.distributed.run it reads RANK / LOCAL_RANK / WORLD_SIZE.  ``value``
is whole-job iterations/s of the metric's configuration (the iterations are collective: every rank advances the same ALM
iteration), "scaling": "strong"; ``sharded_configs`` times the configurations BASELINE.json places on several GPUs
(torus100k: configs[3], torus65k_T127: configs[4]) on the same ranks: steady it/s, exchange seconds, per-rank device bytes, a
run to tol.  With fewer GPUs than ranks (a one-GPU box) the ranks share the GPUs over gloo (a rehearsal of the command, not a
scaling measurement; DOTS_DIST_BACKEND overrides).
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured copy)

WORKLOADS = {
    # configs[0] stand-in, the configuration the metric is quoted on: knots_5-like tube, ntime=31, tol 1e-3
    "knot": dict(example="knot", kw={}, n_time=31, congestion=0.0, tol=1e-3, config=0),
    # configs[1]: "sphere mesh ~10k vertices, ntime=31, 1xMI355X, congestion=0.0"
    "sphere10k": dict(example="sphere", kw=dict(level=5), n_time=31, congestion=0.0, tol=1e-3, config=1),
    # configs[2]: knots mesh, ntime=63, congestion 0.1 (time pitch 64)
    "knot63": dict(example="knot", kw={}, n_time=63, congestion=0.1, tol=1e-3, config=2),
    # configs[3]: ~100k-vertex torus, ntime=31
    "torus100k": dict(example="torus", kw=dict(nu=400, nv=250), n_time=31, congestion=0.0, tol=1e-3, config=3),
    # configs[4] stand-in: 360 x 180 torus, V = 64 800, ntime=127 (time pitch 128), tol 1e-5
    "torus65k_T127": dict(example="torus", kw=dict(nu=360, nv=180), n_time=127, congestion=0.0, tol=1e-5, config=4),
    # scale check beyond the configurations (not a bench line): 1000 x 500 torus, V = 500 000, factor ~10 GB
    "torus500k": dict(example="torus", kw=dict(nu=1000, nv=500), n_time=31, congestion=0.0, tol=1e-3, config=None),
    # the survey's analytic case
    "plane20": dict(example="plane", kw=dict(n=20), n_time=31, congestion=0.0, tol=1e-3, config=None),
}
ALL_CONFIGS = ["knot", "sphere10k", "knot63", "torus100k", "torus65k_T127"]
SHARDED_CONFIGS = ["torus100k", "torus65k_T127"]      # the configurations BASELINE.json places on 4 and 8 GPUs
STEADY = {"torus65k_T127": (40, 100), "torus100k": (100, 200), "torus500k": (20, 40)}     # (first timed iteration, timed iterations); default (100, 200)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=150)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--workload", default="knot", choices=sorted(WORKLOADS))
    ap.add_argument("--lap-solver", default="modal_direct", choices=["modal_direct", "modal_pcg", "spacetime_pcg"])
    ap.add_argument("--preconditioner", default="multigrid", choices=["multigrid", "jacobi"])
    ap.add_argument("--cg-tol", type=float, default=None)
    ap.add_argument("--mg-coarsest", type=int, default=None, help="rows of the dense coarsest multigrid level (default: solver default)")
    ap.add_argument("--nd-leaf", type=int, default=None, help="leaf size of the nested dissection (default: solver default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-time-to-tol", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the table of the five BASELINE configurations (N > 1: the sharded ones)")
    ap.add_argument("--no-alternatives", action="store_true", help="skip the PCG alternatives of the Laplacian solve on torus100k")
    ap.add_argument("--tol-seconds", type=float, default=None,
                    help="wall-clock cap of a sharded run to tol (default: 150 s over RCCL; 20 s when the ranks share GPUs over gloo, a rehearsal)")
    ap.add_argument("--no-reorder", action="store_true", help="keep the generator's vertex numbering (A/B of the renumbering)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="CPU budget of the cpu_baseline sample")
    return ap.parse_args()


def git_commit():
    try:
        return subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True, timeout=5).stdout.strip() or None
    except Exception:
        return None


def iteration_bytes(T, V, F, solve_bytes, z_mid_stored=False, carried=True):
    """Algorithmic bytes ONE iteration of the implemented algorithm moves, N = (T+1) V.
    ``carried`` (round 4, the default path of the direct solver up to a time pitch of 128: DOTS_STEP_CARRY): steps 2+3 leave per-corner
    sums (cn_sq: 6TF values, cn_g: 3(T+1)F) and the right-hand side / projection stream those instead of B, E and beta_mid:
        right-hand side        A, lambda_c, mu r (3TV); cn_g r (3(T+1)F); b-hat w (N)
        cone projection        cn_sq r (6TF); A, beta_fst, beta_end r, z_fst, z_end, lambda w (6TV)
        solve                  the factor twice + four vector passes (``solve_bytes``)
        inverse transform      2N
        steps 2+3, triangles   beta_mid r+w (36TF); B, E r+w (12(T+1)F); phi r (N); cone multiplier r (TV); cn_sq, cn_g w (6TF + 3(T+1)F)
        steps 2+3, vertices    11TV
    not carried (rounds 1-3; today the iteration after a penalty update and time pitches above 128):
        right-hand side        ... B, E r (6(T+1)F) instead of cn_g
        cone projection        beta_mid r (18TF); B r (3(T+1)F) instead of cn_sq
        steps 2+3              without the cn_* stores
    z_mid (18TF) is written only on the iterations whose results are read back, never read."""
    N = (T + 1) * V
    if carried:
        words = (3 * T * V + 3 * (T + 1) * F + N) + (6 * T * F + 6 * T * V) + 2 * N \
            + (36 * T * F + 12 * (T + 1) * F + N + T * V + 6 * T * F + 3 * (T + 1) * F) + 11 * T * V
    else:
        words = (3 * T * V + 6 * (T + 1) * F + N) + (18 * T * F + 3 * (T + 1) * F + 6 * T * V) + 2 * N \
            + (36 * T * F + 12 * (T + 1) * F + N + T * V) + 11 * T * V
    if z_mid_stored:
        words += 18 * T * F
    return 8.0 * words + solve_bytes


def cpu_baseline(geom, n_time, congestion, budget_s):
    """Time the CPU oracle (numpy + SuperLU restatement of the reference) on the same workload.

    Bounded sample: operator setup is excluded (as for the GPU), then as many ALM iterations as fit in ~budget_s / 2
    seconds (at least 2), once on one thread and once the way the reference runs by default (is_multi_threads=True,
    solver_socp.py:674-696: the Laplacian solve and the cone projection on two threads).  The faster one is `value`."""
    import importlib.util
    import threading

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

    def threaded_iterate(s):
        a, b = threading.Thread(target=s.step_laplacian), threading.Thread(target=s.step_soc_projection)
        a.start(); b.start(); a.join(); b.join()
        s.step_q_lambda()
        s.step_multipliers()

    with limit:
        t0 = time.perf_counter()
        s = O.OracleSolver(n_time, geom, congestion=congestion)
        s.scale_z(2.0)
        setup = time.perf_counter() - t0
        s.iterate()   # warm caches
        rates = {}
        for tag, step in (("1 thread", s.iterate), ("2 threads", lambda: threaded_iterate(s))):
            n, t0 = 0, time.perf_counter()
            while True:
                step()
                n += 1
                el = time.perf_counter() - t0
                if (el > 0.5 * budget_s and n >= 2) or n >= 400:
                    break
            rates[tag] = (n / el, n)
    best = max(rates, key=lambda k: rates[k][0])
    return {
        "value": rates[best][0], "unit": "ALM iterations/s", "cores": 1 if best == "1 thread" else 2, "kind": "port",
        "sample": f"numpy/SuperLU oracle after setup ({setup:.1f} s setup excluded): {rates['1 thread'][1]} iterations on 1 thread = "
                  f"{rates['1 thread'][0]:.3f} it/s, {rates['2 threads'][1]} iterations with step 1 on 2 threads (the reference's default "
                  f"is_multi_threads; it also runs numexpr on 4 threads, which numpy does not) = {rates['2 threads'][0]:.3f} it/s; "
                  f"1 BLAS thread, host {os.cpu_count()} logical CPUs",
    }


class Timer:
    def __init__(self, alm, dist):
        self.alm, self.dist = alm, dist

    def barrier(self):
        import torch

        self.alm.dev.sync()
        torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()

    def run(self, n):
        """n iterations between two barriers; seconds (max over ranks)."""
        import torch

        self.barrier()
        t0 = time.perf_counter()
        for _ in range(n):
            self.alm.iterate()
        self.barrier()
        el = time.perf_counter() - t0
        if self.dist is not None:
            t = torch.tensor([el], dtype=torch.float64, device="cuda" if self.dist.get_backend() == "nccl" else "cpu")
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            el = float(t.item())
        return el


def build_solver(args, wl, geom, nit, local_rank, world, tol=1e-30, time_limit=float("inf"), lap_solver=None, preconditioner=None):
    from dots_socp_amd.socp.solver_socp import AlmSolver, DEFAULT_CG_TOL

    cg_tol = args.cg_tol if args.cg_tol is not None else DEFAULT_CG_TOL
    kw = {} if args.mg_coarsest is None else {"mg_coarsest": args.mg_coarsest}
    if args.nd_leaf is not None:
        kw["nd_leaf"] = args.nd_leaf
    lap_solver = lap_solver or args.lap_solver
    common = dict(congestion=wl["congestion"], nit=nit, tol=tol, cg_tol=cg_tol, device=local_rank, preconditioner=preconditioner or args.preconditioner,
                  lap_solver=lap_solver, time_limit=time_limit, reorder=not args.no_reorder, **kw)
    if world > 1:
        from dots_socp_amd.distributed import ShardedAlmSolver, TorchComm

        if lap_solver == "spacetime_pcg":
            raise SystemExit("--gpus N>1 shards the time modes: use --lap-solver modal_direct or modal_pcg")
        return ShardedAlmSolver(wl["n_time"], geom, comm=TorchComm(), **common)
    return AlmSolver(wl["n_time"], geom, **common)


def solve_roofline(alm, V, n_time):
    """Roofline of the dominant kernel(s), hipEvent-timed on the context's own stream in the state the run left behind."""
    if getattr(alm, "front_summary", None):
        fs = alm.front_summary
        ms_solve, bytes_solve = alm.dev.bench_kernel(which=3, reps=100)
        achieved = bytes_solve / (ms_solve * 1e-3) / 1e9
        launches = alm.dev.front_launches() if hasattr(alm.dev, "front_launches") else 2 * fs["levels"]
        return {
            "bound": "hbm", "kernel": "k_front_fwd + k_front_bwd (the two triangular sweeps of the multifrontal factor)",
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
            "ms_per_solve": ms_solve, "launches_per_solve": launches, "ms_per_launch": ms_solve / max(launches, 1),
            "algorithmic_bytes_per_solve": bytes_solve, "factor": fs,
            "bytes_note": "achieved / frac use the algorithmic bytes of the solve: the factor with one block per tree node, twice, + 4 vector "
                          "passes; the sweeps run on bands of merged tree heights (factor.bands) whose blocks hold more entries "
                          "(factor.bytes_per_solve_as_installed): the extra bytes are the price of fewer dependent launches and are NOT counted",
        }, bytes_solve
    ms_apply, bytes_apply = alm.dev.bench_kernel(which=0, reps=200)
    ms_update, bytes_update = alm.dev.bench_kernel(which=1, reps=200)
    achieved = bytes_apply / (ms_apply * 1e-3) / 1e9
    roofline = {
        "bound": "hbm", "kernel": "k_cg_apply (fused direction update + K p + p.Kp partials)",
        "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
        "ms_per_launch": ms_apply, "algorithmic_bytes_per_launch": bytes_apply,
        "second_kernel": {"kernel": "k_cg_update", "ms_per_launch": ms_update,
                          "achieved": bytes_update / (ms_update * 1e-3) / 1e9, "algorithmic_bytes_per_launch": bytes_update},
        "working_set_note": "CG working set fits the 256 MiB Infinity Cache at this size" if V * (n_time + 1) * 8 * 6 < 256e6 else "",
    }
    if getattr(alm, "mg_summary", None):
        ms_vc, bytes_vc = alm.dev.bench_kernel(which=2, reps=100)
        roofline["multigrid_vcycle"] = {"ms_per_cycle": ms_vc, "finest_level_algorithmic_bytes": bytes_vc,
                                        "achieved_finest_only": bytes_vc / (ms_vc * 1e-3) / 1e9, "hierarchy": alm.mg_summary}
    return roofline, None


def whole_iteration(T, V, F, solve_bytes, ms_per_step, label):
    b = iteration_bytes(T, V, F, solve_bytes, carried=T + 1 <= 128)
    return {"algorithmic_bytes": b, "achieved": b / (ms_per_step * 1e-3) / 1e9, "unit": "GB/s", "frac": b / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "over": label, "note": "bytes the implemented iteration moves when z_mid is not materialised and the next iteration's gathers are carried "
                                   "(bench.py: iteration_bytes; round 3's iteration moved 72 T F bytes more); KKT evaluations and penalty updates inside the "
                                   "window are not counted as bytes"}


def time_to_tol(args, wl, geom, local_rank):
    t1 = time.perf_counter()
    solver = build_solver(args, wl, geom, 20000, local_rank, 1, tol=wl["tol"], time_limit=1000)
    setup_s = time.perf_counter() - t1
    t1 = time.perf_counter()
    while not solver.iterate():
        pass
    _, hist = solver.finalize(download=False)
    solve_s = time.perf_counter() - t1
    solver.close()
    return {"tol": wl["tol"], "seconds": solve_s, "setup_seconds": setup_s, "iterations": int(hist.kkt_iteration[-1]) + 1,
            "transport_cost": float(hist.history["Transportation cost"][-1]), "max_kkt": float(max(hist.kkt_errors[-1])),
            "pcg_iterations": hist.solver_stats["cg_iterations"]}


def config_entry(args, name, local_rank):
    """One BASELINE configuration on one GPU: steady-state rate, sweeps' roofline, time to tol."""
    from dots_socp_amd import meshes

    wl = WORKLOADS[name]
    geom, _ = meshes.example(wl["example"], **wl["kw"])
    V, F, T = geom["vertices"].shape[0], geom["triangles"].shape[0], wl["n_time"]
    first, count = STEADY.get(name, (100, 200))
    alm = build_solver(args, wl, geom, first + count + 8, local_rank, 1)
    tm = Timer(alm, None)
    for _ in range(first):
        alm.iterate()
    el = tm.run(count)
    ms = 1e3 * el / count
    roof, solve_bytes = solve_roofline(alm, V, T)
    entry = {
        "workload": name, "baseline_config": wl["config"], "V": V, "F": F, "ntime": T, "congestion": wl["congestion"], "tol": wl["tol"],
        "iterations_per_s": count / el, "ms_per_step": ms, "window": f"iterations {first}..{first + count - 1}",
        "device_bytes": alm.dev.device_bytes(),
        "roofline": {k: roof[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "ms_per_solve", "launches_per_solve",
                                          "algorithmic_bytes_per_solve") if k in roof},
    }
    if solve_bytes is not None:
        entry["roofline"]["whole_iteration"] = whole_iteration(T, V, F, solve_bytes, ms, entry["window"])
    alm.close()
    if not args.no_time_to_tol:
        entry["time_to_tol"] = time_to_tol(args, wl, geom, local_rank)
    return entry


def gather_ints(dist, value):
    """``value`` of every rank (a list on every rank)."""
    import torch

    if dist is None:
        return [int(value)]
    t = torch.tensor([int(value)], dtype=torch.int64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
    out = torch.zeros(dist.get_world_size(), dtype=torch.int64, device=t.device)
    dist.all_gather_into_tensor(out, t)
    return [int(x) for x in out.cpu()]


def sharded_entry(args, name, local_rank, world, dist):
    """One multi-GPU configuration of BASELINE.json on the ranks of this job: steady-state rate (barriers, max over ranks), what
    the exchanges cost, per-rank device memory, and a run to the configuration's tol (capped at --tol-seconds)."""
    from dots_socp_amd import meshes
    from dots_socp_amd.distributed import ShardedAlmSolver

    wl = WORKLOADS[name]
    geom, _ = meshes.example(wl["example"], **wl["kw"])
    V, F, T = geom["vertices"].shape[0], geom["triangles"].shape[0], wl["n_time"]
    first, count = STEADY.get(name, (100, 200))
    t0 = time.perf_counter()
    alm = build_solver(args, wl, geom, first + count + 8, local_rank, world)
    setup_s = time.perf_counter() - t0
    tm = Timer(alm, dist)
    for _ in range(first):
        alm.iterate()
    comm0 = dict(alm.comm.bytes)
    el = tm.run(count)
    alm._collect_step_times(wait=True)
    sent = {k: alm.comm.bytes[k] - comm0[k] for k in comm0}
    entry = {
        "workload": name, "baseline_config": wl["config"], "V": V, "F": F, "ntime": T, "congestion": wl["congestion"], "tol": wl["tol"],
        "ranks": world, "nodes_per_rank": alm.dev.slab[2], "iterations_per_s": count / el, "ms_per_step": 1e3 * el / count,
        "window": f"iterations {first}..{first + count - 1}", "setup_seconds": setup_s,
        "step_seconds_estimated_for_all_iterations": dict(alm.run_history.steps_time),
        "exchange_seconds_per_iteration": alm.run_history.steps_time.get(ShardedAlmSolver.EXCHANGE_TAG, 0.0) / max(alm.counter_main + 1, 1),
        "bytes_handed_to_collectives_per_iteration_rank0": {k: v / count for k, v in sent.items() if v},
        "device_bytes_per_rank": gather_ints(dist, alm.dev.device_bytes()),
    }
    alm.close()
    if not args.no_time_to_tol:
        t1 = time.perf_counter()
        solver = build_solver(args, wl, geom, 20000, local_rank, world, tol=wl["tol"], time_limit=args.tol_seconds)
        setup_s = time.perf_counter() - t1
        t1 = time.perf_counter()
        while not solver.iterate():
            pass
        _, hist = solver.finalize(download=False)
        solve_s = time.perf_counter() - t1
        max_kkt = float(max(hist.kkt_errors[-1]))
        entry["time_to_tol"] = {"tol": wl["tol"], "seconds": solve_s, "setup_seconds": setup_s, "iterations": int(hist.kkt_iteration[-1]) + 1,
                                "transport_cost": float(hist.history["Transportation cost"][-1]), "max_kkt": max_kkt,
                                "reached_tol": bool(max_kkt < wl["tol"]), "wall_clock_cap_seconds": args.tol_seconds}
        solver.close()
    return entry


def solver_alternatives(args, local_rank, name="torus100k", budget_s=20.0):
    """The north star's named solvers of step 1 beside the direct one, on the ~100k-vertex configuration, N = 1: the batched
    multigrid-PCG on the time modes and the Jacobi-PCG on the assembled space-time operator (CSR row blocks staged in LDS):
    ALM it/s, PCG iterations per step, and the hipEvent-timed bandwidth of the operator application k_cg_apply."""
    from dots_socp_amd import meshes

    wl = WORKLOADS[name]
    geom, _ = meshes.example(wl["example"], **wl["kw"])
    V, T = geom["vertices"].shape[0], wl["n_time"]
    out = []
    for lap, pre, warm, count in (("modal_pcg", "multigrid", 20, 30), ("spacetime_pcg", "jacobi", 10, 10)):
        t0 = time.perf_counter()
        alm = build_solver(args, wl, geom, warm + count + 8, local_rank, 1, lap_solver=lap, preconditioner=pre)
        tm = Timer(alm, None)
        for _ in range(warm):
            alm.iterate()
        cg0 = alm.cg_total
        el = tm.run(count)
        roof, _ = solve_roofline(alm, V, T)
        out.append({"workload": name, "lap_solver": lap, "preconditioner": pre, "iterations_per_s": count / el, "ms_per_step": 1e3 * el / count,
                    "window": f"iterations {warm}..{warm + count - 1}", "pcg_iterations_per_step": (alm.cg_total - cg0) / count,
                    "pcg_not_converged": int(alm.cg_fail), "cg_tol": alm.dev.params.cg_tol,
                    "roofline": {k: roof[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "ms_per_launch",
                                                      "algorithmic_bytes_per_launch") if k in roof},
                    "seconds_spent": time.perf_counter() - t0})
        alm.close()
        if time.perf_counter() - t0 > budget_s:
            break
    return out


def traffic_record(workload):
    """PMC traffic of one solve (roofline.traffic).  Counters cannot be read inside this process: the figure is the one the
    rocprofv3 --pmc passes of THIS command recorded (profiles/tools/collect_traffic.sh).  It is stamped with the hash of the
    sweep kernels' sources it was measured on and dropped (traffic = null) when they have changed since."""
    path = os.path.join(ROOT, "profiles", f"traffic_{workload}.json")
    if not os.path.exists(path):
        return None, "no recorded PMC passes for this workload"
    with open(path) as fh:
        tr = json.load(fh)
    now = kernel_digest()
    src = tr["source"] + "; NOT measured in this run"
    if tr.get("kernel_sources_sha256") != now:
        return None, src + f"; STALE: the sweep kernels changed since (recorded {str(tr.get('kernel_sources_sha256'))[:12]}, now {now[:12]}): traffic withheld"
    return tr["bytes_per_solve"], src


def kernel_digest():
    import hashlib

    h = hashlib.sha256()
    for f in ("kernels_front.hip", "kernels_factor.hip"):      # what the sweeps' traffic depends on
        with open(os.path.join(ROOT, "dots_socp_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def launch_ranks(args):
    """``python bench.py --gpus N`` without a launcher: start the N ranks (one per GPU) before anything in this process touches
    the GPU, let rank 0's JSON line through, return the ranks' exit code."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    # --standalone: the launcher picks a free rendezvous port itself (no bind-and-close race), on the loopback address
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           os.path.abspath(__file__)] + sys.argv[1:]
    # rank 0 prints the one JSON line; anything else the ranks write to stdout (gloo announces its connections there) goes to stderr
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = proc.stdout.splitlines()
    result = [ln for ln in lines if ln.startswith('{"metric"')]
    for ln in lines:
        if ln not in result and ln.strip():
            print(ln, file=sys.stderr)
    if result:
        print(result[-1])
    return proc.returncode if result or proc.returncode else 1


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world

    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists for the product path)")
    # one rank per GPU over RCCL; with fewer GPUs than ranks (a one-GPU box) the ranks share the GPUs and exchange over gloo:
    # RCCL refuses two ranks on one device.  DOTS_DIST_BACKEND overrides.
    from dots_socp_amd._lib import env_choice

    n_dev = torch.cuda.device_count()
    backend = env_choice("DOTS_DIST_BACKEND", ("nccl", "gloo"), "nccl" if n_dev >= world else "gloo")
    local_rank = local_rank % n_dev
    torch.cuda.set_device(local_rank)
    if args.tol_seconds is None:
        args.tol_seconds = 150.0 if backend == "nccl" else 20.0

    from dots_socp_amd import meshes

    wl = WORKLOADS[args.workload]
    geom, _scale = meshes.example(wl["example"], **wl["kw"])
    n_time, congestion = wl["n_time"], wl["congestion"]
    V, F = geom["vertices"].shape[0], geom["triangles"].shape[0]

    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)

    steady_first, steady_count = STEADY.get(args.workload, (100, 200))
    total = max(args.warmup + args.steps, steady_first) + steady_count + 8
    alm = build_solver(args, wl, geom, total, local_rank, world)
    tm = Timer(alm, dist)

    for _ in range(args.warmup):
        alm.iterate()
    cg0 = alm.cg_total
    elapsed = tm.run(args.steps)                                   # the driver's K steps after W warm-up steps
    cg_per_it = (alm.cg_total - cg0) / max(args.steps, 1)
    done = args.warmup + args.steps
    for _ in range(max(0, steady_first - done)):
        alm.iterate()
    steady_from = max(done, steady_first)
    steady_el = tm.run(steady_count)
    steady_ms = 1e3 * steady_el / steady_count

    roofline, solve_bytes = solve_roofline(alm, V, n_time)
    if solve_bytes is None:
        solve_bytes = cg_per_it * (roofline["algorithmic_bytes_per_launch"] + roofline["second_kernel"]["algorithmic_bytes_per_launch"])
    roofline["whole_iteration"] = whole_iteration(n_time, V, F, solve_bytes, steady_ms, f"iterations {steady_from}..{steady_from + steady_count - 1}")
    # one streaming launch of known size: calibrates the FETCH_SIZE / WRITE_SIZE counters when this command runs
    # under rocprofv3 --pmc (profiles/tools/pmc_summary.py); a single launch, outside the timed region
    _, calib_bytes = alm.dev.bench_kernel(which=4, reps=1)
    roofline["pmc_calibration_bytes_each_way"] = calib_bytes
    if world == 1 and args.lap_solver == "modal_direct":
        roofline["traffic"], roofline["traffic_source"] = traffic_record(args.workload)
    roofline["kernel_sources_sha256"] = kernel_digest()
    dev_bytes = alm.dev.device_bytes()
    dev_bytes_ranks = gather_ints(dist, dev_bytes)
    alm._collect_step_times(wait=True)
    steps_time_all = dict(alm.run_history.steps_time)
    steps_note = alm.run_history.steps_time_note
    alm.close()

    extra = {}
    if rank == 0 and world == 1 and not args.no_time_to_tol:
        extra["time_to_tol"] = time_to_tol(args, wl, geom, local_rank)
    if rank == 0 and world == 1 and not args.no_configs and args.lap_solver == "modal_direct":
        table = []
        for name in ALL_CONFIGS:
            if name == args.workload:
                e = {"workload": name, "baseline_config": wl["config"], "V": V, "F": F, "ntime": n_time, "congestion": congestion, "tol": wl["tol"],
                     "iterations_per_s": steady_count / steady_el, "ms_per_step": steady_ms,
                     "window": f"iterations {steady_from}..{steady_from + steady_count - 1}", "device_bytes": dev_bytes,
                     "roofline": {k: roofline[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "ms_per_solve",
                                                           "launches_per_solve", "algorithmic_bytes_per_solve", "whole_iteration") if k in roofline}}
                if "time_to_tol" in extra:
                    e["time_to_tol"] = extra["time_to_tol"]
                table.append(e)
            else:
                table.append(config_entry(args, name, local_rank))
        extra["configs"] = table
    if rank == 0 and world == 1 and not args.no_alternatives and not args.no_configs and args.lap_solver == "modal_direct":
        extra["solver_alternatives"] = solver_alternatives(args, local_rank)
    if world > 1 and not args.no_configs and args.lap_solver == "modal_direct":      # collective: every rank takes part
        extra["sharded_configs"] = [sharded_entry(args, name, local_rank, world, dist) for name in SHARDED_CONFIGS]

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
                        f"lap_solver={args.lap_solver}{'' if world == 1 else ' (sharded over ' + str(world) + ' ranks)'}"
                        + ("" if args.lap_solver == "modal_direct" else f", preconditioner={args.preconditioner}"),
            "baseline_config": wl["config"],
            "unknowns": V * (n_time + 1), "state_bytes": 8 * ((n_time + 1) * V + 7 * n_time * V + 6 * (n_time + 1) * F + 36 * n_time * F),
            "device_bytes": dev_bytes, "device_bytes_per_rank": dev_bytes_ranks, "pcg_iterations_per_step": cg_per_it,
            "step_seconds": steps_time_all, "step_seconds_note": steps_note, "commit": git_commit(),
            "world_size": world, "backend": ("rccl (torch.distributed nccl)" if backend == "nccl" else backend) if world > 1 else None,
            "gpus_visible": n_dev,
        },
        "steady_state": {"iterations_per_s": steady_count / steady_el, "ms_per_step": steady_ms,
                         "window": f"iterations {steady_from}..{steady_from + steady_count - 1}"},
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
