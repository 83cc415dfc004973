"""GPU tests of the multi-GPU path on ONE device: the ranks are separate device contexts (time slabs of the state,
time-mode ranges of the solve) driven by threads of this process, exchanging through ``ThreadComm``.  The sharded
solve must reproduce the single-context solver to rounding (and therefore the reference decision for decision) and be
bit-identical across its ranks."""
import os
import threading

import numpy as np
import pytest

from conftest import GOLDEN_DIR, load_oracle

pytestmark = pytest.mark.gpu
O = load_oracle()


def golden(name):
    return np.load(os.path.join(GOLDEN_DIR, name))


def geom_of(g):
    return dict(vertices=g["vertices"], triangles=g["triangles"], mu0=g["mu0"], mu1=g["mu1"])


def run_ranks(n_ranks, fn):
    """Run fn(comm) on n_ranks threads; returns the list of results (exceptions are re-raised)."""
    from dots_socp_amd.distributed import ThreadComm

    comms = ThreadComm.group(n_ranks)
    out, err = [None] * n_ranks, [None] * n_ranks

    def work(r):
        try:
            out[r] = fn(comms[r])
        except BaseException as e:   # noqa: BLE001 - reported below
            err[r] = e
            comms[r].shared.barrier.abort()

    threads = [threading.Thread(target=work, args=(r,)) for r in range(n_ranks)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for e in err:
        if e is not None and not isinstance(e, threading.BrokenBarrierError):
            raise e
    for e in err:
        if e is not None:
            raise e
    return out


class SlabGroup:
    """n time-slab contexts on one GPU driven from ONE thread: the exchanges are plain tensor copies."""

    def __init__(self, T, geom, n_ranks, **kw):
        import torch

        from dots_socp_amd.device import DeviceProblem

        self.torch, self.n = torch, n_ranks
        self.devs = [DeviceProblem(T, geom, lap_solver="modal_pcg", time_slab=(r, n_ranks), **kw) for r in range(n_ranks)]
        self.bufs = []
        for d in self.devs:
            nv, nf, nb, nx = (d.slab_elems(k) for k in ("vertex_halo", "triangle_halo", "b_chunk", "x_chunk"))
            z = lambda n: torch.zeros(n, dtype=torch.float64, device="cuda")       # noqa: E731
            b = {"send_x": z(nv), "send_nsq": z(nv), "recv_x": z(nv), "recv_nsq": z(nv), "b_send": z(nb), "b_recv": z(nb * n_ranks),
                 "x_send": z(nx), "x_recv": z(nx * n_ranks), "send_mu": z(nv), "send_b": z(nf), "recv_mu": z(nv), "recv_b": z(nf)}
            d.slab_set_buffers(**{k: t.data_ptr() for k, t in b.items()})
            self.bufs.append(b)
        self.active = self.devs[0].active_ranks

    def each(self, fn):
        return [fn(d) for d in self.devs]

    def upload(self, state):
        for d in self.devs:
            if d.nl:
                for k, v in state.items():
                    d.upload(k, d.to_slab(k, v))

    def neighbours(self, fwd, bwd):
        for d in self.devs:
            d.sync()
        for r in range(self.active):
            if r > 0:
                self.bufs[r][fwd[1]].copy_(self.bufs[r - 1][fwd[0]])
            if r + 1 < self.active:
                self.bufs[r][bwd[1]].copy_(self.bufs[r + 1][bwd[0]])
        self.torch.cuda.synchronize()

    def gather(self, send, recv):
        for d in self.devs:
            d.sync()
        allv = self.torch.cat([b[send] for b in self.bufs])
        for b in self.bufs:
            b[recv].copy_(allv)
        self.torch.cuda.synchronize()

    def iterate(self, split=None):
        """``split``: stage 1 in its two halves (6: right-hand side, 5: projection) around the all-gather of b; default: every other iteration."""
        self.n_iterations = getattr(self, "n_iterations", 0) + 1
        if split is None:
            split = self.n_iterations % 2 == 0
        self.each(lambda d: d.slab_stage(0))
        self.neighbours(("send_x", "recv_x"), ("send_nsq", "recv_nsq"))
        if split:
            self.each(lambda d: d.slab_stage(6))
            self.gather("b_send", "b_recv")
            self.each(lambda d: d.slab_stage(5))
            with pytest.raises(Exception, match="order"):
                self.devs[0].slab_stage(5)
        else:
            self.each(lambda d: d.slab_stage(1))
            self.gather("b_send", "b_recv")
        self.each(lambda d: d.slab_stage(2))
        self.gather("x_send", "x_recv")
        self.each(lambda d: d.slab_stage(3))

    def kkt(self, conditions):
        self.each(lambda d: d.slab_stage(4))
        self.neighbours(("send_mu", "recv_mu"), ("send_b", "recv_b"))
        total = sum(d.kkt_sums(conditions) for d in self.devs)
        return self.devs[0].kkt_combine(conditions, total)

    def objective(self):
        return self.devs[0].objective_combine(sum(d.objective_sums() for d in self.devs))

    def download(self, name):
        full = np.zeros(self.devs[0].full_shape(name))
        for d in self.devs:
            if d.nl:
                d.from_slab(name, d.download(name), full)
        return full

    def close(self):
        self.each(lambda d: d.close())


@pytest.mark.parametrize("direct", [False, True])
@pytest.mark.parametrize("n_ranks", [2, 3, 4, 6])
def test_slab_iterations_equal_the_single_context(n_ranks, direct):
    """Two ALM iterations + KKT residuals + objective on n time slabs == the same on one context, to rounding: the
    element-wise steps form the same sums in the same order, the sweeps split their dot products by the mode pitch (which
    is smaller on a slab), the PCG stops at its tolerance (and fixes the free constant of phi differently)."""
    from dots_socp_amd.device import DeviceProblem

    g = golden("ops_torus8x6.npz")
    geom, T = geom_of(g), int(g["n_time"])
    s = O.OracleSolver(T, geom)
    rng = np.random.default_rng(3)
    state = {k: rng.standard_normal(getattr(s, k).shape) for k in O.OracleSolver.STATE}
    single = DeviceProblem(T, geom, lap_solver="modal_pcg")
    group = SlabGroup(T, geom, n_ranks)
    for k, v in state.items():
        single.upload(k, v)
    group.upload(state)
    params = dict(r=1.3, scale_z=2.0, const_d=2.0, congestion=0.07, cg_tol=1e-13)
    for d in [single] + group.devs:
        d.set_params(**params)
        if direct and d.nl:
            d.setup_frontal(leaf=4)
    for _ in range(2):
        single.step(1)
        group.iterate()
    tol = 1e-12 if direct else 1e-8
    mass = O.OracleSolver(T, geom).mass_v[None, :]
    for k in state:
        want, got = single.download(k), group.download(k)
        if k == "phi":       # eps = 0: phi is defined up to a constant
            want, got = want - np.sum(want * mass) / (np.sum(mass) * (T + 1)), got - np.sum(got * mass) / (np.sum(mass) * (T + 1))
        scale = max(np.max(np.abs(want)), 1e-300)
        assert np.max(np.abs(got - want)) <= tol * scale, k
    conds = list(range(7))
    want, got = single.kkt(conds), group.kkt(conds)
    for i in conds:
        assert got[i][0] == pytest.approx(want[i][0], rel=1e-10 if direct else 1e-7)
    assert group.objective() == pytest.approx(single.objective(), rel=1e-10 if direct else 1e-7)
    # per-rank device memory: the state is divided, not replicated
    state_bytes = sum(np.prod(single.shape(k)) for k in state) * 8
    stride = -(-(T + 1) // n_ranks)
    pitch = max(4, 1 << (stride - 1).bit_length())
    full_pitch = max(8, 1 << T.bit_length())
    for d in group.devs:
        assert d.device_bytes() < single.device_bytes() * (pitch / full_pitch) * 1.35 + 2e6, (d.device_bytes(), single.device_bytes(), state_bytes)
    single.close()
    group.close()


def test_slab_stage_order_and_stale_halos_are_errors():
    from dots_socp_amd import _lib

    g = golden("ops_ico1.npz")
    group = SlabGroup(int(g["n_time"]), geom_of(g), 2)
    d = group.devs[0]
    with pytest.raises(_lib.HipLibraryError, match="order"):
        d.slab_stage(2)
    with pytest.raises(_lib.HipLibraryError, match="order"):
        d.slab_stage(5)               # the projection alone comes after the right-hand side alone (stage 6)
    with pytest.raises(_lib.HipLibraryError, match="slab"):
        d.step(1)
    with pytest.raises(_lib.HipLibraryError, match="slab"):
        d.kkt([0])
    group.iterate()
    with pytest.raises(_lib.HipLibraryError, match="stale"):
        d.kkt_sums([2])
    d.kkt_sums([0, 1, 3, 6])      # conditions without a time stencil across the slab boundary need no halo
    group.close()


@pytest.mark.parametrize("fname,n_ranks,mg", [
    ("run_ico2_T15_cong_tol1e-3.npz", 2, "mg"),
    ("run_ico2_T15_cong_tol1e-3.npz", 4, "jacobi"),
    ("run_ico2_T15_cong_tol1e-3.npz", 3, "direct"),
    ("run_ico2_T15_cong_tol1e-3.npz", 8, "direct"),
    ("run_ico2_T15_palm_tol1e-3.npz", 4, "direct"),
    ("run_ico2_T15_ckpt_tol1e-3.npz", 2, "direct"),
    ("run_torus_T7_tol1e-4.npz", 3, "mg"),
    ("run_torus_T7_tol1e-4.npz", 2, "direct"),
    ("run_torus_T5_eps_k15_steps.npz", 3, "direct"),
    ("run_refplane4_T8_tol1e-3.npz", 2, "jacobi"),
    ("run_refplane4_T8_tol1e-3.npz", 4, "direct"),
    ("run_ico1_T6_k150_lazy.npz", 7, "direct"),
])
def test_sharded_runs_match_reference(fname, n_ranks, mg):
    """Whole solves on n_ranks 'GPUs': every rank stops at the reference's iteration with its cost and KKT, all ranks
    hold the same answer bit for bit, and it is the single-GPU solver's to rounding."""
    from dots_socp_amd.distributed import solver_socp_sharded
    from dots_socp_amd.socp import solver_socp

    g = golden(fname)
    kw = {k[3:]: (g[k].tolist() if g[k].ndim else g[k].item()) for k in g.files if k.startswith("kw_")}
    if mg == "direct":
        kw.update(lap_solver="modal_direct")
    else:
        kw.update(lap_solver="modal_pcg", cg_tol=1e-11, preconditioner="multigrid" if mg == "mg" else "jacobi", mg_coarsest=6)

    def rank_main(comm):
        return solver_socp_sharded(int(g["n_time"]), geom_of(g), comm=comm, device=0, **{k: (list(v) if isinstance(v, list) else v) for k, v in kw.items()})

    results = run_ranks(n_ranks, rank_main)
    want = g["hist_kkt_errors"]
    for sol, hist in results:
        assert int(hist.kkt_iteration[-1]) == int(g["last_iteration"])
        assert np.array_equal(np.isnan(hist.kkt_errors), np.isnan(want))
        m = ~np.isnan(want)
        assert np.allclose(hist.kkt_errors[m], want[m], rtol=1e-6, atol=1e-13)
        assert np.allclose(hist.history["Transportation cost"], g["hist_Transportation_cost"], rtol=1e-6, equal_nan=True)
        assert np.max(np.abs(sol["mu"] - g["sol_mu"])) < 1e-5 * np.max(np.abs(g["sol_mu"]))
        if "ckpt_iteration" in g.files:
            assert [c["iteration"] for c in sol["checkpoints"]] == g["ckpt_iteration"].tolist()
            assert np.max(np.abs(np.stack([c["mu"] for c in sol["checkpoints"]]) - g["ckpt_mu"])) < 1e-5 * np.max(np.abs(g["ckpt_mu"]))
    # all ranks hold the same answer bit for bit ...
    for sol, _ in results[1:]:
        for k in ("mu", "E", "phi", "beta_mid"):
            assert np.array_equal(sol[k], results[0][0][k]), k
    if mg == "direct":      # ... and it is the single-GPU solver's (the sweeps round differently: their dot products are split by the mode pitch)
        one, one_hist = solver_socp(int(g["n_time"]), geom_of(g), **{k: (list(v) if isinstance(v, list) else v) for k, v in kw.items()})
        assert int(one_hist.kkt_iteration[-1]) == int(results[0][1].kkt_iteration[-1])
        for k in ("mu", "E", "A", "B", "z_fst", "beta_mid", "beta_end"):
            assert np.max(np.abs(one[k] - results[0][0][k])) <= 1e-8 * np.max(np.abs(one[k])), k


def test_more_ranks_than_nodes():
    """T+1 = 5 nodes on 8 ranks: three ranks hold no node / mode and still take part in the collectives."""
    from dots_socp_amd import meshes
    from dots_socp_amd.distributed import slab_partition, solver_socp_sharded
    from dots_socp_amd.socp import solver_socp

    geom, _ = meshes.example("sphere", level=1)
    stride, parts = slab_partition(5, 8)
    assert stride == 1 and [c for _, c in parts] == [1, 1, 1, 1, 1, 0, 0, 0]
    ref_sol, ref_hist = solver_socp(4, geom, nit=25, tol=1e-12)
    results = run_ranks(8, lambda comm: solver_socp_sharded(4, geom, comm=comm, nit=25, tol=1e-12))
    for sol, hist in results:
        assert np.allclose(hist.history["Transportation cost"], ref_hist.history["Transportation cost"], rtol=1e-11, equal_nan=True)
        assert np.max(np.abs(sol["mu"] - ref_sol["mu"])) < 1e-10 * np.max(np.abs(ref_sol["mu"]))


@pytest.mark.parametrize("T,n_ranks", [(127, 4), (63, 3)])
def test_slabs_at_long_time_axes(T, n_ranks):
    """T + 1 = 128 / 64: the tiled transforms of a slab context stage Q in chunks (time pitch >= 128) -- same iterates as
    one context, which runs the matrix-core transforms there."""
    from dots_socp_amd import meshes
    from dots_socp_amd.distributed import solver_socp_sharded
    from dots_socp_amd.socp import solver_socp

    geom, _ = meshes.example("sphere", level=2)
    kw = dict(nit=12, tol=1e-12, congestion=0.02)
    one, one_hist = solver_socp(T, geom, **kw)
    results = run_ranks(n_ranks, lambda comm: solver_socp_sharded(T, geom, comm=comm, **kw))
    for sol, hist in results:
        assert np.allclose(hist.history["Transportation cost"], one_hist.history["Transportation cost"], rtol=1e-9, equal_nan=True)
        for k in ("mu", "phi", "B"):
            assert np.max(np.abs(sol[k] - one[k])) <= 1e-9 * np.max(np.abs(one[k])), k


def test_finite_time_limit_adds_no_collectives_to_quiet_iterations():
    """The wall-clock decision is shared only on iterations that read back anyway: between them an
    iteration costs exactly one neighbour exchange and two all-gathers, whatever time_limit is."""
    from dots_socp_amd import meshes
    from dots_socp_amd.distributed import ShardedAlmSolver

    geom, _ = meshes.example("sphere", level=1)

    def rank_main(comm):
        alm = ShardedAlmSolver(6, geom, comm=comm, nit=400, tol=1e-30, time_limit=1000)
        for _ in range(150):
            alm.iterate()
        quiet = alm.quiet_steps
        calls = dict(comm.calls)
        clock = alm.clock_exchanges
        alm.close()
        return quiet, calls, clock

    for quiet, calls, clock in run_ranks(2, rank_main):
        assert quiet > 90                                   # most iterations only enqueue
        assert calls["all_gather"] == 2 * 150
        assert calls["flag"] == clock == 150 - quiet        # one clock exchange per iteration that reads back
        assert calls["exchange"] >= 150 and calls["exchange"] <= 150 + (150 - quiet)


def test_time_limit_stops_every_rank_on_the_same_iteration():
    from dots_socp_amd import meshes
    from dots_socp_amd.distributed import solver_socp_sharded

    geom, _ = meshes.example("sphere", level=2)
    results = run_ranks(3, lambda comm: solver_socp_sharded(15, geom, comm=comm, nit=100000, tol=1e-30, time_limit=0.3))
    its = [int(h.kkt_iteration[-1]) for _, h in results]
    assert len(set(its)) == 1 and 5 < its[0] < 100000


def _rccl_worker(rank, world, port, fname, out_dir):
    """One rank of a torch.distributed job over RCCL (backend "nccl") driving the real device path."""
    import sys

    from conftest import ROOT

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    dev = rank % torch.cuda.device_count()          # one rank per GPU (a single rank on a one-GPU box)
    torch.cuda.set_device(dev)
    dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
    try:
        from dots_socp_amd.distributed import TorchComm, solver_socp_sharded

        g = golden(fname)
        kw = {k[3:]: (g[k].tolist() if g[k].ndim else g[k].item()) for k in g.files if k.startswith("kw_")}
        comm = TorchComm()
        assert comm.backend == "nccl" and comm.size == world
        sol, hist = solver_socp_sharded(int(g["n_time"]), geom_of(g), comm=comm, device=dev, lap_solver="modal_direct", **kw)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), kkt=hist.kkt_errors, it=hist.kkt_iteration,
                 cost=hist.history["Transportation cost"], mu=sol["mu"], calls=np.array([comm.calls[k] for k in sorted(comm.calls)]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("fname", ["run_ico2_T15_cong_tol1e-3.npz", "run_torus_T7_tol1e-4.npz"])
def test_rccl_backend_single_rank(fname, tmp_path):
    """The sharded driver over RCCL itself (device tensors handed to all_gather_into_tensor / all_reduce, ordering against the
    library's stream): one GPU allows one rank, so the neighbour exchanges stay with the thread and gloo tests; the collectives,
    the buffers they are given and the stream hand-overs are the ones an N-GPU job runs."""
    import socket

    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_rccl_worker, args=(1, port, fname, str(tmp_path)), nprocs=1, join=True)
    g = golden(fname)
    r = np.load(tmp_path / "rank0.npz")
    want = g["hist_kkt_errors"]
    assert int(r["it"][-1]) == int(g["last_iteration"])
    assert np.array_equal(np.isnan(r["kkt"]), np.isnan(want))
    m = ~np.isnan(want)
    assert np.allclose(r["kkt"][m], want[m], rtol=1e-6, atol=1e-13)
    assert np.allclose(r["cost"], g["hist_Transportation_cost"], rtol=1e-6, equal_nan=True)
    assert np.max(np.abs(r["mu"] - g["sol_mu"])) < 1e-5 * np.max(np.abs(g["sol_mu"]))
    calls = dict(zip(sorted(["all_gather", "exchange", "all_reduce", "flag", "gather_array"]), r["calls"].tolist()))
    assert calls["all_gather"] > 0 and calls["all_reduce"] > 0 and calls["gather_array"] == 12


@pytest.mark.parametrize("name,nu,nv,T,n_ranks,nit", [("torus100k", 400, 250, 31, 4, 30), ("torus65k_T127", 360, 180, 127, 8, 30)])
def test_sharded_parity_at_the_sizes_the_configurations_name(name, nu, nv, T, n_ranks, nit):
    """BASELINE configs[3] (~100k vertices, T = 31, 4 ranks: slab pitch 8) and configs[4] (V = 64 800, T = 127, 8 ranks: pitch
    16) sharded at FULL size -- where the 32-bit indices, the chunked-Q tiled transforms and the halo packs see their real
    extents: 30 iterations with every KKT residual and the objective evaluated each iteration, against the single-context
    run (KKT / cost 1e-9, mu 1e-8), and for T = 127 against the first 10 iterations the REFERENCE recorded
    (headline_torus65k_T127.npz, 1e-6); per-rank device memory <= 1.35 x single / R."""
    from dots_socp_amd import meshes
    from dots_socp_amd.distributed import ShardedAlmSolver
    from dots_socp_amd.socp.solver_socp import AlmSolver

    geom, _ = meshes.example("torus", nu=nu, nv=nv)
    kw = dict(nit=nit, tol=1e-5, check_kkt_step_by_step=True, time_limit=1e9)

    def run(alm):
        for _ in range(nit):
            if alm.iterate():
                break
        _, hist = alm.finalize(download=False)
        mu = alm.recovered("mu", alm._download("mu"))
        out = (hist.kkt_errors.copy(), np.array(hist.history["Transportation cost"]), np.array(hist.history["Objective value"]), mu, alm.dev.device_bytes())
        alm.close()
        return out

    kkt1, cost1, obj1, mu1, bytes1 = run(AlmSolver(T, geom, **kw))
    assert kkt1.shape == (nit, 7) and np.all(np.isfinite(kkt1))
    results = run_ranks(n_ranks, lambda comm: run(ShardedAlmSolver(T, geom, comm=comm, **kw)))
    scale = np.max(np.abs(mu1))
    for kkt, cost, obj, mu, nbytes in results:
        assert kkt.shape == kkt1.shape
        assert np.allclose(kkt, kkt1, rtol=1e-9, atol=1e-15)
        assert np.allclose(cost, cost1, rtol=1e-9) and np.allclose(obj, obj1, rtol=1e-9)
        assert np.max(np.abs(mu - mu1)) <= 1e-8 * scale
        assert nbytes <= 1.35 * bytes1 / n_ranks, (nbytes, bytes1)
    for _, _, _, mu, _ in results[1:]:
        assert np.array_equal(mu, results[0][3])          # every rank assembles the same solution bit for bit
    fixture = os.path.join(GOLDEN_DIR, f"headline_{name}.npz")
    g = np.load(fixture)
    if "check_kkt_step_by_step" in g.files and bool(g["check_kkt_step_by_step"]):      # the reference's own first iterations
        n = int(g["nit"])
        chk = np.array([geom["vertices"].sum(), np.abs(geom["vertices"]).sum(), float(geom["triangles"].sum())])
        assert np.allclose(chk, g["vertices_checksum"], rtol=1e-13)
        kkt = results[0][0]
        assert np.allclose(kkt[:n], g["hist_kkt_errors"], rtol=1e-6, atol=1e-13)
        assert np.allclose(results[0][1][:n], g["hist_Transportation_cost"], rtol=1e-6)


def test_constant_scaling_on_time_slabs():
    """is_constant_scaling (solver_socp.py:324-365, :574-586) sharded: every norm it needs is a sum over space-time -- the slabs'
    shares go through one small all-reduce.  Against the reference's recorded run and the single-context solver."""
    from dots_socp_amd.distributed import solver_socp_sharded
    from dots_socp_amd.socp import solver_socp

    g = golden("run_ico1_T6_cscale_k30_steps.npz")
    kw = {k[3:]: (g[k].tolist() if g[k].ndim else g[k].item()) for k in g.files if k.startswith("kw_")}
    assert kw["is_constant_scaling"]
    one, one_hist = solver_socp(int(g["n_time"]), geom_of(g), **kw)
    want = g["hist_kkt_errors"]
    for n_ranks in (2, 3):
        results = run_ranks(n_ranks, lambda comm: solver_socp_sharded(int(g["n_time"]), geom_of(g), comm=comm, **kw))
        for sol, hist in results:
            assert int(hist.kkt_iteration[-1]) == int(g["last_iteration"])
            assert np.array_equal(np.isnan(hist.kkt_errors), np.isnan(want))
            m = ~np.isnan(want)
            assert np.allclose(hist.kkt_errors[m], want[m], rtol=1e-6, atol=1e-13)
            assert np.allclose(hist.history["Transportation cost"], g["hist_Transportation_cost"], rtol=1e-6, equal_nan=True)
            assert np.max(np.abs(sol["mu"] - g["sol_mu"])) < 1e-6 * np.max(np.abs(g["sol_mu"]))
            for k in ("mu", "E", "A", "B", "beta_mid"):
                assert np.max(np.abs(sol[k] - one[k])) <= 1e-8 * np.max(np.abs(one[k])), k


def test_kkt_sums_reach_the_all_reduce_without_a_host_round_trip():
    """The 24 KKT sums of a slab stay on the device between the kernels that form them and the all-reduce (dots_kkt_sums_device);
    the whole arrays are assembled by all-gathers of the slabs.  Counted per collective on two thread-ranks."""
    from dots_socp_amd import meshes
    from dots_socp_amd.distributed import ShardedAlmSolver

    geom, _ = meshes.example("sphere", level=2)
    V, F, T = geom["vertices"].shape[0], geom["triangles"].shape[0], 9

    def rank_main(comm):
        alm = ShardedAlmSolver(T, geom, comm=comm, nit=40, tol=1e-30)
        for _ in range(40):
            alm.iterate()
        host = alm.dev.kkt_sums(list(range(7)))
        alm.dev.kkt_sums_device(list(range(7)), alm.kkt_buf.data_ptr())
        alm.dev.sync()
        dev_sums = alm.kkt_buf.cpu().numpy().copy()
        before = dict(comm.bytes)
        sol, _ = alm.finalize()
        moved = {k: comm.bytes[k] - before[k] for k in before}
        stride = alm.dev.slab[2]
        alm.close()
        return host, dev_sums, moved, stride, sol["mu"]

    out = run_ranks(2, rank_main)
    for host, dev_sums, moved, stride, _ in out:
        assert np.array_equal(host, dev_sums)                                        # same kernels, same sums
        assert moved["gather_array"] == 8 * stride * (8 * V + 42 * F)                # 12 arrays, 1 / R of each (padded to the slab)
        assert moved["all_reduce"] == 8 * (24 + 3)                                   # the final KKT sums + the objective's three
    assert np.array_equal(out[0][4], out[1][4])


def _gpus():
    import torch

    return torch.cuda.device_count()


@pytest.mark.parametrize("fname,world", [("run_ico2_T15_cong_tol1e-3.npz", 2), ("run_torus_T7_tol1e-4.npz", 4), ("run_ico2_T15_palm_tol1e-3.npz", 2)])
def test_rccl_backend_several_ranks(fname, world, tmp_path):
    """The FIRST thing to run on a node with several GPUs: the recorded runs of the reference through the sharded driver
    over RCCL with one rank per GPU -- the neighbour exchanges over xGMI (batch_isend_irecv), the two all-gathers, the all-reduced
    KKT sums, enqueue-only iterations ordered against RCCL's streams -- with the same assertions as the thread and gloo tests:
    every rank stops at the reference's iteration with its KKT values and cost, all ranks hold the same solution bit for bit.
    Skipped on boxes with fewer GPUs than ranks (RCCL refuses two ranks on one device)."""
    if _gpus() < world:
        pytest.skip(f"needs {world} GPUs (one rank per GPU over RCCL); this box has {_gpus()}")
    import socket

    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_rccl_worker, args=(world, port, fname, str(tmp_path)), nprocs=world, join=True)
    g = golden(fname)
    ranks = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    want = g["hist_kkt_errors"]
    for r in ranks:
        assert int(r["it"][-1]) == int(g["last_iteration"])
        assert np.array_equal(np.isnan(r["kkt"]), np.isnan(want))
        m = ~np.isnan(want)
        assert np.allclose(r["kkt"][m], want[m], rtol=1e-6, atol=1e-13)
        assert np.allclose(r["cost"], g["hist_Transportation_cost"], rtol=1e-6, equal_nan=True)
        assert np.max(np.abs(r["mu"] - g["sol_mu"])) < 1e-5 * np.max(np.abs(g["sol_mu"]))
        assert np.array_equal(r["mu"], ranks[0]["mu"])
