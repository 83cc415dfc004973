"""GPU tests of the multi-GPU path on ONE device: the ranks are separate device contexts (separate
time-mode ranges) driven by threads of this process, exchanging through ``ThreadComm``.  The sharded
solve must reproduce the single-context solver (and therefore the reference) decision for decision."""
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


@pytest.mark.parametrize("direct", [False, True])
@pytest.mark.parametrize("n_ranks", [2, 3, 4])
def test_sharded_laplacian_equals_single_context(n_ranks, direct):
    """One ALM iteration through begin / gather / end on n_ranks contexts == dots_step on one context."""
    import torch

    from dots_socp_amd.device import DeviceProblem

    g = golden("ops_torus8x6.npz")
    geom, T = geom_of(g), int(g["n_time"])
    s = O.OracleSolver(T, geom)
    rng = np.random.default_rng(3)
    state = {k: rng.standard_normal(getattr(s, k).shape) for k in O.OracleSolver.STATE}
    single = DeviceProblem(T, geom, lap_solver="modal_pcg")
    devs = [DeviceProblem(T, geom, lap_solver="modal_pcg", mode_shard=(r, n_ranks)) for r in range(n_ranks)]
    for d in [single] + devs:
        for k, v in state.items():
            d.upload(k, v)
        d.set_params(r=1.3, scale_z=2.0, const_d=2.0, cg_tol=1e-12)
        if direct:
            d.setup_frontal(leaf=4)
    single.step(1)
    elems = devs[0].shard_elems()
    sends = [torch.zeros(elems, dtype=torch.float64, device="cuda") for _ in devs]
    for d, snd in zip(devs, sends):
        d.step_begin(snd.data_ptr(), elems)
    recv = torch.cat(sends)
    for d in devs:
        d.step_end(recv.data_ptr(), recv.numel())
    want = single.download_all()
    for d in devs:
        got = d.download_all()
        for k in want:
            scale = max(np.max(np.abs(want[k])), 1e-300)
            assert np.max(np.abs(got[k] - want[k])) < 1e-9 * scale, k
    # replicated state is bit-identical across ranks
    a, b = devs[0].download_all(), devs[-1].download_all()
    assert all(np.array_equal(a[k], b[k]) for k in a)
    for d in [single] + devs:
        d.close()


@pytest.mark.parametrize("fname,n_ranks,mg", [
    ("run_ico2_T15_cong_tol1e-3.npz", 2, "mg"),
    ("run_ico2_T15_cong_tol1e-3.npz", 4, "jacobi"),
    ("run_ico2_T15_cong_tol1e-3.npz", 3, "direct"),
    ("run_torus_T7_tol1e-4.npz", 3, "mg"),
    ("run_torus_T7_tol1e-4.npz", 2, "direct"),
    ("run_refplane4_T8_tol1e-3.npz", 2, "jacobi"),
    ("run_refplane4_T8_tol1e-3.npz", 4, "direct"),
])
def test_sharded_runs_match_reference(fname, n_ranks, mg):
    """Whole solves on n_ranks 'GPUs': every rank stops at the reference's iteration with its cost and KKT."""
    from dots_socp_amd.distributed import solver_socp_sharded

    g = golden(fname)
    kw = {k[3:]: (g[k].tolist() if g[k].ndim else g[k].item()) for k in g.files if k.startswith("kw_")}
    if mg == "direct":
        kw.update(lap_solver="modal_direct")
    else:
        kw.update(lap_solver="modal_pcg", cg_tol=1e-11, preconditioner="multigrid" if mg == "mg" else "jacobi", mg_coarsest=6)

    def rank_main(comm):
        return solver_socp_sharded(int(g["n_time"]), geom_of(g), comm=comm, device=0, **kw)

    results = run_ranks(n_ranks, rank_main)
    want = g["hist_kkt_errors"]
    for sol, hist in results:
        assert int(hist.kkt_iteration[-1]) == int(g["last_iteration"])
        assert np.array_equal(np.isnan(hist.kkt_errors), np.isnan(want))
        m = ~np.isnan(want)
        assert np.allclose(hist.kkt_errors[m], want[m], rtol=1e-6, atol=1e-13)
        assert np.allclose(hist.history["Transportation cost"], g["hist_Transportation_cost"], rtol=1e-6, equal_nan=True)
        assert np.max(np.abs(sol["mu"] - g["sol_mu"])) < 1e-5 * np.max(np.abs(g["sol_mu"]))
    # all ranks hold the same answer bit for bit
    for sol, _ in results[1:]:
        assert np.array_equal(sol["mu"], results[0][0]["mu"])


def test_more_ranks_than_modes():
    """T+1 = 5 modes on 8 ranks: three ranks own no mode and still take part in the exchange."""
    from dots_socp_amd import meshes
    from dots_socp_amd.distributed import mode_partition, solver_socp_sharded
    from dots_socp_amd.socp import solver_socp

    geom, _ = meshes.example("sphere", level=1)
    stride, parts = mode_partition(5, 8)
    assert stride == 1 and [c for _, c in parts] == [1, 1, 1, 1, 1, 0, 0, 0]
    ref_sol, ref_hist = solver_socp(4, geom, nit=25, tol=1e-12)
    results = run_ranks(8, lambda comm: solver_socp_sharded(4, geom, comm=comm, nit=25, tol=1e-12))
    for sol, hist in results:
        assert np.allclose(hist.history["Transportation cost"], ref_hist.history["Transportation cost"], rtol=1e-8, equal_nan=True)
        assert np.max(np.abs(sol["mu"] - ref_sol["mu"])) < 1e-7 * np.max(np.abs(ref_sol["mu"]))
