"""CPU test of the N > 1 path: two / three processes, torch.distributed with the gloo backend, the real sharded
driver (``ShardedAlmSolver`` + ``TorchComm``: stage order, neighbour exchanges, the two all-gathers, the all-reduced KKT
sums) on top of the numpy stand-in of a time-slab device context (tests/fake_device.py), which checks every payload it
receives against what the neighbour must hold.  Checks the slab partition, that all ranks take identical decisions,
and that the result equals the reference's recorded run."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import GOLDEN_DIR, ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, fname, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist

    from dots_socp_amd.distributed import ShardedAlmSolver, TorchComm
    from fake_device import FakeDeviceProblem

    # the driver under test stays the product's own; only the device API underneath is the CPU stand-in
    # (the package re-exports the function `solver_socp`, so the module is taken from sys.modules)
    sys.modules["dots_socp_amd.socp.solver_socp"].DeviceProblem = FakeDeviceProblem
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = np.load(os.path.join(GOLDEN_DIR, fname))
        geom = dict(vertices=g["vertices"], triangles=g["triangles"], mu0=g["mu0"], mu1=g["mu1"])
        kw = {k[3:]: (g[k].tolist() if g[k].ndim else g[k].item()) for k in g.files if k.startswith("kw_")}
        comm = TorchComm()
        assert comm.size == world and comm.backend == "gloo"
        nit = kw.pop("nit")
        alm = ShardedAlmSolver(int(g["n_time"]), geom, comm=comm, buffer_device="cpu", nit=nit, **kw)
        for _ in range(nit):
            if alm.iterate():
                break
        sol, hist = alm.finalize()
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), kkt=hist.kkt_errors, it=hist.kkt_iteration,
                 cost=hist.history["Transportation cost"], mu=sol["mu"], beta_mid=sol["beta_mid"],
                 checked=np.array([alm.dev.checked[k] for k in sorted(alm.dev.checked)]),
                 calls=np.array([comm.calls[k] for k in sorted(comm.calls)]), nbytes=np.array([comm.bytes[k] for k in sorted(comm.bytes)]),
                 iterations=np.array(alm.counter_main + 1), pitch=np.array(alm.dev.pitch), V=np.array(alm.dev.V), F=np.array(alm.dev.F),
                 stride=np.array(alm.dev.stride), steps_time=np.array(sum(hist.steps_time.values())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("fname,world", [("run_refplane4_T8_tol1e-3.npz", 2), ("run_torus_T5_cong_k15_steps.npz", 3),
                                         ("run_ico1_T6_palm_k12_steps.npz", 2), ("run_ico1_T6_cscale_k30_steps.npz", 2)])
def test_gloo_run_matches_reference(fname, world, tmp_path):
    import torch.multiprocessing as mp

    mp.spawn(_worker, args=(world, _free_port(), fname, str(tmp_path)), nprocs=world, join=True)
    g = np.load(os.path.join(GOLDEN_DIR, fname))
    ranks = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    r0 = ranks[0]
    for r in ranks[1:]:
        for k in ("kkt", "it", "cost", "mu", "beta_mid"):
            assert np.array_equal(r0[k], r[k], equal_nan=True), f"ranks disagree on {k}"
    want = g["hist_kkt_errors"]
    assert int(r0["it"][-1]) == int(g["last_iteration"])
    assert np.array_equal(np.isnan(r0["kkt"]), np.isnan(want))
    m = ~np.isnan(want)
    assert np.allclose(r0["kkt"][m], want[m], rtol=1e-6, atol=1e-13)
    assert np.allclose(r0["cost"], g["hist_Transportation_cost"], rtol=1e-6, equal_nan=True)
    assert np.max(np.abs(r0["mu"] - g["sol_mu"])) < 1e-6 * np.max(np.abs(g["sol_mu"]))
    assert np.max(np.abs(r0["beta_mid"] - g["sol_beta_mid"])) < 1e-6 * np.max(np.abs(g["sol_beta_mid"]))   # assembled from the slabs
    # the wiring was exercised: every received payload was checked on the ranks that have the neighbour
    n_it = int(r0["iterations"])
    names = sorted(["recv_x", "recv_nsq", "lamc_lo", "b", "x", "recv_mu", "recv_b"])
    for rank, r in enumerate(ranks):
        checked = dict(zip(names, r["checked"].tolist()))
        kinds = sorted(["all_gather", "exchange", "all_reduce", "flag", "gather_array"])
        calls, nbytes = dict(zip(kinds, r["calls"].tolist())), dict(zip(kinds, r["nbytes"].tolist()))
        assert checked["b"] == checked["x"] == n_it and calls["all_gather"] == 2 * n_it
        assert checked["recv_x"] == (n_it if rank > 0 else 0) and checked["lamc_lo"] == (n_it if rank > 0 else 0)
        assert checked["recv_nsq"] == (n_it if rank + 1 < world else 0)
        assert (checked["recv_mu"] > 0) == (rank > 0) and (checked["recv_b"] > 0) == (rank + 1 < world)
        assert calls["all_reduce"] > 0
        # bytes per collective (what this rank hands over): the two all-gathers carry one slab chunk each; an all-reduce carries
        # the 24 KKT sums, the 3 objective sums or the 12 norms of is_constant_scaling -- never a state array; the solution
        # is assembled by all-gathers of the slabs (1 / R of each array per rank, padded to the slab length)
        V, F, pitch, stride, T = int(r["V"]), int(r["F"]), int(r["pitch"]), int(r["stride"]), int(g["n_time"])
        assert nbytes["all_gather"] == 8 * n_it * ((V * pitch + V) + V * pitch)
        assert nbytes["all_reduce"] <= 8 * 24 * calls["all_reduce"] and nbytes["all_reduce"] >= 8 * 3 * calls["all_reduce"]
        sends = (1 if rank + 1 < world else 0) + (1 if rank > 0 else 0)
        kkt_exchanges = calls["exchange"] - n_it
        assert nbytes["exchange"] == 8 * (n_it * sends * V + kkt_exchanges * ((V if rank + 1 < world else 0) + (3 * F if rank > 0 else 0)))
        full = 8 * ((T + 1) * V + 7 * T * V + 6 * (T + 1) * F + 36 * T * F)
        assert calls["gather_array"] == 12 and nbytes["gather_array"] == 8 * stride * (8 * V + 6 * F + 36 * F)
        assert nbytes["gather_array"] < 1.35 * full / world + 8 * (8 * V + 42 * F)
        assert float(r["steps_time"]) > 0.0


def test_slab_partition():
    from dots_socp_amd.distributed import mode_partition, slab_partition

    assert mode_partition is slab_partition

    assert mode_partition(32, 8) == (4, [(4 * r, 4) for r in range(8)])
    assert mode_partition(32, 1) == (32, [(0, 32)])
    stride, parts = mode_partition(9, 4)
    assert stride == 3 and parts == [(0, 3), (3, 3), (6, 3), (9, 0)]
    stride, parts = mode_partition(128, 8)
    assert stride == 16 and sum(c for _, c in parts) == 128
