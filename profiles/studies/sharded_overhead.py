"""Cost of the time-slab iteration (dots_slab_stage 0-3 with its three exchanges) on ONE GPU: a single rank that owns all nodes and modes, and two
ranks as threads sharing the GPU (their kernels serialise, so only the 1-rank line is a timing; the others check that the
path runs).  usage: python profiles/studies/sharded_overhead.py [workload]"""
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from dots_socp_amd import meshes  # noqa: E402
from dots_socp_amd.distributed import ShardedAlmSolver, ThreadComm  # noqa: E402
from dots_socp_amd.socp.solver_socp import AlmSolver  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "sphere10k"
wl = {"sphere10k": ("sphere", dict(level=5)), "torus100k": ("torus", dict(nu=400, nv=250))}[name]
geom, _ = meshes.example(wl[0], **wl[1])
steps, warm = 150, 30


def timed(alm):
    for _ in range(warm):
        alm.iterate()
    alm.dev.sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        alm.iterate()
    alm.dev.sync()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


alm = AlmSolver(31, geom, nit=steps + warm + 8, tol=1e-30, time_limit=float("inf"))
print(f"{name}: plain step           {timed(alm):.4f} ms")
alm.close()
for n in (1, 2):
    comms = ThreadComm.group(n)
    out = [None] * n

    def run(r):
        a = ShardedAlmSolver(31, geom, comm=comms[r], nit=steps + warm + 8, tol=1e-30, time_limit=float("inf"))
        out[r] = timed(a)
        a.close()

    th = [threading.Thread(target=run, args=(r,)) for r in range(n)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    print(f"{name}: sharded, {n} rank(s) on one GPU  {max(out):.4f} ms")
