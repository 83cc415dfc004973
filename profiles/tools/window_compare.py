"""The driver's window (iterations 5..24 of `knot`, --steps 20 --warmup 5) and the cost of the three kinds of iteration, for one source
tree: usage  python profiles/tools/window_compare.py <tree root> [workload]   (profiles/tools/window_compare.sh runs it on several
trees on ONE box: the driver's record moved between rounds while the sweeps did not)."""
import os, sys, time
root = os.path.abspath(sys.argv[1])
sys.path.insert(0, root)
import numpy as np
import torch
torch.cuda.init()
from dots_socp_amd import meshes
from dots_socp_amd.socp.solver_socp import AlmSolver
wl = sys.argv[2] if len(sys.argv) > 2 else "knot"
geom, _ = meshes.example(wl) if wl == "knot" else meshes.example("sphere", level=5)
win, steady = [], []
for rep in range(7):
    alm = AlmSolver(31, geom, nit=1000, tol=1e-30, time_limit=1e9)
    for _ in range(5):
        alm.iterate()
    alm.dev.sync()
    t0 = time.perf_counter()
    for _ in range(20):
        alm.iterate()
    alm.dev.sync()
    win.append(20 / (time.perf_counter() - t0))
    for _ in range(155):
        alm.iterate()
    alm.dev.sync()
    t0 = time.perf_counter()
    for _ in range(200):
        alm.iterate()
    alm.dev.sync()
    steady.append(200 / (time.perf_counter() - t0))
    alm.close()
# kinds of iteration, host-synchronised (each iteration alone on the device: kernels + launch latencies + the host's own work)
alm = AlmSolver(31, geom, nit=1000, tol=1e-30, time_limit=1e9)
kinds = {}
host = {}
for i in range(120):
    adj = alm.adjust_params.peek_adjust(alm.counter_main + 1)
    val = alm.kkt_validator.will_validate_next()
    kind = "penalty update" if adj else ("validation" if val else "quiet")
    alm.dev.sync()
    t0 = time.perf_counter()
    alm.iterate()
    t1 = time.perf_counter()
    alm.dev.sync()
    t2 = time.perf_counter()
    if i >= 30:
        kinds.setdefault(kind, []).append((t2 - t0) * 1e6)
        host.setdefault(kind, []).append((t1 - t0) * 1e6)
alm.close()
win.sort(); steady.sort()
print(f"{root}: {wl} window {win[3]:.0f} it/s (min {win[0]:.0f}, max {win[-1]:.0f}); steady {steady[3]:.0f} (min {steady[0]:.0f}, max {steady[-1]:.0f}); "
      + "; ".join(f"{k}: {np.median(v):.0f} us synchronised, host returns after {np.median(host[k]):.0f} us (n={len(v)})" for k, v in sorted(kinds.items())))
