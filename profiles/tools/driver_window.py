import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
torch.cuda.init()
from dots_socp_amd import meshes
from dots_socp_amd.socp.solver_socp import AlmSolver
geom, _ = meshes.example("knot")
first = int(sys.argv[1]); every = int(sys.argv[2])
vals = []
for rep in range(7):
    alm = AlmSolver(31, geom, nit=1000, tol=1e-30, time_limit=1e9)
    alm.step_timers.first, alm.step_timers.every = first, every
    for _ in range(5):
        alm.iterate()
    alm.dev.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        alm.iterate()
    alm.dev.sync(); torch.cuda.synchronize()
    vals.append(20 / (time.perf_counter() - t0))
    alm.close()
vals.sort()
print(f"first={first} every={every}: driver window it/s median {vals[3]:.0f} (min {vals[0]:.0f}, max {vals[-1]:.0f})")
