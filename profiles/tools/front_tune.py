import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
torch.cuda.init()
from dots_socp_amd import meshes
from dots_socp_amd.socp.solver_socp import AlmSolver
name = sys.argv[1]
W = {"knot": ("knot", {}, 31), "sphere10k": ("sphere", dict(level=5), 31), "knot63": ("knot", {}, 63), "torus100k": ("torus", dict(nu=400, nv=250), 31),
     "torus65k_T127": ("torus", dict(nu=360, nv=180), 127), "plane20": ("plane", dict(n=20), 31)}
ex, kw, T = W[name]
geom, _ = meshes.example(ex, **kw)
print("==", name, file=sys.stderr, flush=True)
alm = AlmSolver(T, geom, nit=50, tol=1e-30, time_limit=1e9)
for _ in range(10):
    alm.iterate()
ms, nbytes = alm.dev.bench_kernel(which=3, reps=100)
print(f"{name}: solve {ms*1e3:.1f} us, {nbytes/ms/1e6:.0f} GB/s algorithmic, launches {alm.dev.front_launches()}", file=sys.stderr, flush=True)
alm.close()
