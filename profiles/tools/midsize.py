"""Steady ALM iterations/s and solve time of a torus of nu x nv vertices at T time intervals (sizes between the bench workloads):
    python profiles/tools/midsize.py <nu> <nv> <T>      (on the GPU box; DOTS_* switches from the environment)"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
torch.cuda.init()
from dots_socp_amd import meshes
from dots_socp_amd.socp.solver_socp import AlmSolver
nu, nv, T = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
geom, _ = meshes.example("torus", nu=nu, nv=nv)
alm = AlmSolver(T, geom, nit=700, tol=1e-30, time_limit=1e9)
for _ in range(150):
    alm.iterate()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 400
for _ in range(n):
    alm.iterate()
alm.dev.synchronize() if hasattr(alm.dev, "synchronize") else torch.cuda.synchronize()
torch.cuda.synchronize()
el = time.perf_counter() - t0
ms, nbytes = alm.dev.bench_kernel(which=3, reps=50)
s = alm.dev.front_summary
print(f"torus {nu}x{nv} V={nu*nv} T={T} env={ {k:v for k,v in os.environ.items() if k.startswith('DOTS_')} }: {n/el:.1f} it/s, solve {ms*1e3:.1f} us, {nbytes/ms/1e9:.2f} TB/s algorithmic, bands {s['bands']}, top_inverse {s['top_inverse']}, leaf_inverse {s.get('leaf_inverse')}", flush=True)
alm.close()
