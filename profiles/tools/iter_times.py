import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dots_socp_amd import meshes
from dots_socp_amd.socp.solver_socp import AlmSolver
geom,_ = meshes.example("knot")
alm = AlmSolver(31, geom, nit=400, tol=1e-30, time_limit=float("inf"))
ts=[]
for i in range(60):
    alm.dev.sync(); t0=time.perf_counter(); alm.iterate(); alm.dev.sync(); ts.append((time.perf_counter()-t0)*1e6)
print("per-iteration us (synchronised):", " ".join(f"{i}:{t:.0f}" for i,t in enumerate(ts)))
