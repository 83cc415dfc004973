"""One rank that owns all nodes and modes runs the TIME-SLAB iteration (dots_slab_stage 0-3, three exchanges through ThreadComm):
what the slab formulation itself costs on one GPU.  usage: python profiles/tools/slab_profile.py [workload] [iterations]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

torch.cuda.init()
from dots_socp_amd import meshes  # noqa: E402
from dots_socp_amd.distributed import ShardedAlmSolver, ThreadComm  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "torus100k"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 150
wl = {"sphere10k": ("sphere", dict(level=5), 31), "torus100k": ("torus", dict(nu=400, nv=250), 31), "knot": ("knot", {}, 31),
      "torus65k_T127": ("torus", dict(nu=360, nv=180), 127)}[name]
geom, _ = meshes.example(wl[0], **wl[1])
alm = ShardedAlmSolver(wl[2], geom, comm=ThreadComm.group(1)[0], nit=steps + 40, tol=1e-30, time_limit=float("inf"))
for _ in range(30):
    alm.iterate()
alm.dev.sync()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    alm.iterate()
alm.dev.sync()
torch.cuda.synchronize()
print(f"{name}: slab iteration on one rank {(time.perf_counter() - t0) / steps * 1e3:.4f} ms", file=sys.stderr)
alm.close()
