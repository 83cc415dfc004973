"""solve time of the direct solver for many band cuts: python scratch/sweep_bands.py <workload> [max parts]"""
import itertools, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import dots_socp_amd
from dots_socp_amd import meshes
from dots_socp_amd.socp.solver_socp import AlmSolver

name = sys.argv[1]
bench.WORKLOADS["torus30k"] = dict(example="torus", kw=dict(nu=200, nv=150), n_time=31, congestion=0.0, tol=1e-3, config=None)
bench.WORKLOADS["sphere40k"] = dict(example="sphere", kw=dict(level=6), n_time=31, congestion=0.0, tol=1e-3, config=None)
bench.WORKLOADS["sphere2k"] = dict(example="sphere", kw=dict(level=4), n_time=31, congestion=0.0, tol=1e-3, config=None)
wl = bench.WORKLOADS[name]
geom, _ = meshes.example(wl["example"], **wl["kw"])

def run(spec, top="0"):
    os.environ["DOTS_FRONT_BANDS"] = spec
    os.environ["DOTS_FRONT_TOPINV"] = top
    alm = AlmSolver(wl["n_time"], geom, congestion=wl["congestion"], nit=10, tol=1e-30, lap_solver="modal_direct", time_limit=float("inf"))
    for _ in range(3):
        alm.iterate()
    ms, _b = alm.dev.bench_kernel(which=3, reps=200)
    fs = alm.front_summary
    out = (ms, fs["levels"], fs["bytes_per_solve_as_installed"] / 1e6, fs["bands"])
    alm.close()
    return out

ms0, H, mb0, _ = run("off")
print(f"{name} heights {H}: off {ms0*1e3:.1f} us {mb0:.0f} MB", flush=True)
def comps(total, parts):
    if total == 0:
        yield ()
        return
    for p in parts:
        if p <= total:
            for rest in comps(total - p, parts):
                yield (p,) + rest
maxparts = int(sys.argv[2]) if len(sys.argv) > 2 else 5
res = []
for c in comps(H, (1, 2, 3, 4)):
    if len(c) > maxparts or len(c) < 3:
        continue
    cuts = np.concatenate([[0], np.cumsum(c)])
    spec = ",".join(str(int(x)) for x in cuts)
    for top in ("0", "1"):
        try:
            ms, _, mb, bands = run(spec, top)
        except Exception as e:
            print(spec, top, "failed", e, flush=True)
            continue
        res.append((ms, spec + ("+inv" if top == "1" else ""), mb, bands))
res.sort()
for ms, spec, mb, bands in res[:25]:
    print(f"{ms*1e3:7.1f} us  {mb:7.0f} MB  asked {spec}  got {bands}", flush=True)
print("worst", res[-1][:3])
import json
json.dump({"workload": name, "with_top_inverse": True, "off_ms": ms0, "heights": H, "results": [(ms, spec, mb) for ms, spec, mb, _ in res]}, open(f"gpurun_out/sweep_inv_{name}.json", "w"))
