"""Factor bytes one sweep reads per band of tree heights (CPU only): python profiles/studies/band_bytes.py <workload>"""
import sys, numpy as np
sys.path.insert(0, '/root/repo')
from dots_socp_amd import meshes, frontal, geometry
ex, kw, T = {"torus100k": ("torus", dict(nu=400, nv=250), 31), "torus65k_T127": ("torus", dict(nu=360, nv=180), 127), "sphere10k": ("sphere", dict(level=5), 31), "knot": ("knot", {}, 31), "torus500k": ("torus", dict(nu=1000, nv=500), 31)}[sys.argv[1]]
geom, _ = meshes.example(ex, **kw)
V = np.asarray(geom["vertices"]); tri = np.asarray(geom["triangles"]).astype(np.int64)
K0 = geometry.mesh_adjacency(V.shape[0], tri)
pitch = max(8, 1 << int(np.ceil(np.log2(T + 1))))
diss = frontal.nested_dissection(K0.indptr, K0.indices, V, leaf=16)
n = np.diff(diss.sep_ptr).astype(np.int64)
b = np.asarray(frontal.symbolic_native(diss, K0.indptr, K0.indices)[0], dtype=np.int64)
bands, top = frontal.plan_bands(diss, n, b, pitch)
print("bands", list(bands), "top inverse", top)
unit = pitch * 8
tot = 0
for lo, hi in zip(bands[:-1], bands[1:]):
    e, rows, vec = frontal.band_entries(diss, n, b, lo, hi)
    inb = (diss.height >= lo) & (diss.height < hi)
    tops = frontal._band_tops(diss, n, lo, hi)
    tot += e * unit
    print(f"band [{lo},{hi}): nodes {inb.sum():6d} rows {rows:7d} factor {e*unit/1e6:8.1f} MB  max n {max(tops):4d} mean n {np.mean(tops):6.1f} max b {b[inb].max():4d} mean b {b[inb].mean():6.1f}")
print("total per sweep", tot / 1e6, "MB")
