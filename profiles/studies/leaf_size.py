"""Tree heights, bands and factor bytes per sweep by leaf size, with the leaves as local inverses (CPU only): python profiles/studies/leaf_size.py <workload>"""
import sys, numpy as np
sys.path.insert(0, '/root/repo')
from dots_socp_amd import meshes, frontal, geometry
ex, kw, T = {"torus100k": ("torus", dict(nu=400, nv=250), 31), "torus65k_T127": ("torus", dict(nu=360, nv=180), 127), "sphere10k": ("sphere", dict(level=5), 31), "knot": ("knot", {}, 31)}[sys.argv[1]]
geom, _ = meshes.example(ex, **kw)
V = np.asarray(geom["vertices"]); tri = np.asarray(geom["triangles"]).astype(np.int64)
K0 = geometry.mesh_adjacency(V.shape[0], tri)
pitch = max(8, 1 << int(np.ceil(np.log2(T + 1))))
for leaf in (12, 16, 20, 24, 32, 40, 48):
    diss = frontal.nested_dissection(K0.indptr, K0.indices, V, leaf=leaf)
    n = np.diff(diss.sep_ptr).astype(np.int64)
    b = np.asarray(frontal.symbolic_native(diss, K0.indptr, K0.indices)[0], dtype=np.int64)
    bands, top = frontal.plan_bands(diss, n, b, pitch)
    unit = pitch * 8
    tot = 0; per = []
    for lo, hi in zip(bands[:-1], bands[1:]):
        e = frontal.band_entries(diss, n, b, lo, hi, leaf_inverse=(bands[1] == 1))[0]
        tot += e * unit; per.append(round(e * unit / 1e6))
    H = int(diss.height.max()) + 1
    lf = diss.height == 0
    print(f"leaf {leaf}: heights {H} bands {len(bands)-1} top_inv {top} per sweep {tot/1e6:.0f} MB; leaves {lf.sum()} n mean {n[lf].mean():.1f} max {n[lf].max()}; bands MB {per}")
