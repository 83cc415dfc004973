"""CPU tests of the host side of the direct solver (dots_socp_amd/frontal.py): the nested-dissection tree,
the symbolic structure and the batched multifrontal factor, checked by running the device's sweeps in numpy
(tests/frontal_cpu.py) against scipy's sparse solve."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spl

import frontal_cpu
from dots_socp_amd import frontal, geometry, meshes


def problem(name, reorder, **kw):
    if name == "refplane":
        v, t = meshes.plane(kw.get("n", 12))
        g, _ = meshes.make_geometry(v, t, np.ones(v.shape[0]) / v.shape[0], np.ones(v.shape[0]) / v.shape[0])
    else:
        g, _ = meshes.example(name, **kw)
    plan = geometry.build_plan(7, g, reorder=reorder, nd_leaf=kw.get("leaf", 8))
    K = sp.csr_matrix((plan.lap_val, plan.lap_col, plan.lap_rowptr), shape=(plan.n_vertices,) * 2)
    return plan, K


CASES = [("sphere", dict(level=3)), ("refplane", dict(n=12)), ("torus", dict(nu=24, nv=16)), ("knot", dict(nu=60, nv=8))]


@pytest.mark.parametrize("name,kw", CASES)
@pytest.mark.parametrize("reorder", [True, "nd", "python"])
def test_tree_and_structure(name, kw, reorder):
    """The C++ dissection (csrc/dissect.hip: the default, also behind reorder="nd") and the numpy reference."""
    plan, K = problem(name, reorder == "nd" and "nd" or True, **kw)
    V = plan.n_vertices
    diss = plan.dissection if reorder == "nd" else frontal.nested_dissection(K.indptr, K.indices, plan.vertices, leaf=8,
                                                                             native=reorder != "python")
    if reorder == "nd":      # the device numbering IS the order in which the sweeps walk the vertices
        assert np.array_equal(frontal.sweep_order(diss, diss.bands), np.arange(V))
    assert np.array_equal(np.sort(diss.order), np.arange(V))
    assert diss.parent[-1] == -1 and np.all(diss.parent[:-1] > np.arange(diss.n_nodes - 1))
    # separator property: no edge joins the two subtrees of a node
    owner = np.empty(V, dtype=np.int64)
    for p in range(diss.n_nodes):
        owner[diss.order[diss.sep_ptr[p]:diss.sep_ptr[p + 1]]] = p
    anc = [set() for _ in range(diss.n_nodes)]
    for p in range(diss.n_nodes - 1, -1, -1):
        if diss.parent[p] >= 0:
            anc[p] = anc[diss.parent[p]] | {int(diss.parent[p])}
    coo = K.tocoo()
    for i, j in zip(coo.row, coo.col):
        a, b = int(owner[i]), int(owner[j])
        assert a == b or a in anc[b] or b in anc[a]
    bds, pos = frontal.symbolic(diss, K.indptr.astype(np.int64), K.indices.astype(np.int64))
    for p in range(diss.n_nodes):
        assert all(int(owner[v]) in anc[p] for v in bds[p])


@pytest.mark.parametrize("name,kw", CASES)
@pytest.mark.parametrize("reorder", [False, "nd"])
def test_factor_solves_every_mode(name, kw, reorder):
    plan, K = problem(name, reorder, **kw)
    diss = plan.dissection if reorder == "nd" else frontal.nested_dissection(K.indptr, K.indices, plan.vertices, leaf=5)
    shifts = plan.time_eigs + 0.0
    ff = frontal.factorize(K, plan.mass_vert, shifts, diss, pitch=8, workers=2)
    assert ff.values.shape[1] == 8 and ff.grounded.tolist() == [0]
    assert ff.stats["levels"] >= 3 and ff.level_ptr[-1] == ff.node_n.size
    rng = np.random.default_rng(3)
    b = rng.standard_normal((plan.n_vertices, shifts.size))
    b[:, 0] -= b[:, 0].mean()                         # the singular mode needs a right-hand side in the range
    x = frontal_cpu.solve(ff, b)
    M = sp.diags(plan.mass_vert)
    for a in range(shifts.size):
        A = (K + shifts[a] * M).tocsr()
        assert np.max(np.abs(A @ x[:, a] - b[:, a])) < 1e-10 * np.max(np.abs(b[:, a])), a
    # a regular mode agrees with SuperLU
    xs = spl.spsolve((K + shifts[3] * M).tocsc(), b[:, 3])
    assert np.max(np.abs(xs - x[:, 3])) < 1e-10 * np.max(np.abs(xs))
    # eps > 0: nothing is grounded
    ff2 = frontal.factorize(K, plan.mass_vert, shifts[:2] + 0.5, diss, workers=1)
    assert ff2.grounded.size == 0
    x2 = frontal_cpu.solve(ff2, b[:, :2])
    assert np.max(np.abs((K + 0.5 * M) @ x2[:, 0] - b[:, 0])) < 1e-10 * np.max(np.abs(b[:, 0]))


@pytest.mark.parametrize("name,kw", CASES)
def test_native_symbolic_equals_python_symbolic(name, kw):
    plan, K = problem(name, "nd", **kw)
    shifts = plan.time_eigs[:2] + 0.3
    a = frontal.factorize(K, plan.mass_vert, shifts, plan.dissection, numeric=False)     # C++ symbolic phase
    b = frontal.factorize(K, plan.mass_vert, shifts, plan.dissection, numeric=True)      # numpy
    for key in ("node_n", "node_b", "node_foff", "node_ioff", "node_uoff", "node_child", "front_idx", "pull0", "pull1", "level_ptr", "level_nodes"):
        assert np.array_equal(getattr(a, key), getattr(b, key)), key
    assert a.values is None and b.values is not None and a.update_rows == b.update_rows


def test_native_dissection_is_at_least_as_good_as_the_reference_one():
    g, _ = meshes.example("sphere", level=4)
    plan = geometry.build_plan(7, g, reorder=False)
    K = sp.csr_matrix((plan.lap_val, plan.lap_col, plan.lap_rowptr), shape=(plan.n_vertices,) * 2)
    size = {}
    for native in (True, False):
        d = frontal.nested_dissection(K.indptr, K.indices, plan.vertices, leaf=16, native=native)
        size[native] = frontal.factorize(K, plan.mass_vert, plan.time_eigs[:1] + 1.0, d, numeric=False).stats["factor_entries_per_mode"]
    assert size[True] <= 1.05 * size[False]


@pytest.mark.parametrize("name,kw", CASES)
@pytest.mark.parametrize("cuts", ["pairs", "triples", "all", [0, 2, 3]])
def test_merged_heights_solve_the_same_systems(name, kw, cuts):
    """The merged form of the sweeps (dots_front_desc.band_ptr, csrc/kernels_merge.hip) restated in numpy: the blocks of a
    band of tree heights are combined algebraically, the solution is the unmerged one to rounding."""
    plan, K = problem(name, "nd", **kw)
    shifts = plan.time_eigs[:3] + 0.0
    ff = frontal.factorize(K, plan.mass_vert, shifts, plan.dissection, workers=1)
    H = ff.level_ptr.size - 1
    if cuts == "pairs":
        cuts = [0] + list(range(1, H, 2)) + [H]
    elif cuts == "triples":
        cuts = sorted(set([0, 1] + list(range(H, 1, -3))))
    elif cuts == "all":
        cuts = [0, H]
    else:
        cuts = sorted(set(cuts + [H]))
    rng = np.random.default_rng(5)
    b = rng.standard_normal((plan.n_vertices, shifts.size))
    b[:, 0] -= b[:, 0].mean()
    x0 = frontal_cpu.solve(ff, b)
    bands = frontal_cpu.merge(ff, cuts)
    assert sum(g["sep"].size for band in bands for g in band) == plan.n_vertices
    x1 = frontal_cpu.solve_merged(ff, bands, b)
    assert np.max(np.abs(x1 - x0)) < 1e-10 * np.max(np.abs(x0))
    x2 = frontal_cpu.solve_merged(ff, bands, b, top_inverse=True)      # the top band as one explicit inverse
    assert np.max(np.abs(x2 - x0)) < 1e-10 * np.max(np.abs(x0))


@pytest.mark.parametrize("name,kw", CASES)
def test_band_planner(name, kw, monkeypatch):
    """frontal.plan_bands: valid cuts for every specification, the same answer on every call (the ranks of a sharded run
    must cut alike), and the numbering of the plan is the order in which the merged sweeps walk the vertices."""
    plan, K = problem(name, "nd", **kw)
    d = plan.dissection
    H = int(d.height.max()) + 1
    n = np.diff(d.sep_ptr)
    b = frontal.symbolic_native(d, K.indptr, K.indices)[0]
    for spec in ("auto", "off", ",".join(str(x) for x in sorted(set(range(0, H, 2)) | {H}))):
        for top in ("auto", "0", "1"):
            cuts, inv = frontal.plan_bands(d, n, b, 8, spec=spec, top_spec=top)
            assert cuts[0] == 0 and cuts[-1] == H and np.all(np.diff(cuts) >= 1) and np.all(np.diff(cuts) <= 4)
            assert (top != "0" or not inv) and (top != "1" or inv or spec == "auto")
            again = frontal.plan_bands(d, n, b, 8, spec=spec, top_spec=top)
            assert np.array_equal(cuts, again[0]) and inv == again[1]
    with pytest.raises(ValueError):
        frontal.plan_bands(d, n, b, 8, spec="0,1")                  # does not reach the last height
    # what one sweep reads: merging never reads fewer factor entries than one launch per height, and a band of one height
    # reads exactly its nodes' blocks
    e_unit = sum(frontal.band_entries(d, n, b, l, l + 1)[0] for l in range(H))
    assert e_unit == int((n.astype(np.int64) * (n + 1) // 2 + b.astype(np.int64) * n).sum())
    for lo in range(0, H - 1):
        hi = min(H, lo + 3)
        assert frontal.band_entries(d, n, b, lo, hi)[0] >= sum(frontal.band_entries(d, n, b, l, l + 1)[0] for l in range(lo, hi)) - \
            sum(int(b[p]) * int(n[p]) for p in np.flatnonzero((d.height >= lo) & (d.height < hi)))
    monkeypatch.setenv("DOTS_FRONT_BANDS", "off")
    plan_off, _ = problem(name, "nd", **kw)
    assert np.array_equal(plan_off.dissection.order, np.arange(plan.n_vertices))      # one launch per height: the elimination order
    assert np.array_equal(frontal.sweep_order(d, d.bands), np.arange(plan.n_vertices))
