"""GPU tests of the direct (multifrontal) Laplacian solve on meshes large enough for multi-panel fronts,
1024-thread levels and several workgroups per node: device factorisation against the numpy one and against
the multigrid-PCG solve of the same right-hand side."""
import numpy as np
import pytest

from conftest import has_gpu
from dots_socp_amd import meshes

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs a GPU")]


def rel(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def gauge(phi, mass):
    w = np.broadcast_to(mass[None, :], phi.shape)
    return phi - np.sum(phi * w) / np.sum(w)


def make(geom, T, eps, reorder, seed=3, **kw):
    from dots_socp_amd.device import DeviceProblem

    dev = DeviceProblem(T, geom, lap_solver="modal_pcg", reorder=reorder, **kw)
    rng = np.random.default_rng(seed)
    for name in ("A", "lambda_c", "mu", "B", "E"):
        dev.upload(name, rng.standard_normal(dev.shape(name)))
    dev.set_params(r=1.3, eps=eps, cg_tol=1e-12, cg_max_iter=5000)
    return dev


@pytest.mark.parametrize("mesh,kw,T", [("sphere", dict(level=4), 15), ("torus", dict(nu=72, nv=40), 31), ("knot", dict(nu=240, nv=10), 7)])
@pytest.mark.parametrize("eps", [0.0, 1e-3])
def test_device_factor_matches_host_factor_and_pcg(mesh, kw, T, eps):
    geom, _ = meshes.example(mesh, **kw)
    mass = None
    out = {}
    for tag, reorder, numeric in (("dev_nd", "nd", "device"), ("host_nd", "nd", "host"), ("dev_rcm", True, "device"), ("pcg", True, None)):
        dev = make(geom, T, eps, reorder)
        mass = dev.plan.mass_vert[np.argsort(dev.plan.perm_vert)] if mass is None else mass
        if numeric is None:
            assert dev.setup_multigrid(eps=eps) is not None
        else:
            s = dev.setup_frontal(eps=eps, numeric=numeric)
            assert s["levels"] >= 6 and s["root_rows"] > 32        # several panels at the top of the tree
        st = dev.run_phase("laplacian")
        assert st.cg_not_converged == 0
        phi = dev.download("phi")
        assert np.all(np.isfinite(phi)), tag
        out[tag] = gauge(phi, mass) if eps == 0.0 else phi
        dev.close()
    assert rel(out["dev_nd"], out["host_nd"]) < 1e-10
    assert rel(out["dev_rcm"], out["host_nd"]) < 1e-9
    assert rel(out["dev_nd"], out["pcg"]) < 1e-8


def test_indefinite_operator_is_reported():
    """A non-positive pivot of the device factorisation (kernels_factor.hip: bad_pivot) comes back as
    DOTS_ERR_ARGUMENT with its message, leaves no factor installed and releases the factor's allocations.
    Trigger: a stiffness matrix with a positive diagonal (dots_create accepts it) whose off-diagonal entries are
    tripled, so K + sigma M is indefinite for the low modes."""
    from dots_socp_amd import _lib
    from dots_socp_amd.device import DeviceProblem
    from dots_socp_amd.geometry import build_plan

    geom, _ = meshes.example("sphere", level=3)
    plan = build_plan(7, geom, reorder="nd")
    rows = np.repeat(np.arange(plan.n_vertices), np.diff(plan.lap_rowptr))
    plan.lap_val = np.where(plan.lap_col == rows, plan.lap_val, 3.0 * plan.lap_val)
    dev = DeviceProblem(7, geom, lap_solver="modal_pcg", plan=plan)
    with pytest.raises(_lib.HipLibraryError, match="pivot"):
        dev.setup_frontal(eps=0.0)
    with pytest.raises(_lib.HipLibraryError, match="no factor"):
        dev.enable_frontal(True)
    # the context is still usable with a sound factor source: a second, healthy context on the same process
    dev.close()
    dev = DeviceProblem(7, geom, lap_solver="modal_pcg", reorder="nd")
    assert dev.setup_frontal(eps=0.0)["levels"] >= 3
    dev.close()


def test_direct_solver_is_deterministic_and_exact():
    """Two contexts give bit-identical phi; K phi reproduces the right-hand side to rounding."""
    geom, _ = meshes.example("sphere", level=4)
    got = []
    for _ in range(2):
        dev = make(geom, 15, 1e-2, "nd")
        dev.setup_frontal(eps=1e-2)
        dev.run_phase("laplacian")
        got.append(dev.download("phi"))
        dev.close()
    assert np.array_equal(got[0], got[1])


@pytest.mark.parametrize("T", [63, 127, 20])
def test_time_pitches_64_128_and_ragged(T):
    """T = 63 / 127 (time pitch 64 / 128: BASELINE configs 3 and 5) and T = 20 (pitch 32 with 11 padding columns):
    ten full ALM iterations with the direct solver stay on the oracle's trajectory."""
    from conftest import load_oracle
    from dots_socp_amd.device import DeviceProblem

    O = load_oracle()
    geom, _ = meshes.example("sphere", level=2)
    s = O.OracleSolver(T, geom, congestion=0.02)
    s.scale_z(2.0)
    dev = DeviceProblem(T, geom, lap_solver="modal_pcg", reorder="nd", nd_leaf=8)
    assert dev.setup_frontal()["levels"] >= 4
    dev.scale_z(2.0, 0.5, 2.0)
    dev.set_params(scale_z=2.0, const_d=2.0, norm_d=s.norm_d, congestion=0.02)
    for _ in range(10):
        s.iterate()
    dev.step(10)
    for k in ("A", "B", "mu", "E", "z_mid", "beta_mid"):
        assert rel(dev.download(k), getattr(s, k)) < 1e-8, (T, k)
    want, got = s.kkt_all(), dev.kkt(range(7))
    for i in range(7):
        assert abs(got[i][0] - want[i]) <= 1e-7 * abs(want[i]) + 1e-14, (T, i)
    dev.close()



@pytest.mark.parametrize("T,prec", [(63, "jacobi"), (63, "multigrid"), (127, "jacobi")])
def test_pcg_alternatives_at_large_time_pitch(T, prec):
    """The PCG solvers at time pitch 64 / 128 give the direct solver's phi."""
    geom, _ = meshes.example("sphere", level=2)
    out = {}
    for tag in ("direct", "pcg"):
        dev = make(geom, T, 1e-3, "nd" if tag == "direct" else True)
        if tag == "direct":
            dev.setup_frontal(eps=1e-3)
        elif prec == "multigrid":
            assert dev.setup_multigrid(eps=1e-3, coarsest=12) is not None
        st = dev.run_phase("laplacian")
        assert st.cg_not_converged == 0
        out[tag] = dev.download("phi")
        dev.close()
    assert rel(out["pcg"], out["direct"]) < 1e-8


@pytest.mark.parametrize("mesh,kw,T,eps", [("sphere", dict(level=4), 15, 0.0), ("torus", dict(nu=72, nv=40), 31, 1e-3), ("knot", dict(nu=240, nv=10), 63, 0.0)])
def test_merged_tree_heights_give_the_same_solution(mesh, kw, T, eps):
    """Bands of tree heights handled by one launch per sweep (dots_front_desc.band_ptr, csrc/kernels_front.hip: the blocks of
    a band are merged on the device from the factor): every cut -- pairs, triples, four heights, ragged ones, on the
    plan's numbering and on a foreign one (index map in the sweeps) -- solves the systems like one launch per height, and
    a solve takes 2 x bands launches (one less with the top band stored as explicit inverses, dots_front_desc.top_inverse).
    A band of five heights is refused."""
    from dots_socp_amd import _lib

    geom, _ = meshes.example(mesh, **kw)
    out, launches = {}, {}
    H = None
    for tag, reorder, cuts in (("off", "nd", "unit"), ("auto", "nd", None), ("pairs", "nd", 2), ("triples", "nd", 3), ("fours", "nd", 4),
                               ("ragged", "nd", "ragged"), ("rcm_triples", True, 3), ("host_pairs", "nd", 2),
                               ("off_topinv", "nd", "unit"), ("triples_topinv", "nd", 3), ("rcm_fours_topinv", True, 4)):
        dev = make(geom, T, eps, reorder)
        if H is None:
            H = int(dev.plan.dissection.height.max()) + 1
        if cuts == "unit":
            bands = np.arange(H + 1)
        elif cuts == "ragged":
            bands = np.asarray(sorted(set(x for x in (0, 1, 4, 5, H - 1, H) if 0 <= x <= H)))
            if np.any(np.diff(bands) > 4):
                bands = np.asarray(sorted(set(range(0, H, 3)) | {H}))
        elif cuts is None:
            bands = None
        else:      # from the top: `cuts` heights per band, the leaves' band takes the rest
            bands = np.asarray(sorted(set(range(H, 0, -cuts)) | {0}))
        top = tag.endswith("topinv")       # the top band as explicit inverses: its forward launch writes the solution
        s = dev.setup_frontal(eps=eps, bands=bands, numeric="host" if tag.startswith("host") else "device", top_inverse=top if bands is not None else None)
        assert s["launches_per_solve"] == 2 * (len(s["bands"]) - 1) - (1 if s["top_inverse"] else 0)
        assert s["top_inverse"] == top or bands is None
        if not s["top_inverse"]:     # (an explicit inverse is read once: n^2 entries instead of n (n + 1) / 2 twice)
            assert s["bytes_per_solve_as_installed"] >= s["bytes_per_solve_one_block_per_node"] * (1.0 - 1e-12)
            assert not s["leaf_inverse"] or (len(s["bands"]) > 2 and s["bands"][1] == 1)
            if reorder == "nd":      # the planner's count of what the merged sweeps read = the device's own
                import scipy.sparse as sp

                from dots_socp_amd import frontal

                p_, d_ = dev.plan, dev.plan.dissection
                K_ = sp.csr_matrix((p_.lap_val, p_.lap_col, p_.lap_rowptr), shape=(p_.n_vertices,) * 2)
                nb_ = frontal.symbolic_native(d_, K_.indptr, K_.indices)[0]
                n_ = np.diff(d_.sep_ptr)
                e_ = sum(frontal.band_entries(d_, n_, nb_, int(lo), int(hi), leaf_inverse=s["leaf_inverse"])[0] for lo, hi in zip(s["bands"][:-1], s["bands"][1:]))
                assert s["bytes_per_solve_as_installed"] == 2.0 * e_ * (T + 1) * 8
        launches[tag] = s["launches_per_solve"]
        st = dev.run_phase("laplacian")
        assert st.cg_not_converged == 0
        phi = dev.download("phi")
        assert np.all(np.isfinite(phi)), tag
        mass = dev.plan.mass_vert[np.argsort(dev.plan.perm_vert)]
        out[tag] = gauge(phi, mass) if eps == 0.0 else phi
        if tag == "fours":
            with pytest.raises(_lib.HipLibraryError, match="band"):
                dev.setup_frontal(eps=eps, bands=np.asarray([0, 5, H]) if H > 5 else np.asarray([0, H + 1]))
        dev.close()
    assert launches["off"] == 2 * H and launches["pairs"] < launches["off"] and launches["fours"] <= launches["triples"] <= launches["pairs"]
    for tag in out:
        assert rel(out[tag], out["off"]) < 1e-10, tag


def test_two_components_and_empty_separators():
    """A mesh of two disjoint spheres: the dissection's first cut falls between the components (an empty separator whose node
    only passes updates on), the operator is regular with eps > 0: merged bands and the top inverse solve it like the PCG."""
    a, _ = meshes.example("sphere", level=3)
    va, ta = np.asarray(a["vertices"]), np.asarray(a["triangles"])
    v = np.concatenate([va, va * 0.7 + np.array([3.0, 0.2, -0.1])])
    t = np.concatenate([ta, ta + va.shape[0]])
    mu0 = np.concatenate([a["mu0"], a["mu0"][::-1]])
    mu1 = np.concatenate([a["mu1"], a["mu1"][::-1]])
    geom = dict(vertices=v, triangles=t, mu0=mu0 / mu0.sum(), mu1=mu1 / mu1.sum())
    out = {}
    for tag in ("direct", "direct_off", "pcg"):
        dev = make(geom, 15, 1e-2, "nd" if tag != "pcg" else True)
        if tag == "pcg":
            assert dev.setup_multigrid(eps=1e-2) is not None
        else:
            H = int(dev.plan.dissection.height.max()) + 1
            s = dev.setup_frontal(eps=1e-2, bands=None if tag == "direct" else np.arange(H + 1))
            assert s["empty_separators"] >= 1
        st = dev.run_phase("laplacian")
        assert st.cg_not_converged == 0
        out[tag] = dev.download("phi")
        dev.close()
    assert rel(out["direct"], out["direct_off"]) < 1e-10
    assert rel(out["direct"], out["pcg"]) < 1e-8


@pytest.mark.parametrize("mesh,kw,T", [("sphere", dict(level=4), 31), ("torus", dict(nu=72, nv=40), 15), ("knot", dict(nu=240, nv=10), 7), ("sphere", dict(level=3), 63)])
def test_forward_kernel_variants_agree(mesh, kw, T, monkeypatch):
    """The forward sweep of a band runs in the fold kernel (k_front_fwd: a row split over the whole workgroup, folded through LDS)
    or in the row kernel (k_front_fwd_rows: Q lane groups of a wavefront per row, the right-hand side staged once per workgroup),
    chosen per band by a rule.  Forced onto EVERY band that fits them, with and without merged tree heights, all variants solve
    the same systems: the solutions agree to rounding (the sums are formed in different orders), and a bad DOTS_FRONT_CFG is an error."""
    import os

    from dots_socp_amd import _lib

    geom, _ = meshes.example(mesh, **kw)
    pitch = max(8, 1 << int(np.ceil(np.log2(T + 1))))
    groups = 64 // max(pitch // 2, 1)              # lane groups of a wavefront = the largest Q of the row kernel
    out = {}
    settings = [("rule", {}), ("fold", {"DOTS_FRONT_ROWS": "0"}), ("rows_everywhere", {"DOTS_FRONT_ROWS": "2"}),
                ("leaves_in_band_kernels", {"DOTS_FRONT_LEAFINV": "0"}), ("unmerged_leaves_in_band_kernels", {"DOTS_FRONT_BANDS": "off", "DOTS_FRONT_LEAFINV": "0"}),
                ("unmerged_rule", {"DOTS_FRONT_BANDS": "off"}), ("unmerged_fold", {"DOTS_FRONT_BANDS": "off", "DOTS_FRONT_ROWS": "0"})]
    for q in (1, 2, 4, 8):
        if q <= groups:      # (the leaf band always fits the row kernel: its rows hold <= 16 columns)
            settings.append((f"r{q}_leaves", {"DOTS_FRONT_BANDS": "off", "DOTS_FRONT_CFG": f"fwd:r{q}"}))
    mass = None
    for tag, env in settings:
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        dev = make(geom, T, 1e-3, "nd")
        mass = dev.plan.mass_vert[np.argsort(dev.plan.perm_vert)] if mass is None else mass
        dev.setup_frontal(eps=1e-3)
        st = dev.run_phase("laplacian")
        assert st.cg_not_converged == 0
        out[tag] = dev.download("phi")
        dev.close()
        for k in env:
            monkeypatch.delenv(k)
    for tag, phi in out.items():
        assert np.all(np.isfinite(phi)), tag
        assert rel(phi, out["fold"]) < 1e-11, tag
    monkeypatch.setenv("DOTS_FRONT_CFG", "fwd:r3")
    dev = make(geom, T, 1e-3, "nd")
    with pytest.raises(_lib.HipLibraryError, match="DOTS_FRONT_CFG"):
        dev.setup_frontal(eps=1e-3)
    dev.close()
    monkeypatch.delenv("DOTS_FRONT_CFG")
    monkeypatch.setenv("DOTS_FRONT_ROWS", "7")
    with pytest.raises(_lib.HipLibraryError, match="DOTS_FRONT_ROWS"):
        make(geom, T, 1e-3, "nd")


@pytest.mark.parametrize("mesh,kw,T,eps", [("sphere", dict(level=4), 15, 0.0), ("torus", dict(nu=72, nv=40), 31, 1e-3), ("knot", dict(nu=240, nv=10), 63, 0.0),
                                           ("sphere", dict(level=3), 127, 1e-3), ("sphere", dict(level=3), 128, 1e-3), ("torus", dict(nu=30, nv=12), 5, 0.0)])
def test_leaves_as_local_inverses(mesh, kw, T, eps, monkeypatch):
    """Where the leaves' band is not merged, a leaf stores S = A_ss^-1 instead of [L^-1 ; G] and both sweeps take its coupling to the
    boundary from the CSR of K (kernels_front.hip: k_front_leaf_fwd / _bwd; the leaf's front holds original matrix entries only, the
    coupling is the same for every mode): n (n + 1) / 2 (S is symmetric) instead of n (n + 1) / 2 + b n entries per leaf, mode and sweep, the same solution to
    rounding, the same number of launches; the coupling comes from per-row records copied from the CSR at setup (one load behind the leaf's
    record) or from the CSR itself: bit-identical.  A mode pitch of 256 does not fit a leaf's vectors in LDS: the band kernels stay."""
    geom, _ = meshes.example(mesh, **kw)
    monkeypatch.setenv("DOTS_FRONT_BANDS", "off")
    out, info = {}, {}
    for tag in ("1", "0", "2"):      # 2: local inverses with the coupling read from the CSR instead of the per-row records
        monkeypatch.setenv("DOTS_FRONT_LEAFINV", tag)
        dev = make(geom, T, eps, "nd")
        s = dev.setup_frontal(eps=eps)
        info[tag] = (s["leaf_inverse"], dev.debug_counter(4), s["launches_per_solve"], s["bytes_per_solve_as_installed"], s["bytes_per_solve_one_block_per_node"])
        records = dev.debug_counter(5)
        assert records == (1 if tag == "1" and s["leaf_inverse"] else 0)
        st = dev.run_phase("laplacian")
        assert st.cg_not_converged == 0
        phi = dev.download("phi")
        assert np.all(np.isfinite(phi))
        mass = dev.plan.mass_vert[np.argsort(dev.plan.perm_vert)]
        out[tag] = gauge(phi, mass) if eps == 0.0 else phi
        n_leaves = int(np.sum(dev.plan.dissection.height == 0))
        dev.close()
    assert info["0"][:2] == (False, 0)
    if T == 128:
        assert info["1"][:2] == (False, 0)
    else:
        assert info["1"][0] and info["1"][1] == n_leaves
        assert info["1"][3] < info["0"][3] and info["1"][4] < info["0"][4] and info["1"][3] >= info["1"][4]
    assert info["1"][2] == info["0"][2]
    assert rel(out["1"], out["0"]) < 1e-11
    assert info["2"] == info["1"] and np.array_equal(out["2"], out["1"])      # the records hold the CSR's entries in its order


def test_leaf_coupling_falls_back_to_the_csr_on_high_degree_vertices(monkeypatch):
    """A latitude-longitude sphere of 64 meridians (poles with 64 neighbours) cut into leaves of up to 48 vertices: a leaf next to a pole has a
    boundary row with 8 entries inside the leaf -- more than a coupling record of the leaf kernels holds (LEAF_KC = 6).  The setup notices
    it and the leaf kernels walk the CSR of K instead (debug counter 5 = 0 with counter 4 > 0); the solution is the band kernels' one."""
    nlat, nlon = 12, 64
    th = np.pi * np.arange(1, nlat + 1) / (nlat + 1)
    ph = 2.0 * np.pi * np.arange(nlon) / nlon
    rings = np.stack([np.outer(np.sin(th), np.cos(ph)), np.outer(np.sin(th), np.sin(ph)), np.outer(np.cos(th), np.ones(nlon))], axis=2).reshape(-1, 3)
    v = np.concatenate([rings, [[0.0, 0.0, 1.0]], [[0.0, 0.0, -1.0]]])
    north, south = nlat * nlon, nlat * nlon + 1
    t = []
    for j in range(nlon):
        k = (j + 1) % nlon
        t.append((north, j, k))
        t.append((south, (nlat - 1) * nlon + k, (nlat - 1) * nlon + j))
        for i in range(nlat - 1):
            a0, a1, b0, b1 = i * nlon + j, i * nlon + k, (i + 1) * nlon + j, (i + 1) * nlon + k
            t += [(a0, b0, a1), (a1, b0, b1)]
    t = np.asarray(t)
    mu = np.ones(v.shape[0])
    mu0, mu1 = mu * (1.0 + v[:, 2]), mu * (1.0 - v[:, 2])
    geom = dict(vertices=v, triangles=t, mu0=mu0 / mu0.sum(), mu1=mu1 / mu1.sum())
    monkeypatch.setenv("DOTS_FRONT_BANDS", "off")
    out, counters = {}, {}
    for tag in ("1", "0"):
        monkeypatch.setenv("DOTS_FRONT_LEAFINV", tag)
        dev = make(geom, 15, 1e-2, "nd", nd_leaf=48)
        dev.setup_frontal(eps=1e-2)
        counters[tag] = (dev.debug_counter(4), dev.debug_counter(5))
        st = dev.run_phase("laplacian")
        assert st.cg_not_converged == 0
        out[tag] = dev.download("phi")
        dev.close()
    assert counters["1"][0] > 0 and counters["1"][1] == 0 and counters["0"] == (0, 0)
    assert np.all(np.isfinite(out["1"])) and rel(out["1"], out["0"]) < 1e-11


@pytest.mark.parametrize("mesh,kw,T,want", [("knot", {}, 31, 0), ("knot", {}, 63, 1), ("sphere", dict(level=5), 31, 1), ("torus", dict(nu=250, nv=160), 31, 1),
                                            ("torus", dict(nu=400, nv=250), 31, 0)])
def test_beta_mid_streaming_rule(mesh, kw, T, want, monkeypatch):
    """dots_front_setup decides per context whether steps 2+3 stream beta_mid with the non-temporal hint (Ctx::bm_nt; debug counter 6): not where
    factor + state fit the 256 MB Infinity Cache (knot), not where the sweeps touch more than 0.9 GB (torus100k), in between yes; DOTS_BM_NT overrides.
    (The iterates do not depend on it: test_hip_phases / test_hip_solver run with either setting.)"""
    geom, _ = meshes.example(mesh, **kw)
    for env, expect in ((None, want), ("0", 0), ("1", 1)):
        if env is None:
            monkeypatch.delenv("DOTS_BM_NT", raising=False)
        else:
            monkeypatch.setenv("DOTS_BM_NT", env)
        dev = make(geom, T, 0.0, "nd")
        dev.setup_frontal(eps=0.0)
        assert dev.debug_counter(6) == expect
        dev.close()
        if mesh == "torus" and kw["nu"] == 400:
            break      # (one setup of the large mesh is enough)


@pytest.mark.parametrize("seed", range(10))
def test_leaf_inverses_on_random_triangulations(seed, monkeypatch):
    """Irregular meshes with a boundary (Delaunay triangulations of random points, lifted off the plane), random sizes, leaf sizes, time
    grids and shifts: the leaves as local inverses solve the systems like the band kernels do."""
    from scipy.spatial import Delaunay

    rng = np.random.default_rng(100 + seed)
    n = int(rng.integers(150, 2500))
    pts = rng.random((n, 2))
    tri = Delaunay(pts).simplices.astype(np.int64)
    e1, e2 = pts[tri[:, 1]] - pts[tri[:, 0]], pts[tri[:, 2]] - pts[tri[:, 0]]
    area2 = e1[:, 0] * e2[:, 1] - e1[:, 1] * e2[:, 0]
    tri = tri[np.abs(area2) > 1e-7]                                   # (slivers on the hull)
    used = np.unique(tri)
    remap = -np.ones(n, dtype=np.int64)
    remap[used] = np.arange(used.size)
    pts, tri = pts[used], remap[tri]
    v = np.column_stack([pts, 0.2 * np.sin(3.0 * pts[:, 0]) * np.cos(2.0 * pts[:, 1])])
    mu0 = 1.0 + pts[:, 0]
    mu1 = 2.0 - pts[:, 1]
    geom = dict(vertices=v, triangles=tri, mu0=mu0 / mu0.sum(), mu1=mu1 / mu1.sum())
    T = int(rng.choice([5, 12, 31, 63]))
    leaf = int(rng.choice([8, 16, 24]))
    eps = float(rng.choice([0.0, 1e-2]))
    monkeypatch.setenv("DOTS_FRONT_BANDS", "off")
    out, leaves = {}, {}
    for tag in ("1", "0"):
        monkeypatch.setenv("DOTS_FRONT_LEAFINV", tag)
        dev = make(geom, T, eps, "nd", nd_leaf=leaf)
        dev.setup_frontal(eps=eps)
        leaves[tag] = dev.debug_counter(4)
        st = dev.run_phase("laplacian")
        assert st.cg_not_converged == 0
        phi = dev.download("phi")
        assert np.all(np.isfinite(phi))
        mass = dev.plan.mass_vert[np.argsort(dev.plan.perm_vert)]
        out[tag] = gauge(phi, mass) if eps == 0.0 else phi
        dev.close()
    assert leaves["1"] > 0 and leaves["0"] == 0
    assert rel(out["1"], out["0"]) < 1e-10
