"""CPU tests of the host-side multigrid setup (dots_socp_amd/multigrid.py): hierarchy invariants, the
three-kernel form of the V-cycle the device uses, and its quality as a PCG preconditioner."""
import numpy as np
import pytest
import scipy.sparse as sp

from dots_socp_amd import geometry, meshes, multigrid


def problem(name, **kw):
    if name == "refplane":
        v, t = meshes.plane(kw.get("n", 12))
        g, _ = meshes.make_geometry(v, t, np.ones(v.shape[0]) / v.shape[0], np.ones(v.shape[0]) / v.shape[0])
    else:
        g, _ = meshes.example(name, **kw)
    plan = geometry.build_plan(7, g)
    K = sp.csr_matrix((plan.lap_val, plan.lap_col, plan.lap_rowptr), shape=(plan.n_vertices,) * 2)
    return plan, K


CASES = [("sphere", dict(level=3)), ("refplane", dict(n=12)), ("torus", dict(nu=24, nv=16)), ("knot", dict(nu=60, nv=8))]


@pytest.mark.parametrize("name,kw", CASES)
def test_hierarchy_invariants(name, kw):
    plan, K = problem(name, **kw)
    levels = multigrid.build_hierarchy(K, plan.mass_vert, coarsest=12)
    assert len(levels) >= 2
    summary = multigrid.hierarchy_summary(levels)
    assert summary["operator_complexity"] < 2.5
    M = sp.diags(plan.mass_vert).tocsr()
    Kl = K
    for lv in levels:
        # K_l, M_l share one pattern and are symmetric; constants stay in the null space of K_l
        assert np.array_equal(lv.K.indptr, lv.M.indptr) and np.array_equal(lv.K.indices, lv.M.indices)
        assert abs(lv.K - lv.K.T).max() < 1e-12 * abs(lv.K).max()
        assert abs(lv.K - Kl).max() < 1e-12 * abs(Kl).max() and abs(lv.M - M).max() < 1e-12 * abs(M).max()
        if lv.P is not None:
            assert abs(lv.R - lv.P.T).max() == 0.0
            assert np.array_equal(lv.KP.indptr, lv.MP.indptr) and np.array_equal(lv.KP.indices, lv.MP.indices)
            assert abs(lv.KP - lv.K @ lv.P).max() < 1e-12 * abs(lv.KP).max()
            assert abs(lv.MP - lv.M @ lv.P).max() < 1e-12 * max(abs(lv.MP).max(), 1e-300)
            assert np.array_equal(lv.PP.indices, lv.KP.indices) and abs(lv.PP - lv.P).max() == 0.0
            Kl, M = (lv.R @ lv.K @ lv.P).tocsr(), (lv.R @ lv.M @ lv.P).tocsr()
    # the near-null-space candidate is carried exactly: K_l c_l = 0 and P_l c_{l+1} = c_l
    assert np.allclose(levels[0].cand, 1.0)
    for lv, nxt in zip(levels[:-1], levels[1:]):
        assert np.max(np.abs(lv.K @ lv.cand)) < 1e-10 * abs(lv.K).max() * np.max(np.abs(lv.cand))
        assert np.max(np.abs(lv.P @ nxt.cand - lv.cand)) < 1e-10 * np.max(np.abs(lv.cand))


@pytest.mark.parametrize("name,kw", CASES)
@pytest.mark.parametrize("shift", [0.0, 9.3, 1900.0])
def test_fused_cycle_equals_plain_cycle(name, kw, shift):
    plan, K = problem(name, **kw)
    levels = multigrid.build_hierarchy(K, plan.mass_vert, coarsest=12)
    vc = multigrid.CpuVcycle(levels, shift)
    b = np.random.default_rng(1).standard_normal(plan.n_vertices)
    if shift == 0.0:
        b -= b.mean()
    plain, fused = vc.cycle(b), vc.cycle_fused(b)
    assert np.max(np.abs(plain - fused)) < 1e-12 * np.max(np.abs(plain))


def pcg_iterations(A, b, prec, dinv, tol=1e-10, maxit=2000):
    x = np.zeros_like(b)
    r = b.copy()
    z = prec(r)
    p = z.copy()
    rz = r @ z
    bref = b @ (dinv * b)
    for it in range(1, maxit + 1):
        Ap = A @ p
        alpha = rz / (p @ Ap)
        x += alpha * p
        r -= alpha * Ap
        if r @ (dinv * r) <= tol * tol * bref:
            return it, x
        z = prec(r)
        rzn = r @ z
        p = z + (rzn / rz) * p
        rz = rzn
    return maxit, x


@pytest.mark.parametrize("name,kw", [("sphere", dict(level=4)), ("torus", dict(nu=60, nv=40))])
def test_multigrid_beats_jacobi(name, kw):
    plan, K = problem(name, **kw)
    levels = multigrid.build_hierarchy(K, plan.mass_vert)
    rng = np.random.default_rng(2)
    for shift in (0.0, plan.time_eigs[1], plan.time_eigs[-1]):
        A = (K + shift * sp.diags(plan.mass_vert)).tocsr()
        dinv = 1.0 / A.diagonal()
        b = rng.standard_normal(plan.n_vertices)
        if shift == 0.0:
            b -= b.mean()
        it_mg, x = pcg_iterations(A, b, multigrid.CpuVcycle(levels, shift).cycle_fused, dinv)
        it_j, _ = pcg_iterations(A, b, lambda r: dinv * r, dinv)
        assert it_mg <= 40 and (it_mg * 3 <= it_j or it_j <= 40), (shift, it_mg, it_j)
        assert np.max(np.abs(A @ x - b)) < 1e-7 * np.max(np.abs(b))
