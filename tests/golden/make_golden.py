"""Record golden vectors from the REFERENCE implementation (build container only).

Runs the reference's own functions (loaded from /root/reference by ref_shim.py, never
copied) on small seeded inputs and stores inputs + outputs as ``*.npz`` next to this
file.  The fixtures are data; they travel to the GPU box, the reference does not.

    python tests/golden/make_golden.py            # (re)generate everything
    python tests/golden/make_golden.py ops runs   # subsets: ops | runs | full | palm | f1 | settings

Fixture families
    ops_<mesh>.npz    per-function input/output vectors
    run_<case>.npz    solver_socp runs: k-iteration states, KKT trajectories, lazy-schedule
                      histories, stopping iteration, cost, objective (rows a1, a11-a14, a16); *_palm_*: is_palm=True
    f1_<case>.npz     outputs of the solver / solver_raw decorators and of utils/evaluate_solution.py (row f1)
    settings_get_mu.npz   get_mu of every data/settings/*.py on one synthetic mesh (row f3)
"""
from __future__ import annotations

import io
import os
import sys
import contextlib
import logging

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_shim  # noqa: E402
from dots_socp_amd import meshes  # noqa: E402


def csr_parts(m, prefix):
    m = m.tocsr()
    m.sort_indices()
    return {f"{prefix}_data": m.data, f"{prefix}_indices": m.indices, f"{prefix}_indptr": m.indptr,
            f"{prefix}_shape": np.array(m.shape)}


def tiny_meshes(ref):
    v, t, _ = ref.plane_mesh.generate_mesh(4)
    out = {"refplane4": (v, np.asarray(t))}
    out["ico1"] = meshes.icosphere(1)
    out["torus8x6"] = meshes.torus(8, 6)
    return out


def geometry_for(ref, v, t, centers=None, normalize=True):
    """GeometryData built the way load_example + normalize_geometry do (hand normalisation)."""
    g, scale = meshes.make_geometry(v, t, normalize=normalize)
    if centers is None:
        centers = meshes.farthest_vertices(g["vertices"], 0, 3)
    g["mu0"] = meshes.bump_density(g["vertices"], g["area_vertices"], [centers[0]], 0.6, 0.2)
    g["mu1"] = meshes.bump_density(g["vertices"], g["area_vertices"], [centers[1], centers[2]], 0.6, 0.2)
    return g, scale


# --------------------------------------------------------------------------- #
# per-function vectors
# --------------------------------------------------------------------------- #
def make_ops(ref, name, v, t, n_time, seed):
    import scipy.sparse as scsp

    S = ref.solver_module
    rng = np.random.default_rng(seed)
    g, _ = geometry_for(ref, v, t)
    v, t, edges = g["vertices"], g["triangles"], g["edges"]
    V, F, T = v.shape[0], t.shape[0], n_time
    h = 1.0 / T

    area, ang, base = ref.pre.geometricQuantities(v, t, edges)
    G, Dv, L = ref.pre.geometricMatrices(v, t, edges, area, ang, base)
    c2v, area_v_raw, v2c, area_vc_raw = ref.pre.trianglesToVertices(v, t, area)
    area_v = area_v_raw / 3.0
    area_vc = area_vc_raw / 3.0

    out = dict(vertices=v, triangles=t, n_time=np.array(T), mu0=g["mu0"], mu1=g["mu1"],
               area_triangles=area, angle_triangles=ang, base_function=base, area_vertices_raw=area_v_raw)
    out.update(csr_parts(G, "G"))
    out.update(csr_parts(L, "L"))

    # a4
    x_t = rng.standard_normal((T, V))
    x_c = rng.standard_normal((T + 1, V))
    out.update(x_t=x_t, x_c=x_c,
               grad_time=S.vanilla_grad_time(h, x_c), div_time=S.vanilla_div_time(h, x_t))
    # a5
    x_s = rng.standard_normal((T + 1, F, 3))
    out.update(x_s=x_s,
               grad_space=S.vanilla_grad_space(T, F, G, x_c), div_space=S.vanilla_div_space(T, V, F, Dv, x_s))
    # a6
    x_d = rng.standard_normal((T, 2, 3, F, 3))
    out.update(x_d=x_d,
               decouple=S.decouple_spacial(x_s, scale_z=1.7), decouple_adjoint=S.decouple_adjoin_spacial(x_d, scale_z=1.7),
               decouple_adjoint_time=S.decouple_adjoint_time(x_t))
    # a10
    w_t = np.kron(np.ones(T), area_v).reshape(T, V)
    w_d = np.kron(np.kron(np.ones(6 * T), area), np.ones(3)).reshape(T, 2, 3, F, 3)
    out.update(nsq_time=np.array(S.norm_square_weight(w_t, T, x_t)),
               nsq_space_dec=np.array(S.norm_square_weight(w_d, T, x_d)))

    # a7: SOC projection with the constants built as solver_socp.py:161-192 builds them
    d_corner = np.sqrt(np.kron(np.ones(3), area) / area_vc)
    diag_soc = np.kron(np.ones(T), d_corner).reshape(T, 3, F)
    diag_soc_glob = np.kron(np.kron(np.ones(2 * T), d_corner), np.ones(3)).reshape(T, 2, 3, F, 3)
    m_v2c = scsp.kron(scsp.eye(T), v2c).tocsr()
    m_c2v_one = scsp.kron(scsp.eye(T), v2c.transpose()).tocsr()
    st = {k: rng.standard_normal(s) for k, s in dict(
        A=(T, V), B=(T + 1, F, 3), lambda_c=(T, V), mu=(T, V), E=(T + 1, F, 3),
        beta_fst=(T, V), beta_mid=(T, 2, 3, F, 3), beta_end=(T, V)).items()}
    # make a share of the cones hit each branch of the projection (inside, polar, boundary)
    st["beta_fst"][:, ::3] -= 6.0
    st["beta_fst"][:, 1::3] += 6.0
    const_d, scale_z = 1.3, 2.5
    memo = (np.zeros((T, V)), np.zeros((T, 2, 3, F, 3)), np.zeros((T, V)), np.zeros((T, 2, 3, F, 3)))
    z_fst, z_mid, z_end = np.zeros((T, V)), np.zeros((T, 2, 3, F, 3)), np.zeros((T, V))
    S.vanilla_solve_proj_soc(diag_soc_glob, diag_soc, m_c2v_one, m_v2c,
                             st["A"], st["B"], st["beta_fst"], st["beta_mid"], st["beta_end"],
                             memo=memo, const_d=const_d, scale_z=scale_z, output=(z_fst, z_mid, z_end))
    out.update({f"st_{k}": a for k, a in st.items()})
    out.update(soc_const_d=np.array(const_d), soc_scale_z=np.array(scale_z), soc_z_fst=z_fst, soc_z_mid=z_mid, soc_z_end=z_end)

    # a8: (q, lambda_c) closed form
    congestion, r = 0.1, 1.7
    diag_b = 1.0 + (2.0 * scale_z ** 2) * np.ones(T + 1)
    diag_b[0] = diag_b[-1] = 1.0 + scale_z ** 2
    diag_b = np.kron(diag_b, np.ones(F * 3)).reshape(T + 1, F, 3)
    phi = rng.standard_normal((T + 1, V))
    dt_phi, dx_phi = S.vanilla_grad_time(h, phi), S.vanilla_grad_space(T, F, G, phi)
    qA, qB, qL = np.zeros((T, V)), np.zeros((T + 1, F, 3)), np.zeros((T, V))
    S.vanilla_solve_q_lambda(scale_z, diag_b, congestion, r, dt_phi, dx_phi, st["mu"], st["E"], z_fst, z_mid, z_end,
                             st["beta_fst"], st["beta_mid"], st["beta_end"],
                             memo=(np.zeros((T, V)), np.zeros((T + 1, F, 3))), output=(qA, qB, qL))
    out.update(q_phi=phi, q_congestion=np.array(congestion), q_r=np.array(r), q_A=qA, q_B=qB, q_lambda_c=qL)

    # a2/a3: Laplacian step (eps = 0 and eps > 0)
    w_s = np.kron(np.kron(np.ones(T + 1), area), np.ones(3)).reshape(T + 1, F, 3)
    w_c = np.kron(np.ones(T + 1), area_v).reshape(T + 1, V)
    bnd = np.zeros((T + 1, V))
    bnd[0], bnd[-1] = -g["mu0"] / (r * h), g["mu1"] / (r * h)
    div_time = lambda x: S.vanilla_div_time(h, x)  # noqa: E731
    div_space = lambda x: S.vanilla_div_space(T, V, F, Dv, x)  # noqa: E731
    for tag, eps in (("eps0", 0.0), ("eps1", 1e-2)):
        inv = ref.lap.buildLaplacianMatrix(n_time=T, stepsize_time=h, n_vertices=V, area_vertices=area_v,
                                           laplacian_space=L, eps=eps)
        phi_io = phi.copy()
        S.vanilla_solve_laplacian(inv, div_time, div_space, w_t, w_s, w_c, eps,
                                  st["A"], st["B"], st["lambda_c"], st["mu"], st["E"], bnd, output=phi_io)
        out[f"lap_{tag}_phi"] = phi_io
        out[f"lap_{tag}_eps"] = np.array(eps)
    out["lap_bnd"] = bnd

    np.savez_compressed(os.path.join(HERE, f"ops_{name}.npz"), **out)
    print("wrote", f"ops_{name}.npz")


# --------------------------------------------------------------------------- #
# solver runs
# --------------------------------------------------------------------------- #
def run_reference(ref, geometry, n_time, **kw):
    buf = io.StringIO()
    logging.disable(logging.CRITICAL)
    try:
        with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(buf):
            sol, hist = ref.solver_socp(n_time, dict(geometry), **kw)
    finally:
        logging.disable(logging.NOTSET)
    return sol, hist


def save_run(name, geometry, n_time, kw, sol, hist, keep, scale_factor=None):
    out = dict(vertices=geometry["vertices"], triangles=geometry["triangles"], mu0=geometry["mu0"], mu1=geometry["mu1"],
               n_time=np.array(n_time))
    for k, val in kw.items():
        if k == "tol_checkpoints":
            out["kw_tol_checkpoints"] = np.array(val, dtype=np.float64)
        elif k == "init_solution":
            continue
        else:
            out[f"kw_{k}"] = np.array(val)
    for k in keep:
        out[f"sol_{k}"] = sol[k]
    out["hist_kkt_errors"] = np.asarray(hist.kkt_errors, dtype=np.float64)
    out["hist_kkt_iteration"] = np.asarray(hist.kkt_iteration, dtype=np.float64)
    for k, val in hist.history.items():
        out["hist_" + k.replace(" ", "_")] = np.asarray(val, dtype=np.float64)
    out["last_iteration"] = np.array(int(hist.kkt_iteration[-1]))
    if scale_factor is not None:  # interface.py:303-308 reports cost / scale_factor**2
        out["scale_factor"] = np.array(scale_factor)
    if sol.get("checkpoints"):
        out["ckpt_iteration"] = np.array([c["iteration"] for c in sol["checkpoints"]])
        out["ckpt_mu"] = np.stack([c["mu"] for c in sol["checkpoints"]])
    np.savez_compressed(os.path.join(HERE, f"run_{name}.npz"), **out)
    print("wrote", f"run_{name}.npz", "last iteration", int(hist.kkt_iteration[-1]),
          "cost", hist.history["Transportation cost"][-1])


ALL_STATE = ("phi", "A", "B", "lambda_c", "z_fst", "z_mid", "z_end", "mu", "E", "beta_fst", "beta_mid", "beta_end")
SMALL_STATE = ("phi", "A", "lambda_c", "mu")


def make_runs(ref, full, small=True):
    tm = tiny_meshes(ref)
    cases = []
    g_ico1, _ = geometry_for(ref, *tm["ico1"])
    g_tor, _ = geometry_for(ref, *tm["torus8x6"])
    g_pl4, _ = geometry_for(ref, *tm["refplane4"], normalize=False)
    g_ico2, _ = geometry_for(ref, *meshes.icosphere(2))
    # k-iteration states with every KKT value and the objective recorded each iteration
    cases.append(("ico1_T6_k12_steps", g_ico1, 6, dict(nit=12, tol=1e-12, check_kkt_step_by_step=True), ALL_STATE))
    cases.append(("ico1_T6_k1_steps", g_ico1, 6, dict(nit=1, tol=1e-12, check_kkt_step_by_step=True), ALL_STATE))
    cases.append(("torus_T5_eps_k15_steps", g_tor, 5, dict(nit=15, tol=1e-12, eps=1e-3, check_kkt_step_by_step=True), ALL_STATE))
    cases.append(("torus_T5_cong_k15_steps", g_tor, 5, dict(nit=15, tol=1e-12, congestion=0.2, check_kkt_step_by_step=True), ALL_STATE))
    cases.append(("ico1_T6_noz_k10_steps", g_ico1, 6, dict(nit=10, tol=1e-12, is_z_scaling=False, check_kkt_step_by_step=True), ALL_STATE))
    cases.append(("ico1_T6_cscale_k30_steps", g_ico1, 6, dict(nit=30, tol=1e-12, is_constant_scaling=True, check_kkt_step_by_step=True), ALL_STATE))
    # lazy KKT schedule + penalty updates + z re-scale (it >= 100)
    cases.append(("ico1_T6_k150_lazy", g_ico1, 6, dict(nit=150, tol=1e-12, congestion=0.05), ALL_STATE))
    # converged runs
    cases.append(("refplane4_T8_tol1e-3", g_pl4, 8, dict(nit=3000, tol=1e-3), ALL_STATE))
    cases.append(("ico2_T15_cong_tol1e-3", g_ico2, 15, dict(nit=3000, tol=1e-3, congestion=0.1), SMALL_STATE))
    cases.append(("ico2_T15_ckpt_tol1e-3", g_ico2, 15, dict(nit=3000, tol=1e-3, tol_checkpoints=[1e-1, 1e-2]), SMALL_STATE))
    cases.append(("torus_T7_tol1e-4", g_tor, 7, dict(nit=5000, tol=1e-4), SMALL_STATE))
    if not small:
        cases = []
    if full:
        # the survey's headline cases: reference plane n=20, T=31, tol=1e-3
        v, t, _ = ref.plane_mesh.generate_mesh(20)
        t = np.asarray(t)
        g20, scale20 = meshes.make_geometry(v, t, normalize=True)
        mu0, mu1 = ref.plane_setting.get_mu(meshes.vertex_areas(v.shape[0], t, meshes.triangle_areas(v, t)), v)
        g20["mu0"], g20["mu1"] = mu0 / mu0.sum(), mu1 / mu1.sum()
        cases.append(("refplane20_T31_tol1e-3", g20, 31, dict(nit=1000, tol=1e-3), SMALL_STATE, scale20))
        cases.append(("refplane20_T31_cong_tol1e-3", g20, 31, dict(nit=1000, tol=1e-3, congestion=0.1), SMALL_STATE, scale20))
    for name, g, T, kw, keep, *rest in cases:
        kw_run = dict(kw)
        if "tol_checkpoints" in kw_run:
            kw_run["tol_checkpoints"] = list(kw_run["tol_checkpoints"])
        sol, hist = run_reference(ref, g, T, **kw_run)
        save_run(name, g, T, kw, sol, hist, keep, *rest)


# --------------------------------------------------------------------------- #
# the steps either side of the path and the is_palm variant
# --------------------------------------------------------------------------- #
def quiet(fn, *a, **kw):
    buf = io.StringIO()
    logging.disable(logging.CRITICAL)
    try:
        with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(buf):
            return fn(*a, **kw)
    finally:
        logging.disable(logging.NOTSET)


def make_palm(ref):
    """solver_socp(is_palm=True) (solver_socp.py:668-672): an extra (q, lambda) solve ahead of step 1."""
    tm = tiny_meshes(ref)
    g_ico1, _ = geometry_for(ref, *tm["ico1"])
    g_ico2, _ = geometry_for(ref, *meshes.icosphere(2))
    for name, g, T, kw, keep in (
        ("ico1_T6_palm_k12_steps", g_ico1, 6, dict(nit=12, tol=1e-12, is_palm=True, check_kkt_step_by_step=True), ALL_STATE),
        ("ico1_T6_palm_cong_k40_lazy", g_ico1, 6, dict(nit=40, tol=1e-12, is_palm=True, congestion=0.05), ALL_STATE),
        ("ico2_T15_palm_tol1e-3", g_ico2, 15, dict(nit=3000, tol=1e-3, is_palm=True), SMALL_STATE),
    ):
        sol, hist = run_reference(ref, g, T, **kw)
        save_run(name, g, T, kw, sol, hist, keep)


def make_f1(ref):
    """Outputs of the reference's solver / solver_raw decorators (socp/solver_decorator.py:10-72, utils/type.py:48-65)
    and of utils/evaluate_solution.py:7-58 on recorded solutions."""
    E = ref.evaluate_solution
    g_ico2, _ = geometry_for(ref, *meshes.icosphere(2))
    g_tor, _ = geometry_for(ref, *ref_tiny(ref)["torus8x6"])
    for name, g, T, kw in (
        ("ico2_T15_ckpt", g_ico2, 15, dict(nit=3000, tol=1e-3, tol_checkpoints=[1e-1, 1e-2])),
        ("torus_T7_cong", g_tor, 7, dict(nit=400, tol=1e-3, congestion=0.05)),
    ):
        out = dict(vertices=g["vertices"], triangles=g["triangles"], mu0=g["mu0"], mu1=g["mu1"], n_time=np.array(T))
        for k, val in kw.items():
            out[f"kw_{k}"] = np.array(val, dtype=np.float64) if k == "tol_checkpoints" else np.array(val)
        for tag, fn in (("raw", ref.decorator.solver_raw), ("center", ref.decorator.solver)):
            kw_run = {k: (list(v) if k == "tol_checkpoints" else v) for k, v in kw.items()}
            sol, hist = quiet(fn, T, dict(g), **kw_run)
            out[f"{tag}_mu"], out[f"{tag}_E"] = sol["mu"], sol["E"]
            out[f"{tag}_last_iteration"] = np.array(int(hist.kkt_iteration[-1]))
            if sol.get("checkpoints"):
                out[f"{tag}_ckpt_mu"] = np.stack([c["mu"] for c in sol["checkpoints"]])
                out[f"{tag}_ckpt_E"] = np.stack([c["E"] for c in sol["checkpoints"]])
                out[f"{tag}_ckpt_iteration"] = np.array([c["iteration"] for c in sol["checkpoints"]])
            # the three checks of evaluate_solution.py on this solution; "exact" = a seeded perturbation of it
            mu = sol["mu"]
            rng = np.random.default_rng(7)
            exact = mu * (1.0 + 0.05 * rng.standard_normal(mu.shape)) + 1e-4 * rng.standard_normal(mu.shape)
            out[f"{tag}_exact"] = exact
            out[f"{tag}_mass_conservation"] = np.array(quiet(E.check_mass_conservation, mu, verbose=False))
            neg, layers = quiet(E.check_negative_mass, mu, verbose=False)
            out[f"{tag}_negative_mass"], out[f"{tag}_negative_mass_layers"] = np.array(neg), layers
            err = quiet(E.compare_with_exact_transportation, mu, exact, g, verbose=False)
            out[f"{tag}_versus_exact"] = np.array([err["l1"], err["l2"], err["linf"]])
        np.savez_compressed(os.path.join(HERE, f"f1_{name}.npz"), **out)
        print("wrote", f"f1_{name}.npz")
    # the plane example's analytic transport (data/settings/plane.py:29-46) on the reference's own plane mesh
    v, t, _ = ref.plane_mesh.generate_mesh(6)
    t = np.asarray(t)
    av = meshes.vertex_areas(v.shape[0], t, meshes.triangle_areas(v, t))
    tt = np.linspace(0.0, 1.0, 5)
    np.savez_compressed(os.path.join(HERE, "f1_plane_exact.npz"), vertices=v, triangles=t, area_vertices=av, t_array=tt,
                        exact=ref.plane_setting.get_exact_transportation(tt, v, av))
    print("wrote f1_plane_exact.npz")


def ref_tiny(ref):
    return tiny_meshes(ref)


def make_settings(ref):
    """get_mu of every data/settings/*.py on ONE synthetic mesh with enough vertices for the fixed
    vertex indices the recipes use (the packaged .off meshes are LFS pointers): torus 130 x 100 scaled to [-1.2, 1.2]."""
    v, t = meshes.torus(130, 100)
    v = 1.2 * v / np.abs(v).max()
    av = meshes.vertex_areas(v.shape[0], t, meshes.triangle_areas(v, t))
    out = dict(vertices=v, triangles=t, area_vertices=av)
    for name, mod in sorted(ref.settings.items()):
        if name == "sphere":      # reads data_mu/*.txt: no recipe to restate
            continue
        mu0, mu1 = mod.get_mu(av.copy(), v.copy())
        out[f"{name}_mu0"], out[f"{name}_mu1"] = np.asarray(mu0, float), np.asarray(mu1, float)
    # a second mesh (sphere of radius 1.45) for the two recipes whose second density has no mass on the torus
    v2, t2 = meshes.icosphere(4, radius=1.45)
    av2 = meshes.vertex_areas(v2.shape[0], t2, meshes.triangle_areas(v2, t2))
    out.update(sphere_vertices=v2, sphere_triangles=t2, sphere_area_vertices=av2)
    for name in ("eight", "knots_3"):
        mu0, mu1 = ref.settings[name].get_mu(av2.copy(), v2.copy())
        assert mu0.sum() > 0 and mu1.sum() > 0, name
        out[f"sphere_{name}_mu0"], out[f"sphere_{name}_mu1"] = np.asarray(mu0, float), np.asarray(mu1, float)
    np.savez_compressed(os.path.join(HERE, "settings_get_mu.npz"), **out)
    print("wrote settings_get_mu.npz", len(out) // 2 - 1, "settings")


def main(argv):
    what = set(argv) or {"ops", "runs", "full", "palm", "f1", "settings"}
    ref = ref_shim.load_reference()
    if "ops" in what:
        for i, (name, (v, t)) in enumerate(tiny_meshes(ref).items()):
            make_ops(ref, name, v, t, n_time=(4, 6, 5)[i], seed=100 + i)
    if "runs" in what or "full" in what:
        make_runs(ref, full="full" in what, small="runs" in what)
    if "palm" in what:
        make_palm(ref)
    if "f1" in what:
        make_f1(ref)
    if "settings" in what:
        make_settings(ref)


if __name__ == "__main__":
    main(sys.argv[1:])
