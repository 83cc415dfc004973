"""GPU parity, function by function: HIP path (through the C ABI) vs the CPU oracle and the
reference's own golden vectors, on the small golden meshes."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR, load_oracle

pytestmark = pytest.mark.gpu

O = load_oracle()
OPS = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN_DIR, "ops_*.npz")))
STATE = ("phi", "A", "B", "lambda_c", "z_fst", "z_mid", "z_end", "mu", "E", "beta_fst", "beta_mid", "beta_end")
FP_TOL = 1e-12   # fp64 element-wise kernels vs numpy: summation order only


def rel(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def remove_gauge(phi, mass_v):
    w = np.broadcast_to(mass_v[None, :], phi.shape)
    return phi - np.sum(phi * w) / np.sum(w)


def golden(name):
    return np.load(os.path.join(GOLDEN_DIR, name))


def random_state(s, seed):
    rng = np.random.default_rng(seed)
    for k in STATE:
        setattr(s, k, rng.standard_normal(getattr(s, k).shape))
    s.beta_fst[:, ::3] -= 6.0     # reach all three branches of the cone projection
    s.beta_fst[:, 1::3] += 6.0


def make_pair(g, lap_solver="spacetime_pcg", reorder=True, eps=0.0, congestion=0.0, seed=7):
    from dots_socp_amd.device import DeviceProblem

    geom = dict(vertices=g["vertices"], triangles=g["triangles"], mu0=g["mu0"], mu1=g["mu1"])
    T = int(g["n_time"])
    s = O.OracleSolver(T, geom, congestion=congestion, eps=eps)
    random_state(s, seed)
    s.r, s.sz, s.d = 1.7, 2.5, 1.3
    s.norm_d *= 1.3
    s.bnd /= s.r
    dev = DeviceProblem(T, geom, lap_solver=lap_solver, reorder=reorder)
    for k in STATE:
        dev.upload(k, getattr(s, k))
    dev.set_params(r=s.r, scale_z=s.sz, const_d=s.d, norm_d=s.norm_d, congestion=congestion, eps=eps, tau=s.tau,
                   cg_tol=1e-12, cg_max_iter=20000)
    return s, dev


@pytest.mark.parametrize("fname", OPS)
@pytest.mark.parametrize("reorder", [False, True])
def test_upload_download_roundtrip(fname, reorder):
    s, dev = make_pair(golden(fname), reorder=reorder)
    for k in STATE:
        assert np.array_equal(dev.download(k), getattr(s, k)), k
    dev.close()


@pytest.mark.parametrize("fname", OPS)
def test_soc_projection_a7(fname):
    g = golden(fname)
    s, dev = make_pair(g)
    s.step_soc_projection()
    dev.run_phase("soc_projection")
    for k in ("z_fst", "z_mid", "z_end"):
        assert rel(dev.download(k), getattr(s, k)) < FP_TOL, k
    # and directly against the reference's recorded output for its own inputs
    for k in ("A", "B", "beta_fst", "beta_mid", "beta_end"):
        dev.upload(k, g[f"st_{k}"])
    dev.set_params(scale_z=float(g["soc_scale_z"]), const_d=float(g["soc_const_d"]))
    dev.run_phase("soc_projection")
    for k in ("z_fst", "z_mid", "z_end"):
        assert rel(dev.download(k), g[f"soc_{k}"]) < FP_TOL, k
    dev.close()


@pytest.mark.parametrize("fname", OPS)
@pytest.mark.parametrize("congestion", [0.0, 0.1])
def test_q_lambda_and_multipliers_a8_a9(fname, congestion):
    s, dev = make_pair(golden(fname), congestion=congestion)
    s.step_q_lambda()
    s.step_multipliers()
    dev.run_phase("q_lambda_mult")
    for k in ("A", "B", "lambda_c", "mu", "E", "beta_fst", "beta_mid", "beta_end"):
        assert rel(dev.download(k), getattr(s, k)) < FP_TOL, k
    for k in ("phi", "z_fst", "z_mid", "z_end"):
        assert np.array_equal(dev.download(k), getattr(s, k)), k
    dev.close()


@pytest.mark.parametrize("fname", OPS)
@pytest.mark.parametrize("lap_solver", ["spacetime_pcg", "modal_pcg", "modal_pcg+mg", "modal_pcg+direct", "modal_pcg+direct_host", "modal_pcg+direct_nd"])
@pytest.mark.parametrize("eps", [0.0, 1e-2])
def test_laplacian_step_a2_a3(fname, lap_solver, eps):
    g = golden(fname)
    s, dev = make_pair(g, lap_solver=lap_solver.split("+")[0], eps=eps, reorder="nd" if lap_solver.endswith("_nd") else True)
    if lap_solver.endswith("+mg"):
        summary = dev.setup_multigrid(eps=eps, coarsest=6)
        assert summary is not None and summary["levels"] >= 2
    direct = "+direct" in lap_solver
    if direct:    # tiny leaves: a tree of several levels even on the fixture meshes
        nd = lap_solver.endswith("_nd")
        summary = dev.setup_frontal(eps=eps, leaf=None if nd else 4, numeric="host" if lap_solver.endswith("_host") else "device")
        assert summary["levels"] >= (2 if nd else 3)
    s.step_laplacian()
    st = dev.run_phase("laplacian")
    assert st.cg_not_converged == 0 and (direct or st.cg_last_iterations > 0)
    got, want = dev.download("phi"), s.phi
    if eps == 0.0:
        got, want = remove_gauge(got, s.mass_v), remove_gauge(want, s.mass_v)
    assert rel(got, want) < 1e-9, (st.cg_last_iterations, st.cg_last_rel_residual)
    dev.close()


@pytest.mark.parametrize("fname", OPS)
def test_laplacian_matches_reference_vector(fname):
    g = golden(fname)
    for tag in ("eps0", "eps1"):
        eps = float(g[f"lap_{tag}_eps"])
        s, dev = make_pair(g, eps=eps)
        for k in ("A", "B", "lambda_c", "mu", "E"):
            dev.upload(k, g[f"st_{k}"])
        dev.upload("phi", g["q_phi"])
        dev.set_params(r=float(g["q_r"]))
        dev.run_phase("laplacian")
        got, want = dev.download("phi"), g[f"lap_{tag}_phi"]
        if eps == 0.0:
            got, want = remove_gauge(got, s.mass_v), remove_gauge(want, s.mass_v)
        assert rel(got, want) < 1e-9
        dev.close()


@pytest.mark.parametrize("fname", OPS)
def test_operators_a4_a5_a6(fname):
    g = golden(fname)
    s, dev = make_pair(g, eps=1e-2)
    assert rel(dev.apply_operator("grad_time", g["x_c"]), g["grad_time"]) < FP_TOL
    assert rel(dev.apply_operator("div_time", g["x_t"]), g["div_time"]) < FP_TOL
    assert rel(dev.apply_operator("grad_space", g["x_c"]), g["grad_space"]) < FP_TOL
    assert rel(dev.apply_operator("div_space", g["x_s"]), g["div_space"]) < FP_TOL
    assert rel(dev.apply_operator("decouple", g["x_s"], 1.7), g["decouple"]) < FP_TOL
    assert rel(dev.apply_operator("decouple_adjoint", g["x_d"], 1.7), g["decouple_adjoint"]) < FP_TOL
    assert rel(dev.apply_operator("time_avg_adjoint", g["x_t"]), g["decouple_adjoint_time"]) < FP_TOL
    # K x against the assembled N x N space-time CSR of the oracle (sign: K = -Laplacian + eps M)
    K = -O.assemble_spacetime_laplacian(s.T, s.h, s.mass_v, s.L, 1e-2)
    want = K.dot(g["x_c"].reshape(-1)).reshape(g["x_c"].shape)
    assert rel(dev.apply_operator("laplacian_apply", g["x_c"]), want) < 1e-11
    dev.close()


@pytest.mark.parametrize("fname", OPS)
@pytest.mark.parametrize("congestion", [0.0, 0.15])
def test_kkt_objective_norms_a10_a11_a12(fname, congestion):
    s, dev = make_pair(golden(fname), congestion=congestion)
    # state after a step so that dt_phi / dx_phi / dec_B are the ones the closures see
    s.step_q_lambda()
    s.step_multipliers()
    dev.run_phase("q_lambda_mult")
    want = [f() for f in s.kkt_functions()]
    got = dev.kkt(range(7))
    for i in range(7):
        assert abs(got[i][0] - want[i][0]) <= 1e-11 * abs(want[i][0]), i
        if i < 4:
            assert abs(got[i][1] - want[i][1]) <= 1e-11 * abs(want[i][1]), i
        else:
            assert got[i][1] is None
    # single conditions evaluated lazily give the same numbers
    for i in range(7):
        one = dev.kkt([i])
        assert one[i][0] == pytest.approx(got[i][0], rel=1e-14, abs=0.0)
    cost, obj = dev.objective()
    wc, wo = s.objective()
    assert abs(cost - wc) < 1e-12 * abs(wc) and abs(obj - wo) < 1e-12 * abs(wo)
    assert dev.norm_square("mu") == pytest.approx(s.nsq_time(s.mu), rel=1e-13)
    assert dev.norm_square("phi") == pytest.approx(s.nsq_center(s.phi), rel=1e-13)
    assert dev.norm_square("phi", 1) == pytest.approx(s.nsq_time(s.dt_phi), rel=1e-12)
    assert dev.norm_square("phi", 2) == pytest.approx(s.nsq_space(s.dx_phi), rel=1e-12)
    assert dev.norm_square("E") == pytest.approx(s.nsq_space(s.E), rel=1e-13)
    assert dev.norm_square("beta_mid") == pytest.approx(s.nsq_space_dec(s.beta_mid), rel=1e-13)
    dev.close()


@pytest.mark.parametrize("fname", OPS)
def test_scaling_tools_a13(fname):
    s, dev = make_pair(golden(fname))
    s.adjust_penalty(1.35)
    dev.adjust_penalty(1.35)
    dev.set_params(r=s.r)
    for k in ("mu", "E", "beta_fst", "beta_mid", "beta_end"):
        assert rel(dev.download(k), getattr(s, k)) < 1e-15, k
    sz_old = s.sz
    s.scale_z(1.6)
    dev.scale_z(s.sz, 1.0 / s.sz, s.sz)
    assert s.sz == pytest.approx(sz_old * 1.6)
    for k in STATE:
        assert rel(dev.download(k), getattr(s, k)) < 1e-14, k
    dev.close()


def test_full_iterations_track_oracle():
    """20 complete ALM iterations from the reference's start state stay on the oracle's trajectory."""
    from dots_socp_amd.device import DeviceProblem

    g = golden("ops_ico1.npz")
    geom = dict(vertices=g["vertices"], triangles=g["triangles"], mu0=g["mu0"], mu1=g["mu1"])
    T = int(g["n_time"])
    for lap in ("spacetime_pcg", "modal_pcg", "modal_pcg+mg", "modal_pcg+direct"):
        s = O.OracleSolver(T, geom)
        s.scale_z(2.0)
        dev = DeviceProblem(T, geom, lap_solver=lap.split("+")[0])
        if lap.endswith("+mg"):
            assert dev.setup_multigrid(coarsest=6)["levels"] >= 2
        if lap.endswith("+direct"):
            assert dev.setup_frontal(leaf=4)["levels"] >= 3
        dev.scale_z(2.0, 0.5, 2.0)
        dev.set_params(scale_z=2.0, const_d=2.0, norm_d=s.norm_d, cg_tol=1e-11)
        for _ in range(20):
            s.iterate()
        st = dev.step(20)
        assert st.alm_iterations == 20 and st.cg_not_converged == 0
        for k in ("A", "B", "mu", "E", "z_mid", "beta_mid"):
            assert rel(dev.download(k), getattr(s, k)) < 1e-8, (lap, k)
        want = s.kkt_all()
        got = dev.kkt(range(7))
        for i in range(7):
            assert abs(got[i][0] - want[i]) <= 1e-7 * abs(want[i]) + 1e-14, (lap, i)
        dev.close()


def test_lazy_z_mid_steps_are_bit_identical():
    """Iterations that rebuild z_mid on the fly (DOTS_STEP_SKIP_Z_MID, enqueued without waiting) give the same
    iterate bit for bit as stored-z_mid iterations; reading z_mid while it is stale fails loudly."""
    from dots_socp_amd._lib import HipLibraryError
    from dots_socp_amd.device import DeviceProblem

    g = golden("ops_ico1.npz")
    geom = dict(vertices=g["vertices"], triangles=g["triangles"], mu0=g["mu0"], mu1=g["mu1"])
    T = int(g["n_time"])
    out = []
    for lazy in (False, True):
        dev = DeviceProblem(T, geom, lap_solver="modal_pcg")
        dev.setup_frontal(leaf=4)
        dev.scale_z(2.0, 0.5, 2.0)
        dev.set_params(scale_z=2.0, const_d=2.0, congestion=0.05)
        if lazy:
            dev.step_flags(skip_z_mid=True)
            for _ in range(11):
                dev.step(1, wait=False)
            with pytest.raises(HipLibraryError):
                dev.kkt([1])
            with pytest.raises(HipLibraryError):
                dev.download("z_mid")
            assert dev.kkt([0, 2, 3])[0][0] > 0          # conditions that do not read z_mid still work
            dev.step_flags(skip_z_mid=False)
            dev.step(1)
        else:
            dev.step(12)
        out.append((dev.download_all(), dev.kkt(range(7))))
        dev.close()
    for k in out[0][0]:
        assert np.array_equal(out[0][0][k], out[1][0][k]), k
    assert out[0][1] == out[1][1]


def test_right_hand_side_ahead_is_bit_identical_and_dropped_on_changes():
    """DOTS_STEP_RHS_AHEAD: the KKT read-back after a step enqueues the next iteration's right-hand side; the next step starts at
    the solve.  Same iterates bit for bit as without it; a penalty update, a rescaling or an upload in between drops the
    right-hand side (the step computes it again from the changed state)."""
    from dots_socp_amd._lib import HipLibraryError
    from dots_socp_amd.device import DeviceProblem

    g = golden("ops_ico1.npz")
    geom = dict(vertices=g["vertices"], triangles=g["triangles"], mu0=g["mu0"], mu1=g["mu1"])
    T = int(g["n_time"])

    def run(ahead):
        dev = DeviceProblem(T, geom, lap_solver="modal_pcg")
        dev.setup_frontal(leaf=4)
        dev.set_params(congestion=0.05)
        res = []
        for k in range(12):
            reads = k % 3 != 1
            dev.step_flags(skip_z_mid=not reads, rhs_ahead=ahead and reads)
            dev.step(1, wait=False)
            if reads:
                res.append(dev.kkt([0, 2, 3]))          # with the flag: followed on the stream by the next right-hand side
                res.append(dev.kkt([4]))                # a second read-back of the same iteration launches nothing more
            if k == 3:
                dev.adjust_penalty(1.7)                 # changes r and five arrays: the right-hand side ahead is dropped
                dev.set_params(r=1.7)
            if k == 6:
                dev.scale_z(2.0, 0.5, 2.0)
                dev.set_params(scale_z=2.0, const_d=2.0)
            if k == 8:
                dev.upload("mu", dev.download("mu") * 1.01)
        dev.step_flags(skip_z_mid=False)
        dev.step(1)
        out = (dev.download_all(), dev.kkt(range(7)), res)
        dev.close()
        return out

    a, b = run(False), run(True)
    for k in a[0]:
        assert np.array_equal(a[0][k], b[0][k]), k
    assert a[1] == b[1] and a[2] == b[2]
    # the flag cannot be combined with is_palm's step 0
    dev = DeviceProblem(T, geom, lap_solver="modal_pcg")
    dev.setup_frontal(leaf=4)
    with pytest.raises(HipLibraryError):
        dev.step_flags(palm=True, rhs_ahead=True)
    dev.close()
    # ... and is a no-op without the direct solver
    dev = DeviceProblem(T, geom, lap_solver="modal_pcg")
    dev.set_params(cg_tol=1e-10)
    dev.step_flags(rhs_ahead=True)
    dev.step(1)
    assert np.isfinite(dev.kkt([0])[0][0])
    dev.close()


@pytest.mark.parametrize("fixture,T", [("ops_ico1.npz", None), ("ops_torus8x6.npz", None), ("ops_ico1.npz", 12), ("ops_torus8x6.npz", 31),
                                       ("ops_refplane4.npz", 63), ("ops_ico1.npz", 64), ("ops_torus8x6.npz", 127)])
def test_carried_gathers_are_bit_identical_and_dropped_on_changes(fixture, T):
    """DOTS_STEP_CARRY: steps 2+3 also store, per corner, the sums the next right-hand side and cone projection would gather from
    B, E and beta_mid; the next step streams them instead (k_q_lambda_mult_carry -> soc_element2 / rhs_value2 <CARRIED>).  Same iterates
    bit for bit as without the flag -- enqueued and timed steps, with the right-hand side ahead, odd and even numbers of nodes, every
    time pitch up to 128 (LDS and MFMA transforms) -- and a penalty update, a rescaling, an upload or a single phase in between
    drops what was carried (the step gathers from the changed arrays)."""
    from dots_socp_amd.device import DeviceProblem

    g = golden(fixture)
    geom = dict(vertices=g["vertices"], triangles=g["triangles"], mu0=g["mu0"], mu1=g["mu1"])
    T = int(g["n_time"]) if T is None else T

    def run(carry):
        dev = DeviceProblem(T, geom, lap_solver="modal_pcg")
        dev.setup_frontal(leaf=4)
        dev.scale_z(2.0, 0.5, 2.0)
        dev.set_params(scale_z=2.0, const_d=2.0, congestion=0.05)
        res = []
        for k in range(16):
            reads = k % 3 != 1
            dev.step_flags(skip_z_mid=not reads, rhs_ahead=reads and k % 2 == 0, carry=carry and k != 9)
            if k in (4, 11):
                dev.step(1)                             # the synchronous (timed) path: right-hand side, solve, projection + inverse transform
            else:
                dev.step(1, wait=False)
            if reads:
                res.append(dev.kkt([0, 2, 3]))
            if k == 3:
                dev.adjust_penalty(1.7)                 # changes r and five arrays: the carried sums are dropped
                dev.set_params(r=1.7)
            if k == 6:
                dev.scale_z(3.0, 1.0 / 3.0, 3.0)
                dev.set_params(scale_z=3.0, const_d=3.0)
            if k == 8:
                dev.upload("beta_mid", dev.download("beta_mid") * 1.01)
            if k == 12:
                dev.run_phase("q_lambda_mult")          # a single phase (no carry): B, E and beta_mid move
        dev.step_flags(skip_z_mid=False)
        dev.step(1)
        out = (dev.download_all(), dev.kkt(range(7)), res)
        dev.close()
        return out

    a, b = run(False), run(True)
    for k in a[0]:
        assert np.array_equal(a[0][k], b[0][k]), k
    assert a[1] == b[1] and a[2] == b[2]


@pytest.mark.parametrize("fixture,T", [("ops_ico1.npz", None), ("ops_torus8x6.npz", 31), ("ops_refplane4.npz", 64), ("ops_ico1.npz", 127)])
@pytest.mark.parametrize("congestion", [0.0, 0.05])
def test_kkt_sums_formed_by_steps_2_and_3(fixture, T, congestion):
    """DOTS_STEP_KKT_SUMS: steps 2+3 accumulate the sums of Prim(phi, q), Prim(q, z), Dual(beta), Comp(rho, cong.) from their registers;
    dots_kkt then reduces them (plus one vertex pass for Dual(alpha)) instead of reading the state again.  Same residuals to
    rounding as the stand-alone kernels, with and without the carry mapping, alone / together / mixed with conditions 4 and 5;
    the iterate itself does not depend on the flag; a change of state or parameters in between drops the sums."""
    from dots_socp_amd.device import DeviceProblem

    g = golden(fixture)
    geom = dict(vertices=g["vertices"], triangles=g["triangles"], mu0=g["mu0"], mu1=g["mu1"])
    T = int(g["n_time"]) if T is None else T

    def run(fused):
        dev = DeviceProblem(T, geom, lap_solver="modal_pcg")
        dev.setup_frontal(leaf=4)
        dev.scale_z(2.0, 0.5, 2.0)
        dev.set_params(scale_z=2.0, const_d=2.0, congestion=congestion)
        res = []
        for k in range(8):
            dev.step_flags(carry=k % 2 == 0, kkt_sums=fused)
            dev.step(1, wait=False)
            if k == 0:
                res += [dev.kkt([i]) for i in (6, 2, 0, 3, 1)]             # one at a time, the validator's order
            elif k == 1:
                res.append(dev.kkt([0, 1, 2, 3]))                            # a penalty update's request
            elif k == 2:
                res.append(dev.kkt([0, 1, 2, 3, 6]))
                res.append(dev.kkt(range(7)))                                # conditions 4, 5: the stand-alone kernels for all seven
            elif k == 3:
                dev.adjust_penalty(1.3)
                dev.set_params(r=1.3)
                res.append(dev.kkt([0, 1, 3, 6]))                            # after a change of state: evaluated from the arrays
            elif k == 4:
                res.append(dev.kkt([1]))
                res.append(dev.kkt([4, 5]))
                res.append(dev.kkt([3, 6]))
        out = (dev.download_all(), res)
        dev.close()
        return out

    a, b = run(False), run(True)
    for k in a[0]:
        assert np.array_equal(a[0][k], b[0][k]), k
    assert len(a[1]) == len(b[1])
    for x, y in zip(a[1], b[1]):
        assert x.keys() == y.keys()
        for i in x:
            for p, q in zip(x[i], y[i]):
                assert (p is None) == (q is None)
                if p is not None:
                    assert abs(p - q) <= 1e-12 * abs(p) + 1e-300, (i, p, q)


@pytest.mark.parametrize("fixture,T", [("ops_ico1.npz", None), ("ops_torus8x6.npz", 31), ("ops_refplane4.npz", 63), ("ops_ico1.npz", 128)])
def test_penalty_division_left_to_the_next_iteration(fixture, T, monkeypatch):
    """dots_adjust_penalty does not divide the five dual arrays at once (solver_socp.py:367-371): the next iteration's kernels divide
    as they read and steps 2+3 write the arrays back divided; any other access (residuals, a download, norms, another update, a step
    whose kernels cannot apply it) carries the division out first.  DOTS_LAZY_DIV=0 divides at once: the same numbers bit for bit."""
    from dots_socp_amd.device import DeviceProblem

    g = golden(fixture)
    geom = dict(vertices=g["vertices"], triangles=g["triangles"], mu0=g["mu0"], mu1=g["mu1"])
    T = int(g["n_time"]) if T is None else T

    def run(lazy):
        monkeypatch.setenv("DOTS_LAZY_DIV", lazy)
        dev = DeviceProblem(T, geom, lap_solver="modal_pcg")
        dev.setup_frontal(leaf=4)
        dev.scale_z(2.0, 0.5, 2.0)
        dev.set_params(scale_z=2.0, const_d=2.0, congestion=0.05)
        res, r = [], 1.0
        for k in range(14):
            reads = k % 3 == 2
            dev.step_flags(skip_z_mid=not reads, carry=k not in (7, 10), kkt_sums=reads)
            if k == 12:
                dev.step(1)                                  # the synchronous path carries a pending division out first
            else:
                dev.step(1, wait=False)
            if reads:
                res.append(dev.kkt([0, 1, 2, 3]))
            if k in (2, 5, 6, 9, 11):                        # (6: applied by the kernels of a quiet iteration; 9: by a step without carry)
                f = (1.7, 0.6, 1.3, 2.1, 0.8)[(2, 5, 6, 9, 11).index(k)]
                r *= f
                dev.adjust_penalty(f)
                dev.set_params(r=r)
                if k == 5:
                    res.append(dev.kkt([3]))                 # residuals right after the update: the division is carried out for them
                if k == 11:
                    dev.adjust_penalty(1.1)                  # two updates in a row
                    r *= 1.1
                    dev.set_params(r=r)
                    res.append(float(np.abs(dev.download("beta_mid")).sum()))
        dev.step_flags(skip_z_mid=False)
        dev.step(1)
        out = (dev.download_all(), dev.kkt(range(7)), res)
        dev.close()
        return out

    a, b = run("0"), run("1")
    for k in a[0]:
        assert np.array_equal(a[0][k], b[0][k]), k
    assert a[1] == b[1] and a[2] == b[2]


@pytest.mark.parametrize("fixture,T", [("ops_ico1.npz", None), ("ops_torus8x6.npz", 30), ("ops_refplane4.npz", 63)])
def test_z_mid_on_demand_equals_the_stored_one(fixture, T, monkeypatch):
    """Iterations after which z_mid may be read do not store it: steps 2+3 write the new B / beta_mid into alternate buffers and z_mid is
    rebuilt from the OLD ones when something asks for it (download, Prim(q, z) through the stand-alone kernels, norms, a rescaling,
    is_palm's step 0, a single phase) -- with a penalty division pending during the step too.  DOTS_ZMID_DEFER=0 stores it as before:
    the same numbers bit for bit, whatever is asked for and in whatever order."""
    from dots_socp_amd.device import DeviceProblem

    g = golden(fixture)
    geom = dict(vertices=g["vertices"], triangles=g["triangles"], mu0=g["mu0"], mu1=g["mu1"])
    T = int(g["n_time"]) if T is None else T

    def run(defer):
        monkeypatch.setenv("DOTS_ZMID_DEFER", defer)
        dev = DeviceProblem(T, geom, lap_solver="modal_pcg")
        dev.setup_frontal(leaf=4)
        dev.scale_z(2.0, 0.5, 2.0)
        dev.set_params(scale_z=2.0, const_d=2.0, congestion=0.05)
        res, r = [], 1.0
        for k in range(14):
            reads = k % 2 == 1
            dev.step_flags(skip_z_mid=not reads and k != 12, carry=True, kkt_sums=reads and k != 9, palm=k == 12)
            dev.step(1, wait=False)
            if k == 1:
                res.append(dev.download("z_mid").copy())                 # rebuilt for the download
                res.append(dev.kkt([1]))                                   # (already materialised)
            if k == 3:
                res.append(dev.kkt([0, 1, 2, 3]))                          # the sums steps 2+3 left: z_mid is not touched
                res.append(dev.kkt(range(7)))                              # conditions 4, 5: the stand-alone kernels read z_mid
                r *= 1.6
                dev.adjust_penalty(1.6)
                dev.set_params(r=r)
            if k == 5:                                                     # the step applied the pending division itself
                res.append(dev.norm_square("z_mid"))
                res.append(dev.download("beta_mid").copy())
            if k == 7:
                dev.scale_z(3.0, 1.0 / 3.0, 3.0)                           # scales z_mid: rebuilt first
                dev.set_params(scale_z=3.0, const_d=3.0)
                res.append(dev.download("z_mid").copy())
            if k == 9:
                res.append(dev.kkt([1, 3]))                                # no fused sums on this step: the stand-alone kernels
            if k == 11:
                pass                                                       # nobody asks: the next step (is_palm: step 0 reads z_mid) must rebuild it
            if k == 13:
                res.append(dev.download("B").copy())
        dev.step_flags(skip_z_mid=False)
        dev.step(1)
        out = (dev.download_all(), dev.kkt(range(7)), res)
        dev.close()
        return out

    a, b = run("0"), run("1")
    for k in a[0]:
        assert np.array_equal(a[0][k], b[0][k]), k
    assert a[1] == b[1]
    for x, y in zip(a[2], b[2]):
        if isinstance(x, np.ndarray):
            assert np.array_equal(x, y)
        else:
            assert x == y


def test_carry_flag_is_a_hint():
    """DOTS_STEP_CARRY is ignored without the direct solver and with is_palm's step 0 (which moves B before the right-hand side)."""
    from dots_socp_amd.device import DeviceProblem

    g = golden("ops_ico1.npz")
    geom = dict(vertices=g["vertices"], triangles=g["triangles"], mu0=g["mu0"], mu1=g["mu1"])
    T = int(g["n_time"])
    dev = DeviceProblem(T, geom, lap_solver="modal_pcg")
    dev.set_params(cg_tol=1e-10)
    dev.step_flags(carry=True)
    dev.step(2)
    assert np.isfinite(dev.kkt([0])[0][0])
    dev.close()
    out = []
    for carry in (False, True):
        dev = DeviceProblem(T, geom, lap_solver="modal_pcg")
        dev.setup_frontal(leaf=4)
        dev.step_flags(palm=True, carry=carry)
        dev.step(5)
        out.append(dev.download_all())
        dev.close()
    for k in out[0]:
        assert np.array_equal(out[0][k], out[1][k]), k


@pytest.mark.parametrize("fixture", ["ops_ico1.npz", "ops_torus8x6.npz"])
def test_kkt_sums_with_two_nodes_per_lane(fixture, monkeypatch):
    """The KKT kernels take two nodes per lane on one GPU (kkt_vertex_body2 / kkt_triangle_body2); DOTS_KKT_TWO=0 keeps the one-node
    bodies (which the time slabs use): same residuals to rounding (the sums are formed in a different order), every condition
    alone and all together, odd and even numbers of nodes."""
    from dots_socp_amd.device import DeviceProblem

    g = golden(fixture)
    geom = dict(vertices=g["vertices"], triangles=g["triangles"], mu0=g["mu0"], mu1=g["mu1"])
    for T in (int(g["n_time"]), int(g["n_time"]) + 1):
        res = []
        for two in ("0", "1"):
            monkeypatch.setenv("DOTS_KKT_TWO", two)
            dev = DeviceProblem(T, geom, lap_solver="modal_pcg")
            for name in ("A", "lambda_c", "mu", "B", "E", "phi", "z_fst", "z_mid", "z_end", "beta_fst", "beta_mid", "beta_end"):
                dev.upload(name, np.random.default_rng(sum(map(ord, name))).standard_normal(dev.shape(name)))
            dev.set_params(r=1.3, scale_z=1.7, const_d=0.9, congestion=0.05)
            out = [dev.kkt([i]) for i in range(7)] + [dev.kkt(range(7))]
            res.append(out)
            dev.close()
        for a, b in zip(*res):
            assert a.keys() == b.keys()
            for i in a:
                for x, y in zip(a[i], b[i]):
                    assert (x is None) == (y is None)
                    if x is not None:
                        assert abs(x - y) <= 1e-12 * max(abs(x), 1e-300), (i, x, y)
