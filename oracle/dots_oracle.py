"""CPU oracle for the DOTs-SOCP ALM hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT.

A plain numpy / scipy.sparse restatement of the reference algorithm
(`/root/reference/dot_surface_socp/socp/solver_socp.py:25-1065` and the files it
calls).  It exists to *check* the HIP path, never to serve it: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import this module.  Nothing under ``dots_socp_amd/`` imports it.

Parity status: PINNED.  ``tests/golden/*.npz`` hold input/output vectors recorded
from the reference itself in the build container (``tests/golden/make_golden.py``
through ``tests/golden/ref_shim.py``); ``tests/test_oracle_golden.py`` checks this
file against every one of them (per-function vectors, k-iteration states, full-run
KKT trajectories, stopping iteration and transport cost).

Arrays use the reference's own layouts:
    phi                     (T+1, V)         time nodes x vertices
    A, lambda_c, mu,
    z_fst, z_end,
    beta_fst, beta_end      (T,   V)         time intervals x vertices
    B, E                    (T+1, F, 3)
    z_mid, beta_mid         (T, 2, 3, F, 3)  [interval, end s, corner k, triangle, xyz]
All arithmetic is IEEE fp64.

Each function cites the reference lines it restates.
"""
from __future__ import annotations

import math
import time
from contextlib import contextmanager

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

SQRT3 = math.sqrt(3.0)

KKT_LABELS = [
    "SOC & Org : Primal Feasibility (q)",
    "SOC       : Primal Feasibility (z)",
    "SOC & Org : Dual Feasibility (alpha)",
    "SOC       : Dual Feasibility (beta)",
    "      Org : ||rho - Pi+(rho + Fq)||",
    "      Org : ||m - rho o B||",
    "      Org : ||cong. rho - lambda_c||",
]
KKT_SHORT = [
    "Prim(phi, q)", "Prim(q, z)", "Dual(alpha)", "Dual(beta)",
    "Comp(rho, f(q))", "Comp(m, rho o B)", "Comp(rho, cong.)",
]


# --------------------------------------------------------------------------- #
# a15  operator assembly   (utils/surface_pre_computations_socp.py:11-132)
# --------------------------------------------------------------------------- #
def triangle_quantities(vertices, triangles):
    """Areas, corner angles and hat-function gradients, vectorised over triangles.

    Restates geometricQuantities (surface_pre_computations_socp.py:11-39): edge
    vectors v01, v12, v20; area = |v01 x v12| / 2 (:25); the angle at corner k is
    the arccos of the normalised dot product of the two edges leaving it (:27-29);
    the hat gradient at corner k is the altitude vector onto the opposite edge
    divided by its squared length (:31-37).
    """
    v = np.asarray(vertices, dtype=np.float64)
    t = np.asarray(triangles)
    p0, p1, p2 = v[t[:, 0]], v[t[:, 1]], v[t[:, 2]]
    e01, e12, e20 = p1 - p0, p2 - p1, p0 - p2

    def dot(a, b):
        return np.einsum("ij,ij->i", a, b)

    def nrm(a):
        return np.sqrt(dot(a, a))

    area = nrm(np.cross(e01, e12)) / 2.0
    ang = np.stack(
        [
            np.arccos(dot(e01, -e20) / (nrm(e01) * nrm(e20))),
            np.arccos(dot(e12, -e01) / (nrm(e12) * nrm(e01))),
            np.arccos(dot(e20, -e12) / (nrm(e20) * nrm(e12))),
        ],
        axis=1,
    )
    alt = np.stack(
        [
            -e01 + e12 * (dot(e01, e12) / dot(e12, e12))[:, None],
            -e12 + e20 * (dot(e12, e20) / dot(e20, e20))[:, None],
            -e20 + e01 * (dot(e20, e01) / dot(e01, e01))[:, None],
        ],
        axis=1,
    )  # (F, corner, xyz)
    hat = alt / np.einsum("fkc,fkc->fk", alt, alt)[:, :, None]
    return area, ang, hat


def surface_matrices(n_vertices, triangles, angles, hat):
    """Gradient (3F x V), divergence (= -G^T) and cotangent Laplacian (V x V).

    Restates geometricMatrices (surface_pre_computations_socp.py:42-86): row
    3f+c of G holds component c of the three hat gradients of triangle f
    (:55-63); the Laplacian gets +cot/2 on the edge opposite each corner and
    -cot/2 on the two diagonal entries of that edge (:68-84).
    """
    t = np.asarray(triangles)
    n_tri = t.shape[0]
    rows = (3 * np.arange(n_tri)[:, None, None] + np.arange(3)[None, None, :]) + np.zeros((1, 3, 1), dtype=np.int64)
    cols = np.broadcast_to(t[:, :, None], (n_tri, 3, 3))
    grad = sp.coo_matrix((hat.reshape(-1), (rows.reshape(-1), cols.reshape(-1))), shape=(3 * n_tri, n_vertices)).tocsr()
    div = (-grad.transpose()).tocsr()

    w = 0.5 * np.cos(angles) / np.sin(angles)
    ii, jj, vv = [], [], []
    for k in range(3):
        a, b = t[:, (k + 1) % 3], t[:, (k + 2) % 3]
        ii += [a, b, a, b]
        jj += [b, a, a, b]
        vv += [w[:, k], w[:, k], -w[:, k], -w[:, k]]
    lap = sp.coo_matrix((np.concatenate(vv), (np.concatenate(ii), np.concatenate(jj))), shape=(n_vertices, n_vertices)).tocsr()
    return grad, div, lap


def corner_maps(n_vertices, triangles, area_triangles):
    """Vertex<->corner incidence (corner index i = k*F + f  <->  vertex triangles[f, k]).

    Restates trianglesToVertices (surface_pre_computations_socp.py:88-132):
    returns the area-weighted corner->vertex map (V x 3F), the raw vertex areas
    (sum of incident triangle areas), the 0/1 vertex->corner map (3F x V) and the
    vertex area seen from every corner.
    """
    t = np.asarray(triangles)
    n_tri = t.shape[0]
    corner_vertex = t.T.reshape(-1)  # i = k*F + f
    corner_area = np.tile(area_triangles, 3)
    idx = np.arange(3 * n_tri)
    c2v_area = sp.coo_matrix((corner_area, (corner_vertex, idx)), shape=(n_vertices, 3 * n_tri)).tocsr()
    area_v_raw = c2v_area.dot(np.ones(3 * n_tri))
    v2c = sp.coo_matrix((np.ones(3 * n_tri), (idx, corner_vertex)), shape=(3 * n_tri, n_vertices)).tocsr()
    return c2v_area, area_v_raw, v2c, area_v_raw[corner_vertex]


# --------------------------------------------------------------------------- #
# a4-a6  stencils and incidence operators   (solver_socp.py:881-974)
# --------------------------------------------------------------------------- #
def grad_time(h, x):
    """solver_socp.py:881-884."""
    return np.diff(x, axis=0) / h


def div_time(h, x):
    """Negative adjoint of grad_time (solver_socp.py:886-896)."""
    out = np.zeros((x.shape[0] + 1, x.shape[1]))
    out[:-1] += x / h
    out[1:] -= x / h
    return out


def grad_space(grad_mat, n_tri, x):
    """solver_socp.py:898-907: (T+1, V) -> (T+1, F, 3)."""
    return grad_mat.dot(x.T).T.reshape(x.shape[0], n_tri, 3)


def div_space(div_mat, x):
    """solver_socp.py:909-921: (T+1, F, 3) -> (T+1, V)."""
    return div_mat.dot(x.reshape(x.shape[0], -1).T).T


def decouple(b, scale_z=1.0):
    """L: (T+1, F, 3) -> (T, 2, 3, F, 3)   (solver_socp.py:923-942)."""
    s = (scale_z / SQRT3) * b
    out = np.empty((b.shape[0] - 1, 2, 3) + b.shape[1:])
    out[:, 0] = s[:-1, None]
    out[:, 1] = s[1:, None]
    return out


def decouple_adjoint(x, scale_z=1.0):
    """L^T: (T, 2, 3, F, 3) -> (T+1, F, 3)   (solver_socp.py:944-959)."""
    s = (scale_z / SQRT3) * x.sum(axis=2)
    out = np.zeros((x.shape[0] + 1,) + x.shape[3:])
    out[:-1] = s[:, 0]
    out[1:] += s[:, 1]
    return out


def time_average_adjoint(x):
    """(T, V) -> (T+1, V): out[t] = (x[t-1] + x[t]) / 2 with zeros outside.

    Restates decouple_adjoint_time (solver_socp.py:961-974): zero-pad one row at
    the end, correlate with [1/2, 1/2] (scipy origin = len // 2 = 1).
    """
    out = np.zeros((x.shape[0] + 1,) + x.shape[1:])
    out[:-1] += 0.5 * x
    out[1:] += 0.5 * x
    return out


# --------------------------------------------------------------------------- #
# a3  space-time Laplacian inverse   (utils/laplacian_inverse_socp.py:11-61)
# --------------------------------------------------------------------------- #
def time_neumann_laplacian(n_time, h):
    """(T+1) x (T+1) Neumann second difference / h^2  (laplacian_inverse_socp.py:15-26)."""
    n = n_time + 1
    lt = np.zeros((n, n))
    i = np.arange(1, n_time)
    lt[i, i] = -2.0
    lt[i, i + 1] = 1.0
    lt[i, i - 1] = 1.0
    lt[0, 0], lt[0, 1] = -1.0, 1.0
    lt[-1, -1], lt[-1, -2] = -1.0, 1.0
    return lt / (h * h)


class LaplacianInverse:
    """Eigen-decomposition in time + one sparse LU per mode in space
    (laplacian_inverse_socp.py:31-41 setup, :52-61 apply)."""

    def __init__(self, n_time, h, mass_v, lap_space, eps=0.0):
        self.eigval, self.eigvec = np.linalg.eigh(time_neumann_laplacian(n_time, h))
        self.lu = [
            spla.splu((lap_space + (lam - eps) * sp.diags([mass_v], [0])).tocsc())
            for lam in self.eigval
        ]

    def __call__(self, rhs):
        hat = self.eigvec.T @ rhs
        sol = np.empty_like(hat)
        for a, lu in enumerate(self.lu):
            sol[a] = lu.solve(hat[a])
        return self.eigvec @ sol


def assemble_spacetime_laplacian(n_time, h, mass_v, lap_space, eps=0.0):
    """The operator LaplacianInverse inverts, as one N x N CSR, N = (T+1) V, row = t*V + v:
    kron(L_time, diag(mass)) + kron(I, L_space) - eps*kron(I, diag(mass)).
    (Used by tests to check the HIP PCG operator; not part of the reference.)"""
    lt = sp.csr_matrix(time_neumann_laplacian(n_time, h))
    m = sp.diags([mass_v], [0])
    eye_t = sp.identity(n_time + 1)
    return (sp.kron(lt, m) + sp.kron(eye_t, lap_space) - eps * sp.kron(eye_t, m)).tocsr()


# --------------------------------------------------------------------------- #
# a5/a14  host control policies   (utils/admm_tools.py:19-171, condition_validator*.py)
# --------------------------------------------------------------------------- #
class PenaltyPolicy:
    """AdjustAdmmParam (admm_tools.py:19-114)."""

    _TABLE = [(50, 2.00), (35, 1.75), (20, 1.60), (10, 1.40), (5, 1.35), (3, 1.32),
              (2.5, 1.28), (2, 1.26), (1.5, 1.20), (1.2, 1.10)]

    def __init__(self):
        self.last_it = -1
        self.hi, self.lo = 1e3, 1e-3
        self.z_rescales = 0

    def is_to_adjust(self, it):  # :30-52
        gap = it - self.last_it
        due = ((it < 20 and gap >= 3) or (it < 50 and gap >= 7) or (it < 100 and gap >= 11)
               or (it < 200 and gap >= 17) or (it < 500 and gap >= 31) or gap >= 43)
        if due:
            self.last_it = it
        return due

    @classmethod
    def factor(cls, ratio):  # :65-95
        inv = ratio < 1.0
        if inv:
            ratio = 1.0 / ratio
        f = 1.0
        for bound, val in cls._TABLE:
            if ratio > bound:
                f = val
                break
        return 1.0 / f if inv else f

    def updated(self, sigma, ratio):  # :54-62
        return max(min(sigma * self.factor(ratio), self.hi), self.lo)

    @staticmethod
    def is_to_scale(it):  # :98-105
        return it == 10 or it == 50 or it % 100 == 50

    def is_to_rescale_z(self, it, last_row, min_it=100, max_times=1, tol=5e-3):  # :107-114
        # builtin max on purpose: NaN entries make the comparison order-dependent, as in the reference
        if it >= min_it and self.z_rescales < max_times and max(list(last_row)) < tol:
            self.z_rescales += 1
            return True
        return False

    @staticmethod
    def scale_factors(prim_norm, dual_norm):  # :117-171
        return float(np.max(prim_norm)), float(np.max(dual_norm))


def max_skip_none(values):
    """condition_validator.py:165-168."""
    vals = [v for v in values if v is not None]
    return max(vals) if vals else None


class LazyKKT:
    """Circular-queue validator + adaptive interval, collapsed into one object.

    Restates ConditionValidator.validate (condition_validator.py:236-331) with the
    forced queue order (solver_socp.py:644), ErrorConditionWrapperEx (:105-162),
    the collector (:171-191) and AdaptiveValidatorWrapper
    (condition_validator_wrapper.py:45-131).
    ``funcs[i]()`` returns the pair [value, value_at_unit_scale] of condition i.
    """

    def __init__(self, funcs, tol, order, min_interval=1, max_interval=37):
        self.funcs = funcs
        self.tol = tol
        self.n = len(funcs)
        self.order = list(order)  # queue position -> condition
        self.pos = {c: i for i, c in enumerate(self.order)}
        self.front = 0
        self.last = [[None, None] for _ in range(self.n)]
        self.interval, self.counter = 1, 0
        self.min_interval, self.max_interval = min_interval, max_interval

    # -- AdaptiveValidatorWrapper
    def reset_counter(self):
        self.counter = 0

    def set_error(self, error, tol):
        ratio = float(np.max(np.array([error]) / np.maximum(np.array([tol]), 1e-10)))
        if ratio <= 1.0:
            self.interval = self.min_interval
            return
        lg = np.log10(ratio)
        if lg > 1.0:
            self.interval = self.max_interval
        else:
            self.interval = max(self.min_interval, int(self.min_interval + lg * (self.max_interval - self.min_interval)))

    def validate(self, required=None):
        due = (self.counter % self.interval) == 0
        self.counter += 1
        if not (due or required):
            return False, {}
        return self._validate(required or [])

    # -- ConditionValidator
    def _check(self, cond):
        try:
            val = self.funcs[cond]()
            self.last[cond] = val
            return bool(val[0] < self.tol)
        except Exception:  # reference swallows and reports inf (condition_validator.py:145-149)
            self.last[cond] = [float("inf"), float("inf")]
            return False

    def _validate(self, required):
        checked, failing = set(), []

        def check(cond):
            if cond in checked:
                return True
            checked.add(cond)
            ok = self._check(cond)
            if not ok:
                failing.append(cond)
            return ok

        req_ok = [check(c) for c in required]
        passed = False
        if all(req_ok) and len(checked) < self.n:
            start = self.front
            while len(checked) < self.n:
                cond = self.order[self.front % self.n]
                if cond not in checked and not check(cond):
                    break
                self.front = (self.front + 1) % self.n
                if self.front == start:
                    passed = True
                    break
            # NB: when the 7th check happens before the queue is back at `start`, the loop ends on its
            # condition with passed == False although nothing failed -- same as the reference (:291-303).
        elif all(req_ok):
            passed = True
        info = {"all_passed": passed, "num_conditions_passed": len(checked), "failing_conditions": failing}
        return passed, info

    def collect(self):
        """get_and_reset of all conditions in original order -> (org, scaled) lists."""
        out = self.last
        self.last = [[None, None] for _ in range(self.n)]
        return [o[0] for o in out], [o[1] for o in out]


class History:
    """RunningHistory subset (admm_tools.py:174-251, 399-442)."""

    def __init__(self, max_records, labels=KKT_LABELS, short=KKT_SHORT, name="SOCP"):
        self.kkt_labels, self.kkt_short_labels, self.name = labels, short, name
        self._n, self._max = 0, max_records
        self.kkt_errors = np.full((max_records, len(labels)), np.inf)
        self.kkt_iteration = np.full(max_records, np.inf)
        self.kkt_time = np.full(max_records, np.inf)
        self.last_record_it = -1
        self.running_time = np.inf
        self.steps_time, self.history = {}, {}
        self._t0 = np.inf

    def start(self):
        self._t0 = time.perf_counter()

    def get_running_time(self):
        return time.perf_counter() - self._t0

    def end(self):
        self.running_time = time.perf_counter() - self._t0
        n = self._n
        self.kkt_errors, self.kkt_iteration, self.kkt_time = self.kkt_errors[:n], self.kkt_iteration[:n], self.kkt_time[:n]
        for k in self.history:
            self.history[k] = self.history[k][:n]

    @contextmanager
    def timer(self, tag):
        t0 = time.perf_counter()
        yield
        self.steps_time[tag] = self.steps_time.get(tag, 0.0) + time.perf_counter() - t0

    def record(self, current_it, kkt_errors, history=None):
        if current_it < self.last_record_it:
            raise ValueError("iteration went backwards")
        if current_it == self.last_record_it:
            self._n -= 1  # overwrite the last row (admm_tools.py:411-414)
        if self._n >= self._max:
            raise ValueError("history full")
        self.last_record_it = current_it
        self.kkt_errors[self._n] = np.array([np.nan if e is None else e for e in kkt_errors], dtype=float)
        self.kkt_iteration[self._n] = current_it
        self.kkt_time[self._n] = time.perf_counter() - self._t0
        if history:
            for k, v in history.items():
                if k not in self.history:
                    self.history[k] = np.full_like(self.kkt_iteration, np.inf)
                self.history[k][self._n] = v
        self._n += 1

    def get_current_kkt_errors(self):
        if self._n == 0:
            return np.full(self.kkt_errors.shape[1], np.inf)
        return self.kkt_errors[self._n - 1]


# --------------------------------------------------------------------------- #
# a1-a13  the solver
# --------------------------------------------------------------------------- #
class OracleSolver:
    """State + steps of the reference solver, one method per reference step.

    Setup restates solver_socp.py:96-313; see the method docstrings for the loop.
    """

    STATE = ("phi", "A", "B", "lambda_c", "z_fst", "z_mid", "z_end", "mu", "E", "beta_fst", "beta_mid", "beta_end")

    def __init__(self, n_time, geometry, congestion=0.0, eps=0.0, tau=1.9, init_solution=None):
        self.T = T = int(n_time)
        self.h = 1.0 / T
        self.tau, self.eps, self.congestion = tau, eps, congestion
        self.r = 1.0
        vertices = np.asarray(geometry["vertices"], dtype=np.float64)
        self.tri = tri = np.asarray(geometry["triangles"])
        self.V, self.F = V, F = vertices.shape[0], tri.shape[0]

        self.area_f, ang, self.hat = triangle_quantities(vertices, tri)
        self.G, self.Dv, self.L = surface_matrices(V, tri, ang, self.hat)
        c2v_area, area_raw, v2c, area_corner = corner_maps(V, tri, self.area_f)
        self.mass_v = area_raw / 3.0                    # solver_socp.py:112
        mass_corner = area_corner / 3.0                 # :113
        self.area_mesh = float(np.sum(self.area_f))

        eyeT = sp.identity(T, format="csr")
        self.v2c_T = sp.kron(eyeT, v2c).tocsr()                                   # :161
        self.c2v_one_T = sp.kron(eyeT, v2c.transpose()).tocsr()                   # :170
        self.c2v_area_T = sp.kron(eyeT, c2v_area).tocsr()                         # :168
        third = v2c[:F] + v2c[F:2 * F] + v2c[2 * F:]
        self.v2f_third = sp.kron(sp.identity(T + 1, format="csr") / 3.0, third).tocsr()  # :163-166
        self.D = np.sqrt(np.tile(self.area_f, 3) / mass_corner).reshape(3, F)      # :172-180, D[k, f]

        self.lap_inv = LaplacianInverse(T, self.h, self.mass_v, self.L, eps)       # :205-212

        # weights of the four scalar products (:139-157, :215-218)
        self.w_v = self.mass_v[None, :]
        self.w_f = self.area_f[None, :, None]
        self.w_fd = self.area_f[None, None, None, :, None]

        ini = dict(init_solution or {})
        r = self.r
        z = np.zeros
        self.phi = ini.get("phi", z((T + 1, V)))
        self.A = ini.get("A", grad_time(self.h, self.phi))
        self.B = ini.get("B", grad_space(self.G, F, self.phi))
        self.lambda_c = ini.get("lambda_c", z((T, V)))
        self.z_fst = ini.get("z_fst", z((T, V)))
        self.z_end = ini.get("z_end", z((T, V)))
        self.z_mid = ini.get("z_mid", z((T, 2, 3, F, 3)))
        self.beta_fst = (1.0 / r) * ini.get("beta_fst", z((T, V)))
        self.beta_end = (1.0 / r) * ini.get("beta_end", z((T, V)))
        self.beta_mid = (1.0 / r) * ini.get("beta_mid", z((T, 2, 3, F, 3)))
        self.mu = (1.0 / r) * ini.get("mu", r * (self.beta_fst - self.beta_end))
        self.E = (1.0 / r) * ini.get("E", -r * decouple_adjoint(self.beta_mid))     # :239-250
        self.dt_phi = np.array(0.0)
        self.dx_phi = np.array(0.0)
        self.dec_B = z((T, 2, 3, F, 3))  # memo_z_mid of the reference (:262), refreshed in step 3

        self.bnd = z((T + 1, V))                                                   # :267-270
        self.bnd[0] = -np.asarray(geometry["mu0"], dtype=np.float64) / (r * self.h)
        self.bnd[-1] = np.asarray(geometry["mu1"], dtype=np.float64) / (r * self.h)

        self.norm_boundary = r * self.h * math.sqrt(self.nsq_center(self.bnd / self.w_v))   # :296
        self.norm_d = math.sqrt(2 * self.area_mesh)                                           # :297
        m_c = float(np.mean(np.broadcast_to(self.w_v, (T + 1, V))))
        m_t = float(np.mean(np.broadcast_to(self.w_v, (T, V))))
        m_s = float(np.mean(np.broadcast_to(self.w_f, (T + 1, F, 3))))
        m_sd = float(np.mean(np.broadcast_to(self.w_fd, (T, 2, 3, F, 3))))
        self.c_prim_q = float(np.mean([m_t, m_s]))            # :308-313
        self.c_prim_z = float(np.mean([m_t, m_sd, m_t]))
        self.c_dual_alpha = m_c
        self.c_dual_beta = float(np.mean([m_t, m_s]))
        self.c_comp_rho = m_t
        self.c_comp_m = m_s

        self.prim_scale = self.dual_scale = 1.0
        self.d = 1.0          # constant_d (:320)
        self.sz = 1.0         # scale_factor_z (:321)

    # -- a10 weighted squared norms (solver_socp.py:875-878, :215-218)
    def nsq_center(self, a):
        return float(np.sum(a ** 2 * self.w_v)) / (self.T + 1)

    def nsq_time(self, a):
        return float(np.sum(a ** 2 * self.w_v)) / self.T

    def nsq_space(self, a):
        return float(np.sum(a ** 2 * self.w_f)) / (self.T + 1)

    def nsq_space_dec(self, a):
        return float(np.sum(a ** 2 * self.w_fd)) / self.T

    def diag_b(self):
        """assemble_diagonal_b (solver_socp.py:194-202) as a (T+1, 1, 1) column."""
        d = np.full(self.T + 1, 1.0 + 2.0 * self.sz ** 2)
        d[0] = d[-1] = 1.0 + self.sz ** 2
        return d[:, None, None]

    # -- a2  right-hand side of the phi step (solver_socp.py:983-986)
    def laplacian_rhs(self):
        return (
            div_time(self.h, (self.A + self.lambda_c - self.mu) * self.w_v)
            + div_space(self.Dv, (self.B - self.E) * self.w_f)
            - self.bnd
            - self.eps * self.w_v * self.phi
        )

    def step_laplacian(self):
        """Step 1-1 (solver_socp.py:699-700, :976-986)."""
        self.phi[:] = self.lap_inv(self.laplacian_rhs())

    # -- a7  second-order-cone projection (solver_socp.py:988-1042)
    def step_soc_projection(self):
        T, V, F = self.T, self.V, self.F
        w_fst = self.d - self.sz * self.A - self.beta_fst
        w_mid = self.D[None, None, :, :, None] * (decouple(self.B, self.sz) - self.beta_mid)
        w_end = self.d + self.sz * self.A - self.beta_end
        per_corner = (w_mid ** 2).sum(axis=(1, 4))                                   # (T, 3, F)
        nrm = self.c2v_one_T.dot(per_corner.reshape(-1)).reshape(T, V)
        nrm = np.sqrt(nrm + w_end ** 2)
        with np.errstate(divide="ignore", invalid="ignore"):
            lam = np.clip(0.5 * (1.0 + w_fst / nrm), 0.0, 1.0)
        lam_corner = self.v2c_T.dot(lam.reshape(-1)).reshape(T, 3, F) / self.D[None]
        self.z_fst[:] = np.where(lam >= 1.0, w_fst, lam * nrm)
        self.z_mid[:] = lam_corner[:, None, :, :, None] * w_mid
        self.z_end[:] = lam * w_end

    # -- a8  (q, lambda_c) closed form (solver_socp.py:709-714, :1044-1065)
    def step_q_lambda(self, refresh_gradients=True):
        """``refresh_gradients=False``: is_palm's step 0 (solver_socp.py:668-672) -- the closed form with dt_phi / dx_phi
        as they stand (initially the gradients of the initial phi, :253-255)."""
        c, r, sz = self.congestion, self.r, self.sz
        if refresh_gradients:
            self.dt_phi = grad_time(self.h, self.phi)
            self.dx_phi = grad_space(self.G, self.F, self.phi)
        a1 = sz * (1.0 + c * r)
        a2 = 1.0 + 2.0 * sz * a1
        memo_a = self.dt_phi + self.mu
        memo_b = decouple_adjoint(self.z_mid + self.beta_mid, sz)
        self.A[:] = (1.0 / a2) * memo_a + (a1 / a2) * (self.z_end + self.beta_end - self.z_fst - self.beta_fst)
        self.B[:] = (self.dx_phi + self.E + memo_b) / self.diag_b()
        self.lambda_c[:] = (c * r / (1.0 + c * r)) * (memo_a - self.A)

    # -- a9  multiplier update (solver_socp.py:716-722)
    def step_multipliers(self):
        tau, sz, d = self.tau, self.sz, self.d
        self.dec_B = decouple(self.B, sz)
        self.mu += tau * (self.dt_phi - self.A - self.lambda_c)
        self.E += tau * (self.dx_phi - self.B)
        self.beta_fst += tau * (self.z_fst + sz * self.A - d)
        self.beta_mid += tau * (self.z_mid - self.dec_B)
        self.beta_end += tau * (self.z_end - sz * self.A - d)

    def iterate(self):
        """One ALM iteration, steps 1-3 (solver_socp.py:674-722, is_palm=False)."""
        self.step_laplacian()
        self.step_soc_projection()
        self.step_q_lambda()
        self.step_multipliers()

    # -- a13  penalty / scaling tools (solver_socp.py:324-412)
    def adjust_penalty(self, factor):
        self.r *= factor
        for a in (self.mu, self.E, self.bnd, self.beta_fst, self.beta_mid, self.beta_end):
            a /= factor

    def scale_z(self, factor):
        """scale_variable_z (:373-395).  The *cumulative* factor multiplies z, as in the reference."""
        self.sz *= factor
        self.d *= factor
        self.norm_d *= factor
        for a in (self.z_fst, self.z_mid, self.z_end):
            a *= self.sz
        for a in (self.beta_fst, self.beta_mid, self.beta_end):
            a *= 1.0 / self.sz
        self.mu = self.sz * (self.beta_fst - self.beta_end)
        self.E = -decouple_adjoint(self.beta_mid, self.sz)

    def scale_prim_dual(self, prim, dual):
        """scale_prim_dual with explicit factors (:324-365); only reached with is_constant_scaling."""
        if max(prim, dual) / min(prim, dual) > 2.0:
            self.prim_scale *= prim
            self.dual_scale *= dual
            for name in ("phi", "A", "B", "lambda_c", "dt_phi", "dx_phi", "z_fst", "z_mid", "z_end"):
                setattr(self, name, getattr(self, name) / prim)
            f = dual ** 2 / prim
            for name in ("bnd", "mu", "E", "beta_fst", "beta_mid", "beta_end"):
                setattr(self, name, getattr(self, name) / f)
            self.r *= dual / prim
            self.congestion *= dual / prim
            self.d /= prim
            self.norm_d /= prim
            self.norm_boundary /= dual

    def scale_prim_dual_auto(self):
        prim = [
            math.sqrt(self.nsq_time(self.dt_phi) + self.nsq_space(self.dx_phi)),
            math.sqrt(self.nsq_time(self.A) + self.nsq_space(self.B)),
            math.sqrt(self.nsq_time(self.z_fst) + self.nsq_space_dec(self.z_mid) + self.nsq_time(self.z_end)),
        ]
        dual = [
            self.r * math.sqrt(self.nsq_time(self.mu) + self.nsq_space(self.E)),
            self.r * math.sqrt(self.nsq_time(self.beta_fst) + self.nsq_space_dec(self.beta_mid) + self.nsq_time(self.beta_end)),
        ]
        self.scale_prim_dual(*PenaltyPolicy.scale_factors(prim, dual))

    def recovered_solution(self):
        """recorver_scaled_solution (:397-405) on copies."""
        ps, ds, r, sz = self.prim_scale, self.dual_scale, self.r, self.sz
        out = {k: ps * getattr(self, k) for k in ("phi", "A", "B", "lambda_c")}
        out.update({k: (ps / sz) * getattr(self, k) for k in ("z_fst", "z_mid", "z_end")})
        out.update({k: (r * ds) * getattr(self, k) for k in ("mu", "E")})
        out.update({k: (r * sz * ds) * getattr(self, k) for k in ("beta_fst", "beta_mid", "beta_end")})
        return out

    # -- a12  objective (solver_socp.py:417-431, called as at :773-775 / :829-831)
    def objective(self):
        phi = self.prim_scale * self.phi
        lam = self.prim_scale * self.lambda_c
        bnd = (self.dual_scale * self.r) * self.bnd
        cong = self.congestion * self.prim_scale / self.dual_scale
        cost = self.h * (float(np.dot(phi[0], bnd[0])) + float(np.dot(phi[-1], bnd[-1])))
        if cong > 1e-10:
            return cost, cost - self.nsq_time(lam) / (2.0 * cong)
        return cost, cost

    # -- a11  the seven KKT residuals (solver_socp.py:433-559, wired at :589-643)
    def _pair(self, num, const, denom_rest, scale):
        return [num / (const / s + denom_rest) for s in (scale, 1.0)]

    def kkt_primal_q(self):  # :433-450 with the residuals of :592-593
        r_mu = self.dt_phi - self.A - self.lambda_c
        r_e = self.dx_phi - self.B
        nsum = (
            math.sqrt(self.nsq_time(self.dt_phi) + self.nsq_space(self.dx_phi))
            + math.sqrt(self.nsq_time(self.A) + self.nsq_space(self.B))
            + math.sqrt(self.nsq_time(self.lambda_c))
        )
        res = math.sqrt(self.nsq_time(r_mu) + self.nsq_space(r_e))
        return self._pair(res, self.c_prim_q, nsum, self.prim_scale)

    def kkt_primal_z(self):  # :452-464 with :598-600
        r_fst = self.z_fst + self.sz * self.A - self.d
        r_mid = self.sz * (self.z_mid - self.dec_B)
        r_end = self.z_end - self.sz * self.A - self.d
        res = math.sqrt(self.nsq_time(r_fst) + self.nsq_time(r_end) + self.nsq_space_dec(r_mid))
        return self._pair(res, self.c_prim_z, self.norm_d, self.prim_scale)

    def kkt_dual_alpha(self):  # :466-482
        aux = (self.r * self.h) * (
            self.bnd + div_time(self.h, self.mu * self.w_v) + div_space(self.Dv, self.E * self.w_f)
        ) / self.w_v
        res = math.sqrt(self.nsq_center(aux))
        return self._pair(res, self.c_dual_alpha, self.norm_boundary, self.dual_scale)

    def kkt_dual_beta(self):  # :484-503
        a1 = self.sz * (self.beta_end - self.beta_fst)
        a2 = decouple_adjoint(self.beta_mid, self.sz)
        nsum = self.r * (
            math.sqrt(self.nsq_time(self.mu) + self.nsq_space(self.E))
            + math.sqrt(self.nsq_time(a1) + self.nsq_space(a2))
        )
        res = self.r * math.sqrt(self.nsq_time(self.mu + a1) + self.nsq_space(self.E + a2))
        return self._pair(res, self.c_dual_beta, nsum, self.dual_scale)

    def kkt_comp_rho_fq(self):  # :505-526 with :615-619
        rho = (self.dual_scale * self.r) * self.mu
        qa = self.prim_scale * self.A
        qb = self.prim_scale * self.B
        sq = np.sum(np.square(decouple(qb)), axis=(1, 4)).reshape(-1)
        aux = qa + 0.25 * self.c2v_area_T.dot(sq).reshape(self.T, self.V) / self.w_v
        nsum = math.sqrt(self.nsq_time(rho)) + math.sqrt(self.nsq_time(aux))
        res = math.sqrt(self.nsq_time(np.maximum(0.0, aux + rho) - rho))
        return [res / (self.c_comp_rho + nsum), None]

    def kkt_comp_m_rho_b(self):  # :528-547 with :624-628
        m = (self.dual_scale * self.r) * self.E
        rho = (self.dual_scale * self.r) * self.mu
        b = self.prim_scale * self.B
        rho_f = self.v2f_third.dot(time_average_adjoint(rho).reshape(-1)).reshape(self.T + 1, self.F, 1)
        aux = rho_f * b
        nsum = math.sqrt(self.nsq_space(m)) + math.sqrt(self.nsq_space(aux))
        res = math.sqrt(self.nsq_space(aux - m))
        return [res / (self.c_comp_m + nsum), None]

    def kkt_comp_congestion(self):  # :549-559 with :633-636
        rho = (self.dual_scale * self.r) * self.mu
        lam = self.prim_scale * self.lambda_c
        nsum = math.sqrt(self.nsq_time(rho)) + math.sqrt(self.nsq_time(lam))
        res = math.sqrt(self.nsq_time(self.congestion * rho - lam))
        return [res / (self.c_comp_rho + nsum), None]

    def kkt_functions(self):
        return [self.kkt_primal_q, self.kkt_primal_z, self.kkt_dual_alpha, self.kkt_dual_beta,
                self.kkt_comp_rho_fq, self.kkt_comp_m_rho_b, self.kkt_comp_congestion]

    def kkt_all(self):
        return [f()[0] for f in self.kkt_functions()]


def solver_socp(n_time, geometry, congestion=0.0, nit=1000, eps=0.0, tol=1e-4, tau=1.90,
                is_palm=False, is_multi_threads=True, is_z_scaling=True, is_constant_scaling=False,
                check_kkt_step_by_step=False, init_solution=None, tol_checkpoints=None, time_limit=1000,
                trace=None):
    """Same contract as the reference's solver_socp (solver_socp.py:25-41, :855-871).

    ``is_palm``: an extra (q, lambda_c) solve opens every iteration (:668-672).  ``trace``: optional list that
    receives one dict per iteration (decisions taken), used by tests only.
    """
    checkpoints = []
    if tol_checkpoints is not None:  # :85-94
        if not isinstance(tol_checkpoints, list) or not tol_checkpoints:
            raise ValueError("tol_checkpoints must be a non-empty list")
        for i, c in enumerate(tol_checkpoints):
            if not (isinstance(c, (int, float)) and 0 < c < 1):
                raise ValueError(f"Invalid checkpoint value at index {i}: {c}. Must be between 0 and 1")
            if c < tol:
                raise ValueError(f"Checkpoint value must be greater than tol. However, checkpoint ({c}) < tol ({tol})")
        tol_checkpoints = sorted(tol_checkpoints, reverse=True)

    s = OracleSolver(n_time, geometry, congestion=congestion, eps=eps, tau=tau, init_solution=init_solution)
    if is_palm:                                                     # :253-255
        s.dt_phi, s.dx_phi = grad_time(s.h, s.phi), grad_space(s.G, s.F, s.phi)
    hist = History(nit)
    policy = PenaltyPolicy()
    stop_set, prim_pos, dual_pos = [0, 2, 4, 5], [0, 1], [2, 3]   # :299-301
    is_org_kkt = False

    hist.start()
    it = -1
    prim_gap = 1.0 + 1.0 * np.exp(-100 * congestion)               # :568
    if is_z_scaling:
        s.scale_z(2.0)                                              # :571-572
    if is_constant_scaling:                                         # :574-586
        bt = s.r * s.bnd / s.w_v
        norm_c = math.sqrt(s.nsq_center(bt))
        norm_ac = math.sqrt(s.nsq_time(grad_time(s.h, bt)) + s.nsq_space(grad_space(s.G, s.F, bt)))
        s.scale_prim_dual(s.norm_d, math.sqrt(n_time) * norm_c ** 2 / norm_ac)
        s.adjust_penalty(1.0 / s.r)

    kkt = LazyKKT(s.kkt_functions(), tol, order=[6, 2, 0, 3, 1, 4, 5])   # :589-645
    t_start = time.perf_counter()
    passed = False
    for it in range(nit):                                          # :656
        if is_constant_scaling and policy.is_to_scale(it):
            s.scale_prim_dual_auto()
        if is_z_scaling and policy.is_to_rescale_z(it, hist.get_current_kkt_errors()):   # :661-666
            row = hist.get_current_kkt_errors()
            rescale = prim_gap * math.sqrt(row[1] / row[0]) if (row[1] / row[0]) >= 0 else float("nan")
            if rescale > 1.25:
                s.scale_z(rescale)

        if is_palm:                                                 # :668-672
            with hist.timer("Step 0 (Q & Lambda)"):
                s.step_q_lambda(refresh_gradients=False)
        with hist.timer("Step 1-1 (Laplacian)"):
            s.step_laplacian()
        with hist.timer("Step 1-2 (SOC-Projection)"):
            s.step_soc_projection()
        with hist.timer("Step 2 (Q & Lambda)"):
            s.step_q_lambda()
        with hist.timer("Step 3 (Multiplier)"):
            s.step_multipliers()

        time_up = (time.perf_counter() - t_start) > time_limit     # :725-726
        adjust = policy.is_to_adjust(it) or time_up
        required = prim_pos + dual_pos if adjust else None

        if not check_kkt_step_by_step:                              # :733-768
            if adjust:
                kkt.reset_counter()
            passed, info = kkt.validate(required)
            org, scaled = kkt.collect()
            if adjust:
                kkt.reset_counter()
            hist.record(it, org)
            error = max_skip_none([org[i] for i in stop_set])
            if error is not None:
                kkt.set_error(error, tol)
        else:                                                       # :769-787
            passed, info = kkt._validate(list(range(7)))
            org, scaled = kkt.collect()
            cost, obj = s.objective()
            hist.record(it, org, {"Transportation cost": cost, "Objective value": obj})
            error = max_skip_none([org[i] for i in stop_set])

        if trace is not None:
            trace.append({"it": it, "adjust": adjust, "checked": [i for i, e in enumerate(org) if e is not None],
                          "r": s.r, "sz": s.sz})

        if tol_checkpoints and error is not None and error <= tol_checkpoints[0]:   # :790-801
            checkpoints.append({
                "mu": (s.r * s.dual_scale) * s.mu, "E": (s.r * s.dual_scale) * s.E,
                "iteration": it, "time": hist.get_running_time(), "kkt": np.array(org, dtype=object),
            })
            tol_checkpoints.pop(0)

        if passed or time_up:                                       # :804-805
            break

        mx = max_skip_none(scaled)                                  # :808-810
        if mx is not None and mx < 5 * tol:
            is_org_kkt = True

        if adjust:                                                  # :813-823
            src = org if is_org_kkt else scaled
            gap = max_skip_none([src[i] for i in prim_pos]) / max_skip_none([src[i] for i in dual_pos])
            s.adjust_penalty(policy.updated(s.r, gap) / s.r)

    kkt._validate(list(range(7)))                                   # :826-841
    org, _ = kkt.collect()
    cost, obj = s.objective()
    hist.record(it, org, {"Transportation cost": cost, "Objective value": obj})
    hist.end()

    solution = s.recovered_solution()
    solution["checkpoints"] = checkpoints if checkpoints else None
    return solution, hist


# --------------------------------------------------------------------------- #
# boundary wrappers (socp/solver_decorator.py:10-72, utils/type.py:48-65)
# --------------------------------------------------------------------------- #
def socp_to_dot(solution, geometry):
    """translate_solution_socp_to_dot (utils/type.py:48-65)."""
    return {
        "mu": solution["mu"] * (np.asarray(geometry["area_vertices"])[None, :] / 3.0),
        "E": solution["E"] * np.asarray(geometry["area_triangles"])[None, :, None],
    }


def to_time_centered(mu_staggered, mu0, mu1):
    """solver_decorator.py:32-34."""
    mid = 0.5 * (mu_staggered[:-1] + mu_staggered[1:])
    return np.concatenate([mu0[None, :], mid, mu1[None, :]], axis=0)
