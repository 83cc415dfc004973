"""CPU stand-in for ``dots_socp_amd.device.DeviceProblem`` (TEST INFRASTRUCTURE).

Implements the same calls with numpy on top of the oracle so that the host-side driver
(``AlmSolver`` / ``ShardedAlmSolver``: control logic, mode partition, exchange layout) can run in CPU-only
multi-process tests with a real ``torch.distributed`` (gloo) all-gather.  It is never used by the product.
"""
import ctypes
import types

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from conftest import load_oracle

O = load_oracle()


def _view(ptr, count):
    return np.ctypeslib.as_array((ctypes.c_double * int(count)).from_address(int(ptr)))


class FakeDeviceProblem:
    def __init__(self, n_time, geometry, lap_solver="modal_pcg", device=0, reorder=True, plan=None, mode_shard=None, **_ignored):
        from dots_socp_amd.geometry import time_modes

        self.s = O.OracleSolver(n_time, geometry)
        s = self.s
        self.T, self.V, self.F = s.T, s.V, s.F
        self.lap_solver = lap_solver
        self.mu0, self.mu1 = np.asarray(geometry["mu0"], float), np.asarray(geometry["mu1"], float)
        self.Q, self.sigma = time_modes(n_time)
        self.params = types.SimpleNamespace(
            r=1.0, scale_z=1.0, const_d=1.0, norm_d=s.norm_d, norm_boundary=s.norm_boundary, congestion=0.0, tau=1.9,
            eps=0.0, prim_scale=1.0, dual_scale=1.0, boundary_scale=1.0, cg_tol=1e-10, cg_max_iter=1000)
        self.mode_slice = None
        if mode_shard is not None:
            rank, n_ranks = mode_shard
            self.stride = -(-(n_time + 1) // n_ranks)
            b = min(rank * self.stride, n_time + 1)
            self.mode_slice = slice(b, b + max(0, min(self.stride, n_time + 1 - b)))
            self.n_ranks = n_ranks
        self._lu = {}
        self.plan = types.SimpleNamespace(perm_vert=None, mass_vert=s.mass_v, mu0=self.mu0, mu1=self.mu1)

    # ---- bookkeeping
    def close(self):
        pass

    def shape(self, name):
        return getattr(self.s, name).shape

    def device_bytes(self):
        return 0

    def sync(self):
        pass

    def set_params(self, **kw):
        for k, v in kw.items():
            setattr(self.params, k, v)
        p, s = self.params, self.s
        s.r, s.sz, s.d, s.norm_d, s.norm_boundary = p.r, p.scale_z, p.const_d, p.norm_d, p.norm_boundary
        s.congestion, s.tau, s.eps, s.prim_scale, s.dual_scale = p.congestion, p.tau, p.eps, p.prim_scale, p.dual_scale
        s.bnd[:] = 0.0
        s.bnd[0] = -p.boundary_scale * self.mu0 / (p.r * s.h)
        s.bnd[-1] = p.boundary_scale * self.mu1 / (p.r * s.h)

    def step_flags(self, skip_z_mid=False):
        pass

    def setup_frontal(self, **kw):
        return {"levels": 1}

    def setup_multigrid(self, **kw):
        return None

    def upload(self, name, arr):
        setattr(self.s, name, np.array(arr, dtype=float))

    def download(self, name):
        return np.array(getattr(self.s, name))

    # ---- the sharded iteration
    def shard_elems(self):
        return self.V * self.stride

    def _solve_mode(self, a, rhs_a):
        if a not in self._lu:
            shift = self.sigma[a] + self.params.eps
            if shift == 0.0:
                shift = 1e-9     # singular mode: a tiny shift only moves the (gauge) constant of phi
            A = (self.s.L - shift * sp.diags(self.s.mass_v)).tocsc()
            self._lu[a] = spla.splu(A)
        return self._lu[a].solve(rhs_a)

    def step_begin(self, send_ptr, count):
        rhs = self.s.laplacian_rhs()
        out = _view(send_ptr, count).reshape(self.V, self.stride)
        out[:] = 0.0
        for j, a in enumerate(range(self.mode_slice.start, self.mode_slice.stop)):
            hat = self.Q[:, a] @ rhs
            if self.sigma[a] + self.params.eps == 0.0:
                hat = hat - hat.mean()
            out[:, j] = self._solve_mode(a, hat)
        return types.SimpleNamespace(cg_iterations=1, cg_not_converged=0, ms_rhs=0.0, ms_laplacian=0.0, ms_soc=0.0,
                                     ms_q_lambda_multiplier=0.0, ms_total=0.0, alm_iterations=0)

    def step_end(self, recv_ptr, count):
        g = _view(recv_ptr, count).reshape(-1, self.V, self.stride)
        xhat = np.zeros((self.T + 1, self.V))
        for a in range(self.T + 1):
            xhat[a] = g[a // self.stride, :, a % self.stride]
        self.s.phi[:] = self.Q @ xhat
        self.s.step_soc_projection()
        self.s.step_q_lambda()
        self.s.step_multipliers()
        return types.SimpleNamespace(cg_iterations=0, cg_not_converged=0, ms_rhs=0.0, ms_laplacian=0.0, ms_soc=0.0,
                                     ms_q_lambda_multiplier=0.0, ms_total=0.0, alm_iterations=1)

    # ---- scalars
    def kkt(self, conditions):
        self.s.dt_phi = O.grad_time(self.s.h, self.s.phi)
        self.s.dx_phi = O.grad_space(self.s.G, self.s.F, self.s.phi)
        self.s.dec_B = O.decouple(self.s.B, self.s.sz)
        fns = self.s.kkt_functions()
        return {int(i): list(fns[int(i)]()) for i in conditions}

    def objective(self):
        return self.s.objective()

    def adjust_penalty(self, factor):
        for k in ("mu", "E", "beta_fst", "beta_mid", "beta_end"):
            setattr(self.s, k, getattr(self.s, k) / factor)

    def scale_z(self, z_mul, beta_mul, sz_new):
        s = self.s
        for k in ("z_fst", "z_mid", "z_end"):
            setattr(s, k, getattr(s, k) * z_mul)
        for k in ("beta_fst", "beta_mid", "beta_end"):
            setattr(s, k, getattr(s, k) * beta_mul)
        s.mu = sz_new * (s.beta_fst - s.beta_end)
        s.E = -O.decouple_adjoint(s.beta_mid, sz_new)
