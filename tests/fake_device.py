"""CPU stand-in for ``dots_socp_amd.device.DeviceProblem`` (TEST INFRASTRUCTURE).

Implements the calls of a TIME-SLAB context with numpy on top of the oracle so that the host-side driver
(``AlmSolver`` / ``ShardedAlmSolver``: control logic, slab partition, stage order, buffer layouts, exchanges) can run in
CPU-only multi-process tests with a real ``torch.distributed`` (gloo) backend.  It is never used by the product.

Every fake rank keeps the WHOLE state (the oracle's arrays) and advances it redundantly; what the test is about is the
wiring: each stage writes exactly the payload a real slab would send, and every payload RECEIVED through the driver's
exchanges is compared with the value this rank's own whole state says it must have (direction, neighbour, buffer
layout, ordering).  The slab numerics themselves are the business of the GPU tests (tests/test_hip_sharded.py).
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


def _stats(**kw):
    base = dict(cg_iterations=0, cg_not_converged=0, ms_rhs=0.0, ms_laplacian=0.0, ms_soc=0.0, ms_q_lambda_multiplier=0.0,
                ms_total=0.0, alm_iterations=0, cg_last_iterations=0)
    base.update(kw)
    return types.SimpleNamespace(**base)


class FakeDeviceProblem:
    def __init__(self, n_time, geometry, lap_solver="modal_pcg", device=0, reorder=True, plan=None, time_slab=None, **_ignored):
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
        assert time_slab is not None, "the CPU stand-in only plays time-slab contexts"
        rank, n_ranks = time_slab
        self.rank, self.n_ranks = rank, n_ranks
        self.stride = -(-(n_time + 1) // n_ranks)
        self.node0 = min(rank * self.stride, n_time + 1)
        self.nl = max(0, min(self.stride, n_time + 1 - self.node0))
        self.ni = max(0, min(self.nl, n_time - self.node0))
        self.slab = (rank, n_ranks, self.stride)
        self.mode_slice = slice(self.node0, self.node0 + self.nl)
        self.active_ranks = -(-(n_time + 1) // self.stride)
        self.pitch = max(4, 1 << (self.stride - 1).bit_length())
        self._lu = {}
        from dots_socp_amd.geometry import build_plan

        real = build_plan(n_time, geometry, reorder=False, native=False)       # hat gradients / areas for the closed forms of the driver
        self.plan = types.SimpleNamespace(perm_vert=None, mass_vert=s.mass_v, mu0=self.mu0, mu1=self.mu1, area_tri=real.area_tri,
                                          hat_grad=real.hat_grad, triangles=real.triangles)
        self.v2c = O.corner_maps(s.V, s.tri, s.area_f)[2]          # (3F, V) 0/1
        self.stage = 0
        self.kkt_halo_fresh = False
        self.checked = {"recv_x": 0, "recv_nsq": 0, "lamc_lo": 0, "b": 0, "x": 0, "recv_mu": 0, "recv_b": 0}

    # ---- bookkeeping
    def close(self):
        pass

    def device_bytes(self):
        return 0

    def sync(self):
        pass

    def stream_wait(self, other, ctx_waits):
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

    def step_flags(self, skip_z_mid=False, palm=False, timed=False, carry=False):
        self.palm = palm

    def step_times(self, wait=False, capacity=64):
        return []

    def setup_frontal(self, **kw):
        return {"levels": 1}

    def setup_multigrid(self, **kw):
        return None

    # ---- host arrays: a slab's own time extent (the real class's slicing helpers work on this object too)
    def shape(self, name):
        from dots_socp_amd.device import DeviceProblem

        return DeviceProblem.shape(self, name)

    def full_shape(self, name):
        return getattr(self.s, name).shape

    def to_slab(self, name, full):
        from dots_socp_amd.device import DeviceProblem

        return DeviceProblem.to_slab(self, name, full)

    def from_slab(self, name, part, full):
        from dots_socp_amd.device import DeviceProblem

        return DeviceProblem.from_slab(self, name, part, full)

    def upload(self, name, arr):
        raise NotImplementedError("the CPU stand-in starts from the zero state")

    def download(self, name):
        return self.to_slab(name, getattr(self.s, name))

    # ---- exchange buffers
    def slab_elems(self, which):
        return {"vertex_halo": self.V, "b_chunk": self.V * self.pitch, "x_chunk": self.V * self.pitch + self.V, "triangle_halo": 3 * self.F}[which]

    def slab_set_buffers(self, **pointers):
        n = self.n_ranks
        size = {"b_send": "b_chunk", "x_send": "x_chunk", "send_b": "triangle_halo", "recv_b": "triangle_halo"}
        self.buf = {}
        for k, ptr in pointers.items():
            if k in ("b_recv", "x_recv"):
                count = n * self.slab_elems("b_chunk" if k == "b_recv" else "x_chunk")
            else:
                count = self.slab_elems(size.get(k, "vertex_halo"))
            self.buf[k] = _view(ptr, count)

    # ---- whole-state helpers
    def _X(self):
        s = self.s
        return s.A + s.lambda_c - s.mu

    def _half_norms(self, t):
        """The s = 1 half of the cone's squared norm of interval t, per vertex (compared with B[t + 1])."""
        s = self.s
        w = s.D[:, :, None] * (O.decouple(s.B, s.sz)[t, 1] - s.beta_mid[t, 1])          # (3, F, 3)
        return self.v2c.T.dot((w ** 2).sum(axis=2).reshape(-1))

    def _solve_mode(self, a, rhs_a):
        if a not in self._lu:
            shift = self.sigma[a] + self.params.eps
            if shift == 0.0:
                shift = 1e-9     # singular mode: a tiny shift only moves the (gauge) constant of phi
            A = (self.s.L - shift * sp.diags(self.s.mass_v)).tocsc()
            self._lu[a] = spla.splu(A)
        return self._lu[a].solve(rhs_a)

    def _check(self, what, got, want):
        assert np.allclose(got, want, rtol=1e-12, atol=1e-300), f"rank {self.rank}: received {what} is not what the neighbour holds"
        self.checked[what] += 1

    # ---- the four stages of one iteration (+ the KKT halos)
    def slab_stage(self, stage, wait=False):
        s, b = self.s, self.buf
        n0, nl, ni, T = self.node0, self.nl, self.ni, self.T
        assert stage == 4 or stage == self.stage or (stage == 6 and self.stage == 1), "stages out of order"
        has_next, has_prev = n0 + nl <= T and nl > 0, n0 > 0 and nl > 0
        if stage == 0:
            if getattr(self, "palm", False):
                s.step_q_lambda(refresh_gradients=False)
            if has_next:
                b["send_x"][:] = self._X()[n0 + nl - 1]
            if has_prev:
                b["send_nsq"][:] = self._half_norms(n0 - 1)
        elif stage in (1, 6):      # right-hand side (1: and the projection, in one stage)
            if has_prev:
                self._check("recv_x", b["recv_x"], self._X()[n0 - 1])
            self.rhs = s.laplacian_rhs()
            out = b["b_send"]
            out[:] = 0.0
            if nl:
                out.reshape(self.V, self.pitch)[:, :nl] = self.rhs[n0:n0 + nl].T
        if stage in (1, 5):        # cone projection; the multipliers of the slab's last interval ride in the solution's all-gather
            if has_next:
                self._check("recv_nsq", b["recv_nsq"], self._half_norms(n0 + nl - 1))
            self.B_old, self.bm_old = s.B.copy(), s.beta_mid.copy()
            s.step_soc_projection()
            w_fst = s.d - s.sz * s.A - s.beta_fst          # the cone multiplier, as the projection forms it
            w_mid = s.D[None, None, :, :, None] * (O.decouple(self.B_old, s.sz) - self.bm_old)
            nrm = np.sqrt(s.c2v_one_T.dot((w_mid ** 2).sum(axis=(1, 4)).reshape(-1)).reshape(T, s.V) + (s.d + s.sz * s.A - s.beta_end) ** 2)
            with np.errstate(divide="ignore", invalid="ignore"):
                self.lam = np.clip(0.5 * (1.0 + w_fst / nrm), 0.0, 1.0)
            b["x_send"][self.V * self.pitch:] = self.lam[n0 + nl - 1] if (nl and has_next) else 0.0
        elif stage == 2:
            chunks = b["b_recv"].reshape(self.n_ranks, -1)
            rhs = np.zeros_like(self.rhs)
            for t in range(T + 1):
                p, j = divmod(t, self.stride)
                rhs[t] = chunks[p].reshape(self.V, self.pitch)[:, j]
            self._check("b", rhs, self.rhs)
            out = b["x_send"][:self.V * self.pitch].reshape(self.V, self.pitch)
            out[:] = 0.0
            for j, a in enumerate(range(n0, n0 + nl)):
                hat = self.Q[:, a] @ rhs
                if self.sigma[a] + self.params.eps == 0.0:
                    hat = hat - hat.mean()
                out[:, j] = self._solve_mode(a, hat)
        elif stage == 3:
            chunks = b["x_recv"].reshape(self.n_ranks, -1)
            if has_prev:
                self._check("lamc_lo", chunks[self.rank - 1, self.V * self.pitch:], self.lam[n0 - 1])
            g = chunks[:, :self.V * self.pitch].reshape(self.n_ranks, self.V, self.pitch)
            xhat = np.zeros((T + 1, self.V))
            for a in range(T + 1):
                xhat[a] = g[a // self.stride, :, a % self.stride]
            if nl:
                own = np.stack([self._solve_mode(a, self.Q[:, a] @ self.rhs - ((self.Q[:, a] @ self.rhs).mean() if self.sigma[a] + self.params.eps == 0.0 else 0.0))
                                for a in range(n0, n0 + nl)])
                self._check("x", xhat[n0:n0 + nl], own)
            s.phi[:] = self.Q @ xhat
            s.step_q_lambda()
            s.step_multipliers()
            self.kkt_halo_fresh = False
        elif stage == 4:
            if has_next:
                b["send_mu"][:] = s.mu[n0 + nl - 1]
            if has_prev:
                b["send_b"][:] = s.B[n0].reshape(-1)
            self.kkt_halo_fresh = True
        if stage <= 3:
            self.stage = (stage + 1) & 3
        elif stage == 6:
            self.stage = 5
        elif stage == 5:
            self.stage = 2
        return _stats(alm_iterations=1 if stage == 3 else 0) if wait else None

    # ---- scalars: every rank contributes (slot 0 counts the contributions), the residuals come from the whole state
    def kkt_sums(self, conditions):
        if set(conditions) & {2, 4, 5}:
            assert self.kkt_halo_fresh, "KKT sums with time stencils before the halo exchange"
            n0, nl = self.node0, self.nl
            if n0 > 0 and nl > 0:
                self._check("recv_mu", self.buf["recv_mu"], self.s.mu[n0 - 1])
            if n0 + nl <= self.T and nl > 0:
                self._check("recv_b", self.buf["recv_b"], self.s.B[n0 + nl].reshape(-1))
        out = np.zeros(24)
        out[0] = 1.0
        return out

    def kkt_combine(self, conditions, sums):
        assert sums[0] == self.n_ranks, "the KKT sums were not added over all ranks"
        self.s.dt_phi = O.grad_time(self.s.h, self.s.phi)
        self.s.dx_phi = O.grad_space(self.s.G, self.s.F, self.s.phi)
        self.s.dec_B = O.decouple(self.s.B, self.s.sz)
        fns = self.s.kkt_functions()
        return {int(i): list(fns[int(i)]()) for i in conditions}

    def objective_sums(self):
        return np.array([1.0, 0.0, 0.0])

    def objective_combine(self, sums):
        assert sums[0] == self.n_ranks
        return self.s.objective()

    # ---- is_constant_scaling: the slab's SHARE of a weighted norm (the driver adds the shares of all ranks)
    def norm_square(self, name, part=0):
        s, n0, nl, ni, T = self.s, self.node0, self.nl, self.ni, self.T
        if nl == 0:
            return 0.0
        mass, area = s.mass_v, s.area_f
        if name == "phi" and part == 1:
            x = O.grad_time(s.h, s.phi)[n0:n0 + ni]
            return float(np.sum(x ** 2 * mass[None, :])) / T
        if name == "phi" and part == 2:
            x = O.grad_space(s.G, s.F, s.phi)[n0:n0 + nl]
            return float(np.sum(x ** 2 * area[None, :, None])) / (T + 1)
        a = getattr(s, name)
        if name == "phi":
            return float(np.sum(a[n0:n0 + nl] ** 2 * mass[None, :])) / (T + 1)
        if name in ("B", "E"):
            return float(np.sum(a[n0:n0 + nl] ** 2 * area[None, :, None])) / (T + 1)
        if name in ("z_mid", "beta_mid"):      # an entry belongs to the slab of the NODE it is compared with
            w = area[None, None, :, None]
            lo = max(n0 - 1, 0)
            return (float(np.sum(a[n0:n0 + ni, 0] ** 2 * w)) + float(np.sum(a[lo:n0 + nl - 1, 1] ** 2 * w))) / T
        return float(np.sum(a[n0:n0 + ni] ** 2 * mass[None, :])) / T

    def scale_arrays(self, names, factor):
        self.kkt_halo_fresh = False
        for k in names:
            setattr(self.s, k, getattr(self.s, k) * factor)

    def adjust_penalty(self, factor):
        self.kkt_halo_fresh = False
        for k in ("mu", "E", "beta_fst", "beta_mid", "beta_end"):
            setattr(self.s, k, getattr(self.s, k) / factor)

    def scale_z(self, z_mul, beta_mul, sz_new):
        self.kkt_halo_fresh = False
        s = self.s
        for k in ("z_fst", "z_mid", "z_end"):
            setattr(s, k, getattr(s, k) * z_mul)
        for k in ("beta_fst", "beta_mid", "beta_end"):
            setattr(s, k, getattr(s, k) * beta_mul)
        s.mu = sz_new * (s.beta_fst - s.beta_end)
        s.E = -O.decouple_adjoint(s.beta_mid, sz_new)
