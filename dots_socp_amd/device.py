"""DeviceProblem: one device-resident DOTs-SOCP problem behind the C ABI.

Thin object wrapper over ``libdotsocp_hip.so``: builds the problem description from the host-side
plan (``geometry.build_plan``), owns the context handle, and exposes the calls of
``include/dots_socp_hip.h`` with numpy arrays in the reference's layouts.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .geometry import DevicePlan, build_plan

STATE_NAMES = tuple(_lib.ARRAY_IDS)


def _ptr(a, ctype):
    return None if a is None else a.ctypes.data_as(C.POINTER(ctype))


def place_slab(name, part, full, n0, nl, ni):
    """Write the part of the slab that holds the nodes ``[n0, n0 + nl)`` (``ni`` intervals) into a whole array in the reference
    layout; ``part`` may be longer than the slab along its first axis (padding is ignored).  Corner arrays are indexed by the
    node an entry is compared with (include/dots_socp_hip.h)."""
    if name in ("z_mid", "beta_mid"):
        full[n0:n0 + ni, 0] = part[:ni, 0]
        lo = 1 if n0 == 0 else 0                           # entry [j][1] is the reference's [n0 + j - 1][1]
        full[n0 + lo - 1:n0 + nl - 1, 1] = part[lo:nl, 1]
    else:
        n = nl if name in ("phi", "B", "E") else ni
        full[n0:n0 + n] = part[:n]
    return full


class DeviceProblem:
    def __init__(self, n_time, geometry, lap_solver="spacetime_pcg", device=0, reorder=True, plan: DevicePlan | None = None,
                 time_slab=None, nd_leaf=16):
        """``time_slab = (rank, n_ranks)``: this context is one TIME SLAB of a multi-GPU solve (distributed.py): it holds
        the nodes [rank * stride, ...) of every state array and solves the time modes with the same indices; the caller
        drives ``slab_stage`` around its exchanges."""
        self.lib = _lib.load()
        self.plan = plan if plan is not None else build_plan(n_time, geometry, reorder=reorder, nd_leaf=nd_leaf)
        p = self.plan
        self.T, self.V, self.F = p.n_time, p.n_vertices, p.n_triangles
        if lap_solver not in _lib.LAP_SOLVERS:
            raise ValueError(f"lap_solver must be one of {list(_lib.LAP_SOLVERS)}")
        self.lap_solver = lap_solver
        d = _lib.ProblemDesc()
        d.abi_version = _lib.ABI_VERSION
        d.device = int(device)
        d.n_time, d.n_vertices, d.n_triangles = p.n_time, p.n_vertices, p.n_triangles
        d.n_corners = int(p.corner_idx.size)
        d.lap_nnz = int(p.lap_val.size)
        d.lap_solver = _lib.LAP_SOLVERS[lap_solver]
        d.triangles = _ptr(p.triangles, C.c_int32)
        d.hat_grad = _ptr(p.hat_grad, C.c_double)
        d.area_tri = _ptr(p.area_tri, C.c_double)
        d.mass_vert = _ptr(p.mass_vert, C.c_double)
        d.corner_ptr = _ptr(p.corner_ptr, C.c_int32)
        d.corner_idx = _ptr(p.corner_idx, C.c_int32)
        d.lap_rowptr = _ptr(p.lap_rowptr, C.c_int32)
        d.lap_col = _ptr(p.lap_col, C.c_int32)
        d.lap_val = _ptr(p.lap_val, C.c_double)
        d.mu0 = _ptr(p.mu0, C.c_double)
        d.mu1 = _ptr(p.mu1, C.c_double)
        d.perm_vert = _ptr(p.perm_vert, C.c_int32)
        d.perm_tri = _ptr(p.perm_tri, C.c_int32)
        d.time_modes = _ptr(p.time_modes, C.c_double)
        d.time_eigs = _ptr(p.time_eigs, C.c_double)
        d.patch_order = _ptr(getattr(p, "patch_order", None), C.c_int32)
        self.mode_slice = None
        self.node0, self.nl, self.ni = 0, p.n_time + 1, p.n_time
        self.slab = None
        if time_slab is not None:
            rank, n_ranks = time_slab
            stride = -(-(p.n_time + 1) // n_ranks)
            begin = min(rank * stride, p.n_time + 1)
            count = max(0, min(stride, p.n_time + 1 - begin))
            d.slab_begin, d.slab_count, d.slab_stride = (begin if count else 0), count, stride
            self.mode_slice = slice(begin, begin + count)       # modes solved here = nodes held here
            self.slab = (rank, n_ranks, stride)
            self.node0, self.nl, self.ni = begin, count, max(0, min(count, p.n_time - begin))
            self.active_ranks = -(-(p.n_time + 1) // stride)    # ranks that hold at least one node
        self._h = C.c_void_p()
        _lib.check(self.lib.dots_create(C.byref(d), C.byref(self._h)), "dots_create")
        self.params = _lib.Params()
        _lib.check(self.lib.dots_get_params(self._h, C.byref(self.params)), "dots_get_params")

    # ---- lifecycle
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.dots_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- shapes of the host arrays: the reference layouts on one GPU, the slab's time extent on a time slab
    def shape(self, name):
        V, F = self.V, self.F
        if name == "phi":
            return (self.nl, V)
        if name in ("B", "E"):
            return (self.nl, F, 3)
        if name in ("z_mid", "beta_mid"):
            return ((self.nl if self.slab else self.ni), 2, 3, F, 3)
        return (self.ni, V)

    def full_shape(self, name):
        T, V, F = self.T, self.V, self.F
        return {"phi": (T + 1, V), "B": (T + 1, F, 3), "E": (T + 1, F, 3), "z_mid": (T, 2, 3, F, 3), "beta_mid": (T, 2, 3, F, 3)}.get(name, (T, V))

    def to_slab(self, name, full):
        """This slab's part of a whole array in the reference layout (see include/dots_socp_hip.h: the corner arrays
        are indexed by the node an entry is compared with)."""
        full = np.asarray(full, dtype=np.float64)
        n0, nl, ni = self.node0, self.nl, self.ni
        if name in ("z_mid", "beta_mid"):
            out = np.zeros(self.shape(name))
            out[:ni, 0] = full[n0:n0 + ni, 0]
            lo = 1 if n0 == 0 else 0                           # entry [j][1] is the reference's [n0 + j - 1][1]
            out[lo:nl, 1] = full[n0 + lo - 1:n0 + nl - 1, 1]
            return out
        return np.ascontiguousarray(full[n0:n0 + (nl if name in ("phi", "B", "E") else ni)])

    def from_slab(self, name, part, full):
        """Write this slab's part into a whole array in the reference layout (entries of other slabs untouched)."""
        return place_slab(name, part, full, self.node0, self.nl, self.ni)

    # ---- parameters
    def set_params(self, **kw):
        for k, v in kw.items():
            if not hasattr(self.params, k):
                raise AttributeError(k)
            setattr(self.params, k, v)
        _lib.check(self.lib.dots_set_params(self._h, C.byref(self.params)), "dots_set_params")

    # ---- state transfer
    def upload(self, name, array):
        a = np.ascontiguousarray(array, dtype=np.float64)
        if a.shape != self.shape(name):
            raise ValueError(f"{name}: expected shape {self.shape(name)}, got {a.shape}")
        _lib.check(self.lib.dots_upload(self._h, _lib.ARRAY_IDS[name], _ptr(a, C.c_double), a.size), f"upload {name}")

    def download(self, name):
        out = np.empty(self.shape(name), dtype=np.float64)
        _lib.check(self.lib.dots_download(self._h, _lib.ARRAY_IDS[name], _ptr(out, C.c_double), out.size), f"download {name}")
        return out

    def download_all(self):
        return {n: self.download(n) for n in STATE_NAMES}

    # ---- the hot loop
    def step(self, n_iters=1, wait=True):
        """``wait=False``: only enqueue (direct solver); returns None, nothing is timed."""
        if not wait:
            _lib.check(self.lib.dots_step(self._h, int(n_iters), None), "dots_step")
            return None
        st = _lib.StepStats()
        _lib.check(self.lib.dots_step(self._h, int(n_iters), C.byref(st)), "dots_step")
        return st

    def step_flags(self, skip_z_mid=False, palm=False, rhs_ahead=False, timed=False, carry=False, kkt_sums=False):
        """``rhs_ahead`` (DOTS_STEP_RHS_AHEAD): the first KKT read-back after the next step also enqueues the right-hand side of
        the iteration after it; only meaningful with the direct solver on one GPU.  ``timed`` (DOTS_STEP_TIMED): enqueue-only
        steps record phase events that ``step_times`` collects later.  ``carry`` (DOTS_STEP_CARRY): the next iteration starts
        from the state this one leaves, so steps 2+3 also store the per-corner sums its right-hand side and projection gather.
        ``kkt_sums`` (DOTS_STEP_KKT_SUMS): residuals are read after the step; steps 2+3 also form the sums of conditions 0, 1, 3, 6."""
        flags = ((_lib.STEP_SKIP_Z_MID if skip_z_mid else 0) | (_lib.STEP_PALM if palm else 0) | (_lib.STEP_RHS_AHEAD if rhs_ahead else 0)
                 | (_lib.STEP_TIMED if timed else 0) | (_lib.STEP_CARRY if carry else 0) | (_lib.STEP_KKT_SUMS if kkt_sums else 0))
        _lib.check(self.lib.dots_step_flags(self._h, flags), "dots_step_flags")

    def penalty_ahead(self, tol, is_org_kkt, r_lower, r_upper, table):
        """dots_penalty_ahead: the next evaluation of conditions 0-3 takes the reference's penalty decision inside the library and starts
        the next iteration's first launch with it (``table``: (threshold, factor) pairs of admm_tools.py:79-90); a hint."""
        pol = getattr(self, "_penalty_policy", None)
        if pol is None:
            pol = self._penalty_policy = _lib.PenaltyPolicy()
            pol.n_steps = len(table)
            for i, (bound, val) in enumerate(table):
                pol.threshold[i], pol.factor[i] = bound, val
        pol.tol, pol.r_lower, pol.r_upper, pol.is_org_kkt = float(tol), float(r_lower), float(r_upper), 1 if is_org_kkt else 0
        _lib.check(self.lib.dots_penalty_ahead(self._h, C.byref(pol)), "dots_penalty_ahead")

    def step_times(self, wait=False, capacity=64):
        """Phase times of the timed enqueue-only steps that have finished (``wait``: of all of them), oldest first."""
        buf = getattr(self, "_times_buf", None)
        if buf is None or len(buf) < capacity:
            buf = self._times_buf = (_lib.StepStats * capacity)()
        n = C.c_int(0)
        _lib.check(self.lib.dots_step_times(self._h, buf, capacity, 1 if wait else 0, C.byref(n)), "dots_step_times")
        out = []
        for i in range(n.value):       # copies: the buffer is reused by the next call
            st = _lib.StepStats()
            C.pointer(st)[0] = buf[i]
            out.append(st)
        return out

    # ---- time slab (multi-GPU): stages of one iteration around the caller's exchanges (dots_slab_stage)
    def slab_elems(self, which):
        return int(self.lib.dots_slab_elems(self._h, _lib.SLAB_SIZES[which]))

    def slab_set_buffers(self, **pointers):
        b = _lib.SlabBuffers()
        for name in _lib.SlabBuffers.NAMES:
            setattr(b, name, int(pointers[name]))
        _lib.check(self.lib.dots_slab_set_buffers(self._h, C.byref(b)), "dots_slab_set_buffers")

    def slab_stage(self, stage, wait=False):
        """``wait=False``: only enqueue on the context's stream (order the exchange with ``stream_wait``)."""
        st = _lib.StepStats() if wait else None
        _lib.check(self.lib.dots_slab_stage(self._h, int(stage), C.byref(st) if wait else None), f"dots_slab_stage {stage}")
        return st

    def stream_wait(self, other_stream, ctx_waits):
        """Order the context's stream against another HIP stream (an integer handle, 0 = default stream)."""
        _lib.check(self.lib.dots_stream_wait(self._h, C.c_void_p(int(other_stream)), 1 if ctx_waits else 0), "dots_stream_wait")

    def run_phase(self, phase):
        st = _lib.StepStats()
        _lib.check(self.lib.dots_run_phase(self._h, _lib.PHASES[phase], C.byref(st)), f"phase {phase}")
        return st

    def kkt(self, conditions):
        """Evaluate the listed KKT conditions; returns {i: [value, value_at_unit_scale or None]}."""
        mask = 0
        for i in conditions:
            mask |= 1 << int(i)
        out = getattr(self, "_kkt_out", None)      # (the host's part of a read-back sits on the critical path: no allocations here)
        if out is None:
            out = self._kkt_out = np.empty(14)
            self._kkt_out_p = _ptr(out, C.c_double)
        out.fill(np.nan)
        _lib.check(self.lib.dots_kkt(self._h, mask, self._kkt_out_p), "dots_kkt")
        vals = out.tolist()
        return {int(i): [vals[2 * i], None if i >= 4 else vals[2 * i + 1]] for i in conditions}

    @staticmethod
    def _mask(conditions):
        mask = 0
        for i in conditions:
            mask |= 1 << int(i)
        return mask

    def kkt_sums(self, conditions):
        """The weighted sums of this context's time slab that the listed conditions need (to be added over the slabs)."""
        sums = np.zeros(_lib.KKT_N_SUMS)
        _lib.check(self.lib.dots_kkt_sums(self._h, self._mask(conditions), _ptr(sums, C.c_double)), "dots_kkt_sums")
        return sums

    def kkt_sums_device(self, conditions, device_ptr):
        """``kkt_sums`` left in the caller's DEVICE buffer of ``KKT_N_SUMS`` doubles (an integer address, e.g. a torch
        tensor's ``data_ptr()``): only enqueued on the context's stream -- order the consumer with ``stream_wait``."""
        _lib.check(self.lib.dots_kkt_sums_device(self._h, self._mask(conditions), C.c_void_p(int(device_ptr))), "dots_kkt_sums_device")

    def debug_counter(self, which=0):
        return int(self.lib.dots_debug_counter(self._h, int(which)))

    def kkt_combine(self, conditions, sums):
        """The listed KKT residuals from the sums of the whole problem; same return value as ``kkt``."""
        out = np.full(14, np.nan)
        sums = np.ascontiguousarray(sums, dtype=np.float64)
        _lib.check(self.lib.dots_kkt_combine(self._h, self._mask(conditions), _ptr(sums, C.c_double), _ptr(out, C.c_double)), "dots_kkt_combine")
        return {int(i): [float(out[2 * i]), None if i >= 4 else float(out[2 * i + 1])] for i in conditions}

    def objective(self):
        out = np.zeros(2)
        _lib.check(self.lib.dots_objective(self._h, _ptr(out, C.c_double)), "dots_objective")
        return float(out[0]), float(out[1])

    def objective_sums(self):
        sums = np.zeros(3)
        _lib.check(self.lib.dots_objective_sums(self._h, _ptr(sums, C.c_double)), "dots_objective_sums")
        return sums

    def objective_combine(self, sums):
        out, sums = np.zeros(2), np.ascontiguousarray(sums, dtype=np.float64)
        _lib.check(self.lib.dots_objective_combine(self._h, _ptr(sums, C.c_double), _ptr(out, C.c_double)), "dots_objective_combine")
        return float(out[0]), float(out[1])

    def adjust_penalty(self, factor):
        _lib.check(self.lib.dots_adjust_penalty(self._h, float(factor)), "dots_adjust_penalty")

    def scale_z(self, z_mul, beta_mul, scale_z_new):
        _lib.check(self.lib.dots_scale_z(self._h, float(z_mul), float(beta_mul), float(scale_z_new)), "dots_scale_z")

    def scale_arrays(self, names, factor):
        mask = 0
        for n in names:
            mask |= 1 << _lib.ARRAY_IDS[n]
        _lib.check(self.lib.dots_scale_arrays(self._h, mask, float(factor)), "dots_scale_arrays")

    def norm_square(self, name, part=0):
        out = C.c_double()
        _lib.check(self.lib.dots_norm_square(self._h, _lib.ARRAY_IDS[name], int(part), C.byref(out)), "dots_norm_square")
        return out.value

    def apply_operator(self, op, x, scale=1.0):
        shapes = {
            "grad_time": ("phi", "A"), "div_time": ("A", "phi"), "time_avg_adjoint": ("A", "phi"),
            "grad_space": ("phi", "B"), "div_space": ("B", "phi"), "decouple": ("B", "z_mid"),
            "decouple_adjoint": ("z_mid", "B"), "laplacian_apply": ("phi", "phi"),
        }
        sin, sout = shapes[op]
        a = np.ascontiguousarray(x, dtype=np.float64)
        if a.shape != self.shape(sin):
            raise ValueError(f"{op}: expected input shape {self.shape(sin)}, got {a.shape}")
        out = np.empty(self.shape(sout), dtype=np.float64)
        _lib.check(
            self.lib.dots_apply_operator(self._h, _lib.OPERATORS[op], float(scale), _ptr(a, C.c_double), a.size,
                                         _ptr(out, C.c_double), out.size),
            f"operator {op}",
        )
        return out

    # ---- multigrid preconditioner of the modal PCG
    def setup_multigrid(self, eps=0.0, omega=2.0 / 3.0, coarsest=256, mode_slice=None):
        """Build the smoothed-aggregation hierarchy on the host and upload it (see multigrid.py).

        ``mode_slice``: the time modes this context solves (default: all T+1).  Returns the
        hierarchy summary, or None when the mesh is too small for a second level (Jacobi stays)."""
        import scipy.sparse as sp

        from . import multigrid

        if self.lap_solver != "modal_pcg":
            raise ValueError("the multigrid preconditioner needs lap_solver='modal_pcg'")
        p = self.plan
        K = sp.csr_matrix((p.lap_val, p.lap_col, p.lap_rowptr), shape=(p.n_vertices, p.n_vertices))
        levels = multigrid.build_hierarchy(K, p.mass_vert, coarsest=coarsest)
        if len(levels) < 2:
            return None
        if mode_slice is None:
            mode_slice = self.mode_slice
        sigma = p.time_eigs if mode_slice is None else p.time_eigs[mode_slice]
        if sigma.size == 0:
            return None      # a rank without modes solves nothing
        last = levels[-1]
        inv = np.empty((last.n, last.n, sigma.size))
        for k, s in enumerate(sigma):
            inv[:, :, k] = multigrid.coarse_inverse(last, float(s + eps))
        keep = []   # host arrays must stay alive until dots_mg_setup returns

        def arr(a, dtype):
            a = np.ascontiguousarray(a, dtype=dtype)
            keep.append(a)
            return a

        lv = (_lib.MgLevel * len(levels))()
        for l, L in enumerate(levels):
            h = lv[l]
            h.n = L.n
            if l > 0:
                h.nnz = int(L.K.nnz)
                h.rowptr = _ptr(arr(L.K.indptr, np.int32), C.c_int32)
                h.col = _ptr(arr(L.K.indices, np.int32), C.c_int32)
                h.val_k = _ptr(arr(L.K.data, np.float64), C.c_double)
                h.val_m = _ptr(arr(L.M.data, np.float64), C.c_double)
                h.diag_k = _ptr(arr(L.dK, np.float64), C.c_double)
                h.diag_m = _ptr(arr(L.dM, np.float64), C.c_double)
            if L.P is not None:
                h.n_coarse = L.P.shape[1]
                h.p_nnz = int(L.P.nnz)
                h.p_rowptr = _ptr(arr(L.P.indptr, np.int32), C.c_int32)
                h.p_col = _ptr(arr(L.P.indices, np.int32), C.c_int32)
                h.p_val = _ptr(arr(L.P.data, np.float64), C.c_double)
                h.r_rowptr = _ptr(arr(L.R.indptr, np.int32), C.c_int32)
                h.r_col = _ptr(arr(L.R.indices, np.int32), C.c_int32)
                h.r_val = _ptr(arr(L.R.data, np.float64), C.c_double)
                h.ap_nnz = int(L.KP.nnz)
                h.ap_rowptr = _ptr(arr(L.KP.indptr, np.int32), C.c_int32)
                h.ap_col = _ptr(arr(L.KP.indices, np.int32), C.c_int32)
                h.ap_val_k = _ptr(arr(L.KP.data, np.float64), C.c_double)
                h.ap_val_m = _ptr(arr(L.MP.data, np.float64), C.c_double)
                h.ap_val_p = _ptr(arr(L.PP.data, np.float64), C.c_double)
        desc = _lib.MgDesc()
        desc.n_levels = len(levels)
        desc.n_cols = int(sigma.size)
        desc.omega = float(omega)
        desc.levels = lv
        desc.coarse_inverse = _ptr(arr(inv, np.float64), C.c_double)
        _lib.check(self.lib.dots_mg_setup(self._h, C.byref(desc)), "dots_mg_setup")
        self.mg_summary = multigrid.hierarchy_summary(levels)
        return self.mg_summary

    # ---- direct (multifrontal) solve of the modal problems
    def setup_frontal(self, eps=0.0, leaf=None, mode_slice=None, numeric="device", bands=None, top_inverse=None):
        """Factorise K + (sigma_a + eps) M for this context's modes on one nested-dissection tree
        (frontal.py) and install the factor: step 1 then runs two triangular sweeps instead of the PCG.
        ``numeric``: "device" (HIP kernels, the default) or "host" (numpy reference, small meshes only).
        ``bands``: tree heights one launch of a sweep handles, ``top_inverse``: the top band as explicit inverses, one launch
        for both sweeps (frontal.plan_bands; default: the plan's own choice)."""
        import scipy.sparse as sp

        from . import frontal

        if self.lap_solver != "modal_pcg":
            raise ValueError("the direct solve needs the modal solver")
        p = self.plan
        K = sp.csr_matrix((p.lap_val, p.lap_col, p.lap_rowptr), shape=(p.n_vertices, p.n_vertices))
        diss = p.dissection if leaf is None else None     # the plan's own tree (reorder="nd") unless a leaf size is forced
        if diss is None:
            diss = frontal.nested_dissection(K.indptr, K.indices, p.vertices, leaf=leaf or 16)
        if mode_slice is None:
            mode_slice = self.mode_slice
        sigma = p.time_eigs if mode_slice is None else p.time_eigs[mode_slice]
        if sigma.size == 0:
            return None
        pitch = int(self.lib.dots_front_pitch(self._h))
        p_pitch = max(8, 1 << int(np.ceil(np.log2(p.n_time + 1))))     # the whole problem's pitch (a time slab's own one is smaller)
        if numeric not in ("device", "host"):
            raise ValueError("numeric must be 'device' or 'host'")
        self.set_params(eps=float(eps))      # the device factorisation reads eps from the context
        ff = frontal.factorize(K, p.mass_vert, sigma + float(eps), diss, pitch=pitch, numeric=numeric == "host")
        if bands is None:
            if diss.bands is not None:
                bands, auto_top = diss.bands, diss.top_inverse
            else:
                bands, auto_top = frontal.plan_bands(diss, ff.node_n, ff.node_b, p_pitch)
            if top_inverse is None:
                top_inverse = auto_top
        bands = np.ascontiguousarray(bands, dtype=np.int32)
        d = _lib.FrontDesc()
        d.band_ptr, d.n_bands, d.top_inverse = _ptr(bands, C.c_int32), bands.size - 1, 1 if top_inverse else 0
        d.n_nodes, d.n_levels, d.n_modes, d.pitch = ff.node_n.size, ff.level_ptr.size - 1, ff.n_modes, ff.pitch
        d.n_front_rows, d.n_entries, d.update_rows = ff.front_idx.size, ff.stats["factor_entries_per_mode"], ff.update_rows
        flags = np.zeros(ff.n_modes, dtype=np.int32)
        flags[ff.grounded] = 1
        keep = [np.ascontiguousarray(a) for a in (ff.node_n, ff.node_b, ff.node_foff, ff.node_ioff, ff.node_uoff, ff.node_child,
                                                   ff.front_idx, ff.pull0, ff.pull1, ff.level_ptr, ff.level_nodes)]
        keep.append(None if ff.values is None else np.ascontiguousarray(ff.values))
        d.grounded = _ptr(flags, C.c_int32)
        d.node_n, d.node_b = _ptr(keep[0], C.c_int32), _ptr(keep[1], C.c_int32)
        d.node_foff, d.node_ioff, d.node_uoff = _ptr(keep[2], C.c_int64), _ptr(keep[3], C.c_int64), _ptr(keep[4], C.c_int64)
        d.node_child, d.front_idx = _ptr(keep[5], C.c_int32), _ptr(keep[6], C.c_int32)
        d.pull0, d.pull1 = _ptr(keep[7], C.c_int32), _ptr(keep[8], C.c_int32)
        d.level_ptr, d.level_nodes, d.values = _ptr(keep[9], C.c_int32), _ptr(keep[10], C.c_int32), _ptr(keep[11], C.c_double)
        _lib.check(self.lib.dots_front_setup(self._h, C.byref(d)), "dots_front_setup")
        self.front_summary = dict(ff.stats)
        self.front_summary["launches_per_solve"] = self.front_launches()
        info = (C.c_double * 4)()
        _lib.check(self.lib.dots_front_info(self._h, info), "dots_front_info")
        self.front_summary.update(bands=[int(x) for x in bands], top_inverse=bool(top_inverse), bytes_per_solve_one_block_per_node=float(info[0]),
                                  bytes_per_solve_as_installed=float(info[1]), leaf_inverse=self.debug_counter(4) > 0)
        return self.front_summary

    def front_launches(self):
        return int(self.lib.dots_front_launches(self._h))

    def enable_frontal(self, on=True):
        _lib.check(self.lib.dots_front_enable(self._h, 1 if on else 0), "dots_front_enable")

    def enable_multigrid(self, on=True):
        _lib.check(self.lib.dots_mg_enable(self._h, 1 if on else 0), "dots_mg_enable")

    def bench_kernel(self, which=0, reps=50):
        ms, nbytes = C.c_double(), C.c_double()
        _lib.check(self.lib.dots_bench_kernel(self._h, int(which), int(reps), C.byref(ms), C.byref(nbytes)), "dots_bench_kernel")
        return ms.value, nbytes.value

    def sync(self):
        _lib.check(self.lib.dots_sync(self._h), "dots_sync")

    def device_bytes(self):
        return int(self.lib.dots_device_bytes(self._h))
