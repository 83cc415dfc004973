"""ctypes binding of libdotsocp_hip.so (declarations mirror include/dots_socp_hip.h).

There is no CPU fallback: if the shared library is missing or fails to load, every entry point of
the package that needs it raises ``HipLibraryError``.
"""
from __future__ import annotations

import ctypes as C
import os

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG_DIR, "libdotsocp_hip.so")
ABI_VERSION = 7

ARRAY_IDS = {
    "phi": 0, "A": 1, "B": 2, "lambda_c": 3, "z_fst": 4, "z_mid": 5, "z_end": 6,
    "mu": 7, "E": 8, "beta_fst": 9, "beta_mid": 10, "beta_end": 11,
}
LAP_SOLVERS = {"spacetime_pcg": 0, "modal_pcg": 1}
PHASES = {"laplacian": 0, "soc_projection": 1, "q_lambda_mult": 2, "q_lambda": 3}
STEP_SKIP_Z_MID, STEP_PALM, STEP_RHS_AHEAD, STEP_TIMED, STEP_CARRY, STEP_KKT_SUMS = 1, 2, 4, 8, 16, 32
OPERATORS = {
    "grad_time": 0, "div_time": 1, "grad_space": 2, "div_space": 3, "decouple": 4,
    "decouple_adjoint": 5, "time_avg_adjoint": 6, "laplacian_apply": 7,
}
EXPORTS = [
    "dots_abi_version", "dots_last_error", "dots_create", "dots_destroy", "dots_set_params", "dots_get_params",
    "dots_sync", "dots_upload", "dots_download", "dots_array_count", "dots_step", "dots_run_phase", "dots_kkt",
    "dots_objective", "dots_adjust_penalty", "dots_scale_z", "dots_scale_arrays", "dots_norm_square",
    "dots_apply_operator", "dots_bench_kernel", "dots_device_bytes", "dots_mg_setup", "dots_mg_enable",
    "dots_slab_elems", "dots_slab_set_buffers", "dots_slab_stage", "dots_kkt_sums", "dots_kkt_sums_device", "dots_debug_counter", "dots_kkt_combine", "dots_objective_sums",
    "dots_objective_combine", "dots_front_launches", "dots_front_info", "dots_front_setup", "dots_front_enable", "dots_front_pitch", "dots_penalty_ahead", "dots_step_flags", "dots_step_times", "dots_stream_wait", "dots_tree_build", "dots_tree_nodes", "dots_tree_copy", "dots_tree_free",
    "dots_patch_order", "dots_assemble", "dots_assemble_nnz", "dots_assemble_copy", "dots_assemble_free", "dots_symbolic_build", "dots_symbolic_front_rows", "dots_symbolic_copy", "dots_symbolic_free",
]


class PenaltyPolicy(C.Structure):      # dots_penalty_policy
    _fields_ = [("tol", C.c_double), ("r_lower", C.c_double), ("r_upper", C.c_double), ("is_org_kkt", C.c_int32), ("n_steps", C.c_int32),
                ("threshold", C.c_double * 16), ("factor", C.c_double * 16)]


class HipLibraryError(RuntimeError):
    status = None      # the library's status code (dots_status) when the error comes from an entry point


ERR_MEMORY = -6


_i32p = C.POINTER(C.c_int32)
_f64p = C.POINTER(C.c_double)


class ProblemDesc(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("device", C.c_int32), ("n_time", C.c_int32), ("n_vertices", C.c_int32),
        ("n_triangles", C.c_int32), ("n_corners", C.c_int32), ("lap_nnz", C.c_int32), ("lap_solver", C.c_int32),
        ("triangles", _i32p), ("hat_grad", _f64p), ("area_tri", _f64p), ("mass_vert", _f64p),
        ("corner_ptr", _i32p), ("corner_idx", _i32p), ("lap_rowptr", _i32p), ("lap_col", _i32p), ("lap_val", _f64p),
        ("mu0", _f64p), ("mu1", _f64p), ("perm_vert", _i32p), ("perm_tri", _i32p),
        ("time_modes", _f64p), ("time_eigs", _f64p),
        ("slab_begin", C.c_int32), ("slab_count", C.c_int32), ("slab_stride", C.c_int32), ("reserved", C.c_int32),
        ("patch_order", _i32p),
    ]


class Params(C.Structure):
    _fields_ = [
        ("r", C.c_double), ("scale_z", C.c_double), ("const_d", C.c_double), ("norm_d", C.c_double),
        ("norm_boundary", C.c_double), ("congestion", C.c_double), ("tau", C.c_double), ("eps", C.c_double),
        ("prim_scale", C.c_double), ("dual_scale", C.c_double), ("boundary_scale", C.c_double), ("cg_tol", C.c_double),
        ("cg_max_iter", C.c_int32), ("reserved", C.c_int32),
    ]


class MgLevel(C.Structure):
    _fields_ = [
        ("n", C.c_int32), ("nnz", C.c_int32), ("rowptr", _i32p), ("col", _i32p), ("val_k", _f64p), ("val_m", _f64p),
        ("diag_k", _f64p), ("diag_m", _f64p), ("n_coarse", C.c_int32), ("p_nnz", C.c_int32),
        ("p_rowptr", _i32p), ("p_col", _i32p), ("p_val", _f64p), ("r_rowptr", _i32p), ("r_col", _i32p), ("r_val", _f64p),
        ("ap_nnz", C.c_int32), ("reserved", C.c_int32), ("ap_rowptr", _i32p), ("ap_col", _i32p),
        ("ap_val_k", _f64p), ("ap_val_m", _f64p), ("ap_val_p", _f64p),
    ]


class MgDesc(C.Structure):
    _fields_ = [
        ("n_levels", C.c_int32), ("n_cols", C.c_int32), ("omega", C.c_double),
        ("levels", C.POINTER(MgLevel)), ("coarse_inverse", _f64p),
    ]


class FrontDesc(C.Structure):
    _fields_ = [
        ("n_nodes", C.c_int32), ("n_levels", C.c_int32), ("n_modes", C.c_int32), ("pitch", C.c_int32),
        ("n_front_rows", C.c_int64), ("n_entries", C.c_int64), ("update_rows", C.c_int64),
        ("node_n", _i32p), ("node_b", _i32p), ("node_foff", C.POINTER(C.c_int64)), ("node_ioff", C.POINTER(C.c_int64)),
        ("node_uoff", C.POINTER(C.c_int64)), ("node_child", _i32p), ("front_idx", _i32p), ("pull0", _i32p), ("pull1", _i32p),
        ("level_ptr", _i32p), ("level_nodes", _i32p), ("values", _f64p), ("grounded", _i32p),
        ("band_ptr", _i32p), ("n_bands", C.c_int32), ("top_inverse", C.c_int32),
    ]


class StepStats(C.Structure):
    _fields_ = [
        ("alm_iterations", C.c_int32), ("cg_iterations", C.c_int32), ("cg_last_iterations", C.c_int32),
        ("cg_not_converged", C.c_int32), ("cg_last_rel_residual", C.c_double),
        ("ms_rhs", C.c_double), ("ms_laplacian", C.c_double), ("ms_soc", C.c_double),
        ("ms_q_lambda_multiplier", C.c_double), ("ms_total", C.c_double),
    ]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


_lib = None


class SlabBuffers(C.Structure):
    """dots_slab_buffers: device pointers of the exchange buffers of a time-slab context."""
    NAMES = ("send_x", "send_nsq", "recv_x", "recv_nsq", "b_send", "b_recv", "x_send", "x_recv", "send_mu", "send_b", "recv_mu", "recv_b")
    _fields_ = [(n, C.c_void_p) for n in NAMES]


KKT_N_SUMS = 24
SLAB_SIZES = {"vertex_halo": 0, "b_chunk": 1, "x_chunk": 2, "triangle_halo": 3}


# Every DOTS_* environment switch the package reads (INTEGRATION.md has the table).  They are measurement aids; a name that is
# not in this list, or a value outside a switch's domain, is an error: a typo must not silently select the default.
KNOWN_ENV = {
    # read by the library (csrc/dots_api.hip: env_int)
    "DOTS_CG_STAGE_LDS", "DOTS_MG_TAIL_ROWS", "DOTS_SOC_WITH_RHS", "DOTS_QL_TWO", "DOTS_KKT_TWO", "DOTS_RHS_TWO", "DOTS_RHS_TILES", "DOTS_CARRY", "DOTS_CARRY_MIN", "DOTS_BM_NT", "DOTS_LAZY_DIV", "DOTS_ZMID_DEFER", "DOTS_MEM_BUDGET", "DOTS_SPIN_FETCH",
    "DOTS_FRONT_VEC2", "DOTS_FRONT_RB", "DOTS_FRONT_ROWS", "DOTS_FRONT_XCD", "DOTS_FRONT_LEAFINV", "DOTS_FRONT_TUNE", "DOTS_FRONT_CFG", "DOTS_MAIL_TEST_DROP",
    "DOTS_MAIL_SPINS", "DOTS_ND_PCA_MIN",
    # read by the host side
    "DOTS_RHS_AHEAD", "DOTS_TIME_EVERY", "DOTS_FRONT_BANDS", "DOTS_FRONT_TOPINV", "DOTS_TORCH_FIRST", "DOTS_DIST_BACKEND", "DOTS_HIPCC_FLAGS",
}


def check_environment():
    unknown = sorted(k for k in os.environ if k.startswith("DOTS_") and k not in KNOWN_ENV)
    if unknown:
        raise HipLibraryError(f"unknown DOTS_* environment switch(es) {unknown}; known: {sorted(KNOWN_ENV)}")


def env_choice(name, choices, default, integer=None):
    """The value of switch ``name``: one of ``choices`` (strings), or with ``integer=(lo, hi)`` an integer in that range
    (returned as a string).  Unset: ``default``.  Anything else raises."""
    v = os.environ.get(name)
    if v is None:
        return default
    if integer is not None:
        try:
            ok = integer[0] <= int(v) <= integer[1]
        except ValueError:
            ok = False
        if not ok:
            raise HipLibraryError(f"environment: {name}={v!r} is not an integer in [{integer[0]}, {integer[1]}]")
        return v
    if v not in choices:
        raise HipLibraryError(f"environment: {name}={v!r} is not one of {list(choices)}")
    return v


def library_path() -> str:
    return LIB_PATH


def _torch_runtime_first(init=True):
    """PyTorch-ROCm wheels bundle their own copy of the HIP runtime under the same SONAME (libamdhip64.so.7) this library
    links against, so a process ends up with ONE runtime: whichever copy is loaded first.  Loaded after torch, this library
    runs on torch's copy (streams, events and device pointers are then interchangeable between the two, which distributed.py
    relies on); loaded BEFORE torch, the system copy is bound and torch's own build fails on it with "No HIP GPUs are
    available" (measured: ROCm 7.2 system runtime, torch 2.10+rocm7.0).  So torch goes first: if it is installed but not yet
    imported it is imported (and initialised) here, whatever order the caller's own imports have (DOTS_TORCH_FIRST=0: leave a
    torch that is not imported yet alone -- for processes that never use torch on the GPU)."""
    import importlib.util
    import sys

    torch = sys.modules.get("torch")
    if torch is None and env_choice("DOTS_TORCH_FIRST", ("0", "1"), "1") == "1" and importlib.util.find_spec("torch") is not None:
        import torch
    if torch is None or not init:
        return
    try:
        if torch.cuda.is_available() and not torch.cuda.is_initialized():
            torch.cuda.init()
    except Exception:  # pragma: no cover - a torch without devices is not this library's business
        pass


_runtime_ready = False


def load(host_only=False):
    """Load the shared library (once) and declare the signatures.  ``host_only``: the caller uses host entry points only
    (dots_assemble, dots_patch_order, dots_tree_*, dots_symbolic_*): torch is still imported first (one copy of the HIP runtime per
    process, see _torch_runtime_first) but the GPU runtime is not initialised -- a process that only assembles operators, or a parent
    that is about to spawn ranks, stays clear of the device."""
    global _lib, _runtime_ready
    if _lib is not None:
        if not host_only and not _runtime_ready:
            _torch_runtime_first(init=True)
            _runtime_ready = True
        return _lib
    check_environment()
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            f"{LIB_PATH} is missing: build it with `python -m dots_socp_amd.build` (hipcc, gfx950). "
            "dots_socp_amd has no CPU fallback."
        )
    _torch_runtime_first(init=not host_only)
    _runtime_ready = not host_only
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover - depends on the host
        raise HipLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    missing = [n for n in EXPORTS if not hasattr(lib, n)]
    if missing:
        raise HipLibraryError(f"{LIB_PATH} does not export {missing}")
    vp = C.c_void_p
    lib.dots_abi_version.restype = C.c_int
    lib.dots_last_error.restype = C.c_char_p
    lib.dots_create.argtypes = [C.POINTER(ProblemDesc), C.POINTER(vp)]
    lib.dots_destroy.argtypes = [vp]
    lib.dots_set_params.argtypes = [vp, C.POINTER(Params)]
    lib.dots_get_params.argtypes = [vp, C.POINTER(Params)]
    lib.dots_sync.argtypes = [vp]
    lib.dots_upload.argtypes = [vp, C.c_int, _f64p, C.c_int64]
    lib.dots_download.argtypes = [vp, C.c_int, _f64p, C.c_int64]
    lib.dots_array_count.argtypes = [vp, C.c_int]
    lib.dots_array_count.restype = C.c_int64
    lib.dots_step.argtypes = [vp, C.c_int, C.POINTER(StepStats)]
    lib.dots_run_phase.argtypes = [vp, C.c_int, C.POINTER(StepStats)]
    lib.dots_kkt.argtypes = [vp, C.c_uint32, _f64p]
    lib.dots_objective.argtypes = [vp, _f64p]
    lib.dots_adjust_penalty.argtypes = [vp, C.c_double]
    lib.dots_scale_z.argtypes = [vp, C.c_double, C.c_double, C.c_double]
    lib.dots_scale_arrays.argtypes = [vp, C.c_uint32, C.c_double]
    lib.dots_norm_square.argtypes = [vp, C.c_int, C.c_int, _f64p]
    lib.dots_apply_operator.argtypes = [vp, C.c_int, C.c_double, _f64p, C.c_int64, _f64p, C.c_int64]
    lib.dots_bench_kernel.argtypes = [vp, C.c_int, C.c_int, _f64p, _f64p]
    lib.dots_slab_elems.argtypes = [vp, C.c_int]
    lib.dots_slab_elems.restype = C.c_int64
    lib.dots_slab_set_buffers.argtypes = [vp, C.POINTER(SlabBuffers)]
    lib.dots_slab_stage.argtypes = [vp, C.c_int, C.POINTER(StepStats)]
    lib.dots_kkt_sums.argtypes = [vp, C.c_uint32, _f64p]
    lib.dots_kkt_sums_device.argtypes = [vp, C.c_uint32, vp]
    lib.dots_debug_counter.argtypes = [vp, C.c_int]
    lib.dots_debug_counter.restype = C.c_int64
    lib.dots_kkt_combine.argtypes = [vp, C.c_uint32, _f64p, _f64p]
    lib.dots_objective_sums.argtypes = [vp, _f64p]
    lib.dots_objective_combine.argtypes = [vp, _f64p, _f64p]
    lib.dots_mg_setup.argtypes = [vp, C.POINTER(MgDesc)]
    lib.dots_mg_enable.argtypes = [vp, C.c_int]
    lib.dots_front_setup.argtypes = [vp, C.POINTER(FrontDesc)]
    lib.dots_front_enable.argtypes = [vp, C.c_int]
    lib.dots_front_pitch.argtypes = [vp]
    lib.dots_front_launches.argtypes = [vp]
    lib.dots_front_info.argtypes = [vp, _f64p]
    lib.dots_step_flags.argtypes = [vp, C.c_uint32]
    lib.dots_penalty_ahead.argtypes = [vp, C.POINTER(PenaltyPolicy)]
    lib.dots_step_times.argtypes = [vp, C.POINTER(StepStats), C.c_int, C.c_int, C.POINTER(C.c_int)]
    lib.dots_stream_wait.argtypes = [vp, vp, C.c_int]
    i64p = C.POINTER(C.c_int64)
    lib.dots_tree_build.argtypes = [C.c_int32, _i32p, _i32p, _f64p, C.c_int32, C.POINTER(vp)]
    lib.dots_patch_order.argtypes = [C.c_int32, _f64p, C.c_int32, _i32p]
    lib.dots_assemble.argtypes = [C.c_int32, C.c_int32, _f64p, _i32p, C.POINTER(vp)]
    lib.dots_assemble_nnz.argtypes = [vp]
    lib.dots_assemble_nnz.restype = C.c_int64
    lib.dots_assemble_copy.argtypes = [vp, _f64p, _f64p, _f64p, _i32p, _i32p, _i32p, _i32p, _f64p]
    lib.dots_assemble_free.argtypes = [vp]
    lib.dots_assemble_free.restype = None
    lib.dots_tree_nodes.argtypes = [vp]
    lib.dots_tree_nodes.restype = C.c_int64
    lib.dots_tree_copy.argtypes = [vp, i64p, i64p, _i32p, _i32p, _i32p]
    lib.dots_tree_free.argtypes = [vp]
    lib.dots_tree_free.restype = None
    lib.dots_symbolic_build.argtypes = [C.c_int32, _i32p, _i32p, C.c_int64, i64p, i64p, _i32p, C.POINTER(vp)]
    lib.dots_symbolic_front_rows.argtypes = [vp]
    lib.dots_symbolic_front_rows.restype = C.c_int64
    lib.dots_symbolic_copy.argtypes = [vp, _i32p, _i32p, _i32p, _i32p]
    lib.dots_symbolic_free.argtypes = [vp]
    lib.dots_symbolic_free.restype = None
    lib.dots_device_bytes.argtypes = [vp]
    lib.dots_device_bytes.restype = C.c_int64
    for n in EXPORTS:
        f = getattr(lib, n)
        if n not in ("dots_last_error", "dots_array_count", "dots_device_bytes", "dots_slab_elems", "dots_tree_nodes", "dots_tree_free", "dots_debug_counter",
                     "dots_symbolic_front_rows", "dots_symbolic_free", "dots_assemble_nnz", "dots_assemble_free"):
            f.restype = C.c_int
    if lib.dots_abi_version() != ABI_VERSION:
        raise HipLibraryError("libdotsocp_hip.so ABI version mismatch; rebuild it")
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().dots_last_error().decode("utf-8", "replace")
        err = HipLibraryError(f"{what or 'libdotsocp_hip'} failed with status {rc}: {msg}")
        err.status = rc
        raise err
