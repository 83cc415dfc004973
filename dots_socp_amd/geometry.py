"""Host-side operator assembly for the HIP path (vectorised numpy, runs once per solve).

Produces the flat arrays of ``dots_problem_desc`` (include/dots_socp_hip.h).  It covers what the
reference computes in ``utils/surface_pre_computations_socp.py:11-132`` and
``socp/solver_socp.py:102-113,161-192`` -- triangle areas, hat-function gradients, vertex masses,
the surface stiffness matrix, the vertex<->corner incidence -- but keeps every constant at its
natural size (per triangle / per vertex / per corner) instead of broadcasting it to the size of
the state, and never builds the Kronecker incidence matrices: the kernels index the corner lists.

Also computes an optional locality-preserving renumbering (reverse Cuthill-McKee on the mesh
graph) so that neighbouring rows sit in neighbouring tiles / the same XCD's L2.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import scipy.sparse as sp
from scipy.sparse.csgraph import reverse_cuthill_mckee

from ._lib import env_choice


@dataclass
class DevicePlan:
    n_time: int
    n_vertices: int
    n_triangles: int
    triangles: np.ndarray     # (F,3) int32, device numbering
    hat_grad: np.ndarray      # (F,3,3) float64   [f, corner, xyz]
    area_tri: np.ndarray      # (F,)
    mass_vert: np.ndarray     # (V,)  = incident area / 3
    corner_ptr: np.ndarray    # (V+1,) int32
    corner_idx: np.ndarray    # (3F,)  int32, f*3+k
    lap_rowptr: np.ndarray    # (V+1,) int32
    lap_col: np.ndarray       # (nnz,) int32
    lap_val: np.ndarray       # (nnz,) float64,  K = G^T diag(area) G = - cotangent Laplacian
    mu0: np.ndarray           # (V,) device numbering
    mu1: np.ndarray
    perm_vert: np.ndarray | None   # device vertex i = caller vertex perm_vert[i]
    perm_tri: np.ndarray | None
    time_modes: np.ndarray    # (T+1, T+1) Q[t, a]
    time_eigs: np.ndarray     # (T+1,) sigma_a >= 0
    area_mesh: float
    vertices: np.ndarray | None = None    # (V,3) device numbering (used by the nested dissection of frontal.py)
    patch_order: np.ndarray | None = None  # (V,) int32: device vertices as compact patches (geometry.patch_order)
    dissection: object | None = None      # frontal.Dissection in device numbering when reorder == "nd"


def hat_gradients(vertices, triangles):
    """Triangle areas and the gradients of the three hat functions of every triangle.

    The gradient of the hat function of corner k is the altitude vector from the opposite edge to
    that corner divided by its squared length (same quantity as the reference's ``base_function``,
    surface_pre_computations_socp.py:31-37)."""
    v = np.asarray(vertices, dtype=np.float64)
    t = np.asarray(triangles)
    p = v[t]                                             # (F, 3, 3)
    area = 0.5 * np.linalg.norm(np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 1]), axis=1)
    g = np.empty_like(p)
    for k in range(3):
        a, b = p[:, (k + 1) % 3], p[:, (k + 2) % 3]
        e = b - a
        w = p[:, k] - a
        alt = w - e * (np.einsum("ij,ij->i", w, e) / np.einsum("ij,ij->i", e, e))[:, None]
        g[:, k] = alt / np.einsum("ij,ij->i", alt, alt)[:, None]
    return area, g


def stiffness_matrix(n_vertices, triangles, area, hat):
    """K = G^T diag(area) G  (V x V, CSR, sorted): minus the cotangent Laplacian.  Formed as the sparse product it is
    (G: 3F x V, row (f, c) holds the c-th component of the three hat gradients of triangle f, surface_pre_computations_socp.py:55-65)
    rather than by summing 9 F coordinate entries: 2.7x faster at 10^5 vertices, equal to 4e-15."""
    t = np.asarray(triangles)
    F = t.shape[0]
    indptr = np.arange(0, 9 * F + 1, 3, dtype=np.int64)
    cols = np.repeat(t, 3, axis=0).reshape(-1)                            # row (f, c): the columns t[f, 0..2]
    data = np.ascontiguousarray(np.transpose(hat, (0, 2, 1))).reshape(-1)  # [f][c][k]
    G = sp.csr_matrix((data, cols, indptr), shape=(3 * F, n_vertices))
    K = (G.T @ sp.diags(np.repeat(area, 3)) @ G).tocsr()
    K.sum_duplicates()
    K.sort_indices()
    return K


def assemble_native(vertices, triangles):
    """``hat_gradients`` + vertex masses + ``corner_lists`` + ``stiffness_matrix`` in one call of the library's host code
    (``dots_assemble``: the same formulas in the same order of operations; the numpy functions of this module stay the
    reference implementations the tests compare it with).  Returns (area, hat, mass, corner_ptr, corner_idx, K)."""
    import ctypes as C

    from . import _lib

    v = np.ascontiguousarray(vertices, dtype=np.float64)
    t = np.ascontiguousarray(triangles, dtype=np.int32)
    V, F = v.shape[0], t.shape[0]
    lib = _lib.load(host_only=True)      # host code of the library: no GPU runtime is initialised for it
    h = C.c_void_p()
    f64, i32 = C.POINTER(C.c_double), C.POINTER(C.c_int32)
    _lib.check(lib.dots_assemble(V, F, v.ctypes.data_as(f64), t.ctypes.data_as(i32), C.byref(h)), "dots_assemble")
    try:
        nnz = int(lib.dots_assemble_nnz(h))
        area, hat, mass = np.empty(F), np.empty((F, 3, 3)), np.empty(V)
        cptr, cidx = np.empty(V + 1, dtype=np.int32), np.empty(3 * F, dtype=np.int32)
        rowptr, col, val = np.empty(V + 1, dtype=np.int32), np.empty(nnz, dtype=np.int32), np.empty(nnz)
        _lib.check(lib.dots_assemble_copy(h, area.ctypes.data_as(f64), hat.ctypes.data_as(f64), mass.ctypes.data_as(f64), cptr.ctypes.data_as(i32),
                                          cidx.ctypes.data_as(i32), rowptr.ctypes.data_as(i32), col.ctypes.data_as(i32), val.ctypes.data_as(f64)), "dots_assemble_copy")
    finally:
        lib.dots_assemble_free(h)
    return area, hat, mass, cptr, cidx, sp.csr_matrix((val, col, rowptr), shape=(V, V))


def corner_lists(n_vertices, triangles):
    """vertex -> corners CSR.  Corners of a vertex are ordered by (k, f), i.e. by the reference's
    corner index i = k*F + f (surface_pre_computations_socp.py:114-118)."""
    t = np.asarray(triangles)
    F = t.shape[0]
    vert_of_corner = t.T.reshape(-1)                                    # i = k*F + f
    order = np.argsort(vert_of_corner, kind="stable")
    k, f = np.divmod(order, F)
    ptr = np.zeros(n_vertices + 1, dtype=np.int64)
    np.add.at(ptr, vert_of_corner + 1, 1)
    return np.cumsum(ptr).astype(np.int32), (f * 3 + k).astype(np.int32)


def time_modes(n_time):
    """Orthonormal eigenvectors Q[t, a] and eigenvalues sigma_a >= 0 of minus the Neumann second
    difference on n_time+1 nodes with step h = 1/n_time (the matrix of
    laplacian_inverse_socp.py:15-26): the DCT-II basis."""
    n = n_time + 1
    h = 1.0 / n_time
    a = np.arange(n)
    j = np.arange(n)
    Q = np.cos(np.pi * np.outer(j + 0.5, a) / n) * np.sqrt(np.where(a == 0, 1.0, 2.0) / n)[None, :]
    sigma = (2.0 - 2.0 * np.cos(np.pi * a / n)) / (h * h)
    return Q, sigma


def mesh_adjacency(n_vertices, triangles):
    """CSR pattern (no values) of the vertex graph of the mesh: what the renumberings need, at a fraction of the
    cost of assembling the stiffness matrix."""
    t = np.asarray(triangles)
    rows = np.concatenate([t[:, 0], t[:, 1], t[:, 1], t[:, 2], t[:, 2], t[:, 0]])
    cols = np.concatenate([t[:, 1], t[:, 0], t[:, 2], t[:, 1], t[:, 0], t[:, 2]])
    A = sp.csr_matrix((np.ones(rows.size, dtype=np.int8), (rows, cols)), shape=(n_vertices, n_vertices))
    A.sum_duplicates()
    A.sort_indices()
    return A


def locality_order(K, triangles):
    """Reverse Cuthill-McKee numbering of the vertices; triangles sorted by their smallest new vertex."""
    perm_v = np.asarray(reverse_cuthill_mckee(K, symmetric_mode=True), dtype=np.int64)
    inv = np.empty_like(perm_v)
    inv[perm_v] = np.arange(perm_v.size)
    t_new = inv[np.asarray(triangles)]
    perm_f = np.argsort(t_new.min(axis=1), kind="stable")
    return perm_v, perm_f


def dissection_order(K, vertices, triangles, leaf=16, pitch=32):
    """Nested-dissection numbering (the order in which the sweeps of the direct solver walk the vertices, frontal.py:
    the elimination order, with the separators of merged tree heights pulled together): leaves are compact patches of
    the surface, so it is also a locality-preserving numbering for the gathers."""
    from . import frontal

    diss = frontal.nested_dissection(K.indptr, K.indices, vertices, leaf=leaf)
    node_b = frontal.symbolic_native(diss, K.indptr, K.indices)[0]
    diss.bands, diss.top_inverse = frontal.plan_bands(diss, np.diff(diss.sep_ptr), node_b, pitch)
    perm_v = frontal.sweep_order(diss, diss.bands).astype(np.int64)
    inv = np.empty_like(perm_v)
    inv[perm_v] = np.arange(perm_v.size)
    t_new = inv[np.asarray(triangles)]
    perm_f = np.argsort(t_new.min(axis=1), kind="stable")
    diss.order = inv[diss.order]      # elimination order in the new numbering (the identity when no heights are merged)
    return perm_v, perm_f, diss


def patch_order(vertices, unit=16):
    """A sequence of all vertices in which every aligned run of ``unit * 2^k`` entries is a compact patch of the surface:
    recursive coordinate bisection (longest side of the bounding box, cut at a multiple of ``unit``), leaves of <= ``unit``
    vertices, siblings next to each other (host code of the library: ``dots_patch_order``).  The element-wise kernels that walk
    the corner lists of a tile of vertices (right-hand side, cone projection) take their tiles from this sequence instead of
    from the numbering: the triangles of a compact patch are shared by its own vertices (~1.6 distinct triangle fetches per
    triangle and 16-vertex tile instead of ~2.0 for 16 consecutive vertices of the sweep order, whose separators are
    lines), while the rows of a vertex are contiguous in memory whatever the order the vertices are visited in."""
    import ctypes as C

    from . import _lib

    v = np.ascontiguousarray(vertices, dtype=np.float64)
    out = np.empty(v.shape[0], dtype=np.int32)
    lib = _lib.load(host_only=True)
    _lib.check(lib.dots_patch_order(v.shape[0], v.ctypes.data_as(C.POINTER(C.c_double)), int(unit), out.ctypes.data_as(C.POINTER(C.c_int32))),
               "dots_patch_order")
    return out


def assemble_reference(vertices, triangles):
    """The same six results from the numpy reference functions of this module (no library needed: host-only tools and tests)."""
    V = vertices.shape[0]
    area, hat = hat_gradients(vertices, triangles)
    cptr, cidx = corner_lists(V, triangles)
    mass = np.zeros(V)
    np.add.at(mass, triangles.reshape(-1), np.repeat(area, 3) / 3.0)
    return area, hat, mass, cptr.astype(np.int32), cidx.astype(np.int32), stiffness_matrix(V, triangles, area, hat)


def build_plan(n_time, geometry, reorder=True, nd_leaf=16, native=True) -> DevicePlan:
    """``reorder``: True / "rcm" reverse Cuthill-McKee, "nd" nested dissection (direct solver), False none.
    ``native``: the library's host assembly (dots_assemble; the product path) or the numpy reference functions."""
    vertices = np.asarray(geometry["vertices"], dtype=np.float64)
    triangles = np.asarray(geometry["triangles"]).astype(np.int64)
    mu0 = np.asarray(geometry["mu0"], dtype=np.float64)
    mu1 = np.asarray(geometry["mu1"], dtype=np.float64)
    V, F = vertices.shape[0], triangles.shape[0]
    if triangles.min() < 0 or triangles.max() >= V:
        raise ValueError("triangle index out of range")
    if mu0.shape != (V,) or mu1.shape != (V,):
        raise ValueError("mu0/mu1 must have one entry per vertex")

    perm_v = perm_f = diss = None
    if reorder:
        K0 = mesh_adjacency(V, triangles)
        if reorder == "nd":
            pitch = max(8, 1 << int(np.ceil(np.log2(n_time + 1))))
            perm_v, perm_f, diss = dissection_order(K0, vertices, triangles, leaf=nd_leaf, pitch=pitch)
        else:
            perm_v, perm_f = locality_order(K0, triangles)
        inv = np.empty_like(perm_v)
        inv[perm_v] = np.arange(V)
        vertices = vertices[perm_v]
        triangles = inv[triangles[perm_f]]
        mu0, mu1 = mu0[perm_v], mu1[perm_v]

    area, hat, mass, cptr, cidx, K = assemble_native(vertices, triangles) if native else assemble_reference(vertices, triangles)
    if not np.all(area > 0):
        raise ValueError("degenerate triangle (zero area)")
    if not np.all(mass > 0):
        raise ValueError("isolated vertex (no incident triangle)")
    Q, sigma = time_modes(n_time)
    c = np.ascontiguousarray
    return DevicePlan(
        n_time=int(n_time), n_vertices=V, n_triangles=F,
        triangles=c(triangles.astype(np.int32)), hat_grad=c(hat), area_tri=c(area), mass_vert=c(mass),
        corner_ptr=c(cptr), corner_idx=c(cidx),
        lap_rowptr=c(K.indptr.astype(np.int32)), lap_col=c(K.indices.astype(np.int32)), lap_val=c(K.data.astype(np.float64)),
        mu0=c(mu0), mu1=c(mu1),
        perm_vert=None if perm_v is None else c(perm_v.astype(np.int32)),
        perm_tri=None if perm_f is None else c(perm_f.astype(np.int32)),
        time_modes=c(Q), time_eigs=c(sigma), area_mesh=float(area.sum()), vertices=c(vertices), dissection=diss,
        patch_order=c(patch_order(vertices)) if env_choice("DOTS_RHS_TILES", ("0", "1", "2"), "0") != "0" else None,
    )
