"""Host-side setup of the direct (multifrontal) Laplacian solve of step 1.

The reference inverts the space-time Laplacian exactly: an eigen-decomposition in time followed by T+1
sparse LU factorisations of the shifted surface operators ``K + (sigma_a + eps) M`` (SuperLU,
``utils/laplacian_inverse_socp.py:11-61``).  This module builds the MI355X form of that algorithm: ONE
nested-dissection elimination tree of the mesh graph shared by all time modes, and per tree node a small
dense block per mode, laid out so that both triangular sweeps are batched dense matrix-vector products
that stream the factor once from HBM with the mode index fastest (the device layout of every node array):

    node p:  separator rows  sep_p  (n_p vertices, eliminated at p)
             boundary rows   bd_p   (b_p vertices of ancestors' separators that the subtree touches)
             F_p = [ L_pp^-1 ; A_bs A_ss^-1 ]   (n_p + b_p) x n_p   per mode, stored [row][col][mode]

    forward  (leaves -> root, one launch per tree height):
             w      = b[sep_p] - (updates pulled from the two children)
             y_p    = L_pp^-1 w                                   rows 0..n_p of F_p
             u_p    = (children's updates on bd_p) + G_p w        rows n_p.. of F_p      (pull form: no atomics)
    backward (root -> leaves):   x_p = F_p^T [ y_p ; -x[bd_p] ]

The numeric factorisation (dense frontal Cholesky, batched over the modes with numpy) runs once per solve
on the host, as the reference's SuperLU factorisations do.  Nothing here is a fallback for the device
path: the sweeps exist only as HIP kernels (csrc/kernels_front.hip).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np
import scipy.sparse as sp
from scipy.linalg.lapack import dtrtri as _trtri


# ------------------------------------------------------------------------------------------------
# nested dissection
# ------------------------------------------------------------------------------------------------
@dataclass
class Dissection:
    """Elimination tree.  Node ids are in post-order (children before parents; the root is last)."""
    order: np.ndarray          # (V,) vertex eliminated at position k
    sep_ptr: np.ndarray        # (n_nodes+1,) node p eliminates order[sep_ptr[p]:sep_ptr[p+1]]
    child: np.ndarray          # (n_nodes, 2) children ids or -1
    parent: np.ndarray         # (n_nodes,)
    height: np.ndarray         # (n_nodes,) 0 for leaves
    bands: np.ndarray | None = None    # (n_bands+1,) tree heights one launch of a sweep handles (plan_bands); None: one each
    top_inverse: bool = False          # the top band stores explicit inverses: one launch for both sweeps (plan_bands)

    @property
    def n_nodes(self):
        return self.parent.size


def _neighbour_ranges(indptr, rows):
    """Concatenated CSR positions of `rows` and the local row number of each position."""
    starts = indptr[rows]
    lens = indptr[rows + 1] - starts
    total = int(lens.sum())
    first = np.cumsum(lens) - lens
    local = np.repeat(np.arange(rows.size), lens)
    pos = np.arange(total) - np.repeat(first, lens) + np.repeat(starts, lens)
    return pos, local


def nested_dissection(indptr, indices, coords, leaf=16, native=True) -> Dissection:
    """Geometric nested dissection (``native``: the C++ implementation in libdotsocp_hip.so, csrc/dissect.hip --
    host code, ~10x faster; ``native=False``: this module's numpy recursion, the reference for the tests): a subset is cut at the median of its principal axis, the separator is
    the smaller of the two one-sided vertex boundaries of the cut (so it is a separator of the GRAPH
    whatever the embedding looks like); subsets of at most `leaf` vertices become dense leaves."""
    if native:
        return _nested_dissection_native(indptr, indices, coords, leaf)
    indptr = np.asarray(indptr, dtype=np.int64)
    indices = np.asarray(indices, dtype=np.int64)
    coords = np.asarray(coords, dtype=np.float64)
    V = indptr.size - 1
    mark = np.zeros(V, dtype=np.int8)
    seps, childs = [], []

    def cut(S):
        X = coords[S]
        lo, hi = X.min(axis=0), X.max(axis=0)
        ext = hi - lo
        ax = int(np.argmax(ext))
        if S.size > 512:      # large subsets: cut across the principal axis (fewer separator vertices on slanted pieces)
            Xc = X - X.mean(axis=0)
            _, U = np.linalg.eigh(Xc.T @ Xc)
            proj = Xc @ U[:, -1]
        else:                 # small ones: the longest side of the bounding box is as good and much cheaper
            proj = X[:, ax]
        half = S.size // 2
        o = np.argpartition(proj, half)
        A, B = S[o[:half]], S[o[half:]]
        mark[A], mark[B] = 1, 2
        pa, la = _neighbour_ranges(indptr, A)
        inA = np.bincount(la, weights=(mark[indices[pa]] == 2), minlength=A.size) > 0
        pb, lb = _neighbour_ranges(indptr, B)
        inB = np.bincount(lb, weights=(mark[indices[pb]] == 1), minlength=B.size) > 0
        mark[S] = 0
        if int(inA.sum()) <= int(inB.sum()):
            return A[~inA], B, A[inA]
        return A, B[~inB], B[inB]

    def build(S):
        if S.size <= leaf:
            seps.append(S)
            childs.append((-1, -1))
            return len(seps) - 1
        A, B, sep = cut(S)
        if A.size == 0 or B.size == 0:      # degenerate cut (e.g. a thin strip): eliminate the subset densely
            seps.append(S)
            childs.append((-1, -1))
            return len(seps) - 1
        a = build(A)
        b = build(B)
        seps.append(sep)
        childs.append((a, b))
        return len(seps) - 1

    import sys
    old = sys.getrecursionlimit()
    sys.setrecursionlimit(max(old, 10000))
    try:
        build(np.arange(V, dtype=np.int64))
    finally:
        sys.setrecursionlimit(old)
    n = len(seps)
    child = np.asarray(childs, dtype=np.int32).reshape(n, 2)
    parent = np.full(n, -1, dtype=np.int32)
    height = np.zeros(n, dtype=np.int32)
    for p in range(n):
        for c in child[p]:
            if c >= 0:
                parent[c] = p
                height[p] = max(height[p], height[c] + 1)
    sep_ptr = np.zeros(n + 1, dtype=np.int64)
    sep_ptr[1:] = np.cumsum([s.size for s in seps])
    order = np.concatenate(seps)
    assert order.size == V and np.array_equal(np.sort(order), np.arange(V))
    return Dissection(order=order, sep_ptr=sep_ptr, child=child, parent=parent, height=height)


def _ptr(a, ctype):
    import ctypes as C
    return a.ctypes.data_as(C.POINTER(ctype))


def _nested_dissection_native(indptr, indices, coords, leaf):
    import ctypes as C

    from . import _lib

    lib = _lib.load(host_only=True)
    ip = np.ascontiguousarray(indptr, dtype=np.int32)
    ix = np.ascontiguousarray(indices, dtype=np.int32)
    xyz = np.ascontiguousarray(coords, dtype=np.float64)
    V = ip.size - 1
    h = C.c_void_p()
    _lib.check(lib.dots_tree_build(V, _ptr(ip, C.c_int32), _ptr(ix, C.c_int32), _ptr(xyz, C.c_double), int(leaf), C.byref(h)), "dots_tree_build")
    try:
        n = int(lib.dots_tree_nodes(h))
        order, sep_ptr = np.empty(V, dtype=np.int64), np.empty(n + 1, dtype=np.int64)
        child, parent, height = np.empty((n, 2), dtype=np.int32), np.empty(n, dtype=np.int32), np.empty(n, dtype=np.int32)
        _lib.check(lib.dots_tree_copy(h, _ptr(order, C.c_int64), _ptr(sep_ptr, C.c_int64), _ptr(child, C.c_int32), _ptr(parent, C.c_int32),
                                      _ptr(height, C.c_int32)), "dots_tree_copy")
    finally:
        lib.dots_tree_free(h)
    return Dissection(order=order, sep_ptr=sep_ptr, child=child, parent=parent, height=height)


def symbolic_native(diss: Dissection, indptr, indices):
    """(node_b, front_idx, pull0, pull1) of the tree on the graph, from csrc/dissect.hip (same result as
    ``symbolic`` + the front / pull loops of ``factorize``)."""
    import ctypes as C

    from . import _lib

    lib = _lib.load(host_only=True)
    ip = np.ascontiguousarray(indptr, dtype=np.int32)
    ix = np.ascontiguousarray(indices, dtype=np.int32)
    order = np.ascontiguousarray(diss.order, dtype=np.int64)
    sep_ptr = np.ascontiguousarray(diss.sep_ptr, dtype=np.int64)
    child = np.ascontiguousarray(diss.child, dtype=np.int32)
    n = diss.n_nodes
    h = C.c_void_p()
    _lib.check(lib.dots_symbolic_build(ip.size - 1, _ptr(ip, C.c_int32), _ptr(ix, C.c_int32), n, _ptr(order, C.c_int64), _ptr(sep_ptr, C.c_int64),
                                       _ptr(child, C.c_int32), C.byref(h)), "dots_symbolic_build")
    try:
        rows = int(lib.dots_symbolic_front_rows(h))
        node_b = np.empty(n, dtype=np.int32)
        front_idx, pull0, pull1 = (np.empty(rows, dtype=np.int32) for _ in range(3))
        _lib.check(lib.dots_symbolic_copy(h, _ptr(node_b, C.c_int32), _ptr(front_idx, C.c_int32), _ptr(pull0, C.c_int32), _ptr(pull1, C.c_int32)),
                   "dots_symbolic_copy")
    finally:
        lib.dots_symbolic_free(h)
    return node_b, front_idx, pull0, pull1


# ------------------------------------------------------------------------------------------------
# bands of tree heights that one launch of a sweep handles (csrc/kernels_front.hip, "merged heights")
# ------------------------------------------------------------------------------------------------
def _band_tops_full(diss: Dissection, n, b, lo, hi):
    """(entries of the members' rows, columns of every merged node, boundary rows of every merged node) of the band of
    heights [lo, hi), vectorised over the nodes of one height at a time (children before parents)."""
    h, child, parent = diss.height, diss.child, diss.parent
    desc = np.zeros(n.size, dtype=np.int64)
    for k in range(lo + 1, hi):
        P = np.flatnonzero(h == k)
        for s in range(2):
            c = child[P, s]
            ok = (c >= 0) & (h[np.maximum(c, 0)] >= lo)
            cc = np.maximum(c, 0)
            desc[P] += np.where(ok, desc[cc] + n[cc], 0)
    inb = (h >= lo) & (h < hi)
    entries = int((n[inb] * (n[inb] + 1) // 2 + n[inb] * desc[inb]).sum())
    top = inb & ((parent < 0) | (h[np.maximum(parent, 0)] >= hi))
    return entries, (desc + n)[top], b[top]


def band_entries(diss: Dissection, node_n, node_b, lo, hi, leaf_inverse=False):
    """(factor entries one sweep reads, front rows, vector entries the forward workgroups read) when the heights
    [lo, hi) are merged: a member's rows hold its own triangle plus the columns of its descendants inside the band, the
    top member's boundary rows all columns; every row block of a merged node reads the right-hand side and the update
    planes of its columns (2, 4 or 8 planes for bands of 1, 2 or more heights above the leaves).
    ``leaf_inverse``: the band [0, 1) stored as explicit local inverses (n (n + 1) / 2 entries per leaf; kernels_front.hip)."""
    n = np.asarray(node_n, dtype=np.int64)
    b = np.asarray(node_b, dtype=np.int64)
    entries, nm, bm = _band_tops_full(diss, n, b, lo, hi)
    if leaf_inverse and lo == 0 and hi == 1:
        entries = int((nm * (nm + 1) // 2).sum())
    else:
        entries += int((bm * nm).sum())
    rows = int((nm + bm).sum())
    planes = 0 if lo == 0 else (2 if hi - lo == 1 else (4 if hi - lo == 2 else 8))
    rb = 4 if rows >= 4096 else (2 if rows >= 2048 else 1)
    vec = float((0.5 * nm * nm + nm * bm).sum()) / rb * (1 + planes)
    return entries, rows, vec


# Cost model of the sweeps, fitted to the solve times of ~3 000 cuts measured on MI355X (meshes of 2.5 k to 41 k vertices,
# T = 31 and 63, with and without the explicit top inverse; profiles/studies/band_cuts.txt).  Microseconds per launch:
#     launch + factor MB / bandwidth + steps * BAND_STEP_US        steps = columns of the longest row / 32
# The longest row is the serial part of a launch (a row is split over 32 lanes per mode); the bandwidth a factor is
# streamed at falls with its size (cache-resident ... far beyond the 256 MB Infinity Cache).
BAND_LAUNCH_US = (3.7, 8.0)     # small factors ... factors of 1 GB and more (more workgroup rounds per launch)
BAND_FACTOR_TBS = ((1.0e8, 7.3), (3.0e8, 5.3), (1.0e9, 4.5), (3.0e9, 3.0))      # (factor bytes read per solve, TB/s)
BAND_STEP_US = 0.45
BAND_VECTOR_TBS = 67.0          # right-hand side and update planes re-read by every row block (L2)
BAND_INVERSE_MAX_COLS = 450     # an explicit top inverse pays up to about this many columns per node (knot 398: -5 us; sphere10k 559: +14 us)


def _factor_tbs(total):
    pts = BAND_FACTOR_TBS
    if total <= pts[0][0]:
        return pts[0][1]
    for (x0, y0), (x1, y1) in zip(pts[:-1], pts[1:]):
        if total <= x1:
            w = (np.log(total) - np.log(x0)) / (np.log(x1) - np.log(x0))
            return y0 + w * (y1 - y0)
    return pts[-1][1]


def plan_bands(diss: Dissection, node_n, node_b, pitch, max_heights=4, spec=None, top_spec=None):
    """Cut the tree heights into bands, each handled by ONE launch per sweep: the cuts that minimise the modelled solve
    time (see BAND_* above).  On small meshes a launch per height costs more than the bytes it streams, so two to four
    heights are merged per band; on large ones (bandwidth-bound) almost nothing is.  Returns (cuts, top_inverse):
    ``top_inverse`` = the top band (its nodes have no boundary rows) stores the explicit inverse S^-1 = L'^-T L'^-1 of its
    merged block, read ONCE per solve by a single launch instead of L'^-1 twice by two (n^2 entries against ~n^2 / 2 twice:
    the same bytes, one launch less -- and a larger top band becomes affordable).
    ``spec`` (or the environment variable DOTS_FRONT_BANDS): "auto", "off", or explicit cuts "0,2,5,8,10"; ``top_spec``
    (DOTS_FRONT_TOPINV): "auto", "0", "1".  Depends on the tree and the pitch of the whole problem only, so that every rank
    of a sharded run cuts alike."""
    import os

    from ._lib import env_choice

    H = int(diss.height.max()) + 1
    spec = os.environ.get("DOTS_FRONT_BANDS", "auto") if spec is None else spec
    top_spec = env_choice("DOTS_FRONT_TOPINV", ("auto", "0", "1"), "auto") if top_spec is None else str(top_spec)
    if top_spec not in ("auto", "0", "1"):
        raise ValueError(f"top_inverse must be 'auto', '0' or '1', not {top_spec!r}")
    if spec == "off":
        return np.arange(H + 1, dtype=np.int32), top_spec == "1"
    if spec != "auto":
        try:
            cuts = np.asarray([int(x) for x in str(spec).split(",")], dtype=np.int32)
        except ValueError:
            raise ValueError(f"band cuts must be 'auto', 'off' or a comma-separated list of tree heights, not {spec!r}") from None
        if cuts[0] != 0 or cuts[-1] != H or np.any(np.diff(cuts) < 1) or np.any(np.diff(cuts) > max_heights):
            raise ValueError(f"bad band cuts {spec!r} for a tree of {H} heights")
        return cuts, top_spec == "1"
    unit = float(pitch) * 8.0
    n = np.asarray(node_n, dtype=np.int64)
    b = np.asarray(node_b, dtype=np.int64)
    total = float((n * (n + 1) // 2 + b * n).sum()) * unit * 2.0
    big = min(1.0, total / 1.0e9)
    launch_us = BAND_LAUNCH_US[0] + (BAND_LAUNCH_US[1] - BAND_LAUNCH_US[0]) * big
    factor_tbs = _factor_tbs(total)
    cost, inv_cost = {}, {}
    for hi in range(1, H + 1):
        for lo in range(max(0, hi - max_heights), hi):
            e, rows, vec = band_entries(diss, n, b, lo, hi)
            tops = _band_tops(diss, n, lo, hi)
            steps = max(tops) / 32.0
            cost[(lo, hi)] = launch_us + e * unit / (factor_tbs * 1e6) + (vec + 3.0 * rows) * unit / (BAND_VECTOR_TBS * 1e6) * big \
                + 0.75 * steps * BAND_STEP_US
            if hi == H and top_spec != "0" and (max(tops) <= BAND_INVERSE_MAX_COLS or top_spec == "1"):       # per sweep: half of ONE launch that reads n^2 entries per node, in full rows
                full = float(sum(t * t for t in tops))
                inv_cost[lo] = 0.5 * (launch_us + full * unit / (factor_tbs * 1e6) + steps * BAND_STEP_US)
    best = {0: (0.0, [0], False)}
    for hi in range(1, H + 1):
        cands = []
        for lo in range(max(0, hi - max_heights), hi):
            cands.append((best[lo][0] + cost[(lo, hi)], best[lo][1] + [hi], False))
            if hi == H and lo in inv_cost and top_spec in ("auto", "1"):
                cands.append((best[lo][0] + inv_cost[lo], best[lo][1] + [hi], True))
        if hi == H and top_spec == "1":
            cands = [c for c in cands if c[2]] or cands
        best[hi] = min(cands, key=lambda c: c[0])
    return np.asarray(best[H][1], dtype=np.int32), bool(best[H][2])


def _band_tops(diss: Dissection, n, lo, hi):
    """columns of the merged nodes of the band of heights [lo, hi)"""
    return [int(x) for x in _band_tops_full(diss, n, np.zeros_like(n), lo, hi)[1]]


def sweep_order(diss: Dissection, bands):
    """Vertex sequence of the sweeps: the separators of the nodes of a band that hang together are consecutive
    (sorted by the band's top node, then by node).  With bands of one height this is the elimination order."""
    h, parent = diss.height, diss.parent
    band_of = np.searchsorted(np.asarray(bands), h, side="right") - 1
    root = np.arange(diss.n_nodes)
    for p in range(diss.n_nodes - 1, -1, -1):
        if parent[p] >= 0 and band_of[parent[p]] == band_of[p]:
            root[p] = root[parent[p]]
    seq = np.lexsort((np.arange(diss.n_nodes), root))
    return np.concatenate([diss.order[diss.sep_ptr[p]:diss.sep_ptr[p + 1]] for p in seq])


# ------------------------------------------------------------------------------------------------
# symbolic + numeric factorisation
# ------------------------------------------------------------------------------------------------
@dataclass
class FrontalFactor:
    """Flat arrays of the device solver (see include/dots_socp_hip.h, dots_front_desc)."""
    n_vertices: int
    n_modes: int
    pitch: int                 # doubles per (row, col) entry of F = modes padded to the device pitch
    node_n: np.ndarray         # (n_nodes,) separator size
    node_b: np.ndarray         # (n_nodes,) boundary size
    node_foff: np.ndarray      # (n_nodes,) int64 offset of F_p in entries (row*col units, multiply by pitch)
    node_ioff: np.ndarray      # (n_nodes,) int64 offset into front_idx / pull0 / pull1
    node_uoff: np.ndarray      # (n_nodes,) int64 first row of u_p in the update buffer
    node_child: np.ndarray     # (n_nodes, 2)
    front_idx: np.ndarray      # vertex of every front row (separator rows first, then boundary rows)
    pull0: np.ndarray          # position of the front row in child 0's boundary (or -1); same length as front_idx
    pull1: np.ndarray
    level_ptr: np.ndarray      # (n_levels+1,) into level_nodes
    level_nodes: np.ndarray    # nodes sorted by height
    values: np.ndarray         # F of all nodes, (sum m_p n_p, pitch)
    update_rows: int           # rows of the update buffer
    grounded: np.ndarray       # modes whose operator is singular (root pivot grounded)
    stats: dict = field(default_factory=dict)


def symbolic(diss: Dissection, indptr, indices):
    """Boundary sets: bd_p = (adj(sep_p) U bd(children)) minus everything eliminated at or below p,
    ordered by elimination position."""
    V = diss.order.size
    pos = np.empty(V, dtype=np.int64)
    pos[diss.order] = np.arange(V)
    bds = []
    for p in range(diss.n_nodes):
        sep = diss.order[diss.sep_ptr[p]:diss.sep_ptr[p + 1]]
        last = diss.sep_ptr[p + 1]
        pr, _ = _neighbour_ranges(indptr, sep)
        cand = [indices[pr]]
        for c in diss.child[p]:
            if c >= 0:
                cand.append(bds[c])
        cand = np.unique(np.concatenate(cand))
        cand = cand[pos[cand] >= last]
        bds.append(cand[np.argsort(pos[cand], kind="stable")])
    return bds, pos


def _numeric(K_parts, mass, shifts, singular, diss, bds, node_foff, values, cols):
    """Dense frontal Cholesky of the modes `cols` (a slice of the mode axis); writes values[:, cols]."""
    indptr, indices, data = K_parts
    V = indptr.size - 1
    A = shifts.size
    frontpos = np.full(V, -1, dtype=np.int64)
    schur = {}
    for p in range(diss.n_nodes):
        sep = diss.order[diss.sep_ptr[p]:diss.sep_ptr[p + 1]]
        bd = bds[p]
        n, b = sep.size, bd.size
        front = np.concatenate([sep, bd])
        frontpos[front] = np.arange(n + b)
        pr, loc = _neighbour_ranges(indptr, sep)
        rows = frontpos[indices[pr]]
        ok = rows >= 0
        rows, loc, val = rows[ok], loc[ok], data[pr][ok]
        # columns of the front that are eliminated here: A[front, sep]; the boundary block starts from the children
        C = np.zeros((A, n + b, n))
        C[:, rows, loc] = val[None, :]
        ar = np.arange(n)
        C[:, ar, ar] += shifts[:, None] * mass[sep][None, :]
        Abb = np.zeros((A, b, b))
        for c in diss.child[p]:
            if c < 0:
                continue
            cm = frontpos[bds[c]]
            Sc = schur.pop(c)
            lo = cm < n                      # child boundary rows that are separator rows of p
            ilo, ihi = np.flatnonzero(lo), np.flatnonzero(~lo)
            C[:, cm[:, None], cm[ilo][None, :]] += Sc[:, :, ilo]
            if ihi.size:
                h = cm[ihi] - n
                Abb[:, h[:, None], h[None, :]] += Sc[:, ihi[:, None], ihi[None, :]]
        frontpos[front] = -1
        if n == 0:      # empty separator (the cut fell between two components): the node only passes updates on
            if b:
                schur[p] = Abb
            continue
        Ass = np.ascontiguousarray(C[:, :n])
        root = diss.parent[p] < 0
        if root and singular.any():
            Ass[singular, n - 1, :] = 0.0
            Ass[singular, :, n - 1] = 0.0
            Ass[singular, n - 1, n - 1] = 1.0
        L = np.linalg.cholesky(Ass)
        Linv = np.empty_like(L)
        for a in range(A):
            Linv[a] = _trtri(L[a], lower=1)[0]
        if root and singular.any():
            Linv[singular, n - 1, :] = 0.0
        fo = node_foff[p]
        out = values[fo:fo + (n + b) * n].reshape(n + b, n, -1)
        out[:n, :, cols] = np.moveaxis(Linv, 0, -1)
        if b:
            Abs = np.ascontiguousarray(C[:, n:])
            G = (Abs @ np.ascontiguousarray(np.swapaxes(Linv, 1, 2))) @ Linv
            S = Abb - G @ np.ascontiguousarray(np.swapaxes(Abs, 1, 2))
            schur[p] = 0.5 * (S + np.swapaxes(S, 1, 2))
            out[n:, :, cols] = np.moveaxis(G, 0, -1)


def factorize(K: sp.csr_matrix, mass, shifts, diss: Dissection, pitch=None, workers=1, numeric=True) -> FrontalFactor:
    """Symbolic structure of the multifrontal factor of K + shifts[a] * diag(mass) on the tree `diss` and,
    with ``numeric=True``, its values computed on the host with numpy (a reference for the device
    factorisation of csrc/kernels_factor.hip, which is what the solver uses: ``numeric=False`` leaves
    ``values`` None)."""
    import os
    from concurrent.futures import ThreadPoolExecutor

    K = sp.csr_matrix(K)
    K.sort_indices()
    indptr, indices, data = K.indptr.astype(np.int64), K.indices.astype(np.int64), K.data.astype(np.float64)
    mass = np.asarray(mass, dtype=np.float64)
    shifts = np.asarray(shifts, dtype=np.float64)
    A = shifts.size
    P = int(pitch) if pitch is not None else A
    V = K.shape[0]
    nn = diss.n_nodes
    node_n = np.diff(diss.sep_ptr).astype(np.int32)
    if numeric:
        bds, pos = symbolic(diss, indptr, indices)
        node_b = np.asarray([b.size for b in bds], dtype=np.int32)
    else:       # structure only: the C++ symbolic phase
        bds = None
        node_b, front_idx, pull0n, pull1n = symbolic_native(diss, K.indptr, K.indices)
    m = node_n.astype(np.int64) + node_b
    node_foff = np.zeros(nn, dtype=np.int64)
    node_foff[1:] = np.cumsum(m * node_n)[:-1]
    node_ioff = np.zeros(nn, dtype=np.int64)
    node_ioff[1:] = np.cumsum(m)[:-1]
    node_uoff = np.zeros(nn, dtype=np.int64)
    node_uoff[1:] = np.cumsum(node_b.astype(np.int64))[:-1]
    total_f = int((m * node_n).sum())
    if numeric:
        front_idx = np.empty(int(m.sum()), dtype=np.int32)
        pull = [np.full(front_idx.size, -1, dtype=np.int32), np.full(front_idx.size, -1, dtype=np.int32)]
        frontpos = np.full(V, -1, dtype=np.int64)
        for p in range(nn):
            front = np.concatenate([diss.order[diss.sep_ptr[p]:diss.sep_ptr[p + 1]], bds[p]])
            io = node_ioff[p]
            front_idx[io:io + front.size] = front
            frontpos[front] = np.arange(front.size)
            for k, c in enumerate(diss.child[p]):
                if c >= 0:
                    cm = frontpos[bds[c]]
                    pull[k][io + cm] = np.arange(cm.size, dtype=np.int32)
            frontpos[front] = -1
    else:
        pull = [pull0n, pull1n]

    scale = float(np.abs(K.diagonal()).max())
    singular = np.abs(shifts) * float(mass.max()) <= 1e-13 * scale
    values = np.zeros((total_f, P), dtype=np.float64) if numeric else None
    if workers is None:
        workers = max(1, min(A, (os.cpu_count() or 2) - 1, 16))
    chunks = [c for c in np.array_split(np.arange(A), workers) if c.size] if numeric else []
    parts = (indptr, indices, data)

    def run(cols):
        _numeric(parts, mass, shifts[cols], singular[cols], diss, bds, node_foff, values, cols)

    if len(chunks) == 1:
        run(chunks[0])
    elif chunks:
        try:
            from threadpoolctl import threadpool_limits
            limit = threadpool_limits(limits=1)
        except Exception:  # pragma: no cover
            import contextlib
            limit = contextlib.nullcontext()
        with limit, ThreadPoolExecutor(len(chunks)) as ex:
            list(ex.map(run, chunks))
    order_by_h = np.argsort(diss.height, kind="stable").astype(np.int32)
    n_levels = int(diss.height.max()) + 1
    level_ptr = np.searchsorted(diss.height[order_by_h], np.arange(n_levels + 1)).astype(np.int32)
    stats = {
        "nodes": int(nn), "levels": n_levels, "leaf_rows_max": int(node_n[diss.height == 0].max()),
        "root_rows": int(node_n[-1]), "empty_separators": int((node_n == 0).sum()), "max_front": int(m.max()), "factor_entries_per_mode": total_f,
        "factor_bytes": int(total_f) * P * 8, "update_rows": int(node_b.sum()),
    }
    return FrontalFactor(
        n_vertices=V, n_modes=A, pitch=P, node_n=node_n, node_b=node_b, node_foff=node_foff, node_ioff=node_ioff,
        node_uoff=node_uoff, node_child=diss.child.astype(np.int32), front_idx=front_idx, pull0=pull[0], pull1=pull[1],
        level_ptr=level_ptr, level_nodes=order_by_h, values=values, update_rows=int(node_b.sum()),
        grounded=np.flatnonzero(singular).astype(np.int32), stats=stats,
    )
