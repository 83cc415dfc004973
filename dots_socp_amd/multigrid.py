"""Host-side setup of the smoothed-aggregation multigrid hierarchy used to precondition step 1's PCG.

The T+1 time modes of the space-time Laplacian share one spatial operator up to a shift:
``A_a = K + (sigma_a + eps) M`` with K the surface stiffness matrix and M the vertex mass matrix.
One hierarchy is built from K alone (the hardest, unshifted mode) and used for every mode; each
level keeps K_l and M_l with a common sparsity pattern so the device applies
``K_l + (sigma_a + eps) M_l`` to all modes at once (mode index fastest in memory).

Setup is plain numpy/scipy and runs once per solve (the reference spends its setup on T+1 sparse LU
factorisations instead, utils/laplacian_inverse_socp.py:31-41).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List

import numpy as np
import scipy.sparse as sp


@dataclass
class MgLevel:
    n: int
    K: sp.csr_matrix            # stiffness on this level
    M: sp.csr_matrix            # mass on this level, same pattern as K (explicit zeros kept)
    P: sp.csr_matrix | None     # prolongation to this level from the next coarser one (n x n_coarse)
    R: sp.csr_matrix | None     # restriction = P^T (n_coarse x n)
    KP: sp.csr_matrix | None = None   # K @ P  (n x n_coarse), used by the fused post-smoothing kernel
    MP: sp.csr_matrix | None = None   # M @ P on the pattern of KP
    PP: sp.csr_matrix | None = None   # P itself on the pattern of KP (zeros where P has no entry)
    dK: np.ndarray = field(default=None)   # diag(K)
    dM: np.ndarray = field(default=None)   # diag(M)
    cand: np.ndarray = field(default=None)  # near-null-space candidate of K on this level (K @ cand = 0)


def _align_to_pattern(K: sp.csr_matrix, M: sp.csr_matrix) -> sp.csr_matrix:
    """M re-stored on the sparsity pattern of K (pattern(M) must be a subset; missing entries are 0.0)."""
    n, ncol = K.shape
    M = M.tocsr()
    M.sort_indices()
    rows_k = np.repeat(np.arange(n, dtype=np.int64), np.diff(K.indptr))
    rows_m = np.repeat(np.arange(n, dtype=np.int64), np.diff(M.indptr))
    key_k = rows_k * ncol + K.indices
    key_m = rows_m * ncol + M.indices
    pos = np.searchsorted(key_k, key_m)
    if pos.size and (pos.max() >= key_k.size or not np.array_equal(key_k[pos], key_m)):
        raise ValueError("mass pattern is not contained in the stiffness pattern")
    data = np.zeros(K.nnz)
    data[pos] = M.data
    return sp.csr_matrix((data, K.indices.copy(), K.indptr.copy()), shape=K.shape)


def _pattern_union(*mats) -> sp.csr_matrix:
    """CSR matrix of ones on the union of the (non-negative) arguments' patterns, sorted indices."""
    acc = None
    for m in mats:
        m = sp.csr_matrix(m)
        m = sp.csr_matrix((np.ones(m.nnz), m.indices, m.indptr), shape=m.shape)
        acc = m if acc is None else acc + m
    acc = acc.tocsr()
    acc.sort_indices()
    acc.data[:] = 1.0
    return acc


def _strength_graph(K: sp.csr_matrix, theta: float) -> sp.csr_matrix:
    """Symmetric strength-of-connection: |K_ij| >= theta * sqrt(K_ii K_jj), i != j."""
    d = K.diagonal()
    C = K.tocoo()
    keep = (C.row != C.col) & (np.abs(C.data) >= theta * np.sqrt(np.abs(d[C.row] * d[C.col])))
    S = sp.coo_matrix((np.abs(C.data[keep]), (C.row[keep], C.col[keep])), shape=K.shape).tocsr()
    S.sort_indices()
    return S


def _aggregate(S: sp.csr_matrix) -> np.ndarray:
    """Greedy aggregation (Vanek-Mandel-Brezina): returns the aggregate id of every vertex."""
    n = S.shape[0]
    ptr, idx, val = S.indptr, S.indices, S.data
    agg = np.full(n, -1, dtype=np.int64)
    n_agg = 0
    # pass 1: a vertex whose whole strong neighbourhood is free seeds an aggregate
    for i in range(n):
        if agg[i] != -1:
            continue
        nb = idx[ptr[i]:ptr[i + 1]]
        if nb.size and np.all(agg[nb] == -1):
            agg[i] = n_agg
            agg[nb] = n_agg
            n_agg += 1
    # pass 2: leftovers join the aggregate of their strongest aggregated neighbour (as of pass 1)
    snapshot = agg.copy()
    for i in np.flatnonzero(agg == -1):
        nb = idx[ptr[i]:ptr[i + 1]]
        w = val[ptr[i]:ptr[i + 1]]
        ok = snapshot[nb] != -1
        if np.any(ok):
            agg[i] = snapshot[nb[ok][np.argmax(w[ok])]]
    # pass 3: anything still free forms aggregates with its free neighbours (isolated vertices alone)
    for i in np.flatnonzero(agg == -1):
        if agg[i] != -1:
            continue
        agg[i] = n_agg
        nb = idx[ptr[i]:ptr[i + 1]]
        agg[nb[agg[nb] == -1]] = n_agg
        n_agg += 1
    return agg


def build_hierarchy(K: sp.csr_matrix, mass: np.ndarray, max_levels: int = 8, coarsest: int = 96, theta: float = 0.08,
                    omega_p: float = 4.0 / 3.0) -> List[MgLevel]:
    """Smoothed-aggregation hierarchy for the constant near-null-space of K."""
    K = sp.csr_matrix(K, dtype=np.float64)
    K.sort_indices()
    M = sp.diags(np.asarray(mass, dtype=np.float64)).tocsr()
    cand = np.ones(K.shape[0])
    levels: List[MgLevel] = []
    for lvl in range(max_levels):
        n = K.shape[0]
        # K and M on one pattern (explicit zeros where needed) so the device stores one index array per
        # level.  scipy prunes numerically-zero results of sparse products (a flat mesh has cot 90 = 0
        # entries), so the union pattern is formed from absolute values, which cannot cancel.
        pat = _pattern_union(abs(K), abs(M))
        K, Mp = _align_to_pattern(pat, K), _align_to_pattern(pat, M)
        level = MgLevel(n=n, K=K, M=Mp, P=None, R=None, dK=K.diagonal().copy(), dM=Mp.diagonal().copy(), cand=cand.copy())
        levels.append(level)
        if n <= coarsest or lvl == max_levels - 1:
            break
        S = _strength_graph(K, theta * (0.5 ** lvl))
        agg = _aggregate(S)
        n_c = int(agg.max()) + 1
        if n_c >= n * 0.9 or n_c < 2:
            break
        # tentative prolongator from the near-null-space candidate B (B = 1 on the finest level; on
        # coarser levels its coarse representation, so that P_l B_{l+1} = B_l exactly)
        norms = np.sqrt(np.bincount(agg, weights=cand * cand, minlength=n_c))
        Pt = sp.csr_matrix((cand / norms[agg], (np.arange(n), agg)), shape=(n, n_c))
        cand = norms
        d = K.diagonal()
        rho = float(np.max(np.asarray(abs(K).sum(axis=1)).ravel() / d))        # Gershgorin bound of rho(D^-1 K)
        P = (Pt - sp.diags(omega_p / rho / d) @ (K @ Pt)).tocsr()
        P.sort_indices()
        level.P = P
        level.R = P.T.tocsr()
        level.R.sort_indices()
        pat_ap = _pattern_union(abs(K) @ abs(P), abs(Mp) @ abs(P), abs(P))
        level.KP = _align_to_pattern(pat_ap, (K @ P).tocsr())
        level.MP = _align_to_pattern(pat_ap, (Mp @ P).tocsr())
        level.PP = _align_to_pattern(pat_ap, P)
        K = (level.R @ K @ P).tocsr()
        M = (level.R @ Mp @ P).tocsr()
        K.sort_indices()
        M.sort_indices()
    return levels


def hierarchy_summary(levels: List[MgLevel]) -> dict:
    n0, nnz0 = levels[0].n, levels[0].K.nnz
    return {
        "levels": len(levels),
        "sizes": [lv.n for lv in levels],
        "nnz": [int(lv.K.nnz) for lv in levels],
        "grid_complexity": sum(lv.n for lv in levels) / n0,
        "operator_complexity": sum(lv.K.nnz for lv in levels) / nnz0,
    }


def coarse_inverse(level: MgLevel, shift: float) -> np.ndarray:
    """Dense (pseudo-)inverse of K_L + shift M_L on the coarsest level.  With shift == 0 the matrix is
    singular (one null vector); the cutoff is explicit because the null eigenvalue (~1e-17 relative)
    sits right at numpy's default rank threshold."""
    A = level.K.toarray() + shift * level.M.toarray()
    A = 0.5 * (A + A.T)
    if shift == 0.0:
        return np.linalg.pinv(A, rcond=1e-10, hermitian=True)
    return np.linalg.inv(A)


class CpuVcycle:
    """Reference V(nu,nu)-cycle with damped Jacobi (CPU, one shifted system).  Mirrors what the device
    kernels do for every mode; used by tests and for tuning, not by the product path."""

    def __init__(self, levels: List[MgLevel], shift: float, omega: float = 2.0 / 3.0, nu: int = 1):
        self.levels, self.shift, self.omega, self.nu = levels, shift, omega, nu
        self.A = [(lv.K + shift * lv.M).tocsr() for lv in levels]
        self.dinv = [1.0 / (lv.dK + shift * lv.dM) for lv in levels]
        self.coarse_inv = coarse_inverse(levels[-1], shift)

    def cycle(self, b, l=0):
        if l == len(self.levels) - 1:
            return self.coarse_inv @ b
        A, dinv, lv = self.A[l], self.dinv[l], self.levels[l]
        x = self.omega * dinv * b
        for _ in range(self.nu - 1):
            x = x + self.omega * dinv * (b - A @ x)
        xc = self.cycle(lv.R @ (b - A @ x), l + 1)
        x = x + lv.P @ xc
        for _ in range(self.nu):
            x = x + self.omega * dinv * (b - A @ x)
        return x

    def cycle_fused(self, b, l=0):
        """The same V(1,1) cycle in the three-kernel form the device uses (t = A D^-1 b kept, A P precomputed)."""
        if l == len(self.levels) - 1:
            return self.coarse_inv @ b
        A, dinv, lv, w = self.A[l], self.dinv[l], self.levels[l], self.omega
        bt = dinv * b
        t = A @ bt                                            # k_mg_down
        xc = self.cycle_fused(lv.R @ (b - w * t), l + 1)      # k_mg_restrict, recursion
        AP = lv.KP + self.shift * lv.MP
        return w * bt + lv.P @ xc + w * dinv * (b - w * t - AP @ xc)   # k_mg_post

    def __call__(self, b):
        return self.cycle(b)
