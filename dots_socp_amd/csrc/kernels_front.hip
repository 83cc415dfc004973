// Direct solve of the modal surface problems  (K + (sigma_a + eps) M) x_a = b_a  for all time modes a:
// the two triangular sweeps of a multifrontal Cholesky factorisation (replaces the per-iteration
// SuperLU solves of the reference, utils/laplacian_inverse_socp.py:46-60; the factor itself is built once
// per solve by dots_socp_amd/frontal.py, as the reference builds its T+1 LU factors at :40-44).
//
// One nested-dissection tree is shared by all modes.  Node p eliminates n_p separator vertices and
// touches b_p boundary vertices of its ancestors; its dense block per mode is
//     F_p = [ L_pp^-1 ; G_p ],   G_p = A_bs A_ss^-1,   (n_p + b_p) x n_p,   stored [row][col][mode]
// so that BOTH sweeps are batched dense matrix-vector products that stream F once, with the mode index
// fastest (a wavefront reads two 256-byte runs per load at T = 31, like every other kernel of the path):
//     forward   w   = b[sep_p] - (update rows pulled from the two children)
//               y_p = L_pp^-1 w                    rows [0, n_p) of F_p (lower triangle only)
//               u_p = (children's updates on bd_p) + G_p w          rows [n_p, n_p + b_p)
//     backward  x_p = F_p^T [ y_p ; -x[bd_p] ]
// Every child owns one plane of its parent's update buffer W (front-ordered, position map `cmap` precomputed from
// the pull maps of the ABI): the child writes its update rows there, the parent reads its two planes contiguously.
// No two writers share an address, so no atomics are needed and results are deterministic; entries no child
// writes stay zero.  All nodes of one tree height are independent: one launch per height and sweep, a
// workgroup = (node, block of rows | columns), the dot products split over the workgroup's lanes that do not
// index the mode and folded by wave shuffles + LDS.
// HBM-bound: one solve reads sum_p (n_p (n_p + 1) / 2 + b_p n_p) * modes * 8 bytes twice and touches the
// vectors (V * modes * 8 bytes) a handful of times.
//
// MERGED HEIGHTS (dots_front_desc.band_ptr).  On small meshes a launch per tree height costs more than the bytes it
// streams (4-5 us of dependent latencies for a few MB).  The nodes of a band of heights [lo, hi) that hang together
// are therefore combined into ONE sweep node, algebraically and without a new factorisation (k_merge_member):
//     member s, child c inside the band:  L'^-1[rows of s, columns of c's subtree] = -L_s^-1 U_c[rows of sep_s]
//                                         U_s[columns of c's subtree]              = U_c[rows of bd_s] - G_s U_c[rows of sep_s]
//     own columns:                        L_s^-1 and U_s = G_s;       G' = U of the band's top node
// F' = [L'^-1 ; G'] has the shape the sweeps expect (the blocks of unrelated members stay zero and are never read:
// the forward rows carry their first column, the backward columns their row ranges), the nodes below the band write
// their updates into planes of the merged node (nodes whose boundaries do not meet share a plane), and a solve takes
// 2 x (number of bands) launches.  numpy restatement: tests/frontal_cpu.py (merge, solve_merged).
#include "dots_dev.h"

#include <algorithm>
#include <array>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

#ifndef DOTS_FRONT_UNROLL
#define DOTS_FRONT_UNROLL 1
#endif
#ifndef DOTS_FRONT_U2
#define DOTS_FRONT_U2 1      // steps of the dot product in flight with two-mode lanes, forward fold kernel (A/B: -DDOTS_FRONT_U2=2)
#endif
#ifndef DOTS_FRONT_U2B
#define DOTS_FRONT_U2B DOTS_FRONT_U2      // ... backward kernel
#endif

namespace dots {

struct FrontArgs {
    int sh, TP, ncol;          // mode pitch (log2, value) and live modes
};

__device__ __forceinline__ int64_t front_row(const FrontDev &f, int k) { return f.vmap ? f.vmap[k] : k; }

// VEC consecutive modes per lane (VEC = 2: 16-byte loads, twice as many parts q of a dot product per workgroup).
template <int VEC> struct Vd { double v[VEC]; };
template <int VEC> __device__ __forceinline__ Vd<VEC> vload(const double *p) {
    Vd<VEC> o;
    if (VEC == 2) {
        const double2 t = *reinterpret_cast<const double2 *>(p);
        o.v[0] = t.x;
        o.v[VEC - 1] = t.y;
    } else {
        o.v[0] = *p;
    }
    return o;
}
template <int VEC> __device__ __forceinline__ void vstore(double *p, const Vd<VEC> &x) {
    if (VEC == 2) *reinterpret_cast<double2 *>(p) = make_double2(x.v[0], x.v[VEC - 1]);
    else *p = x.v[0];
}

// Sum `acc` over the threads of the workgroup that share their modes (the parts q of the dot products): lanes of a
// wavefront first (xor-shuffles), then one LDS slot per (row, wavefront, mode); the caller reads them back with
// front_folded after the barrier inside.  TPv = TP / VEC lanes hold one row part; TPv <= 64.
template <int NB, int RB, int VEC>
__device__ __forceinline__ void front_fold(Vd<VEC> (&acc)[RB], double *red, int TP, int a, int tid) {
    constexpr int NW = NB / 64;
    const int TPv = TP / VEC;
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
            double s = acc[r].v[c];
            for (int o = 32; o >= TPv; o >>= 1) s += __shfl_xor(s, o, 64);
            acc[r].v[c] = s;
        }
    if ((tid & 63) < TPv) {
#pragma unroll
        for (int r = 0; r < RB; ++r)
#pragma unroll
            for (int c = 0; c < VEC; ++c) red[(r * NW + (tid >> 6)) * TP + a + c] = acc[r].v[c];
    }
    __syncthreads();
}
template <int NB, int VEC>
__device__ __forceinline__ Vd<VEC> front_folded(const double *red, int r, int TP, int a) {
    constexpr int NW = NB / 64;
    Vd<VEC> s;
#pragma unroll
    for (int c = 0; c < VEC; ++c) {
        double t = 0.0;
        for (int w = 0; w < NW; ++w) t += red[(r * NW + w) * TP + a + c];
        s.v[c] = t;
    }
    return s;
}
// one thread per mode and row part when a row of modes is wider than a wavefront (TP / VEC > 64)
template <int NB, int RB>
__device__ __forceinline__ void front_fold_wide(Vd<1> (&acc)[RB], double *red, int tid) {
#pragma unroll
    for (int r = 0; r < RB; ++r) red[r * NB + tid] = acc[r].v[0];
    __syncthreads();
}
template <int NB>
__device__ __forceinline__ Vd<1> front_folded_wide(const double *red, int r, int sh, int a) {
    Vd<1> s;
    s.v[0] = 0.0;
    for (int k = 0; k < (NB >> sh); ++k) s.v[0] += red[r * NB + (k << sh) + a];
    return s;
}

// forward sweep of one band of tree heights.  Workgroup = (node, rb <= RB rows); thread = (VEC modes from a, part q of
// the dot product).  Every load of the loop body is unconditional (rows past the block are clamped to its first row
// and their sums dropped; planes no child writes hold zeros; the upper triangle of L^-1 is stored as zeros), so that
// the compiler issues the RB + 1 + KP loads of a step back to back and waits once.
// RB is the band's exact block size (1, 2 or 4: no duplicate loads); KP the update planes the band's nodes read
// (0 on the bottom band: leaves only).
template <int NB, int RB, bool VMAP, int KP, int VEC>
__global__ __launch_bounds__(NB) void k_front_fwd(FrontArgs g, FrontDev f, const FrontWork *__restrict__ desc, int rb, const double *__restrict__ bhat,
                                                  double *__restrict__ Y) {
    __shared__ double red[RB * (NB / 64) * 64 * VEC];
    const FrontWork wk = desc[blockIdx.x];
    const SweepNode &nd = wk.nd;
    const int row0 = wk.first;
    const int sh = g.sh, tid = threadIdx.x;
    const int shv = VEC == 2 ? sh - 1 : sh;                      // log2 of the lanes per row part
    const int a = (tid & ((g.TP / VEC) - 1)) * VEC, q = tid >> shv, Q = NB >> shv;
    const bool wide = (g.TP / VEC) > 64;                         // only with VEC == 1
    const int n = nd.n, m = n + nd.b;
    const double *__restrict__ Fp = f.F + (nd.foff << sh) + a;
    const double *__restrict__ W0 = f.W + (nd.woff << sh) + a;         // plane 0; plane k is k * m rows further
    const int64_t plane = (int64_t)m << sh;
    const bool live = a < g.ncol;
    const int nr = min(rb, m - row0);

    Vd<VEC> acc[RB];
    const double *rowp[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) {
#pragma unroll
        for (int c = 0; c < VEC; ++c) acc[r].v[c] = 0.0;
        rowp[r] = Fp + (((int64_t)(r < nr ? row0 + r : row0) * n) << sh);
    }
    // where this thread's (first) update row goes in the parent's plane: loaded now, needed after the fold
    const bool upd0 = q < nr && row0 + q >= n;
    const int cm0 = upd0 ? f.cmap[nd.bdoff + (row0 + q - n)] : 0;
    Vd<VEC> cp0[KP > 0 ? KP : 1];      // ... and what the children carried to that row
    if (KP > 0 && upd0 && live) {
#pragma unroll
        for (int k = 0; k < KP; ++k) cp0[k] = vload<VEC>(W0 + k * plane + ((int64_t)(row0 + q) << sh));
    }
    // rows of L^-1 only need the columns j <= i: the block's last row bounds the loop; wk.lo: the first column of the
    // block's rows that is not in a zero block of the merged node
    const int last = row0 + nr - 1;
    const int jmax = wk.end > 0 ? wk.end : (last < n ? last + 1 : n);
    // U steps of the dot product are loaded before the first is used (merged nodes have long rows: a step per
    // memory round trip would leave the workgroup waiting on latency).  One-mode lanes only: measured +1...7 % on the
    // merged small meshes, -2 % on the bandwidth-bound two-mode sweeps of torus100k (profiles/studies/band_cuts.txt)
    constexpr int U = !DOTS_FRONT_UNROLL ? 1 : (VEC > 1 ? DOTS_FRONT_U2 : ((1 + KP + RB) <= 6) ? 4 : (((1 + KP + RB) <= 12) ? 2 : 1));
    if (live) {
        for (int j0 = wk.lo + q; j0 < jmax; j0 += U * Q) {
            Vd<VEC> wb[U], wp[U][KP > 0 ? KP : 1], fv[U][RB];
            bool ok[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = j0 + u * Q;
                ok[u] = u == 0 || j < jmax;
                if (ok[u]) {
                    const int64_t jo = (int64_t)j << sh;
                    const int64_t row = VMAP ? (int64_t)f.vmap[nd.k0 + j] : (int64_t)(nd.k0 + j);
                    wb[u] = vload<VEC>(bhat + (row << sh) + a);
#pragma unroll
                    for (int k = 0; k < KP; ++k) wp[u][k] = vload<VEC>(W0 + k * plane + jo);
#pragma unroll
                    for (int r = 0; r < RB; ++r) fv[u][r] = vload<VEC>(rowp[r] + jo);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int c = 0; c < VEC; ++c) {
                    double w = wb[u].v[c];
                    if (KP > 0) {
                        double t = wp[u][0].v[c];
#pragma unroll
                        for (int k = 1; k < KP; ++k) t += wp[u][k].v[c];
                        w -= t;
                    }
                    if (ok[u]) {
#pragma unroll
                        for (int r = 0; r < RB; ++r) acc[r].v[c] += fv[u][r].v[c] * w;
                    }
                }
        }
    }
    if (VEC == 1 && wide) front_fold_wide<NB, RB>(reinterpret_cast<Vd<1>(&)[RB]>(acc), red, tid);
    else front_fold<NB, RB, VEC>(acc, red, g.TP, a, tid);
    for (int r = q; r < nr && live; r += Q) {
        Vd<VEC> s;
        if (VEC == 1 && wide) s.v[0] = front_folded_wide<NB>(red, r, sh, a).v[0];
        else s = front_folded<NB, VEC>(red, r, g.TP, a);
        const int i = row0 + r;
        if (i < n) {
            vstore<VEC>(Y + (front_row(f, nd.k0 + i) << sh) + a, s);
        } else {   // update row: carry the children's contributions on, hand the sum to the parent's plane
            if (KP > 0) {
                Vd<VEC> cp[KP > 0 ? KP : 1];
#pragma unroll
                for (int k = 0; k < KP; ++k) cp[k] = r == q ? cp0[k] : vload<VEC>(W0 + k * plane + ((int64_t)i << sh));
#pragma unroll
                for (int c = 0; c < VEC; ++c) {
                    double u = cp[0].v[c];
#pragma unroll
                    for (int k = 1; k < KP; ++k) u += cp[k].v[c];
                    s.v[c] += u;
                }
            }
            const int cm = r == q ? cm0 : f.cmap[nd.bdoff + (i - n)];
            vstore<VEC>(f.W + ((nd.parent_w + cm) << sh) + a, s);
        }
    }
}

// ---- forward sweep on bands of SHORT rows: lane groups instead of workgroup folds ----------------------------------------
// On the lower tree heights a row of F has 5-40 columns.  Split over the 16-64 parts of k_front_fwd most lanes load nothing, and
// a wavefront executes ~185 vector + ~105 scalar instructions (address arithmetic, the fold through LDS, its barrier) for ONE row:
// those launches are ISSUE-bound at full occupancy, not memory-bound (SQ counters of the round-2 kernels at torus100k,
// profiles/r03/r03a_torus100k_sq_counters.txt: 3.5 vector loads per wave, active / wave cycles 0.22-0.24 with 34-49 waves in flight).
// Here a wavefront is cut into G = 64 / (TP / VEC) lane groups of TP / VEC lanes (one row of modes each); QW = 2^qw_shift
// consecutive groups share a row of F (QW = 1: a lane walks its row alone), so a wave holds G / QW rows and folds with
// log2(QW) xor-shuffles; the right-hand side w = b - (planes) of the block's columns is formed ONCE per workgroup, in LDS.
// A block never spans two members of a merged node (FrontWork.pad = its rows), so wk.lo is the first column of ALL its rows.
// Sums are formed per part in column order, parts folded pairwise: the order depends on QW only, not on the mode pitch.
constexpr size_t FWD_ROWS_LDS_MAX = 40 * 1024;      // LDS a workgroup of the row kernel may take for w (4 workgroups per CU stay resident)
constexpr double FWD_ROWS_MEAN_MAX = 30.0;          // bands whose rows are longer on average keep the fold kernel unless they read 4+ planes (DOTS_FRONT_ROWS=2: no limit)
constexpr int FWD_ROWS_PAD = 2;      // doubles of padding per staged row of w (rows of exactly TP doubles would share their banks)
template <bool VMAP, int KP, int VEC>
__global__ __launch_bounds__(256) void k_front_fwd_rows(FrontArgs g, FrontDev f, const FrontWork *__restrict__ desc, int qw_shift,
                                                       const double *__restrict__ bhat, double *__restrict__ Y) {
    extern __shared__ __attribute__((aligned(16))) double wsh[];      // [columns of the block][TP + FWD_ROWS_PAD]
    const FrontWork wk = desc[blockIdx.x];
    const SweepNode &nd = wk.nd;
    const int sh = g.sh, tid = threadIdx.x;
    const int shv = VEC == 2 ? sh - 1 : sh;                      // log2 of the lanes per row of modes
    const int a = (tid & ((1 << shv) - 1)) * VEC;
    const int grp = tid >> shv;                                  // lane group of the workgroup
    const int part = grp & ((1 << qw_shift) - 1), QW = 1 << qw_shift;
    const int r = grp >> qw_shift;                               // row of the block
    const int n = nd.n, m = n + nd.b;
    const int row0 = wk.first, nr = wk.pad, lo = wk.lo;
    const int i = row0 + r;
    const bool live = a < g.ncol, rowok = r < nr;
    const int last = row0 + nr - 1;
    const int jmax = wk.end > 0 ? wk.end : (last < n ? last + 1 : n);
    const int64_t plane = (int64_t)m << sh;
    const double *__restrict__ W0 = f.W + (nd.woff << sh) + a;         // plane 0; plane k is k * m rows further
    const int ldw = g.TP + FWD_ROWS_PAD;
    // where an update row goes in the parent's plane, and what the children carried to it: loaded now, needed after the loop
    const bool store = part == 0 && rowok && live;
    const bool upd = store && i >= n;
    const int cm = upd ? f.cmap[nd.bdoff + (i - n)] : 0;
    Vd<VEC> cp[KP > 0 ? KP : 1];
    if (KP > 0 && upd) {
#pragma unroll
        for (int k = 0; k < KP; ++k) cp[k] = vload<VEC>(W0 + k * plane + ((int64_t)i << sh));
    }
    // The row's entries are loaded one BATCH of U columns ahead of their use: the first batch is in flight while w is staged
    // (it does not depend on w), every later one while the batch before it is multiplied -- a row of 25 columns is 4-5 memory
    // round trips instead of 8 (the launches of the middle heights run ONE round of workgroups: their time is that chain)
    constexpr int U = 4;
    const bool walk = rowok && live;
    const int jend = !walk ? 0 : (wk.end > 0 ? wk.end : (i < n ? i + 1 : n));
    const double *__restrict__ Fi = f.F + (nd.foff << sh) + a + (((int64_t)(walk ? i : row0) * n) << sh);
    Vd<VEC> fv[U];
    int j = lo + part;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int ju = j + u * QW;
#pragma unroll
        for (int c = 0; c < VEC; ++c) fv[u].v[c] = 0.0;
        if (ju < jend) fv[u] = vload<VEC>(Fi + ((int64_t)ju << sh));
    }
    if (live) {
        for (int js = lo + grp; js < jmax; js += 256 >> shv) {
            const int64_t row = VMAP ? (int64_t)f.vmap[nd.k0 + js] : (int64_t)(nd.k0 + js);
            Vd<VEC> w = vload<VEC>(bhat + (row << sh) + a);
            if (KP > 0) {
                Vd<VEC> wp[KP > 0 ? KP : 1];
#pragma unroll
                for (int k = 0; k < KP; ++k) wp[k] = vload<VEC>(W0 + k * plane + ((int64_t)js << sh));
#pragma unroll
                for (int c = 0; c < VEC; ++c) {
                    double t = wp[0].v[c];
#pragma unroll
                    for (int k = 1; k < KP; ++k) t += wp[k].v[c];
                    w.v[c] -= t;
                }
            }
            vstore<VEC>(wsh + (js - lo) * ldw + a, w);
        }
    }
    __syncthreads();
    Vd<VEC> acc;
#pragma unroll
    for (int c = 0; c < VEC; ++c) acc.v[c] = 0.0;
    const double *ws = wsh + a - lo * ldw;
    while (j < jend) {
        Vd<VEC> nx[U], wv[U];
        const int jn = j + U * QW;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ju = jn + u * QW;
#pragma unroll
            for (int c = 0; c < VEC; ++c) nx[u].v[c] = 0.0;
            if (ju < jend) nx[u] = vload<VEC>(Fi + ((int64_t)ju << sh));
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ju = j + u * QW;
#pragma unroll
            for (int c = 0; c < VEC; ++c) wv[u].v[c] = 0.0;
            if (ju < jend) wv[u] = vload<VEC>(ws + ju * ldw);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int c = 0; c < VEC; ++c) acc.v[c] += fv[u].v[c] * wv[u].v[c];
#pragma unroll
        for (int u = 0; u < U; ++u) fv[u] = nx[u];
        j = jn;
    }
    for (int o = 1 << shv; o < (1 << (shv + qw_shift)); o <<= 1) {
#pragma unroll
        for (int c = 0; c < VEC; ++c) acc.v[c] += __shfl_xor(acc.v[c], o, 64);
    }
    if (store) {
        if (i < n) {
            vstore<VEC>(Y + (front_row(f, nd.k0 + i) << sh) + a, acc);
        } else {   // update row: carry the children's contributions on, hand the sum to the parent's plane
            if (KP > 0) {
#pragma unroll
                for (int c = 0; c < VEC; ++c) {
                    double u = cp[0].v[c];
#pragma unroll
                    for (int k = 1; k < KP; ++k) u += cp[k].v[c];
                    acc.v[c] += u;
                }
            }
            vstore<VEC>(f.W + ((nd.parent_w + cm) << sh) + a, acc);
        }
    }
}

// backward sweep of one band.  Workgroup = (node, cb <= RB columns of ONE member of the node).  Same load discipline.
// The rows that can be nonzero in these columns: the member's own rows from the block's first column on, the rows of
// its ancestors inside the band (wk.rs / wk.re, wk.lo ranges), the boundary rows.
template <int NB, int RB, bool VMAP, int VEC>
__global__ __launch_bounds__(NB) void k_front_bwd(FrontArgs g, FrontDev f, const FrontWork *__restrict__ desc, int cb, const double *__restrict__ Y,
                                                  double *X) {
    __shared__ double red[RB * (NB / 64) * 64 * VEC];
    const FrontWork wk = desc[blockIdx.x];
    const SweepNode &nd = wk.nd;
    const int col0 = wk.first;
    const int sh = g.sh, tid = threadIdx.x;
    const int shv = VEC == 2 ? sh - 1 : sh;
    const int a = (tid & ((g.TP / VEC) - 1)) * VEC, q = tid >> shv, Q = NB >> shv;
    const bool wide = (g.TP / VEC) > 64;
    const int n = nd.n, m = n + nd.b;
    const double *__restrict__ Fp = f.F + (nd.foff << sh) + a;
    const int *__restrict__ bdv = f.bd_vertex + nd.bdoff;
    const bool live = a < g.ncol;
    const int nc = min(cb, wk.end - col0);

    Vd<VEC> acc[RB];
    int64_t co[RB];      // column offsets (columns past the block: its first column, sums dropped)
#pragma unroll
    for (int r = 0; r < RB; ++r) {
#pragma unroll
        for (int c = 0; c < VEC; ++c) acc[r].v[c] = 0.0;
        co[r] = (int64_t)(r < nc ? col0 + r : col0) << sh;
    }
    constexpr int U = !DOTS_FRONT_UNROLL ? 1 : (VEC > 1 ? DOTS_FRONT_U2B : ((1 + RB) <= 4) ? 4 : 2);      // as in the forward sweep
    if (live) {
        // rows of the separators: y.  Column i of L^-1 is zero above the diagonal: start at the block's first column
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            if (rg < wk.lo) {
                const int r1 = wk.re[rg];
                for (int j0 = wk.rs[rg] + q; j0 < r1; j0 += U * Q) {
                    Vd<VEC> v[U], fv[U][RB];
                    bool ok[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int j = j0 + u * Q;
                        ok[u] = u == 0 || j < r1;
                        if (ok[u]) {
                            const int64_t row = VMAP ? (int64_t)f.vmap[nd.k0 + j] : (int64_t)(nd.k0 + j);
                            v[u] = vload<VEC>(Y + (row << sh) + a);
                            const double *__restrict__ Fj = Fp + (((int64_t)j * n) << sh);
#pragma unroll
                            for (int r = 0; r < RB; ++r) fv[u][r] = vload<VEC>(Fj + co[r]);
                        }
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u)
                        if (ok[u]) {
#pragma unroll
                            for (int r = 0; r < RB; ++r)
#pragma unroll
                                for (int c = 0; c < VEC; ++c) acc[r].v[c] += fv[u][r].v[c] * v[u].v[c];
                        }
                }
            }
        }
        // boundary rows: -x of the ancestors (written by the launches of the bands above)
        for (int j0 = n + q; j0 < m; j0 += U * Q) {
            Vd<VEC> v[U], fv[U][RB];
            bool ok[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = j0 + u * Q;
                ok[u] = u == 0 || j < m;
                if (ok[u]) {
                    v[u] = vload<VEC>(X + ((int64_t)bdv[j - n] << sh) + a);
                    const double *__restrict__ Fj = Fp + (((int64_t)j * n) << sh);
#pragma unroll
                    for (int r = 0; r < RB; ++r) fv[u][r] = vload<VEC>(Fj + co[r]);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (ok[u]) {
#pragma unroll
                    for (int r = 0; r < RB; ++r)
#pragma unroll
                        for (int c = 0; c < VEC; ++c) acc[r].v[c] -= fv[u][r].v[c] * v[u].v[c];
                }
        }
    }
    if (VEC == 1 && wide) front_fold_wide<NB, RB>(reinterpret_cast<Vd<1>(&)[RB]>(acc), red, tid);
    else front_fold<NB, RB, VEC>(acc, red, g.TP, a, tid);
    for (int r = q; r < nc && live; r += Q) {
        Vd<VEC> s;
        if (VEC == 1 && wide) s.v[0] = front_folded_wide<NB>(red, r, sh, a).v[0];
        else s = front_folded<NB, VEC>(red, r, g.TP, a);
        vstore<VEC>(X + (front_row(f, nd.k0 + col0 + r) << sh) + a, s);
    }
}

// ---- the leaves as explicit local inverses ------------------------------------------------------------------------------
// A leaf p of the tree has no children: its front holds ORIGINAL matrix entries only, A_ss = K_ss + (sigma_a + eps) M_ss and the coupling
// A_bs = K_bs -- sparse (a boundary vertex touches two or three vertices of the leaf) and the SAME for every mode (M is diagonal).  The band
// kernels nevertheless stream the dense G_p = K_bs A_ss^-1 (b x n per mode: three quarters of a leaf's block, and the leaves' band is a fifth of the
// factor on a large mesh).  Here a leaf stores S_p = A_ss^-1 = L^-T L^-1 (symmetric: its lower triangle packed by rows, k_top_inverse from its L^-1)
// and both sweeps take the coupling from the CSR of K that the context holds anyway:
//     forward    t = S_p b[sep_p]                       u_i = sum_{v in sep_p} K[bd_i, v] t_v        -> the leaf's plane of its parent
//     backward   x[sep_p] = S_p (b[sep_p] - g),         g_j = sum_{u not in sep_p} K[sep_j, u] x_u   (every such u is a boundary vertex of the leaf)
// n (n + 1) / 2 entries per leaf, mode and sweep instead of n (n + 1) / 2 + b n (torus100k: 85 instead of 329 MB per sweep; a workgroup reads every
// entry twice, as a row and as a column entry: the second time from its caches), no y of the leaves stored.
// One workgroup per leaf; thread = (VEC modes, lane group); a lane group owns a row.  Needs the device numbering to be the sweep order
// (FrontDev::vmap == nullptr: a vertex's position in its leaf is its index minus k0) and an un-merged band of leaves.
// Row i of the symmetric S from its packed lower triangle P[i (i + 1) / 2 + j] (j <= i): the entries left of the diagonal are one run, those right of
// it are column i of the rows below (every entry is one run of modes: the lanes of a group still load 16 consecutive bytes each).  The first
// LEAF_PRE entries of the row are loaded into registers BEFORE the vector they multiply is staged (leaf_row_load: the factor streams from memory while
// the workgroup gathers its right-hand side), the rest -- leaves of more than LEAF_PRE vertices: degenerate cuts only -- behind it.
// (A/B on one box, -DDOTS_LEAF_PRE=16 / 8 / 4: torus100k solve 671 / 660 / 660 us -- 94 VGPRs and 5 waves per SIMD against ~60 and 8 --, torus65k_T127 1 422 / 1 426 / 1 424)
#ifndef DOTS_LEAF_PRE
#define DOTS_LEAF_PRE 8
#endif
constexpr int LEAF_PRE = DOTS_LEAF_PRE;
__device__ __forceinline__ int64_t leaf_entry(int i, int j) { return j <= i ? (int64_t)i * (i + 1) / 2 + j : (int64_t)j * (j + 1) / 2 + i; }
template <int VEC>
__device__ __forceinline__ void leaf_row_load(Vd<VEC> (&s)[LEAF_PRE], const double *__restrict__ P, int i, int n, int sh) {
#pragma unroll
    for (int u = 0; u < LEAF_PRE; ++u) {
#pragma unroll
        for (int c = 0; c < VEC; ++c) s[u].v[c] = 0.0;
        if (u < n && i < n) s[u] = vload<VEC>(P + (leaf_entry(i, u) << sh));
    }
}
template <int VEC>
__device__ __forceinline__ Vd<VEC> leaf_row_dot(const Vd<VEC> (&s)[LEAF_PRE], const double *__restrict__ P, const double *vsh, int i, int n, int TP, int sh) {
    Vd<VEC> acc;
#pragma unroll
    for (int c = 0; c < VEC; ++c) acc.v[c] = 0.0;
#pragma unroll
    for (int u = 0; u < LEAF_PRE; ++u)
        if (u < n) {
            const Vd<VEC> v = vload<VEC>(vsh + u * TP);
#pragma unroll
            for (int c = 0; c < VEC; ++c) acc.v[c] += s[u].v[c] * v.v[c];
        }
    for (int j = LEAF_PRE; j < n; ++j) {
        const Vd<VEC> s0 = vload<VEC>(P + (leaf_entry(i, j) << sh)), v0 = vload<VEC>(vsh + j * TP);
#pragma unroll
        for (int c = 0; c < VEC; ++c) acc.v[c] += s0.v[c] * v0.v[c];
    }
    return acc;
}
template <int VEC>
__device__ __forceinline__ Vd<VEC> leaf_row(const double *__restrict__ P, const double *vsh, int i, int n, int TP, int sh) {
    Vd<VEC> s[LEAF_PRE];
    leaf_row_load<VEC>(s, P, i, n, sh);
    return leaf_row_dot<VEC>(s, P, vsh, i, n, TP, sh);
}

template <int VEC, int NB, bool TAB>
__global__ __launch_bounds__(NB) void k_front_leaf_fwd(FrontArgs g, FrontDev f, const int *__restrict__ rowptr, const int *__restrict__ col,
                                                       const double *__restrict__ val, const double *__restrict__ bhat) {
    extern __shared__ __attribute__((aligned(16))) double lsh[];      // w [n][TP], then t [n][TP]
    const LeafWork lw = f.leaf_desc[blockIdx.x];
    const int sh = g.sh, TP = g.TP, tid = threadIdx.x;
    const int shv = VEC == 2 ? sh - 1 : sh;
    const int a = (tid & ((1 << shv) - 1)) * VEC, grp = tid >> shv, NG = NB >> shv;
    const int n = lw.n, b = lw.b, k0 = lw.k0;
    const bool live = a < g.ncol;
    double *wsh = lsh + a, *tsh = lsh + n * TP + a;
    const double *__restrict__ S = f.leafS + (lw.soff << sh) + a;
    Vd<VEC> srow[LEAF_PRE];
    if (live) leaf_row_load<VEC>(srow, S, grp, n, sh);
    // the first boundary row(s) of this lane group, loaded now and needed after the barriers: its record, or (no records) its vertex, where it
    // goes in the parent's plane and its row of K
    const bool upd0 = grp < b, upd1 = grp + NG < b;
    const LeafBdRow *__restrict__ br = TAB ? f.leaf_bd + lw.rowoff : nullptr;
    int cm0 = 0, ru0[LEAF_KC];
    double rv0[LEAF_KC];
    int vb0 = 0, vb1 = 0, cm1 = 0, eb0 = 0, ee0 = 0, eb1 = 0, ee1 = 0;
    if (TAB) {
#pragma unroll
        for (int k = 0; k < LEAF_KC; ++k) { ru0[k] = 0; rv0[k] = 0.0; }
        if (upd0) {
            cm0 = br[grp].cm;
#pragma unroll
            for (int k = 0; k < LEAF_KC; ++k) { ru0[k] = br[grp].u[k]; rv0[k] = br[grp].v[k]; }
        }
    } else {
        vb0 = upd0 ? f.bd_vertex[lw.bdoff + grp] : 0; cm0 = upd0 ? f.cmap[lw.bdoff + grp] : 0;
        vb1 = upd1 ? f.bd_vertex[lw.bdoff + grp + NG] : 0; cm1 = upd1 ? f.cmap[lw.bdoff + grp + NG] : 0;
        eb0 = upd0 ? rowptr[vb0] : 0; ee0 = upd0 ? rowptr[vb0 + 1] : 0;
        eb1 = upd1 ? rowptr[vb1] : 0; ee1 = upd1 ? rowptr[vb1 + 1] : 0;
    }
    if (live)
        for (int j = grp; j < n; j += NG) vstore<VEC>(wsh + j * TP, vload<VEC>(bhat + ((int64_t)(k0 + j) << sh) + a));
    __syncthreads();
    if (live) {
        if (grp < n) vstore<VEC>(tsh + grp * TP, leaf_row_dot<VEC>(srow, S, wsh, grp, n, TP, sh));
        for (int i = grp + NG; i < n; i += NG) vstore<VEC>(tsh + i * TP, leaf_row<VEC>(S, wsh, i, n, TP, sh));
    }
    __syncthreads();
    if (!live) return;
    for (int r = grp; r < b; r += NG) {
        Vd<VEC> acc;
#pragma unroll
        for (int c = 0; c < VEC; ++c) acc.v[c] = 0.0;
        int cm;
        if (TAB) {      // (padded entries: value 0 at position 0 -- t is finite)
            cm = cm0;
            if (r != grp) {
                cm = br[r].cm;
#pragma unroll
                for (int k = 0; k < LEAF_KC; ++k) { ru0[k] = br[r].u[k]; rv0[k] = br[r].v[k]; }
            }
#pragma unroll
            for (int k = 0; k < LEAF_KC; ++k) {
                const Vd<VEC> t = vload<VEC>(tsh + ru0[k] * TP);
#pragma unroll
                for (int c = 0; c < VEC; ++c) acc.v[c] += rv0[k] * t.v[c];
            }
        } else {
            const bool p0 = r == grp, p1 = r == grp + NG;
            const int vb = p0 ? vb0 : (p1 ? vb1 : f.bd_vertex[lw.bdoff + r]);
            cm = p0 ? cm0 : (p1 ? cm1 : f.cmap[lw.bdoff + r]);
            const int e1 = p0 ? ee0 : (p1 ? ee1 : rowptr[vb + 1]);
            for (int e = p0 ? eb0 : (p1 ? eb1 : rowptr[vb]); e < e1; ++e) {
                const unsigned u = (unsigned)(col[e] - k0);
                if (u < (unsigned)n) {
                    const double kv = val[e];
                    const Vd<VEC> t = vload<VEC>(tsh + (int)u * TP);
#pragma unroll
                    for (int c = 0; c < VEC; ++c) acc.v[c] += kv * t.v[c];
                }
            }
        }
        vstore<VEC>(f.W + ((lw.parent_w + cm) << sh) + a, acc);
    }
}

template <int VEC, int NB, bool TAB>
__global__ __launch_bounds__(NB) void k_front_leaf_bwd(FrontArgs g, FrontDev f, const int *__restrict__ rowptr, const int *__restrict__ col,
                                                       const double *__restrict__ val, const double *__restrict__ bhat, double *X) {
    extern __shared__ __attribute__((aligned(16))) double lsh[];      // b[sep] - g  [n][TP]
    const LeafWork lw = f.leaf_desc[blockIdx.x];
    const int sh = g.sh, TP = g.TP, tid = threadIdx.x;
    const int shv = VEC == 2 ? sh - 1 : sh;
    const int a = (tid & ((1 << shv) - 1)) * VEC, grp = tid >> shv, NG = NB >> shv;
    const int n = lw.n, k0 = lw.k0;
    const bool live = a < g.ncol;
    double *rsh = lsh + a;
    const double *__restrict__ S = f.leafS + (lw.soff << sh) + a;
    Vd<VEC> srow[LEAF_PRE];
    if (live) leaf_row_load<VEC>(srow, S, grp, n, sh);
    if (live)
        for (int j = grp; j < n; j += NG) {
            const int v = k0 + j;
            Vd<VEC> acc = vload<VEC>(bhat + ((int64_t)v << sh) + a);
            if (TAB) {
                const LeafSepRow &R = f.leaf_sep[v];
                const int cnt = R.cnt;
                for (int k = 0; k < cnt; ++k) {
                    const double kv = R.v[k];
                    const Vd<VEC> x = vload<VEC>(X + ((int64_t)R.u[k] << sh) + a);
#pragma unroll
                    for (int c = 0; c < VEC; ++c) acc.v[c] -= kv * x.v[c];
                }
            } else {
                const int e1 = rowptr[v + 1];
                for (int e = rowptr[v]; e < e1; ++e) {
                    const int u = col[e];
                    if ((unsigned)(u - k0) >= (unsigned)n) {      // outside the leaf: a boundary vertex, solved by a launch of the bands above
                        const double kv = val[e];
                        const Vd<VEC> x = vload<VEC>(X + ((int64_t)u << sh) + a);
#pragma unroll
                        for (int c = 0; c < VEC; ++c) acc.v[c] -= kv * x.v[c];
                    }
                }
            }
            vstore<VEC>(rsh + j * TP, acc);
        }
    __syncthreads();
    if (!live) return;
    if (grp < n) vstore<VEC>(X + ((int64_t)(k0 + grp) << sh) + a, leaf_row_dot<VEC>(srow, S, rsh, grp, n, TP, sh));
    for (int i = grp + NG; i < n; i += NG) vstore<VEC>(X + ((int64_t)(k0 + i) << sh) + a, leaf_row<VEC>(S, rsh, i, n, TP, sh));
}

// the coupling records of every leaf from the CSR (once per factorisation); *overflow is set when a row holds more entries than a record
__global__ __launch_bounds__(64) void k_leaf_tables(FrontDev f, const int *__restrict__ rowptr, const int *__restrict__ col, const double *__restrict__ val,
                                                   LeafBdRow *__restrict__ bt, LeafSepRow *__restrict__ st, int *overflow) {
    const LeafWork lw = f.leaf_desc[blockIdx.x];
    const int n = lw.n, k0 = lw.k0;
    for (int r = threadIdx.x; r < lw.b; r += 64) {
        LeafBdRow R{};
        R.cm = f.cmap[lw.bdoff + r];
        const int vb = f.bd_vertex[lw.bdoff + r];
        int cnt = 0;
        for (int e = rowptr[vb]; e < rowptr[vb + 1]; ++e) {
            const unsigned u = (unsigned)(col[e] - k0);
            if (u < (unsigned)n) {
                if (cnt < LEAF_KC) { R.u[cnt] = (int)u; R.v[cnt] = val[e]; }
                ++cnt;
            }
        }
        if (cnt > LEAF_KC) atomicOr(overflow, 1);
        R.cnt = cnt < LEAF_KC ? cnt : LEAF_KC;
        bt[lw.rowoff + r] = R;
    }
    for (int j = threadIdx.x; j < n; j += 64) {
        LeafSepRow R{};
        const int v = k0 + j;
        int cnt = 0;
        for (int e = rowptr[v]; e < rowptr[v + 1]; ++e) {
            const int u = col[e];
            if ((unsigned)(u - k0) >= (unsigned)n) {
                if (cnt < LEAF_KE) { R.u[cnt] = u; R.v[cnt] = val[e]; }
                ++cnt;
            }
        }
        if (cnt > LEAF_KE) atomicOr(overflow, 1);
        R.cnt = cnt < LEAF_KE ? cnt : LEAF_KE;
        st[v] = R;
    }
}

// ---- merged bands: F' of a merged node from its members' blocks (see the header comment) ---------------------
struct MergeArgs {
    int sh, TP, ncol;
    double *F;                  // original blocks, then the merged ones
    double *scratch;            // U_s of the members that are neither at the bottom nor at the top of their band
    const int *pull0, *pull1;   // front position -> row in the child's boundary, or -1
    const MergeMember *mem;
    const int *list;            // member records this launch handles (one tree height of one band)
};

// grid (blocks of entries, members).  A thread = (mode, entry (i, column) of [rows of s's front] x [columns of s's
// subtree in the merged node]); the members' children inside the band were handled by the launches before.
__global__ __launch_bounds__(256) void k_merge_member(MergeArgs g) {
    const MergeMember s = g.mem[g.list[blockIdx.y]];
    const int sh = g.sh, tid = threadIdx.x;
    const int a = tid & (g.TP - 1), q = tid >> sh, Q = 256 >> sh;
    if (a >= g.ncol) return;
    const int n = s.n, m = n + s.b, w = s.o + n - s.c0;
    const double *__restrict__ Fs = g.F + (s.foff << sh) + a;                 // entry (i, t): ((i * n + t) << sh)
    double *__restrict__ Fd = g.F + (s.dst << sh) + a;                        // entry (row, col): ((row * ns + col) << sh)
    double *__restrict__ Us = (s.uin == 1 ? g.scratch : g.F) + (s.uoff << sh) + a;
    const int64_t total = (int64_t)m * w;
    for (int64_t e = (int64_t)blockIdx.x * Q + q; e < total; e += (int64_t)gridDim.x * Q) {
        const int i = (int)(e / w), jj = (int)(e % w), col = s.c0 + jj;
        double val;
        if (col >= s.o) {
            if (i >= n && s.uin == 0) continue;           // its own G rows already are U_s
            val = Fs[((int64_t)i * n + (col - s.o)) << sh];
        } else {
            int k = 0;
            MergeMember c = g.mem[s.ch[0] >= 0 ? s.ch[0] : s.ch[1]];
            if (s.ch[0] < 0 || col < c.c0 || col >= c.o + c.n) { k = 1; c = g.mem[s.ch[1]]; }
            const int *__restrict__ pull = (k == 0 ? g.pull0 : g.pull1) + s.ioff;
            const double *__restrict__ Uc = (c.uin == 1 ? g.scratch : g.F) + (c.uoff << sh) + a + ((int64_t)(col - c.c0) << sh);
            const double *__restrict__ Fi = Fs + (((int64_t)i * n) << sh);
            const int tmax = i < n ? i + 1 : n;
            double acc = 0.0;
            for (int t = 0; t < tmax; ++t) {
                const int r = pull[t];
                if (r >= 0) acc += Fi[(int64_t)t << sh] * Uc[((int64_t)r * c.ustride) << sh];
            }
            if (i < n) val = -acc;
            else {
                const int r = pull[i];
                val = (r >= 0 ? Uc[((int64_t)r * c.ustride) << sh] : 0.0) - acc;
            }
        }
        if (i < n) Fd[((int64_t)(s.o + i) * s.ns + col) << sh] = val;
        else Us[((int64_t)(i - n) * s.ustride + jj) << sh] = val;
    }
}

// ---- top band as an explicit inverse (dots_front_desc.top_inverse): S^-1 = L'^-T L'^-1 of a node without boundary rows ----
// grid (blocks of entries, nodes); thread = (mode, entry (i, j)); L = the node's merged block, row stride n
struct TopInvArgs {
    int sh, TP, ncol;
    const double *F;
    double *out;                // S^-1 of every node of the list, one after the other
    const int64_t *foff, *ooff; // per node: its block in F, its block in out
    const int *n;
    int packed;                 // 1: only the lower triangle is written, packed by rows (out[i (i + 1) / 2 + j], j <= i; ooff counts packed entries)
};
__global__ __launch_bounds__(256) void k_top_inverse(TopInvArgs g) {
    const int nd = blockIdx.y, n = g.n[nd];
    const int sh = g.sh, tid = threadIdx.x;
    const int a = tid & (g.TP - 1), q = tid >> sh, Q = 256 >> sh;
    if (a >= g.ncol) return;
    const double *__restrict__ L = g.F + (g.foff[nd] << sh) + a;
    double *__restrict__ S = g.out + (g.ooff[nd] << sh) + a;
    const int64_t total = (int64_t)n * n;
    for (int64_t e = (int64_t)blockIdx.x * Q + q; e < total; e += (int64_t)gridDim.x * Q) {
        const int i = (int)(e / n), j = (int)(e % n);
        if (g.packed && j > i) continue;
        double s0 = 0.0, s1 = 0.0;
        int k = max(i, j);
        for (; k + 2 <= n; k += 2) {
            s0 += L[((int64_t)k * n + i) << sh] * L[((int64_t)k * n + j) << sh];
            s1 += L[((int64_t)(k + 1) * n + i) << sh] * L[((int64_t)(k + 1) * n + j) << sh];
        }
        for (; k < n; ++k) s0 += L[((int64_t)k * n + i) << sh] * L[((int64_t)k * n + j) << sh];
        S[(g.packed ? (int64_t)i * (i + 1) / 2 + j : e) << sh] = s0 + s1;
    }
}

// reads `n` doubles and writes nothing (the sum is never NaN-compared true): evicts the caches without leaving dirty lines (tuner)
__global__ __launch_bounds__(256) void k_flush_read(const double *__restrict__ x, int64_t n, double *sink) {
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) s += x[i];
    if (s == 1.2345e300) *sink = s;
}

// two modes per lane (16-byte loads, half the waves).  Round 1 (one launch per tree height): +6 % at torus100k, +13 % at T = 127,
// -2 % on the latency-bound sphere10k; with merged bands it pays there too (knot solve 53.6 -> 51.9 us, sphere10k 97 -> 95.5 us):
// on wherever the pitch allows.  DOTS_FRONT_VEC2=0 turns it off, =3 restores the round-1 rule (pitch >= 64 or a factor > 1 GB).
static bool front_two_modes(const Ctx *c) {
    const Dev &d = c->dcg;
    if (!c->front_vec2 || d.TP < 4 || d.TP > 128) return false;
    return c->front_vec2 != 3 || d.TP >= 64 || c->front_bytes > 1.0e9;
}

// one band of the forward sweep: n workgroups of nbt threads, blk rows each, kp update planes per node
static void front_launch_fwd(Ctx *c, const FrontDev &f, const FrontWork *ptr, int n, int nbt, int blk, int kp, const double *bhat, double *y) {
    const Dev &d = c->dcg;
    const FrontArgs g{d.tp_shift, d.TP, d.cg_ncol};
    const bool vm = f.vmap != nullptr, v2 = front_two_modes(c);
#define FRONT_FWD4(NBV, RBV, KPV)                                                                                                  \
    do {                                                                                                                           \
        if (v2) {                                                                                                                  \
            if (vm) hipLaunchKernelGGL((k_front_fwd<NBV, RBV, true, KPV, 2>), dim3(n), dim3(NBV), 0, c->stream, g, f, ptr, blk, bhat, y);  \
            else hipLaunchKernelGGL((k_front_fwd<NBV, RBV, false, KPV, 2>), dim3(n), dim3(NBV), 0, c->stream, g, f, ptr, blk, bhat, y);    \
        } else {                                                                                                                   \
            if (vm) hipLaunchKernelGGL((k_front_fwd<NBV, RBV, true, KPV, 1>), dim3(n), dim3(NBV), 0, c->stream, g, f, ptr, blk, bhat, y);  \
            else hipLaunchKernelGGL((k_front_fwd<NBV, RBV, false, KPV, 1>), dim3(n), dim3(NBV), 0, c->stream, g, f, ptr, blk, bhat, y);    \
        }                                                                                                                          \
    } while (0)
#define FRONT_FWD(NBV, RBV)                                                                                                        \
    do {                                                                                                                           \
        if (kp == 0) FRONT_FWD4(NBV, RBV, 0);                                                                                      \
        else if (kp == 2) FRONT_FWD4(NBV, RBV, 2);                                                                                 \
        else if (kp == 4) FRONT_FWD4(NBV, RBV, 4);                                                                                 \
        else FRONT_FWD4(NBV, RBV, 8);                                                                                              \
    } while (0)
    if (nbt == 1024) { if (blk == 1) FRONT_FWD(1024, 1); else if (blk == 2) FRONT_FWD(1024, 2); else FRONT_FWD(1024, 4); }
    else { if (blk == 1) FRONT_FWD(256, 1); else if (blk == 2) FRONT_FWD(256, 2); else FRONT_FWD(256, 4); }
#undef FRONT_FWD
#undef FRONT_FWD4
}

// one band of the forward sweep with the row kernel: n workgroups of 256 threads, 2^qw_shift lane groups per row
static void front_launch_fwd_rows(Ctx *c, const FrontDev &f, const FrontWork *ptr, int n, int qw_shift, int kp, int lds_cols, const double *bhat, double *y) {
    const Dev &d = c->dcg;
    const FrontArgs g{d.tp_shift, d.TP, d.cg_ncol};
    const bool vm = f.vmap != nullptr, v2 = front_two_modes(c);
    const size_t lds = sizeof(double) * (size_t)std::max(lds_cols, 1) * (size_t)(d.TP + FWD_ROWS_PAD);
#define FRONT_ROWS4(KPV)                                                                                                           \
    do {                                                                                                                           \
        if (v2) {                                                                                                                  \
            if (vm) hipLaunchKernelGGL((k_front_fwd_rows<true, KPV, 2>), dim3(n), dim3(256), lds, c->stream, g, f, ptr, qw_shift, bhat, y);  \
            else hipLaunchKernelGGL((k_front_fwd_rows<false, KPV, 2>), dim3(n), dim3(256), lds, c->stream, g, f, ptr, qw_shift, bhat, y);    \
        } else {                                                                                                                   \
            if (vm) hipLaunchKernelGGL((k_front_fwd_rows<true, KPV, 1>), dim3(n), dim3(256), lds, c->stream, g, f, ptr, qw_shift, bhat, y);  \
            else hipLaunchKernelGGL((k_front_fwd_rows<false, KPV, 1>), dim3(n), dim3(256), lds, c->stream, g, f, ptr, qw_shift, bhat, y);    \
        }                                                                                                                          \
    } while (0)
    if (kp == 0) FRONT_ROWS4(0);
    else if (kp == 2) FRONT_ROWS4(2);
    else if (kp == 4) FRONT_ROWS4(4);
    else FRONT_ROWS4(8);
#undef FRONT_ROWS4
}

static void front_launch_bwd(Ctx *c, const FrontDev &f, const FrontWork *ptr, int n, int nbt, int blk, const double *y, double *x) {
    const Dev &d = c->dcg;
    const FrontArgs g{d.tp_shift, d.TP, d.cg_ncol};
    const bool vm = f.vmap != nullptr, v2 = front_two_modes(c);
#define FRONT_BWD(NBV, RBV)                                                                                                        \
    do {                                                                                                                           \
        if (v2) {                                                                                                                  \
            if (vm) hipLaunchKernelGGL((k_front_bwd<NBV, RBV, true, 2>), dim3(n), dim3(NBV), 0, c->stream, g, f, ptr, blk, y, x);  \
            else hipLaunchKernelGGL((k_front_bwd<NBV, RBV, false, 2>), dim3(n), dim3(NBV), 0, c->stream, g, f, ptr, blk, y, x);    \
        } else {                                                                                                                   \
            if (vm) hipLaunchKernelGGL((k_front_bwd<NBV, RBV, true, 1>), dim3(n), dim3(NBV), 0, c->stream, g, f, ptr, blk, y, x);  \
            else hipLaunchKernelGGL((k_front_bwd<NBV, RBV, false, 1>), dim3(n), dim3(NBV), 0, c->stream, g, f, ptr, blk, y, x);    \
        }                                                                                                                          \
    } while (0)
    if (nbt == 1024) { if (blk == 1) FRONT_BWD(1024, 1); else if (blk == 2) FRONT_BWD(1024, 2); else FRONT_BWD(1024, 4); }
    else { if (blk == 1) FRONT_BWD(256, 1); else if (blk == 2) FRONT_BWD(256, 2); else FRONT_BWD(256, 4); }
#undef FRONT_BWD
}

void front_release(Ctx *c) {
    for (int i = 0; i < c->n_front_allocs; ++i) (void)hipFree(c->front_allocs[i]);
    c->n_front_allocs = 0;
    c->front = FrontDev{};
    c->use_front = 0;
    c->front_bytes = c->front_bytes_unmerged = 0.0;
    c->front_heights = 0;
    c->front_top_inverse = 0;
}

namespace {
struct Group {                 // one node of the sweeps
    int root = -1, band = 0;
    std::vector<int> members;  // original nodes, ascending = children first
    int n = 0, b = 0, k0 = 0, planes = 0, colour = -1, parent = -1;
    int64_t foff = 0, woff = 0;
};
}  // namespace

int front_setup(Ctx *c, const dots_front_desc *h) {
    const Dev &d = c->dcg;
    auto bad = [&](const char *what) {
        set_error(std::string("front_setup: ") + what);
        return (int)DOTS_ERR_ARGUMENT;
    };
    if (!h || h->n_nodes < 1 || h->n_levels < 1 || h->n_levels > 64) return bad("bad description");
    if (!h->node_n || !h->node_b || !h->node_foff || !h->node_ioff || !h->node_uoff || !h->node_child || !h->front_idx || !h->pull0 ||
        !h->pull1 || !h->level_ptr || !h->level_nodes || (!h->values && !h->grounded))
        return bad("null array");
    if (h->pitch != d.TP || h->n_modes != d.cg_ncol) return bad("pitch / mode count does not match the context");
    // ---- index sanity: a wrong index would fault on the device -------------------------------------
    const int nn = h->n_nodes;
    int64_t fo = 0, io = 0, uo = 0, eliminated = 0;
    double entries_unmerged = 0.0;
    for (int p = 0; p < nn; ++p) {
        const int64_t n = h->node_n[p], b = h->node_b[p];
        if (n < 0 || b < 0) return bad("node size");      // (n + b = 0: the empty top separator of a mesh of several components)
        if (h->node_foff[p] != fo || h->node_ioff[p] != io || h->node_uoff[p] != uo) return bad("node offsets are not the running sums");
        for (int k = 0; k < 2; ++k) {
            const int ch = h->node_child[2 * p + k];
            if (ch < -1 || ch >= p) return bad("child index (nodes must be numbered children first)");
        }
        const int c0 = h->node_child[2 * p], c1 = h->node_child[2 * p + 1];
        for (int64_t i = 0; i < n + b; ++i) {
            const int v = h->front_idx[io + i];
            if (v < 0 || v >= d.V) return bad("front vertex out of range");
            const int k0 = h->pull0[io + i], k1 = h->pull1[io + i];
            if (k0 < -1 || (k0 >= 0 && (c0 < 0 || k0 >= h->node_b[c0]))) return bad("pull0 out of range");
            if (k1 < -1 || (k1 >= 0 && (c1 < 0 || k1 >= h->node_b[c1]))) return bad("pull1 out of range");
        }
        fo += (n + b) * n;
        io += n + b;
        uo += b;
        eliminated += n;
        entries_unmerged += 0.5 * n * (n + 1) + (double)b * n;
    }
    if (fo != h->n_entries || io != h->n_front_rows || uo != h->update_rows || eliminated != d.V) return bad("totals do not match");
    if (h->level_ptr[0] != 0 || h->level_ptr[h->n_levels] != nn) return bad("level_ptr");
    std::vector<int> level_of(nn, -1), parent(nn, -1);
    for (int l = 0; l < h->n_levels; ++l) {
        if (h->level_ptr[l + 1] < h->level_ptr[l]) return bad("level_ptr not monotone");
        for (int k = h->level_ptr[l]; k < h->level_ptr[l + 1]; ++k) {
            const int p = h->level_nodes[k];
            if (p < 0 || p >= nn || level_of[p] != -1) return bad("level_nodes is not a permutation");
            level_of[p] = l;
        }
    }
    for (int p = 0; p < nn; ++p)
        for (int k = 0; k < 2; ++k) {
            const int ch = h->node_child[2 * p + k];
            if (ch < 0) continue;
            if (level_of[ch] >= level_of[p]) return bad("a child is not on a lower level than its parent");
            if (parent[ch] != -1) return bad("a node has two parents");
            parent[ch] = p;
        }
    for (int p = 0; p < nn; ++p)
        if (parent[p] == -1 && h->node_b[p] != 0) return bad("a node without parent has boundary rows");
    // ---- bands of tree heights that one launch handles (default: one height each) --------------------
    std::vector<int> cuts;
    if (h->band_ptr) {
        if (h->n_bands < 1 || h->n_bands > h->n_levels) return bad("n_bands");
        for (int k = 0; k <= h->n_bands; ++k) cuts.push_back(h->band_ptr[k]);
        if (cuts.front() != 0 || cuts.back() != h->n_levels) return bad("band_ptr must run from 0 to n_levels");
        for (int k = 0; k < h->n_bands; ++k)
            if (cuts[k + 1] <= cuts[k] || cuts[k + 1] - cuts[k] > 4) return bad("a band holds 1 to 4 tree heights");
    } else {
        for (int l = 0; l <= h->n_levels; ++l) cuts.push_back(l);
    }

    const bool top_inv = h->top_inverse != 0;      // the top band stores explicit inverses (its nodes have no boundary rows)

    DOTS_HIP(hipStreamSynchronize(c->stream));
    front_release(c);
    // ---- the original tree: node records of the factorisation, elimination order --------------------
    std::vector<FrontNode> nodes((size_t)nn);
    std::vector<int> vmap0((size_t)d.V), bd_vertex((size_t)std::max<int64_t>(h->update_rows, 1));
    {
        std::vector<char> seen((size_t)d.V, 0);
        int k0 = 0;
        int64_t soff = 0;
        for (int p = 0; p < nn; ++p) {
            FrontNode &nd = nodes[(size_t)p];
            nd.n = h->node_n[p];
            nd.b = h->node_b[p];
            nd.k0 = k0;
            nd.foff = h->node_foff[p];
            nd.bdoff = h->node_uoff[p];
            nd.parent = parent[p];
            nd.c0 = h->node_child[2 * p];
            nd.c1 = h->node_child[2 * p + 1];
            nd.ioff = h->node_ioff[p];
            nd.soff = soff;
            soff += (int64_t)nd.b * nd.b;
            const int64_t io2 = h->node_ioff[p];
            for (int i = 0; i < nd.n; ++i) {
                const int v = h->front_idx[io2 + i];
                if (seen[(size_t)v]) return bad("a vertex is eliminated twice");
                seen[(size_t)v] = 1;
                vmap0[(size_t)(k0 + i)] = v;
            }
            for (int i = 0; i < nd.b; ++i) bd_vertex[(size_t)(nd.bdoff + i)] = h->front_idx[io2 + nd.n + i];
            k0 += nd.n;
        }
        for (int p = 0; p < nn; ++p)       // every boundary row of a child must be pulled exactly once by its parent
            for (int k = 0; k < 2; ++k) {
                const int ch = h->node_child[2 * p + k];
                if (ch < 0) continue;
                const int32_t *pull = k == 0 ? h->pull0 : h->pull1;
                std::vector<char> got((size_t)h->node_b[ch], 0);
                for (int fpos = 0; fpos < h->node_n[p] + h->node_b[p]; ++fpos) {
                    const int r = pull[h->node_ioff[p] + fpos];
                    if (r < 0) continue;
                    if (got[(size_t)r]) return bad("a child boundary row is pulled twice");
                    got[(size_t)r] = 1;
                }
                for (int r = 0; r < h->node_b[ch]; ++r)
                    if (!got[(size_t)r]) return bad("a child boundary row is not pulled by its parent");
            }
    }

    // ---- the nodes of the sweeps: the members of a band that hang together ----------------------------
    std::vector<Group> groups;
    std::vector<int> band_of_level, root_of((size_t)nn), gidx((size_t)nn, -1), off_in((size_t)nn, 0), c0_in((size_t)nn, 0);
    std::vector<int> vmap, cmap, band_planes;
    std::vector<MergeMember> members;
    std::vector<int> member_of((size_t)nn, -1);
    int64_t merged_entries = 0, scratch_entries = 0;
    double entries_read = 0.0;
    for (int attempt = 0;; ++attempt) {
        const int nb = (int)cuts.size() - 1;
        band_of_level.assign((size_t)h->n_levels, 0);
        for (int k = 0; k < nb; ++k)
            for (int l = cuts[k]; l < cuts[k + 1]; ++l) band_of_level[(size_t)l] = k;
        auto band = [&](int p) { return band_of_level[(size_t)level_of[p]]; };
        for (int p = nn - 1; p >= 0; --p) root_of[(size_t)p] = (parent[p] >= 0 && band(parent[p]) == band(p)) ? root_of[(size_t)parent[p]] : p;
        groups.clear();
        std::fill(gidx.begin(), gidx.end(), -1);
        for (int p = 0; p < nn; ++p)
            if (root_of[(size_t)p] == p) {
                gidx[(size_t)p] = (int)groups.size();
                Group G;
                G.root = p;
                G.band = band(p);
                G.b = h->node_b[p];
                groups.push_back(G);
            }
        for (int p = 0; p < nn; ++p) groups[(size_t)gidx[(size_t)root_of[(size_t)p]]].members.push_back(p);
        merged_entries = scratch_entries = 0;
        entries_read = 0.0;
        int k0 = 0;
        members.clear();
        std::fill(member_of.begin(), member_of.end(), -1);
        for (Group &G : groups) {
            for (int s : G.members) {
                off_in[(size_t)s] = G.n;
                int c0 = G.n;
                for (int k = 0; k < 2; ++k) {
                    const int ch = h->node_child[2 * s + k];
                    if (ch >= 0 && root_of[(size_t)ch] == G.root) c0 = std::min(c0, c0_in[(size_t)ch]);
                }
                c0_in[(size_t)s] = c0;
                const double ns = h->node_n[s];
                entries_read += 0.5 * ns * (ns + 1) + ns * (double)(G.n - c0);
                G.n += h->node_n[s];
            }
            entries_read += (double)G.b * G.n;
            if (top_inv && G.band == nb - 1) {     // S^-1: n x n entries read ONCE per solve = n * n / 2 per sweep in this count
                double tri = 0.0;
                for (int s : G.members) tri += 0.5 * h->node_n[s] * (h->node_n[s] + 1.0) + (double)h->node_n[s] * (off_in[(size_t)s] - c0_in[(size_t)s]);
                entries_read += 0.5 * (double)G.n * G.n - tri;
            }
            G.k0 = k0;
            k0 += G.n;
            if (G.members.size() == 1) {
                G.foff = h->node_foff[G.root];
            } else {
                G.foff = h->n_entries + merged_entries;
                merged_entries += (int64_t)(G.n + G.b) * G.n;
                for (int s : G.members) {          // member records of the merge kernel
                    MergeMember mm{};
                    mm.n = h->node_n[s];
                    mm.b = h->node_b[s];
                    mm.o = off_in[(size_t)s];
                    mm.c0 = c0_in[(size_t)s];
                    mm.foff = h->node_foff[s];
                    mm.ioff = h->node_ioff[s];
                    mm.dst = G.foff;
                    mm.ns = G.n;
                    mm.ch[0] = mm.ch[1] = -1;
                    bool inner = false;
                    for (int k = 0; k < 2; ++k) {
                        const int ch = h->node_child[2 * s + k];
                        if (ch >= 0 && root_of[(size_t)ch] == G.root) { mm.ch[k] = member_of[(size_t)ch]; inner = true; }
                    }
                    if (s == G.root) {
                        mm.uin = 2;
                        mm.uoff = G.foff + (int64_t)G.n * G.n;
                        mm.ustride = G.n;
                    } else if (!inner) {
                        mm.uin = 0;
                        mm.uoff = mm.foff + (int64_t)mm.n * mm.n;
                        mm.ustride = mm.n;
                    } else {
                        mm.uin = 1;
                        mm.uoff = scratch_entries;
                        mm.ustride = mm.o + mm.n - mm.c0;
                        scratch_entries += (int64_t)mm.b * mm.ustride;
                    }
                    member_of[(size_t)s] = (int)members.size();
                    members.push_back(mm);
                }
            }
        }
        // sweep order, positions of the update rows in the parents' fronts, planes
        vmap.assign((size_t)d.V, 0);
        cmap.assign((size_t)std::max<int64_t>(h->update_rows, 1), 0);
        std::vector<int> rowpos((size_t)d.V, -1);
        band_planes.assign((size_t)nb, 0);
        int over = -1;
        for (size_t gi = 0; gi < groups.size(); ++gi) {
            Group &G = groups[gi];
            for (int s : G.members)
                for (int i = 0; i < h->node_n[s]; ++i) {
                    const int v = h->front_idx[h->node_ioff[s] + i];
                    vmap[(size_t)(G.k0 + off_in[(size_t)s] + i)] = v;
                    rowpos[(size_t)v] = off_in[(size_t)s] + i;
                }
            const int64_t iop = h->node_ioff[G.root] + h->node_n[G.root];
            for (int i = 0; i < G.b; ++i) rowpos[(size_t)h->front_idx[iop + i]] = G.n + i;
            // nodes below the band that hang from this one: where their update rows land, and a plane for each such
            // that no two writers of a plane share a row -- the fewest planes (the forward kernel reads every plane of
            // every column): exact colouring of the conflict graph (at most 16 nodes)
            std::vector<int> ext;                          // the children's group indices
            for (int s : G.members)
                for (int k = 0; k < 2; ++k) {
                    const int ch = h->node_child[2 * s + k];
                    if (ch < 0 || root_of[(size_t)ch] == G.root) continue;
                    const int32_t *pull = k == 0 ? h->pull0 : h->pull1;
                    const int64_t ios = h->node_ioff[s], uoc = h->node_uoff[ch];
                    for (int fpos = 0; fpos < h->node_n[s] + h->node_b[s]; ++fpos) {
                        const int r = pull[ios + fpos];
                        if (r >= 0) cmap[(size_t)(uoc + r)] = rowpos[(size_t)h->front_idx[ios + fpos]];
                    }
                    groups[(size_t)gidx[(size_t)ch]].parent = (int)gi;
                    ext.push_back(gidx[(size_t)ch]);
                }
            const int ne = (int)ext.size();
            std::vector<uint32_t> clash((size_t)ne, 0);    // bit j: child i and child j write a common row
            {
                std::vector<uint32_t> who((size_t)(G.n + G.b), 0);
                for (int i = 0; i < ne && i < 32; ++i) {
                    const int ch = groups[(size_t)ext[(size_t)i]].root;
                    for (int r = 0; r < h->node_b[ch]; ++r) who[(size_t)cmap[(size_t)(h->node_uoff[ch] + r)]] |= 1u << i;
                }
                for (uint32_t m2 : who)
                    for (int i = 0; i < ne && i < 32; ++i)
                        if (m2 >> i & 1u) clash[(size_t)i] |= m2 & ~(1u << i);
            }
            std::vector<int> colour((size_t)ne, -1);
            int planes = 0;
            if (ne > 0 && ne <= 20) {
                std::vector<int> order((size_t)ne);
                for (int i = 0; i < ne; ++i) order[(size_t)i] = i;
                std::sort(order.begin(), order.end(), [&](int x, int y) {
                    const int dx = __builtin_popcount(clash[(size_t)x]), dy = __builtin_popcount(clash[(size_t)y]);
                    return dx != dy ? dx > dy : x < y;
                });
                for (planes = 1; planes <= ne; ++planes) {
                    std::fill(colour.begin(), colour.end(), -1);
                    int64_t budget = 200000;      // backtracking steps; beyond: take the next larger count
                    std::function<bool(int)> place = [&](int at) -> bool {
                        if (at == ne) return true;
                        if (--budget < 0) return false;
                        const int v = order[(size_t)at];
                        for (int col = 0; col < planes && col <= at; ++col) {     // col <= at: planes are interchangeable
                            bool free = true;
                            for (int j = 0; j < ne && free; ++j)
                                free = !((clash[(size_t)v] >> j & 1u) && colour[(size_t)j] == col);
                            if (!free) continue;
                            colour[(size_t)v] = col;
                            if (place(at + 1)) return true;
                            colour[(size_t)v] = -1;
                        }
                        return false;
                    };
                    if (place(0)) break;
                }
            } else {                                       // (not reachable with bands of at most 4 heights) one plane each
                for (int i = 0; i < ne; ++i) colour[(size_t)i] = i;
                planes = ne;
            }
            for (int i = 0; i < ne; ++i) groups[(size_t)ext[(size_t)i]].colour = colour[(size_t)i];
            G.planes = planes;
            band_planes[(size_t)G.band] = std::max(band_planes[(size_t)G.band], G.planes);
            if (G.planes > 8 && over < 0) over = G.band;
            for (int s : G.members)
                for (int i = 0; i < h->node_n[s]; ++i) rowpos[(size_t)h->front_idx[h->node_ioff[s] + i]] = -1;
            for (int i = 0; i < G.b; ++i) rowpos[(size_t)h->front_idx[iop + i]] = -1;
        }
        if (over < 0) break;
        // more than 8 planes on a merged node (cannot happen with one height per band: two children): split that band
        if (attempt > 64 || cuts[(size_t)over + 1] - cuts[(size_t)over] < 2) return bad("update planes");
        std::vector<int> split;
        for (size_t k = 0; k < cuts.size(); ++k) {
            split.push_back(cuts[k]);
            if ((int)k == over)
                for (int l = cuts[k] + 1; l < cuts[k + 1]; ++l) split.push_back(l);
        }
        cuts.swap(split);
    }
    const int nb = (int)cuts.size() - 1;
    bool identity = true;
    for (int k = 0; k < d.V && identity; ++k) identity = vmap[(size_t)k] == k;
    bool identity0 = true;
    for (int k = 0; k < d.V && identity0; ++k) identity0 = vmap0[(size_t)k] == k;
    // The leaves as explicit local inverses (k_front_leaf_fwd / _bwd): an un-merged band of leaves below at least one other band; every leaf's
    // vertices numbered as the sweeps walk them (device vertex = sweep-order index: the plan's own numbering, or any that keeps the leaves in place);
    // a row of modes within a workgroup.  DOTS_FRONT_CFG / DOTS_FRONT_TUNE choose among the BAND kernels, also for band 0: the leaves then stay with them.
    bool leaf_inv = c->front_leafinv && !getenv("DOTS_FRONT_CFG") && !c->front_tune && nb >= 2 && cuts[1] == 1 && d.rowptr && d.col && d.val &&
                    (d.TP / (front_two_modes(c) ? 2 : 1)) <= 1024;
    for (size_t gi = 0; gi < groups.size() && leaf_inv; ++gi) {
        const Group &G = groups[gi];
        if (G.band != 0) continue;
        if (G.n == 0 && G.b > 0) leaf_inv = false;      // (cannot happen: a leaf's boundary comes from its own vertices)
        for (int j = 0; j < G.n && leaf_inv; ++j) leaf_inv = vmap[(size_t)(G.k0 + j)] == G.k0 + j;
    }
    // planes: the band's bucket (0, 2, 4, 8) of m rows each per node; W starts with a few zero rows (woff = 0 is never read)
    int64_t wrows = 8;
    for (int k = 0; k < nb; ++k) {
        int &bp = band_planes[(size_t)k];
        bp = bp == 0 ? 0 : (bp <= 2 ? 2 : (bp <= 4 ? 4 : 8));
        c->front_planes[k] = bp;
    }
    for (Group &G : groups) {
        G.woff = wrows;
        wrows += (int64_t)band_planes[(size_t)G.band] * (G.n + G.b);
    }

    // ---- workgroup lists per band: (node, first row) for the forward sweep, (node, first column, row ranges)
    // backward
    std::vector<std::vector<int>> by_band((size_t)nb);
    for (size_t gi = 0; gi < groups.size(); ++gi) by_band[(size_t)groups[gi].band].push_back((int)gi);
    auto sweep_node = [&](const Group &G) {
        SweepNode sn{};
        sn.n = G.n;
        sn.b = G.b;
        sn.k0 = G.k0;
        sn.planes = G.planes;
        sn.foff = G.foff;
        sn.woff = G.woff;
        sn.parent_w = G.parent < 0 ? -1 : groups[(size_t)G.parent].woff + (int64_t)G.colour * (groups[(size_t)G.parent].n + groups[(size_t)G.parent].b);
        sn.bdoff = h->node_uoff[G.root];
        return sn;
    };
    auto make_fwd = [&](int k, int rb, std::vector<FrontWork> &out) {
        for (int gi : by_band[(size_t)k]) {
            const Group &G = groups[(size_t)gi];
            const SweepNode sn = sweep_node(G);
            std::vector<int> first_col((size_t)G.n, 0);     // first column of its member's subtree, per separator row
            for (int s : G.members)
                for (int i = 0; i < h->node_n[s]; ++i) first_col[(size_t)(off_in[(size_t)s] + i)] = c0_in[(size_t)s];
            for (int r = 0; r < G.n + G.b; r += rb) {
                FrontWork w{};
                w.nd = sn;
                w.first = r;
                int lo = r < G.n ? first_col[(size_t)r] : 0;
                for (int i = r; i < std::min(r + rb, G.n + G.b); ++i) lo = std::min(lo, i < G.n ? first_col[(size_t)i] : 0);
                w.lo = lo;
                if (top_inv && k == nb - 1) { w.lo = 0; w.end = G.n; }     // full rows of S^-1
                out.push_back(w);
            }
        }
    };
    // row kernel (k_front_fwd_rows): blocks of up to rows_wg rows that never span two members of a merged node (so that wk.lo is
    // the first column of every row of the block) nor the separator / boundary rows; returns the longest column range a block stages
    auto make_fwd_rows = [&](int k, int rows_wg, std::vector<FrontWork> &out) {
        int lds_cols = 1;
        for (int gi : by_band[(size_t)k]) {
            const Group &G = groups[(size_t)gi];
            const SweepNode sn = sweep_node(G);
            const bool full = top_inv && k == nb - 1;
            std::vector<std::array<int, 3>> segs;      // (first row, one past the last, first column)
            if (full) segs.push_back({0, G.n, 0});
            else {
                for (int s : G.members)
                    if (h->node_n[s] > 0) segs.push_back({off_in[(size_t)s], off_in[(size_t)s] + h->node_n[s], c0_in[(size_t)s]});
                if (G.b > 0) segs.push_back({G.n, G.n + G.b, 0});
            }
            for (const auto &sg : segs)
                for (int r = sg[0]; r < sg[1]; r += rows_wg) {
                    FrontWork w{};
                    w.nd = sn;
                    w.first = r;
                    w.pad = std::min(rows_wg, sg[1] - r);
                    w.lo = sg[2];
                    const int last = r + w.pad - 1;
                    int jmax = last < G.n ? last + 1 : G.n;
                    if (full) { w.end = G.n; jmax = G.n; }
                    lds_cols = std::max(lds_cols, jmax - w.lo);
                    out.push_back(w);
                }
        }
        return lds_cols;
    };
    auto make_bwd = [&](int k, int cb, std::vector<FrontWork> &out) {
        for (int gi : by_band[(size_t)k]) {
            const Group &G = groups[(size_t)gi];
            const SweepNode sn = sweep_node(G);
            for (int s : G.members) {
                const int o = off_in[(size_t)s], e = o + h->node_n[s];
                for (int col = o; col < e; col += cb) {
                    FrontWork w{};
                    w.nd = sn;
                    w.first = col;
                    w.end = e;
                    int nr = 0;
                    w.rs[nr] = col;
                    w.re[nr] = e;
                    ++nr;
                    for (int a2 = parent[s]; a2 >= 0 && root_of[(size_t)a2] == G.root; a2 = parent[a2]) {
                        const int ao = off_in[(size_t)a2], ae = ao + h->node_n[a2];
                        if (ae == ao) continue;
                        if (w.re[nr - 1] == ao) w.re[nr - 1] = ae;          // adjacent: one range
                        else if (nr < 4) {
                            w.rs[nr] = ao;
                            w.re[nr] = ae;
                            ++nr;
                        }
                    }
                    w.lo = nr;
                    out.push_back(w);
                }
            }
        }
    };
    // Threads and rows (columns) per workgroup of each band (measured per band with DOTS_FRONT_TUNE, profiles/studies/
    // band_cuts.txt).  Many rows: 256-thread workgroups of up to 4 rows; few rows (the large nodes near the root): 1024-thread
    // workgroups, so that the long dot products are split 4x finer.  Forward bands whose nodes read 4 or 8 update planes take
    // 4 rows per workgroup earlier (the planes are read once per workgroup, not per row).
    std::vector<int64_t> band_rows((size_t)nb, 0), band_cols((size_t)nb, 0);
    const bool two_modes = front_two_modes(c);
    // lane groups of a wavefront in the row kernel: 64 / (lanes per row of modes); 0 = a row of modes is wider than a wavefront
    const int lanes_row = two_modes ? d.TP / 2 : d.TP;
    const int rows_groups = lanes_row <= 64 ? 64 / lanes_row : 0;
    auto rows_per_wg = [&](int qs) { return std::max(1, ((256 / lanes_row) >> qs)); };
    for (int k = 0; k < nb; ++k) {
        for (int gi : by_band[(size_t)k]) {
            band_rows[(size_t)k] += groups[(size_t)gi].n + groups[(size_t)gi].b;
            band_cols[(size_t)k] += groups[(size_t)gi].n;
        }
        const int64_t rows = band_rows[(size_t)k], cols = band_cols[(size_t)k];
        const bool big_ok = d.TP <= 256;      // 1024-thread workgroups need a row of modes to fit
        int fnb, frb, bnb, bcb;
        if (top_inv && k == nb - 1 && big_ok) {      // full rows of S^-1: every workgroup reads the whole right-hand side and all planes
            fnb = 1024;
            frb = rows >= 480 ? 4 : (rows >= 64 ? 2 : 1);
        } else if (band_planes[(size_t)k] >= 4) {
            if (rows >= 600 || !big_ok) { fnb = 256; frb = 4; }
            else { fnb = 1024; frb = rows >= 300 ? 2 : 1; }
        } else {      // (thresholds from the tables of profiles/studies/shape_tuner.txt)
            fnb = (rows < 1024 && big_ok) ? 1024 : 256;
            frb = fnb == 1024 ? (rows >= 512 ? 2 : 1) : (rows >= 2048 ? 4 : 2);
        }
        // (two-mode lanes split a dot product over twice as many parts per workgroup: 256 threads reach further down; at a pitch of
        // 128 a 256-thread workgroup splits a dot product only 4 ways: 1024 threads up to 3000 columns)
        if (cols >= (d.TP >= 128 ? 3000 : (two_modes ? 1024 : 1536)) || !big_ok) { bnb = 256; bcb = cols >= 4096 ? 4 : (cols >= 2048 ? 2 : 1); }
        else { bnb = 1024; bcb = (d.TP >= 128 && cols >= 600) ? 4 : (cols >= 250 ? 2 : 1); }
        c->front_fwd_rb[k] = std::min(frb, c->front_rb_max);
        c->front_bwd_cb[k] = std::min(bcb, c->front_rb_max);
        c->front_fwd_nb[k] = fnb;
        c->front_bwd_nb[k] = bnb;
        // Bands of short rows take the row kernel (k_front_fwd_rows): QW lane groups per row by the band's mean row length
        // (rules from the DOTS_FRONT_TUNE tables of profiles/studies/shape_tuner.txt, round 3)
        c->front_fwd_qw[k] = -1;
        c->front_fwd_lds[k] = 0;
        if (c->front_rows && rows_groups >= 1) {
            double len = 0.0;      // columns a row of the band reads, summed
            int longest = 1;
            for (int gi : by_band[(size_t)k]) {
                const Group &G = groups[(size_t)gi];
                if (top_inv && k == nb - 1) { len += (double)G.n * G.n; longest = std::max(longest, G.n); continue; }
                for (int s : G.members) {
                    const double ns = h->node_n[s], w0 = off_in[(size_t)s] - c0_in[(size_t)s];
                    len += ns * w0 + 0.5 * ns * (ns + 1);
                    longest = std::max(longest, off_in[(size_t)s] + h->node_n[s] - c0_in[(size_t)s]);
                }
                len += (double)G.b * G.n;
                if (G.b > 0) longest = std::max(longest, G.n);
            }
            const double mean = rows > 0 ? len / (double)rows : 0.0;
            const bool fits = (size_t)longest * (size_t)(d.TP + FWD_ROWS_PAD) * sizeof(double) <= FWD_ROWS_LDS_MAX;
            // (round-3 tables: with rows of <= ~30 columns one lane group per row wins by 10-60 %; on merged bands of 4+ planes the
            // sixteen rows of a workgroup share ONE staged right-hand side: four groups per row draw level or win; elsewhere
            // the fold kernels keep the longer rows)
            int qs = mean <= FWD_ROWS_MEAN_MAX ? 0 : 2;
            while ((1 << qs) > rows_groups) --qs;
            if (fits && (c->front_rows >= 2 || mean <= FWD_ROWS_MEAN_MAX || (band_planes[(size_t)k] >= 4 && rows_groups >= 4))) c->front_fwd_qw[k] = qs;
        }
    }

    // ---- device: the original tree, the factor, the merged blocks ----------------------------------------
    FrontDev f0{};      // the original tree as the factorisation sees it
    f0.n_nodes = nn;
    f0.n_levels = h->n_levels;
    int rc;
#define FUP(dev, field, src, n) if ((rc = front_upload(c, &dev.field, src, (int64_t)(n)))) { front_release(c); return rc; }
    FUP(f0, nodes, nodes.data(), nn);
    if (!identity0) FUP(f0, vmap, vmap0.data(), d.V);
    FUP(f0, bd_vertex, bd_vertex.data(), bd_vertex.size());
    const int64_t all_entries = h->n_entries + merged_entries;
    {   // Does it fit?  The factor (with the merged blocks) stays; the numeric factorisation needs a copy of the fronts and the
        // Schur complements beside it; the iteration's carried gathers are allocated behind it (dots_front_setup).  The
        // reference just factorises (laplacian_inverse_socp.py:34-41); here the caller gets a status it can act on
        // (the Python driver falls back to the multigrid-PCG) instead of a failed allocation halfway through.
        int64_t srows = 0;
        for (const FrontNode &nd : nodes) srows += (int64_t)nd.b * nd.b;
        const double per = 8.0 * (double)d.TP;
        double leaf_entries = 0.0;      // the leaves' explicit inverses, stored beside their blocks (see below)
        if (leaf_inv)
            for (int gi : by_band[0]) leaf_entries += 0.5 * (double)groups[(size_t)gi].n * (groups[(size_t)gi].n + 1.0);
        const double factor_b = per * ((double)all_entries + leaf_entries), work_b = h->values ? 0.0 : per * (double)(h->n_entries + srows);
        const double carry_b = (c->d.TP <= 128 && c->carry_arrays) ? 8.0 * (c->shard_stride == 0 ? 12.0 : 9.0) * (double)c->d.F * (double)c->d.TP : 0.0;
        size_t free_b = 0, total_b = 0;
        DOTS_HIP(hipMemGetInfo(&free_b, &total_b));
        double budget = 0.97 * (double)free_b;
        int mb = -1;
        if (!env_int("DOTS_MEM_BUDGET", 0, 1 << 30, &mb)) return DOTS_ERR_ARGUMENT;      // MB the factor may take, whatever is free (tests)
        if (mb >= 0) budget = 1048576.0 * mb;
        if (factor_b + work_b + carry_b > budget) {
            char buf[512];
            snprintf(buf, sizeof buf, "front_setup: the factor does not fit: %.3f GB (factor %.3f GB for %d modes of %d vertices, %.3f GB while it is "
                     "computed, %.3f GB of per-corner sums) against %.3f GB available", (factor_b + work_b + carry_b) * 1e-9, factor_b * 1e-9, h->n_modes, d.V,
                     work_b * 1e-9, carry_b * 1e-9, budget * 1e-9);
            front_release(c);
            set_error(buf);
            return DOTS_ERR_MEMORY;
        }
    }
    const double *Fall = nullptr;
    if ((rc = front_upload<double>(c, &Fall, nullptr, all_entries << d.tp_shift))) { front_release(c); return rc; }
    if (h->values) {
        hipError_t e = hipMemcpyAsync(const_cast<double *>(Fall), h->values, sizeof(double) * ((size_t)h->n_entries << d.tp_shift), hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { front_release(c); return hip_fail(e, "factor upload", __FILE__, __LINE__); }
    } else {   // numeric factorisation on the device (kernels_factor.hip)
        std::vector<int> grounded((size_t)d.TP, 0);
        for (int a = 0; a < h->n_modes; ++a) grounded[(size_t)a] = h->grounded[a] ? 1 : 0;
        if ((rc = front_factorize(c, h, f0, nodes, const_cast<double *>(Fall), grounded.data()))) { front_release(c); return rc; }
    }
    if (!members.empty()) {      // merged bands: one launch per tree height inside a band, children first
        std::vector<void *> tmp;
        auto release_tmp = [&]() { for (void *p : tmp) (void)hipFree(p); };
        auto dalloc = [&](void **out, size_t bytes, const void *host) -> hipError_t {
            void *p = nullptr;
            hipError_t e = hipMalloc(&p, std::max<size_t>(bytes, 8));
            if (e != hipSuccess) return e;
            tmp.push_back(p);
            *out = p;
            return host ? hipMemcpyAsync(p, host, bytes, hipMemcpyHostToDevice, c->stream) : hipSuccess;
        };
        std::vector<int> order;                      // member records sorted by (band, tree height)
        std::vector<int> launch_ptr{0};
        std::vector<int64_t> launch_items;
        std::vector<int> node_of_member(members.size());
        for (int p = 0; p < nn; ++p)
            if (member_of[(size_t)p] >= 0) node_of_member[(size_t)member_of[(size_t)p]] = p;
        for (int l = 0; l < h->n_levels; ++l) {
            int64_t items = 0;
            for (size_t mi = 0; mi < members.size(); ++mi)
                if (level_of[node_of_member[mi]] == l) {
                    order.push_back((int)mi);
                    const MergeMember &mm = members[mi];
                    items = std::max(items, (int64_t)(mm.n + mm.b) * (mm.o + mm.n - mm.c0));
                }
            if ((int)order.size() > launch_ptr.back()) {
                launch_ptr.push_back((int)order.size());
                launch_items.push_back(items);
            }
        }
        void *dm = nullptr, *dl = nullptr, *p0 = nullptr, *p1 = nullptr, *sc = nullptr;
        hipError_t e = dalloc(&dm, sizeof(MergeMember) * members.size(), members.data());
        if (e == hipSuccess) e = dalloc(&dl, sizeof(int) * order.size(), order.data());
        if (e == hipSuccess) e = dalloc(&p0, sizeof(int) * (size_t)h->n_front_rows, h->pull0);
        if (e == hipSuccess) e = dalloc(&p1, sizeof(int) * (size_t)h->n_front_rows, h->pull1);
        if (e == hipSuccess) e = dalloc(&sc, sizeof(double) * ((size_t)std::max<int64_t>(scratch_entries, 1) << d.tp_shift), nullptr);
        if (e == hipSuccess) {
            MergeArgs g{};
            g.sh = d.tp_shift; g.TP = d.TP; g.ncol = d.cg_ncol;
            g.F = const_cast<double *>(Fall);
            g.scratch = (double *)sc;
            g.pull0 = (const int *)p0; g.pull1 = (const int *)p1;
            g.mem = (const MergeMember *)dm;
            const int Q = 256 >> d.tp_shift;
            for (size_t k = 0; k + 1 < launch_ptr.size(); ++k) {
                g.list = (const int *)dl + launch_ptr[k];
                const unsigned bx = (unsigned)std::min<int64_t>(std::max<int64_t>((launch_items[k] + Q - 1) / Q, 1), 2048);
                hipLaunchKernelGGL(k_merge_member, dim3(bx, (unsigned)(launch_ptr[k + 1] - launch_ptr[k])), dim3(256), 0, c->stream, g);
            }
            e = hipGetLastError();
        }
        hipError_t e2 = hipStreamSynchronize(c->stream);
        release_tmp();
        if (e != hipSuccess || e2 != hipSuccess) { front_release(c); return hip_fail(e != hipSuccess ? e : e2, "merged bands", __FILE__, __LINE__); }
    }

    if (top_inv) {      // the top band's blocks L'^-1 (n x n, no boundary rows) become S^-1 = L'^-T L'^-1, in place
        std::vector<int64_t> foffs, ooffs;
        std::vector<int> ns;
        int64_t total = 0, biggest = 0;
        for (int gi : by_band[(size_t)(nb - 1)]) {
            const Group &G = groups[(size_t)gi];
            if (G.b != 0) { front_release(c); return bad("top_inverse: a node of the top band has boundary rows"); }
            if (G.n == 0) continue;
            foffs.push_back(G.foff);
            ooffs.push_back(total);
            ns.push_back(G.n);
            total += (int64_t)G.n * G.n;
            biggest = std::max<int64_t>(biggest, (int64_t)G.n * G.n);
        }
        if (!ns.empty()) {
            void *dS = nullptr, *dfo = nullptr, *doo = nullptr, *dn = nullptr;
            hipError_t e = hipMalloc(&dS, sizeof(double) * ((size_t)total << d.tp_shift));
            if (e == hipSuccess) e = hipMalloc(&dfo, sizeof(int64_t) * foffs.size());
            if (e == hipSuccess) e = hipMalloc(&doo, sizeof(int64_t) * ooffs.size());
            if (e == hipSuccess) e = hipMalloc(&dn, sizeof(int) * ns.size());
            if (e == hipSuccess) e = hipMemcpyAsync(dfo, foffs.data(), sizeof(int64_t) * foffs.size(), hipMemcpyHostToDevice, c->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(doo, ooffs.data(), sizeof(int64_t) * ooffs.size(), hipMemcpyHostToDevice, c->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(dn, ns.data(), sizeof(int) * ns.size(), hipMemcpyHostToDevice, c->stream);
            if (e == hipSuccess) {
                TopInvArgs g{};
                g.sh = d.tp_shift; g.TP = d.TP; g.ncol = d.cg_ncol;
                g.F = Fall; g.out = (double *)dS;
                g.foff = (const int64_t *)dfo; g.ooff = (const int64_t *)doo; g.n = (const int *)dn;
                const int Q = 256 >> d.tp_shift;
                const unsigned bx = (unsigned)std::min<int64_t>(std::max<int64_t>((biggest + Q - 1) / Q, 1), 4096);
                hipLaunchKernelGGL(k_top_inverse, dim3(bx, (unsigned)ns.size()), dim3(256), 0, c->stream, g);
                e = hipGetLastError();
                for (size_t k = 0; k < ns.size() && e == hipSuccess; ++k)
                    e = hipMemcpyAsync(const_cast<double *>(Fall) + (foffs[k] << d.tp_shift), (const double *)dS + (ooffs[k] << d.tp_shift),
                                       sizeof(double) * ((size_t)ns[k] * ns[k] << d.tp_shift), hipMemcpyDeviceToDevice, c->stream);
            }
            hipError_t e2 = hipStreamSynchronize(c->stream);
            for (void *p2 : {dS, dfo, doo, dn}) if (p2) (void)hipFree(p2);
            if (e != hipSuccess || e2 != hipSuccess) { front_release(c); return hip_fail(e != hipSuccess ? e : e2, "top inverse", __FILE__, __LINE__); }
        }
    }

    FrontDev f{};
    f.n_nodes = nn;
    f.n_levels = nb;
    f.nodes = f0.nodes;
    f.bd_vertex = f0.bd_vertex;
    f.F = Fall;
    if (!identity) FUP(f, vmap, vmap.data(), d.V);
    FUP(f, cmap, cmap.data(), cmap.size());
    const double *w = nullptr;
    if ((rc = front_upload<double>(c, &w, nullptr, std::max<int64_t>(wrows, 1) << d.tp_shift))) { front_release(c); return rc; }
    f.W = const_cast<double *>(w);
    // ---- the leaves as explicit local inverses (leaf_inv above; w and t of the largest leaf must fit the LDS a workgroup may take)
    if (leaf_inv) {
        std::vector<LeafWork> leaves;
        std::vector<int64_t> foffs, ooffs;
        std::vector<int> ns;
        int64_t total = 0, biggest = 1, bd_rows = 0;
        int nmax = 0;
        double saved_read = 0.0, saved_alg = 0.0;
        for (int gi : by_band[0]) {
            const Group &G = groups[(size_t)gi];
            if (G.n == 0) continue;      // (nothing to eliminate: nothing to send either -- its plane stays zero)
            LeafWork lw{};
            lw.k0 = G.k0; lw.n = G.n; lw.b = G.b;
            lw.soff = total;
            lw.bdoff = h->node_uoff[G.root];
            lw.rowoff = bd_rows;
            lw.parent_w = G.parent < 0 ? 0 : groups[(size_t)G.parent].woff + (int64_t)G.colour * (groups[(size_t)G.parent].n + groups[(size_t)G.parent].b);
            if (G.parent < 0) lw.b = 0;
            bd_rows += lw.b;
            leaves.push_back(lw);
            foffs.push_back(G.foff);
            ooffs.push_back(total);
            ns.push_back(G.n);
            total += (int64_t)G.n * (G.n + 1) / 2;      // S is symmetric: its lower triangle, packed by rows
            biggest = std::max<int64_t>(biggest, (int64_t)G.n * G.n);
            nmax = std::max(nmax, G.n);
            saved_read += (double)G.b * G.n;
            saved_alg += (double)G.b * G.n;
        }
        const size_t lds = sizeof(double) * 2 * (size_t)nmax * (size_t)d.TP;
        if (!leaves.empty() && lds <= 48 * 1024) {
            const double *dS = nullptr;
            const LeafWork *dl = nullptr;
            if ((rc = front_upload<double>(c, &dS, nullptr, total << d.tp_shift)) || (rc = front_upload(c, &dl, leaves.data(), (int64_t)leaves.size()))) { front_release(c); return rc; }
            void *dfo = nullptr, *doo = nullptr, *dn = nullptr;
            hipError_t e = hipMalloc(&dfo, sizeof(int64_t) * foffs.size());
            if (e == hipSuccess) e = hipMalloc(&doo, sizeof(int64_t) * ooffs.size());
            if (e == hipSuccess) e = hipMalloc(&dn, sizeof(int) * ns.size());
            if (e == hipSuccess) e = hipMemcpyAsync(dfo, foffs.data(), sizeof(int64_t) * foffs.size(), hipMemcpyHostToDevice, c->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(doo, ooffs.data(), sizeof(int64_t) * ooffs.size(), hipMemcpyHostToDevice, c->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(dn, ns.data(), sizeof(int) * ns.size(), hipMemcpyHostToDevice, c->stream);
            if (e == hipSuccess) {
                const int Q = 256 >> d.tp_shift;
                const unsigned bx = (unsigned)std::min<int64_t>(std::max<int64_t>((biggest + std::max(Q, 1) - 1) / std::max(Q, 1), 1), 4096);
                for (size_t at = 0; at < ns.size() && e == hipSuccess; at += 32768) {      // (grid.y is limited to 65535)
                    TopInvArgs g{};
                    g.sh = d.tp_shift; g.TP = d.TP; g.ncol = d.cg_ncol;
                    g.F = Fall; g.out = const_cast<double *>(dS);
                    g.foff = (const int64_t *)dfo + at; g.ooff = (const int64_t *)doo + at; g.n = (const int *)dn + at;
                    g.packed = 1;
                    hipLaunchKernelGGL(k_top_inverse, dim3(bx, (unsigned)std::min<size_t>(ns.size() - at, 32768)), dim3(256), 0, c->stream, g);
                    e = hipGetLastError();
                }
            }
            hipError_t e2 = hipStreamSynchronize(c->stream);
            for (void *p2 : {dfo, doo, dn}) if (p2) (void)hipFree(p2);
            if (e != hipSuccess || e2 != hipSuccess) { front_release(c); return hip_fail(e != hipSuccess ? e : e2, "leaf inverses", __FILE__, __LINE__); }
            f.leafS = dS;
            f.leaf_desc = dl;
            f.n_leaves = (int)leaves.size();
            f.leaf_nmax = nmax;
            if (c->front_leafinv == 1) {      // coupling records (a row with more entries than a record holds: the kernels walk the CSR instead)
                const LeafBdRow *dbt = nullptr;
                const LeafSepRow *dst = nullptr;
                const int *dov = nullptr;
                if ((rc = front_upload<LeafBdRow>(c, &dbt, nullptr, std::max<int64_t>(bd_rows, 1))) || (rc = front_upload<LeafSepRow>(c, &dst, nullptr, d.V)) ||
                    (rc = front_upload<int>(c, &dov, nullptr, 1))) { front_release(c); return rc; }
                hipLaunchKernelGGL(k_leaf_tables, dim3(f.n_leaves), dim3(64), 0, c->stream, f, d.rowptr, d.col, d.val, const_cast<LeafBdRow *>(dbt),
                                   const_cast<LeafSepRow *>(dst), const_cast<int *>(dov));
                int over = 0;
                hipError_t e3 = hipGetLastError();
                if (e3 == hipSuccess) e3 = hipMemcpyAsync(&over, dov, sizeof(int), hipMemcpyDeviceToHost, c->stream);
                if (e3 == hipSuccess) e3 = hipStreamSynchronize(c->stream);
                if (e3 != hipSuccess) { front_release(c); return hip_fail(e3, "leaf coupling records", __FILE__, __LINE__); }
                if (!over) { f.leaf_bd = dbt; f.leaf_sep = dst; }
            }
            entries_read -= saved_read;          // n (n + 1) / 2 per leaf and sweep (the packed triangle of S) instead of n (n + 1) / 2 + b n;
            entries_unmerged -= saved_alg;       // the coupling is the mode-independent CSR
        }
    }
    c->front_bytes = 2.0 * entries_read * d.cg_ncol * sizeof(double);
    c->front_bytes_unmerged = 2.0 * entries_unmerged * d.cg_ncol * sizeof(double);
    if (const char *e = getenv("DOTS_FRONT_CFG")) {      // "fwd:1024x2,256x4,r1,...;bwd:..." one entry per band (A/B measurements); rQ = row kernel, Q lane groups per row
        const std::string spec(e);
        bool ok = spec.find("fwd:") != std::string::npos || spec.find("bwd:") != std::string::npos;
        for (int sweep = 0; sweep < 2 && ok; ++sweep) {
            size_t pos = spec.find(sweep == 0 ? "fwd:" : "bwd:");
            if (pos == std::string::npos) continue;
            pos += 4;
            for (int k = 0; k < nb && pos < spec.size() && spec[pos] != ';'; ++k) {
                int tnb = 0, trb = 0, q = 0;
                if (sweep == 0 && sscanf(spec.c_str() + pos, "r%d", &q) == 1 && q >= 1 && q <= rows_groups && (q & (q - 1)) == 0) {
                    int qs = 0;
                    while ((1 << qs) < q) ++qs;
                    c->front_fwd_qw[k] = qs;
                } else if (sscanf(spec.c_str() + pos, "%dx%d", &tnb, &trb) == 2 && (tnb == 256 || tnb == 1024) && (trb == 1 || trb == 2 || trb == 4)) {
                    (sweep == 0 ? c->front_fwd_nb : c->front_bwd_nb)[k] = tnb;
                    (sweep == 0 ? c->front_fwd_rb : c->front_bwd_cb)[k] = trb;
                    if (sweep == 0) c->front_fwd_qw[k] = -1;
                } else if (spec[pos] != '-') {      // "-" keeps the rule's choice for the band
                    ok = false;
                    break;
                }
                pos = spec.find_first_of(",;", pos);
                if (pos == std::string::npos || spec[pos] == ';') break;
                ++pos;
            }
        }
        if (!ok) { front_release(c); return bad("DOTS_FRONT_CFG: expected 'fwd:<entry>,...;bwd:<entry>,...' with entries 256xR, 1024xR (R = 1, 2, 4), rQ (forward: row kernel) or -"); }
    }
    // Workgroups are dealt round-robin over the 8 XCDs (blockIdx mod 8): every XCD gets one contiguous run of a launch's list, so
    // that the row blocks of a node, which read the same right-hand-side and plane rows, share an L2.  Measured (solve, us):
    // knot 49.3 -> 44.0, sphere10k 97.4 -> 89.5, knot63 80.0 -> 74.0; launches of few large nodes (>= 100 workgroups per node:
    // the top of torus100k) lose 4 % with it and keep the plain order; the large launches below them do not care.
    // DOTS_FRONT_XCD=0: plain order everywhere.
    const bool xcd_deal = c->front_xcd != 0;
    auto deal = [&](std::vector<FrontWork> &list, size_t from, size_t n_nodes) {
        const size_t n = list.size() - from;
        if (!xcd_deal || n_nodes < 8 || n < 16 || n > 80 * n_nodes) return;
        const size_t per = (n + 7) / 8;
        std::vector<FrontWork> out;
        out.reserve(n);
        for (size_t s2 = 0; s2 < per; ++s2)
            for (size_t x = 0; x < 8; ++x)
                if (x * per + s2 < n) out.push_back(list[from + x * per + s2]);
        std::copy(out.begin(), out.end(), list.begin() + (std::ptrdiff_t)from);
    };
    if (c->front_tune) {     // DOTS_FRONT_TUNE: time every (threads, rows) choice per band and sweep on this device; prints the table
        std::vector<void *> tmp;
        double *vec[3] = {nullptr, nullptr, nullptr};
        bool ok = true;
        for (int i = 0; i < 3 && ok; ++i) {
            void *p2 = nullptr;
            ok = hipMalloc(&p2, sizeof(double) * ((size_t)d.V << d.tp_shift)) == hipSuccess;
            if (ok) { tmp.push_back(p2); vec[i] = (double *)p2; ok = hipMemsetAsync(p2, 0, sizeof(double) * ((size_t)d.V << d.tp_shift), c->stream) == hipSuccess; }
        }
        hipEvent_t e0 = nullptr, e1 = nullptr;
        ok = ok && hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess;
        const int apply = c->front_tune;
        // A factor larger than the Infinity Cache streams from HBM in the real solve; a launch repeated back to back would find its
        // band (60-130 MB) in the cache.  There every timed launch is preceded by a read sweep over 512 MB (cold caches, no dirty lines, one event pair per
        // launch); small factors ARE cache-resident in the real solve and are timed back to back.
        const bool cold = c->front_bytes > 400.0e6;
        void *flushbuf = nullptr;
        const size_t flush_bytes = (size_t)512 << 20;
        if (cold && ok) {
            ok = hipMalloc(&flushbuf, flush_bytes) == hipSuccess;
            if (ok) { tmp.push_back(flushbuf); ok = hipMemsetAsync(flushbuf, 0, flush_bytes, c->stream) == hipSuccess; }
        }
        auto time_us = [&](const std::function<void()> &launch) -> double {
            if (!cold) {
                const int reps = 20;
                for (int rep = -3; rep < reps; ++rep) {
                    if (rep == 0) (void)hipEventRecord(e0, c->stream);
                    launch();
                }
                (void)hipEventRecord(e1, c->stream);
                (void)hipEventSynchronize(e1);
                float ms = 0.f;
                (void)hipEventElapsedTime(&ms, e0, e1);
                return 1e3 * ms / reps;
            }
            const int reps = 6;
            double total = 0.0;
            for (int rep = -1; rep < reps; ++rep) {
                hipLaunchKernelGGL(k_flush_read, dim3(4096), dim3(256), 0, c->stream, (const double *)flushbuf, (int64_t)(flush_bytes / sizeof(double)), (double *)flushbuf);
                (void)hipEventRecord(e0, c->stream);
                launch();
                (void)hipEventRecord(e1, c->stream);
                (void)hipEventSynchronize(e1);
                float ms = 0.f;
                (void)hipEventElapsedTime(&ms, e0, e1);
                if (rep >= 0) total += ms;
            }
            return 1e3 * total / reps;
        };
        if (cold) fprintf(stderr, "[front tune] cold caches: every timed launch follows a read sweep over 512 MB\n");
        for (int k = 0; k < nb && ok; ++k)
            for (int sweep = 0; sweep < 2 && ok; ++sweep) {
                if (sweep == 1 && top_inv && k == nb - 1) continue;      // no backward launch there
                double best = 1e30;
                int bnb = 0, brb = 0;
                fprintf(stderr, "[front tune] band %d (heights %d-%d, %s, %lld %s, planes %d):", k, cuts[(size_t)k], cuts[(size_t)k + 1] - 1, sweep == 0 ? "fwd" : "bwd",
                        (long long)(sweep == 0 ? band_rows[(size_t)k] : band_cols[(size_t)k]), sweep == 0 ? "rows" : "cols", c->front_planes[k]);
                for (int tnb : {256, 1024})
                    for (int trb : {1, 2, 4}) {
                        if (tnb == 1024 && d.TP > 256) continue;
                        std::vector<FrontWork> list;
                        if (sweep == 0) make_fwd(k, trb, list); else make_bwd(k, trb, list);
                        deal(list, 0, by_band[(size_t)k].size());
                        if (list.empty()) continue;
                        void *dl = nullptr;
                        if (hipMalloc(&dl, sizeof(FrontWork) * list.size()) != hipSuccess) { ok = false; break; }
                        (void)hipMemcpyAsync(dl, list.data(), sizeof(FrontWork) * list.size(), hipMemcpyHostToDevice, c->stream);
                        const double us = time_us([&]() {
                            if (sweep == 0) front_launch_fwd(c, f, (const FrontWork *)dl, (int)list.size(), tnb, trb, c->front_planes[k], vec[0], vec[1]);
                            else front_launch_bwd(c, f, (const FrontWork *)dl, (int)list.size(), tnb, trb, vec[1], vec[2]);
                        });
                        (void)hipFree(dl);
                        const bool cur = tnb == (sweep == 0 ? c->front_fwd_nb : c->front_bwd_nb)[k] && trb == (sweep == 0 ? c->front_fwd_rb : c->front_bwd_cb)[k] &&
                                         !(sweep == 0 && c->front_fwd_qw[k] >= 0);
                        fprintf(stderr, " %dx%d %.2f%s", tnb, trb, us, cur ? "*" : "");
                        if (us < best) { best = us; bnb = tnb; brb = trb; }
                    }
                int bqs = -1;
                if (sweep == 0) {
                    for (int qs = 0; (1 << qs) <= rows_groups && qs <= 4; ++qs) {
                        std::vector<FrontWork> list;
                        const int lds_cols = make_fwd_rows(k, rows_per_wg(qs), list);
                        if ((size_t)lds_cols * (size_t)(d.TP + FWD_ROWS_PAD) * sizeof(double) > FWD_ROWS_LDS_MAX || list.empty()) continue;
                        deal(list, 0, by_band[(size_t)k].size());
                        void *dl = nullptr;
                        if (hipMalloc(&dl, sizeof(FrontWork) * list.size()) != hipSuccess) { ok = false; break; }
                        (void)hipMemcpyAsync(dl, list.data(), sizeof(FrontWork) * list.size(), hipMemcpyHostToDevice, c->stream);
                        const double us = time_us([&]() { front_launch_fwd_rows(c, f, (const FrontWork *)dl, (int)list.size(), qs, c->front_planes[k], lds_cols, vec[0], vec[1]); });
                        (void)hipFree(dl);
                        fprintf(stderr, " r%d %.2f%s", 1 << qs, us, c->front_fwd_qw[k] == qs ? "*" : "");
                        if (us < best) { best = us; bqs = qs; }
                    }
                }
                if (bqs >= 0) fprintf(stderr, "  -> r%d\n", 1 << bqs);
                else fprintf(stderr, "  -> %dx%d\n", bnb, brb);
                if (apply > 1 && bnb) {
                    (sweep == 0 ? c->front_fwd_nb : c->front_bwd_nb)[k] = bnb;
                    (sweep == 0 ? c->front_fwd_rb : c->front_bwd_cb)[k] = brb;
                    if (sweep == 0) c->front_fwd_qw[k] = bqs;
                }
            }
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        (void)hipStreamSynchronize(c->stream);
        (void)hipMemsetAsync(f.W, 0, sizeof(double) * ((size_t)std::max<int64_t>(wrows, 1) << d.tp_shift), c->stream);
        (void)hipStreamSynchronize(c->stream);
        for (void *p2 : tmp) (void)hipFree(p2);
    }
    // (Tried in round 3 and rejected: the independent subtrees below the top of the tree on 2 or 4 streams, so that one stream's
    // ramp-up and drain overlap another's streaming -- launches from several streams do not overlap here, every launch costs
    // ~10 us more: sphere10k solve 85 -> 169 -> 272 us, torus100k 753 -> 804 -> 1061 us; profiles/studies/r03_lanes_experiment.txt)
    std::vector<FrontWork> fwd, bwd;
    for (int k = 0; k < nb; ++k) {
        c->front_fwd_ptr[k] = (int)fwd.size();
        c->front_bwd_ptr[k] = (int)bwd.size();
        if (c->front_fwd_qw[k] >= 0) {
            c->front_fwd_lds[k] = make_fwd_rows(k, rows_per_wg(c->front_fwd_qw[k]), fwd);
            if ((size_t)c->front_fwd_lds[k] * (size_t)(d.TP + FWD_ROWS_PAD) * sizeof(double) > FWD_ROWS_LDS_MAX) {
                front_release(c);
                return bad("DOTS_FRONT_CFG: the row kernel does not fit a band it was forced on (its right-hand side exceeds the LDS budget)");
            }
        } else {
            make_fwd(k, c->front_fwd_rb[k], fwd);
        }
        make_bwd(k, c->front_bwd_cb[k], bwd);
        deal(fwd, (size_t)c->front_fwd_ptr[k], by_band[(size_t)k].size());
        deal(bwd, (size_t)c->front_bwd_ptr[k], by_band[(size_t)k].size());
    }
    c->front_fwd_ptr[nb] = (int)fwd.size();
    c->front_bwd_ptr[nb] = (int)bwd.size();
    FUP(f, fwd_desc, fwd.data(), std::max<size_t>(fwd.size(), 1)); FUP(f, bwd_desc, bwd.data(), std::max<size_t>(bwd.size(), 1));
#undef FUP
    c->front = f;
    c->use_front = 1;
    c->front_heights = h->n_levels;
    c->front_top_inverse = top_inv ? 1 : 0;
    return 0;
}

// the leaves' band as explicit local inverses: one workgroup per leaf
static void front_launch_leaves(Ctx *c, const FrontDev &f, bool forward, const double *bhat, double *x) {
    const Dev &d = c->dcg;
    const FrontArgs g{d.tp_shift, d.TP, d.cg_ncol};
    const size_t lds = sizeof(double) * (forward ? 2 : 1) * (size_t)f.leaf_nmax * (size_t)d.TP;
    const bool v2 = front_two_modes(c);
    const int lanes = d.TP / (v2 ? 2 : 1);
    // (sixteen rows in flight per workgroup at every pitch -- 1024 threads at a pitch of 128 -- lose: torus65k_T127 110 / 87 -> 115 / 110 us per launch,
    // one workgroup per CU instead of four; workgroups grow only where a row of modes needs more than 256 lanes)
#ifdef DOTS_LEAF_NB      // (A/B)
    const int nbt = DOTS_LEAF_NB;
#else
    const int nbt = lanes <= 256 ? 256 : (lanes <= 512 ? 512 : 1024);
#endif
    const bool tab = f.leaf_bd != nullptr;
#define LEAF_LAUNCH2(VECV, NBV, TABV)                                                                                                                \
    do {                                                                                                                                             \
        if (forward) hipLaunchKernelGGL((k_front_leaf_fwd<VECV, NBV, TABV>), dim3(f.n_leaves), dim3(NBV), lds, c->stream, g, f, d.rowptr, d.col, d.val, bhat);      \
        else hipLaunchKernelGGL((k_front_leaf_bwd<VECV, NBV, TABV>), dim3(f.n_leaves), dim3(NBV), lds, c->stream, g, f, d.rowptr, d.col, d.val, bhat, x);           \
    } while (0)
#define LEAF_LAUNCH(VECV, NBV) do { if (tab) LEAF_LAUNCH2(VECV, NBV, true); else LEAF_LAUNCH2(VECV, NBV, false); } while (0)
    if (v2) { if (nbt == 256) LEAF_LAUNCH(2, 256); else if (nbt == 512) LEAF_LAUNCH(2, 512); else LEAF_LAUNCH(2, 1024); }
    else { if (nbt == 256) LEAF_LAUNCH(1, 256); else if (nbt == 512) LEAF_LAUNCH(1, 512); else LEAF_LAUNCH(1, 1024); }
#undef LEAF_LAUNCH
#undef LEAF_LAUNCH2
}

int front_solve(Ctx *c, const double *bhat, double *y, double *x) {
    const FrontDev &f = c->front;
    if (f.n_nodes == 0) { set_error("front_solve: no factor installed"); return DOTS_ERR_STATE; }
    for (int l = 0; l < f.n_levels; ++l) {
        if (l == 0 && f.n_leaves > 0) { front_launch_leaves(c, f, true, bhat, x); continue; }
        const int n = c->front_fwd_ptr[l + 1] - c->front_fwd_ptr[l];
        // (a top band of explicit inverses writes the solution itself)
        double *out = (c->front_top_inverse && l == f.n_levels - 1) ? x : y;
        if (n > 0 && c->front_fwd_qw[l] >= 0) front_launch_fwd_rows(c, f, f.fwd_desc + c->front_fwd_ptr[l], n, c->front_fwd_qw[l], c->front_planes[l], c->front_fwd_lds[l], bhat, out);
        else if (n > 0) front_launch_fwd(c, f, f.fwd_desc + c->front_fwd_ptr[l], n, c->front_fwd_nb[l], c->front_fwd_rb[l], c->front_planes[l], bhat, out);
    }
    for (int l = f.n_levels - 1 - (c->front_top_inverse ? 1 : 0); l >= 0; --l) {
        if (l == 0 && f.n_leaves > 0) { front_launch_leaves(c, f, false, bhat, x); continue; }
        const int n = c->front_bwd_ptr[l + 1] - c->front_bwd_ptr[l];
        if (n > 0) front_launch_bwd(c, f, f.bwd_desc + c->front_bwd_ptr[l], n, c->front_bwd_nb[l], c->front_bwd_cb[l], y, x);
    }
    DOTS_HIP(hipGetLastError());
    return 0;
}

}  // namespace dots
