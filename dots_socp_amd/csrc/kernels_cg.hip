// Preconditioned CG for the space-time Laplacian of step 1 (replaces the reference's
// eigen-decomposition + SuperLU solve, utils/laplacian_inverse_socp.py:11-61).
//
// Operator.  K = -(L_time (x) M + I (x) L_space) + eps I (x) M, with L_space the cotangent
// Laplacian (V x V CSR, assembled once from the mesh), M = diag(vertex mass) and L_time the
// Neumann second difference on the T+1 time nodes.  K is never expanded to an N x N matrix: with
// the time-fastest layout x[v][t] it is applied as a sparse-matrix x dense-block product
//     (K x)[v][:] = sum_u K_space[v,u] x[u][:]  +  mass_v * (time stencil of x[v][:])
// so the CSR (12 B/nnz) is read once per application instead of once per time node and every
// neighbour gather is one contiguous row of T+1 doubles.  Each workgroup stages the CSR entries of
// its block of rows in LDS (coalesced load, then broadcast reads).
//
// Two solvers share the kernels (template MODAL):
//   SPACETIME  one Jacobi-PCG on the coupled operator, one set of scalars.
//   MODAL      the orthonormal time eigen-basis Q (DCT-II) block-diagonalises K into T+1 shifted
//              surface problems K_space + (sigma_a + eps) M; they are solved as one batched PCG with
//              per-column scalars; converged columns are frozen and skipped.  Preconditioner: Jacobi,
//              or the multigrid V-cycle of kernels_mg.hip when a hierarchy has been uploaded.
//
// One Jacobi-PCG iteration is TWO kernels and no host round trip:
//   k_cg_apply   beta from the r.z partial sums of the previous kernel; p = z + beta p_old formed on the
//                fly at every gathered entry; writes p, Ap; emits partial sums of p.Ap
//   k_cg_update  alpha from the p.Ap partial sums; x += alpha p, r -= alpha Ap, z = D^-1 r; emits r.z
// (with multigrid the V-cycle kernels run between the two and the last of them emits r.z).
// Every workgroup re-reduces the (few hundred) per-workgroup partial sums of the kernel before it, in
// the same fixed order, so all workgroups see bit-identical scalars: no atomics, no separate reduction
// kernels, deterministic results.  Scalars that outlive a kernel (r.z totals, done flags, iteration
// counter) are written by workgroup 0 only, into the buffer of the current parity; readers of the
// same launch use the other parity.  The loop body is captured once in a hipGraph; the host polls
// the done flags between replays.
// Stopping rule (both preconditioners): r^T D^-1 r <= cg_tol^2 * b^T D^-1 b with D = diag(K).
#include "dots_dev.h"

#include <algorithm>
#include <vector>

namespace dots {

using S = CgScalOffsets;
constexpr int CG_NB = 1024;          // largest workgroup of the PCG kernels (16 wave64); large problems use CG_NB_LARGE
constexpr int CG_NB_LARGE = 256;

template <bool MODAL>
__device__ __forceinline__ double shift_of(const Dev &d, int t) {
    if (MODAL) return d.sigma[t];
    return ((t == 0 || t == d.T) ? 1.0 : 2.0) / (d.h * d.h);
}
template <bool MODAL>
__device__ __forceinline__ int n_cols(const Dev &d) { return MODAL ? d.cg_ncol : d.T + 1; }

// Dynamic LDS carve-up of the PCG kernels.
struct CgLds {
    double *val;    // [cap]   staged CSR values
    double *red;    // [blockDim] reduction scratch
    double *tot;    // [blockDim] column totals / broadcast
    int *col;       // [cap]
    int *rp;        // [VT + 2]
};
__device__ __forceinline__ CgLds carve(unsigned char *base, int cap) {
    CgLds l;
    const int nb = blockDim.x;
    l.val = reinterpret_cast<double *>(base);
    l.red = l.val + cap;
    l.tot = l.red + nb;
    l.col = reinterpret_cast<int *>(l.tot + nb);
    l.rp = l.col + cap;
    return l;
}
static size_t cg_lds_bytes(int cap, int vt, int nb) { return sizeof(double) * ((size_t)cap + 2 * nb) + sizeof(int) * ((size_t)cap + vt + 2); }

// Sum over workgroups of a partial array part[G][TP] (column fastest: one coalesced row per workgroup)
// for the column of the calling thread (MODAL: c = tid & (TP-1); otherwise the single column 0).
template <bool MODAL>
__device__ __forceinline__ double column_total(const Dev &d, const double *part, int G, const CgLds &l) {
    const int tid = threadIdx.x, NB = blockDim.x;
    if (G == 1) {   // already collapsed by k_collapse: one row, no reduction and no barrier
        if (!MODAL) return part[0];
        const int c = tid & (d.TP - 1);
        return c < d.cg_ncol ? part[c] : 0.0;
    }
    double s = 0.0;
    if (MODAL) {
        const int c = tid & (d.TP - 1), j = tid >> d.tp_shift, J = NB >> d.tp_shift;
        if (c < d.cg_ncol)
            for (int g = j; g < G; g += J) s += part[((int64_t)g << d.tp_shift) + c];
        l.red[tid] = s;
        __syncthreads();
        if (j == 0) {
            double t = 0.0;
            for (int k = 0; k < J; ++k) t += l.red[c + (k << d.tp_shift)];
            l.tot[c] = t;
        }
        __syncthreads();
        const double out = l.tot[c];
        __syncthreads();
        return out;
    } else {
        for (int g = tid; g < G; g += NB) s += part[g];
        s = wave_sum(s);
        if ((tid & 63) == 0) l.red[tid >> 6] = s;
        __syncthreads();
        if (tid == 0) {
            double t = 0.0;
            for (int k = 0; k < NB / 64; ++k) t += l.red[k];
            l.tot[0] = t;
        }
        __syncthreads();
        const double out = l.tot[0];
        __syncthreads();
        return out;
    }
}

// Emit the per-workgroup partial sum(s) of one value per thread into part[G][TP] (fixed order).
template <bool MODAL>
__device__ __forceinline__ void emit_partial(const Dev &d, double acc, double *part, const CgLds &l) {
    const int tid = threadIdx.x, NB = blockDim.x;
    if (MODAL && d.TP <= 64) {
        // lanes of a wave that share a column are TP apart: fold them with xor-shuffles, then one LDS
        // slot per (wave, column) and a single barrier
        double s = acc;
        for (int o = 32; o >= d.TP; o >>= 1) s += __shfl_xor(s, o, 64);
        const int lane = tid & 63, w = tid >> 6, nw = NB >> 6;
        if (lane < d.TP) l.red[w * d.TP + lane] = s;
        __syncthreads();
        if (tid < d.cg_ncol) {
            double t = 0.0;
            for (int k = 0; k < nw; ++k) t += l.red[k * d.TP + tid];
            part[((int64_t)blockIdx.x << d.tp_shift) + tid] = t;
        }
        __syncthreads();
    } else if (MODAL) {
        const int c = tid & (d.TP - 1), j = tid >> d.tp_shift, J = NB >> d.tp_shift;
        l.red[tid] = acc;
        __syncthreads();
        if (j == 0 && c < d.cg_ncol) {
            double t = 0.0;
            for (int k = 0; k < J; ++k) t += l.red[c + (k << d.tp_shift)];
            part[((int64_t)blockIdx.x << d.tp_shift) + c] = t;
        }
        __syncthreads();
    } else {
        double s = wave_sum(acc);
        if ((tid & 63) == 0) l.red[tid >> 6] = s;
        __syncthreads();
        if (tid == 0) {
            double t = 0.0;
            for (int k = 0; k < NB / 64; ++k) t += l.red[k];
            part[blockIdx.x] = t;
        }
        __syncthreads();
    }
}

struct CgArgs {
    const double *zin;     // z (FUSE_P) or x (plain apply)
    const double *p_old;
    double *p_new;
    double *out;           // Ap / y
    double *x;
    double eps, tol2;
    int parity;            // which r.z buffer this iteration reads
    int ept;               // elements per thread
    int nb;                // threads per workgroup (1024, or 256 for large problems)
    int vt;                // vertices per tile = nb * ept / TP
    int cap;               // staged CSR capacity
    int G;                 // workgroups (= tiles, padded to a multiple of 8)
    int nc;                // scalar columns
    int pstride;           // doubles per partial array = G * TP (MODAL) or G
    int stage;             // 1: stage the CSR row block in LDS, 0: read it through L1
    int mg;                // 1: z comes from the multigrid V-cycle (r.z and the stopping norm are separate sums)
    int collapse;          // 1: a small kernel sums the G partial rows once and consumers read that single row
    int Gr;                // rows a consumer re-reduces: G, or 1 when collapsed
    int prow;              // doubles per partial row: TP (MODAL) or 1
};

// Partial-sum arrays of the PCG in Dev::partials: r.z (two parities), p.Ap, bref, stopping norm (two
// parities), G rows each; behind them one collapsed row per array (used when a.collapse).
constexpr int N_PART_ARRAYS = 6;
__device__ __host__ __forceinline__ int64_t part_at(const CgArgs &a, int k) { return (int64_t)k * a.pstride; }
__device__ __host__ __forceinline__ int64_t coll_at(const CgArgs &a, int k) { return (int64_t)N_PART_ARRAYS * a.pstride + (int64_t)k * a.prow; }
__device__ __host__ __forceinline__ int idx_rz(int parity) { return parity; }
__device__ __host__ __forceinline__ int idx_crit(const CgArgs &a, int parity) { return a.mg ? 4 + parity : parity; }
constexpr int IDX_PAP = 2, IDX_BREF = 3;
// where producers write ...
__device__ __host__ __forceinline__ int64_t part_rz(const CgArgs &a, int parity) { return part_at(a, idx_rz(parity)); }
__device__ __host__ __forceinline__ int64_t part_pap(const CgArgs &a) { return part_at(a, IDX_PAP); }
__device__ __host__ __forceinline__ int64_t part_bref(const CgArgs &a) { return part_at(a, IDX_BREF); }
__device__ __host__ __forceinline__ int64_t part_crit(const CgArgs &a, int parity) { return part_at(a, idx_crit(a, parity)); }
// ... and where consumers read (a.Gr rows)
__device__ __host__ __forceinline__ int64_t read_at(const CgArgs &a, int k) { return a.collapse ? coll_at(a, k) : part_at(a, k); }
constexpr int SC_RZ0 = S::RZ, SC_RZ1 = S::PAP;   // the two parities of the r.z totals

// ------------------------------------------------------------------------------------------
// Operator application with the CSR row block staged in LDS.
// ------------------------------------------------------------------------------------------
template <bool MODAL, bool FUSE_P>
__global__ __launch_bounds__(CG_NB) void k_cg_apply(Dev d, CgArgs a) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const CgLds l = carve(lds_raw, a.cap);
    const int tid = threadIdx.x;
    const int col_of_thread = MODAL ? (tid & (d.TP - 1)) : 0;
    const int ncols = n_cols<MODAL>(d);

    double beta = 0.0;
    bool frozen = false;
    if (FUSE_P) {
        // scalars of this iteration, recomputed identically by every workgroup
        const double rz_new = column_total<MODAL>(d, d.partials + read_at(a, idx_rz(a.parity)), a.Gr, l);
        const double crit = a.mg ? column_total<MODAL>(d, d.partials + read_at(a, idx_crit(a, a.parity)), a.Gr, l) : rz_new;
        const double rz_old = d.scal[(a.parity ? SC_RZ0 : SC_RZ1) + col_of_thread];
        const double bref = d.scal[S::BREF + col_of_thread];
        frozen = (d.flags[col_of_thread] != 0) || (crit <= a.tol2 * bref);
        beta = (rz_old > 0.0) ? rz_new / rz_old : 0.0;
        const bool col_valid = !MODAL || col_of_thread < ncols;
        const int all = __syncthreads_and((!col_valid || frozen) ? 1 : 0);
        if (blockIdx.x == 0) {
            if (tid < a.nc) {   // tid == column for the first nc threads (MODAL: TP <= CG_NB; else nc == 1)
                d.scal[(a.parity ? SC_RZ1 : SC_RZ0) + tid] = rz_new;
                d.scal[S::ALPHA + tid] = crit;      // last stopping-norm value (diagnostics)
                d.flags[tid] = frozen ? 1 : 0;
            }
            if (tid == 0 && !all) d.flags[FLAG_ITERS] += 1;
        }
        if (all) return;
    }

    const int tile = xcd_tile(blockIdx.x, a.G);
    double acc = 0.0;
    if (tile < a.G && tile * a.vt < d.V) {
        const int v0 = tile * a.vt;
        const int nrows = min(a.vt, d.V - v0);
        const int rp0 = d.rowptr[v0];
        const int cap = a.stage ? a.cap : 0;
        if (a.stage) {
            for (int i = tid; i <= nrows; i += a.nb) l.rp[i] = d.rowptr[v0 + i] - rp0;
            __syncthreads();
            const int nloc = min(l.rp[nrows], cap);
            for (int i = tid; i < nloc; i += a.nb) {
                l.col[i] = d.col[rp0 + i];
                l.val[i] = d.val[rp0 + i];
            }
            __syncthreads();
        }

        const bool use_p = FUSE_P && (beta != 0.0);
        for (int q = 0; q < a.ept; ++q) {
            const int e = tid + q * a.nb;
            const int vl = e >> d.tp_shift, t = e & (d.TP - 1);
            if (vl >= nrows || t >= ncols) continue;
            if (FUSE_P && MODAL && frozen) continue;
            const int v = v0 + vl;
            auto fetch = [&](int i) -> double {
                double x = a.zin[i];
                if (use_p) x += beta * a.p_old[i];
                return x;
            };
            double sum = 0.0;
            int j = a.stage ? l.rp[vl] : d.rowptr[v] - rp0;
            const int jend = a.stage ? l.rp[vl + 1] : d.rowptr[v + 1] - rp0;
            // gathers in batches of four independent loads (rows have ~7 entries)
            for (; j + 4 <= jend && j + 4 <= cap; j += 4) {
                const int u0 = l.col[j], u1 = l.col[j + 1], u2 = l.col[j + 2], u3 = l.col[j + 3];
                const double w0 = l.val[j], w1 = l.val[j + 1], w2 = l.val[j + 2], w3 = l.val[j + 3];
                const double x0 = fetch(idxV(d, u0, t)), x1 = fetch(idxV(d, u1, t)), x2 = fetch(idxV(d, u2, t)),
                             x3 = fetch(idxV(d, u3, t));
                sum += (w0 * x0 + w1 * x1) + (w2 * x2 + w3 * x3);
            }
            for (; j + 4 <= jend; j += 4) {   // entries that are not staged: straight from global (L1 broadcast)
                const int *cp = d.col + rp0 + j;
                const double *vp = d.val + rp0 + j;
                const int u0 = cp[0], u1 = cp[1], u2 = cp[2], u3 = cp[3];
                const double x0 = fetch(idxV(d, u0, t)), x1 = fetch(idxV(d, u1, t)), x2 = fetch(idxV(d, u2, t)),
                             x3 = fetch(idxV(d, u3, t));
                sum += (vp[0] * x0 + vp[1] * x1) + (vp[2] * x2 + vp[3] * x3);
            }
            for (; j < jend; ++j) {
                const int u = (j < cap) ? l.col[j] : d.col[rp0 + j];
                const double w = (j < cap) ? l.val[j] : d.val[rp0 + j];
                sum += w * fetch(idxV(d, u, t));
            }
            const int iv = idxV(d, v, t);
            const double pv = fetch(iv);
            const double m = d.mass_v[v];
            if (MODAL) {
                sum += (d.sigma[t] + a.eps) * m * pv;
            } else {
                const double pm = (t > 0) ? fetch(iv - 1) : 0.0;
                const double pp = (t < d.T) ? fetch(iv + 1) : 0.0;
                const double ct = (t == 0 || t == d.T) ? 1.0 : 2.0;
                sum += m * ((ct * pv - pm - pp) / (d.h * d.h) + a.eps * pv);
            }
            a.out[iv] = sum;
            if (FUSE_P) {
                a.p_new[iv] = pv;
                acc += pv * sum;
            }
        }
    }
    if (FUSE_P) emit_partial<MODAL>(d, acc, d.partials + part_pap(a), l);
}

// r = (b - bmean) - K x ;  z = D^-1 r (Jacobi only) ;  partial sums of r^T D^-1 r (parity 0) and b~^T D^-1 b~
template <bool MODAL>
__global__ __launch_bounds__(CG_NB) void k_cg_r0(Dev d, CgArgs a, const double *__restrict__ b, const double *__restrict__ Kx) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const CgLds l = carve(lds_raw, a.cap);
    const int tile = xcd_tile(blockIdx.x, a.G);
    const int ncols = n_cols<MODAL>(d);
    double a0 = 0.0, a1 = 0.0;
    if (tile < a.G) {
        for (int q = 0; q < a.ept; ++q) {
            const int e = threadIdx.x + q * a.nb;
            const int v = tile * a.vt + (e >> d.tp_shift), t = e & (d.TP - 1);
            if ((e >> d.tp_shift) >= a.vt || v >= d.V || t >= ncols) continue;
            const int iv = idxV(d, v, t);
            const int cidx = MODAL ? t : 0;
            const double bt = b[iv] - d.scal[S::BMEAN + cidx];
            const double r = bt - Kx[iv];
            const double dinv = 1.0 / (d.kdiag[v] + (shift_of<MODAL>(d, t) + a.eps) * d.mass_v[v]);
            const double z = dinv * r;
            d.cg_r[iv] = r;
            d.cg_z[iv] = z;   // Jacobi z; with multigrid it is the D^-1 r the V-cycle starts from
            a0 += r * z;
            a1 += bt * dinv * bt;
        }
    }
    emit_partial<MODAL>(d, a0, d.partials + part_crit(a, 0), l);
    emit_partial<MODAL>(d, a1, d.partials + part_bref(a), l);
}

// once per solve: bref totals, zero the r.z totals of both parities, clear flags and the counter
template <bool MODAL>
__global__ __launch_bounds__(CG_NB) void k_cg_begin(Dev d, CgArgs a) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const CgLds l = carve(lds_raw, a.cap);
    const double bref = column_total<MODAL>(d, d.partials + read_at(a, IDX_BREF), a.Gr, l);
    const int tid = threadIdx.x;
    if (tid < a.nc) {
        d.scal[S::BREF + tid] = bref;
        d.scal[SC_RZ0 + tid] = 0.0;
        d.scal[SC_RZ1 + tid] = 0.0;
        d.flags[tid] = 0;
    }
    if (tid == 0) d.flags[FLAG_ITERS] = 0;
}

// x += alpha p ; r -= alpha Ap ; z = D^-1 r (Jacobi) ; partial sums of r^T D^-1 r into the other parity
template <bool MODAL>
__global__ __launch_bounds__(CG_NB) void k_cg_update(Dev d, CgArgs a) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const CgLds l = carve(lds_raw, a.cap);
    const int tid = threadIdx.x;
    const int col_of_thread = MODAL ? (tid & (d.TP - 1)) : 0;
    const int ncols = n_cols<MODAL>(d);
    const double pap = column_total<MODAL>(d, d.partials + read_at(a, IDX_PAP), a.Gr, l);
    const bool frozen = d.flags[col_of_thread] != 0;
    const bool col_valid = !MODAL || col_of_thread < ncols;
    if (__syncthreads_and((!col_valid || frozen) ? 1 : 0)) return;
    const double rz = d.scal[(a.parity ? SC_RZ1 : SC_RZ0) + col_of_thread];
    const double alpha = (frozen || !(pap > 0.0)) ? 0.0 : rz / pap;

    const int tile = xcd_tile(blockIdx.x, a.G);
    double acc = 0.0;
    if (tile < a.G) {
        for (int q = 0; q < a.ept; ++q) {
            const int e = tid + q * a.nb;
            const int v = tile * a.vt + (e >> d.tp_shift), t = e & (d.TP - 1);
            if ((e >> d.tp_shift) >= a.vt || v >= d.V || t >= ncols) continue;
            const int iv = idxV(d, v, t);
            const double dinv = 1.0 / (d.kdiag[v] + (shift_of<MODAL>(d, t) + a.eps) * d.mass_v[v]);
            if (MODAL && frozen) {      // keep the frozen column's norm in the sum so its total stays put
                const double r = d.cg_r[iv];
                acc += r * dinv * r;
                continue;
            }
            a.x[iv] += alpha * a.p_new[iv];
            const double r = d.cg_r[iv] - alpha * a.out[iv];
            const double z = r * dinv;
            d.cg_r[iv] = r;
            d.cg_z[iv] = z;   // Jacobi z; with multigrid it is the D^-1 r the V-cycle starts from
            acc += r * z;
        }
    }
    emit_partial<MODAL>(d, acc, d.partials + part_crit(a, a.parity ^ 1), l);
}

// mean of b over the null space (only when the operator is singular: eps == 0)
__global__ __launch_bounds__(BLOCK) void k_cg_bmean(Dev d, int nblk, int nc, double mean_scale) {
    __shared__ double lds[4];
    double v[1] = {0.0};
    for (int g = threadIdx.x; g < nblk; g += BLOCK) v[0] += d.partials[g];
    block_sum<1>(v, lds);
    if (threadIdx.x == 0) d.scal[S::BMEAN] = v[0] * mean_scale;
    for (int c = 1 + threadIdx.x; c < nc; c += BLOCK) d.scal[S::BMEAN + c] = 0.0;
}

// Time-mode transform of node arrays: FWD  y[v][a] = sum_t Q[t][a] x[v][t]   (time -> modes)
//                                     INV  y[v][t] = sum_a Q[t][a] x[v][a]   (modes -> time)
// The forward transform also emits the partial sums of column 0 (mean removal of the singular mode).
template <bool FWD>
__global__ __launch_bounds__(BLOCK) void k_time_modes(Dev d, const double *__restrict__ x, double *__restrict__ y, int emit_col0) {
    __shared__ double lds[4];
    const int tile = xcd_tile(blockIdx.x, d.n_vtiles);
    const int n = d.T + 1;
    double part[1] = {0.0};
    if (tile < d.n_vtiles) {
        for (int e = threadIdx.x; e < TILE_ELEMS; e += BLOCK) {
            const int v = tile * d.VT + (e >> d.tp_shift), j = e & (d.TP - 1);
            if (v >= d.V || j >= n) continue;
            const double *row = x + idxV(d, v, 0);
            double s = 0.0;
            if (FWD) {
                for (int i = 0; i < n; ++i) s += d.Q[i * n + j] * row[i];
            } else {
                for (int i = 0; i < n; ++i) s += d.Q[j * n + i] * row[i];
            }
            y[idxV(d, v, j)] = s;
            if (FWD && j == 0) part[0] += s;
        }
    }
    if (FWD && emit_col0) {
        block_sum<1>(part, lds);
        if (threadIdx.x == 0) d.partials[blockIdx.x] = part[0];
    }
}

// The same transform with the tile of x and Q (in chunks) staged in LDS: every x row and every Q entry is read
// from memory once per workgroup instead of once per output; a thread computes four outputs that share their Q
// column.  x rows are padded by one double so that the rows a wavefront broadcasts sit in different banks.
// NB = 1024 (one output per thread) where the launch is latency-bound: the inverse transform after the direct solve.
template <bool FWD, int NB = BLOCK>
__global__ __launch_bounds__(NB) void k_time_modes_tile(Dev d, const double *__restrict__ x, double *__restrict__ y, int IC) {
    extern __shared__ double tm_lds[];
    const int n = d.T + 1, TP = d.TP, TPp = TP + 1;
    double *Qs = tm_lds;                 // [IC][TP]
    double *xs = tm_lds + IC * TP;       // [VT][TPp]
    const int tile = xcd_tile(blockIdx.x, d.n_vtiles);
    if (tile >= d.n_vtiles) return;
    const int v0 = tile * d.VT;
    stage_q_chunk<FWD, NB>(d, d.Q, Qs, 0, min(IC, n));      // together with the tile's loads: one round trip, one barrier
    for (int e = threadIdx.x; e < TILE_ELEMS; e += NB) {
        const int vl = e >> d.tp_shift, t = e & (TP - 1);
        xs[vl * TPp + t] = (v0 + vl < d.V && t < n) ? x[idxV(d, v0 + vl, t)] : 0.0;
    }
    modes_from_tile<FWD, NB>(d, d.Q, xs, Qs, IC, v0, y, -1, 0, 1 << 30, true);
}

// The transform as a GEMM on the matrix cores (T + 1 >= 64; modes_from_tile_mfma in dots_dev.h): a workgroup
// stages the rows of 32 vertices in LDS, its four wavefronts split the 16 x 16 output tiles.
__global__ __launch_bounds__(BLOCK) void k_time_modes_mfma(Dev d, const double *__restrict__ Qe, const double *__restrict__ x, double *__restrict__ y) {
    extern __shared__ double xs_m[];                    // [TM_ROWS][TP + 1]
    const int n = d.T + 1, TP = d.TP, TPp = TP + 1;
    const int v0 = blockIdx.x * TM_ROWS;
    for (int e = threadIdx.x; e < TM_ROWS * TP; e += BLOCK) {
        const int vl = e >> d.tp_shift, t = e & (TP - 1);
        xs_m[vl * TPp + t] = (v0 + vl < d.V && t < n) ? x[idxV(d, v0 + vl, t)] : 0.0;
    }
    __syncthreads();
    modes_from_tile_mfma<BLOCK / 64>(d, Qe, xs_m, v0, y);
}

// ------------------------------------------------------------------------------------------
// host driver
// ------------------------------------------------------------------------------------------
// Sum the G rows of a partial array into one row (fixed order).  Used for large problems, where every
// workgroup re-reducing thousands of rows would cost more than one extra tiny kernel.
// Workgroup b sums the rows g = b, b + gridDim.x, ... and writes row b of dst: launched twice
// (COLLAPSE_STAGE workgroups, then one) it reduces G rows to one in two short kernels.
constexpr int COLLAPSE_STAGE = 64;
__global__ __launch_bounds__(CG_NB) void k_collapse(const double *__restrict__ src, int G, int row, double *__restrict__ dst) {
    __shared__ double red[CG_NB];
    const int tid = threadIdx.x;
    const int c = tid & (row - 1), j = tid / row, J = CG_NB / row;
    double s = 0.0;
    for (int g = blockIdx.x + j * gridDim.x; g < G; g += J * gridDim.x) s += src[(int64_t)g * row + c];
    red[tid] = s;
    __syncthreads();
    if (j == 0) {
        double t = 0.0;
        for (int k = 0; k < J; ++k) t += red[c + k * row];
        dst[(int64_t)blockIdx.x * row + c] = t;
    }
}

// (Round 4: both stages in ONE launch -- the last workgroup to finish, by a ticket, adds the stage rows -- gives the same totals bit for bit and is
// SLOWER: modal_pcg at torus100k 113.9 -> 99.5 it/s.  Even 64 agent-scope fences per launch write back and invalidate the L2s the
// cache-resident PCG lives on; profiles/studies/r04_reduce_with_dual_alpha.txt has the same finding for the KKT reduction.  Not kept.)
// One element per thread keeps the rows in flight per XCD well inside its L2 (measured: a 7-neighbour
// gather at V = 100k takes 39 us with 1 element per thread and 56 us with 4, profiles/micro).  Up to 1024
// workgroups their partial sums are re-reduced inside the consumer kernels; beyond that k_collapse does it.
static void cg_tiling(const Dev &d, int *nb_out, int *ept_out, int *vt_out, int *G_out, int *collapse_out) {
    const int ept = 1;
    int nb = CG_NB;
    int vt = std::max(1, nb * ept / d.TP);
    int G = xcd_grid((d.V + vt - 1) / vt);
    const int collapse = G > 1024 ? 1 : 0;
    if (collapse) {   // the partial rows are summed by k_collapse anyway: small workgroups, no barriers in the prologue
        nb = std::max(CG_NB_LARGE, d.TP);
        vt = std::max(1, nb * ept / d.TP);
        G = xcd_grid((d.V + vt - 1) / vt);
    }
    *nb_out = nb;
    *ept_out = ept;
    *vt_out = vt;
    *G_out = G;
    *collapse_out = collapse;
}

static CgArgs make_args(Ctx *c, bool modal) {
    const Dev &d = modal ? c->dcg : c->d;
    CgArgs a{};
    cg_tiling(d, &a.nb, &a.ept, &a.vt, &a.G, &a.collapse);
    a.cap = a.vt * 12 + 64;          // ~7 entries per row on a triangle mesh; entries beyond cap are read from global
    if (a.cap > 3072) a.cap = 3072;  // keep the dynamic LDS below 64 KB
    a.nc = modal ? d.cg_ncol : 1;
    a.prow = modal ? d.TP : 1;
    a.pstride = a.G * a.prow;
    a.Gr = a.collapse ? 1 : a.G;
    a.stage = c->cg_stage_lds;
    a.mg = (modal && c->mg.nlev > 1 && c->use_mg) ? 1 : 0;
    a.eps = c->prm.eps;
    a.tol2 = c->prm.cg_tol * c->prm.cg_tol;
    return a;
}

int64_t cg_partials_needed(const Dev &d) {
    int nb, ept, vt, G, collapse;
    cg_tiling(d, &nb, &ept, &vt, &G, &collapse);
    // G rows per array + one collapsed row each + the intermediate rows of the two-stage collapse
    return N_PART_ARRAYS * (int64_t)d.TP * (G + 1) + (int64_t)COLLAPSE_STAGE * d.TP;
}

// sum partial array k into its collapsed row (no-op for problems small enough to re-reduce in the consumers)
static void collapse_if_needed(Ctx *c, const Dev &d, const CgArgs &a, int k) {
    if (!a.collapse) return;
    double *stage = d.partials + coll_at(a, N_PART_ARRAYS);   // scratch behind the collapsed rows
    hipLaunchKernelGGL(k_collapse, dim3(COLLAPSE_STAGE), dim3(CG_NB), 0, c->stream, d.partials + part_at(a, k), a.G, a.prow, stage);
    hipLaunchKernelGGL(k_collapse, dim3(1), dim3(CG_NB), 0, c->stream, stage, COLLAPSE_STAGE, a.prow, d.partials + coll_at(a, k));
}

template <bool MODAL>
static int cg_iterations(Ctx *c, CgArgs a, double *x, int n_iter) {
    const Dev &d = MODAL ? c->dcg : c->d;
    const size_t lds = cg_lds_bytes(a.cap, a.vt, a.nb);
    for (int it = 0; it < n_iter; ++it) {
        const bool odd = (it & 1) != 0;
        a.parity = odd ? 1 : 0;
        a.zin = d.cg_z;
        a.p_old = odd ? d.cg_p1 : d.cg_p0;
        a.p_new = odd ? d.cg_p0 : d.cg_p1;
        a.out = d.cg_Ap;
        a.x = x;
        hipLaunchKernelGGL((k_cg_apply<MODAL, true>), dim3(a.G), dim3(a.nb), lds, c->stream, d, a);
        collapse_if_needed(c, d, a, IDX_PAP);
        hipLaunchKernelGGL((k_cg_update<MODAL>), dim3(a.G), dim3(a.nb), lds, c->stream, d, a);
        collapse_if_needed(c, d, a, idx_crit(a, a.parity ^ 1));
        if (a.mg) {
            int rc = mg_vcycle(c, d.cg_r, d.cg_z, d.cg_Ap, d.partials + part_rz(a, a.parity ^ 1), a.nb, a.ept, a.vt, a.G);
            if (rc) return rc;
            collapse_if_needed(c, d, a, idx_rz(a.parity ^ 1));
        }
    }
    DOTS_HIP(hipGetLastError());
    return 0;
}

// Instantiate (once per eps / tolerance / preconditioner) a hipGraph holding `unit` PCG iterations.
template <bool MODAL>
static int cg_graph_prepare(Ctx *c, const CgArgs &a, double *x, int unit) {
    if (c->cg_graph && c->cg_graph_iters == unit && c->cg_graph_eps == c->prm.eps && c->cg_graph_tol == c->prm.cg_tol &&
        c->cg_graph_mg == a.mg)
        return 0;
    if (c->cg_graph) {
        (void)hipGraphExecDestroy(c->cg_graph);
        c->cg_graph = nullptr;
    }
    hipGraph_t graph = nullptr;
    DOTS_HIP(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
    int rc = cg_iterations<MODAL>(c, a, x, unit);
    hipError_t e = hipStreamEndCapture(c->stream, &graph);
    if (rc != 0) return rc;
    DOTS_HIP(e);
    DOTS_HIP(hipGraphInstantiate(&c->cg_graph, graph, nullptr, nullptr, 0));
    (void)hipGraphDestroy(graph);
    c->cg_graph_iters = unit;
    c->cg_graph_eps = c->prm.eps;
    c->cg_graph_tol = c->prm.cg_tol;
    c->cg_graph_mg = a.mg;
    return 0;
}

// PCG on the node-layout right-hand side b (already in mode space when MODAL), solution in x.
template <bool MODAL>
static int cg_core(Ctx *c, const double *b, double *x, dots_step_stats *stats) {
    const Dev &d = MODAL ? c->dcg : c->d;
    CgArgs a = make_args(c, MODAL);
    const size_t lds = cg_lds_bytes(a.cap, a.vt, a.nb);
    a.zin = x;
    a.out = d.cg_Ap;
    hipLaunchKernelGGL((k_cg_apply<MODAL, false>), dim3(a.G), dim3(a.nb), lds, c->stream, d, a);
    hipLaunchKernelGGL((k_cg_r0<MODAL>), dim3(a.G), dim3(a.nb), lds, c->stream, d, a, b, d.cg_Ap);
    collapse_if_needed(c, d, a, idx_crit(a, 0));
    collapse_if_needed(c, d, a, IDX_BREF);
    hipLaunchKernelGGL((k_cg_begin<MODAL>), dim3(1), dim3(a.nb), lds, c->stream, d, a);
    DOTS_HIP(hipGetLastError());
    if (a.mg) {
        int rc = mg_vcycle(c, d.cg_r, d.cg_z, d.cg_Ap, d.partials + part_rz(a, 0), a.nb, a.ept, a.vt, a.G);
        if (rc) return rc;
        collapse_if_needed(c, d, a, idx_rz(0));
    }

    const int max_iter = c->prm.cg_max_iter > 0 ? c->prm.cg_max_iter : 10000;
    const int unit = a.mg ? 2 : 8;   // iterations per graph replay (even: the ping-pong parity is preserved)
    int rc = cg_graph_prepare<MODAL>(c, a, x, unit);
    if (rc != 0) return rc;
    int launched = 0, iters = 0;
    bool done = false;
    // first burst sized from the previous solve (iteration counts drift slowly along the ALM)
    int burst = c->last_cg_iters > 2 * unit ? ((c->last_cg_iters - unit / 2) / unit) * unit : unit;
    while (!done && launched < max_iter) {
        for (int k = 0; k < burst / unit; ++k) DOTS_HIP(hipGraphLaunch(c->cg_graph, c->stream));
        launched += burst;
        burst = unit;
        DOTS_HIP(hipMemcpyAsync(c->h_flags, d.flags, sizeof(int) * FLAG_TOTAL, hipMemcpyDeviceToHost, c->stream));
        DOTS_HIP(hipStreamSynchronize(c->stream));
        done = true;
        for (int k = 0; k < a.nc; ++k) done = done && (c->h_flags[k] != 0);
        iters = c->h_flags[FLAG_ITERS];
    }
    c->last_cg_iters = iters;
    if (stats) {
        DOTS_HIP(hipMemcpyAsync(c->h_pinned, d.scal, sizeof(double) * S::TOTAL, hipMemcpyDeviceToHost, c->stream));
        DOTS_HIP(hipStreamSynchronize(c->stream));
        double worst = 0.0;
        for (int k = 0; k < a.nc; ++k) {
            const double br = c->h_pinned[S::BREF + k], cr = c->h_pinned[S::ALPHA + k];
            if (br > 0.0 && cr / br > worst) worst = cr / br;
        }
        stats->cg_last_rel_residual = sqrt(worst);
        stats->cg_last_iterations = iters;
        stats->cg_iterations += iters;
        if (!done) stats->cg_not_converged += 1;
    }
    return 0;
}

// ---- transforms of a time-slab context (multi-GPU) -------------------------------------------------------------
// Both read an all-gathered buffer of R chunks: chunk p holds [V][pitch] doubles with `stride` live columns, column j of
// chunk p = global index p * stride + j (time node for the right-hand side, time mode for the solution).
struct Gathered {
    const double *base;
    int64_t chunk;       // doubles per rank
    int shift;           // log2 of the pitch inside a chunk
    int stride;          // live columns per chunk
};
__device__ __forceinline__ double gathered_at(const Gathered &G, int v, int i) {
    const int p = i / G.stride, j = i - p * G.stride;
    return G.base[p * G.chunk + ((int64_t)v << G.shift) + j];
}

// Forward transform restricted to the modes [a0, a0 + g.cg_ncol) this context solves, from the gathered right-hand
// side:  y[v][j] = sum_t Q[t][a0 + j] b[v][t]  (pitch of the PCG view).  Emits the column-0 partial sums (mean removal
// of the singular mode) when asked to.   dt: the global-time view (T, Q).
__global__ __launch_bounds__(BLOCK) void k_time_modes_fwd_sub(Dev dt, Dev g, int a0, Gathered x, double *__restrict__ y, int emit_col0) {
    __shared__ double lds[4];
    const int n = dt.T + 1, nloc = g.cg_ncol;
    double part[1] = {0.0};
    const int64_t total = (int64_t)dt.V << g.tp_shift;
    for (int64_t e = (int64_t)blockIdx.x * BLOCK + threadIdx.x; e < total; e += (int64_t)gridDim.x * BLOCK) {
        const int v = (int)(e >> g.tp_shift), j = (int)(e & (g.TP - 1));
        if (j >= nloc) continue;
        double s = 0.0;
        for (int i = 0; i < n; ++i) s += dt.Q[i * n + a0 + j] * gathered_at(x, v, i);
        y[idxV(g, v, j)] = s;
        if (emit_col0 && j == 0) part[0] += s;
    }
    if (emit_col0) {
        block_sum<1>(part, lds);
        if (threadIdx.x == 0) dt.partials[blockIdx.x] = part[0];
    }
}

// Tiled forms (same staging as k_time_modes_tile; dt = the global-time view: pitch >= T + 1).  Forward: stores only
// this context's modes with the PCG view's pitch.
__global__ __launch_bounds__(BLOCK) void k_time_modes_fwd_sub_tile(Dev dt, int a0, int nloc, int out_shift, Gathered x, double *__restrict__ y, int IC) {
    extern __shared__ double tm_lds[];
    const int n = dt.T + 1, TP = dt.TP, TPp = TP + 1;
    double *Qs = tm_lds, *xs = tm_lds + IC * TP;
    const int tile = xcd_tile(blockIdx.x, dt.n_vtiles);
    if (tile >= dt.n_vtiles) return;
    const int v0 = tile * dt.VT;
    for (int e = threadIdx.x; e < TILE_ELEMS; e += BLOCK) {
        const int vl = e >> dt.tp_shift, t = e & (TP - 1);
        xs[vl * TPp + t] = (v0 + vl < dt.V && t < n) ? gathered_at(x, v0 + vl, t) : 0.0;
    }
    modes_from_tile<true>(dt, dt.Q, xs, Qs, IC, v0, y, out_shift, a0, nloc);
}

// Inverse: phi of this slab's nodes [t0, t0 + nl) (pitch of the slab, 1 << out_shift) from the gathered mode-space
// solution, and phi at node t0 + nl (the next slab's first node) into phi_hi: the same sum, in the same order, as the
// owner of that node forms -- no exchange of phi is ever needed.
__global__ __launch_bounds__(BLOCK) void k_time_modes_inv_gathered_tile(Dev dt, Gathered x, double *__restrict__ y, int out_shift, int t0, int nl,
                                                                        double *__restrict__ phi_hi, int IC) {
    extern __shared__ double tm_lds[];
    const int n = dt.T + 1, TP = dt.TP, TPp = TP + 1;
    double *Qs = tm_lds, *xs = tm_lds + IC * TP;
    const int tile = xcd_tile(blockIdx.x, dt.n_vtiles);
    if (tile >= dt.n_vtiles) return;
    const int v0 = tile * dt.VT;
    for (int e = threadIdx.x; e < TILE_ELEMS; e += BLOCK) {
        const int vl = e >> dt.tp_shift, a = e & (TP - 1);
        xs[vl * TPp + a] = (v0 + vl < dt.V && a < n) ? gathered_at(x, v0 + vl, a) : 0.0;
    }
    modes_from_tile<false>(dt, dt.Q, xs, Qs, IC, v0, y, out_shift, t0, nl);
    const int th = t0 + nl;      // first node of the next slab
    if (th < n && (int)threadIdx.x < dt.VT && v0 + (int)threadIdx.x < dt.V) {
        const double *x0 = xs + threadIdx.x * TPp;
        double acc = 0.0;
        for (int a = 0; a < n; ++a) acc += dt.Q[th * n + a] * x0[a];
        phi_hi[v0 + threadIdx.x] = acc;
    }
}

// Step 1 for the modes of this context.  Unsharded: also transforms back (phi is complete on return).
// Sharded: the local mode-space solution stays in dcg.cg_x for the caller to exchange (cg_finish_sharded).
template <bool MODAL>
static int cg_solve_impl(Ctx *c, dots_step_stats *stats, bool defer_inverse) {
    const Dev &d = c->d;
    const Dev &g = c->dcg;
    const int gt = xcd_grid(d.n_vtiles);
    const bool singular = (c->prm.eps == 0.0);
    double *x = MODAL ? g.cg_x : d.phi;
    const double *b = d.cg_b;
    const bool sharded = MODAL && c->shard_count > 0;
    const bool owns_mode0 = !sharded || c->shard_begin == 0;
    if (MODAL && g.cg_ncol == 0) return 0;   // a rank without modes has nothing to solve
    const bool direct = MODAL && c->use_front && c->front.n_nodes > 0;   // no warm start, no mean removal needed
    if (sharded) {
        // time-slab context: the right-hand side of ALL nodes was all-gathered into slab.b_recv; this rank transforms the
        // modes it solves.  The PCG warm start is its own mode-space solution of the previous iteration (g.cg_x).
        const Dev &dt = c->dgt;
        const Gathered bg{c->slab.b_recv, c->slab_b_chunk, d.tp_shift, c->shard_stride};
        const int gs = 1024;   // grid-stride; k_cg_bmean sums exactly this many partial sums
        if (direct && time_modes_tile_ok(dt))     // no mean removal with the direct solver: the tiled transform
            hipLaunchKernelGGL(k_time_modes_fwd_sub_tile, dim3(xcd_grid(dt.n_vtiles)), dim3(BLOCK), time_modes_tile_lds(dt), c->stream, dt, c->shard_begin,
                               g.cg_ncol, g.tp_shift, bg, g.cg_p0, time_modes_chunk(dt));
        else
            hipLaunchKernelGGL(k_time_modes_fwd_sub, dim3(gs), dim3(BLOCK), 0, c->stream, dt, g, c->shard_begin, bg, g.cg_p0, owns_mode0 ? 1 : 0);
        b = g.cg_p0;
        const double mean_scale = (singular && owns_mode0) ? 1.0 / d.V : 0.0;
        if (!direct) hipLaunchKernelGGL(k_cg_bmean, dim3(1), dim3(BLOCK), 0, c->stream, d, owns_mode0 ? gs : 0, g.cg_ncol, mean_scale);
    } else {
        if (MODAL) {
            // b^ = Q^T b (into p0 as scratch), x^ = Q^T phi (warm start in mode space)
            if (rhs_writes_modes(c)) {
                // k_rhs_modes (kernels_alm.hip) already left the mode-space right-hand side in cg_p0
            } else if (direct && time_modes_tile_ok(d))
                hipLaunchKernelGGL((k_time_modes_tile<true>), dim3(gt), dim3(BLOCK), time_modes_tile_lds(d), c->stream, d, d.cg_b, d.cg_p0, time_modes_chunk(d));
            else
                hipLaunchKernelGGL((k_time_modes<true>), dim3(gt), dim3(BLOCK), 0, c->stream, d, d.cg_b, d.cg_p0, 1);
            if (!direct) hipLaunchKernelGGL((k_time_modes<true>), dim3(gt), dim3(BLOCK), 0, c->stream, d, d.phi, d.cg_x, 0);
            b = d.cg_p0;   // consumed by k_cg_r0 before iteration 0 (which reads no p_old: beta = 0) writes p1
        }
        const double mean_scale = !singular ? 0.0 : (MODAL ? 1.0 / d.V : 1.0 / ((double)d.V * (d.T + 1)));
        if (!direct) hipLaunchKernelGGL(k_cg_bmean, dim3(1), dim3(BLOCK), 0, c->stream, d, gt, MODAL ? d.cg_ncol : 1, mean_scale);
    }
    DOTS_HIP(hipGetLastError());
    int rc;
    if (direct) {
        // direct solve: the two triangular sweeps of the multifrontal factor (kernels_front.hip)
        rc = front_solve(c, b, g.cg_z, x);
        c->last_cg_iters = 0;
        if (stats) stats->cg_last_iterations = 0;
    } else {
        rc = cg_core<MODAL>(c, b, x, stats);
    }
    if (rc) return rc;
    if (MODAL && !sharded && !defer_inverse) {
        if (time_modes_mfma_ok(d))
            hipLaunchKernelGGL(k_time_modes_mfma, dim3((d.V + TM_ROWS - 1) / TM_ROWS), dim3(BLOCK), sizeof(double) * TM_ROWS * (d.TP + 1), c->stream, d, d.QpadT,
                               d.cg_x, d.phi);
        else if (time_modes_tile_ok(d) && direct && d.n_vtiles <= 512)      // small meshes, behind the sweeps: latency-bound, one output per thread (knot: 9.5 -> 6 us; no gain at 10^5 vertices)
            hipLaunchKernelGGL((k_time_modes_tile<false, 1024>), dim3(gt), dim3(1024), time_modes_tile_lds(d), c->stream, d, d.cg_x, d.phi, time_modes_chunk(d));
        else if (time_modes_tile_ok(d))
            hipLaunchKernelGGL((k_time_modes_tile<false>), dim3(gt), dim3(BLOCK), time_modes_tile_lds(d), c->stream, d, d.cg_x, d.phi, time_modes_chunk(d));
        else
            hipLaunchKernelGGL((k_time_modes<false>), dim3(gt), dim3(BLOCK), 0, c->stream, d, d.cg_x, d.phi, 0);
        DOTS_HIP(hipGetLastError());
    }
    return 0;
}

int cg_solve(Ctx *c, dots_step_stats *stats, bool defer_inverse) {
    return c->lap_solver == DOTS_LAP_MODAL_PCG ? cg_solve_impl<true>(c, stats, defer_inverse) : cg_solve_impl<false>(c, stats, false);
}

// phi of this slab from the mode-space solutions of all ranks (slab.x_recv = [n_ranks][V][mode pitch])
int cg_finish_sharded(Ctx *c) {
    const Dev &d = c->d, &dt = c->dgt;
    if (d.nl == 0) return 0;
    if (!time_modes_tile_ok(dt)) { set_error("time slabs need T + 1 <= 256"); return DOTS_ERR_STATE; }
    const Gathered xg{c->slab.x_recv, c->slab_x_chunk, c->dcg.tp_shift, c->shard_stride};
    hipLaunchKernelGGL(k_time_modes_inv_gathered_tile, dim3(xcd_grid(dt.n_vtiles)), dim3(BLOCK), time_modes_tile_lds(dt), c->stream, dt, xg, d.phi,
                       d.tp_shift, d.t0, d.nl, d.phi_hi, time_modes_chunk(dt));
    DOTS_HIP(hipGetLastError());
    return 0;
}

// y = K x on node-layout arrays with the coupled space-time operator (tests, operator parity)
int cg_apply_operator(Ctx *c, const double *x, double *y) {
    CgArgs a = make_args(c, false);
    a.zin = x;
    a.out = y;
    hipLaunchKernelGGL((k_cg_apply<false, false>), dim3(a.G), dim3(a.nb), cg_lds_bytes(a.cap, a.vt, a.nb), c->stream, c->d, a);
    DOTS_HIP(hipGetLastError());
    return 0;
}

// Time `reps` launches of one PCG kernel between two hipEvents.
//   which 0: k_cg_apply (fused direction update + operator + p.Ap)     1: k_cg_update     2: one multigrid V-cycle
int cg_bench(Ctx *c, int which, int reps, double *ms, double *bytes) {
    const bool modal = c->lap_solver == DOTS_LAP_MODAL_PCG;
    const Dev &d = modal ? c->dcg : c->d;
    CgArgs a = make_args(c, modal);
    const size_t lds = cg_lds_bytes(a.cap, a.vt, a.nb);
    if (which == 4) {   // one streaming launch of known size (calibrates the PMC traffic counters)
        *ms = 0.0;
        return launch_calibration(c, bytes);
    }
    if (which == 3) {   // both sweeps of the direct solve on whatever the vectors hold
        if (!modal || c->front.n_nodes == 0) {
            set_error("bench: no multifrontal factor on this context");
            return DOTS_ERR_STATE;
        }
        int rcf = 0;
        for (int i = 0; i < 2; ++i) rcf |= front_solve(c, d.cg_p0, d.cg_z, d.cg_x);
        DOTS_HIP(hipEventRecord(c->ev[6], c->stream));
        for (int i = 0; i < reps; ++i) rcf |= front_solve(c, d.cg_p0, d.cg_z, d.cg_x);
        DOTS_HIP(hipEventRecord(c->ev[7], c->stream));
        DOTS_HIP(hipEventSynchronize(c->ev[7]));
        if (rcf) return rcf;
        float tf = 0.f;
        DOTS_HIP(hipEventElapsedTime(&tf, c->ev[6], c->ev[7]));
        *ms = (double)tf / reps;
        *bytes = c->front_bytes_unmerged + 8.0 * (double)d.V * d.cg_ncol * 4.0;   // factor twice (one block per tree node: merged bands read more, dots_front_info); b read, y written + read, x written
        return 0;
    }
    if (which == 2 && !a.mg) {
        set_error("bench: no multigrid hierarchy on this context");
        return DOTS_ERR_STATE;
    }
    // neutral scalars so that the kernels do their full work: no column frozen, beta = 1/2, alpha = 1
    DOTS_HIP(hipMemsetAsync(d.flags, 0, sizeof(int) * FLAG_TOTAL, c->stream));
    DOTS_HIP(hipMemsetAsync(d.cg_p0, 0, sizeof(double) * (size_t)d.V * d.TP, c->stream));
    DOTS_HIP(hipMemsetAsync(d.cg_p1, 0, sizeof(double) * (size_t)d.V * d.TP, c->stream));
    const int64_t npart = cg_partials_needed(d);
    std::vector<double> ones((size_t)npart, 0.0);
    for (int k = 0; k < a.nc; ++k) {   // workgroup 0 carries the totals: r.z = crit = 1 (both parities), p.Ap = 2
        for (int64_t base : {part_rz(a, 0), part_rz(a, 1), part_crit(a, 0), part_crit(a, 1), coll_at(a, idx_rz(0)),
                             coll_at(a, idx_rz(1)), coll_at(a, idx_crit(a, 0)), coll_at(a, idx_crit(a, 1))})
            ones[(size_t)(base + k)] = 1.0;
        ones[(size_t)(part_pap(a) + k)] = 2.0;
        ones[(size_t)(coll_at(a, IDX_PAP) + k)] = 2.0;
    }
    std::vector<double> sc(S::TOTAL, 0.0);
    for (int k = 0; k < S::NCMAX; ++k) {
        sc[SC_RZ0 + k] = 2.0;
        sc[SC_RZ1 + k] = 2.0;
        sc[S::BREF + k] = 1e30;
    }
    DOTS_HIP(hipMemcpyAsync(d.scal, sc.data(), sizeof(double) * S::TOTAL, hipMemcpyHostToDevice, c->stream));
    DOTS_HIP(hipMemcpyAsync(d.partials, ones.data(), sizeof(double) * (size_t)npart, hipMemcpyHostToDevice, c->stream));
    DOTS_HIP(hipStreamSynchronize(c->stream));
    a.tol2 = 0.0;   // nothing freezes
    a.zin = d.cg_z;
    a.out = d.cg_Ap;
    a.x = d.cg_x;
    int rc_launch = 0;
    auto launch = [&](int i) {
        // the parity stays 0 so that the neutral r.z partial sums the apply kernel reads are never overwritten
        a.parity = 0;
        a.p_old = (i & 1) ? d.cg_p1 : d.cg_p0;
        a.p_new = (i & 1) ? d.cg_p0 : d.cg_p1;
        if (which == 2) {
            rc_launch |= mg_vcycle(c, d.cg_r, d.cg_z, d.cg_Ap, d.partials + part_rz(a, 1), a.nb, a.ept, a.vt, a.G);
        } else if (which == 1) {
            if (modal) hipLaunchKernelGGL((k_cg_update<true>), dim3(a.G), dim3(a.nb), lds, c->stream, d, a);
            else hipLaunchKernelGGL((k_cg_update<false>), dim3(a.G), dim3(a.nb), lds, c->stream, d, a);
        } else {
            if (modal) hipLaunchKernelGGL((k_cg_apply<true, true>), dim3(a.G), dim3(a.nb), lds, c->stream, d, a);
            else hipLaunchKernelGGL((k_cg_apply<false, true>), dim3(a.G), dim3(a.nb), lds, c->stream, d, a);
        }
    };
    for (int i = 0; i < 2; ++i) launch(i);
    DOTS_HIP(hipMemcpyAsync(d.partials, ones.data(), sizeof(double) * (size_t)npart, hipMemcpyHostToDevice, c->stream));
    DOTS_HIP(hipEventRecord(c->ev[6], c->stream));
    for (int i = 0; i < reps; ++i) launch(i);
    DOTS_HIP(hipEventRecord(c->ev[7], c->stream));
    DOTS_HIP(hipEventSynchronize(c->ev[7]));
    DOTS_HIP(hipGetLastError());
    if (rc_launch) return rc_launch;
    float t = 0.f;
    DOTS_HIP(hipEventElapsedTime(&t, c->ev[6], c->ev[7]));
    *ms = (double)t / reps;
    const double N = (double)d.V * (modal ? d.cg_ncol : d.T + 1);
    if (which == 2) *bytes = 8.0 * N * 8.0;                                        // fine level: 8 vector passes
    else if (which == 1) *bytes = 8.0 * N * (a.mg ? 6.0 : 7.0) + 16.0 * d.V;       // x,p,r,Ap read; x,r(,z) written
    else *bytes = 12.0 * c->nnz + 4.0 * (d.V + 1) + 16.0 * d.V + 8.0 * N * 4.0;    // CSR + z,p_old read; p_new,Ap written
    return 0;
}

void preload_transform_kernels() {      // (see preload_alm_kernels)
    const void *fns[] = {(const void *)k_time_modes_tile<true, BLOCK>, (const void *)k_time_modes_tile<false, BLOCK>,
                         (const void *)k_time_modes_tile<false, 1024>, (const void *)k_time_modes_mfma,
                         (const void *)k_time_modes<true>, (const void *)k_time_modes<false>};
    hipFuncAttributes a;
    for (const void *f : fns) (void)hipFuncGetAttributes(&a, f);
    (void)hipGetLastError();
}

}  // namespace dots
