// Multigrid V-cycle preconditioner for the batched modal PCG (all time modes at once).
//
// Every time mode a solves  A_a = K + (sigma_a + eps) M  on the surface; the hierarchy (smoothed
// aggregation on K, built on the host by dots_socp_amd/multigrid.py) is shared by all modes and each
// level stores K_l and M_l on one sparsity pattern, so a level operator is applied to every mode in
// one sparse-matrix x dense-block product with the mode index fastest in memory: a gathered
// neighbour row is one contiguous run of doubles, exactly as in the PCG operator.
//
// V(1,1) cycle with damped Jacobi (weight w), symmetric as PCG requires, written so that a level
// costs three kernels and a single neighbour gather pass.  With bt = D^-1 b (delivered by the kernel
// that produced b) and the product A P precomputed on the host:
//   down      r   = b - w A bt                                     (the only SpMM on the level's own graph;
//                                                                   pre-smoothing x = w bt folded in: r = b - A x)
//   restrict  b'  = P^T r,   bt' = D'^-1 b'
//   post      z   = w bt + w D^-1 r + sum_j (P_ij - w D^-1 (A P)_ij) x'_j      (x = w bt + P x';  b - A x = r - (A P) x')
// P is stored on the pattern of A P, so post is one pass over ~10 entries of the (much smaller) coarse
// vector and otherwise reads only its own entries of bt and r: z may overwrite bt in place.
// On the finest level post also emits the r.z partial sums that the next PCG kernel re-reduces.
//
// The coarse tail.  The smallest levels (at most Ctx::mg_tail_rows rows, default 256) are pure launch
// latency.  The time-mode columns are independent of each other, so ONE kernel (k_mg_tail) walks those
// levels -- down, restrict, ..., dense coarsest solve, ..., post -- with one workgroup per column and
// workgroup barriers between the phases.  Measured on MI355X (us per V-cycle, sphere10k / torus100k):
// tail of <= 64 rows 71.2 / 207.8, <= 256 rows 67.0 / 206.4, <= 2048 rows 93.7 / 286.0: a single
// workgroup per column starves once a level has more than a few hundred rows, so only the last two
// levels go into the tail.
// Frozen (converged) modes are skipped everywhere.
#include "dots_dev.h"

namespace dots {

constexpr int MG_NB = 256;
constexpr int MG_TAIL_NB = 1024;

struct MgArgs {
    double eps, omega;
    int ncol;          // active columns (time modes)
};

__device__ __forceinline__ double mg_shift(const Dev &d, const MgArgs &a, int c) { return d.sigma[c] + a.eps; }

// ---- per-element work of the three level kernels -----------------------------------------------
// r_i = b_i - w (A bt)_i  with A = K + s M on the level's pattern (level 0: M is the diagonal mass, vM == nullptr)
__device__ __forceinline__ void mg_down_elem(const Dev &d, const MgLevelDev &L, const MgArgs &a, const double *b, const double *bt, double *r,
                                             int i, int c) {
    const double s = mg_shift(d, a, c);
    const int sh = d.tp_shift;
    double sum = 0.0;
    int j = L.rp[i];
    const int j1 = L.rp[i + 1];
    if (L.vM) {
        for (; j + 4 <= j1; j += 4) {
            const double x0 = bt[(L.col[j] << sh) + c], x1 = bt[(L.col[j + 1] << sh) + c], x2 = bt[(L.col[j + 2] << sh) + c],
                         x3 = bt[(L.col[j + 3] << sh) + c];
            sum += ((L.vK[j] + s * L.vM[j]) * x0 + (L.vK[j + 1] + s * L.vM[j + 1]) * x1) +
                   ((L.vK[j + 2] + s * L.vM[j + 2]) * x2 + (L.vK[j + 3] + s * L.vM[j + 3]) * x3);
        }
        for (; j < j1; ++j) sum += (L.vK[j] + s * L.vM[j]) * bt[(L.col[j] << sh) + c];
    } else {
        for (; j + 4 <= j1; j += 4) {
            const double x0 = bt[(L.col[j] << sh) + c], x1 = bt[(L.col[j + 1] << sh) + c], x2 = bt[(L.col[j + 2] << sh) + c],
                         x3 = bt[(L.col[j + 3] << sh) + c];
            sum += (L.vK[j] * x0 + L.vK[j + 1] * x1) + (L.vK[j + 2] * x2 + L.vK[j + 3] * x3);
        }
        for (; j < j1; ++j) sum += L.vK[j] * bt[(L.col[j] << sh) + c];
        sum += s * L.dM[i] * bt[(i << sh) + c];
    }
    r[(i << sh) + c] = b[(i << sh) + c] - a.omega * sum;
}

// b'_i = (R r)_i,  bt'_i = b'_i / diag(A')_i   for the next coarser level (dKc, dMc: its diagonals)
__device__ __forceinline__ void mg_restrict_elem(const Dev &d, const MgLevelDev &L, const MgArgs &a, const double *r, const double *dKc,
                                                 const double *dMc, double *bc, double *btc, int i, int c) {
    const int sh = d.tp_shift;
    double sum = 0.0;
    int j = L.r_rp[i];
    const int j1 = L.r_rp[i + 1];
    for (; j + 4 <= j1; j += 4) {
        const double r0 = r[(L.r_col[j] << sh) + c], r1 = r[(L.r_col[j + 1] << sh) + c], r2 = r[(L.r_col[j + 2] << sh) + c],
                     r3 = r[(L.r_col[j + 3] << sh) + c];
        sum += (L.r_val[j] * r0 + L.r_val[j + 1] * r1) + (L.r_val[j + 2] * r2 + L.r_val[j + 3] * r3);
    }
    for (; j < j1; ++j) sum += L.r_val[j] * r[(L.r_col[j] << sh) + c];
    const int ic = (i << sh) + c;
    bc[ic] = sum;
    btc[ic] = sum / (dKc[i] + mg_shift(d, a, c) * dMc[i]);
}

// z_i = w bt_i + w dinv r_i + sum_j (P_ij - w dinv (A P)_ij) x'_j      (P aligned to the pattern of A P)
__device__ __forceinline__ double mg_post_value(const Dev &d, const MgLevelDev &L, const MgArgs &a, const double *bt, const double *r,
                                                const double *xc, int i, int c) {
    const int sh = d.tp_shift;
    const double s = mg_shift(d, a, c);
    const double wdinv = a.omega / (L.dK[i] + s * L.dM[i]);
    double q = 0.0;
    int j = L.ap_rp[i];
    const int j1 = L.ap_rp[i + 1];
    for (; j + 4 <= j1; j += 4) {
        const double x0 = xc[(L.ap_col[j] << sh) + c], x1 = xc[(L.ap_col[j + 1] << sh) + c], x2 = xc[(L.ap_col[j + 2] << sh) + c],
                     x3 = xc[(L.ap_col[j + 3] << sh) + c];
        q += ((L.ap_vP[j] - wdinv * (L.ap_vK[j] + s * L.ap_vM[j])) * x0 + (L.ap_vP[j + 1] - wdinv * (L.ap_vK[j + 1] + s * L.ap_vM[j + 1])) * x1) +
             ((L.ap_vP[j + 2] - wdinv * (L.ap_vK[j + 2] + s * L.ap_vM[j + 2])) * x2 + (L.ap_vP[j + 3] - wdinv * (L.ap_vK[j + 3] + s * L.ap_vM[j + 3])) * x3);
    }
    for (; j < j1; ++j) q += (L.ap_vP[j] - wdinv * (L.ap_vK[j] + s * L.ap_vM[j])) * xc[(L.ap_col[j] << sh) + c];
    const int iv = (i << sh) + c;
    return a.omega * bt[iv] + wdinv * r[iv] + q;
}

// ---- one kernel per phase (levels above the tail) --------------------------------------------------
#define MG_THREAD_SETUP(nrows)                                                        \
    const int64_t e = (int64_t)blockIdx.x * MG_NB + threadIdx.x;                       \
    const int i = (int)(e >> d.tp_shift), c = (int)(e & (d.TP - 1));                   \
    if (i >= (nrows) || c >= a.ncol) return;                                           \
    if (d.flags[c]) return;

__global__ __launch_bounds__(MG_NB) void k_mg_down(Dev d, MgLevelDev L, MgArgs a, const double *__restrict__ b, const double *__restrict__ bt,
                                                 double *__restrict__ r) {
    MG_THREAD_SETUP(L.n)
    mg_down_elem(d, L, a, b, bt, r, i, c);
}

__global__ __launch_bounds__(MG_NB) void k_mg_restrict(Dev d, MgLevelDev L, MgArgs a, const double *__restrict__ r, const double *__restrict__ dKc,
                                                     const double *__restrict__ dMc, double *__restrict__ bc, double *__restrict__ btc) {
    MG_THREAD_SETUP(L.nc)
    mg_restrict_elem(d, L, a, r, dKc, dMc, bc, btc, i, c);
}

// restriction onto a SMALL coarse level: a workgroup per coarse row, its ~20-60 entries split over the lanes that do not index the column
// (one thread per (row, column) leaves a level of 200-2 000 rows with 26-250 workgroups each walking a whole row: 9.5-11 us at torus100k)
__global__ __launch_bounds__(MG_NB) void k_mg_restrict_rows(Dev d, MgLevelDev L, MgArgs a, const double *__restrict__ r, const double *__restrict__ dKc,
                                                          const double *__restrict__ dMc, double *__restrict__ bc, double *__restrict__ btc) {
    __shared__ double red[MG_NB];
    const int i = blockIdx.x, tid = threadIdx.x, sh = d.tp_shift;
    const int c = tid & (d.TP - 1), q = tid >> sh, Q = MG_NB >> sh;
    const bool live = c < a.ncol && !d.flags[c];
    double s = 0.0;
    if (live)
        for (int j = L.r_rp[i] + q; j < L.r_rp[i + 1]; j += Q) s += L.r_val[j] * r[(L.r_col[j] << sh) + c];
    red[tid] = s;
    __syncthreads();
    if (q == 0 && live) {
        double t = 0.0;
        for (int k = 0; k < Q; ++k) t += red[c + (k << sh)];
        const int ic = (i << sh) + c;
        bc[ic] = t;
        btc[ic] = t / (dKc[i] + mg_shift(d, a, c) * dMc[i]);
    }
}

__global__ __launch_bounds__(MG_NB) void k_mg_post(Dev d, MgLevelDev L, MgArgs a, const double *bt, const double *__restrict__ r,
                                                 const double *__restrict__ xc, double *z) {
    MG_THREAD_SETUP(L.n)
    z[(i << d.tp_shift) + c] = mg_post_value(d, L, a, bt, r, xc, i, c);
}

// dense per-mode solve on the coarsest level, one thread per (row, column): x[i][c] = sum_j inv[i][j][c] b[j][c]
// (loads coalesce over the column index; used when the coarsest level is too large for the tail kernel)
__global__ __launch_bounds__(MG_NB) void k_mg_coarse(Dev d, MgArgs a, int n, const double *__restrict__ inv, const double *__restrict__ b,
                                                   double *__restrict__ x) {
    MG_THREAD_SETUP(n)
    const int sh = d.tp_shift;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    const double *row = inv + (((int64_t)i * n) << sh) + c;
    int j = 0;
    for (; j + 4 <= n; j += 4) {
        s0 += row[(int64_t)j << sh] * b[(j << sh) + c];
        s1 += row[(int64_t)(j + 1) << sh] * b[((j + 1) << sh) + c];
        s2 += row[(int64_t)(j + 2) << sh] * b[((j + 2) << sh) + c];
        s3 += row[(int64_t)(j + 3) << sh] * b[((j + 3) << sh) + c];
    }
    for (; j < n; ++j) s0 += row[(int64_t)j << sh] * b[(j << sh) + c];
    x[(i << sh) + c] = (s0 + s1) + (s2 + s3);
}

// the same with a workgroup per row, its dot product split over the lanes that do not index the column: a coarsest level of ~200 rows
// is 26 workgroups of the kernel above (22 us at torus100k, each thread walking 208 entries) and ~200 of this one
__global__ __launch_bounds__(MG_NB) void k_mg_coarse_rows(Dev d, MgArgs a, int n, const double *__restrict__ inv, const double *__restrict__ b,
                                                        double *__restrict__ x) {
    __shared__ double red[MG_NB];
    const int i = blockIdx.x, tid = threadIdx.x, sh = d.tp_shift;
    const int c = tid & (d.TP - 1), q = tid >> sh, Q = MG_NB >> sh;
    const bool live = c < a.ncol && !d.flags[c];
    double s = 0.0;
    if (live) {
        const double *row = inv + (((int64_t)i * n) << sh) + c;
        for (int j = q; j < n; j += Q) s += row[(int64_t)j << sh] * b[(j << sh) + c];
    }
    red[tid] = s;
    __syncthreads();
    if (q == 0 && live) {
        double t = 0.0;
        for (int k = 0; k < Q; ++k) t += red[c + (k << sh)];
        x[(i << sh) + c] = t;
    }
}

// ---- the coarse tail in one launch: one workgroup per time-mode column ------------------------------
struct MgTail {
    int first, nlev;            // levels [first, nlev) of the hierarchy
    MgLevelDev lv[8];           // lv[k] = level first + k
    const double *coarse_inv;   // [nL][nL][TP]
};

__global__ __launch_bounds__(MG_TAIL_NB) void k_mg_tail(Dev d, MgTail T, MgArgs a) {
    const int c = blockIdx.x;
    if (c >= a.ncol || d.flags[c]) return;
    const int tid = threadIdx.x, sh = d.tp_shift;
    const int nk = T.nlev - T.first;
    for (int k = 0; k + 1 < nk; ++k) {
        const MgLevelDev &L = T.lv[k];
        const MgLevelDev &C = T.lv[k + 1];
        for (int i = tid; i < L.n; i += MG_TAIL_NB) mg_down_elem(d, L, a, L.b, L.bt, L.t, i, c);
        __syncthreads();
        for (int i = tid; i < L.nc; i += MG_TAIL_NB) mg_restrict_elem(d, L, a, L.t, C.dK, C.dM, C.b, C.bt, i, c);
        __syncthreads();
    }
    {   // dense per-mode solve on the coarsest level; the solution lands in its bt slot
        const MgLevelDev &L = T.lv[nk - 1];
        const int n = L.n;
        for (int i = tid; i < n; i += MG_TAIL_NB) {
            double sum = 0.0;
            for (int j = 0; j < n; ++j) sum += T.coarse_inv[(((int64_t)i * n + j) << sh) + c] * L.b[(j << sh) + c];
            L.bt[(i << sh) + c] = sum;
        }
        __syncthreads();
    }
    for (int k = nk - 2; k >= 0; --k) {
        const MgLevelDev &L = T.lv[k];
        const double *xc = T.lv[k + 1].bt;
        for (int i = tid; i < L.n; i += MG_TAIL_NB) {
            const double z = mg_post_value(d, L, a, L.bt, L.t, xc, i, c);
            L.bt[(i << sh) + c] = z;     // own entry only: in place is safe
        }
        __syncthreads();
    }
}

// Finest-level post-smoothing on the PCG's own tiling (so that the r.z partial sums land where the
// next k_cg_apply expects them).  z overwrites bt in place.  Workgroup size = the PCG's (blockDim.x).
__global__ __launch_bounds__(1024) void k_mg_post_fine(Dev d, MgLevelDev L, MgArgs a, const double *__restrict__ b, const double *bt,
                                                       const double *__restrict__ r, const double *__restrict__ xc, double *z,
                                                       double *__restrict__ part, int ept, int vt) {
    __shared__ double red[1024];
    const int tid = threadIdx.x, NB = blockDim.x;
    const int c = tid & (d.TP - 1);
    const int tile = xcd_tile(blockIdx.x, gridDim.x);
    double acc = 0.0;
    const bool live = c < a.ncol && !d.flags[c];
    if (live) {
        for (int q = 0; q < ept; ++q) {
            const int el = tid + q * NB;
            const int vl = el >> d.tp_shift;
            const int i = tile * vt + vl;
            if (vl >= vt || i >= L.n) continue;
            const int iv = (i << d.tp_shift) + c;
            const double zi = mg_post_value(d, L, a, bt, r, xc, i, c);
            z[iv] = zi;
            acc += b[iv] * zi;     // r.z with the PCG residual b (the level-0 right-hand side)
        }
    }
    if (d.TP <= 64) {   // lanes sharing a column are TP apart inside a wave: fold by shuffles, one barrier
        double sacc = acc;
        for (int o = 32; o >= d.TP; o >>= 1) sacc += __shfl_xor(sacc, o, 64);
        const int lane = tid & 63, wv = tid >> 6, nw = NB >> 6;
        if (lane < d.TP) red[wv * d.TP + lane] = sacc;
        __syncthreads();
        if (tid < a.ncol) {
            double tsum = 0.0;
            for (int k = 0; k < nw; ++k) tsum += red[k * d.TP + tid];
            part[((int64_t)blockIdx.x << d.tp_shift) + tid] = tsum;
        }
    } else {
        red[tid] = acc;
        __syncthreads();
        const int j = tid >> d.tp_shift, J = NB >> d.tp_shift;
        if (j == 0 && c < a.ncol) {
            double tsum = 0.0;
            for (int k = 0; k < J; ++k) tsum += red[c + (k << d.tp_shift)];
            part[((int64_t)blockIdx.x << d.tp_shift) + c] = tsum;
        }
    }
}

static inline int mg_grid(const Dev &d, int rows) { return (int)((((int64_t)rows << d.tp_shift) + MG_NB - 1) / MG_NB); }

// Enqueue one V-cycle.  r: residual; z: holds D^-1 r on entry (written by the PCG update kernel) and the
// preconditioned residual on exit; r.z partial sums go to `rz_part`.  Level 0 uses `t0` as scratch.
int mg_vcycle(Ctx *c, const double *r, double *z, double *t0, double *rz_part, int nb, int ept, int vt, int G) {
    const Dev &d = c->dcg;   // the PCG's view: its own column range, pitch and sigma
    const MgDev &m = c->mg;
    MgArgs a{c->prm.eps, m.omega, d.cg_ncol};
    const int nl = m.nlev;
    // first level of the tail: the first level >= 1 with few enough rows (always at least the coarsest)
    int first_tail = nl - 1;
    for (int l = 1; l < nl; ++l)
        if (m.lv[l].n <= c->mg_tail_rows && nl - l <= 8) { first_tail = l; break; }
    // down sweep above the tail
    for (int l = 0; l < first_tail; ++l) {
        const MgLevelDev &L = m.lv[l];
        const MgLevelDev &C = m.lv[l + 1];
        const double *b = (l == 0) ? r : L.b;
        const double *bt = (l == 0) ? z : L.bt;
        double *t = (l == 0) ? t0 : L.t;
        hipLaunchKernelGGL(k_mg_down, dim3(mg_grid(d, L.n)), dim3(MG_NB), 0, c->stream, d, L, a, b, bt, t);
        if (L.nc <= 4096 && d.TP <= MG_NB / 2) hipLaunchKernelGGL(k_mg_restrict_rows, dim3(L.nc), dim3(MG_NB), 0, c->stream, d, L, a, t, C.dK, C.dM, C.b, C.bt);
        else hipLaunchKernelGGL(k_mg_restrict, dim3(mg_grid(d, L.nc)), dim3(MG_NB), 0, c->stream, d, L, a, t, C.dK, C.dM, C.b, C.bt);
    }
    // the tail (at least the dense coarsest solve): one workgroup per column; a large coarsest level on
    // its own is solved by the flat kernel instead (one thread per entry, coalesced over the columns)
    if (first_tail == nl - 1 && m.lv[nl - 1].n > 64) {
        const MgLevelDev &L = m.lv[nl - 1];
        if (d.TP <= MG_NB / 2) hipLaunchKernelGGL(k_mg_coarse_rows, dim3(L.n), dim3(MG_NB), 0, c->stream, d, a, L.n, m.coarse_inv, L.b, L.bt);
        else hipLaunchKernelGGL(k_mg_coarse, dim3(mg_grid(d, L.n)), dim3(MG_NB), 0, c->stream, d, a, L.n, m.coarse_inv, L.b, L.bt);
    } else {
        MgTail T{};
        T.first = first_tail;
        T.nlev = nl;
        for (int l = first_tail; l < nl; ++l) T.lv[l - first_tail] = m.lv[l];
        T.coarse_inv = m.coarse_inv;
        hipLaunchKernelGGL(k_mg_tail, dim3(d.cg_ncol), dim3(MG_TAIL_NB), 0, c->stream, d, T, a);
    }
    // up sweep above the tail: level l's result is written over its bt
    for (int l = first_tail - 1; l >= 0; --l) {
        const MgLevelDev &L = m.lv[l];
        const double *xc = m.lv[l + 1].bt;
        if (l == 0)
            hipLaunchKernelGGL(k_mg_post_fine, dim3(G), dim3(nb), 0, c->stream, d, L, a, r, z, t0, xc, z, rz_part, ept, vt);
        else
            hipLaunchKernelGGL(k_mg_post, dim3(mg_grid(d, L.n)), dim3(MG_NB), 0, c->stream, d, L, a, L.bt, L.t, xc, L.bt);
    }
    DOTS_HIP(hipGetLastError());
    return 0;
}

}  // namespace dots
