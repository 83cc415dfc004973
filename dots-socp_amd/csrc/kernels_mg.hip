// Multigrid V-cycle preconditioner for the batched modal PCG (all time modes at once).
//
// Every time mode a solves  A_a = K + (sigma_a + eps) M  on the surface; the hierarchy (smoothed
// aggregation on K, built on the host by dots-socp_amd/multigrid.py) is shared by all modes and each
// level stores K_l and M_l on one sparsity pattern, so a level operator is applied to every mode in
// one sparse-matrix x dense-block product with the mode index fastest in memory: a gathered
// neighbour row is one contiguous run of doubles, exactly as in the PCG operator.
//
// V(1,1) cycle with damped Jacobi, written so that each level costs four kernels:
//   down     r  = b - A (w D^-1 b)                 (pre-smoothing from a zero guess folded into the residual)
//   restrict b' = P^T r
//   up       x  = w D^-1 b + P x'                  (re-forms the pre-smoothed iterate, adds the correction)
//   post     z  = x + w D^-1 (b - A x)             (on the finest level also emits the r.z partial sums
//                                                   that the next PCG kernel re-reduces)
// and the coarsest level is a dense per-mode inverse.  Frozen (converged) modes are skipped.
// The cycle is symmetric (same smoother before and after, P^T restriction), as PCG requires.
#include "dots_dev.h"

namespace dots {

constexpr int MG_NB = 256;

struct MgArgs {
    double eps, omega;
    int ncol;          // active columns (time modes)
};

__device__ __forceinline__ double mg_shift(const Dev &d, const MgArgs &a, int c) { return d.sigma[c] + a.eps; }

// sum_j A[i,j] * f(j)   with A = K + s M on the level's pattern (level 0: M is the diagonal mass, no vM)
template <typename F>
__device__ __forceinline__ double mg_row(const MgLevelDev &L, int i, double s, F f) {
    double sum = 0.0;
    const int j0 = L.rp[i], j1 = L.rp[i + 1];
    if (L.vM) {
        for (int j = j0; j < j1; ++j) sum += (L.vK[j] + s * L.vM[j]) * f(L.col[j]);
    } else {
        for (int j = j0; j < j1; ++j) sum += L.vK[j] * f(L.col[j]);
        sum += s * L.dM[i] * f(i);
    }
    return sum;
}

#define MG_THREAD_SETUP(nrows)                                                        \
    const int64_t e = (int64_t)blockIdx.x * MG_NB + threadIdx.x;                       \
    const int i = (int)(e >> d.tp_shift), c = (int)(e & (d.TP - 1));                   \
    if (i >= (nrows) || c >= a.ncol) return;                                           \
    if (d.flags[c]) return;

__global__ __launch_bounds__(MG_NB) void k_mg_down(Dev d, MgLevelDev L, MgArgs a, const double *__restrict__ b, double *__restrict__ r) {
    MG_THREAD_SETUP(L.n)
    const double s = mg_shift(d, a, c);
    const double Ax = mg_row(L, i, s, [&](int u) { return a.omega * b[(u << d.tp_shift) + c] / (L.dK[u] + s * L.dM[u]); });
    r[(i << d.tp_shift) + c] = b[(i << d.tp_shift) + c] - Ax;
}

__global__ __launch_bounds__(MG_NB) void k_mg_restrict(Dev d, MgLevelDev L, MgArgs a, const double *__restrict__ r, double *__restrict__ bc) {
    MG_THREAD_SETUP(L.nc)
    double sum = 0.0;
    for (int j = L.r_rp[i]; j < L.r_rp[i + 1]; ++j) sum += L.r_val[j] * r[(L.r_col[j] << d.tp_shift) + c];
    bc[(i << d.tp_shift) + c] = sum;
}

__global__ __launch_bounds__(MG_NB) void k_mg_up(Dev d, MgLevelDev L, MgArgs a, const double *__restrict__ b, const double *__restrict__ xc,
                                               double *__restrict__ x) {
    MG_THREAD_SETUP(L.n)
    const double s = mg_shift(d, a, c);
    double sum = a.omega * b[(i << d.tp_shift) + c] / (L.dK[i] + s * L.dM[i]);
    for (int j = L.p_rp[i]; j < L.p_rp[i + 1]; ++j) sum += L.p_val[j] * xc[(L.p_col[j] << d.tp_shift) + c];
    x[(i << d.tp_shift) + c] = sum;
}

__global__ __launch_bounds__(MG_NB) void k_mg_post(Dev d, MgLevelDev L, MgArgs a, const double *__restrict__ b, const double *__restrict__ x,
                                                 double *__restrict__ z) {
    MG_THREAD_SETUP(L.n)
    const double s = mg_shift(d, a, c);
    const double Ax = mg_row(L, i, s, [&](int u) { return x[(u << d.tp_shift) + c]; });
    const int iv = (i << d.tp_shift) + c;
    z[iv] = x[iv] + a.omega * (b[iv] - Ax) / (L.dK[i] + s * L.dM[i]);
}

// dense per-mode solve on the coarsest level: x[i][c] = sum_j inv[i][j][c] b[j][c]
__global__ __launch_bounds__(MG_NB) void k_mg_coarse(Dev d, MgArgs a, int n, const double *__restrict__ inv, const double *__restrict__ b,
                                                   double *__restrict__ x) {
    MG_THREAD_SETUP(n)
    double sum = 0.0;
    for (int j = 0; j < n; ++j) sum += inv[(((int64_t)i * n + j) << d.tp_shift) + c] * b[(j << d.tp_shift) + c];
    x[(i << d.tp_shift) + c] = sum;
}

// Finest-level post-smoothing on the PCG's own tiling (so that the r.z partial sums land where the
// next k_cg_apply expects them): z = x + w D^-1 (r - A x), partial sums of r.z per workgroup and column.
__global__ __launch_bounds__(1024) void k_mg_post_fine(Dev d, MgLevelDev L, MgArgs a, const double *__restrict__ b, const double *__restrict__ x,
                                                       double *__restrict__ z, double *__restrict__ part, int ept, int vt) {
    __shared__ double red[1024];
    const int tid = threadIdx.x;
    const int c = tid & (d.TP - 1);
    const int tile = xcd_tile(blockIdx.x, gridDim.x);
    double acc = 0.0;
    const bool live = c < a.ncol && !d.flags[c];
    if (live) {
        const double s = mg_shift(d, a, c);
        for (int q = 0; q < ept; ++q) {
            const int el = tid + q * 1024;
            const int vl = el >> d.tp_shift;
            const int i = tile * vt + vl;
            if (vl >= vt || i >= L.n) continue;
            const double Ax = mg_row(L, i, s, [&](int u) { return x[(u << d.tp_shift) + c]; });
            const int iv = (i << d.tp_shift) + c;
            const double zi = x[iv] + a.omega * (b[iv] - Ax) / (L.dK[i] + s * L.dM[i]);
            z[iv] = zi;
            acc += b[iv] * zi;
        }
    }
    red[tid] = acc;
    __syncthreads();
    const int j = tid >> d.tp_shift, J = 1024 >> d.tp_shift;
    if (j == 0 && c < a.ncol) {
        double t = 0.0;
        for (int k = 0; k < J; ++k) t += red[c + (k << d.tp_shift)];
        part[((int64_t)blockIdx.x << d.tp_shift) + c] = t;
    }
}

static inline int mg_grid(const Dev &d, int rows) { return (int)((((int64_t)rows << d.tp_shift) + MG_NB - 1) / MG_NB); }

// Enqueue one V-cycle: z = MG(r) for every live column; r.z partial sums go to `rz_part`.
int mg_vcycle(Ctx *c, const double *r, double *z, double *rz_part, int ept, int vt, int G) {
    const Dev &d = c->d;
    const MgDev &m = c->mg;
    MgArgs a{c->prm.eps, m.omega, d.cg_ncol};
    const int nl = m.nlev;
    const double *b = r;
    // down sweep
    for (int l = 0; l + 1 < nl; ++l) {
        const MgLevelDev &L = m.lv[l];
        hipLaunchKernelGGL(k_mg_down, dim3(mg_grid(d, L.n)), dim3(MG_NB), 0, c->stream, d, L, a, b, L.r);
        hipLaunchKernelGGL(k_mg_restrict, dim3(mg_grid(d, L.nc)), dim3(MG_NB), 0, c->stream, d, L, a, L.r, m.lv[l + 1].b);
        b = m.lv[l + 1].b;
    }
    // coarsest
    {
        const MgLevelDev &L = m.lv[nl - 1];
        hipLaunchKernelGGL(k_mg_coarse, dim3(mg_grid(d, L.n)), dim3(MG_NB), 0, c->stream, d, a, L.n, m.coarse_inv, L.b, L.x);
    }
    // up sweep
    for (int l = nl - 2; l >= 0; --l) {
        const MgLevelDev &L = m.lv[l];
        const double *bl = (l == 0) ? r : L.b;
        const double *xc = (l + 1 == nl - 1) ? m.lv[l + 1].x : m.lv[l + 1].x2;
        hipLaunchKernelGGL(k_mg_up, dim3(mg_grid(d, L.n)), dim3(MG_NB), 0, c->stream, d, L, a, bl, xc, L.x);
        if (l == 0)
            hipLaunchKernelGGL(k_mg_post_fine, dim3(G), dim3(1024), 0, c->stream, d, L, a, bl, L.x, z, rz_part, ept, vt);
        else
            hipLaunchKernelGGL(k_mg_post, dim3(mg_grid(d, L.n)), dim3(MG_NB), 0, c->stream, d, L, a, bl, L.x, L.x2);
    }
    DOTS_HIP(hipGetLastError());
    return 0;
}

}  // namespace dots
