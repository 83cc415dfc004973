// Numeric multifrontal Cholesky factorisation on the device: builds the factor the sweeps of
// kernels_front.hip stream (replaces the T+1 SuperLU factorisations of the reference's setup,
// utils/laplacian_inverse_socp.py:40-44).  One elimination tree for all time modes; every kernel handles all
// modes at once with the mode index fastest, so the work of the T+1 factorisations is one batched pass.
//
// Per tree height (leaves first), all nodes of the height in each launch:
//   assemble   C_p = A[front_p, sep_p] (entries of K + shift * mass, looked up in the CSR) + the children's
//              Schur complements, S_p = the children's Schur complements on bd_p x bd_p        (pull form)
//   for every panel of w columns:
//     panel    Cholesky of the w x w diagonal block in LDS; then L[i, panel] for the rows below it
//     update   C[i][j] -= L[i, panel] . L[j, panel]  on the columns right of the panel, and S_p likewise
//   linv       L_pp^-1 by forward substitution, one (column, mode) per thread
//   gmat       G_p = L_bs L_pp^-1            (= A_bs A_ss^-1)
// after which F_p = [L_pp^-1 ; G_p] is final and S_p waits for the parent.  Deterministic (no atomics).
#include "dots_dev.h"

#include <algorithm>
#include <cstdlib>
#include <vector>

namespace dots {

struct FactArgs {
    int sh, TP, ncol;
    const double *sigma;        // shifts of the context's modes (without eps)
    double eps;
    const int *grounded;        // [TP] 1: the mode's operator is singular, its last root pivot is grounded
    const int *rowptr, *col;    // K (device vertex numbering)
    const double *val, *mass;
    const int *pull0, *pull1;   // front position -> row in the child's boundary, or -1
    double *C, *T, *S;          // work copy of the fronts (becomes L), final factor, Schur complements
    const int *level_nodes;     // nodes of the current height
    int *bad_pivot;             // set to 1 when a pivot is not positive (operator not positive definite)
};

__device__ __forceinline__ int front_vertex(const FrontDev &f, const FrontNode &nd, int i) {
    return i < nd.n ? (int)(f.vmap ? f.vmap[nd.k0 + i] : nd.k0 + i) : f.bd_vertex[nd.bdoff + (i - nd.n)];
}

// ---- assemble -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fact_assemble(FactArgs g, FrontDev f) {
    const FrontNode nd = f.nodes[g.level_nodes[blockIdx.y]];
    const int sh = g.sh, tid = threadIdx.x;
    const int a = tid & (g.TP - 1), q = tid >> sh, Q = 256 >> sh;
    if (a >= g.ncol) return;
    const int n = nd.n, b = nd.b, m = n + b;
    const int64_t nC = (int64_t)m * n, total = nC + (int64_t)b * b;
    const double shift = g.sigma[a] + g.eps;
    const bool ground = nd.parent < 0 && g.grounded[a];
    const FrontNode c0 = nd.c0 >= 0 ? f.nodes[nd.c0] : nd;
    const FrontNode c1 = nd.c1 >= 0 ? f.nodes[nd.c1] : nd;
    for (int64_t e = (int64_t)blockIdx.x * Q + q; e < total; e += (int64_t)gridDim.x * Q) {
        int i, j;
        if (e < nC) { i = (int)(e / n); j = (int)(e % n); }
        else { const int64_t e2 = e - nC; i = n + (int)(e2 / b); j = n + (int)(e2 % b); }
        double v = 0.0;
        if (j < n) {
            const int vj = front_vertex(f, nd, j), vi = front_vertex(f, nd, i);
            for (int k = g.rowptr[vj]; k < g.rowptr[vj + 1]; ++k)
                if (g.col[k] == vi) v = g.val[k];
            if (i == j) v += shift * g.mass[vj];
        }
        if (nd.c0 >= 0) {
            const int ri = g.pull0[nd.ioff + i], rj = g.pull0[nd.ioff + j];
            if (ri >= 0 && rj >= 0) v += g.S[((c0.soff + (int64_t)ri * c0.b + rj) << sh) + a];
        }
        if (nd.c1 >= 0) {
            const int ri = g.pull1[nd.ioff + i], rj = g.pull1[nd.ioff + j];
            if (ri >= 0 && rj >= 0) v += g.S[((c1.soff + (int64_t)ri * c1.b + rj) << sh) + a];
        }
        if (ground && j < n && (i == n - 1 || j == n - 1)) v = (i == j) ? 1.0 : 0.0;
        if (e < nC) g.C[((nd.foff + e) << sh) + a] = v;
        else g.S[((nd.soff + (e - nC)) << sh) + a] = v;
    }
}

// ---- panel: Cholesky of the w x w diagonal block (FACTOR: one workgroup per node, in LDS, written back), then
// L[i, panel] for the rows below it (!FACTOR: grid (row blocks, nodes), every workgroup loads the FACTORED block).
// Two launches, so that no workgroup can read a diagonal block another one is overwriting.
template <bool FACTOR>
__global__ __launch_bounds__(256) void k_fact_panel(FactArgs g, FrontDev f, int k0, int w, int rows_per_wg) {
    extern __shared__ double D[];      // [w][w][TP]
    const FrontNode nd = f.nodes[g.level_nodes[blockIdx.y]];
    const int n = nd.n, m = n + nd.b;
    if (k0 >= n) return;
    const int we = min(w, n - k0);
    const int first = k0 + we + blockIdx.x * rows_per_wg;     // first row of this workgroup below the diagonal block
    if (!FACTOR && first >= m) return;
    const int sh = g.sh, tid = threadIdx.x;
    const int a = tid & (g.TP - 1), q = tid >> sh, Q = 256 >> sh;
    const bool live = a < g.ncol;
    double *__restrict__ Cp = g.C + (nd.foff << sh) + a;
    auto Dd = [&](int i, int j) -> double & { return D[((i * w + j) << sh) + a]; };
    if (live)
        for (int e = q; e < we * we; e += Q) {
            const int i = e / we, j = e % we;
            Dd(i, j) = j <= i ? Cp[((int64_t)(k0 + i) * n + (k0 + j)) << sh] : 0.0;
        }
    __syncthreads();
    if (FACTOR) {
        for (int t = 0; t < we; ++t) {
            const double dtt = live ? Dd(t, t) : 1.0;
            if (q == 0 && !(dtt > 0.0)) *g.bad_pivot = 1;      // also catches NaN; every writer stores the same value
            const double piv = sqrt(dtt);
            __syncthreads();
            if (live) {
                if (q == 0) Dd(t, t) = piv;
                for (int i = t + 1 + q; i < we; i += Q) Dd(i, t) /= piv;
            }
            __syncthreads();
            if (live)
                for (int e = q; e < (we - t - 1) * (we - t - 1); e += Q) {
                    const int i = t + 1 + e / (we - t - 1), j = t + 1 + e % (we - t - 1);
                    if (j <= i) Dd(i, j) -= Dd(i, t) * Dd(j, t);
                }
            __syncthreads();
        }
        if (live)
            for (int e = q; e < we * we; e += Q) {
                const int i = e / we, j = e % we;
                if (j <= i) Cp[((int64_t)(k0 + i) * n + (k0 + j)) << sh] = Dd(i, j);
            }
        return;
    }
    if (!live) return;
    // rows below: L[i][k0 + t] = (C[i][k0 + t] - sum_{s<t} L[i][k0 + s] D[t][s]) / D[t][t]
    for (int i = first + q; i < min(first + rows_per_wg, m); i += Q) {
        double *__restrict__ row = Cp + (((int64_t)i * n + k0) << sh);
        double x[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            if (t < we) {
                double s = row[(int64_t)t << sh];
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    if (u < t) s -= x[u] * Dd(t, u);
                x[t] = s / Dd(t, t);
                row[(int64_t)t << sh] = x[t];
            }
        }
    }
}

// ---- trailing update with the panel [k0, k0 + we): 16 x 16 tiles of the m x m front (columns >= n live in S); the 8 threads
// of a mode take 4 rows x 8 columns each, so that a panel entry loaded once serves 8 (4) products: 12 loads per 32
// multiply-adds.  (An 8 x 8 tiling with one row per thread, 9 loads per 8 multiply-adds, took 2.6 x as long: DESIGN.md 8.)
__global__ __launch_bounds__(256) void k_fact_update(FactArgs g, FrontDev f, int k0, int w, int tiles_per_row) {
    const FrontNode nd = f.nodes[g.level_nodes[blockIdx.y]];
    const int n = nd.n, b = nd.b, m = n + b;
    if (k0 >= n) return;
    const int we = min(w, n - k0);
    const int ti = blockIdx.x / tiles_per_row, tj = blockIdx.x % tiles_per_row;
    const int i0 = ti * 16, j0 = tj * 16;
    if (i0 >= m || j0 >= m) return;
    const int jlo = k0 + we;                    // first column that still changes
    if (j0 + 16 <= jlo) return;
    if (j0 < n) { if (i0 + 16 <= j0) return; }  // strictly above the diagonal of the L part
    else if (i0 + 16 <= n) return;              // S columns only take S rows
    const int sh = g.sh, tid = threadIdx.x;
    const int a = tid & (g.TP - 1);
    if (a >= g.ncol) return;
    const int Q = 256 >> sh;                    // threads per mode
    const double *__restrict__ Cp = g.C + (nd.foff << sh) + a;
    // sub-tiles of 4 rows x 8 columns: 8 per tile, dealt to the mode's threads
    for (int st = tid >> sh; st < 8; st += Q) {
        const int ib = i0 + (st >> 1) * 4, jb = j0 + (st & 1) * 8;
        double acc[4][8];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int cc = 0; cc < 8; ++cc) acc[r][cc] = 0.0;
        const double *ri[4], *rj[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) ri[r] = Cp + (((int64_t)min(ib + r, m - 1) * n + k0) << sh);
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) rj[cc] = Cp + (((int64_t)min(jb + cc, m - 1) * n + k0) << sh);
        for (int t = 0; t < we; ++t) {
            double li[4], lj[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) li[r] = ri[r][(int64_t)t << sh];
#pragma unroll
            for (int cc = 0; cc < 8; ++cc) lj[cc] = rj[cc][(int64_t)t << sh];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int cc = 0; cc < 8; ++cc) acc[r][cc] += li[r] * lj[cc];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = ib + r;
            if (i >= m) continue;
#pragma unroll
            for (int cc = 0; cc < 8; ++cc) {
                const int j = jb + cc;
                if (!(j < m && j >= jlo && (j < n ? i >= j : i >= n))) continue;
                if (j < n) g.C[((nd.foff + (int64_t)i * n + j) << sh) + a] -= acc[r][cc];
                else g.S[((nd.soff + (int64_t)(i - n) * b + (j - n)) << sh) + a] -= acc[r][cc];
            }
        }
    }
}

// ---- L^-1, one (column, mode) per thread: x_j = 1 / L_jj ;  x_i = -(sum_{k=j}^{i-1} L_ik x_k) / L_ii ---------
__global__ __launch_bounds__(256) void k_fact_linv(FactArgs g, FrontDev f) {
    const FrontNode nd = f.nodes[g.level_nodes[blockIdx.y]];
    const int n = nd.n;
    const int sh = g.sh, tid = threadIdx.x;
    const int a = tid & (g.TP - 1), Q = 256 >> sh;
    const int j = blockIdx.x * Q + (tid >> sh);
    if (j >= n || a >= g.ncol) return;
    const double *__restrict__ L = g.C + (nd.foff << sh) + a;
    double *__restrict__ X = g.T + (nd.foff << sh) + a;
    const bool ground = nd.parent < 0 && g.grounded[a];
    for (int i = j; i < n; ++i) {
        const double *__restrict__ Li = L + (((int64_t)i * n) << sh);
        // 8 partial sums: the loop is a chain of L2 round trips, 16 loads in flight per lane instead of 4
        double s[8] = {(i == j) ? 1.0 : 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        int k = j;
        for (; k + 8 <= i; k += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) s[u] -= Li[(int64_t)(k + u) << sh] * X[((int64_t)(k + u) * n + j) << sh];
        }
        for (; k < i; ++k) s[0] -= Li[(int64_t)k << sh] * X[((int64_t)k * n + j) << sh];
        double x = (((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]))) / Li[(int64_t)i << sh];
        if (ground && i == n - 1) x = 0.0;
        X[((int64_t)i * n + j) << sh] = x;
    }
}

// ---- G = L_bs L^-1 (rows n.. of the final block): G[r][j] = sum_{k>=j} L_bs[r][k] Linv[k][j] ------------------
__global__ __launch_bounds__(256) void k_fact_gmat(FactArgs g, FrontDev f) {
    const FrontNode nd = f.nodes[g.level_nodes[blockIdx.y]];
    const int n = nd.n, b = nd.b;
    const int sh = g.sh, tid = threadIdx.x;
    const int a = tid & (g.TP - 1), q = tid >> sh, Q = 256 >> sh;
    if (a >= g.ncol) return;
    const double *__restrict__ L = g.C + (nd.foff << sh) + a;
    double *__restrict__ X = g.T + (nd.foff << sh) + a;
    const int64_t total = (int64_t)b * n;
    for (int64_t e = (int64_t)blockIdx.x * Q + q; e < total; e += (int64_t)gridDim.x * Q) {
        const int r = (int)(e / n), j = (int)(e % n);
        const double *__restrict__ Lr = L + (((int64_t)(n + r) * n) << sh);
        double s[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        int k = j;
        for (; k + 8 <= n; k += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) s[u] += Lr[(int64_t)(k + u) << sh] * X[((int64_t)(k + u) * n + j) << sh];
        }
        for (; k < n; ++k) s[0] += Lr[(int64_t)k << sh] * X[((int64_t)k * n + j) << sh];
        X[((int64_t)(n + r) * n + j) << sh] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    }
}

// Factorise on the device.  `f` holds the tree (nodes, vmap, bd_vertex already uploaded); T (zeroed,
// n_entries << sh doubles) receives the factor.  Host arrays: per-level node lists and the pull maps.
int front_factorize(Ctx *c, const dots_front_desc *h, FrontDev &f, const std::vector<FrontNode> &nodes, double *T, const int *grounded_host) {
    const Dev &d = c->dcg;
    const int sh = d.tp_shift;
    int rc = 0;
    std::vector<void *> tmp;
    auto release = [&]() { for (void *p : tmp) (void)hipFree(p); };
    auto dalloc = [&](void **out, size_t bytes, const void *host) -> int {
        void *p = nullptr;
        DOTS_HIP(hipMalloc(&p, std::max<size_t>(bytes, 8)));
        tmp.push_back(p);
        if (host) DOTS_HIP(hipMemcpyAsync(p, host, bytes, hipMemcpyHostToDevice, c->stream));
        else DOTS_HIP(hipMemsetAsync(p, 0, std::max<size_t>(bytes, 8), c->stream));
        *out = p;
        return 0;
    };
    int64_t srows = 0;
    for (const FrontNode &nd : nodes) srows = std::max(srows, nd.soff + (int64_t)nd.b * nd.b);
    void *C = nullptr, *S = nullptr, *p0 = nullptr, *p1 = nullptr, *ln = nullptr, *gr = nullptr, *bp = nullptr;
    if ((rc = dalloc(&C, sizeof(double) * ((size_t)h->n_entries << sh), nullptr)) ||
        (rc = dalloc(&S, sizeof(double) * ((size_t)std::max<int64_t>(srows, 1) << sh), nullptr)) ||
        (rc = dalloc(&p0, sizeof(int) * (size_t)h->n_front_rows, h->pull0)) || (rc = dalloc(&p1, sizeof(int) * (size_t)h->n_front_rows, h->pull1)) ||
        (rc = dalloc(&ln, sizeof(int) * (size_t)h->n_nodes, h->level_nodes)) || (rc = dalloc(&gr, sizeof(int) * (size_t)d.TP, grounded_host)) ||
        (rc = dalloc(&bp, sizeof(int), nullptr))) {
        release();
        return rc;
    }
    FactArgs g{};
    g.sh = sh; g.TP = d.TP; g.ncol = d.cg_ncol;
    g.sigma = d.sigma; g.eps = c->prm.eps; g.grounded = (const int *)gr;
    g.rowptr = d.rowptr; g.col = d.col; g.val = d.val; g.mass = d.mass_v;
    g.pull0 = (const int *)p0; g.pull1 = (const int *)p1;
    g.C = (double *)C; g.T = T; g.S = (double *)S;
    g.bad_pivot = (int *)bp;
    // panel width: the diagonal block of all modes must fit in LDS (w * w * TP doubles <= 64 KB)
    int w = 16;
    while ((size_t)w * w * d.TP * sizeof(double) > 65536 && w > 1) w /= 2;
    const int Q = 256 >> sh;
    for (int l = 0; l < h->n_levels && !rc; ++l) {
        const int cnt = h->level_ptr[l + 1] - h->level_ptr[l];
        if (cnt <= 0) continue;
        int max_n = 0, max_m = 0, max_b = 0;
        int64_t max_e = 0;
        for (int k = h->level_ptr[l]; k < h->level_ptr[l + 1]; ++k) {
            const FrontNode &nd = nodes[(size_t)h->level_nodes[k]];
            max_n = std::max(max_n, nd.n);
            max_b = std::max(max_b, nd.b);
            max_m = std::max(max_m, nd.n + nd.b);
            max_e = std::max(max_e, (int64_t)(nd.n + nd.b) * nd.n + (int64_t)nd.b * nd.b);
        }
        g.level_nodes = (const int *)ln + h->level_ptr[l];
        auto blocks = [&](int64_t items) { return (unsigned)std::min<int64_t>(std::max<int64_t>((items + Q - 1) / Q, 1), 4096); };
        hipLaunchKernelGGL(k_fact_assemble, dim3(blocks(max_e), cnt), dim3(256), 0, c->stream, g, f);
        const int rows_per_wg = 8 * Q;
        const int tiles = (max_m + 15) / 16;
        for (int k0 = 0; k0 < max_n; k0 += w) {
            const int below = std::max(max_m - (k0 + 1), 0);
            const size_t lds = sizeof(double) * (size_t)w * w * d.TP;
            hipLaunchKernelGGL((k_fact_panel<true>), dim3(1, cnt), dim3(256), lds, c->stream, g, f, k0, w, rows_per_wg);
            if (below > 0)
                hipLaunchKernelGGL((k_fact_panel<false>), dim3((below + rows_per_wg - 1) / rows_per_wg, cnt), dim3(256), lds, c->stream, g, f, k0, w, rows_per_wg);
            hipLaunchKernelGGL(k_fact_update, dim3(tiles * tiles, cnt), dim3(256), 0, c->stream, g, f, k0, w, tiles);
        }
        if (max_n > 0) hipLaunchKernelGGL(k_fact_linv, dim3((max_n + Q - 1) / Q, cnt), dim3(256), 0, c->stream, g, f);
        if (max_b > 0 && max_n > 0) hipLaunchKernelGGL(k_fact_gmat, dim3(blocks((int64_t)max_b * max_n), cnt), dim3(256), 0, c->stream, g, f);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) rc = hip_fail(e, "factorisation launch", __FILE__, __LINE__);
    }
    int bad = 0;
    hipError_t e = hipMemcpyAsync(&bad, bp, sizeof(int), hipMemcpyDeviceToHost, c->stream);
    hipError_t e2 = hipStreamSynchronize(c->stream);
    release();
    if (rc) return rc;
    DOTS_HIP(e);
    DOTS_HIP(e2);
    if (bad) {
        set_error("front_setup: a pivot of the factorisation is not positive: K + (sigma + eps) M is not positive definite "
                  "(disconnected mesh, degenerate triangles or eps < 0?)");
        return DOTS_ERR_ARGUMENT;
    }
    return 0;
}

}  // namespace dots
