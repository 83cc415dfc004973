// Jacobi-preconditioned CG for the space-time Laplacian of step 1 (replaces the reference's
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
//   SPACETIME  one PCG on the coupled operator, one set of scalars (alpha, beta, ...).
//   MODAL      the orthonormal time eigen-basis Q (DCT-II) block-diagonalises K into T+1 shifted
//              surface problems K_space + (sigma_a + eps) M; they are solved as one batched PCG with
//              per-column scalars; converged columns are frozen and skipped.
//
// Reductions are two-stage and deterministic: workgroups write partial sums, a one-workgroup-per-column
// kernel adds them in a fixed order and updates the scalars on device.  Nothing returns to the host
// inside an iteration; the host polls the `done` flags between chunks of iterations.
#include "dots_dev.h"

namespace dots {

using S = CgScalOffsets;

template <bool MODAL>
__device__ __forceinline__ double shift_of(const Dev &d, int t) {
    if (MODAL) return d.sigma[t];
    return ((t == 0 || t == d.T) ? 1.0 : 2.0) / (d.h * d.h);
}

// true when every column has converged (uniform over the workgroup)
template <bool MODAL>
__device__ __forceinline__ bool all_done(const Dev &d) {
    if (!MODAL) return d.flags[0] != 0;
    const int nc = d.T + 1;
    int mine = 1;
    for (int c = threadIdx.x; c < nc; c += BLOCK) mine &= (d.flags[c] != 0);
    return __syncthreads_and(mine) != 0;
}

// per-column (MODAL) or whole-block sum of one value per thread -> partials[(slot*NC + c)*nblk + blk]
template <bool MODAL>
__device__ __forceinline__ void emit_partial(const Dev &d, double acc, int slot, double *lds /* [BLOCK] */) {
    const int nblk = gridDim.x;
    if (MODAL) {
        // all elements of a thread share the column (tid & (TP-1)); TP <= BLOCK is enforced at create
        lds[threadIdx.x] = acc;
        __syncthreads();
        const int nc = d.T + 1;
        if ((int)threadIdx.x < d.TP) {
            double s = 0.0;
            for (int j = threadIdx.x; j < BLOCK; j += d.TP) s += lds[j];
            if ((int)threadIdx.x < nc) d.partials[((int64_t)slot * nc + threadIdx.x) * nblk + blockIdx.x] = s;
        }
        __syncthreads();
    } else {
        double v[1] = {acc};
        block_sum<1>(v, lds);
        if (threadIdx.x == 0) d.partials[(int64_t)slot * nblk + blockIdx.x] = v[0];
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// Operator application with the CSR row block staged in LDS.
//   FUSE_P = true : p_new = z + beta * p_old is formed on the fly at every gathered entry (so the
//                   direction update costs no extra kernel), p_new and Ap = K p_new are written for the
//                   tile's own rows, and the partial sums of p^T Ap are emitted.
//   FUSE_P = false: plain y = K x (initial residual, tests, operator parity).
// ------------------------------------------------------------------------------------------
template <bool MODAL, bool FUSE_P>
__global__ __launch_bounds__(BLOCK) void k_cg_apply(Dev d, const double *__restrict__ zin, const double *__restrict__ p_old,
                                                    double *__restrict__ p_new, double *__restrict__ out, double eps,
                                                    int check_done) {
    __shared__ int s_rp[TILE_ELEMS / 8 + 2];
    __shared__ int s_col[LDS_NNZ_CAP];
    __shared__ double s_val[LDS_NNZ_CAP];
    __shared__ double s_red[BLOCK];
    if (check_done && all_done<MODAL>(d)) return;

    const int tile = xcd_tile(blockIdx.x, d.n_vtiles);
    double acc = 0.0;
    if (tile < d.n_vtiles) {
        const int v0 = tile * d.VT;
        const int nrows = min(d.VT, d.V - v0);
        const int rp0 = d.rowptr[v0];
        for (int i = threadIdx.x; i <= nrows; i += BLOCK) s_rp[i] = d.rowptr[v0 + i] - rp0;
        __syncthreads();
        const int nloc = s_rp[nrows];
        for (int i = threadIdx.x; i < min(nloc, LDS_NNZ_CAP); i += BLOCK) {
            s_col[i] = d.col[rp0 + i];
            s_val[i] = d.val[rp0 + i];
        }
        __syncthreads();

        for (int e = threadIdx.x; e < TILE_ELEMS; e += BLOCK) {
            const int vl = e >> d.tp_shift, t = e & (d.TP - 1);
            const int v = v0 + vl;
            if (vl >= nrows || t > d.T) continue;
            const int cidx = MODAL ? t : 0;
            if (FUSE_P && MODAL && d.flags[cidx]) continue;   // frozen column
            const double beta = FUSE_P ? d.scal[S::BETA + cidx] : 0.0;
            const bool use_p = FUSE_P && (beta != 0.0);
            double sum = 0.0;
            for (int j = s_rp[vl]; j < s_rp[vl + 1]; ++j) {
                const int u = (j < LDS_NNZ_CAP) ? s_col[j] : d.col[rp0 + j];
                const double w = (j < LDS_NNZ_CAP) ? s_val[j] : d.val[rp0 + j];
                const int iu = idxV(d, u, t);
                double pu = zin[iu];
                if (use_p) pu += beta * p_old[iu];
                sum += w * pu;
            }
            const int iv = idxV(d, v, t);
            double pv = zin[iv];
            if (use_p) pv += beta * p_old[iv];
            const double m = d.mass_v[v];
            if (MODAL) {
                sum += (d.sigma[t] + eps) * m * pv;
            } else {
                double pm = 0.0, pp = 0.0;
                if (t > 0) {
                    pm = zin[iv - 1];
                    if (use_p) pm += beta * p_old[iv - 1];
                }
                if (t < d.T) {
                    pp = zin[iv + 1];
                    if (use_p) pp += beta * p_old[iv + 1];
                }
                const double ct = (t == 0 || t == d.T) ? 1.0 : 2.0;
                sum += m * ((ct * pv - pm - pp) / (d.h * d.h) + eps * pv);
            }
            out[iv] = sum;
            if (FUSE_P) {
                p_new[iv] = pv;
                acc += pv * sum;
            }
        }
    }
    if (FUSE_P) emit_partial<MODAL>(d, acc, 0, s_red);
}

// r = (b - bmean) - K x ;  z = M^-1 r ;  partial sums of r^T z (slot 0) and b~^T M^-1 b~ (slot 1)
template <bool MODAL>
__global__ __launch_bounds__(BLOCK) void k_cg_r0(Dev d, const double *__restrict__ b, const double *__restrict__ Kx, double eps) {
    __shared__ double s_red[BLOCK];
    const int tile = xcd_tile(blockIdx.x, d.n_vtiles);
    double a0 = 0.0, a1 = 0.0;
    if (tile < d.n_vtiles) {
        for (int e = threadIdx.x; e < TILE_ELEMS; e += BLOCK) {
            const int v = tile * d.VT + (e >> d.tp_shift), t = e & (d.TP - 1);
            if (v >= d.V || t > d.T) continue;
            const int iv = idxV(d, v, t);
            const int cidx = MODAL ? t : 0;
            const double bt = b[iv] - d.scal[S::BMEAN + cidx];
            const double r = bt - Kx[iv];
            const double dinv = 1.0 / (d.kdiag[v] + (shift_of<MODAL>(d, t) + eps) * d.mass_v[v]);
            const double z = dinv * r;
            d.cg_r[iv] = r;
            d.cg_z[iv] = z;
            a0 += r * z;
            a1 += bt * dinv * bt;
        }
    }
    emit_partial<MODAL>(d, a0, 0, s_red);
    emit_partial<MODAL>(d, a1, 1, s_red);
}

// x += alpha p ; r -= alpha Ap ; z = M^-1 r ; partial sums of r^T z
template <bool MODAL>
__global__ __launch_bounds__(BLOCK) void k_cg_update(Dev d, double *__restrict__ x, const double *__restrict__ p, double eps) {
    __shared__ double s_red[BLOCK];
    if (all_done<MODAL>(d)) return;
    const int tile = xcd_tile(blockIdx.x, d.n_vtiles);
    double acc = 0.0;
    if (tile < d.n_vtiles) {
        for (int e = threadIdx.x; e < TILE_ELEMS; e += BLOCK) {
            const int v = tile * d.VT + (e >> d.tp_shift), t = e & (d.TP - 1);
            if (v >= d.V || t > d.T) continue;
            const int cidx = MODAL ? t : 0;
            if (MODAL && d.flags[cidx]) continue;
            const int iv = idxV(d, v, t);
            const double alpha = d.scal[S::ALPHA + cidx];
            x[iv] += alpha * p[iv];
            const double r = d.cg_r[iv] - alpha * d.cg_Ap[iv];
            const double z = r / (d.kdiag[v] + (shift_of<MODAL>(d, t) + eps) * d.mass_v[v]);
            d.cg_r[iv] = r;
            d.cg_z[iv] = z;
            acc += r * z;
        }
    }
    emit_partial<MODAL>(d, acc, 0, s_red);
}

// One workgroup per scalar column: add the partial sums in a fixed order and update the scalars.
//   STAGE 0: after k_cg_r0     rz, bref, beta = 0, done, iteration counter = 0
//   STAGE 1: after k_cg_apply  pAp, alpha
//   STAGE 2: after k_cg_update rz_new, beta, done, iteration counter += 1
//   STAGE 3: mean of b         bmean (only when the operator is singular: eps == 0)
template <bool MODAL, int STAGE>
__global__ __launch_bounds__(BLOCK) void k_cg_reduce(Dev d, int nblk, double tol2, double mean_scale) {
    __shared__ double lds[8];
    const int nc = MODAL ? d.T + 1 : 1;
    const int c = blockIdx.x;
    if (STAGE == 1 || STAGE == 2) {
        if (all_done<MODAL>(d)) return;
    }
    double v[2] = {0.0, 0.0};
    for (int g = threadIdx.x; g < nblk; g += BLOCK) {
        v[0] += d.partials[((int64_t)0 * nc + c) * nblk + g];
        if (STAGE == 0) v[1] += d.partials[((int64_t)1 * nc + c) * nblk + g];
    }
    block_sum<2>(v, lds);
    if (threadIdx.x != 0) return;
    if (STAGE == 0) {
        d.scal[S::RZ + c] = v[0];
        d.scal[S::BREF + c] = v[1];
        d.scal[S::BETA + c] = 0.0;
        d.scal[S::ALPHA + c] = 0.0;
        d.flags[c] = (v[0] <= tol2 * v[1]) ? 1 : 0;
        if (c == 0) d.flags[FLAG_ITERS] = 0;
    } else if (STAGE == 1) {
        const bool done = d.flags[c] != 0;
        d.scal[S::PAP + c] = v[0];
        d.scal[S::ALPHA + c] = (done || !(v[0] > 0.0)) ? 0.0 : d.scal[S::RZ + c] / v[0];
    } else if (STAGE == 2) {
        if (!d.flags[c]) {
            const double rz = d.scal[S::RZ + c];
            d.scal[S::BETA + c] = (rz > 0.0) ? v[0] / rz : 0.0;
            d.scal[S::RZ + c] = v[0];
            d.flags[c] = (v[0] <= tol2 * d.scal[S::BREF + c]) ? 1 : 0;
        }
        if (c == 0) d.flags[FLAG_ITERS] += 1;
    } else {
        d.scal[S::BMEAN + c] = (c == 0) ? v[0] * mean_scale : 0.0;
    }
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

// ------------------------------------------------------------------------------------------
// host driver
// ------------------------------------------------------------------------------------------
template <bool MODAL>
static int cg_iterations(Ctx *c, double *x, int n_iter, int start_parity) {
    const Dev &d = c->d;
    const int g = xcd_grid(d.n_vtiles);
    const int nc = MODAL ? d.T + 1 : 1;
    const double eps = c->prm.eps, tol2 = c->prm.cg_tol * c->prm.cg_tol;
    for (int it = 0; it < n_iter; ++it) {
        const bool odd = ((start_parity + it) & 1) != 0;
        double *p_old = odd ? d.cg_p1 : d.cg_p0;
        double *p_new = odd ? d.cg_p0 : d.cg_p1;
        hipLaunchKernelGGL((k_cg_apply<MODAL, true>), dim3(g), dim3(BLOCK), 0, c->stream, d, d.cg_z, p_old, p_new, d.cg_Ap, eps, 1);
        hipLaunchKernelGGL((k_cg_reduce<MODAL, 1>), dim3(nc), dim3(BLOCK), 0, c->stream, d, g, tol2, 0.0);
        hipLaunchKernelGGL((k_cg_update<MODAL>), dim3(g), dim3(BLOCK), 0, c->stream, d, x, p_new, eps);
        hipLaunchKernelGGL((k_cg_reduce<MODAL, 2>), dim3(nc), dim3(BLOCK), 0, c->stream, d, g, tol2, 0.0);
    }
    DOTS_HIP(hipGetLastError());
    return 0;
}

// Instantiate (once per eps) a hipGraph holding `unit` PCG iterations; replaying it costs one host call.
template <bool MODAL>
static int cg_graph_prepare(Ctx *c, double *x, int unit) {
    if (c->cg_graph && c->cg_graph_iters == unit && c->cg_graph_eps == c->prm.eps && c->cg_graph_tol == c->prm.cg_tol) return 0;
    if (c->cg_graph) {
        (void)hipGraphExecDestroy(c->cg_graph);
        c->cg_graph = nullptr;
    }
    hipGraph_t graph = nullptr;
    DOTS_HIP(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
    int rc = cg_iterations<MODAL>(c, x, unit, 0);
    hipError_t e = hipStreamEndCapture(c->stream, &graph);
    if (rc != 0) return rc;
    DOTS_HIP(e);
    DOTS_HIP(hipGraphInstantiate(&c->cg_graph, graph, nullptr, nullptr, 0));
    (void)hipGraphDestroy(graph);
    c->cg_graph_iters = unit;
    c->cg_graph_eps = c->prm.eps;
    c->cg_graph_tol = c->prm.cg_tol;
    return 0;
}

template <bool MODAL>
static int cg_solve_impl(Ctx *c, dots_step_stats *stats) {
    const Dev &d = c->d;
    const int g = xcd_grid(d.n_vtiles);
    const int nc = MODAL ? d.T + 1 : 1;
    const double eps = c->prm.eps, tol2 = c->prm.cg_tol * c->prm.cg_tol;
    const bool singular = (eps == 0.0);
    double *x = MODAL ? d.cg_x : d.phi;
    const double *b = d.cg_b;

    if (MODAL) {
        // b^ = Q^T b (into cg_Ap as scratch), x^ = Q^T phi (warm start in mode space)
        hipLaunchKernelGGL((k_time_modes<true>), dim3(g), dim3(BLOCK), 0, c->stream, d, d.cg_b, d.cg_p0, 1);
        hipLaunchKernelGGL((k_time_modes<true>), dim3(g), dim3(BLOCK), 0, c->stream, d, d.phi, d.cg_x, 0);
        b = d.cg_p0;   // consumed by k_cg_r0 before the first iteration overwrites p0 (iteration 0 writes p1)
    }
    // mean of b over the null space (partials were emitted by k_rhs / the forward transform)
    const double mean_scale = !singular ? 0.0 : (MODAL ? 1.0 / d.V : 1.0 / ((double)d.V * (d.T + 1)));
    hipLaunchKernelGGL((k_cg_reduce<MODAL, 3>), dim3(nc), dim3(BLOCK), 0, c->stream, d, g, tol2, mean_scale);
    hipLaunchKernelGGL((k_cg_apply<MODAL, false>), dim3(g), dim3(BLOCK), 0, c->stream, d, x, nullptr, nullptr, d.cg_Ap, eps, 0);
    hipLaunchKernelGGL((k_cg_r0<MODAL>), dim3(g), dim3(BLOCK), 0, c->stream, d, b, d.cg_Ap, eps);
    hipLaunchKernelGGL((k_cg_reduce<MODAL, 0>), dim3(nc), dim3(BLOCK), 0, c->stream, d, g, tol2, 0.0);
    DOTS_HIP(hipGetLastError());

    const int max_iter = c->prm.cg_max_iter > 0 ? c->prm.cg_max_iter : 10000;
    const int unit = 8;   // iterations per graph replay (even, so the p ping-pong parity is preserved)
    int rc = cg_graph_prepare<MODAL>(c, x, unit);
    if (rc != 0) return rc;
    int launched = 0, iters = 0;
    bool done = false;
    // first burst sized from the previous solve (iteration counts drift slowly along the ALM)
    int burst = c->last_cg_iters > 2 * unit ? ((c->last_cg_iters - unit) / unit) * unit : unit;
    while (!done && launched < max_iter) {
        for (int k = 0; k < burst / unit; ++k) DOTS_HIP(hipGraphLaunch(c->cg_graph, c->stream));
        launched += burst;
        burst = unit;
        DOTS_HIP(hipMemcpyAsync(c->h_flags, d.flags, sizeof(int) * FLAG_TOTAL, hipMemcpyDeviceToHost, c->stream));
        DOTS_HIP(hipStreamSynchronize(c->stream));
        done = true;
        for (int k = 0; k < nc; ++k) done = done && (c->h_flags[k] != 0);
        iters = c->h_flags[FLAG_ITERS];
    }
    c->last_cg_iters = iters;
    if (MODAL) {
        hipLaunchKernelGGL((k_time_modes<false>), dim3(g), dim3(BLOCK), 0, c->stream, d, d.cg_x, d.phi, 0);
        DOTS_HIP(hipGetLastError());
    }
    if (stats) {
        DOTS_HIP(hipMemcpyAsync(c->h_pinned, d.scal, sizeof(double) * S::TOTAL, hipMemcpyDeviceToHost, c->stream));
        DOTS_HIP(hipStreamSynchronize(c->stream));
        double worst = 0.0;
        for (int k = 0; k < nc; ++k) {
            const double br = c->h_pinned[S::BREF + k], rz = c->h_pinned[S::RZ + k];
            if (br > 0.0 && rz / br > worst) worst = rz / br;
        }
        stats->cg_last_rel_residual = sqrt(worst);
        stats->cg_last_iterations = iters;
        stats->cg_iterations += iters;
        if (!done) stats->cg_not_converged += 1;
    }
    return 0;
}

int cg_solve(Ctx *c, dots_step_stats *stats) {
    return c->lap_solver == DOTS_LAP_MODAL_PCG ? cg_solve_impl<true>(c, stats) : cg_solve_impl<false>(c, stats);
}

// y = K x on node-layout arrays with the coupled space-time operator (tests, operator parity)
int cg_apply_operator(Ctx *c, const double *x, double *y) {
    const int g = xcd_grid(c->d.n_vtiles);
    hipLaunchKernelGGL((k_cg_apply<false, false>), dim3(g), dim3(BLOCK), 0, c->stream, c->d, x, nullptr, nullptr, y, c->prm.eps, 0);
    DOTS_HIP(hipGetLastError());
    return 0;
}

// Time `reps` launches of the dominant kernel (fused operator application) between two hipEvents.
int cg_bench(Ctx *c, int which, int reps, double *ms, double *bytes) {
    const Dev &d = c->d;
    const int g = xcd_grid(d.n_vtiles);
    const bool modal = c->lap_solver == DOTS_LAP_MODAL_PCG;
    // neutral scalars so that the kernel does its full work: beta = 0.5, no column frozen
    DOTS_HIP(hipMemsetAsync(d.flags, 0, sizeof(int) * FLAG_TOTAL, c->stream));
    double *hb = c->h_pinned;
    for (int k = 0; k < S::NCMAX; ++k) hb[k] = 0.5;
    DOTS_HIP(hipMemcpyAsync(d.scal + S::BETA, hb, sizeof(double) * S::NCMAX, hipMemcpyHostToDevice, c->stream));
    DOTS_HIP(hipMemsetAsync(d.cg_p0, 0, sizeof(double) * (size_t)d.V * d.TP, c->stream));
    auto launch = [&](int i) {
        double *po = (i & 1) ? d.cg_p1 : d.cg_p0, *pn = (i & 1) ? d.cg_p0 : d.cg_p1;
        if (which == 1) {
            if (modal) hipLaunchKernelGGL((k_cg_update<true>), dim3(g), dim3(BLOCK), 0, c->stream, d, d.cg_x, pn, c->prm.eps);
            else hipLaunchKernelGGL((k_cg_update<false>), dim3(g), dim3(BLOCK), 0, c->stream, d, d.cg_x, pn, c->prm.eps);
        } else {
            if (modal) hipLaunchKernelGGL((k_cg_apply<true, true>), dim3(g), dim3(BLOCK), 0, c->stream, d, d.cg_z, po, pn, d.cg_Ap, c->prm.eps, 1);
            else hipLaunchKernelGGL((k_cg_apply<false, true>), dim3(g), dim3(BLOCK), 0, c->stream, d, d.cg_z, po, pn, d.cg_Ap, c->prm.eps, 1);
        }
    };
    DOTS_HIP(hipMemsetAsync(d.scal + S::ALPHA, 0, sizeof(double) * S::NCMAX, c->stream));
    for (int i = 0; i < 3; ++i) launch(i);
    DOTS_HIP(hipEventRecord(c->ev[6], c->stream));
    for (int i = 0; i < reps; ++i) launch(i);
    DOTS_HIP(hipEventRecord(c->ev[7], c->stream));
    DOTS_HIP(hipEventSynchronize(c->ev[7]));
    float t = 0.f;
    DOTS_HIP(hipEventElapsedTime(&t, c->ev[6], c->ev[7]));
    *ms = (double)t / reps;
    const double N = (double)d.V * (d.T + 1);
    if (which == 1) *bytes = 8.0 * N * 7.0 + 16.0 * d.V;                      // x,p,r,Ap read; x,r,z written
    else *bytes = 12.0 * c->nnz + 4.0 * (d.V + 1) + 16.0 * d.V + 8.0 * N * 4.0;  // CSR + z,p_old read; p_new,Ap written
    return 0;
}

}  // namespace dots
