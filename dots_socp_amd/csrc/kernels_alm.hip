// ALM element-wise / gather kernels for gfx950: second-order-cone projection, right-hand side
// of the phi step, the fused (q, lambda_c) + multiplier update, scaling tools, layout conversion.
// All HBM-bound fp64; one thread per (row, time) element, rows = vertices or (triangle, xyz).
#include "dots_dev.h"

#include <algorithm>

namespace dots {

// ------------------------------------------------------------------------------------------
// Step 1-2  second-order-cone projection      (reference: solver_socp.py:988-1042)
//
// Cone of (t, v):  z_fst >= || ( z_end, D[k,f] * z_mid[t, s, k, f, c]  over the corners (f,k) of v ) ||.
// The reference does this with two 0/1 incidence SpMVs and >= 9 passes over 18*T*F-sized arrays;
// here one thread owns (v, t): it walks the vertex's corner list once to accumulate the squared
// norm, forms lambda, and walks it again to write z_mid of its own corners (no atomics: every
// corner belongs to exactly one vertex).  beta_mid is read from HBM once, the second walk hits L2.
// ------------------------------------------------------------------------------------------
// ONLY_MULTIPLIER: the iteration driver (dots_step) does not need z_mid in memory: it stores the cone
// multiplier lambda[v][t] (T*V values instead of 18*T*F) and steps 2+3 rebuild z_mid = lambda/D * pre-image on
// the fly from beta_mid and B, which they read anyway: the second corner walk and 18*T*F stores disappear here,
// 18*T*F loads disappear there.
// The squared norm is accumulated in two halves, s = 0 (entries of node t) and s = 1 (entries of node t + 1), each over
// the corner list in order: when node t + 1 belongs to the next time slab that rank forms the s = 1 half from its own
// B and beta_mid (soc_half_of_first_node) and this one reads it from the halo -- bit for bit the same sum.
// two consecutive time columns of a row (16-byte aligned: even column, pitch a power of two)

// A corner's share of a sum over xyz is formed FIRST, as (x + y) + z, and the shares are then added in corner-list order: the
// kernels that gather these terms from memory and the steps-2+3 kernel that forms them from its registers for the next
// iteration (DOTS_STEP_CARRY: cn_sq, cn_g below) then produce the same sums bit for bit (-ffp-contract=off: no fused multiply-adds).
__device__ __forceinline__ double sum3(double x, double y, double z) { return (x + y) + z; }
#ifndef DOTS_CARRY_BATCH
#define DOTS_CARRY_BATCH 2
#endif
constexpr int CARRY_BATCH = DOTS_CARRY_BATCH;      // corners whose carried rows a lane loads before it adds them (closed surfaces: valence ~ 6)
// one pre-image entry of the cone's middle block squared: (D (sB B - beta_mid))^2, sBB = sB * B already rounded
__device__ __forceinline__ double soc_w2(double D, double sBB, double bm) {
    const double w = D * (sBB - bm);
    return w * w;
}

__device__ __forceinline__ double soc_half(const Dev &d, int v, int t, int s, double sB) {
    double acc = 0.0;
    for (int j = d.cptr[v]; j < d.cptr[v + 1]; ++j) {
        const int fk = d.cidx[j];
        const int f = fk / 3;
        const double D = d.c_D[j];
        double q[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) q[c] = soc_w2(D, sB * d.B[idxF(d, f, c, t + s)], d.bm[idxM(d, fk, s, c, t)]);
        acc += sum3(q[0], q[1], q[2]);
    }
    return acc;
}

// CARRIED: the corners' shares of both halves were stored by the last steps-2+3 launch (cn_sq[j][s][interval], in corner-LIST
// order: a vertex's rows are contiguous): two loads per corner, no index chain, neither B nor beta_mid is touched.
// DIV: a penalty update is pending (Ctx::pending_div): the five dual arrays still hold their values BEFORE the division by
// `dv` (admm_tools / solver_socp.py:367-371) and every entry is divided as it is read -- the same IEEE division the stand-alone
// kernel (k_divide_five) performs, so nothing changes bit for bit; steps 2+3 of the same iteration write the arrays back divided.
template <bool ONLY_MULTIPLIER, bool CARRIED = false, bool DIV = false>
__device__ __forceinline__ void soc_element(const Dev &d, int v, int t, double sz, double cd, double dv = 1.0) {
    const double sB = sz * INV_SQRT3;
    const int iv = idxV(d, v, t);
    const int j0 = d.cptr[v], j1 = d.cptr[v + 1];
    const bool own1 = t + 1 < d.nl;      // node t + 1 is held here (always on one GPU)
    double acc0 = 0.0, acc1 = 0.0;
    if (CARRIED) {
        for (int j = j0; j < j1; j += CARRY_BATCH) {
            double q0[CARRY_BATCH], q1[CARRY_BATCH];
#pragma unroll
            for (int i = 0; i < CARRY_BATCH; ++i) {
                const int64_t row = ((int64_t)(2 * min(j + i, j1 - 1)) << d.tp_shift) + t;
                q0[i] = ld1_nt(d.cn_sq + row);
                q1[i] = ld1_nt(d.cn_sq + row + d.TP);
            }
#pragma unroll
            for (int i = 0; i < CARRY_BATCH; ++i) {
                if (j + i >= j1) break;
                acc0 += q0[i];
                acc1 += q1[i];
            }
        }
        if (!own1) acc1 = d.nsq_hi[v];      // (time slab: node t + 1 belongs to the next slab, which formed this half)
    } else {
        for (int j = j0; j < j1; ++j) {
            const int fk = d.cidx[j];
            const int f = fk / 3;
            const double D = d.c_D[j];
            double q[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) q[c] = soc_w2(D, sB * d.B[idxF(d, f, c, t)], DIV ? d.bm[idxM(d, fk, 0, c, t)] / dv : d.bm[idxM(d, fk, 0, c, t)]);
            acc0 += sum3(q[0], q[1], q[2]);
            if (own1) {
#pragma unroll
                for (int c = 0; c < 3; ++c) q[c] = soc_w2(D, sB * d.B[idxF(d, f, c, t + 1)], DIV ? d.bm[idxM(d, fk, 1, c, t)] / dv : d.bm[idxM(d, fk, 1, c, t)]);
                acc1 += sum3(q[0], q[1], q[2]);
            }
        }
        if (!own1) acc1 = d.nsq_hi[v];
    }
    const double acc = acc0 + acc1;
    const double a = d.A[iv];
    const double w_fst = cd - sz * a - (DIV ? d.bf[iv] / dv : d.bf[iv]);
    const double w_end = cd + sz * a - (DIV ? d.be[iv] / dv : d.be[iv]);
    const double nrm = sqrt(acc + w_end * w_end);
    double lam = 0.5 * (1.0 + w_fst / nrm);          // 0/0 -> NaN when the pre-image is 0, as in the reference (:1018)
    if (lam == lam) lam = fmin(fmax(lam, 0.0), 1.0);  // np.clip keeps NaN; fmin/fmax would drop it
    d.zf[iv] = (lam >= 1.0) ? w_fst : lam * nrm;
    d.ze[iv] = lam * w_end;
    if (ONLY_MULTIPLIER) {
        d.lamc[iv] = lam;
        return;
    }
    for (int j = j0; j < j1; ++j) {      // (not reached on a time slab: the full projection needs every node of the interval here)
        const int fk = d.cidx[j];
        const int f = fk / 3;
        const double D = d.c_D[j];
        const double lt = lam / D;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const double w = D * (sB * d.B[idxF(d, f, c, t + s)] - d.bm[idxM(d, fk, s, c, t)]);
                d.zm[idxM(d, fk, s, c, t)] = lt * w;
            }
    }
}

// The projection of the intervals t and t + 1 (t even) of a vertex by ONE lane, multiplier only (one GPU: every node is held
// here): B and the s = 0 entries of both intervals sit in one aligned 16-byte word per row, 15 loads per corner instead of 24;
// half the waves for the same bytes (see k_q_lambda_mult_triangle2).  Interval for interval the arithmetic of soc_element.
// STAGED: the rows of B come from LDS (`rows`: the tile's distinct triangles, [position][c][ldr]; `c_loc`: the position of a
// corner-list entry's triangle) instead of from memory (k_rhs_soc_tiles): the same values, the same arithmetic.
// CARRIED: see soc_element -- one 16-byte load per corner and half (the s = 1 shares are stored in the column of their INTERVAL).
template <bool STAGED, bool CARRIED = false, bool DIV = false>
__device__ __forceinline__ void soc_element2(const Dev &d, int v, int t, double sz, double cd, const double *rows = nullptr, int ldr = 0,
                                             const int *__restrict__ c_loc = nullptr, double dv = 1.0) {
    const double sB = sz * INV_SQRT3;
    const int iv = idxV(d, v, t);
    const bool two = t + 1 < d.ni;                 // the second interval exists (T odd: not for the last pair)
    const int j0 = d.cptr[v], j1 = d.cptr[v + 1];
    double a0[2] = {0.0, 0.0}, a1[2] = {0.0, 0.0};
    if (CARRIED) {      // CARRY_BATCH corners' loads in flight (rows clamped to the list, sums in list order)
        const double *__restrict__ q = d.cn_sq + t;
        for (int j = j0; j < j1; j += CARRY_BATCH) {
            D2 q0[CARRY_BATCH], q1[CARRY_BATCH];
#pragma unroll
            for (int i = 0; i < CARRY_BATCH; ++i) {
                const int64_t row = (int64_t)(2 * min(j + i, j1 - 1)) << d.tp_shift;
                q0[i] = ld2_nt(q + row);
                q1[i] = ld2_nt(q + row + d.TP);
            }
#pragma unroll
            for (int i = 0; i < CARRY_BATCH; ++i) {
                if (j + i >= j1) break;
                a0[0] += q0[i].v[0];
                a1[0] += q1[i].v[0];
                a0[1] += q0[i].v[1];
                a1[1] += q1[i].v[1];
            }
        }
    }
    for (int j = CARRIED ? j1 : j0; j < j1; ++j) {
        const int fk = d.cidx[j];
        const int f = fk / 3;
        const double D = d.c_D[j];
        D2 bt[3], m0[3];
        double b2[3], m1a[3], m1b[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (STAGED) {
                const double *rw = rows + (c_loc[j] * 3 + c) * ldr + t;
                bt[c] = ld2(rw);
                b2[c] = two ? rw[2] : 0.0;
            } else {
                bt[c] = ld2(d.B + idxF(d, f, c, t));
                b2[c] = two ? d.B[idxF(d, f, c, t + 2)] : 0.0;
            }
            m0[c] = ld2(d.bm + idxM(d, fk, 0, c, t));
            m1a[c] = d.bm[idxM(d, fk, 1, c, t)];
            m1b[c] = two ? d.bm[idxM(d, fk, 1, c, t + 1)] : 0.0;
            if (DIV) {
                m0[c].v[0] /= dv;
                m0[c].v[1] /= dv;
                m1a[c] /= dv;
                m1b[c] /= dv;
            }
        }
        double q[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) q[c] = soc_w2(D, sB * bt[c].v[0], m0[c].v[0]);
        a0[0] += sum3(q[0], q[1], q[2]);
#pragma unroll
        for (int c = 0; c < 3; ++c) q[c] = soc_w2(D, sB * bt[c].v[1], m1a[c]);
        a1[0] += sum3(q[0], q[1], q[2]);
#pragma unroll
        for (int c = 0; c < 3; ++c) q[c] = soc_w2(D, sB * bt[c].v[1], m0[c].v[1]);
        a0[1] += sum3(q[0], q[1], q[2]);
#pragma unroll
        for (int c = 0; c < 3; ++c) q[c] = soc_w2(D, sB * b2[c], m1b[c]);
        a1[1] += sum3(q[0], q[1], q[2]);
    }
    const D2 A = ld2(d.A + iv);
    D2 bf = ld2(d.bf + iv), be = ld2(d.be + iv);
    if (DIV) {
        bf.v[0] /= dv; bf.v[1] /= dv;
        be.v[0] /= dv; be.v[1] /= dv;
    }
    D2 zf, ze, lm;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const double acc = a0[u] + a1[u];
        const double a = A.v[u];
        const double w_fst = cd - sz * a - bf.v[u];
        const double w_end = cd + sz * a - be.v[u];
        const double nrm = sqrt(acc + w_end * w_end);
        double lam = 0.5 * (1.0 + w_fst / nrm);          // 0/0 -> NaN when the pre-image is 0, as in the reference (:1018)
        if (lam == lam) lam = fmin(fmax(lam, 0.0), 1.0);  // np.clip keeps NaN; fmin/fmax would drop it
        zf.v[u] = (lam >= 1.0) ? w_fst : lam * nrm;
        ze.v[u] = lam * w_end;
        lm.v[u] = lam;
    }
    if (two) {
        st2(d.zf + iv, zf);
        st2(d.ze + iv, ze);
        st2(d.lamc + iv, lm);
    } else {
        d.zf[iv] = zf.v[0];
        d.ze[iv] = ze.v[0];
        d.lamc[iv] = lm.v[0];
    }
}

template <bool ONLY_MULTIPLIER, bool CARRIED = false>
__global__ __launch_bounds__(BLOCK) void k_soc_projection(Dev d, double sz, double cd, int n_soc, int IC) {
    // Workgroups [n_soc, gridDim.x): the modes -> time transform of phi (independent of the projection, same launch)
    if ((int)blockIdx.x >= n_soc) {
        extern __shared__ double tm_lds[];
        const int n = d.T + 1, TP = d.TP, TPp = TP + 1;
        double *Qs = tm_lds, *xs = tm_lds + IC * TP;
        const int tile = xcd_tile(blockIdx.x - n_soc, d.n_vtiles);
        if (tile >= d.n_vtiles) return;
        const int v0 = tile * d.VT;
        for (int e = threadIdx.x; e < TILE_ELEMS; e += BLOCK) {
            const int vl = e >> d.tp_shift, t = e & (TP - 1);
            xs[vl * TPp + t] = (v0 + vl < d.V && t < n) ? d.cg_x[idxV(d, v0 + vl, t)] : 0.0;
        }
        modes_from_tile<false>(d, d.Q, xs, Qs, IC, v0, d.phi);
        return;
    }
    // one element per thread: a workgroup takes a quarter of a tile (the corner walk is a chain of dependent
    // loads; four elements per thread would serialise four of them)
    // (block b and block b + G8 share a tile and an XCD: G8 is a multiple of 8)
    const int G8 = n_soc / (TILE_ELEMS / BLOCK);
    const int tile = xcd_tile(blockIdx.x % G8, d.n_vtiles);
    if (tile >= d.n_vtiles) return;
    const int v0 = tile * d.VT;
    for (int e = (blockIdx.x / G8) * BLOCK + threadIdx.x; e < TILE_ELEMS; e += TILE_ELEMS) {
        const int v = v0 + (e >> d.tp_shift), t = e & (d.TP - 1);
        if (v >= d.V || t >= d.ni) continue;
        soc_element<ONLY_MULTIPLIER, CARRIED>(d, v, t, sz, cd);
    }
}

// ------------------------------------------------------------------------------------------
// Time slabs: what the neighbouring slabs need from this one (dots_slab_stage 0 and 4).
//   forward  (to the next slab, which starts at node t0 + nl):   X = A + lambda_c - mu (right-hand side at its first node) or
//            mu (KKT) of this slab's last interval nl - 1
//   backward (to the previous slab, which ends with interval t0 - 1): the s = 1 half of that interval's cone norms --
//            its entries are compared with B of THIS slab's first node and live in this slab's column 0 -- or B of the
//            first node (KKT: Comp(rho, f(q)))
// ------------------------------------------------------------------------------------------
template <bool CARRIED>
__global__ __launch_bounds__(BLOCK) void k_slab_pack_iteration(Dev d, double sz, double *__restrict__ send_x, double *__restrict__ send_nsq) {
    const int v = blockIdx.x * BLOCK + threadIdx.x;
    if (v >= d.V) return;
    if (d.t0 + d.nl <= d.T) {        // a next slab exists: interval nl - 1 is this slab's last and ends at its first node
        const int iv = idxV(d, v, d.nl - 1);
        send_x[v] = d.A[iv] + d.lam[iv] - d.mu[iv];
    }
    if (d.t0 > 0) {
        if (CARRIED) {               // the corners' shares were left by steps 2+3 (cn_lo: same values, same order of the sum)
            double acc = 0.0;
            for (int j = d.cptr[v]; j < d.cptr[v + 1]; ++j) acc += d.cn_lo[j];
            send_nsq[v] = acc;
        } else {
            send_nsq[v] = soc_half(d, v, -1, 1, sz * INV_SQRT3);
        }
    }
}
__global__ __launch_bounds__(BLOCK) void k_slab_pack_kkt(Dev d, double *__restrict__ send_mu, double *__restrict__ send_b) {
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i < d.V && d.t0 + d.nl <= d.T) send_mu[i] = d.mu[idxV(d, i, d.nl - 1)];
    if (i < 3 * d.F && d.t0 > 0) send_b[i] = d.B[(int64_t)i << d.tp_shift];
}
int launch_slab_pack_iteration(Ctx *c) {
    if (c->d.nl == 0) return 0;
    if (c->carry_valid) hipLaunchKernelGGL(k_slab_pack_iteration<true>, dim3((c->d.V + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, c->stream, c->d, c->prm.scale_z, c->slab.send_x, c->slab.send_nsq);
    else hipLaunchKernelGGL(k_slab_pack_iteration<false>, dim3((c->d.V + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, c->stream, c->d, c->prm.scale_z, c->slab.send_x, c->slab.send_nsq);
    DOTS_HIP(hipGetLastError());
    return 0;
}
int launch_slab_pack_kkt(Ctx *c) {
    if (c->d.nl == 0) return 0;
    const int n = std::max(c->d.V, 3 * c->d.F);
    hipLaunchKernelGGL(k_slab_pack_kkt, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, c->stream, c->d, c->slab.send_mu, c->slab.send_b);
    DOTS_HIP(hipGetLastError());
    return 0;
}

int launch_soc_projection(Ctx *c, int zmid_mode, bool with_inverse) {
    const int n_soc = xcd_grid(c->d.n_vtiles) * (TILE_ELEMS / BLOCK);
    const int n_inv = with_inverse ? xcd_grid(c->d.n_vtiles) : 0;
    const size_t lds = with_inverse ? time_modes_tile_lds(c->d) : 0;
    const int IC = time_modes_chunk(c->d);
    if (zmid_mode && c->carry_valid)
        hipLaunchKernelGGL((k_soc_projection<true, true>), dim3(n_soc + n_inv), dim3(BLOCK), lds, c->stream, c->d, c->prm.scale_z, c->prm.const_d, n_soc, IC);
    else if (zmid_mode)
        hipLaunchKernelGGL((k_soc_projection<true>), dim3(n_soc + n_inv), dim3(BLOCK), lds, c->stream, c->d, c->prm.scale_z, c->prm.const_d, n_soc, IC);
    else
        hipLaunchKernelGGL((k_soc_projection<false>), dim3(n_soc + n_inv), dim3(BLOCK), lds, c->stream, c->d, c->prm.scale_z, c->prm.const_d, n_soc, IC);
    DOTS_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------
// Step 1-1 right-hand side                    (reference: solver_socp.py:976-986, :886-921)
//   rhs = div_t((A + lambda_c - mu) * mass) + div_x((B - E) * area) - boundary - eps * mass * phi
// The PCG solves K phi = b with K = -Laplacian (SPD-semidefinite), so b = -rhs is stored.
// div_x is a gather over the vertex's corner list (no scatter, no atomics).  Each workgroup also
// emits the partial sum of b (mean removal for the singular eps = 0 operator).
// ------------------------------------------------------------------------------------------
template <bool CARRIED = false, bool DIV = false>
__device__ __forceinline__ double rhs_value(const Dev &d, int v, int t, double r, double eps, double dv = 1.0) {
    const int iv = idxV(d, v, t);
    const double m = d.mass_v[v];
    const double ih = 1.0 / d.h;
    double xt = 0.0, xm = 0.0;
    if (t < d.ni) xt = (d.A[iv] + d.lam[iv] - (DIV ? d.mu[iv] / dv : d.mu[iv])) * m;
    if (t > 0) xm = (d.A[iv - 1] + d.lam[iv - 1] - (DIV ? d.mu[iv - 1] / dv : d.mu[iv - 1])) * m;
    else if (has_prev_interval(d, t)) xm = d.X_lo[v] * m;      // interval t0 - 1 lives in the previous time slab
    double rhs = (xt - xm) * ih;
    double ds = 0.0;
    if (CARRIED) {      // the corners' shares of div_x((B - E) area) were stored by the last steps-2+3 launch (cn_g[j][node])
        const int jc0 = d.cptr[v], jc1 = d.cptr[v + 1];
        for (int j = jc0; j < jc1; j += CARRY_BATCH) {
            double gj[CARRY_BATCH];
#pragma unroll
            for (int i = 0; i < CARRY_BATCH; ++i) gj[i] = ld1_nt(d.cn_g + ((int64_t)min(j + i, jc1 - 1) << d.tp_shift) + t);
#pragma unroll
            for (int i = 0; i < CARRY_BATCH; ++i) {
                if (j + i >= jc1) break;
                ds += gj[i];
            }
        }
    } else {
        for (int j = d.cptr[v]; j < d.cptr[v + 1]; ++j) {
            const int f = d.cidx[j] / 3;
            double g[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int64_t i = idxF(d, f, c, t);
                g[c] = d.c_gA[j * 3 + c] * (d.B[i] - (DIV ? d.E[i] / dv : d.E[i]));
            }
            ds += sum3(g[0], g[1], g[2]);
        }
    }
    rhs -= ds;
    if (first_node(d, t)) rhs += d.mu0[v] / (r * d.h);
    if (last_node(d, t)) rhs -= d.mu1[v] / (r * d.h);
    rhs -= eps * m * d.phi[iv];
    return -rhs;
}

// Workgroups [n_rhs, gridDim.x) (when there are any): the cone projection, a quarter tile each, as in k_rhs_modes.
template <bool CARRIED>
__global__ __launch_bounds__(BLOCK) void k_rhs(Dev d, double r, double eps, int n_rhs, double sz, double cd) {
    __shared__ double lds[4];
    if ((int)blockIdx.x >= n_rhs) {
        const int G8 = (gridDim.x - n_rhs) / (TILE_ELEMS / BLOCK), b = blockIdx.x - n_rhs;
        const int st = xcd_tile(b % G8, d.n_vtiles);
        if (st >= d.n_vtiles) return;
        const int e = (b / G8) * BLOCK + threadIdx.x, v = st * d.VT + (e >> d.tp_shift), t = e & (d.TP - 1);
        if (v < d.V && t < d.ni) soc_element<true, CARRIED>(d, v, t, sz, cd);
        return;
    }
    const int tile = xcd_tile(blockIdx.x, d.n_vtiles);
    double part[1] = {0.0};
    if (tile < d.n_vtiles) {
        const int v0 = tile * d.VT;
        for (int e = threadIdx.x; e < TILE_ELEMS; e += BLOCK) {
            const int v = v0 + (e >> d.tp_shift), t = e & (d.TP - 1);
            if (v >= d.V || t >= d.nl) continue;
            const double b = rhs_value<CARRIED>(d, v, t, r, eps);
            d.cg_b[idxV(d, v, t)] = b;
            part[0] += b;
        }
    }
    block_sum<1>(part, lds);
    if (threadIdx.x == 0) d.partials[blockIdx.x] = part[0];
}

// The same right-hand side followed at once by the time-mode transform of the tile (direct solver): b never goes
// to memory, the solver's input  bhat[v][a] = sum_t Q[t][a] b[v][t]  is written instead.
// 1024 threads: one right-hand-side value (a corner walk) per thread.
constexpr int RHS_NB = 1024;
// Workgroups [n_rhs, gridDim.x): the cone projection of a tile (it reads nothing this kernel or the solve writes
// and is as latency-bound as the corner walks here: one launch instead of two).
__global__ __launch_bounds__(RHS_NB) void k_rhs_modes(Dev d, double r, double eps, double *__restrict__ bhat, int IC, int n_rhs, double sz, double cd) {
    if ((int)blockIdx.x >= n_rhs) {
        const int st = xcd_tile(blockIdx.x - n_rhs, d.n_vtiles);
        if (st >= d.n_vtiles) return;
        const int e = threadIdx.x, v = st * d.VT + (e >> d.tp_shift), t = e & (d.TP - 1);
        if (v < d.V && t < d.ni) soc_element<true>(d, v, t, sz, cd);
        return;
    }
    extern __shared__ double tm_lds[];
    const int n = d.T + 1, TP = d.TP, TPp = TP + 1;
    double *Qs = tm_lds;                 // [IC][TP]
    double *xs = tm_lds + IC * TP;       // [VT][TPp]
    const int tile = xcd_tile(blockIdx.x, d.n_vtiles);
    if (tile >= d.n_vtiles) return;
    const int v0 = tile * d.VT;
    stage_q_chunk<true, RHS_NB>(d, d.Q, Qs, 0, min(IC, n));      // in flight while the corner walks run
    for (int e = threadIdx.x; e < TILE_ELEMS; e += RHS_NB) {
        const int vl = e >> d.tp_shift, t = e & (TP - 1);
        xs[vl * TPp + t] = (v0 + vl < d.V && t < n) ? rhs_value(d, v0 + vl, t, r, eps) : 0.0;
    }
    modes_from_tile<true, RHS_NB>(d, d.Q, xs, Qs, IC, v0, bhat, -1, 0, 1 << 30, true);
}

// The right-hand side at the nodes t and t + 1 (t even) of a vertex by one lane (one GPU), node for node the arithmetic of
// rhs_value: B and E of both nodes in one 16-byte word per row.
// STAGED: B - E of the tile's distinct triangles comes from LDS (see soc_element2).
template <bool STAGED, bool CARRIED = false, bool DIV = false>
__device__ __forceinline__ void rhs_value2(const Dev &d, int v, int t, double r, double eps, double (&out)[2], const double *rows = nullptr, int ldr = 0,
                                           const int *__restrict__ c_loc = nullptr, double dv = 1.0) {
    const int iv = idxV(d, v, t);
    const double m = d.mass_v[v];
    const double ih = 1.0 / d.h;
    const D2 A = ld2(d.A + iv), L = ld2(d.lam + iv), P = ld2(d.phi + iv);
    D2 M = ld2(d.mu + iv);
    double mp = t > 0 ? d.mu[iv - 1] : 0.0;
    if (DIV) {
        M.v[0] /= dv; M.v[1] /= dv;
        mp /= dv;
    }
    const double xp = t > 0 ? (d.A[iv - 1] + d.lam[iv - 1] - mp) * m : 0.0;      // interval t - 1
    const double x0 = t < d.ni ? (A.v[0] + L.v[0] - M.v[0]) * m : 0.0;                      // interval t
    const double x1 = t + 1 < d.ni ? (A.v[1] + L.v[1] - M.v[1]) * m : 0.0;                  // interval t + 1
    double rhs[2] = {(x0 - xp) * ih, (x1 - x0) * ih};
    double ds[2] = {0.0, 0.0};
    const int jc0 = d.cptr[v], jc1 = d.cptr[v + 1];
    if (CARRIED) {
        const double *__restrict__ g = d.cn_g + t;
        for (int j = jc0; j < jc1; j += CARRY_BATCH) {
            D2 gj[CARRY_BATCH];
#pragma unroll
            for (int i = 0; i < CARRY_BATCH; ++i) gj[i] = ld2_nt(g + ((int64_t)min(j + i, jc1 - 1) << d.tp_shift));
#pragma unroll
            for (int i = 0; i < CARRY_BATCH; ++i) {
                if (j + i >= jc1) break;
                ds[0] += gj[i].v[0];
                ds[1] += gj[i].v[1];
            }
        }
    }
    for (int j = CARRIED ? jc1 : jc0; j < jc1; ++j) {
        const int f = d.cidx[j] / 3;
        D2 b[3], e[3];
        double ga[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            ga[c] = d.c_gA[j * 3 + c];
            if (STAGED) {
                b[c] = ld2(rows + (c_loc[j] * 3 + c) * ldr + t);      // B - E, subtracted when it was staged
            } else {
                const int64_t i = idxF(d, f, c, t);
                b[c] = ld2(d.B + i);
                e[c] = ld2(d.E + i);
                if (DIV) { e[c].v[0] /= dv; e[c].v[1] /= dv; }
            }
        }
        double g0[3], g1[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            g0[c] = ga[c] * (STAGED ? b[c].v[0] : b[c].v[0] - e[c].v[0]);
            g1[c] = ga[c] * (STAGED ? b[c].v[1] : b[c].v[1] - e[c].v[1]);
        }
        ds[0] += sum3(g0[0], g0[1], g0[2]);
        ds[1] += sum3(g1[0], g1[1], g1[2]);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        rhs[u] -= ds[u];
        if (t + u == 0) rhs[u] += d.mu0[v] / (r * d.h);
        if (t + u == d.T) rhs[u] -= d.mu1[v] / (r * d.h);
        rhs[u] -= eps * m * P.v[u];
        out[u] = -rhs[u];
    }
}

// k_rhs_modes with two time columns per lane (one GPU, direct solver): 512 threads per tile.
constexpr int RHS_NB2 = RHS_NB / 2;
template <bool CARRIED, bool DIV = false>
__global__ __launch_bounds__(RHS_NB2) void k_rhs_modes2(Dev d, double r, double eps, double *__restrict__ bhat, int IC, int n_rhs, double sz, double cd,
                                                        const int *__restrict__ tile_vertex, double dv) {
    // tile_vertex (or null): the tiles' vertices taken from dots_problem_desc.patch_order instead of from the numbering
    const int e = 2 * threadIdx.x, vl = e >> d.tp_shift, t = e & (d.TP - 1);
    if ((int)blockIdx.x >= n_rhs) {
        const int st = xcd_tile(blockIdx.x - n_rhs, d.n_vtiles);
        if (st >= d.n_vtiles) return;
        const int v = tile_vertex ? tile_vertex[st * d.VT + vl] : (st * d.VT + vl < d.V ? st * d.VT + vl : -1);
        if (v >= 0 && t < d.ni) soc_element2<false, CARRIED, DIV>(d, v, t, sz, cd, nullptr, 0, nullptr, dv);
        return;
    }
    extern __shared__ double tm_lds[];
    const int n = d.T + 1, TP = d.TP, TPp = TP + 1;
    double *Qs = tm_lds;                 // [IC][TP]
    double *xs = tm_lds + IC * TP;       // [VT][TPp]
    const int tile = xcd_tile(blockIdx.x, d.n_vtiles);
    if (tile >= d.n_vtiles) return;
    const int v0 = tile * d.VT;
    const int *__restrict__ tv = tile_vertex ? tile_vertex + v0 : nullptr;
    stage_q_chunk<true, RHS_NB2>(d, d.Q, Qs, 0, min(IC, n));      // in flight while the corner walks run
    for (int ee = e; ee < TILE_ELEMS; ee += 2 * RHS_NB2) {       // (one pass: TILE_ELEMS = 2 * RHS_NB2)
        const int vv = ee >> d.tp_shift, tt = ee & (TP - 1);
        const int v = tv ? tv[vv] : (v0 + vv < d.V ? v0 + vv : -1);
        double b[2] = {0.0, 0.0};
        if (v >= 0 && tt < n) rhs_value2<false, CARRIED, DIV>(d, v, tt, r, eps, b, nullptr, 0, nullptr, dv);
        xs[vv * TPp + tt] = b[0];
        xs[vv * TPp + tt + 1] = tt + 1 < n ? b[1] : 0.0;
    }
    modes_from_tile<true, RHS_NB2>(d, d.Q, xs, Qs, IC, v0, bhat, -1, 0, 1 << 30, true, tv);
}

// The right-hand side + projection launch on PATCH tiles (dots_problem_desc.patch_order): a workgroup of 256 threads takes
// TILE2 / TP vertices of a compact patch of the surface (two time columns per lane) and stages the rows of B of the patch's
// DISTINCT triangles in LDS once; the projection walks its corner lists against them (beta_mid streams from memory, each
// entry once); then E is subtracted in place and the right-hand side walks the same lists against B - E, followed at once by
// the time-mode transform of the tile.  The corner walks of k_rhs_modes2 fetch the three rows of B (and of E) of a triangle
// once per corner VERTEX; the caches catch little of that beside the beta_mid stream (PMC at torus100k: 2.25 GB read for
// 1.50 GB algorithmic).  Here a triangle's rows are read once per tile that touches it (~1.6 tiles), and B serves both halves.
// Vertex for vertex and corner for corner the arithmetic of rhs_value2 / soc_element2: results are bit-identical.
constexpr int TILE2_NB = 256;      // (a tile of k_rhs_soc_tiles: 512 (vertex, time) elements, two per lane)
template <bool WITH_SOC>
__global__ __launch_bounds__(TILE2_NB) void k_rhs_soc_tiles(Dev d, TileDev tl, double r, double eps, double *__restrict__ bhat, int IC, double sz, double cd) {
    extern __shared__ double tm_lds[];
    const int n = d.T + 1, TP = d.TP, TPp = TP + 1, ldr = TP + 2;
    double *Qs = tm_lds;                          // [IC][TP]
    double *xs = tm_lds + IC * TP;                // [VTL][TPp]
    double *rows = xs + tl.VTL * TPp + (((tl.VTL * TPp) & 1) ? 1 : 0);      // [distinct triangles][3][ldr], 16-byte aligned
    const int tile = xcd_tile(blockIdx.x, tl.n_tiles);
    if (tile >= tl.n_tiles) return;
    const int *__restrict__ tv = tl.vertex + tile * tl.VTL;
    const int t0 = tl.tri_ptr[tile], nrow = 3 * (tl.tri_ptr[tile + 1] - t0);
    const int lr = TP >> 1, tid = threadIdx.x;    // lanes per row (two columns each)
    const int ra = (tid & (lr - 1)) * 2, r0 = tid / lr, rstep = TILE2_NB / lr;
    for (int row = r0; row < nrow; row += rstep) {
        const int f = tl.tri[t0 + row / 3], c = row - 3 * (row / 3);
        D2 b = ld2(d.B + idxF(d, f, c, ra));
        if (!WITH_SOC) {
            const D2 e = ld2(d.E + idxF(d, f, c, ra));
            b.v[0] -= e.v[0];
            b.v[1] -= e.v[1];
        }
        st2(rows + row * ldr + ra, b);
    }
    stage_q_chunk<true, TILE2_NB>(d, d.Q, Qs, 0, min(IC, n));
    __syncthreads();
    const int e2 = 2 * tid, vl = e2 >> d.tp_shift, t = e2 & (TP - 1);
    const int v = tv[vl];
    if (WITH_SOC) {
        if (v >= 0 && t < d.ni) soc_element2<true>(d, v, t, sz, cd, rows, ldr, tl.c_loc);
        __syncthreads();
        for (int row = r0; row < nrow; row += rstep) {      // (every thread updates the entries it staged itself)
            const int f = tl.tri[t0 + row / 3], c = row - 3 * (row / 3);
            const D2 e = ld2(d.E + idxF(d, f, c, ra));
            D2 b = ld2(rows + row * ldr + ra);
            b.v[0] -= e.v[0];
            b.v[1] -= e.v[1];
            st2(rows + row * ldr + ra, b);
        }
        __syncthreads();
    }
    double b[2] = {0.0, 0.0};
    if (v >= 0 && t < n) rhs_value2<true>(d, v, t, r, eps, b, rows, ldr, tl.c_loc);
    xs[vl * TPp + t] = b[0];
    xs[vl * TPp + t + 1] = t + 1 < n ? b[1] : 0.0;
    Dev dt = d;
    dt.VT = tl.VTL;
    modes_from_tile<true, TILE2_NB>(dt, d.Q, xs, Qs, IC, 0, bhat, -1, 0, 1 << 30, true, tv);
}

// T + 1 >= 64: 32 vertices per workgroup, the transform on the matrix cores.
template <bool CARRIED, bool DIV = false>
__global__ __launch_bounds__(RHS_NB) void k_rhs_modes_mfma(Dev d, double r, double eps, double *__restrict__ bhat, int n_rhs, int n_tiles, double sz, double cd, double dv) {
    if ((int)blockIdx.x >= n_rhs) {      // riders: the cone projection of a tile, as in k_rhs_modes
        const int st = xcd_tile(blockIdx.x - n_rhs, d.n_vtiles);
        if (st >= d.n_vtiles) return;
        const int e = threadIdx.x, v = st * d.VT + (e >> d.tp_shift), t = e & (d.TP - 1);
        if (v < d.V && t < d.ni) soc_element<true, CARRIED, DIV>(d, v, t, sz, cd, dv);
        return;
    }
    extern __shared__ double xs_m[];                    // [TM_ROWS][TP + 1]
    const int n = d.T + 1, TP = d.TP, TPp = TP + 1;
    const int tile = xcd_tile(blockIdx.x, n_tiles);     // neighbouring tiles walk the same triangles: one XCD (one L2) for a run of them
    if (tile >= n_tiles) return;
    const int v0 = tile * TM_ROWS;
    for (int e = threadIdx.x; e < TM_ROWS * TP; e += RHS_NB) {
        const int vl = e >> d.tp_shift, t = e & (TP - 1);
        xs_m[vl * TPp + t] = (v0 + vl < d.V && t < n) ? rhs_value<CARRIED, DIV>(d, v0 + vl, t, r, eps, dv) : 0.0;
    }
    __syncthreads();
    modes_from_tile_mfma<RHS_NB / 64>(d, d.Qpad, xs_m, v0, bhat);
}

constexpr size_t RHS_TILES_LDS_MAX = 80 * 1024;      // two workgroups per CU stay resident
// LDS of k_rhs_soc_tiles: Q chunk + the tile of right-hand-side values + the staged triangle rows
static size_t rhs_tiles_lds(const Dev &d, const TileDev &tl) {
    const size_t xs = (size_t)tl.VTL * (d.TP + 1);
    return sizeof(double) * ((size_t)time_modes_chunk(d) * d.TP + xs + (xs & 1) + (size_t)tl.ntri_max * 3 * (d.TP + 2));
}
// DOTS_RHS_TILES=1 (default 0).  Measured in round 3 (profiles/studies/r03_rhs_tiles.txt): the staged launch reads what it was
// built to read (a triangle's rows once per tile) and gives the same iterates bit for bit, but 170 B of LDS per thread leave
// 8 of 32 waves per CU resident, and the launch it replaces runs at 5.9 TB/s of traffic BECAUSE it is at full occupancy:
// torus100k 556 -> 500 it/s, sphere10k 4 960 -> 4 330, knot 10 350 -> 9 100.  Kept as an alternative, off by default.
bool rhs_on_tiles(const Ctx *c) {
    const TileDev &tl = c->tiles;
    return c->rhs_tiles == 1 && tl.n_tiles > 0 && rhs_tiles_lds(c->d, tl) <= RHS_TILES_LDS_MAX;
}

// the right-hand-side (+ projection) launch that launch_rhs would pick can divide the dual arrays as it reads them
bool rhs_divides(const Ctx *c) {
    if (!rhs_writes_modes(c) || c->carry_valid) return false;
    if (time_modes_mfma_ok(c->d)) return true;
    return c->rhs_two && c->d.TP >= 4 && !rhs_on_tiles(c);
}

int launch_rhs(Ctx *c, bool with_soc, double dv) {
    const int g = xcd_grid(c->d.n_vtiles);
    if (rhs_writes_modes(c) && time_modes_mfma_ok(c->d)) {      // (two time columns per lane measured here too: knot63 -1.5 %, torus65k_T127 +1.5 %: not kept)
        const int n_tiles = (c->d.V + TM_ROWS - 1) / TM_ROWS, n_rhs = xcd_grid(n_tiles);
        if (c->carry_valid)
            hipLaunchKernelGGL(k_rhs_modes_mfma<true>, dim3(n_rhs + (with_soc ? g : 0)), dim3(RHS_NB), sizeof(double) * TM_ROWS * (c->d.TP + 1), c->stream, c->d,
                               c->prm.r / c->prm.boundary_scale, c->prm.eps, c->d.cg_p0, n_rhs, n_tiles, c->prm.scale_z, c->prm.const_d, 1.0);
        else if (dv != 0.0)
            hipLaunchKernelGGL((k_rhs_modes_mfma<false, true>), dim3(n_rhs + (with_soc ? g : 0)), dim3(RHS_NB), sizeof(double) * TM_ROWS * (c->d.TP + 1), c->stream, c->d,
                               c->prm.r / c->prm.boundary_scale, c->prm.eps, c->d.cg_p0, n_rhs, n_tiles, c->prm.scale_z, c->prm.const_d, dv);
        else
            hipLaunchKernelGGL(k_rhs_modes_mfma<false>, dim3(n_rhs + (with_soc ? g : 0)), dim3(RHS_NB), sizeof(double) * TM_ROWS * (c->d.TP + 1), c->stream, c->d,
                               c->prm.r / c->prm.boundary_scale, c->prm.eps, c->d.cg_p0, n_rhs, n_tiles, c->prm.scale_z, c->prm.const_d, 1.0);
    }
    else if (rhs_writes_modes(c) && c->rhs_two && c->d.TP >= 4 && c->carry_valid)      // the corners' shares come from the last steps-2+3 launch
        hipLaunchKernelGGL(k_rhs_modes2<true>, dim3(with_soc ? 2 * g : g), dim3(RHS_NB2), time_modes_tile_lds(c->d), c->stream, c->d, c->prm.r / c->prm.boundary_scale,
                           c->prm.eps, c->d.cg_p0, time_modes_chunk(c->d), g, c->prm.scale_z, c->prm.const_d, (const int *)nullptr, 1.0);
    else if (dv != 0.0)      // a penalty update is pending: the dual arrays are divided as they are read (rhs_divides told the caller this launch can)
        hipLaunchKernelGGL((k_rhs_modes2<false, true>), dim3(with_soc ? 2 * g : g), dim3(RHS_NB2), time_modes_tile_lds(c->d), c->stream, c->d, c->prm.r / c->prm.boundary_scale,
                           c->prm.eps, c->d.cg_p0, time_modes_chunk(c->d), g, c->prm.scale_z, c->prm.const_d, (const int *)nullptr, dv);
    else if (rhs_writes_modes(c) && c->rhs_two && rhs_on_tiles(c)) {     // patch tiles, triangle rows staged in LDS (large meshes)
        const TileDev &tl = c->tiles;
        const int IC = time_modes_chunk(c->d), gt = xcd_grid(tl.n_tiles);
        const size_t lds = rhs_tiles_lds(c->d, tl);
        static bool raised = false;      // more than the default 64 KB of dynamic LDS needs the attribute (once per process)
        if (!raised) {
            DOTS_HIP(hipFuncSetAttribute((const void *)k_rhs_soc_tiles<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)RHS_TILES_LDS_MAX));
            DOTS_HIP(hipFuncSetAttribute((const void *)k_rhs_soc_tiles<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)RHS_TILES_LDS_MAX));
            raised = true;
        }
        if (with_soc) hipLaunchKernelGGL((k_rhs_soc_tiles<true>), dim3(gt), dim3(TILE2_NB), lds, c->stream, c->d, tl, c->prm.r / c->prm.boundary_scale, c->prm.eps, c->d.cg_p0,
                                         IC, c->prm.scale_z, c->prm.const_d);
        else hipLaunchKernelGGL((k_rhs_soc_tiles<false>), dim3(gt), dim3(TILE2_NB), lds, c->stream, c->d, tl, c->prm.r / c->prm.boundary_scale, c->prm.eps, c->d.cg_p0,
                                IC, c->prm.scale_z, c->prm.const_d);
    }
    else if (rhs_writes_modes(c) && c->rhs_two && c->d.TP >= 4)      // two time columns per lane (16-byte accesses)
        hipLaunchKernelGGL(k_rhs_modes2<false>, dim3(with_soc ? 2 * g : g), dim3(RHS_NB2), time_modes_tile_lds(c->d), c->stream, c->d, c->prm.r / c->prm.boundary_scale,
                           c->prm.eps, c->d.cg_p0, time_modes_chunk(c->d), g, c->prm.scale_z, c->prm.const_d,
                           (c->rhs_tiles == 2 && c->tiles.n_tiles > 0) ? c->tiles.vertex : (const int *)nullptr, 1.0);
    else if (rhs_writes_modes(c))
        hipLaunchKernelGGL(k_rhs_modes, dim3(with_soc ? 2 * g : g), dim3(RHS_NB), time_modes_tile_lds(c->d), c->stream, c->d, c->prm.r / c->prm.boundary_scale,
                           c->prm.eps, c->d.cg_p0, time_modes_chunk(c->d), g, c->prm.scale_z, c->prm.const_d);
    else if (c->carry_valid)      // (a time slab of the direct solver's iteration)
        hipLaunchKernelGGL(k_rhs<true>, dim3(with_soc ? g + g * (TILE_ELEMS / BLOCK) : g), dim3(BLOCK), 0, c->stream, c->d, c->prm.r / c->prm.boundary_scale,
                           c->prm.eps, g, c->prm.scale_z, c->prm.const_d);
    else
        hipLaunchKernelGGL(k_rhs<false>, dim3(with_soc ? g + g * (TILE_ELEMS / BLOCK) : g), dim3(BLOCK), 0, c->stream, c->d, c->prm.r / c->prm.boundary_scale,
                           c->prm.eps, g, c->prm.scale_z, c->prm.const_d);
    DOTS_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------
// Steps 2 + 3 fused                            (reference: solver_socp.py:709-722, :1044-1065)
// Vertex part, thread (v, t < T):
//     dphi = (phi[t+1] - phi[t]) / h
//     A = (dphi + mu)/a2 + (a1/a2)(z_end + beta_end - z_fst - beta_fst);  lambda_c = cr/(1+cr) (dphi + mu - A)
//     mu += tau (dphi - A - lambda_c);  beta_fst += tau (z_fst + sz A - d);  beta_end += tau (z_end - sz A - d)
// Triangle part, thread (f, c, t <= T):
//     gx = sum_k hat[f][k][c] phi[tri[f][k]][t]                       (grad_space, 3 coalesced row gathers)
//     B  = (gx + E + sz/sqrt3 * (sum_k (z_mid+beta_mid)[t][s=0] + sum_k (z_mid+beta_mid)[t-1][s=1])) / diag_b[t]
//     E += tau (gx - B)
//     beta_mid[t][0][k] += tau (z_mid - sz/sqrt3 B[t]);  beta_mid[t-1][1][k] += tau (z_mid - sz/sqrt3 B[t])
// Every beta_mid entry is attached to the node whose B it is compared with, so the thread that
// computes B[t] updates them without any exchange: z_mid and beta_mid are read once, beta_mid written once.
// ------------------------------------------------------------------------------------------
// QONLY: only the (q, lambda_c) closed form (the reference's "Step 0" of is_palm = True, solver_socp.py:668-672):
// A, B and lambda_c are written, no multiplier moves.
// The fused KKT sums of the vertex part (DOTS_STEP_KKT_SUMS): the slots of kkt_vertex_body that need nothing beyond this element
constexpr int KV_N = 10;
constexpr int KV_SLOT[KV_N] = {V_DPHI2, V_A2, V_LAM2, V_RESMU2, V_RFST2, V_REND2, V_MU2, V_AUX1_2, V_MUAUX1_2, V_CONG_RES2};
// ... and of the triangle part
constexpr int KF_N = 7;
constexpr int KF_SLOT[KF_N] = {F_DX2, F_B2, F_RESE2, F_E2, F_AUX2_2, F_EAUX2_2, F_RMID2};

// KKT: the expressions of kkt_vertex_body (kernels_kkt.hip) on the values this element has just computed, accumulated in ks[KV_N]
template <bool QONLY, int NB = BLOCK, bool KKT = false, bool DIV = false>
__device__ __forceinline__ void q_lambda_vertex_tile(const Dev &d, int tile, double sz, double cd, double cr, double tau, const KktArgs *ka = nullptr,
                                                     double *ks = nullptr, double dv = 1.0) {
    const int v0 = tile * d.VT;
    const double a1 = sz * (1.0 + cr);
    const double a2 = 1.0 + 2.0 * sz * a1;
    const double ia2 = 1.0 / a2, a12 = a1 / a2, cl = cr / (1.0 + cr), ih = 1.0 / d.h;
    for (int e = threadIdx.x; e < TILE_ELEMS; e += NB) {
        const int v = v0 + (e >> d.tp_shift), t = e & (d.TP - 1);
        if (v >= d.V || t >= d.ni) continue;
        const int iv = idxV(d, v, t);
        const double dphi = (next_node(d, d.phi, d.phi_hi, v, t) - d.phi[iv]) * ih;
        const double zf = d.zf[iv], ze = d.ze[iv];
        double mu = d.mu[iv], bf = d.bf[iv], be = d.be[iv];
        if (DIV) { mu /= dv; bf /= dv; be /= dv; }      // (a pending penalty update: see soc_element)
        const double memo = dphi + mu;
        const double a = ia2 * memo + a12 * (ze + be - zf - bf);
        const double lc = cl * (memo - a);
        d.A[iv] = a;
        d.lam[iv] = lc;
        if (QONLY) continue;
        const double mun = mu + tau * (dphi - a - lc), bfn = bf + tau * (zf + sz * a - cd), ben = be + tau * (ze - sz * a - cd);
        d.mu[iv] = mun;
        d.bf[iv] = bfn;
        d.be[iv] = ben;
        if (KKT) {
            const double m = d.mass_v[v];
            const double rm = dphi - a - lc;
            ks[0] += dphi * dphi * m;
            ks[1] += a * a * m;
            ks[2] += lc * lc * m;
            ks[3] += rm * rm * m;
            const double rf = zf + ka->sz * a - ka->cd, re = ze - ka->sz * a - ka->cd;
            ks[4] += rf * rf * m;
            ks[5] += re * re * m;
            ks[6] += mun * mun * m;
            const double x1 = ka->sz * (ben - bfn);
            ks[7] += x1 * x1 * m;
            ks[8] += (mun + x1) * (mun + x1) * m;
            const double res = ka->cong * ((ka->ds * ka->r) * mun) - ka->ps * lc;
            ks[9] += res * res * m;
        }
    }
}

// ZMODE 0: z_mid is read from memory; 1: rebuilt from the cone multiplier (same expression as the projection
// kernel: z = (lambda / D) * (D * (sz/sqrt3 * B_old - beta_mid))) and stored; 2: rebuilt, not stored.
// Workgroups [0, nf8) do the triangle part; with VERTEX_TOO the vertex part (independent of it: different
// arrays) rides in the same launch as workgroups [nf8, nf8 + nv8).
template <int ZMODE, bool QONLY = false>
__global__ __launch_bounds__(BLOCK) void k_q_lambda_mult_triangle(Dev d, double sz, double tau, int nf8, double cd, double cr) {
    constexpr int SUB = TILE_ELEMS / BLOCK;     // one element per thread: a workgroup takes a quarter of a triangle tile
    if ((int)blockIdx.x >= nf8 * SUB) {
        const int vt = xcd_tile(blockIdx.x - nf8 * SUB, d.n_vtiles);
        if (vt < d.n_vtiles) q_lambda_vertex_tile<QONLY>(d, vt, sz, cd, cr, tau);
        return;
    }
    const int tile = xcd_tile(blockIdx.x % nf8, d.n_ftiles);
    if (tile >= d.n_ftiles) return;
    const int row0 = tile * d.FT;
    const double sB = sz * INV_SQRT3;
    const double diag_in = 1.0 + 2.0 * sz * sz, diag_bd = 1.0 + sz * sz;
    for (int e = (blockIdx.x / nf8) * BLOCK + threadIdx.x; e < TILE_ELEMS; e += TILE_ELEMS) {
        const int row = row0 + (e >> d.tp_shift), t = e & (d.TP - 1);
        if (row >= 3 * d.F || t >= d.nl) continue;
        const int f = row / 3, c = row - 3 * f;
        // All loads first and unconditional: the compiler then issues them back to back instead of one wait per guarded
        // load.  Both corner entries of this node sit in ITS column (idxM), whether their interval exists or not (then the
        // slot holds a zero that is masked below); the multiplier of interval t - 1 comes from the previous time slab's
        // halo when this is the slab's first node.
        const bool has0 = t < d.ni, has1 = has_prev_interval(d, t);
        const int64_t ie = idxF(d, f, c, t);
        int vk[3];
        double hk[3], Dk[3], phik[3], l0[3], l1[3], b0[3], b1[3], z0[3], z1[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            vk[k] = d.tri[f * 3 + k];
            hk[k] = d.hat[(f * 3 + k) * 3 + c];
            Dk[k] = ZMODE ? d.fk_D[f * 3 + k] : 1.0;
        }
        const double Bold = ZMODE ? d.B[ie] : 0.0;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            phik[k] = d.phi[idxV(d, vk[k], t)];
            b0[k] = d.bm[idxM(d, f * 3 + k, 0, c, t)];
            b1[k] = d.bm[idxM(d, f * 3 + k, 1, c, t - 1)];
            if (ZMODE) {
                l0[k] = d.lamc[idxV(d, vk[k], t)];
                const double *pl = t > 0 ? d.lamc + idxV(d, vk[k], t - 1) : (has1 ? d.lamc_lo + vk[k] : d.lamc + idxV(d, vk[k], t));
                l1[k] = *pl;
            } else {
                z0[k] = d.zm[idxM(d, f * 3 + k, 0, c, t)];
                z1[k] = d.zm[idxM(d, f * 3 + k, 1, c, t - 1)];
            }
        }
        double gx = 0.0;
#pragma unroll
        for (int k = 0; k < 3; ++k) gx += hk[k] * phik[k];
        double S = 0.0;
        const double sBold = sB * Bold;   // both pre-images of this thread's corners use B_old at ITS node
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (ZMODE) {
                z0[k] = (l0[k] / Dk[k]) * (Dk[k] * (sBold - b0[k]));
                z1[k] = (l1[k] / Dk[k]) * (Dk[k] * (sBold - b1[k]));
            }
            if (!has0) z0[k] = b0[k] = 0.0;
            if (!has1) z1[k] = b1[k] = 0.0;
            if (ZMODE == 1) {
                if (has0) d.zm[idxM(d, f * 3 + k, 0, c, t)] = z0[k];
                if (has1) d.zm[idxM(d, f * 3 + k, 1, c, t - 1)] = z1[k];
            }
            S += (z0[k] + b0[k]) + (z1[k] + b1[k]);
        }
        const double Eo = d.E[ie];
        const double Bn = (gx + Eo + sB * S) / ((first_node(d, t) || last_node(d, t)) ? diag_bd : diag_in);
        d.B[ie] = Bn;
        if (QONLY) continue;
        d.E[ie] = Eo + tau * (gx - Bn);
        const double sBn = sB * Bn;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (has0) d.bm[idxM(d, f * 3 + k, 0, c, t)] = b0[k] + tau * (z0[k] - sBn);
            if (has1) d.bm[idxM(d, f * 3 + k, 1, c, t - 1)] = b1[k] + tau * (z1[k] - sBn);
        }
    }
}

// The same kernel with TWO consecutive nodes (t, t + 1; t even) per lane: every array of the triangle part is read and written in
// the column of the node (idxM), so both nodes of a lane sit in one aligned 16-byte word of every row; the mesh constants of
// (f, c) are loaded once for both.  Half the waves for the same bytes: these launches are latency-bound at full occupancy on
// the small meshes (time ~ waves x chain / resident waves).  Element for element the arithmetic of k_q_lambda_mult_triangle.
// (118 VGPRs = 4 waves per SIMD.  Forcing 5 / 6 with amdgpu_waves_per_eu spills 44 / 132 bytes per lane and loses: torus100k 571 -> 553 / 472 it/s,
// knot 10 530 -> 9 750 / 7 980: profiles/studies/r03_steps23_forced_occupancy.txt)
// The lane of k_q_lambda_mult_triangle2 (below) as a function, statement for statement the same arithmetic; that kernel keeps its
// own copy: the register allocation of its ZMODE = 2 instantiation (118 VGPRs = 4 waves per SIMD) did not survive the call.
// CARRY (k_q_lambda_mult_carry): the lane also forms, from the registers that hold the NEW B, E and beta_mid, what the next
// iteration's right-hand side and cone projection would gather from memory -- per corner k of its triangle and node u its
// xyz component's share of  |D (sB B - beta_mid)|^2  (both halves: s = 0 of interval t + u, s = 1 of interval t + u - 1) and of
// area * hat . (B - E) -- and leaves them in LDS (xl: this lane's column, stride NBX) for the fold over xyz.
constexpr int CARRY_NB = 192;         // 3 wavefronts: 2 * 192 / TP rows = whole triangles for every pitch <= 128
constexpr int CARRY_VALUES = 18;      // per lane: 3 corners x (2 halves + 1 divergence share) x 2 nodes
// KKT: the lane also accumulates the sums of kkt_triangle_body2 that need no gather (ks[KF_N]; sz = scale_factor_z).
// BMNT: beta_mid is loaded and stored with the non-temporal hint (Ctx::bm_nt: where the factor can live in the 256 MB Infinity Cache if the
// 36 T F values of beta_mid that stream through every iteration do not displace it -- sphere10k: 5 130-5 190 -> 5 490-5 500 it/s; where
// everything fits (knot) the hint costs 2 %, where nothing does (torus100k) it changes nothing: profiles/studies/r04_nontemporal.txt)
template <int ZMODE, bool QONLY, bool CARRY, bool KKT = false, bool DIV = false, bool BMNT = false>
__device__ __forceinline__ void ql2_lane(const Dev &d, int f, int c, int t, double sB, double diag_in, double diag_bd, double tau, double *xl = nullptr,
                                         double sz = 0.0, double *ks = nullptr, double dv = 1.0) {
    const bool two = t + 1 < d.nl;                  // the second node exists (always, unless the slab holds an odd number of nodes)
    const int64_t ie = idxF(d, f, c, t);
    int vk[3];
    double hk[3], Dk[3];
    D2 phik[3], l0[3], b0[3], b1[3], z0[3], z1[3];
    double lm1[3];                                  // the multiplier of interval t - 1 (that of interval t is l0[.].v[0])
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        vk[k] = d.tri[f * 3 + k];
        hk[k] = d.hat[(f * 3 + k) * 3 + c];
        Dk[k] = (ZMODE || CARRY) ? d.fk_D[f * 3 + k] : 1.0;
    }
    const double area = (CARRY || KKT) ? d.area_f[f] : 0.0;
    const bool has1_0 = has_prev_interval(d, t);
    D2 Bold = {{0.0, 0.0}};
    if (ZMODE) Bold = ld2(d.B + ie);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        phik[k] = ld2(d.phi + idxV(d, vk[k], t));
        b0[k] = BMNT ? ld2_nt(d.bm + idxM(d, f * 3 + k, 0, c, t)) : ld2(d.bm + idxM(d, f * 3 + k, 0, c, t));
        b1[k] = BMNT ? ld2_nt(d.bm + idxM(d, f * 3 + k, 1, c, t - 1)) : ld2(d.bm + idxM(d, f * 3 + k, 1, c, t - 1));
        if (DIV) {      // (a pending penalty update: see soc_element)
            b0[k].v[0] /= dv; b0[k].v[1] /= dv;
            b1[k].v[0] /= dv; b1[k].v[1] /= dv;
        }
        if (ZMODE) {
            l0[k] = ld2(d.lamc + idxV(d, vk[k], t));
            const double *pl = t > 0 ? d.lamc + idxV(d, vk[k], t - 1) : (has1_0 ? d.lamc_lo + vk[k] : d.lamc + idxV(d, vk[k], t));
            lm1[k] = *pl;
        } else {
            z0[k] = ld2(d.zm + idxM(d, f * 3 + k, 0, c, t));
            z1[k] = ld2(d.zm + idxM(d, f * 3 + k, 1, c, t - 1));
        }
    }
    D2 Eo = ld2(d.E + ie);
    if (DIV) { Eo.v[0] /= dv; Eo.v[1] /= dv; }
    D2 Bn, En, n0[3], n1[3];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int tu = t + u;
        const bool has0 = tu < d.ni, has1 = has_prev_interval(d, tu);
        double gx = 0.0;
#pragma unroll
        for (int k = 0; k < 3; ++k) gx += hk[k] * phik[k].v[u];
        double S = 0.0;
        const double sBold = sB * Bold.v[u];
        double zz0[3], zz1[3], bb0[3], bb1[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            bb0[k] = b0[k].v[u];
            bb1[k] = b1[k].v[u];
            if (ZMODE) {
                const double l1 = u == 0 ? lm1[k] : l0[k].v[0];
                zz0[k] = (l0[k].v[u] / Dk[k]) * (Dk[k] * (sBold - bb0[k]));
                zz1[k] = (l1 / Dk[k]) * (Dk[k] * (sBold - bb1[k]));
            } else {
                zz0[k] = z0[k].v[u];
                zz1[k] = z1[k].v[u];
            }
            if (!has0) zz0[k] = bb0[k] = 0.0;
            if (!has1) zz1[k] = bb1[k] = 0.0;
            z0[k].v[u] = zz0[k];
            z1[k].v[u] = zz1[k];
            S += (zz0[k] + bb0[k]) + (zz1[k] + bb1[k]);
        }
        const double bn = (gx + Eo.v[u] + sB * S) / ((first_node(d, tu) || last_node(d, tu)) ? diag_bd : diag_in);
        Bn.v[u] = bn;
        En.v[u] = Eo.v[u] + tau * (gx - bn);
        const double sBn = sB * bn;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            // slots whose interval does not exist keep what they hold (zeros): the scalar kernel does not store them either
            n0[k].v[u] = has0 ? bb0[k] + tau * (zz0[k] - sBn) : b0[k].v[u];
            n1[k].v[u] = has1 ? bb1[k] + tau * (zz1[k] - sBn) : b1[k].v[u];
        }
        if (KKT && tu < d.nl) {      // kkt_triangle_body2, conditions 0, 3 and the z_mid part of 1, on the values just computed
            ks[0] += gx * gx * area;
            ks[1] += bn * bn * area;
            ks[2] += (gx - bn) * (gx - bn) * area;
            const double en = En.v[u];
            ks[3] += en * en * area;
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (has0) s0 += n0[k].v[u];
                if (has1) s1 += n1[k].v[u];
            }
            const double x2 = sB * (s0 + s1);
            ks[4] += x2 * x2 * area;
            ks[5] += (en + x2) * (en + x2) * area;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (has0) {
                    const double q = sz * (zz0[k] - sBn);
                    ks[6] += q * q * area;
                }
                if (has1) {
                    const double q = sz * (zz1[k] - sBn);
                    ks[6] += q * q * area;
                }
            }
        }
        if (CARRY) {      // the expressions of soc_element2 / rhs_value2 on the values those will find in memory
            const bool node = tu < d.nl;
            const double de = bn - En.v[u];
            if (KKT) {    // ... and of Dual(alpha)'s gather (kkt_vertex_body2: sum over the corners of area * hat . E)
#pragma unroll
                for (int k = 0; k < 3; ++k) xl[(CARRY_VALUES + k * 2 + u) * CARRY_NB] = node ? (hk[k] * area) * En.v[u] : 0.0;
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                xl[(k * 6 + u) * CARRY_NB] = has0 ? soc_w2(Dk[k], sBn, n0[k].v[u]) : 0.0;               // s = 0 half of interval tu
                xl[(k * 6 + 2 + u) * CARRY_NB] = (has1 && node) ? soc_w2(Dk[k], sBn, n1[k].v[u]) : 0.0;  // s = 1 half of interval tu - 1
                xl[(k * 6 + 4 + u) * CARRY_NB] = node ? (hk[k] * area) * de : 0.0;                       // div_x share at node tu
            }
        }
    }
    if (two) {
        if (ZMODE == 1) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                // (entries of intervals that do not exist are stored as the zeros computed above; their slots are never read as data)
                st2_nt(d.zm + idxM(d, f * 3 + k, 0, c, t), z0[k]);      // (not read again inside the loop)
                st2_nt(d.zm + idxM(d, f * 3 + k, 1, c, t - 1), z1[k]);
            }
        }
        st2(d.B_st + ie, Bn);
        if (QONLY) return;
        st2(d.E + ie, En);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (BMNT) {
                st2_nt(d.bm_st + idxM(d, f * 3 + k, 0, c, t), n0[k]);
                st2_nt(d.bm_st + idxM(d, f * 3 + k, 1, c, t - 1), n1[k]);
            } else {
                st2(d.bm_st + idxM(d, f * 3 + k, 0, c, t), n0[k]);
                st2(d.bm_st + idxM(d, f * 3 + k, 1, c, t - 1), n1[k]);
            }
        }
    } else {      // only the first node of the pair exists: element-wise stores
        const bool has0 = t < d.ni, has1 = has1_0;
        if (ZMODE == 1) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (has0) d.zm[idxM(d, f * 3 + k, 0, c, t)] = z0[k].v[0];
                if (has1) d.zm[idxM(d, f * 3 + k, 1, c, t - 1)] = z1[k].v[0];
            }
        }
        d.B_st[ie] = Bn.v[0];
        if (QONLY) return;
        d.E[ie] = En.v[0];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (has0) d.bm_st[idxM(d, f * 3 + k, 0, c, t)] = n0[k].v[0];
            if (has1) d.bm_st[idxM(d, f * 3 + k, 1, c, t - 1)] = n1[k].v[0];
        }
    }
}

template <int ZMODE, bool QONLY = false>
__global__ __launch_bounds__(BLOCK) void k_q_lambda_mult_triangle2(Dev d, double sz, double tau, int nf8, double cd, double cr) {
    constexpr int SUB = TILE_ELEMS / (2 * BLOCK);     // two elements per thread: a workgroup takes half of a triangle tile
    if ((int)blockIdx.x >= nf8 * SUB) {
        const int vt = xcd_tile(blockIdx.x - nf8 * SUB, d.n_vtiles);
        if (vt < d.n_vtiles) q_lambda_vertex_tile<QONLY>(d, vt, sz, cd, cr, tau);
        return;
    }
    const int tile = xcd_tile(blockIdx.x % nf8, d.n_ftiles);
    if (tile >= d.n_ftiles) return;
    const int row0 = tile * d.FT;
    const double sB = sz * INV_SQRT3;
    const double diag_in = 1.0 + 2.0 * sz * sz, diag_bd = 1.0 + sz * sz;
    const int e = ((blockIdx.x / nf8) * BLOCK + threadIdx.x) * 2;          // first of the lane's two elements in the tile
    const int row = row0 + (e >> d.tp_shift), t = e & (d.TP - 1);
    if (row >= 3 * d.F || t >= d.nl) return;
    const int f = row / 3, c = row - 3 * f;
    const bool two = t + 1 < d.nl;                  // the second node exists (always, unless the slab holds an odd number of nodes)
    const int64_t ie = idxF(d, f, c, t);
    int vk[3];
    double hk[3], Dk[3];
    D2 phik[3], l0[3], b0[3], b1[3], z0[3], z1[3];
    double lm1[3];                                  // the multiplier of interval t - 1 (that of interval t is l0[.].v[0])
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        vk[k] = d.tri[f * 3 + k];
        hk[k] = d.hat[(f * 3 + k) * 3 + c];
        Dk[k] = ZMODE ? d.fk_D[f * 3 + k] : 1.0;
    }
    const bool has1_0 = has_prev_interval(d, t);
    D2 Bold = {{0.0, 0.0}};
    if (ZMODE) Bold = ld2(d.B + ie);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        phik[k] = ld2(d.phi + idxV(d, vk[k], t));
        b0[k] = ld2(d.bm + idxM(d, f * 3 + k, 0, c, t));
        b1[k] = ld2(d.bm + idxM(d, f * 3 + k, 1, c, t - 1));
        if (ZMODE) {
            l0[k] = ld2(d.lamc + idxV(d, vk[k], t));
            const double *pl = t > 0 ? d.lamc + idxV(d, vk[k], t - 1) : (has1_0 ? d.lamc_lo + vk[k] : d.lamc + idxV(d, vk[k], t));
            lm1[k] = *pl;
        } else {
            z0[k] = ld2(d.zm + idxM(d, f * 3 + k, 0, c, t));
            z1[k] = ld2(d.zm + idxM(d, f * 3 + k, 1, c, t - 1));
        }
    }
    const D2 Eo = ld2(d.E + ie);
    D2 Bn, En, n0[3], n1[3];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int tu = t + u;
        const bool has0 = tu < d.ni, has1 = has_prev_interval(d, tu);
        double gx = 0.0;
#pragma unroll
        for (int k = 0; k < 3; ++k) gx += hk[k] * phik[k].v[u];
        double S = 0.0;
        const double sBold = sB * Bold.v[u];
        double zz0[3], zz1[3], bb0[3], bb1[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            bb0[k] = b0[k].v[u];
            bb1[k] = b1[k].v[u];
            if (ZMODE) {
                const double l1 = u == 0 ? lm1[k] : l0[k].v[0];
                zz0[k] = (l0[k].v[u] / Dk[k]) * (Dk[k] * (sBold - bb0[k]));
                zz1[k] = (l1 / Dk[k]) * (Dk[k] * (sBold - bb1[k]));
            } else {
                zz0[k] = z0[k].v[u];
                zz1[k] = z1[k].v[u];
            }
            if (!has0) zz0[k] = bb0[k] = 0.0;
            if (!has1) zz1[k] = bb1[k] = 0.0;
            z0[k].v[u] = zz0[k];
            z1[k].v[u] = zz1[k];
            S += (zz0[k] + bb0[k]) + (zz1[k] + bb1[k]);
        }
        const double bn = (gx + Eo.v[u] + sB * S) / ((first_node(d, tu) || last_node(d, tu)) ? diag_bd : diag_in);
        Bn.v[u] = bn;
        En.v[u] = Eo.v[u] + tau * (gx - bn);
        const double sBn = sB * bn;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            // slots whose interval does not exist keep what they hold (zeros): the scalar kernel does not store them either
            n0[k].v[u] = has0 ? bb0[k] + tau * (zz0[k] - sBn) : b0[k].v[u];
            n1[k].v[u] = has1 ? bb1[k] + tau * (zz1[k] - sBn) : b1[k].v[u];
        }
    }
    if (two) {
        if (ZMODE == 1) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                // (entries of intervals that do not exist are stored as the zeros computed above; their slots are never read as data)
                st2(d.zm + idxM(d, f * 3 + k, 0, c, t), z0[k]);
                st2(d.zm + idxM(d, f * 3 + k, 1, c, t - 1), z1[k]);
            }
        }
        st2(d.B + ie, Bn);
        if (QONLY) return;
        st2(d.E + ie, En);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            st2(d.bm + idxM(d, f * 3 + k, 0, c, t), n0[k]);
            st2(d.bm + idxM(d, f * 3 + k, 1, c, t - 1), n1[k]);
        }
    } else {      // only the first node of the pair exists: element-wise stores
        const bool has0 = t < d.ni, has1 = has1_0;
        if (ZMODE == 1) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (has0) d.zm[idxM(d, f * 3 + k, 0, c, t)] = z0[k].v[0];
                if (has1) d.zm[idxM(d, f * 3 + k, 1, c, t - 1)] = z1[k].v[0];
            }
        }
        d.B[ie] = Bn.v[0];
        if (QONLY) return;
        d.E[ie] = En.v[0];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (has0) d.bm[idxM(d, f * 3 + k, 0, c, t)] = n0[k].v[0];
            if (has1) d.bm[idxM(d, f * 3 + k, 1, c, t - 1)] = n1[k].v[0];
        }
    }
}

// Steps 2+3 that also CARRY the next iteration's gathers (DOTS_STEP_CARRY; one GPU, pitch <= 128).  A workgroup of 192 lanes takes
// whole triangles (the three xyz rows of a triangle are TP / 2 lanes apart), the lanes exchange their shares through LDS and the
// lane of component c folds corner k = c over xyz -- (x + y) + z, the association of sum3 -- and stores
//     cn_sq[j][0][interval]   the s = 0 half of |D (sB B - beta_mid)|^2 of corner-list entry j,
//     cn_sq[j][1][interval]   the s = 1 half (formed at node interval + 1: it comes from the next lane of the row),
//     cn_g[j][node]           area * hat_k . (B - E),
// j = cpos[f * 3 + k]: the rows of a vertex's corners are CONTIGUOUS, so the next right-hand-side / projection launch streams
// them (soc_element2 / rhs_value2 <CARRIED>) instead of walking index lists into B, E and beta_mid.
// Replaces one of the three passes over beta_mid per iteration (solver_socp.py:997-1017 reads what :716-722 just wrote).
// 148-152 VGPRs = 3 waves per SIMD by themselves; the instantiation with the KKT sums is held there (170 otherwise).
// (A/B: -DDOTS_CARRY_WAVES=4 caps the registers at 128, 20 spilled: knot 10 900 -> 9 500 it/s, torus100k 630 -> 585)
#ifndef DOTS_CARRY_WAVES
#define DOTS_CARRY_WAVES 3
#endif
#define CARRY_OCCUPANCY __attribute__((amdgpu_waves_per_eu(DOTS_CARRY_WAVES)))
// this workgroup's partial sums (thread 0 holds the totals) into the fused-KKT buffers: slot SLOT[i] of block `bid` of `nblk`
template <int N>
__device__ __forceinline__ void store_fused(const double (&v)[N], const int (&slot)[N], int first, double *__restrict__ part, int nblk, int bid) {
#pragma unroll
    for (int i = 0; i < N; ++i) part[(int64_t)(slot[i] - first) * nblk + bid] = v[i];
}
template <int ZMODE, bool KKT = false, bool DIV = false, bool BMNT = false>
__global__ __launch_bounds__(CARRY_NB) CARRY_OCCUPANCY void k_q_lambda_mult_carry(Dev d, double sz, double tau, int n_fwg, int tri_per_wg, double cd, double cr,
                                                                                  KktArgs ka, KktFused kf, double dv, int emit) {
    // emit: bit 0 the gathers of the next right-hand side / projection (cn_sq, cn_g: DOTS_STEP_CARRY), bit 1 those of this
    // iterate's Dual(alpha) residual (cn_e: DOTS_STEP_KKT_SUMS, KKT instantiations only)
    __shared__ double xs[(CARRY_VALUES + (KKT ? 6 : 0)) * CARRY_NB];
    const int tid = threadIdx.x;
    if ((int)blockIdx.x >= n_fwg) {
        const int vb = blockIdx.x - n_fwg, vt = xcd_tile(vb, d.n_vtiles);
        double ks[KV_N] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        if (vt < d.n_vtiles) q_lambda_vertex_tile<false, CARRY_NB, KKT, DIV>(d, vt, sz, cd, cr, tau, &ka, ks, dv);
        if (KKT) {
            block_sum<KV_N, CARRY_NB / 64>(ks, xs);
            if (tid == 0) store_fused<KV_N>(ks, KV_SLOT, 0, kf.part_v, kf.nv, vb);
        }
        return;
    }
    const int wg = xcd_tile(blockIdx.x, (d.F + tri_per_wg - 1) / tri_per_wg);
    const int e = 2 * tid, lr = e >> d.tp_shift, t = e & (d.TP - 1);      // local row (3 per triangle), first node
    const int f = wg * tri_per_wg + lr / 3, c = lr - 3 * (lr / 3);
    const bool active = f < d.F && t < d.nl;      // (a whole triangle is active or not; so is a column over its three rows)
#ifndef DOTS_CARRY_CPOS_LATE
    const int j = active ? d.cpos[f * 3 + c] : 0; // row of this lane's corner k = c in the carried arrays (loaded with the lane's other constants)
#endif
    double kt[KF_N] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (active) ql2_lane<ZMODE, false, true, KKT, DIV, BMNT>(d, f, c, t, sz * INV_SQRT3, 1.0 + 2.0 * sz * sz, 1.0 + sz * sz, tau, xs + tid, sz, kt, dv);
    __syncthreads();
    const int L = d.TP >> 1;                      // lanes per row
    const int t0 = tid - c * L;                   // the lane of component 0 of this triangle and column
    double q[6];                                  // corner k = c: s = 0 halves (nodes t, t + 1), s = 1 halves, divergence shares
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const double *x = xs + (c * 6 + i) * CARRY_NB + t0;
        q[i] = active ? sum3(x[0], x[L], x[2 * L]) : 0.0;
    }
    // q[2], q[3]: s = 1 halves of the intervals t - 1 and t.  Stored by interval: (t, t + 1) = (own q[3], the next lane's q[2]);
    // the row's last lane (t = TP - 2) would need interval TP - 1 >= T, which does not exist.
    const double nxt = __shfl_down(q[2], 1, 64);
    if (active) {
#ifdef DOTS_CARRY_CPOS_LATE
        const int j = d.cpos[f * 3 + c];
#endif
        if (emit & 1) {
            st2_nt(d.cn_sq + ((int64_t)(2 * j) << d.tp_shift) + t, D2{{q[0], q[1]}});
            st2_nt(d.cn_sq + ((int64_t)(2 * j + 1) << d.tp_shift) + t, D2{{q[3], t + 2 < d.TP ? nxt : 0.0}});
            st2_nt(d.cn_g + ((int64_t)j << d.tp_shift) + t, D2{{q[4], q[5]}});
            if (t == 0 && has_prev_interval(d, 0)) d.cn_lo[j] = q[2];      // (time slab: the half of the previous slab's last interval formed here)
        }
        if (KKT && (emit & 2)) {
            const double *x0 = xs + (CARRY_VALUES + c * 2) * CARRY_NB + t0, *x1 = x0 + CARRY_NB;
            st2_nt(d.cn_e + ((int64_t)j << d.tp_shift) + t, D2{{sum3(x0[0], x0[L], x0[2 * L]), sum3(x1[0], x1[L], x1[2 * L])}});
        }
    }
    if (KKT) {
        __syncthreads();      // (the exchange values in xs have been read)
        block_sum<KF_N, CARRY_NB / 64>(kt, xs);
        if (tid == 0) store_fused<KF_N>(kt, KF_SLOT, N_VSUMS, kf.part_f, kf.nf, blockIdx.x);
    }
}

// z_mid on demand (Ctx::zmid_deferred): in place on z_mid's storage, which holds the beta_mid the last projection read;
// Bold: the B it read.  One node per lane; the expression of the steps-2+3 kernels (ZMODE 1), entry for entry.
__global__ __launch_bounds__(BLOCK) void k_rebuild_zmid(Dev d, const double *__restrict__ Bold, double sz, double dv) {
    const int tile = xcd_tile(blockIdx.x, d.n_ftiles);
    if (tile >= d.n_ftiles) return;
    const double sB = sz * INV_SQRT3;
    for (int e = threadIdx.x; e < TILE_ELEMS; e += BLOCK) {
        const int row = tile * d.FT + (e >> d.tp_shift), t = e & (d.TP - 1);
        if (row >= 3 * d.F || t >= d.nl) continue;
        const int f = row / 3, c = row - 3 * f;
        const bool has0 = t < d.ni, has1 = has_prev_interval(d, t);
        const double sBold = sB * Bold[idxF(d, f, c, t)];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int vk = d.tri[f * 3 + k];
            const double Dk = d.fk_D[f * 3 + k];
            if (has0) {
                double b = d.zm[idxM(d, f * 3 + k, 0, c, t)];
                if (dv != 0.0) b /= dv;
                d.zm[idxM(d, f * 3 + k, 0, c, t)] = (d.lamc[idxV(d, vk, t)] / Dk) * (Dk * (sBold - b));
            }
            if (has1) {
                double b = d.zm[idxM(d, f * 3 + k, 1, c, t - 1)];
                if (dv != 0.0) b /= dv;
                d.zm[idxM(d, f * 3 + k, 1, c, t - 1)] = (d.lamc[idxV(d, vk, t - 1)] / Dk) * (Dk * (sBold - b));
            }
        }
    }
}
int materialise_zmid(Ctx *c) {
    if (!c->zmid_deferred) return 0;
    c->zmid_deferred = 0;
    hipLaunchKernelGGL(k_rebuild_zmid, dim3(xcd_grid(c->d.n_ftiles)), dim3(BLOCK), 0, c->stream, c->d, c->B_alt, c->zmid_sz, c->zmid_dv);
    DOTS_HIP(hipGetLastError());
    return 0;
}

// Step 0 of is_palm = True: (A, B, lambda_c) from the current multipliers and the stored z_mid; nothing else moves.
int launch_q_lambda_only(Ctx *c) {
    const dots_params &p = c->prm;
    const int nf8 = xcd_grid(c->d.n_ftiles), nv8 = xcd_grid(c->d.n_vtiles);
    hipLaunchKernelGGL((k_q_lambda_mult_triangle<0, true>), dim3(nf8 * (TILE_ELEMS / BLOCK) + nv8), dim3(BLOCK), 0, c->stream, c->d, p.scale_z, p.tau, nf8,
                       p.const_d, p.congestion * p.r);
    DOTS_HIP(hipGetLastError());
    return 0;
}

// steps 2+3 as launch_q_lambda_mult would run them can divide the dual arrays as they read them (the carry kernels)
bool ql_divides(const Ctx *c, int zmid_mode) {
    return carry_possible(c) && ((c->step_carry && zmid_mode >= 1) || (c->step_kkt && zmid_mode == 1 && c->kkt_fused.part_v && !c->d.slab));
}

int launch_q_lambda_mult(Ctx *c, int zmid_mode, double dv) {
    const dots_params &p = c->prm;
    const int nf8 = xcd_grid(c->d.n_ftiles), nv8 = xcd_grid(c->d.n_vtiles);
    const double cd = p.const_d, cr = p.congestion * p.r;
    c->carry_valid = c->kkt_fused_valid = 0;
    // DOTS_STEP_KKT_SUMS: the launch also leaves the sums of the KKT conditions it can form from its registers (kkt_fused) and the
    // per-corner gather of Dual(alpha) (cn_e); it takes the carry mapping (whole triangles per workgroup) whether or not the next
    // iteration's gathers are wanted (emit)
    const bool kkt = c->step_kkt && zmid_mode == 1 && c->kkt_fused.part_v && carry_possible(c) && !c->d.slab;
    const KktArgs ka{KKT_FUSED_MASK, p.r, p.scale_z, p.const_d, p.congestion, p.prim_scale, p.dual_scale, p.boundary_scale};
    KktFused kf = c->kkt_fused;
    kf.nv = nv8;
    const bool carry = carry_possible(c) && c->step_carry && zmid_mode >= 1;
    if (carry || kkt) {      // k_q_lambda_mult_carry
        const int tw = (2 * CARRY_NB / c->d.TP) / 3, n_fwg = xcd_grid((c->d.F + tw - 1) / tw);
        const dim3 g(n_fwg + nv8);
        kf.nf = n_fwg;
        const bool k = kkt && nv8 <= c->kkt_fused_cap_v && n_fwg <= c->kkt_fused_cap_f;
        const int emit = (carry ? 1 : 0) | (k ? 2 : 0);
        // z_mid on demand: nothing of it is stored; the new B / beta_mid go to the alternate buffers, the old ones stay (Ctx::zmid_deferred)
        const bool defer = zmid_mode == 1 && c->zmid_defer && c->B_alt && !c->d.slab && !c->step_palm;
        Dev dk = c->d;
        if (defer) { dk.B_st = c->B_alt; dk.bm_st = c->d.zm; }
#define CARRY_LAUNCH(Z, K)                                                                                                                                     \
    do {                                                                                                                                                     \
        if (c->bm_nt) {                                                                                                                                      \
            if (dv != 0.0) hipLaunchKernelGGL((k_q_lambda_mult_carry<Z, K, true, true>), g, dim3(CARRY_NB), 0, c->stream, dk, p.scale_z, p.tau, n_fwg, tw, cd, cr, ka, kf, dv, emit); \
            else hipLaunchKernelGGL((k_q_lambda_mult_carry<Z, K, false, true>), g, dim3(CARRY_NB), 0, c->stream, dk, p.scale_z, p.tau, n_fwg, tw, cd, cr, ka, kf, 1.0, emit);        \
        } else {                                                                                                                                             \
            if (dv != 0.0) hipLaunchKernelGGL((k_q_lambda_mult_carry<Z, K, true, false>), g, dim3(CARRY_NB), 0, c->stream, dk, p.scale_z, p.tau, n_fwg, tw, cd, cr, ka, kf, dv, emit); \
            else hipLaunchKernelGGL((k_q_lambda_mult_carry<Z, K, false, false>), g, dim3(CARRY_NB), 0, c->stream, dk, p.scale_z, p.tau, n_fwg, tw, cd, cr, ka, kf, 1.0, emit);        \
        }                                                                                                                                                    \
    } while (0)
        if (zmid_mode == 2) CARRY_LAUNCH(2, false);
        else if (defer && k) CARRY_LAUNCH(2, true);
        else if (defer) CARRY_LAUNCH(2, false);
        else if (k) CARRY_LAUNCH(1, true);
        else CARRY_LAUNCH(1, false);
#undef CARRY_LAUNCH
        DOTS_HIP(hipGetLastError());
        if (defer) {
            std::swap(c->d.B, c->B_alt);          // B_alt: the B the projection read
            std::swap(c->d.bm, c->d.zm);          // z_mid's storage: the beta_mid it read
            c->d.B_st = c->d.B;
            c->d.bm_st = c->d.bm;
            c->dcg.B = c->dgt.B = c->d.B; c->dcg.bm = c->dgt.bm = c->d.bm; c->dcg.zm = c->dgt.zm = c->d.zm;
            c->dcg.B_st = c->dgt.B_st = c->d.B; c->dcg.bm_st = c->dgt.bm_st = c->d.bm;
            c->zmid_deferred = 1;
            c->zmid_dv = dv;
            c->zmid_sz = p.scale_z;
        }
        c->carry_valid = carry ? 1 : 0;
        if (k) { c->kkt_fused = kf; c->kkt_fused_valid = 1; }
        return 0;
    }
    if (c->ql_two && c->d.TP >= 4) {      // two nodes per lane (16-byte accesses): k_q_lambda_mult_triangle2
        const dim3 g2(nf8 * (TILE_ELEMS / (2 * BLOCK)) + nv8);
        if (zmid_mode == 2) hipLaunchKernelGGL((k_q_lambda_mult_triangle2<2>), g2, dim3(BLOCK), 0, c->stream, c->d, p.scale_z, p.tau, nf8, cd, cr);
        else if (zmid_mode == 1) hipLaunchKernelGGL((k_q_lambda_mult_triangle2<1>), g2, dim3(BLOCK), 0, c->stream, c->d, p.scale_z, p.tau, nf8, cd, cr);
        else hipLaunchKernelGGL((k_q_lambda_mult_triangle2<0>), g2, dim3(BLOCK), 0, c->stream, c->d, p.scale_z, p.tau, nf8, cd, cr);
        DOTS_HIP(hipGetLastError());
        return 0;
    }
    const dim3 gf(nf8 * (TILE_ELEMS / BLOCK) + nv8);
    if (zmid_mode == 2) hipLaunchKernelGGL((k_q_lambda_mult_triangle<2>), gf, dim3(BLOCK), 0, c->stream, c->d, p.scale_z, p.tau, nf8, cd, cr);
    else if (zmid_mode == 1) hipLaunchKernelGGL((k_q_lambda_mult_triangle<1>), gf, dim3(BLOCK), 0, c->stream, c->d, p.scale_z, p.tau, nf8, cd, cr);
    else hipLaunchKernelGGL((k_q_lambda_mult_triangle<0>), gf, dim3(BLOCK), 0, c->stream, c->d, p.scale_z, p.tau, nf8, cd, cr);
    DOTS_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------
// scaling tools                                (reference: solver_socp.py:367-395)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_scale(double *x, int64_t n, double f) {
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) x[i] *= f;
}
__global__ __launch_bounds__(BLOCK) void k_divide(double *x, int64_t n, double f) {
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) x[i] /= f;
}

// Calibration launch for the PMC traffic counters (profiles/tools/pmc_summary.py): exactly one 8-byte load and
// one 8-byte store per element of the staging buffer, the access width of every kernel of this library.
__global__ __launch_bounds__(BLOCK) void k_calib_stream(double *x, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) x[i] = 2.0 * x[i];
}
int launch_calibration(Ctx *c, double *bytes_each_way) {
    const int64_t n = c->stage_count;
    hipLaunchKernelGGL(k_calib_stream, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, c->stream, c->stage, n);
    DOTS_HIP(hipGetLastError());
    DOTS_HIP(hipStreamSynchronize(c->stream));
    *bytes_each_way = 8.0 * (double)n;
    return 0;
}

static int grid_for(int64_t n) {
    int64_t g = (n + BLOCK - 1) / BLOCK;
    return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

int launch_scale_array(Ctx *c, int id, double f) {
    const int64_t n = array_count_device(c->d, id);
    hipLaunchKernelGGL(k_scale, dim3(grid_for(n)), dim3(BLOCK), 0, c->stream, c->arr(id), n, f);
    // a time slab keeps phi at the next slab's first node beside its own columns (written by the inverse transform, read by
    // grad_time in the KKT and norm kernels until the next solve): it is scaled with phi
    if (id == DOTS_PHI && c->d.slab && c->d.phi_hi) hipLaunchKernelGGL(k_scale, dim3(grid_for(c->d.V)), dim3(BLOCK), 0, c->stream, c->d.phi_hi, (int64_t)c->d.V, f);
    DOTS_HIP(hipGetLastError());
    return 0;
}

// x /= f for the five dual arrays in ONE launch (same division as k_divide, element for element)
struct DivideFive {
    double *p[5];
    int64_t end[5];      // running element counts
};
__global__ __launch_bounds__(BLOCK) void k_divide_five(DivideFive a, double f) {
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < a.end[4]; i += (int64_t)gridDim.x * BLOCK) {
        int k = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) k += i >= a.end[q];
        const int64_t j = i - (k ? a.end[k - 1] : 0);
        a.p[k][j] /= f;
    }
}

int launch_adjust_penalty(Ctx *c, double factor) {  // :367-371 (the boundary term is rebuilt from r in k_rhs)
    const int ids[5] = {DOTS_BETA_MID, DOTS_E, DOTS_MU, DOTS_BETA_FST, DOTS_BETA_END};
    DivideFive a{};
    int64_t n = 0;
    for (int k = 0; k < 5; ++k) {
        a.p[k] = c->arr(ids[k]);
        n += array_count_device(c->d, ids[k]);
        a.end[k] = n;
    }
    hipLaunchKernelGGL(k_divide_five, dim3(grid_for(n)), dim3(BLOCK), 0, c->stream, a, factor);
    DOTS_HIP(hipGetLastError());
    return 0;
}

// mu = sz (beta_fst - beta_end);  E = - L^T(beta_mid; sz)          (:386-388)
__global__ __launch_bounds__(BLOCK) void k_rebuild_mu(Dev d, double sz) {
    const int tile = xcd_tile(blockIdx.x, d.n_vtiles);
    if (tile >= d.n_vtiles) return;
    for (int e = threadIdx.x; e < TILE_ELEMS; e += BLOCK) {
        const int v = tile * d.VT + (e >> d.tp_shift), t = e & (d.TP - 1);
        if (v >= d.V || t >= d.ni) continue;
        const int iv = idxV(d, v, t);
        d.mu[iv] = sz * (d.bf[iv] - d.be[iv]);
    }
}
__device__ __forceinline__ double dec_adjoint_at(const Dev &d, const double *x, int f, int c, int t) {
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (t < d.ni) s0 += x[idxM(d, f * 3 + k, 0, c, t)];
        if (has_prev_interval(d, t)) s1 += x[idxM(d, f * 3 + k, 1, c, t - 1)];
    }
    return s0 + s1;
}
__global__ __launch_bounds__(BLOCK) void k_rebuild_E(Dev d, double sz) {
    const int tile = xcd_tile(blockIdx.x, d.n_ftiles);
    if (tile >= d.n_ftiles) return;
    for (int e = threadIdx.x; e < TILE_ELEMS; e += BLOCK) {
        const int row = tile * d.FT + (e >> d.tp_shift), t = e & (d.TP - 1);
        if (row >= 3 * d.F || t >= d.nl) continue;
        const int f = row / 3, c = row - 3 * f;
        d.E[idxF(d, f, c, t)] = -(sz * INV_SQRT3) * dec_adjoint_at(d, d.bm, f, c, t);
    }
}

int launch_scale_z(Ctx *c, double z_mul, double beta_mul, double sz_new) {
    const int zs[3] = {DOTS_Z_FST, DOTS_Z_MID, DOTS_Z_END}, bs[3] = {DOTS_BETA_FST, DOTS_BETA_MID, DOTS_BETA_END};
    for (int id : zs) launch_scale_array(c, id, z_mul);
    for (int id : bs) launch_scale_array(c, id, beta_mul);
    hipLaunchKernelGGL(k_rebuild_mu, dim3(xcd_grid(c->d.n_vtiles)), dim3(BLOCK), 0, c->stream, c->d, sz_new);
    hipLaunchKernelGGL(k_rebuild_E, dim3(xcd_grid(c->d.n_ftiles)), dim3(BLOCK), 0, c->stream, c->d, sz_new);
    DOTS_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------
// layout conversion: reference layout (time-major) <-> device layout (time-fastest, permuted)
// ------------------------------------------------------------------------------------------
// Host layouts: one GPU -- the reference's (phi (T+1,V); interval arrays (T,V); B, E (T+1,F,3); z_mid, beta_mid (T,2,3,F,3)).
// Time slab -- the same with the slab's time extent: (nl,V), (ni,V), (nl,F,3), and the corner arrays as (nl,2,3,F,3)
// indexed by the NODE an entry belongs to (entry [j][s] is the reference's [t0 + j - s][s]; entries whose interval
// does not exist are ignored on upload and zero on download).
template <bool TO_DEV>
__global__ __launch_bounds__(BLOCK) void k_convert(Dev d, int kind, double *dev, double *host) {
    // one thread per device element (t fastest -> device side coalesced; host side strided, staging only)
    const int nt = (kind == 0 || kind == 2) ? d.nl : d.ni;
    const int64_t rows = (kind <= 1) ? d.V : (kind == 2 ? (int64_t)3 * d.F : (int64_t)18 * d.F);
    const int64_t n = rows << d.tp_shift;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
        const int t = (int)(i & (d.TP - 1));
        const int64_t row = i >> d.tp_shift;
        int64_t h;
        if (kind <= 1) {
            if (t >= nt) { if (TO_DEV) dev[i] = 0.0; continue; }
            const int v = d.perm_v ? d.perm_v[row] : (int)row;
            h = (int64_t)t * d.V + v;
        } else if (kind == 2) {
            if (t >= nt) { if (TO_DEV) dev[i] = 0.0; continue; }
            const int f = (int)(row / 3), c = (int)(row - 3 * (row / 3));
            const int fo = d.perm_f ? d.perm_f[f] : f;
            h = ((int64_t)t * d.F + fo) * 3 + c;
        } else {  // row = ((f*3+k)*2+s)*3+c, column t = node of the entry: interval t - s
            const int c = (int)(row % 3);
            const int s = (int)((row / 3) % 2);
            const int k = (int)((row / 6) % 3);
            const int f = (int)(row / 18);
            const int fo = d.perm_f ? d.perm_f[f] : f;
            const int ti = t - s, tg = d.t0 + ti;         // local / global interval of this slot
            const bool valid = t < d.nl && tg >= 0 && tg < d.T;
            if (!valid && !(d.slab && t < d.nl)) { if (TO_DEV) dev[i] = 0.0; continue; }
            h = ((((int64_t)(d.slab ? t : ti) * 2 + s) * 3 + k) * d.F + fo) * 3 + c;
            if (!valid) {       // a slab's host array has the slot, the interval does not exist
                if (TO_DEV) dev[i] = 0.0;
                else host[h] = 0.0;
                continue;
            }
        }
        if (TO_DEV) dev[i] = host[h];
        else host[h] = dev[i];
    }
}

int launch_to_device_layout(Ctx *c, int id, const double *staged) {
    const int64_t n = array_count_device(c->d, id);
    hipLaunchKernelGGL(k_convert<true>, dim3(grid_for(n)), dim3(BLOCK), 0, c->stream, c->d, array_kind(id), c->arr(id),
                       const_cast<double *>(staged));
    DOTS_HIP(hipGetLastError());
    return 0;
}
int launch_from_device_layout(Ctx *c, int id, double *staged) {
    const int64_t n = array_count_device(c->d, id);
    hipLaunchKernelGGL(k_convert<false>, dim3(grid_for(n)), dim3(BLOCK), 0, c->stream, c->d, array_kind(id), c->arr(id), staged);
    DOTS_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------
// standalone operators (rows a4-a6): same index arithmetic as the fused kernels, exposed so each
// reference function has a one-to-one parity test.  in/out are device-layout scratch arrays.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_op_vertex(Dev d, int op, const double *x, double *y) {
    const int tile = xcd_tile(blockIdx.x, d.n_vtiles);
    if (tile >= d.n_vtiles) return;
    const double ih = 1.0 / d.h;
    for (int e = threadIdx.x; e < TILE_ELEMS; e += BLOCK) {
        const int v = tile * d.VT + (e >> d.tp_shift), t = e & (d.TP - 1);
        if (v >= d.V) continue;
        const int iv = idxV(d, v, t);
        if (op == DOTS_OP_GRAD_TIME) {
            y[iv] = (t < d.ni) ? (x[iv + 1] - x[iv]) * ih : 0.0;
        } else if (op == DOTS_OP_DIV_TIME || op == DOTS_OP_TIME_AVG_ADJOINT) {
            if (t >= d.nl) { y[iv] = 0.0; continue; }
            const double a = (t < d.ni) ? x[iv] : 0.0, b = (t > 0) ? x[iv - 1] : 0.0;
            y[iv] = (op == DOTS_OP_DIV_TIME) ? (a - b) * ih : 0.5 * (a + b);
        } else if (op == DOTS_OP_DIV_SPACE) {
            if (t >= d.nl) { y[iv] = 0.0; continue; }
            double ds = 0.0;
            for (int j = d.cptr[v]; j < d.cptr[v + 1]; ++j) {
                const int fk = d.cidx[j], f = fk / 3, k = fk - 3 * f;
#pragma unroll
                for (int c = 0; c < 3; ++c) ds += d.hat[(f * 3 + k) * 3 + c] * x[idxF(d, f, c, t)];
            }
            y[iv] = -ds;
        }
    }
}
__global__ __launch_bounds__(BLOCK) void k_op_triangle(Dev d, int op, double scale, const double *x, double *y) {
    const int tile = xcd_tile(blockIdx.x, d.n_ftiles);
    if (tile >= d.n_ftiles) return;
    const double sB = scale * INV_SQRT3;
    for (int e = threadIdx.x; e < TILE_ELEMS; e += BLOCK) {
        const int row = tile * d.FT + (e >> d.tp_shift), t = e & (d.TP - 1);
        if (row >= 3 * d.F || t >= d.nl) continue;
        const int f = row / 3, c = row - 3 * f;
        if (op == DOTS_OP_GRAD_SPACE) {
            double gx = 0.0;
#pragma unroll
            for (int k = 0; k < 3; ++k) gx += d.hat[(f * 3 + k) * 3 + c] * x[idxV(d, d.tri[f * 3 + k], t)];
            y[idxF(d, f, c, t)] = gx;
        } else if (op == DOTS_OP_DECOUPLE) {
            const double b = sB * x[idxF(d, f, c, t)];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (t < d.ni) y[idxM(d, f * 3 + k, 0, c, t)] = b;
                if (t > 0) y[idxM(d, f * 3 + k, 1, c, t - 1)] = b;
            }
        } else if (op == DOTS_OP_DECOUPLE_ADJOINT) {
            y[idxF(d, f, c, t)] = sB * dec_adjoint_at(d, x, f, c, t);
        }
    }
}

int launch_operator(Ctx *c, int op, double scale, const double *in, double *out) {
    if (c->d.slab) { set_error("the standalone operators work on whole arrays: not available on a time slab"); return DOTS_ERR_STATE; }
    switch (op) {
        case DOTS_OP_GRAD_TIME:
        case DOTS_OP_DIV_TIME:
        case DOTS_OP_TIME_AVG_ADJOINT:
        case DOTS_OP_DIV_SPACE:
            hipLaunchKernelGGL(k_op_vertex, dim3(xcd_grid(c->d.n_vtiles)), dim3(BLOCK), 0, c->stream, c->d, op, in, out);
            break;
        case DOTS_OP_GRAD_SPACE:
        case DOTS_OP_DECOUPLE:
        case DOTS_OP_DECOUPLE_ADJOINT:
            hipLaunchKernelGGL(k_op_triangle, dim3(xcd_grid(c->d.n_ftiles)), dim3(BLOCK), 0, c->stream, c->d, op, scale, in, out);
            break;
        default:
            set_error("unknown operator");
            return DOTS_ERR_ARGUMENT;
    }
    DOTS_HIP(hipGetLastError());
    return 0;
}

// The runtime prepares a kernel on its first use (several hundred microseconds each): do that for the kernels of an iteration when
// the context is created, not inside the first iterations that happen to use them.
void preload_alm_kernels() {
    const void *fns[] = {
        (const void *)k_rhs<false>, (const void *)k_rhs<true>, (const void *)k_rhs_modes, (const void *)k_rhs_modes2<false>, (const void *)k_rhs_modes2<true>,
        (const void *)k_rhs_modes_mfma<false>, (const void *)k_rhs_modes_mfma<true>, (const void *)k_rhs_modes_mfma<false, true>, (const void *)k_rhs_modes2<false, true>,
        (const void *)k_soc_projection<true>, (const void *)k_soc_projection<false>, (const void *)k_soc_projection<true, true>,
        (const void *)k_q_lambda_mult_triangle<0>, (const void *)k_q_lambda_mult_triangle<1>, (const void *)k_q_lambda_mult_triangle<2>,
        (const void *)k_q_lambda_mult_triangle<0, true>,
        (const void *)k_q_lambda_mult_triangle2<0>, (const void *)k_q_lambda_mult_triangle2<1>, (const void *)k_q_lambda_mult_triangle2<2>,
        (const void *)k_q_lambda_mult_carry<1>, (const void *)k_q_lambda_mult_carry<2>, (const void *)k_q_lambda_mult_carry<1, true>,
        (const void *)k_q_lambda_mult_carry<1, false, true>, (const void *)k_q_lambda_mult_carry<2, false, true>, (const void *)k_q_lambda_mult_carry<1, true, true>,
        (const void *)k_q_lambda_mult_carry<2, true, false>, (const void *)k_q_lambda_mult_carry<2, true, true>, (const void *)k_rebuild_zmid,
        (const void *)k_q_lambda_mult_carry<2, false, false, true>, (const void *)k_q_lambda_mult_carry<2, false, true, true>,
        (const void *)k_q_lambda_mult_carry<2, true, false, true>, (const void *)k_q_lambda_mult_carry<2, true, true, true>,

        (const void *)k_divide_five, (const void *)k_scale, (const void *)k_divide, (const void *)k_rebuild_mu, (const void *)k_rebuild_E,
    };
    hipFuncAttributes a;
    for (const void *f : fns) (void)hipFuncGetAttributes(&a, f);
    (void)hipGetLastError();
}

}  // namespace dots
