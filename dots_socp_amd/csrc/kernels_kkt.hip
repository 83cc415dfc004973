// KKT residuals, objective and weighted norms (reference: solver_socp.py:417-559, :875-878).
//
// The reference evaluates each residual with several numpy passes and temporaries the size of the
// state.  Here two kernels (one vertex-major, one triangle-major) read the state once and emit all
// weighted sums a requested set of conditions needs; a small kernel adds the per-workgroup partials
// in a fixed order (wave64 __shfl reductions inside the workgroup, no atomics -> deterministic) and
// the ~20 scalars are combined on the host side of the C ABI exactly as the reference's closures do.
#include "dots_dev.h"

#include <atomic>

#include <cmath>

namespace dots {

using S = CgScalOffsets;


// One element per thread (a workgroup takes a quarter of a tile, as the cone projection does): the corner walks of
// Comp(rho, f(q)) and Dual(alpha) are chains of dependent loads, and these kernels run on the iterations whose
// residuals the host waits for.
// (bid of nblk workgroups: the kernels below run it alone or beside the triangle part in one launch)
__device__ __forceinline__ void kkt_vertex_body(const Dev &d, const KktArgs &a, int bid, int nblk, double *__restrict__ part, double *lds) {
    constexpr int SUB = TILE_ELEMS / BLOCK;
    const int G8 = nblk / SUB;
    const int tile = xcd_tile(bid % G8, d.n_vtiles);
    double s[N_VSUMS];
#pragma unroll
    for (int i = 0; i < N_VSUMS; ++i) s[i] = 0.0;
    const bool c0 = a.mask & 1u, c1 = a.mask & 2u, c2 = a.mask & 4u, c3 = a.mask & 8u, c4 = a.mask & 16u, c6 = a.mask & 64u;
    if (tile < d.n_vtiles) {
        const double ih = 1.0 / d.h, rho_s = a.ds * a.r;
        for (int e = (bid / G8) * BLOCK + threadIdx.x; e < TILE_ELEMS; e += TILE_ELEMS) {
            const int v = tile * d.VT + (e >> d.tp_shift), t = e & (d.TP - 1);
            if (v >= d.V || t >= d.nl) continue;
            const int iv = idxV(d, v, t);
            const double m = d.mass_v[v];
            if (t < d.ni) {
                const double A = d.A[iv], mu = d.mu[iv], lc = d.lam[iv];
                if (c0) {   // Prim(phi, q): solver_socp.py:433-450, residuals of :592-593
                    const double dphi = (next_node(d, d.phi, d.phi_hi, v, t) - d.phi[iv]) * ih;
                    const double rm = dphi - A - lc;
                    s[V_DPHI2] += dphi * dphi * m;
                    s[V_A2] += A * A * m;
                    s[V_RESMU2] += rm * rm * m;
                }
                if (c0 || c6) s[V_LAM2] += lc * lc * m;
                if (c1) {   // Prim(q, z): :452-464 with :598-600
                    const double rf = d.zf[iv] + a.sz * A - a.cd, re = d.ze[iv] - a.sz * A - a.cd;
                    s[V_RFST2] += rf * rf * m;
                    s[V_REND2] += re * re * m;
                }
                if (c3 || c4 || c6) s[V_MU2] += mu * mu * m;
                if (c3) {   // Dual(beta): :484-503
                    const double a1 = a.sz * (d.be[iv] - d.bf[iv]);
                    s[V_AUX1_2] += a1 * a1 * m;
                    s[V_MUAUX1_2] += (mu + a1) * (mu + a1) * m;
                }
                if (c4) {   // Comp(rho, f(q)): :505-526 with :615-619
                    double q = 0.0;
                    for (int j = d.cptr[v]; j < d.cptr[v + 1]; ++j) {
                        const int f = d.cidx[j] / 3;
                        double sq = 0.0;
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            const double b0 = a.ps * d.B[idxF(d, f, c, t)], b1 = a.ps * next_node(d, d.B, d.B_hi, f * 3 + c, t);
                            sq += b0 * b0 + b1 * b1;
                        }
                        q += d.c_area[j] * sq * (1.0 / 3.0);
                    }
                    const double rho = rho_s * mu;
                    const double aux = a.ps * A + 0.25 * q / m;
                    const double res = fmax(0.0, aux + rho) - rho;
                    s[V_COMP_AUX2] += aux * aux * m;
                    s[V_COMP_RES2] += res * res * m;
                }
                if (c6) {   // Comp(rho, cong.): :549-559 with :633-636
                    const double res = a.cong * (rho_s * mu) - a.ps * lc;
                    s[V_CONG_RES2] += res * res * m;
                }
            }
            if (c2) {       // Dual(alpha): :466-482
                double x = 0.0;
                if (t < d.ni) x += d.mu[iv] * m;
                if (t > 0) x -= d.mu[iv - 1] * m;
                else if (has_prev_interval(d, t)) x -= d.mu_lo[v] * m;
                x *= ih;
                double dsx = 0.0;
                for (int j = d.cptr[v]; j < d.cptr[v + 1]; ++j) {
                    const int f = d.cidx[j] / 3;
#pragma unroll
                    for (int c = 0; c < 3; ++c) dsx += d.c_gA[j * 3 + c] * d.E[idxF(d, f, c, t)];
                }
                x -= dsx;
                if (first_node(d, t)) x -= a.bs * d.mu0[v] / (a.r * d.h);
                if (last_node(d, t)) x += a.bs * d.mu1[v] / (a.r * d.h);
                const double aux = (a.r * d.h) * x / m;
                s[V_DUALAUX2] += aux * aux * m;
            }
        }
    }
    block_sum<N_VSUMS>(s, lds);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < N_VSUMS; ++i) part[(int64_t)i * nblk + bid] = s[i];
    }
}

__device__ __forceinline__ void kkt_triangle_body(const Dev &d, const KktArgs &a, int bid, int nblk, double *__restrict__ part, double *lds) {
    constexpr int SUB = TILE_ELEMS / BLOCK;
    const int G8 = nblk / SUB;
    const int tile = xcd_tile(bid % G8, d.n_ftiles);
    double s[N_FSUMS];
#pragma unroll
    for (int i = 0; i < N_FSUMS; ++i) s[i] = 0.0;
    const bool c0 = a.mask & 1u, c1 = a.mask & 2u, c3 = a.mask & 8u, c5 = a.mask & 32u;
    if (tile < d.n_ftiles) {
        const double sB = a.sz * INV_SQRT3, rho_s = a.ds * a.r;
        for (int e = (bid / G8) * BLOCK + threadIdx.x; e < TILE_ELEMS; e += TILE_ELEMS) {
            const int row = tile * d.FT + (e >> d.tp_shift), t = e & (d.TP - 1);
            if (row >= 3 * d.F || t >= d.nl) continue;
            const int f = row / 3, c = row - 3 * f;
            const double w = d.area_f[f];
            const int64_t ie = idxF(d, f, c, t);
            const double B = d.B[ie];
            if (c0) {
                double gx = 0.0;
#pragma unroll
                for (int k = 0; k < 3; ++k) gx += d.hat[(f * 3 + k) * 3 + c] * d.phi[idxV(d, d.tri[f * 3 + k], t)];
                s[F_DX2 - N_VSUMS] += gx * gx * w;
                s[F_B2 - N_VSUMS] += B * B * w;
                s[F_RESE2 - N_VSUMS] += (gx - B) * (gx - B) * w;
            }
            if (c3 || c5) {
                const double E = d.E[ie];
                s[F_E2 - N_VSUMS] += E * E * w;
                if (c3) {
                    double s0 = 0.0, s1 = 0.0;
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        if (t < d.ni) s0 += d.bm[idxM(d, f * 3 + k, 0, c, t)];
                        if (has_prev_interval(d, t)) s1 += d.bm[idxM(d, f * 3 + k, 1, c, t - 1)];
                    }
                    const double a2 = sB * (s0 + s1);
                    s[F_AUX2_2 - N_VSUMS] += a2 * a2 * w;
                    s[F_EAUX2_2 - N_VSUMS] += (E + a2) * (E + a2) * w;
                }
                if (c5) {   // Comp(m, rho o B): :528-547 with :624-628
                    double rn = 0.0;
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const int vk = d.tri[f * 3 + k];
                        const int iv = idxV(d, vk, t);
                        double r2 = 0.0;
                        if (t < d.ni) r2 += d.mu[iv];
                        if (t > 0) r2 += d.mu[iv - 1];
                        else if (has_prev_interval(d, t)) r2 += d.mu_lo[vk];
                        rn += 0.5 * rho_s * r2;
                    }
                    const double aux = (rn * (1.0 / 3.0)) * (a.ps * B);
                    const double mm = rho_s * E;
                    s[F_AUX5_2 - N_VSUMS] += aux * aux * w;
                    s[F_AUX5M_2 - N_VSUMS] += (aux - mm) * (aux - mm) * w;
                }
            }
            if (c1) {       // z_mid part of Prim(q, z): sz (z_mid - L B)
                const double sb = sB * B;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    if (t < d.ni) {
                        const double q = a.sz * (d.zm[idxM(d, f * 3 + k, 0, c, t)] - sb);
                        s[F_RMID2 - N_VSUMS] += q * q * w;
                    }
                    if (has_prev_interval(d, t)) {
                        const double q = a.sz * (d.zm[idxM(d, f * 3 + k, 1, c, t - 1)] - sb);
                        s[F_RMID2 - N_VSUMS] += q * q * w;
                    }
                }
            }
        }
    }
    block_sum<N_FSUMS>(s, lds);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < N_FSUMS; ++i) part[(int64_t)i * nblk + bid] = s[i];
    }
}

// ---- the same sums with TWO consecutive nodes (t, t + 1; t even) per lane, one GPU only (no halos): every array is read in the
// column of its node or interval, so both sit in one aligned 16-byte word of the row; the corner walks load the mesh constants
// once for both.  Half the waves for the same bytes (these launches run on the iterations the host waits for).  Element for
// element the expressions of the bodies above; a lane adds its first node's terms, then its second's. -----------------------
__device__ __forceinline__ void kkt_vertex_body2(const Dev &d, const KktArgs &a, int bid, int nblk, double *__restrict__ part, double *lds) {
    constexpr int SUB = TILE_ELEMS / (2 * BLOCK);
    const int G8 = nblk / SUB;
    const int tile = xcd_tile(bid % G8, d.n_vtiles);
    double s[N_VSUMS];
#pragma unroll
    for (int i = 0; i < N_VSUMS; ++i) s[i] = 0.0;
    const bool c0 = a.mask & 1u, c1 = a.mask & 2u, c2 = a.mask & 4u, c3 = a.mask & 8u, c4 = a.mask & 16u, c6 = a.mask & 64u;
    const int e = ((bid / G8) * BLOCK + threadIdx.x) * 2;
    const int v = tile * d.VT + (e >> d.tp_shift), t = e & (d.TP - 1);
    if (tile < d.n_vtiles && v < d.V && t < d.nl) {
        const double ih = 1.0 / d.h, rho_s = a.ds * a.r;
        const int iv = idxV(d, v, t);
        const double m = d.mass_v[v];
        const bool has[2] = {t < d.ni, t + 1 < d.ni};          // the intervals t, t + 1 exist
        const bool node1 = t + 1 < d.nl;                        // the second node exists
        const D2 A = ld2(d.A + iv), mu = ld2(d.mu + iv), lc = ld2(d.lam + iv);
        if (has[0]) {
            D2 ph = {{0.0, 0.0}}, zf = ph, ze = ph, be = ph, bf = ph;
            double ph2 = 0.0;
            if (c0) { ph = ld2(d.phi + iv); if (has[1]) ph2 = d.phi[iv + 2]; }
            if (c1) { zf = ld2(d.zf + iv); ze = ld2(d.ze + iv); }
            if (c3) { be = ld2(d.be + iv); bf = ld2(d.bf + iv); }
            double q[2] = {0.0, 0.0};
            if (c4) {   // one corner walk for both intervals: B at the nodes t, t + 1, t + 2
                for (int j = d.cptr[v]; j < d.cptr[v + 1]; ++j) {
                    const int f = d.cidx[j] / 3;
                    double sq[2] = {0.0, 0.0};
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const int64_t i = idxF(d, f, c, t);
                        const D2 b = ld2(d.B + i);
                        const double b0 = a.ps * b.v[0], b1 = a.ps * b.v[1];
                        sq[0] += b0 * b0 + b1 * b1;
                        if (has[1]) {
                            const double b2 = a.ps * d.B[i + 2];
                            sq[1] += b1 * b1 + b2 * b2;
                        }
                    }
                    const double ca = d.c_area[j];
                    q[0] += ca * sq[0] * (1.0 / 3.0);
                    q[1] += ca * sq[1] * (1.0 / 3.0);
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (!has[u]) continue;
                const double Au = A.v[u], muu = mu.v[u], lcu = lc.v[u];
                if (c0) {
                    const double dphi = ((u == 0 ? ph.v[1] : ph2) - ph.v[u]) * ih;
                    const double rm = dphi - Au - lcu;
                    s[V_DPHI2] += dphi * dphi * m;
                    s[V_A2] += Au * Au * m;
                    s[V_RESMU2] += rm * rm * m;
                }
                if (c0 || c6) s[V_LAM2] += lcu * lcu * m;
                if (c1) {
                    const double rf = zf.v[u] + a.sz * Au - a.cd, re = ze.v[u] - a.sz * Au - a.cd;
                    s[V_RFST2] += rf * rf * m;
                    s[V_REND2] += re * re * m;
                }
                if (c3 || c4 || c6) s[V_MU2] += muu * muu * m;
                if (c3) {
                    const double a1 = a.sz * (be.v[u] - bf.v[u]);
                    s[V_AUX1_2] += a1 * a1 * m;
                    s[V_MUAUX1_2] += (muu + a1) * (muu + a1) * m;
                }
                if (c4) {
                    const double rho = rho_s * muu;
                    const double aux = a.ps * Au + 0.25 * q[u] / m;
                    const double res = fmax(0.0, aux + rho) - rho;
                    s[V_COMP_AUX2] += aux * aux * m;
                    s[V_COMP_RES2] += res * res * m;
                }
                if (c6) {
                    const double res = a.cong * (rho_s * muu) - a.ps * lcu;
                    s[V_CONG_RES2] += res * res * m;
                }
            }
        }
        if (c2) {       // Dual(alpha) at the nodes t and t + 1
            const double mum1 = t > 0 ? d.mu[iv - 1] : 0.0;
            double dsx[2] = {0.0, 0.0};
            for (int j = d.cptr[v]; j < d.cptr[v + 1]; ++j) {
                const int f = d.cidx[j] / 3;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const double ga = d.c_gA[j * 3 + c];
                    const D2 E = ld2(d.E + idxF(d, f, c, t));
                    dsx[0] += ga * E.v[0];
                    dsx[1] += ga * E.v[1];
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (u == 1 && !node1) continue;
                const int tu = t + u;
                double x = 0.0;
                if (tu < d.ni) x += mu.v[u] * m;
                if (tu > 0) x -= (u == 0 ? mum1 : mu.v[0]) * m;
                x *= ih;
                x -= dsx[u];
                if (tu == 0) x -= a.bs * d.mu0[v] / (a.r * d.h);
                if (tu == d.T) x += a.bs * d.mu1[v] / (a.r * d.h);
                const double aux = (a.r * d.h) * x / m;
                s[V_DUALAUX2] += aux * aux * m;
            }
        }
    }
    block_sum<N_VSUMS>(s, lds);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < N_VSUMS; ++i) part[(int64_t)i * nblk + bid] = s[i];
    }
}

__device__ __forceinline__ void kkt_triangle_body2(const Dev &d, const KktArgs &a, int bid, int nblk, double *__restrict__ part, double *lds) {
    constexpr int SUB = TILE_ELEMS / (2 * BLOCK);
    const int G8 = nblk / SUB;
    const int tile = xcd_tile(bid % G8, d.n_ftiles);
    double s[N_FSUMS];
#pragma unroll
    for (int i = 0; i < N_FSUMS; ++i) s[i] = 0.0;
    const bool c0 = a.mask & 1u, c1 = a.mask & 2u, c3 = a.mask & 8u, c5 = a.mask & 32u;
    const int e = ((bid / G8) * BLOCK + threadIdx.x) * 2;
    const int row = tile * d.FT + (e >> d.tp_shift), t = e & (d.TP - 1);
    if (tile < d.n_ftiles && row < 3 * d.F && t < d.nl) {
        const double sB = a.sz * INV_SQRT3, rho_s = a.ds * a.r;
        const int f = row / 3, c = row - 3 * f;
        const double w = d.area_f[f];
        const int64_t ie = idxF(d, f, c, t);
        const int nu = t + 1 < d.nl ? 2 : 1;                    // nodes of this lane
        const D2 B = ld2(d.B + ie);
        int vk[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) vk[k] = d.tri[f * 3 + k];
        D2 E = {{0.0, 0.0}}, phik[3], b0[3], b1[3], z0[3], z1[3], muk[3];
        double mum1[3] = {0.0, 0.0, 0.0}, hk[3] = {0.0, 0.0, 0.0};
        if (c3 || c5) E = ld2(d.E + ie);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (c0) { hk[k] = d.hat[(f * 3 + k) * 3 + c]; phik[k] = ld2(d.phi + idxV(d, vk[k], t)); }
            if (c3) { b0[k] = ld2(d.bm + idxM(d, f * 3 + k, 0, c, t)); b1[k] = ld2(d.bm + idxM(d, f * 3 + k, 1, c, t - 1)); }
            if (c1) { z0[k] = ld2(d.zm + idxM(d, f * 3 + k, 0, c, t)); z1[k] = ld2(d.zm + idxM(d, f * 3 + k, 1, c, t - 1)); }
            if (c5) {
                const int iv = idxV(d, vk[k], t);
                muk[k] = ld2(d.mu + iv);
                if (t > 0) mum1[k] = d.mu[iv - 1];
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (u >= nu) continue;
            const int tu = t + u;
            const bool has0 = tu < d.ni, has1 = tu > 0;          // the intervals tu and tu - 1 exist
            const double Bu = B.v[u];
            if (c0) {
                double gx = 0.0;
#pragma unroll
                for (int k = 0; k < 3; ++k) gx += hk[k] * phik[k].v[u];
                s[F_DX2 - N_VSUMS] += gx * gx * w;
                s[F_B2 - N_VSUMS] += Bu * Bu * w;
                s[F_RESE2 - N_VSUMS] += (gx - Bu) * (gx - Bu) * w;
            }
            if (c3 || c5) {
                const double Eu = E.v[u];
                s[F_E2 - N_VSUMS] += Eu * Eu * w;
                if (c3) {
                    double s0 = 0.0, s1 = 0.0;
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        if (has0) s0 += b0[k].v[u];
                        if (has1) s1 += b1[k].v[u];
                    }
                    const double a2 = sB * (s0 + s1);
                    s[F_AUX2_2 - N_VSUMS] += a2 * a2 * w;
                    s[F_EAUX2_2 - N_VSUMS] += (Eu + a2) * (Eu + a2) * w;
                }
                if (c5) {
                    double rn = 0.0;
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        double r2 = 0.0;
                        if (has0) r2 += muk[k].v[u];
                        if (has1) r2 += (u == 0 ? mum1[k] : muk[k].v[0]);
                        rn += 0.5 * rho_s * r2;
                    }
                    const double aux = (rn * (1.0 / 3.0)) * (a.ps * Bu);
                    const double mm = rho_s * Eu;
                    s[F_AUX5_2 - N_VSUMS] += aux * aux * w;
                    s[F_AUX5M_2 - N_VSUMS] += (aux - mm) * (aux - mm) * w;
                }
            }
            if (c1) {
                const double sb = sB * Bu;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    if (has0) {
                        const double q = a.sz * (z0[k].v[u] - sb);
                        s[F_RMID2 - N_VSUMS] += q * q * w;
                    }
                    if (has1) {
                        const double q = a.sz * (z1[k].v[u] - sb);
                        s[F_RMID2 - N_VSUMS] += q * q * w;
                    }
                }
            }
        }
    }
    block_sum<N_FSUMS>(s, lds);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < N_FSUMS; ++i) part[(int64_t)i * nblk + bid] = s[i];
    }
}

// Dual(alpha) (solver_socp.py:466-482) from the per-corner sums sum_xyz area * hat * E that the last steps-2+3 launch left in cn_e
// (corner-list order: a vertex's rows are contiguous): two nodes per lane, no walk over E.  The expression of kkt_vertex_body2;
// the corner sums are added in list order (a corner's xyz share first: the residual agrees with that body to rounding).
__global__ __launch_bounds__(BLOCK) void k_kkt_dual_alpha_carried(Dev d, KktArgs a, int nblk, double *__restrict__ part) {
    __shared__ double lds[4];
    constexpr int SUB = TILE_ELEMS / (2 * BLOCK);
    const int G8 = nblk / SUB, bid = blockIdx.x;
    const int tile = xcd_tile(bid % G8, d.n_vtiles);
    double s[1] = {0.0};
    const int e = ((bid / G8) * BLOCK + threadIdx.x) * 2;
    const int v = tile * d.VT + (e >> d.tp_shift), t = e & (d.TP - 1);
    if (tile < d.n_vtiles && v < d.V && t < d.nl) {
        const double ih = 1.0 / d.h, m = d.mass_v[v];
        const int iv = idxV(d, v, t);
        const D2 mu = ld2(d.mu + iv);
        const double mum1 = t > 0 ? d.mu[iv - 1] : 0.0;
        double dsx[2] = {0.0, 0.0};
        const int j0 = d.cptr[v], j1 = d.cptr[v + 1];
        const double *__restrict__ g = d.cn_e + t;
        for (int j = j0; j < j1; j += 2) {
            const D2 g0 = ld2_nt(g + ((int64_t)j << d.tp_shift)), g1 = ld2_nt(g + ((int64_t)min(j + 1, j1 - 1) << d.tp_shift));
            dsx[0] += g0.v[0];
            dsx[1] += g0.v[1];
            if (j + 1 < j1) {
                dsx[0] += g1.v[0];
                dsx[1] += g1.v[1];
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int tu = t + u;
            if (tu >= d.nl) continue;
            double x = 0.0;
            if (tu < d.ni) x += mu.v[u] * m;
            if (tu > 0) x -= (u == 0 ? mum1 : mu.v[0]) * m;
            x *= ih;
            x -= dsx[u];
            if (tu == 0) x -= a.bs * d.mu0[v] / (a.r * d.h);
            if (tu == d.T) x += a.bs * d.mu1[v] / (a.r * d.h);
            const double aux = (a.r * d.h) * x / m;
            s[0] += aux * aux * m;
        }
    }
    block_sum<1>(s, lds);
    if (threadIdx.x == 0) part[bid] = s[0];
}

// Workgroups [0, nv): the vertex sums (first: their corner walks are the longer chain), [nv, nv + nf): the triangle sums.
// One launch for both on the iterations that read residuals back (independent sums: two latency-bound kernels overlap).
template <bool TWO>
__global__ __launch_bounds__(BLOCK) void k_kkt_sums(Dev d, KktArgs a, int nv, double *__restrict__ part_v, double *__restrict__ part_f) {
    __shared__ double lds[(N_VSUMS > N_FSUMS ? N_VSUMS : N_FSUMS) * 4];
    if (TWO) {
        if ((int)blockIdx.x < nv) kkt_vertex_body2(d, a, blockIdx.x, nv, part_v, lds);
        else kkt_triangle_body2(d, a, blockIdx.x - nv, gridDim.x - nv, part_f, lds);
    } else {
        if ((int)blockIdx.x < nv) kkt_vertex_body(d, a, blockIdx.x, nv, part_v, lds);
        else kkt_triangle_body(d, a, blockIdx.x - nv, gridDim.x - nv, part_f, lds);
    }
}

// one workgroup per slot: scal[SUMS + first + slot] = sum_blk part[slot][blk]
__global__ __launch_bounds__(BLOCK) void k_reduce_slots(const double *part, int nblk, double *out) {
    __shared__ double lds[4];
    double v[1] = {0.0};
    for (int g = threadIdx.x; g < nblk; g += BLOCK) v[0] += part[(int64_t)blockIdx.x * nblk + g];
    block_sum<1>(v, lds);
    if (threadIdx.x == 0) out[blockIdx.x] = v[0];
}

int reduce_partials(Ctx *c, const double *part, int n_slots, int nblk, int first) {
    hipLaunchKernelGGL(k_reduce_slots, dim3(n_slots), dim3(BLOCK), 0, c->stream, part, nblk, c->d.scal + S::SUMS + first);
    DOTS_HIP(hipGetLastError());
    return 0;
}

// The sums reach the host through a mailbox in coherent pinned memory that a one-wavefront kernel writes itself (values,
// system-scope fence, sequence number); the host spins on the sequence number.  Against copy + hipStreamSynchronize this
// saves the copy's launch and the wake-up of a blocked host thread on every iteration that reads residuals back.
__global__ void k_mail_sums(const double *__restrict__ src, int n, double *mail, double seq) {
    const int i = threadIdx.x;
    if (i < n) mail[i] = src[i];
    __threadfence_system();
    __syncthreads();
    if (i == 0) {
        __threadfence_system();
        __hip_atomic_store(&mail[MAX_SUMS], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// The reduction of the KKT sums and their hand-over in one launch: workgroup = slot (vertex slots sum nv partial blocks,
// triangle slots nf; a part that did not run gives 0); every workgroup writes its sum to scal and to the mailbox, the last one
// to arrive (device counter) publishes the sequence number.
// (src: where every slot's per-workgroup partial sums lie and how many there are -- the KKT kernels of this file, or the
// steps-2+3 launch that formed them from its registers: DOTS_STEP_KKT_SUMS)
struct ReduceSrc {
    const double *p[N_SUMS];
    int n[N_SUMS];
};
static ReduceSrc reduce_src(const double *part_v, int nv, const double *part_f, int nf) {
    ReduceSrc r{};
    for (int i = 0; i < N_SUMS; ++i) {
        r.p[i] = i < N_VSUMS ? part_v + (int64_t)i * nv : part_f + (int64_t)(i - N_VSUMS) * nf;
        r.n[i] = i < N_VSUMS ? nv : nf;
    }
    return r;
}
__global__ __launch_bounds__(BLOCK) void k_reduce_mail(ReduceSrc src, double *__restrict__ out, double *mail, double seq, int *counter) {
    __shared__ double lds[4];
    const int slot = blockIdx.x;
    const double *__restrict__ part = src.p[slot];
    const int nblk = src.n[slot];
    // (four partial sums per thread: at 10^5 vertices a slot has 4 x 10^4 blocks, one chain of loads per thread took 60 us)
    double w[4] = {0.0, 0.0, 0.0, 0.0};
    int g = threadIdx.x;
    for (; g + 3 * BLOCK < nblk; g += 4 * BLOCK) {
#pragma unroll
        for (int u = 0; u < 4; ++u) w[u] += part[g + u * BLOCK];
    }
    for (; g < nblk; g += BLOCK) w[0] += part[g];
    double v[1] = {(w[0] + w[1]) + (w[2] + w[3])};
    block_sum<1>(v, lds);
    if (threadIdx.x == 0) {
        out[slot] = v[0];
        if (!mail) return;               // device hand-over (dots_kkt_sums_device): the caller's stream order publishes `out`
        mail[slot] = v[0];
        __threadfence_system();
        if (atomicAdd(counter, 1) == (int)gridDim.x - 1) {
            *counter = 0;
            __threadfence_system();
            __hip_atomic_store(&mail[MAX_SUMS], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

static int wait_mail(Ctx *c, double seq, int n) {
    volatile double *flag = c->h_mail + MAX_SUMS;
    bool got = false;
    for (int64_t spins = 0; spins < c->mail_spins; ++spins) {   // ~ tens of ms at most; then the blocking wait
        if (*flag == seq) { got = true; break; }
        __builtin_ia32_pause();
    }
    if (!got) {
        // The spin ran out (DOTS_MAIL_SPINS, default ~ tens of ms): block until the stream has drained.  If the sequence number
        // still has not arrived the publish did not happen (e.g. the counter of k_reduce_mail was left non-zero by an
        // interrupted launch): never hand stale sums to the stop / penalty decisions -- reset the counter and copy the sums
        // from the device scalars, where k_reduce_mail / k_mail_sums' source also hold them.
        DOTS_HIP(hipStreamSynchronize(c->stream));
        if (*flag != seq) {
            c->mail_fallbacks += 1;
            if (c->kkt_counter) DOTS_HIP(hipMemsetAsync(c->kkt_counter, 0, 2 * sizeof(int), c->stream));
            DOTS_HIP(hipMemcpyAsync(c->h_pinned, c->d.scal + S::SUMS, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
            DOTS_HIP(hipStreamSynchronize(c->stream));
            return 0;
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    for (int i = 0; i < n; ++i) c->h_pinned[i] = c->h_mail[i];
    return 0;
}

static int fetch_sums(Ctx *c, int n) {
    if (c->spin_fetch && c->h_mail && n <= MAX_SUMS && n <= 64) {
        const double seq = (double)(++c->mail_seq);
        const bool drop = c->mail_test_drop > 0 && c->mail_seq % (uint64_t)c->mail_test_drop == 0;      // tests: this publish goes astray
        hipLaunchKernelGGL(k_mail_sums, dim3(1), dim3(64), 0, c->stream, c->d.scal + S::SUMS, n, c->h_mail, drop ? -seq : seq);
        DOTS_HIP(hipGetLastError());
        return wait_mail(c, seq, n);
    }
    DOTS_HIP(hipMemcpyAsync(c->h_pinned, c->d.scal + S::SUMS, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
    DOTS_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

// The sums left in the caller's DEVICE buffer (a time slab hands them straight to the all-reduce): enqueue only, no host wait.
int kkt_sums_device(Ctx *c, uint32_t mask, double *out) {
    const Dev &d = c->d;
    const dots_params &p = c->prm;
    KktArgs a{mask, p.r, p.scale_z, p.const_d, p.congestion, p.prim_scale, p.dual_scale, p.boundary_scale};
    const bool two = c->kkt_two && !d.slab && d.TP >= 4;
    const int per_tile = two ? TILE_ELEMS / (2 * BLOCK) : TILE_ELEMS / BLOCK;
    const int gv = xcd_grid(d.n_vtiles) * per_tile, gf = xcd_grid(d.n_ftiles) * per_tile;
    double *part_f = d.partials + (int64_t)N_VSUMS * gv;
    const bool need_v = mask & (1u | 2u | 4u | 8u | 16u | 64u), need_f = mask & (1u | 2u | 8u | 32u);
    const int nv = (need_v && d.nl > 0) ? gv : 0, nf = (need_f && d.nl > 0) ? gf : 0;
    if (nv + nf == 0) {      // nothing to sum (no mask bit, or a rank without nodes): zeros
        DOTS_HIP(hipMemsetAsync(out, 0, sizeof(double) * N_SUMS, c->stream));
        return 0;
    }
    if (two) hipLaunchKernelGGL(k_kkt_sums<true>, dim3(nv + nf), dim3(BLOCK), 0, c->stream, d, a, nv, d.partials, part_f);
    else hipLaunchKernelGGL(k_kkt_sums<false>, dim3(nv + nf), dim3(BLOCK), 0, c->stream, d, a, nv, d.partials, part_f);
    // (a part that did not run sums zero blocks: its slots are zero)
    hipLaunchKernelGGL(k_reduce_mail, dim3(N_SUMS), dim3(BLOCK), 0, c->stream, reduce_src(d.partials, nv, part_f, nf), out, (double *)nullptr, 0.0, (int *)nullptr);
    DOTS_HIP(hipGetLastError());
    return 0;
}

// The weighted sums the conditions in `mask` need (of this context's time slab; the whole problem on one GPU).
int kkt_sums(Ctx *c, uint32_t mask, double *sums) {
    const Dev &d = c->d;
    const dots_params &p = c->prm;
    KktArgs a{mask, p.r, p.scale_z, p.const_d, p.congestion, p.prim_scale, p.dual_scale, p.boundary_scale};
    const bool two = c->kkt_two && !d.slab && d.TP >= 4;      // two nodes per lane (kkt_*_body2)
    const int per_tile = two ? TILE_ELEMS / (2 * BLOCK) : TILE_ELEMS / BLOCK;
    const int gv = xcd_grid(d.n_vtiles) * per_tile, gf = xcd_grid(d.n_ftiles) * per_tile;
    double *part_f = d.partials + (int64_t)N_VSUMS * gv;
    const bool need_v = mask & (1u | 2u | 4u | 8u | 16u | 64u), need_f = mask & (1u | 2u | 8u | 32u);
    for (int i = 0; i < N_SUMS; ++i) sums[i] = 0.0;
    if (d.nl == 0) return 0;          // a rank without nodes contributes nothing
    int nv = need_v ? gv : 0, nf = need_f ? gf : 0;
    if (nv + nf == 0) return 0;
    // DOTS_STEP_KKT_SUMS: the last steps-2+3 launch left the sums of Prim(phi, q), Prim(q, z), Dual(beta) and Comp(rho, cong.)
    // (no pass over the state for them); Dual(alpha) gathers E over the corner lists: the vertex kernel alone, with that
    // condition only.  Conditions that gather more (4, 5) take the kernels of this file for everything.
    const bool fused = kkt_takes_fused(c, mask);
    ReduceSrc src = reduce_src(d.partials, nv, part_f, nf);
    bool have[N_SUMS];
    for (int i = 0; i < N_SUMS; ++i) have[i] = (i < N_VSUMS) ? need_v : need_f;
    if (fused) {
        nv = (mask & 4u) ? gv : 0;
        a.mask = 4u;
        for (int i = 0; i < N_SUMS; ++i) { src.p[i] = d.partials; src.n[i] = 0; have[i] = false; }
        src.p[V_DUALAUX2] = d.partials + (int64_t)V_DUALAUX2 * nv;
        src.n[V_DUALAUX2] = nv;
        have[V_DUALAUX2] = nv > 0;
        const KktFused &kf = c->kkt_fused;
        for (int i = 0; i < N_SUMS; ++i) {
            const bool v_slot = i == V_DPHI2 || i == V_A2 || i == V_LAM2 || i == V_RESMU2 || i == V_RFST2 || i == V_REND2 || i == V_MU2 || i == V_AUX1_2 ||
                                i == V_MUAUX1_2 || i == V_CONG_RES2;
            const bool f_slot = i == F_DX2 || i == F_B2 || i == F_RESE2 || i == F_E2 || i == F_AUX2_2 || i == F_EAUX2_2 || i == F_RMID2;
            if (v_slot) { src.p[i] = kf.part_v + (int64_t)i * kf.nv; src.n[i] = kf.nv; have[i] = true; }
            if (f_slot) { src.p[i] = kf.part_f + (int64_t)(i - N_VSUMS) * kf.nf; src.n[i] = kf.nf; have[i] = true; }
        }
        if (nv && two && d.cn_e) hipLaunchKernelGGL(k_kkt_dual_alpha_carried, dim3(nv), dim3(BLOCK), 0, c->stream, d, a, nv, d.partials + (int64_t)V_DUALAUX2 * nv);
        else if (nv) {
            if (two) hipLaunchKernelGGL(k_kkt_sums<true>, dim3(nv), dim3(BLOCK), 0, c->stream, d, a, nv, d.partials, part_f);
            else hipLaunchKernelGGL(k_kkt_sums<false>, dim3(nv), dim3(BLOCK), 0, c->stream, d, a, nv, d.partials, part_f);
        }
    } else {
        if (two) hipLaunchKernelGGL(k_kkt_sums<true>, dim3(nv + nf), dim3(BLOCK), 0, c->stream, d, a, nv, d.partials, part_f);
        else hipLaunchKernelGGL(k_kkt_sums<false>, dim3(nv + nf), dim3(BLOCK), 0, c->stream, d, a, nv, d.partials, part_f);
    }
    int rc;
    if (c->spin_fetch && c->h_mail && c->kkt_counter) {
        const double seq = (double)(++c->mail_seq);
        const bool drop = c->mail_test_drop > 0 && c->mail_seq % (uint64_t)c->mail_test_drop == 0;      // tests: this publish goes astray
        hipLaunchKernelGGL(k_reduce_mail, dim3(N_SUMS), dim3(BLOCK), 0, c->stream, src, c->d.scal + S::SUMS, c->h_mail, drop ? -seq : seq, c->kkt_counter);
        DOTS_HIP(hipGetLastError());
        // DOTS_STEP_RHS_AHEAD: while the host waits for these sums and decides, the device starts on the next iteration's
        // right-hand side (it reads only what the next dots_step would read; dots_api.hip: check() drops it if anything changes)
        if (c->rhs_ahead_armed) {
            c->rhs_ahead_armed = 0;
            if (rhs_takes_soc(c) && c->zf_alt) {      // ... and its cone projection, into the alternate buffers (a dropped launch leaves the state as it is)
                Dev keep = c->d;
                c->d.zf = c->zf_alt;
                c->d.ze = c->ze_alt;
                c->d.lamc = c->lamc_alt;
                rc = launch_rhs(c, true);
                c->d = keep;
                if (rc) return rc;
                c->rhs_ahead = 2;
            } else {
                if ((rc = launch_rhs(c))) return rc;
                c->rhs_ahead = 1;
            }
        }
        rc = wait_mail(c, seq, N_SUMS);
    } else {
        if (nv && (rc = reduce_partials(c, d.partials, N_VSUMS, nv, 0))) return rc;
        if (nf && (rc = reduce_partials(c, part_f, N_FSUMS, nf, N_VSUMS))) return rc;
        DOTS_HIP(hipGetLastError());
        rc = fetch_sums(c, N_SUMS);
    }
    if (rc) return rc;
    // slots of kernels that did not run hold leftovers of earlier calls: report zeros there
    for (int i = 0; i < N_SUMS; ++i) sums[i] = have[i] ? c->h_pinned[i] : 0.0;
    if (c->penalty_armed) {
        c->penalty_armed = 0;
        if ((rc = penalty_decision_ahead(c, mask, sums))) return rc;
    }
    return 0;
}

// dots_penalty_ahead: the reference's penalty decision (solver_socp.py:806-823, admm_tools.py:54-95) taken here, from the sums that have just
// arrived, so that the next iteration's first launch is on the stream before the host has even returned to its own (identical) decision.
int penalty_decision_ahead(Ctx *c, uint32_t mask, const double *sums) {
    const dots_penalty_policy &pp = c->penalty_policy;
    if ((mask & 15u) != 15u || c->d.slab || c->step_palm || !c->lazy_div || !c->zf_alt || c->rhs_ahead || c->carry_valid || !rhs_takes_soc(c) ||
        !rhs_divides(c) || c->pending_div != 0.0 || !carry_possible(c))      // (carry_possible: dots_adjust_penalty will leave the division pending)
        return 0;
    double o[2 * DOTS_N_KKT];
    int rc = kkt_combine(c, 15u, sums, o);
    if (rc) return rc;
    bool fails = false, finite = true;
    double max_unit = o[1];
    for (int i = 0; i < 4; ++i) {
        finite = finite && std::isfinite(o[2 * i]) && std::isfinite(o[2 * i + 1]);
        fails = fails || !(o[2 * i] < pp.tol);
        max_unit = o[2 * i + 1] > max_unit ? o[2 * i + 1] : max_unit;
    }
    if (!finite || !fails) return 0;      // all four pass: the host goes on validating (and may stop); nothing is anticipated
    const bool org = pp.is_org_kkt || max_unit < 5.0 * pp.tol;
    const int k = org ? 0 : 1;
    const double prim = o[k] > o[2 + k] ? o[k] : o[2 + k], dual = o[4 + k] > o[6 + k] ? o[4 + k] : o[6 + k];
    if (!(dual > 0.0) || !(prim >= 0.0)) return 0;
    const double gap = prim / dual;
    const bool prim_win = gap < 1.0;
    const double g = prim_win ? 1.0 / gap : gap;
    double factor = 1.0;
    for (int i = 0; i < pp.n_steps; ++i)
        if (g > pp.threshold[i]) { factor = pp.factor[i]; break; }
    if (prim_win) factor = 1.0 / factor;
    const double r = c->prm.r;
    double r_new = r * factor;
    r_new = r_new < pp.r_upper ? r_new : pp.r_upper;
    r_new = r_new > pp.r_lower ? r_new : pp.r_lower;
    const double dv = r_new / r;          // what the driver hands to dots_adjust_penalty ...
    const double r_next = r * dv;         // ... and the penalty it then sets (solver_socp.py: r *= factor)
    if (!(dv > 0.0) || !std::isfinite(dv)) return 0;
    Dev keep = c->d;
    c->d.zf = c->zf_alt;
    c->d.ze = c->ze_alt;
    c->d.lamc = c->lamc_alt;
    c->prm.r = r_next;
    rc = launch_rhs(c, true, dv);
    c->prm.r = r;
    c->d = keep;
    if (rc) return rc;
    c->rhs_ahead = 3;
    c->penalty_ahead_started += 1;
    c->ahead_dv = dv;
    c->ahead_r = r_next;
    return 0;
}

int kkt_n_sums() { return N_SUMS; }

int kkt_evaluate(Ctx *c, uint32_t mask, double *out) {
    double sums[N_SUMS];
    int rc = kkt_sums(c, mask, sums);
    if (rc) return rc;
    return kkt_combine(c, mask, sums, out);
}

// The reference's closures (solver_socp.py:433-559) on the sums of the WHOLE problem.
int kkt_combine(Ctx *c, uint32_t mask, const double *s, double *out) {
    const Dev &d = c->d;
    const dots_params &p = c->prm;
    const double T = d.T, T1 = d.T + 1;
    auto nt = [&](int i) { return s[i] / T; };     // norm_square_time
    auto nc = [&](int i) { return s[i] / T1; };    // norm_square_center
    auto ns = [&](int i) { return s[i] / T1; };    // norm_square_space
    auto nsd = [&](int i) { return s[i] / T; };    // norm_square_space_decouple
    const double rho2 = (p.dual_scale * p.r) * (p.dual_scale * p.r);
    const double nan = std::nan("");
    if (mask & 1u) {
        const double nsum = std::sqrt(nt(V_DPHI2) + ns(F_DX2)) + std::sqrt(nt(V_A2) + ns(F_B2)) + std::sqrt(nt(V_LAM2));
        const double res = std::sqrt(nt(V_RESMU2) + ns(F_RESE2));
        out[0] = res / (c->c_prim_q / p.prim_scale + nsum);
        out[1] = res / (c->c_prim_q + nsum);
    }
    if (mask & 2u) {
        const double res = std::sqrt(nt(V_RFST2) + nt(V_REND2) + nsd(F_RMID2));
        out[2] = res / (c->c_prim_z / p.prim_scale + p.norm_d);
        out[3] = res / (c->c_prim_z + p.norm_d);
    }
    if (mask & 4u) {
        const double res = std::sqrt(nc(V_DUALAUX2));
        out[4] = res / (c->c_dual_alpha / p.dual_scale + p.norm_boundary);
        out[5] = res / (c->c_dual_alpha + p.norm_boundary);
    }
    if (mask & 8u) {
        const double nsum = p.r * (std::sqrt(nt(V_MU2) + ns(F_E2)) + std::sqrt(nt(V_AUX1_2) + ns(F_AUX2_2)));
        const double res = p.r * std::sqrt(nt(V_MUAUX1_2) + ns(F_EAUX2_2));
        out[6] = res / (c->c_dual_beta / p.dual_scale + nsum);
        out[7] = res / (c->c_dual_beta + nsum);
    }
    if (mask & 16u) {
        const double nsum = std::sqrt(rho2 * nt(V_MU2)) + std::sqrt(nt(V_COMP_AUX2));
        out[8] = std::sqrt(nt(V_COMP_RES2)) / (c->c_comp_rho + nsum);
        out[9] = nan;
    }
    if (mask & 32u) {
        const double nsum = std::sqrt(rho2 * ns(F_E2)) + std::sqrt(ns(F_AUX5_2));
        out[10] = std::sqrt(ns(F_AUX5M_2)) / (c->c_comp_m + nsum);
        out[11] = nan;
    }
    if (mask & 64u) {
        const double nsum = std::sqrt(rho2 * nt(V_MU2)) + std::sqrt(p.prim_scale * p.prim_scale * nt(V_LAM2));
        out[12] = std::sqrt(nt(V_CONG_RES2)) / (c->c_comp_rho + nsum);
        out[13] = nan;
    }
    return 0;
}

// ------------------------------------------------------------------------------------------
// objective (solver_socp.py:417-431 as called at :773-775): with phi_s = ps*phi and the boundary
// term (ds*r)*bnd = ds * (-mu0, +mu1)/h the cost is  ps*ds*( <phi[T],mu1> - <phi[0],mu0> ).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_objective(Dev d) {
    __shared__ double lds[3 * 4];
    double s[3] = {0.0, 0.0, 0.0};
    const int tile = xcd_tile(blockIdx.x, d.n_vtiles);
    if (tile < d.n_vtiles) {
        for (int e = threadIdx.x; e < TILE_ELEMS; e += BLOCK) {
            const int v = tile * d.VT + (e >> d.tp_shift), t = e & (d.TP - 1);
            if (v >= d.V || t >= d.nl) continue;
            const int iv = idxV(d, v, t);
            if (first_node(d, t)) s[0] += d.phi[iv] * d.mu0[v];
            if (last_node(d, t)) s[1] += d.phi[iv] * d.mu1[v];
            if (t < d.ni) s[2] += d.lam[iv] * d.lam[iv] * d.mass_v[v];
        }
    }
    block_sum<3>(s, lds);
    if (threadIdx.x == 0)
        for (int i = 0; i < 3; ++i) d.partials[(int64_t)i * gridDim.x + blockIdx.x] = s[i];
}

int objective_sums(Ctx *c, double *sums) {
    const Dev &d = c->d;
    sums[0] = sums[1] = sums[2] = 0.0;
    if (d.nl == 0) return 0;
    const int gv = xcd_grid(d.n_vtiles);
    hipLaunchKernelGGL(k_objective, dim3(gv), dim3(BLOCK), 0, c->stream, d);
    int rc = reduce_partials(c, d.partials, 3, gv, 0);
    if (rc) return rc;
    rc = fetch_sums(c, 3);
    if (rc) return rc;
    for (int i = 0; i < 3; ++i) sums[i] = c->h_pinned[i];
    return 0;
}

int objective_combine(Ctx *c, const double *s, double *out) {
    const Dev &d = c->d;
    const dots_params &p = c->prm;
    const double cost = p.prim_scale * p.dual_scale * p.boundary_scale * (s[1] - s[0]);
    const double cong = p.congestion * p.prim_scale / p.dual_scale;
    out[0] = cost;
    out[1] = (cong > 1e-10) ? cost - (p.prim_scale * p.prim_scale * s[2] / d.T) / (2.0 * cong) : cost;
    return 0;
}

int objective_evaluate(Ctx *c, double *out) {
    double s[3];
    int rc = objective_sums(c, s);
    if (rc) return rc;
    return objective_combine(c, s, out);
}

// ------------------------------------------------------------------------------------------
// norm_square_weight of one array (solver_socp.py:875-878 with the partials of :215-218)
//   kind 0 node (mass, /(T+1))  1 interval (mass, /T)  2 triangle (area, /(T+1))  3 corner (area, /T)
//   part 1 / 2 with DOTS_PHI: grad_time(phi) / grad_space(phi)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_norm(Dev d, const double *x, int kind, int part) {
    __shared__ double lds[4];
    double s[1] = {0.0};
    const bool vert = (kind <= 1 && part != 2);
    const int ntiles = vert ? d.n_vtiles : (kind == 3 ? 6 * d.n_ftiles : d.n_ftiles);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        for (int e = threadIdx.x; e < TILE_ELEMS; e += BLOCK) {
            const int64_t row = (int64_t)tile * d.VT + (e >> d.tp_shift);
            const int t = e & (d.TP - 1);
            if (vert) {
                if (row >= d.V) continue;
                const int iv = idxV(d, (int)row, t);
                double val = 0.0;
                if (part == 1) {
                    if (t >= d.ni) continue;
                    val = (next_node(d, x, d.phi_hi, row, t) - x[iv]) / d.h;
                } else {
                    if (t >= (kind == 0 ? d.nl : d.ni)) continue;
                    val = x[iv];
                }
                s[0] += val * val * d.mass_v[row];
            } else if (kind == 3) {
                if (row >= (int64_t)18 * d.F || t >= d.nl) continue;      // slots without an interval hold zeros
                const double val = x[(row << d.tp_shift) + t];
                s[0] += val * val * d.area_f[row / 18];
            } else {
                if (row >= (int64_t)3 * d.F || t >= d.nl) continue;
                const int f = (int)(row / 3), c = (int)(row - 3 * (row / 3));
                double val;
                if (part == 2) {
                    val = 0.0;
                    for (int k = 0; k < 3; ++k) val += d.hat[(f * 3 + k) * 3 + c] * x[idxV(d, d.tri[f * 3 + k], t)];
                } else {
                    val = x[(row << d.tp_shift) + t];
                }
                s[0] += val * val * d.area_f[f];
            }
        }
    }
    block_sum<1>(s, lds);
    if (threadIdx.x == 0) d.partials[blockIdx.x] = s[0];
}

int norm_square(Ctx *c, int id, int part, double *out) {
    const Dev &d = c->d;
    // (on a time slab: the slab's SHARE of the norm -- its own nodes / intervals / corner entries over the global average
    //  count; the caller adds the shares of all slabs.  part 1 reads phi_hi, the next slab's first node of the last solve)
    const int kind = array_kind(id);
    if (part != 0 && id != DOTS_PHI) {
        set_error("part != 0 is only defined for DOTS_PHI");
        return DOTS_ERR_ARGUMENT;
    }
    const int g = 512;
    if (d.nl == 0) { *out = 0.0; return 0; }
    hipLaunchKernelGGL(k_norm, dim3(g), dim3(BLOCK), 0, c->stream, d, c->arr(id), kind, part);
    int rc = reduce_partials(c, d.partials, 1, g, 0);
    if (rc) return rc;
    rc = fetch_sums(c, 1);
    if (rc) return rc;
    double avg = (kind == 0 || kind == 2) ? d.T + 1 : d.T;
    if (part == 1) avg = d.T;
    if (part == 2) avg = d.T + 1;
    *out = c->h_pinned[0] / avg;
    return 0;
}

void preload_kkt_kernels() {      // (see preload_alm_kernels)
    const void *fns[] = {(const void *)k_kkt_sums<true>, (const void *)k_kkt_sums<false>, (const void *)k_kkt_dual_alpha_carried, (const void *)k_reduce_slots, (const void *)k_mail_sums, (const void *)k_reduce_mail,
                         (const void *)k_objective, (const void *)k_norm};
    hipFuncAttributes a;
    for (const void *f : fns) (void)hipFuncGetAttributes(&a, f);
    (void)hipGetLastError();
}

}  // namespace dots
