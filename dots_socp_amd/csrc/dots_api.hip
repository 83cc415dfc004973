// C ABI of libdotsocp_hip.so (see include/dots_socp_hip.h): context lifecycle, state transfer,
// the ALM step driver.  Everything here is host code around the kernels of kernels_*.hip.
#include "dots_dev.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

struct dots_ctx : dots::Ctx {};

namespace dots {

static thread_local std::string g_last_error;

void set_error(const std::string &msg) { g_last_error = msg; }

int hip_fail(hipError_t e, const char *what, const char *file, int line) {
    char buf[512];
    snprintf(buf, sizeof buf, "HIP error %d (%s) at %s:%d: %s", (int)e, hipGetErrorString(e), file, line, what);
    set_error(buf);
    return DOTS_ERR_HIP;
}

// An integer switch from the environment: unset keeps *out; anything that is not an integer in [lo, hi] is an error.
bool env_int(const char *name, int lo, int hi, int *out) {
    const char *e = getenv(name);
    if (!e) return true;
    char *end = nullptr;
    const long v = strtol(e, &end, 10);
    if (end == e || *end != '\0' || v < lo || v > hi) {
        char buf[256];
        snprintf(buf, sizeof buf, "environment: %s=%s is not an integer in [%d, %d]", name, e, lo, hi);
        set_error(buf);
        return false;
    }
    *out = (int)v;
    return true;
}

int array_kind(int id) {
    switch (id) {
        case DOTS_PHI: return 0;
        case DOTS_B: case DOTS_E: return 2;
        case DOTS_Z_MID: case DOTS_BETA_MID: return 3;
        default: return 1;
    }
}
int64_t array_count_host(const Dev &d, int id) {
    switch (array_kind(id)) {      // a time slab exchanges its own time extent (corner arrays: one block per node)
        case 0: return (int64_t)d.nl * d.V;
        case 1: return (int64_t)d.ni * d.V;
        case 2: return (int64_t)d.nl * d.F * 3;
        default: return (int64_t)(d.slab ? d.nl : d.ni) * 18 * d.F;
    }
}
int64_t array_count_device(const Dev &d, int id) {
    switch (array_kind(id)) {
        case 0: case 1: return (int64_t)d.V << d.tp_shift;
        case 2: return ((int64_t)3 * d.F) << d.tp_shift;
        default: return ((int64_t)18 * d.F) << d.tp_shift;
    }
}

template <typename T>
static int dev_alloc(Ctx *c, T **out, int64_t count, bool zero = true) {
    void *p = nullptr;
    const size_t bytes = sizeof(T) * (size_t)std::max<int64_t>(count, 1);
    DOTS_HIP(hipMalloc(&p, bytes));
    if (zero) DOTS_HIP(hipMemsetAsync(p, 0, bytes, c->stream));
    if (c->n_allocs >= (int)(sizeof(c->allocs) / sizeof(c->allocs[0]))) {
        set_error("allocation table full");
        return DOTS_ERR_STATE;
    }
    c->allocs[c->n_allocs++] = p;
    c->bytes += (int64_t)bytes;
    *out = (T *)p;
    return 0;
}

template <typename T>
static int dev_upload(Ctx *c, const T **out, const T *host, int64_t count) {
    T *p = nullptr;
    int rc = dev_alloc(c, &p, count, false);
    if (rc) return rc;
    DOTS_HIP(hipMemcpyAsync(p, host, sizeof(T) * (size_t)count, hipMemcpyHostToDevice, c->stream));
    DOTS_HIP(hipStreamSynchronize(c->stream));   // host buffers are borrowed for the call only
    *out = p;
    return 0;
}

static int build(Ctx *c, const dots_problem_desc *p) {
    Dev &d = c->d;
    d.T = p->n_time;
    d.V = p->n_vertices;
    d.F = p->n_triangles;
    int tpg = 8, shg = 3;                     // global pitch: power of two >= T + 1
    while (tpg < d.T + 1) { tpg <<= 1; ++shg; }
    if (tpg > TILE_ELEMS) { set_error("n_time too large: T+1 must be <= 1024"); return DOTS_ERR_ARGUMENT; }
    if (p->lap_solver == DOTS_LAP_MODAL_PCG && tpg > BLOCK) { set_error("modal solver needs T+1 <= 256"); return DOTS_ERR_ARGUMENT; }
    const bool sharded = p->slab_count > 0 || p->slab_stride > 0;
    d.t0 = 0; d.nl = d.T + 1; d.ni = d.T; d.slab = 0;
    int tp = tpg, sh = shg;
    if (sharded) {
        if (p->lap_solver != DOTS_LAP_MODAL_PCG || p->slab_stride < 1 || p->slab_begin < 0 || p->slab_count < 0 ||
            p->slab_begin + p->slab_count > d.T + 1 || p->slab_count > p->slab_stride ||
            (p->slab_count > 0 && p->slab_begin % p->slab_stride != 0) ||
            (p->slab_count < p->slab_stride && p->slab_count > 0 && p->slab_begin + p->slab_count != d.T + 1)) {
            set_error("bad time slab (needs the modal solver, begin = rank * stride, 0 <= count <= stride, a short slab only at the end)");
            return DOTS_ERR_ARGUMENT;
        }
        c->shard_begin = p->slab_begin;
        c->shard_count = p->slab_count;
        c->shard_stride = p->slab_stride;
        c->shard_ranks = (d.T + 1 + p->slab_stride - 1) / p->slab_stride;
        tp = 4; sh = 2;                       // local pitch: power of two >= the nodes per rank
        while (tp < p->slab_stride) { tp <<= 1; ++sh; }
        d.t0 = p->slab_begin; d.nl = p->slab_count; d.slab = 1;
        d.ni = std::max(0, std::min(d.nl, d.T - d.t0));
    }
    d.TP = tp;
    d.tp_shift = sh;
    if (((int64_t)d.V << sh) >= ((int64_t)1 << 31)) {      // node arrays are indexed with 32-bit arithmetic (idxV)
        set_error("V * time pitch must stay below 2^31 per context: cut the time axis into more slabs");
        return DOTS_ERR_ARGUMENT;
    }
    d.VT = d.FT = TILE_ELEMS / tp;
    d.n_vtiles = (d.V + d.VT - 1) / d.VT;
    d.n_ftiles = (3 * d.F + d.FT - 1) / d.FT;
    d.h = 1.0 / d.T;
    d.cg_ncol = d.T + 1;
    c->nnz = p->lap_nnz;
    c->lap_solver = p->lap_solver;

    const int V = d.V, F = d.F, nC = p->n_corners;
    // host-side derived per-corner tables (constants of solver_socp.py:172-192 kept per corner, never broadcast)
    std::vector<double> cD(nC), cgA((size_t)nC * 3), cArea(nC), kdiag(V, 0.0);
    for (int v = 0; v < V; ++v) {
        for (int j = p->corner_ptr[v]; j < p->corner_ptr[v + 1]; ++j) {
            const int fk = p->corner_idx[j], f = fk / 3;
            if (f < 0 || f >= F || p->triangles[fk] != v) { set_error("corner list inconsistent with triangles"); return DOTS_ERR_ARGUMENT; }
            cD[j] = std::sqrt(p->area_tri[f] / p->mass_vert[v]);
            cArea[j] = p->area_tri[f];
            for (int cc = 0; cc < 3; ++cc) cgA[(size_t)j * 3 + cc] = p->hat_grad[(size_t)fk * 3 + cc] * p->area_tri[f];
        }
        bool has_diag = false;
        for (int j = p->lap_rowptr[v]; j < p->lap_rowptr[v + 1]; ++j) {
            if (p->lap_col[j] < 0 || p->lap_col[j] >= V) { set_error("lap_col out of range"); return DOTS_ERR_ARGUMENT; }
            if (p->lap_col[j] == v) { kdiag[v] += p->lap_val[j]; has_diag = true; }
        }
        if (!has_diag || !(kdiag[v] > 0.0)) { set_error("surface stiffness matrix needs a positive diagonal"); return DOTS_ERR_ARGUMENT; }
    }
    for (int f = 0; f < F; ++f)
        for (int k = 0; k < 3; ++k)
            if (p->triangles[f * 3 + k] < 0 || p->triangles[f * 3 + k] >= V) { set_error("triangle index out of range"); return DOTS_ERR_ARGUMENT; }

    int rc = 0;
#define UP(field, src, n) if ((rc = dev_upload(c, &d.field, src, (int64_t)(n)))) return rc
    UP(tri, p->triangles, F * 3);
    UP(hat, p->hat_grad, F * 9);
    UP(area_f, p->area_tri, F);
    UP(mass_v, p->mass_vert, V);
    UP(cptr, p->corner_ptr, V + 1);
    UP(cidx, p->corner_idx, nC);
    UP(c_D, cD.data(), nC);
    {
        std::vector<double> fkD((size_t)nC);
        for (int j = 0; j < nC; ++j) fkD[(size_t)p->corner_idx[j]] = cD[(size_t)j];
        UP(fk_D, fkD.data(), nC);
    }
    {
        std::vector<int> cpos((size_t)nC);
        for (int j = 0; j < nC; ++j) cpos[(size_t)p->corner_idx[j]] = j;
        UP(cpos, cpos.data(), nC);
    }
    UP(c_gA, cgA.data(), nC * 3);
    UP(c_area, cArea.data(), nC);
    UP(rowptr, p->lap_rowptr, V + 1);
    UP(col, p->lap_col, p->lap_nnz);
    UP(val, p->lap_val, p->lap_nnz);
    UP(kdiag, kdiag.data(), V);
    UP(mu0, p->mu0, V);
    UP(mu1, p->mu1, V);
    if (p->perm_vert) UP(perm_v, p->perm_vert, V);
    if (p->perm_tri) UP(perm_f, p->perm_tri, F);
    if (p->lap_solver == DOTS_LAP_MODAL_PCG) {
        if (!p->time_modes || !p->time_eigs) { set_error("modal solver needs time_modes and time_eigs"); return DOTS_ERR_ARGUMENT; }
        UP(Q, p->time_modes, (d.T + 1) * (d.T + 1));
        {
            const int n = d.T + 1;
            std::vector<double> qp((size_t)tpg * tpg, 0.0), qt((size_t)tpg * tpg, 0.0);
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j) {
                    qp[(size_t)i * tpg + j] = p->time_modes[i * n + j];
                    qt[(size_t)j * tpg + i] = p->time_modes[i * n + j];
                }
            UP(Qpad, qp.data(), (int64_t)tpg * tpg);
            UP(QpadT, qt.data(), (int64_t)tpg * tpg);
        }
        std::vector<double> sig(tpg, 0.0);
        for (int i = 0; i <= d.T; ++i) sig[i] = p->time_eigs[i];
        UP(sigma, sig.data(), tpg);
    }
    // patch tiles of the right-hand-side / projection launch (k_rhs_soc_tiles: one GPU, T + 1 < 64)
    if (p->patch_order && !sharded && p->lap_solver == DOTS_LAP_MODAL_PCG && tpg >= 4 && tpg <= 32) {
        TileDev &tl = c->tiles;
        tl.VTL = 512 / tpg;
        tl.n_tiles = (V + tl.VTL - 1) / tl.VTL;
        std::vector<int> tv((size_t)std::max(tl.n_tiles * tl.VTL, d.n_vtiles * d.VT), -1), tptr((size_t)tl.n_tiles + 1, 0), ttri, cloc((size_t)nC, 0), stamp((size_t)F, -1), pos((size_t)F, 0);
        std::vector<char> seen((size_t)V, 0);
        for (int i = 0; i < V; ++i) {
            const int v = p->patch_order[i];
            if (v < 0 || v >= V || seen[(size_t)v]) { set_error("patch_order is not a permutation of the vertices"); return DOTS_ERR_ARGUMENT; }
            seen[(size_t)v] = 1;
            tv[(size_t)i] = v;
        }
        for (int t = 0; t < tl.n_tiles; ++t) {
            int count = 0;
            for (int k = 0; k < tl.VTL; ++k) {
                const int v = tv[(size_t)t * tl.VTL + k];
                if (v < 0) continue;
                for (int j = p->corner_ptr[v]; j < p->corner_ptr[v + 1]; ++j) {
                    const int f = p->corner_idx[j] / 3;
                    if (stamp[(size_t)f] != t) { stamp[(size_t)f] = t; pos[(size_t)f] = count++; ttri.push_back(f); }
                    cloc[(size_t)j] = pos[(size_t)f];
                }
            }
            tptr[(size_t)t + 1] = (int)ttri.size();
            tl.ntri_max = std::max(tl.ntri_max, count);
        }
        if (ttri.empty()) ttri.push_back(0);
        UP(tiles_vertex, tv.data(), tv.size());
        UP(tiles_tri_ptr, tptr.data(), tptr.size());
        UP(tiles_tri, ttri.data(), ttri.size());
        UP(tiles_c_loc, cloc.data(), cloc.size());
        tl.vertex = d.tiles_vertex; tl.tri_ptr = d.tiles_tri_ptr; tl.tri = d.tiles_tri; tl.c_loc = d.tiles_c_loc;
    }
#undef UP

    double **state[12] = {&d.phi, &d.A, &d.B, &d.lam, &d.zf, &d.zm, &d.ze, &d.mu, &d.E, &d.bf, &d.bm, &d.be};
    for (int id = 0; id < DOTS_N_ARRAYS; ++id)
        if ((rc = dev_alloc(c, state[id], array_count_device(d, id)))) return rc;
    const int64_t nnode = (int64_t)V << sh;
    if (!sharded && (rc = dev_alloc(c, &d.cg_b, nnode))) return rc;      // a slab writes its right-hand side into slab.b_send
    if ((rc = dev_alloc(c, &d.lamc, nnode))) return rc;
    if (!sharded && ((rc = dev_alloc(c, &c->zf_alt, nnode)) || (rc = dev_alloc(c, &c->ze_alt, nnode)) || (rc = dev_alloc(c, &c->lamc_alt, nnode)))) return rc;
    if (!sharded && c->zmid_defer && p->lap_solver == DOTS_LAP_MODAL_PCG && (rc = dev_alloc(c, &c->B_alt, array_count_device(d, DOTS_B)))) return rc;
    d.B_st = d.B;
    d.bm_st = d.bm;
    if (sharded) {
        double *hv = nullptr;
        if ((rc = dev_alloc(c, &hv, V))) return rc;
        d.phi_hi = hv;
    }
    if (!sharded) {
        double **cgv[6] = {&d.cg_r, &d.cg_z, &d.cg_p0, &d.cg_p1, &d.cg_Ap, &d.cg_x};
        for (auto q : cgv)
            if ((rc = dev_alloc(c, q, nnode))) return rc;
    }
    const int gv = xcd_grid(d.n_vtiles), gf = xcd_grid(d.n_ftiles);
    // KKT kernels: one workgroup per quarter tile, N_VSUMS + N_FSUMS <= MAX_SUMS slots each
    const int64_t npart = std::max<int64_t>({(int64_t)MAX_SUMS * std::max(gv, gf) * (TILE_ELEMS / BLOCK) * 2, cg_partials_needed(d), 4096});
    if ((rc = dev_alloc(c, &d.partials, npart))) return rc;
    if ((rc = dev_alloc(c, &d.scal, CgScalOffsets::TOTAL))) return rc;
    if ((rc = dev_alloc(c, &d.flags, FLAG_TOTAL))) return rc;
    if (!sharded && tp >= 4) {      // DOTS_STEP_KKT_SUMS: per-workgroup partial sums of the steps-2+3 launch (carry or tile mapping)
        const int64_t tw = tp <= 128 ? (2 * 192 / tp) / 3 : 1;
        c->kkt_fused_cap_v = gv;
        c->kkt_fused_cap_f = xcd_grid((int)((F + tw - 1) / tw));
        if ((rc = dev_alloc(c, &c->kkt_fused.part_v, (int64_t)N_VSUMS * c->kkt_fused_cap_v))) return rc;
        if ((rc = dev_alloc(c, &c->kkt_fused.part_f, (int64_t)N_FSUMS * c->kkt_fused_cap_f))) return rc;
    }
    c->stage_count = array_count_device(d, DOTS_Z_MID);
    if ((rc = dev_alloc(c, &c->stage, c->stage_count))) return rc;
    DOTS_HIP(hipHostMalloc((void **)&c->h_pinned, sizeof(double) * CgScalOffsets::TOTAL, hipHostMallocDefault));
    DOTS_HIP(hipHostMalloc((void **)&c->h_flags, sizeof(int) * FLAG_TOTAL, hipHostMallocDefault));
    if ((rc = dev_alloc(c, &c->kkt_counter, 2))) return rc;
    DOTS_HIP(hipHostMalloc((void **)&c->h_mail, sizeof(double) * (MAX_SUMS + 8), hipHostMallocCoherent | hipHostMallocMapped));
    for (int i = 0; i < MAX_SUMS + 8; ++i) c->h_mail[i] = 0.0;

    // the PCG's view of the device data (see Ctx::dcg), and on a slab the global-time view of the transforms
    c->dcg = d;
    c->dgt = d;
    if (sharded) {
        Dev &g = c->dcg;           // modes [shard_begin, shard_begin + shard_count): same partition, same pitch as the slab
        g.cg_ncol = p->slab_count;
        std::vector<double> sig(tp, 0.0);
        for (int i = 0; i < p->slab_count; ++i) sig[i] = p->time_eigs[p->slab_begin + i];
        if ((rc = dev_upload(c, &g.sigma, sig.data(), tp))) return rc;
        double **cgv[5] = {&g.cg_r, &g.cg_z, &g.cg_p0, &g.cg_p1, &g.cg_Ap};     // cg_x = slab.x_send (dots_slab_set_buffers)
        for (auto q : cgv)
            if ((rc = dev_alloc(c, q, nnode))) return rc;
        Dev &t = c->dgt;
        t.TP = tpg; t.tp_shift = shg; t.t0 = 0; t.nl = d.T + 1; t.ni = d.T; t.slab = 0;
        t.VT = t.FT = TILE_ELEMS / tpg;
        t.n_vtiles = (d.V + t.VT - 1) / t.VT;
        t.n_ftiles = (3 * d.F + t.FT - 1) / t.FT;
        c->slab_b_chunk = nnode;
        c->slab_x_chunk = nnode + V;
    }

    // KKT normalisation constants (solver_socp.py:303-313): means of the broadcast weight arrays
    double sm = 0.0, sa = 0.0;
    for (int v = 0; v < V; ++v) sm += p->mass_vert[v];
    for (int f = 0; f < F; ++f) sa += p->area_tri[f];
    const double mean_v = sm / V, mean_f = sa / F;
    c->c_prim_q = 0.5 * (mean_v + mean_f);
    c->c_prim_z = (mean_v + mean_f + mean_v) / 3.0;
    c->c_dual_alpha = mean_v;
    c->c_dual_beta = 0.5 * (mean_v + mean_f);
    c->c_comp_rho = mean_v;
    c->c_comp_m = mean_f;

    dots_params &q = c->prm;
    q.r = 1.0; q.scale_z = 1.0; q.const_d = 1.0; q.norm_d = std::sqrt(2.0 * sa);
    double nb = 0.0;
    for (int v = 0; v < V; ++v) nb += (p->mu0[v] * p->mu0[v] + p->mu1[v] * p->mu1[v]) / p->mass_vert[v];
    q.norm_boundary = std::sqrt(nb / (d.T + 1));      // = r h sqrt(norm_square_center(boundary / mass)), :296
    q.congestion = 0.0; q.tau = 1.9; q.eps = 0.0; q.prim_scale = 1.0; q.dual_scale = 1.0; q.boundary_scale = 1.0;
    q.cg_tol = 1e-8; q.cg_max_iter = 20000;
    DOTS_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

// `reads_only`: the entry point changes neither the state nor the parameters.  Every other one drops a right-hand side that was
// enqueued ahead of its iteration (DOTS_STEP_RHS_AHEAD): the next dots_step then computes it again.
// A pending penalty division (Ctx::pending_div) carried out now: the stand-alone pass over the five dual arrays.
static int flush_division(Ctx *c) {
    if (c->pending_div == 0.0) return 0;
    const double f = c->pending_div;
    c->pending_div = 0.0;
    return launch_adjust_penalty(c, f);
}

// `keeps_division`: the entry point neither reads nor writes the dual arrays (or, dots_step, applies a pending division itself).
static int check(dots_ctx *ctx, bool reads_only = false, bool keeps_division = false) {
    if (!ctx) { set_error("null context"); return DOTS_ERR_ARGUMENT; }
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) return hip_fail(e, "hipSetDevice", __FILE__, __LINE__);
    if (!reads_only) ctx->rhs_ahead = ctx->rhs_ahead_armed = ctx->penalty_armed = ctx->carry_valid = ctx->kkt_fused_valid = 0;      // (the carried gathers / fused sums belong to the state steps 2+3 left)
    if (!keeps_division) return flush_division(ctx);
    return 0;
}

// first half of an iteration: right-hand side + solve (for this context's modes)
static int palm_step0(Ctx *c) {
    if (!c->step_palm) return 0;
    if (c->zmid_stale) { set_error("step: DOTS_STEP_PALM needs z_mid of the previous iteration in memory"); return DOTS_ERR_STATE; }
    c->carry_valid = c->kkt_fused_valid = 0;      // step 0 moves A, B and lambda_c
    return launch_q_lambda_only(c);
}

// ---- time slabs: the stages of one iteration (include/dots_socp_hip.h, dots_slab_stage) ---------------------------
__global__ __launch_bounds__(BLOCK) void k_slab_append_multiplier(Dev d, double *__restrict__ tail) {
    const int v = blockIdx.x * BLOCK + threadIdx.x;
    // the cone multiplier of the interval that ends at the next slab's first node
    if (v < d.V) tail[v] = (d.ni == d.nl && d.nl > 0) ? d.lamc[idxV(d, v, d.nl - 1)] : 0.0;
}

static int slab_stage(Ctx *c, int stage) {
    int rc;
    const Dev &d = c->d;
    switch (stage) {
        case 0:       // [is_palm: step 0]; halos for the right-hand side and the projection
            if (d.nl > 0 && (rc = palm_step0(c))) return rc;
            return launch_slab_pack_iteration(c);
        case 1:       // right-hand side of this slab's nodes -> b_send, and the cone projection of its intervals (one launch)
        case 6:       // ... the right-hand side alone: the caller starts the all-gather of b behind it and enqueues stage 5
        case 5:       // ... the cone projection alone (it reads nothing the right-hand side or the solve writes); the multipliers of
                      //     the slab's last interval go to the tail of x_send: the NEXT slab's steps 2+3 need them, after the solve
            if (d.nl > 0) {
                if (stage == 5) { if ((rc = launch_soc_projection(c, 1, false))) return rc; }
                else if ((rc = launch_rhs(c, stage == 1))) return rc;
                if (stage != 6) {
                    hipLaunchKernelGGL(k_slab_append_multiplier, dim3((d.V + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, c->stream, d,
                                       c->slab.x_send + ((int64_t)d.V << d.tp_shift));
                    DOTS_HIP(hipGetLastError());
                }
            }
            return 0;
        case 2:       // forward transform of this rank's modes + solve -> x_send
            return cg_solve(c, nullptr);
        case 3:       // inverse transform for this slab, steps 2 and 3
            if ((rc = cg_finish_sharded(c))) return rc;
            c->zmid_stale = c->step_skip_zmid;
            c->kkt_halo_fresh = 0;
            if (d.nl == 0) return 0;
            return launch_q_lambda_mult(c, c->step_skip_zmid ? 2 : 1);
        case 4:       // halos of the KKT kernels
            c->kkt_halo_fresh = 1;
            return launch_slab_pack_kkt(c);
        default:
            set_error("slab_stage: unknown stage");
            return DOTS_ERR_ARGUMENT;
    }
}

// a free slot of the timing ring (its events created on first use), or nullptr when the ring is full: the step is then not timed
constexpr int TKIND_FUSED = 16;      // tkind of a whole iteration whose projection rode in the right-hand-side launch
static hipEvent_t *time_slot(Ctx *c, int kind) {
    if (!c->step_timed || c->t_count >= Ctx::TIME_SLOTS) return nullptr;
    const int s = (c->t_head + c->t_count) % Ctx::TIME_SLOTS;
    for (auto &e : c->tev[s])
        if (!e && hipEventCreate(&e) != hipSuccess) return nullptr;
    c->tkind[s] = kind;
    c->t_count += 1;
    return c->tev[s];
}

static int run_iteration_body(Ctx *c, dots_step_stats *st, hipEvent_t *tv) {
    int rc;
#define MARK(i) do { if (tv) DOTS_HIP(hipEventRecord(tv[i], c->stream)); } while (0)
    MARK(0);
    if (c->zmid_deferred) {      // nobody asked for the last iterate's z_mid: its storage (holding the old beta_mid) is simply stale
        if (c->step_palm) { if ((rc = materialise_zmid(c))) return rc; }      // (step 0 reads z_mid)
        else { c->zmid_deferred = 0; c->zmid_stale = 1; }
    }
    // a pending penalty division rides in this iteration's kernels when both of them can apply it; otherwise it is carried out first
    const int zmode = c->step_skip_zmid ? 2 : 1;
    double dv = 0.0;
    if (c->pending_div != 0.0) {
        if (!st && !c->step_palm && c->rhs_ahead == 2 && c->ahead_div == c->pending_div && ql_divides(c, zmode)) {
            dv = c->pending_div;      // the launch ahead (penalty_decision_ahead) divided as it read: steps 2+3 do the same and write back divided
            c->pending_div = 0.0;
        } else if (!st && !c->step_palm && !c->rhs_ahead && rhs_takes_soc(c) && rhs_divides(c) && ql_divides(c, zmode)) {
            dv = c->pending_div;
            c->pending_div = 0.0;
        } else if ((rc = flush_division(c))) return rc;      // (a launch ahead that divided as it read saw the values the arrays now hold)
    }
    c->ahead_div = 0.0;
    if ((rc = palm_step0(c))) return rc;
    // the right-hand side of this iteration was enqueued behind the KKT kernels of the last one (DOTS_STEP_RHS_AHEAD) and nothing
    // it reads has changed since: start at the solve; the projection then runs with the inverse transform
    const int ahead_kind = (c->step_palm || c->rhs_ahead > 2) ? 0 : c->rhs_ahead;      // (3, 4: an anticipated penalty update the caller did not confirm)
    const bool ahead = ahead_kind != 0;
    c->rhs_ahead = 0;
    if (ahead_kind == 2) {      // the projection ran ahead too: its results become the current z_fst, z_end and cone multiplier
        std::swap(c->d.zf, c->zf_alt);
        std::swap(c->d.ze, c->ze_alt);
        std::swap(c->d.lamc, c->lamc_alt);
        c->dcg.zf = c->dgt.zf = c->d.zf;
        c->dcg.ze = c->dgt.ze = c->d.ze;
        c->dcg.lamc = c->dgt.lamc = c->d.lamc;
    }
    if (!st) {   // asynchronous: enqueue only (the direct solver needs no host round trip); nothing is timed
        c->zmid_stale = c->step_skip_zmid;
        if (rhs_takes_soc(c) && !ahead) {   // [right-hand side + projection] -> sweeps -> inverse transform -> steps 2+3
            if ((rc = launch_rhs(c, true, dv))) return rc;
            MARK(1);
            if ((rc = cg_solve(c, nullptr))) return rc;
            MARK(2);
            MARK(3);      // (no separate projection launch: dots_step_times splits the first phase between ms_rhs and ms_soc)
            if ((rc = launch_q_lambda_mult(c, zmode, dv))) return rc;
            MARK(4);
            MARK(5);      // back to back with 4: what one event costs on the stream, taken off every phase (dots_step_times)
            if (tv) c->tkind[(c->t_head + c->t_count - 1) % Ctx::TIME_SLOTS] = TKIND_FUSED;
            return 0;
        }
        const bool fuse = soc_takes_inverse(c) && ahead_kind != 2;
        if (!ahead && (rc = launch_rhs(c))) return rc;
        MARK(1);
        if ((rc = cg_solve(c, nullptr, fuse))) return rc;
        MARK(2);
        if (ahead_kind != 2 && (rc = launch_soc_projection(c, 1, fuse))) return rc;
        MARK(3);
        if ((rc = launch_q_lambda_mult(c, zmode, dv))) return rc;
        MARK(4);
        MARK(5);
        return 0;
    }
#undef MARK
    DOTS_HIP(hipEventRecord(c->ev[0], c->stream));
    if (!ahead && (rc = launch_rhs(c))) return rc;
    DOTS_HIP(hipEventRecord(c->ev[1], c->stream));
    const bool fuse = soc_takes_inverse(c) && ahead_kind != 2;
    if ((rc = cg_solve(c, st, fuse))) return rc;
    DOTS_HIP(hipEventRecord(c->ev[2], c->stream));
    if (ahead_kind != 2 && (rc = launch_soc_projection(c, 1, fuse))) return rc;
    DOTS_HIP(hipEventRecord(c->ev[3], c->stream));
    if ((rc = launch_q_lambda_mult(c, c->step_skip_zmid ? 2 : 1))) return rc;
    c->zmid_stale = c->step_skip_zmid;
    DOTS_HIP(hipEventRecord(c->ev[4], c->stream));
    DOTS_HIP(hipEventSynchronize(c->ev[4]));
    float t;
    DOTS_HIP(hipEventElapsedTime(&t, c->ev[0], c->ev[1])); st->ms_rhs += t;
    DOTS_HIP(hipEventElapsedTime(&t, c->ev[1], c->ev[2])); st->ms_laplacian += t;
    DOTS_HIP(hipEventElapsedTime(&t, c->ev[2], c->ev[3])); st->ms_soc += t;
    DOTS_HIP(hipEventElapsedTime(&t, c->ev[3], c->ev[4])); st->ms_q_lambda_multiplier += t;
    DOTS_HIP(hipEventElapsedTime(&t, c->ev[0], c->ev[4])); st->ms_total += t;
    st->alm_iterations += 1;
    return 0;
}

// A slot of the timing ring is taken before the phases are enqueued: if enqueueing fails midway the slot is given back, so that a
// later dots_step_times never waits on events that were not recorded (and does not hide the original error behind its own).
static int run_iteration(Ctx *c, dots_step_stats *st) {
    hipEvent_t *tv = st ? nullptr : time_slot(c, 0);
    const int rc = run_iteration_body(c, st, tv);
    if (rc && tv) c->t_count -= 1;
    return rc;
}

static void mg_release(Ctx *c) {
    for (int i = 0; i < c->n_mg_allocs; ++i) (void)hipFree(c->mg_allocs[i]);
    c->n_mg_allocs = 0;
    c->mg = MgDev{};
}

template <typename T>
static int mg_upload(Ctx *c, const T **out, const T *host, int64_t count) {
    void *p = nullptr;
    const size_t bytes = sizeof(T) * (size_t)std::max<int64_t>(count, 1);
    DOTS_HIP(hipMalloc(&p, bytes));
    if (c->n_mg_allocs >= (int)(sizeof(c->mg_allocs) / sizeof(c->mg_allocs[0]))) {
        (void)hipFree(p);
        set_error("multigrid allocation table full");
        return DOTS_ERR_STATE;
    }
    c->mg_allocs[c->n_mg_allocs++] = p;
    // on the context's own (non-blocking) stream: never touch the legacy default stream, another
    // context of this process may be capturing a graph
    if (host) DOTS_HIP(hipMemcpyAsync(p, host, sizeof(T) * (size_t)count, hipMemcpyHostToDevice, c->stream));
    else DOTS_HIP(hipMemsetAsync(p, 0, bytes, c->stream));
    DOTS_HIP(hipStreamSynchronize(c->stream));
    *out = (const T *)p;
    return 0;
}

}  // namespace dots

using namespace dots;

extern "C" {

int dots_abi_version(void) { return DOTS_ABI_VERSION; }
const char *dots_last_error(void) { return g_last_error.c_str(); }

int dots_create(const dots_problem_desc *desc, dots_ctx **out) {
    if (!desc || !out) { set_error("null argument"); return DOTS_ERR_ARGUMENT; }
    *out = nullptr;
    if (desc->abi_version != DOTS_ABI_VERSION) { set_error("ABI version mismatch"); return DOTS_ERR_ARGUMENT; }
    if (desc->n_time < 1 || desc->n_vertices < 3 || desc->n_triangles < 1 || desc->n_corners != 3 * desc->n_triangles) {
        set_error("bad problem sizes");
        return DOTS_ERR_ARGUMENT;
    }
    if (!desc->triangles || !desc->hat_grad || !desc->area_tri || !desc->mass_vert || !desc->corner_ptr || !desc->corner_idx ||
        !desc->lap_rowptr || !desc->lap_col || !desc->lap_val || !desc->mu0 || !desc->mu1) {
        set_error("null array in problem description");
        return DOTS_ERR_ARGUMENT;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { set_error("no HIP device"); return DOTS_ERR_NO_DEVICE; }
    if (desc->device < 0 || desc->device >= ndev) { set_error("device ordinal out of range"); return DOTS_ERR_ARGUMENT; }
    DOTS_HIP(hipSetDevice(desc->device));
    dots_ctx *c = new dots_ctx();
    c->device = desc->device;
    // measurement switches: every value is validated -- a typo must not silently select the default
    {
        bool ok = true;
        ok &= env_int("DOTS_CG_STAGE_LDS", 0, 1, &c->cg_stage_lds);
        ok &= env_int("DOTS_MG_TAIL_ROWS", 0, 1 << 20, &c->mg_tail_rows);
        ok &= env_int("DOTS_SOC_WITH_RHS", 0, 1, &c->soc_with_rhs);
        ok &= env_int("DOTS_QL_TWO", 0, 1, &c->ql_two);
        ok &= env_int("DOTS_KKT_TWO", 0, 1, &c->kkt_two);
        ok &= env_int("DOTS_RHS_TWO", 0, 1, &c->rhs_two);
        ok &= env_int("DOTS_RHS_TILES", 0, 2, &c->rhs_tiles);
        ok &= env_int("DOTS_ZMID_DEFER", 0, 1, &c->zmid_defer);    // 0: read-back iterations store z_mid as before
        ok &= env_int("DOTS_LAZY_DIV", 0, 1, &c->lazy_div);        // 0: a penalty update divides the dual arrays at once
        ok &= env_int("DOTS_CARRY", 0, 1, &c->carry_arrays);       // 0: never allocate the carried gathers (DOTS_STEP_CARRY is then ignored)
        ok &= env_int("DOTS_SPIN_FETCH", 0, 1, &c->spin_fetch);
        ok &= env_int("DOTS_FRONT_VEC2", 0, 3, &c->front_vec2);      // 0 never, 1 / 2 wherever the pitch allows (default), 3 only where bandwidth-bound
        ok &= env_int("DOTS_FRONT_RB", 1, 4, &c->front_rb_max);
        ok &= env_int("DOTS_FRONT_ROWS", 0, 2, &c->front_rows);      // 0: the fold kernels everywhere, 1: row kernels where the rules say (default), 2: wherever they fit
        ok &= env_int("DOTS_FRONT_XCD", 0, 1, &c->front_xcd);
        ok &= env_int("DOTS_FRONT_LEAFINV", 0, 2, &c->front_leafinv);
        ok &= env_int("DOTS_FRONT_TUNE", 0, 2, &c->front_tune);
        ok &= env_int("DOTS_MAIL_TEST_DROP", 0, 1 << 20, &c->mail_test_drop);
        int spins = -1;
        ok &= env_int("DOTS_MAIL_SPINS", 0, 2000000000, &spins);
        if (spins >= 0) c->mail_spins = spins;
        if (!ok) { delete c; return DOTS_ERR_ARGUMENT; }
    }
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete c; return hip_fail(e, "hipStreamCreate", __FILE__, __LINE__); }
    for (auto &ev : c->ev) (void)hipEventCreate(&ev);
    int rc = build(c, desc);
    if (rc) { dots_destroy(c); return rc; }
    preload_alm_kernels();
    preload_kkt_kernels();
    preload_transform_kernels();
    *out = c;
    return 0;
}

int dots_destroy(dots_ctx *c) {
    if (!c) return 0;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->cg_graph) (void)hipGraphExecDestroy(c->cg_graph);
    for (int i = 0; i < c->n_mg_allocs; ++i) (void)hipFree(c->mg_allocs[i]);
    for (int i = 0; i < c->n_front_allocs; ++i) (void)hipFree(c->front_allocs[i]);
    for (int i = 0; i < c->n_allocs; ++i) (void)hipFree(c->allocs[i]);
    if (c->h_pinned) (void)hipHostFree(c->h_pinned);
    if (c->h_flags) (void)hipHostFree(c->h_flags);
    if (c->h_mail) (void)hipHostFree(c->h_mail);
    for (auto &ev : c->ev) if (ev) (void)hipEventDestroy(ev);
    for (auto &slot : c->tev)
        for (auto &ev : slot) if (ev) (void)hipEventDestroy(ev);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return 0;
}

int dots_set_params(dots_ctx *c, const dots_params *p) {
    // the penalty update the library anticipated (penalty_decision_ahead): only r moves, to the anticipated value
    const bool anticipated = c && p && c->rhs_ahead == 4 && p->r == c->ahead_r && p->scale_z == c->prm.scale_z && p->const_d == c->prm.const_d &&
                             p->eps == c->prm.eps && p->boundary_scale == c->prm.boundary_scale && p->congestion == c->prm.congestion && p->tau == c->prm.tau;
    int rc = check(c, false, true);
    if (rc) return rc;
    if (!p || !(p->r > 0) || !(p->scale_z > 0) || !(p->cg_tol > 0) || p->eps < 0 || !(p->boundary_scale > 0)) { set_error("bad parameters"); return DOTS_ERR_ARGUMENT; }
    c->prm = *p;
    if (anticipated && c->pending_div == c->ahead_dv) {
        c->rhs_ahead = 2;
        c->ahead_div = c->ahead_dv;
        c->penalty_ahead_confirmed += 1;
    }
    return 0;
}

int dots_penalty_ahead(dots_ctx *c, const dots_penalty_policy *policy) {
    int rc = check(c, true, true);
    if (rc) return rc;
    if (!policy || policy->n_steps < 0 || policy->n_steps > 16 || !(policy->tol > 0) || !(policy->r_lower > 0) || !(policy->r_upper >= policy->r_lower)) {
        set_error("penalty_ahead: bad policy");
        return DOTS_ERR_ARGUMENT;
    }
    c->penalty_policy = *policy;
    c->penalty_armed = 1;
    return 0;
}
int dots_get_params(dots_ctx *c, dots_params *p) {
    if (!c || !p) { set_error("null argument"); return DOTS_ERR_ARGUMENT; }
    *p = c->prm;
    return 0;
}
int dots_sync(dots_ctx *c) {
    int rc = check(c, true, true);
    if (rc) return rc;
    DOTS_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

int64_t dots_array_count(dots_ctx *c, int id) {
    if (!c || id < 0 || id >= DOTS_N_ARRAYS) return -1;
    return array_count_host(c->d, id);
}
int64_t dots_device_bytes(dots_ctx *c) { return c ? c->bytes : -1; }

int dots_upload(dots_ctx *c, int id, const double *host, int64_t count) {
    int rc = check(c);
    if (rc) return rc;
    if (id < 0 || id >= DOTS_N_ARRAYS || !host || count != array_count_host(c->d, id)) { set_error("upload: bad array id or element count"); return DOTS_ERR_ARGUMENT; }
    if (id != DOTS_Z_MID && (rc = materialise_zmid(c))) return rc;
    DOTS_HIP(hipMemcpyAsync(c->stage, host, sizeof(double) * (size_t)count, hipMemcpyHostToDevice, c->stream));
    if ((rc = launch_to_device_layout(c, id, c->stage))) return rc;
    DOTS_HIP(hipStreamSynchronize(c->stream));
    if (id == DOTS_Z_MID) { c->zmid_stale = 0; c->zmid_deferred = 0; }      // (the upload is what z_mid's storage now holds)
    c->kkt_halo_fresh = 0;
    return 0;
}
int dots_download(dots_ctx *c, int id, double *host, int64_t count) {
    int rc = check(c, true);
    if (rc) return rc;
    if (id < 0 || id >= DOTS_N_ARRAYS || !host || count != array_count_host(c->d, id)) { set_error("download: bad array id or element count"); return DOTS_ERR_ARGUMENT; }
    if (id == DOTS_Z_MID && c->zmid_stale) { set_error("download: z_mid was not materialised by the last step (dots_step_flags)"); return DOTS_ERR_STATE; }
    if (id == DOTS_Z_MID && (rc = materialise_zmid(c))) return rc;
    if ((rc = launch_from_device_layout(c, id, c->stage))) return rc;
    DOTS_HIP(hipMemcpyAsync(host, c->stage, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, c->stream));
    DOTS_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

int64_t dots_slab_elems(dots_ctx *c, int which) {
    if (!c || c->shard_stride == 0) return -1;
    const int64_t nnode = (int64_t)c->d.V << c->d.tp_shift;
    switch (which) {
        case DOTS_SLAB_VERTEX_HALO: return c->d.V;
        case DOTS_SLAB_B_CHUNK: return nnode;
        case DOTS_SLAB_X_CHUNK: return nnode + c->d.V;
        case DOTS_SLAB_TRIANGLE_HALO: return (int64_t)3 * c->d.F;
        default: return -1;
    }
}

int dots_slab_set_buffers(dots_ctx *c, const dots_slab_buffers *b) {
    int rc = check(c);
    if (rc) return rc;
    if (c->shard_stride == 0) { set_error("slab_set_buffers: context is not a time slab"); return DOTS_ERR_STATE; }
    if (!b || !b->send_x || !b->send_nsq || !b->recv_x || !b->recv_nsq || !b->b_send || !b->b_recv || !b->x_send || !b->x_recv ||
        !b->send_mu || !b->send_b || !b->recv_mu || !b->recv_b) {
        set_error("slab_set_buffers: null buffer");
        return DOTS_ERR_ARGUMENT;
    }
    DOTS_HIP(hipStreamSynchronize(c->stream));
    c->slab = *b;
    Dev &d = c->d;
    const int rank = c->shard_begin / c->shard_stride;
    d.cg_b = b->b_send;                         // the right-hand side of this slab is written where the all-gather reads it
    d.X_lo = b->recv_x;
    d.nsq_hi = b->recv_nsq;
    d.mu_lo = b->recv_mu;
    d.B_hi = b->recv_b;
    // the previous slab appended its last interval's cone multipliers to its chunk of the SOLUTION all-gather (they are needed by
    // steps 2+3, after it: the right-hand-side all-gather can then start before the projection has run)
    d.lamc_lo = (rank > 0 && d.nl > 0) ? b->x_recv + (int64_t)(rank - 1) * c->slab_x_chunk + ((int64_t)d.V << d.tp_shift) : b->recv_x;
    const Dev g0 = c->dcg, t0 = c->dgt;
    c->dcg = d;                                 // the solver's view: same pitch, its own vectors, sigma slice, mode count
    c->dcg.cg_ncol = g0.cg_ncol; c->dcg.sigma = g0.sigma;
    c->dcg.cg_r = g0.cg_r; c->dcg.cg_z = g0.cg_z; c->dcg.cg_p0 = g0.cg_p0; c->dcg.cg_p1 = g0.cg_p1; c->dcg.cg_Ap = g0.cg_Ap;
    c->dcg.cg_x = b->x_send;                    // ... the mode-space solution is written where the second all-gather reads it
    c->dgt = d;                                 // the transforms' view: same pointers, global-time geometry
    c->dgt.TP = t0.TP; c->dgt.tp_shift = t0.tp_shift; c->dgt.t0 = 0; c->dgt.nl = d.T + 1; c->dgt.ni = d.T; c->dgt.slab = 0;
    c->dgt.VT = t0.VT; c->dgt.FT = t0.FT; c->dgt.n_vtiles = t0.n_vtiles; c->dgt.n_ftiles = t0.n_ftiles;
    // the PCG's warm start (its own previous solution) and the buffers' padding start from zero
    DOTS_HIP(hipMemsetAsync(b->x_send, 0, sizeof(double) * (size_t)c->slab_x_chunk, c->stream));
    DOTS_HIP(hipMemsetAsync(b->b_send, 0, sizeof(double) * (size_t)c->slab_b_chunk, c->stream));
    DOTS_HIP(hipStreamSynchronize(c->stream));
    c->slab_stage = 0;
    return 0;
}

int dots_slab_stage(dots_ctx *c, int stage, dots_step_stats *stats) {
    int rc = check(c, true);      // (the stages say themselves what they invalidate: carried sums live from stage 3 to the next stages 0 and 1)
    if (rc) return rc;
    if (c->shard_stride == 0) { set_error("slab_stage: context is not a time slab"); return DOTS_ERR_STATE; }
    if (!c->slab.b_send) { set_error("slab_stage: no exchange buffers (dots_slab_set_buffers)"); return DOTS_ERR_STATE; }
    if (stage < 0 || stage > 6) { set_error("slab_stage: unknown stage"); return DOTS_ERR_ARGUMENT; }
    // order: 0, 1, 2, 3 -- or with stage 1 in two halves: 0, 6, 5, 2, 3 (slab_stage = the next stage expected; 5: the projection is due)
    const bool in_order = stage == 4 || stage == c->slab_stage || (stage == 6 && c->slab_stage == 1);
    if (!in_order) { set_error("slab_stage: stages must be called in the order 0, 1, 2, 3 (or 0, 6, 5, 2, 3)"); return DOTS_ERR_STATE; }
    if (stage == 4 && c->slab_stage != 0) { set_error("slab_stage: the KKT halos are packed between iterations"); return DOTS_ERR_STATE; }
    if (!stats) {
        hipEvent_t *tv = stage != 4 ? time_slot(c, 1 + stage) : nullptr;
        rc = tv ? (int)hipEventRecord(tv[0], c->stream) : 0;
        if (!rc) rc = slab_stage(c, stage);
        if (!rc && tv && hipEventRecord(tv[1], c->stream) != hipSuccess) rc = DOTS_ERR_HIP;
        if (rc) {      // (the slot goes back to the ring: see run_iteration)
            if (tv) c->t_count -= 1;
            if (rc > 0) { set_error("slab_stage: hipEventRecord failed"); rc = DOTS_ERR_HIP; }
            return rc;
        }
    } else {
        DOTS_HIP(hipEventRecord(c->ev[0], c->stream));
        if ((rc = slab_stage(c, stage))) return rc;
        DOTS_HIP(hipEventRecord(c->ev[1], c->stream));
        DOTS_HIP(hipEventSynchronize(c->ev[1]));
        float t;
        DOTS_HIP(hipEventElapsedTime(&t, c->ev[0], c->ev[1]));
        memset(stats, 0, sizeof *stats);
        stats->ms_total = t;
        if (stage == 0 || stage == 6) stats->ms_rhs = t;              // packing the halos (as dots_step_times books it); the right-hand side alone
        if (stage == 5) stats->ms_soc = t;
        if (stage == 1) stats->ms_rhs = stats->ms_soc = 0.5 * t;      // one launch: right-hand side and projection together
        if (stage == 2) { stats->ms_laplacian = t; stats->cg_iterations = stats->cg_last_iterations = c->last_cg_iters; }
        if (stage == 3) { stats->ms_q_lambda_multiplier = t; stats->alm_iterations = 1; }
    }
    if (stage <= 3) c->slab_stage = (stage + 1) & 3;
    else if (stage == 6) c->slab_stage = 5;
    else if (stage == 5) c->slab_stage = 2;
    return 0;
}

int dots_stream_wait(dots_ctx *c, void *other_stream, int ctx_waits) {
    int rc = check(c, true, true);      // (orders streams: changes neither state nor parameters)
    if (rc) return rc;
    hipStream_t other = (hipStream_t)other_stream;       // nullptr = the legacy default stream
    if (ctx_waits) {
        DOTS_HIP(hipEventRecord(c->ev[10], other));
        DOTS_HIP(hipStreamWaitEvent(c->stream, c->ev[10], 0));
    } else {
        DOTS_HIP(hipEventRecord(c->ev[11], c->stream));
        DOTS_HIP(hipStreamWaitEvent(other, c->ev[11], 0));
    }
    return 0;
}

int dots_step(dots_ctx *c, int n_iters, dots_step_stats *stats) {
    int rc = check(c, true, true);      // (run_iteration consumes the flags and a pending penalty division itself)
    if (rc) return rc;
    if (n_iters < 0) { set_error("n_iters < 0"); return DOTS_ERR_ARGUMENT; }
    if (c->shard_stride != 0) { set_error("dots_step on a time slab: use dots_slab_stage"); return DOTS_ERR_STATE; }
    dots_step_stats local;
    memset(&local, 0, sizeof local);
    for (int i = 0; i < n_iters; ++i)
        if ((rc = run_iteration(c, stats ? &local : nullptr))) return rc;
    if (stats) *stats = local;
    return 0;
}

int dots_step_flags(dots_ctx *c, uint32_t flags) {
    int rc = check(c, true, true);
    if (rc) return rc;
    if (flags & ~(uint32_t)(DOTS_STEP_SKIP_Z_MID | DOTS_STEP_PALM | DOTS_STEP_RHS_AHEAD | DOTS_STEP_TIMED | DOTS_STEP_CARRY | DOTS_STEP_KKT_SUMS)) { set_error("step_flags: unknown flag"); return DOTS_ERR_ARGUMENT; }
    if ((flags & DOTS_STEP_RHS_AHEAD) && (flags & DOTS_STEP_PALM)) { set_error("step_flags: DOTS_STEP_RHS_AHEAD cannot be combined with DOTS_STEP_PALM (its step 0 changes what the right-hand side reads)"); return DOTS_ERR_ARGUMENT; }
    c->rhs_ahead_armed = ((flags & DOTS_STEP_RHS_AHEAD) && rhs_writes_modes(c)) ? 1 : 0;      // (a hint: ignored without the direct solver / on a time slab)
    if ((flags & DOTS_STEP_SKIP_Z_MID) && (flags & DOTS_STEP_PALM)) { set_error("step_flags: DOTS_STEP_PALM reads z_mid, it cannot be combined with DOTS_STEP_SKIP_Z_MID"); return DOTS_ERR_ARGUMENT; }
    c->step_skip_zmid = (flags & DOTS_STEP_SKIP_Z_MID) ? 1 : 0;
    c->step_palm = (flags & DOTS_STEP_PALM) ? 1 : 0;
    c->step_timed = (flags & DOTS_STEP_TIMED) ? 1 : 0;
    c->step_carry = ((flags & DOTS_STEP_CARRY) && !(flags & DOTS_STEP_PALM)) ? 1 : 0;      // (a hint, like DOTS_STEP_RHS_AHEAD)
    c->step_kkt = ((flags & DOTS_STEP_KKT_SUMS) && !(flags & DOTS_STEP_SKIP_Z_MID)) ? 1 : 0;  // (a hint as well)
    return 0;
}

int dots_step_times(dots_ctx *c, dots_step_stats *out, int capacity, int wait, int *n_out) {
    int rc = check(c, true, true);
    if (rc) return rc;
    if (!out || !n_out || capacity < 0) { set_error("step_times: bad argument"); return DOTS_ERR_ARGUMENT; }
    int n = 0;
    while (n < capacity && c->t_count > 0) {
        hipEvent_t *tv = c->tev[c->t_head];
        const int kind = c->tkind[c->t_head];
        const bool whole = kind == 0 || kind == TKIND_FUSED;
        hipEvent_t last = tv[whole ? 5 : 1];
        if (wait) DOTS_HIP(hipEventSynchronize(last));
        else {
            const hipError_t q = hipEventQuery(last);
            if (q == hipErrorNotReady) break;
            DOTS_HIP(q);
        }
        dots_step_stats &st = out[n];
        memset(&st, 0, sizeof st);
        float t;
        if (whole) {
            // Every phase is bracketed by two events, and an event costs stream time itself: the gap between the two events
            // recorded back to back behind the last kernel (4, 5) is that cost, measured in this very iteration; it is taken off
            // every phase so that the sampled times estimate what the UNTIMED iterations of the kind take.
            float gap, p[4];
            DOTS_HIP(hipEventElapsedTime(&gap, tv[4], tv[5]));
            for (int i = 0; i < 4; ++i) {
                DOTS_HIP(hipEventElapsedTime(&p[i], tv[i], tv[i + 1]));
                p[i] = p[i] > gap ? p[i] - gap : 0.0f;
            }
            if (kind == TKIND_FUSED) {      // one launch for the right-hand side and the projection: half each, as a slab's stage 1
                st.ms_rhs = st.ms_soc = 0.5 * p[0];
            } else {
                st.ms_rhs = p[0];
                st.ms_soc = p[2];
            }
            st.ms_laplacian = p[1];
            st.ms_q_lambda_multiplier = p[3];
            st.ms_total = (double)p[0] + p[1] + (kind == TKIND_FUSED ? 0.0 : (double)p[2]) + p[3];
            st.alm_iterations = 1;
        } else {
            const int stage = kind - 1;
            DOTS_HIP(hipEventElapsedTime(&t, tv[0], tv[1]));
            st.ms_total = t;
            if (stage == 0 || stage == 6) st.ms_rhs = t;                // packing the halos; the right-hand side alone
            if (stage == 5) st.ms_soc = t;
            if (stage == 1) st.ms_rhs = st.ms_soc = 0.5 * t;            // one launch: right-hand side and projection together
            if (stage == 2) st.ms_laplacian = t;
            if (stage == 3) { st.ms_q_lambda_multiplier = t; st.alm_iterations = 1; }
        }
        c->t_head = (c->t_head + 1) % Ctx::TIME_SLOTS;
        c->t_count -= 1;
        ++n;
    }
    *n_out = n;
    return 0;
}

int dots_run_phase(dots_ctx *c, int phase, dots_step_stats *stats) {
    int rc = check(c);
    if (rc) return rc;
    if (c->shard_stride != 0) { set_error("run_phase works on whole arrays: not available on a time slab"); return DOTS_ERR_STATE; }
    dots_step_stats local;
    memset(&local, 0, sizeof local);
    if ((rc = materialise_zmid(c))) return rc;
    switch (phase) {
        case DOTS_PHASE_LAPLACIAN:
            if ((rc = launch_rhs(c))) return rc;
            if ((rc = cg_solve(c, &local))) return rc;
            break;
        case DOTS_PHASE_SOC_PROJECTION: rc = launch_soc_projection(c); c->zmid_stale = 0; break;
        case DOTS_PHASE_Q_LAMBDA_MULT: rc = launch_q_lambda_mult(c); break;
        case DOTS_PHASE_Q_LAMBDA:
            if (c->zmid_stale) { set_error("phase q_lambda: z_mid was not materialised by the last step (dots_step_flags)"); return DOTS_ERR_STATE; }
            rc = launch_q_lambda_only(c);
            break;
        default: set_error("unknown phase"); return DOTS_ERR_ARGUMENT;
    }
    if (rc) return rc;
    DOTS_HIP(hipStreamSynchronize(c->stream));
    if (stats) *stats = local;
    return 0;
}

int dots_kkt(dots_ctx *c, uint32_t mask, double *out) {
    int rc = check(c, true);
    if (rc) return rc;
    if (!out || (mask >> DOTS_N_KKT)) { set_error("kkt: bad mask or null output"); return DOTS_ERR_ARGUMENT; }
    if (!mask) return 0;
    if (c->zmid_stale && (mask & (1u << DOTS_KKT_PRIM_Z))) { set_error("kkt: z_mid was not materialised by the last step (dots_step_flags)"); return DOTS_ERR_STATE; }
    if (c->shard_stride != 0) { set_error("kkt on a time slab: use dots_kkt_sums / dots_kkt_combine around the caller's all-reduce"); return DOTS_ERR_STATE; }
    if ((mask & (1u << DOTS_KKT_PRIM_Z)) && !kkt_takes_fused(c, mask) && (rc = materialise_zmid(c))) return rc;      // (the stand-alone kernels read z_mid)
    return kkt_evaluate(c, mask, out);
}
int dots_kkt_sums(dots_ctx *c, uint32_t mask, double *sums) {
    int rc = check(c, true);
    if (rc) return rc;
    if (!sums || (mask >> DOTS_N_KKT)) { set_error("kkt_sums: bad mask or null output"); return DOTS_ERR_ARGUMENT; }
    static_assert(DOTS_KKT_N_SUMS == MAX_SUMS, "header and kernels disagree on the number of KKT sums");
    for (int i = 0; i < DOTS_KKT_N_SUMS; ++i) sums[i] = 0.0;
    if (!mask) return 0;
    if (c->zmid_stale && (mask & (1u << DOTS_KKT_PRIM_Z))) { set_error("kkt: z_mid was not materialised by the last step (dots_step_flags)"); return DOTS_ERR_STATE; }
    if (c->shard_stride != 0 && !c->kkt_halo_fresh && (mask & ((1u << DOTS_KKT_DUAL_ALPHA) | (1u << DOTS_KKT_COMP_RHO_FQ) | (1u << DOTS_KKT_COMP_M_RHO_B)))) {
        set_error("kkt_sums: the KKT halos are stale (dots_slab_stage 4 + exchange first)");
        return DOTS_ERR_STATE;
    }
    if ((mask & (1u << DOTS_KKT_PRIM_Z)) && !kkt_takes_fused(c, mask) && (rc = materialise_zmid(c))) return rc;
    return kkt_sums(c, mask, sums);
}
int dots_kkt_sums_device(dots_ctx *c, uint32_t mask, double *device_sums) {
    int rc = check(c, true);
    if (rc) return rc;
    if (!device_sums || (mask >> DOTS_N_KKT)) { set_error("kkt_sums_device: bad mask or null output"); return DOTS_ERR_ARGUMENT; }
    if (c->zmid_stale && (mask & (1u << DOTS_KKT_PRIM_Z))) { set_error("kkt: z_mid was not materialised by the last step (dots_step_flags)"); return DOTS_ERR_STATE; }
    if (c->shard_stride != 0 && !c->kkt_halo_fresh && (mask & ((1u << DOTS_KKT_DUAL_ALPHA) | (1u << DOTS_KKT_COMP_RHO_FQ) | (1u << DOTS_KKT_COMP_M_RHO_B)))) {
        set_error("kkt_sums_device: the KKT halos are stale (dots_slab_stage 4 + exchange first)");
        return DOTS_ERR_STATE;
    }
    return kkt_sums_device(c, mask, device_sums);
}
int dots_kkt_combine(dots_ctx *c, uint32_t mask, const double *sums, double *out) {
    int rc = check(c, true);
    if (rc) return rc;
    if (!sums || !out || (mask >> DOTS_N_KKT)) { set_error("kkt_combine: bad argument"); return DOTS_ERR_ARGUMENT; }
    return kkt_combine(c, mask, sums, out);
}
int dots_objective_sums(dots_ctx *c, double *sums) {
    int rc = check(c, true);
    if (rc) return rc;
    if (!sums) { set_error("null output"); return DOTS_ERR_ARGUMENT; }
    return objective_sums(c, sums);
}
int dots_objective_combine(dots_ctx *c, const double *sums, double *out) {
    int rc = check(c, true);
    if (rc) return rc;
    if (!sums || !out) { set_error("null argument"); return DOTS_ERR_ARGUMENT; }
    return objective_combine(c, sums, out);
}

int dots_objective(dots_ctx *c, double *out) {
    int rc = check(c, true);
    if (rc) return rc;
    if (!out) { set_error("null output"); return DOTS_ERR_ARGUMENT; }
    if (c->shard_stride != 0) { set_error("objective on a time slab: use dots_objective_sums / dots_objective_combine"); return DOTS_ERR_STATE; }
    return objective_evaluate(c, out);
}

int dots_adjust_penalty(dots_ctx *c, double factor) {
    const bool anticipated = c && c->rhs_ahead == 3 && factor == c->ahead_dv;      // (penalty_decision_ahead; check() drops the launch ahead)
    int rc = check(c);      // (carries out a division that is still pending)
    if (rc) return rc;
    if (!(factor > 0)) { set_error("factor must be positive"); return DOTS_ERR_ARGUMENT; }
    c->kkt_halo_fresh = 0;
    // one GPU, direct solver: the next iteration's kernels apply the division as they read (run_iteration); any other access to
    // the arrays carries it out first (check)
    if (c->lazy_div && c->shard_stride == 0 && carry_possible(c)) {
        c->pending_div = factor;
        if (anticipated) c->rhs_ahead = 4;      // ... kept if dots_set_params now brings the anticipated penalty
        return 0;
    }
    return launch_adjust_penalty(c, factor);
}
int dots_scale_z(dots_ctx *c, double z_mul, double beta_mul, double sz_new) {
    int rc = check(c);
    if (rc) return rc;
    c->kkt_halo_fresh = 0;
    if ((rc = materialise_zmid(c))) return rc;
    return launch_scale_z(c, z_mul, beta_mul, sz_new);
}
int dots_scale_arrays(dots_ctx *c, uint32_t mask, double factor) {
    int rc = check(c);
    if (rc) return rc;
    if (mask >> DOTS_N_ARRAYS) { set_error("bad array mask"); return DOTS_ERR_ARGUMENT; }
    c->kkt_halo_fresh = 0;
    if ((rc = materialise_zmid(c))) return rc;
    for (int id = 0; id < DOTS_N_ARRAYS; ++id)
        if ((mask >> id) & 1u)
            if ((rc = launch_scale_array(c, id, factor))) return rc;
    return 0;
}
int dots_norm_square(dots_ctx *c, int id, int part, double *out) {
    int rc = check(c);
    if (rc) return rc;
    if (id < 0 || id >= DOTS_N_ARRAYS || !out || part < 0 || part > 2) { set_error("norm_square: bad argument"); return DOTS_ERR_ARGUMENT; }
    if (id == DOTS_Z_MID && c->zmid_stale) { set_error("norm_square: z_mid was not materialised by the last step (dots_step_flags)"); return DOTS_ERR_STATE; }
    if (id == DOTS_Z_MID && (rc = materialise_zmid(c))) return rc;
    return norm_square(c, id, part, out);
}

int dots_apply_operator(dots_ctx *c, int op, double scale, const double *in, int64_t n_in, double *out, int64_t n_out) {
    int rc = check(c);
    if (rc) return rc;
    // (input array class, output array class) per operator, expressed through state arrays of the same shape
    int in_id, out_id;
    switch (op) {
        case DOTS_OP_GRAD_TIME: in_id = DOTS_PHI; out_id = DOTS_A; break;
        case DOTS_OP_DIV_TIME: case DOTS_OP_TIME_AVG_ADJOINT: in_id = DOTS_A; out_id = DOTS_PHI; break;
        case DOTS_OP_GRAD_SPACE: in_id = DOTS_PHI; out_id = DOTS_B; break;
        case DOTS_OP_DIV_SPACE: in_id = DOTS_B; out_id = DOTS_PHI; break;
        case DOTS_OP_DECOUPLE: in_id = DOTS_B; out_id = DOTS_Z_MID; break;
        case DOTS_OP_DECOUPLE_ADJOINT: in_id = DOTS_Z_MID; out_id = DOTS_B; break;
        case DOTS_OP_LAPLACIAN_APPLY: in_id = DOTS_PHI; out_id = DOTS_PHI; break;
        default: set_error("unknown operator"); return DOTS_ERR_ARGUMENT;
    }
    if (!in || !out || n_in != array_count_host(c->d, in_id) || n_out != array_count_host(c->d, out_id)) {
        set_error("apply_operator: bad element counts");
        return DOTS_ERR_ARGUMENT;
    }
    // scratch: device-layout input and output live in two temporary buffers
    double *din = nullptr, *dout = nullptr;
    const int64_t cin = array_count_device(c->d, in_id), cout = array_count_device(c->d, out_id);
    DOTS_HIP(hipMalloc((void **)&din, sizeof(double) * (size_t)cin));
    DOTS_HIP(hipMalloc((void **)&dout, sizeof(double) * (size_t)cout));
    DOTS_HIP(hipMemsetAsync(dout, 0, sizeof(double) * (size_t)cout, c->stream));
    DOTS_HIP(hipMemcpyAsync(c->stage, in, sizeof(double) * (size_t)n_in, hipMemcpyHostToDevice, c->stream));
    // reuse the layout kernels by temporarily pointing the state slot at the scratch buffers
    Ctx tmp = *c;
    double **slot_in[12] = {&tmp.d.phi, &tmp.d.A, &tmp.d.B, &tmp.d.lam, &tmp.d.zf, &tmp.d.zm, &tmp.d.ze, &tmp.d.mu, &tmp.d.E, &tmp.d.bf, &tmp.d.bm, &tmp.d.be};
    *slot_in[in_id] = din;
    rc = launch_to_device_layout(&tmp, in_id, c->stage);
    if (!rc) {
        if (op == DOTS_OP_LAPLACIAN_APPLY) rc = cg_apply_operator(c, din, dout);
        else rc = launch_operator(c, op, scale, din, dout);
    }
    if (!rc) {
        Ctx tmp2 = *c;
        double **slot_out[12] = {&tmp2.d.phi, &tmp2.d.A, &tmp2.d.B, &tmp2.d.lam, &tmp2.d.zf, &tmp2.d.zm, &tmp2.d.ze, &tmp2.d.mu, &tmp2.d.E, &tmp2.d.bf, &tmp2.d.bm, &tmp2.d.be};
        *slot_out[out_id] = dout;
        rc = launch_from_device_layout(&tmp2, out_id, c->stage);
    }
    hipError_t e1 = hipMemcpyAsync(out, c->stage, sizeof(double) * (size_t)n_out, hipMemcpyDeviceToHost, c->stream);
    hipError_t e2 = hipStreamSynchronize(c->stream);
    (void)hipFree(din);
    (void)hipFree(dout);
    if (rc) return rc;
    DOTS_HIP(e1);
    DOTS_HIP(e2);
    return 0;
}

int dots_mg_setup(dots_ctx *c, const dots_mg_desc *m) {
    int rc = check(c);
    if (rc) return rc;
    if (!m || m->n_levels < 2 || m->n_levels > 10 || !m->levels || !m->coarse_inverse) { set_error("mg_setup: bad description"); return DOTS_ERR_ARGUMENT; }
    if (c->lap_solver != DOTS_LAP_MODAL_PCG) { set_error("multigrid needs the modal solver"); return DOTS_ERR_ARGUMENT; }
    if (m->levels[0].n != c->d.V || m->n_cols != c->dcg.cg_ncol) { set_error("mg_setup: level 0 / mode count mismatch"); return DOTS_ERR_ARGUMENT; }
    DOTS_HIP(hipStreamSynchronize(c->stream));
    if (c->cg_graph) { (void)hipGraphExecDestroy(c->cg_graph); c->cg_graph = nullptr; }
    mg_release(c);
    const Dev &d = c->dcg;   // level vectors and the coarse inverse use the PCG view's pitch
    MgDev g{};
    g.nlev = m->n_levels;
    g.omega = m->omega;
    for (int l = 0; l < m->n_levels; ++l) {
        const dots_mg_level &h = m->levels[l];
        MgLevelDev &L = g.lv[l];
        L.n = h.n;
        L.nc = h.n_coarse;
        if (l + 1 < m->n_levels && (h.n_coarse != m->levels[l + 1].n || !h.p_rowptr || !h.r_rowptr)) { set_error("mg_setup: inconsistent level sizes"); mg_release(c); return DOTS_ERR_ARGUMENT; }
        // index sanity (a wrong index would fault on the device)
        if (l > 0) {
            if (!h.rowptr || !h.col || !h.val_k || !h.val_m || !h.diag_k || !h.diag_m) { set_error("mg_setup: null level array"); mg_release(c); return DOTS_ERR_ARGUMENT; }
            for (int j = 0; j < h.nnz; ++j) if (h.col[j] < 0 || h.col[j] >= h.n) { set_error("mg_setup: column index out of range"); mg_release(c); return DOTS_ERR_ARGUMENT; }
            if (h.rowptr[h.n] != h.nnz) { set_error("mg_setup: rowptr/nnz mismatch"); mg_release(c); return DOTS_ERR_ARGUMENT; }
        }
        if (l + 1 < m->n_levels) {
            if (h.p_rowptr[h.n] != h.p_nnz || h.r_rowptr[h.n_coarse] != h.p_nnz) { set_error("mg_setup: P/R size mismatch"); mg_release(c); return DOTS_ERR_ARGUMENT; }
            for (int j = 0; j < h.p_nnz; ++j)
                if (h.p_col[j] < 0 || h.p_col[j] >= h.n_coarse || h.r_col[j] < 0 || h.r_col[j] >= h.n) { set_error("mg_setup: P/R index out of range"); mg_release(c); return DOTS_ERR_ARGUMENT; }
        }
#define MUP(field, src, n) if ((rc = mg_upload(c, &L.field, src, (int64_t)(n)))) { mg_release(c); return rc; }
        if (l == 0) {
            L.rp = d.rowptr; L.col = d.col; L.vK = d.val; L.vM = nullptr; L.dK = d.kdiag; L.dM = d.mass_v;
        } else {
            MUP(rp, h.rowptr, h.n + 1); MUP(col, h.col, h.nnz); MUP(vK, h.val_k, h.nnz); MUP(vM, h.val_m, h.nnz);
            MUP(dK, h.diag_k, h.n); MUP(dM, h.diag_m, h.n);
        }
        if (l + 1 < m->n_levels) {
            MUP(p_rp, h.p_rowptr, h.n + 1); MUP(p_col, h.p_col, h.p_nnz); MUP(p_val, h.p_val, h.p_nnz);
            MUP(r_rp, h.r_rowptr, h.n_coarse + 1); MUP(r_col, h.r_col, h.p_nnz); MUP(r_val, h.r_val, h.p_nnz);
            if (!h.ap_rowptr || !h.ap_col || !h.ap_val_k || !h.ap_val_m || !h.ap_val_p || h.ap_rowptr[h.n] != h.ap_nnz) { set_error("mg_setup: bad A*P arrays"); mg_release(c); return DOTS_ERR_ARGUMENT; }
            for (int j = 0; j < h.ap_nnz; ++j) if (h.ap_col[j] < 0 || h.ap_col[j] >= h.n_coarse) { set_error("mg_setup: A*P index out of range"); mg_release(c); return DOTS_ERR_ARGUMENT; }
            MUP(ap_rp, h.ap_rowptr, h.n + 1); MUP(ap_col, h.ap_col, h.ap_nnz); MUP(ap_vK, h.ap_val_k, h.ap_nnz); MUP(ap_vM, h.ap_val_m, h.ap_nnz); MUP(ap_vP, h.ap_val_p, h.ap_nnz);
        }
#undef MUP
        // level vectors; level 0 works on the PCG's own r, z and Ap buffers
        if (l > 0) {
            const int64_t nv = (int64_t)h.n << d.tp_shift;
            const double *tmp = nullptr;
            double **vecs[3] = {&L.b, &L.bt, &L.t};
            for (auto v : vecs) {
                if ((rc = mg_upload<double>(c, &tmp, nullptr, nv))) { mg_release(c); return rc; }
                *v = const_cast<double *>(tmp);
            }
        }
    }
    // coarse inverse: host [nL][nL][n_cols] -> device [nL][nL][TP]
    const int nL = m->levels[m->n_levels - 1].n;
    std::vector<double> inv((size_t)nL * nL * d.TP, 0.0);
    for (int64_t ij = 0; ij < (int64_t)nL * nL; ++ij)
        for (int k = 0; k < m->n_cols; ++k) inv[(size_t)ij * d.TP + k] = m->coarse_inverse[(size_t)ij * m->n_cols + k];
    if ((rc = mg_upload(c, &g.coarse_inv, inv.data(), (int64_t)inv.size()))) { mg_release(c); return rc; }
    // mg_release() reset c->mg; the level pointers were written into the local copy
    for (int l = 0; l < g.nlev; ++l) c->mg.lv[l] = g.lv[l];
    c->mg.nlev = g.nlev;
    c->mg.omega = g.omega;
    c->mg.coarse_inv = g.coarse_inv;
    return 0;
}

int dots_mg_enable(dots_ctx *c, int on) {
    int rc = check(c);
    if (rc) return rc;
    c->use_mg = on ? 1 : 0;
    return 0;
}

int dots_front_setup(dots_ctx *c, const dots_front_desc *desc) {
    int rc = check(c);
    if (rc) return rc;
    if (c->lap_solver != DOTS_LAP_MODAL_PCG) { set_error("the direct solve needs the modal solver"); return DOTS_ERR_ARGUMENT; }
    c->d.cn_sq = c->d.cn_g = c->d.cn_lo = c->d.cn_e = nullptr;
    if ((rc = front_setup(c, desc))) return rc;
    // DOTS_STEP_CARRY: the per-corner gathers steps 2+3 leave for the next right-hand side / projection (one GPU, pitch <= 128);
    // they belong to the direct solver's iteration and are released with the factor
    {   // beta_mid streamed around the caches (ql2_lane<BMNT>): where factor + the state an iteration touches do not fit the Infinity Cache but the
        // factor is small enough for a good part of it to stay there between the sweeps once beta_mid no longer pushes it out.  Measured, hint against
        // none (profiles/studies/r04_nontemporal.txt): knot -2.5 % (everything fits), knot63 / sphere10k / a 10k torus +9 %, tori of 20k / 40k / 60k vertices
        // +3.3 / +3.7 / +5.0 %, 80k +0.3 %, 100k -0.5 %; T = 63: 20k +2.3 %, 40k +0.1...1.5 %; T = 127: 20k -0.2...+1.7 %, 65k -2 %
        // (the sweeps of the last four touch 0.92-3.1 GB): on up to 0.9 GB
        const double mall = 256.0 * 1048576.0, F8 = 8.0 * (double)c->d.F * (double)c->d.TP, V8 = 8.0 * (double)c->d.V * (double)c->d.TP;
        const double touched = 33.0 * F8 + 12.0 * V8;      // beta_mid, B, E, the carried sums; the vertex arrays
        int nt = -1;
        if (!env_int("DOTS_BM_NT", 0, 1, &nt)) { front_release(c); return DOTS_ERR_ARGUMENT; }
        const double factor = 0.5 * c->front_bytes;      // what the sweeps touch: every block is read by both sweeps (the zero blocks of merged nodes are never read)
        c->bm_nt = nt >= 0 ? nt : (factor < 0.9e9 && factor + touched > mall ? 1 : 0);
    }
    if (c->d.TP <= 128 && c->carry_arrays && c->d.nl > 0) {      // (one GPU or a time slab with nodes)
        const int64_t rows = (int64_t)3 * c->d.F;
        const double *sq = nullptr, *g = nullptr, *lo = nullptr, *e = nullptr;
        if ((rc = front_upload<double>(c, &sq, nullptr, (2 * rows) << c->d.tp_shift)) || (rc = front_upload<double>(c, &g, nullptr, rows << c->d.tp_shift)) ||
            (rc = front_upload<double>(c, &lo, nullptr, rows)) || (c->shard_stride == 0 && (rc = front_upload<double>(c, &e, nullptr, rows << c->d.tp_shift)))) {
            front_release(c);
            return rc;
        }
        c->d.cn_sq = const_cast<double *>(sq);
        c->d.cn_g = const_cast<double *>(g);
        c->d.cn_lo = const_cast<double *>(lo);
        c->d.cn_e = const_cast<double *>(e);
    }
    return 0;
}

int dots_front_enable(dots_ctx *c, int on) {
    int rc = check(c);
    if (rc) return rc;
    if (on && c->front.n_nodes == 0) { set_error("front_enable: no factor installed"); return DOTS_ERR_STATE; }
    c->use_front = on ? 1 : 0;
    return 0;
}

int dots_front_launches(dots_ctx *c) {
    if (check(c) || c->front.n_nodes == 0) return -1;
    return 2 * c->front.n_levels - (c->front_top_inverse ? 1 : 0);
}

int dots_front_info(dots_ctx *c, double *out) {
    int rc = check(c);
    if (rc) return rc;
    if (!out || c->front.n_nodes == 0) { set_error("front_info: no factor installed"); return DOTS_ERR_STATE; }
    out[0] = c->front_bytes_unmerged;
    out[1] = c->front_bytes;
    out[2] = (double)c->front_heights;
    out[3] = (double)c->front.n_levels;
    return 0;
}

int dots_front_pitch(dots_ctx *c) {
    if (check(c)) return -1;
    return c->dcg.TP;
}

int64_t dots_debug_counter(dots_ctx *c, int which) {
    if (!c) return -1;
    switch (which) {
        case 0: return c->mail_fallbacks;
        case 1: return (int64_t)c->mail_seq;
        case 2: return c->penalty_ahead_started;
        case 3: return c->penalty_ahead_confirmed;
        case 4: return c->front.n_leaves;      // leaves the sweeps handle as explicit local inverses (0: band kernels)
        case 5: return c->front.leaf_bd ? 1 : 0;      // ... with their coupling in per-row records (0: read from the CSR)
        case 6: return c->bm_nt;                      // beta_mid streamed around the caches by steps 2+3 (the rule of dots_front_setup, or DOTS_BM_NT)
        default: return -1;
    }
}

int dots_bench_kernel(dots_ctx *c, int which, int reps, double *ms, double *bytes) {
    int rc = check(c);
    if (rc) return rc;
    if (reps < 1 || !ms || !bytes) { set_error("bench: bad argument"); return DOTS_ERR_ARGUMENT; }
    return cg_bench(c, which, reps, ms, bytes);
}

}  // extern "C"
