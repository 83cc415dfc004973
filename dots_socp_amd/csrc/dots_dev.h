// Internal declarations shared by the HIP translation units of libdotsocp_hip.so (gfx950 only).
//
// Device data layout ("time-fastest"): every spatial gather of the path -- a vertex reading its
// neighbours / corners, a triangle reading its three vertices -- touches one contiguous row of
// TP doubles (TP = time pitch, power of two >= T+1), so all HBM reads coalesce:
//     node / interval arrays   x[v][t]                 idxV(v,t)       = v*TP + t
//     triangle arrays          B[f][c][t]              idxF(f,c,t)     = (f*3+c)*TP + t
//     corner arrays            z_mid[f][k][s][c][t]    idxM(f,k,s,c,t) = ((((f*3+k)*2+s)*3+c)*TP + t
// Interval arrays use t in [0,T), node arrays t in [0,T]; padding columns stay zero.
// Corner entries live in the column of the NODE they are compared with (z_mid[t][s] pairs with B[t + s],
// solver_socp.py:934-940): column t + s.  The thread that owns B[f][c][node] then owns every corner entry of
// its column, which is what lets the time axis be cut into slabs without any exchange inside steps 2+3.
//
// Time slabs (multi-GPU, one context per rank): a context holds the columns of the nodes [t0, t0 + nl) only
// (local column j = node t0 + j = interval t0 + j); T stays the GLOBAL number of intervals.  What a kernel needs
// from the neighbouring slabs arrives in small per-vertex / per-triangle halo arrays (X_lo ... B_hi below).
// On one GPU t0 = 0, nl = T + 1, ni = T and no halo is ever read.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/dots_socp_hip.h"

namespace dots {

constexpr int BLOCK = 256;           // threads per workgroup = 4 wave64
constexpr int TILE_ELEMS = 1024;     // (vertex,time) elements per workgroup tile
constexpr int LDS_NNZ_CAP = 1536;    // CSR entries of one row block staged in LDS (18 KB)
constexpr int MAX_SUMS = 24;         // reduction slots per kernel
constexpr double INV_SQRT3 = 0.57735026918962576451;

void set_error(const std::string &msg);
int hip_fail(hipError_t e, const char *what, const char *file, int line);

#define DOTS_HIP(call)                                                        \
    do {                                                                      \
        hipError_t e__ = (call);                                              \
        if (e__ != hipSuccess) return ::dots::hip_fail(e__, #call, __FILE__, __LINE__); \
    } while (0)

// Everything a kernel needs, passed by value.
struct Dev {
    int T, V, F, TP, tp_shift;
    int t0;          // global index of the node in local column 0
    int nl;          // nodes held here (columns [0, nl));  T + 1 on one GPU
    int ni;          // intervals held here = min(nl, T - t0);  T on one GPU
    int slab;        // 1: this context is one time slab of a multi-GPU solve
    int VT;          // vertices per tile = TILE_ELEMS / TP
    int n_vtiles;    // ceil(V / VT)
    int FT;          // (triangle,component) rows per tile = TILE_ELEMS / TP
    int n_ftiles;    // ceil(3F / FT)
    double h;        // 1 / T
    // mesh constants
    const int *tri;          // [F][3]
    const double *hat;       // [F][3][3]
    const double *area_f;    // [F]
    const double *mass_v;    // [V]
    const int *cptr;         // [V+1]
    const int *cidx;         // [3F]  f*3+k
    const double *c_D;       // [3F]  sqrt(area_f / mass_v) per corner-list entry
    const double *c_gA;      // [3F][3] hat gradient * area_f per corner-list entry
    const double *c_area;    // [3F]  area_f per corner-list entry
    const double *fk_D;      // [3F]  the same sqrt(area_f / mass_v), indexed by corner f*3+k
    const int *rowptr;       // [V+1]
    const int *col;          // [nnz]
    const double *val;       // [nnz]
    const double *kdiag;     // [V] diagonal of K
    const double *mu0, *mu1; // [V]
    const int *perm_v, *perm_f;
    const int *tiles_vertex, *tiles_tri_ptr, *tiles_tri, *tiles_c_loc;   // owned arrays behind Ctx::tiles
    const double *Q;         // [(T+1)^2] or null
    const double *Qpad, *QpadT;  // [TP][TP] Q and its transpose, zero padded (operands of the MFMA transform)
    const double *sigma;     // [T+1] or null
    // state
    double *phi, *A, *B, *lam, *zf, *zm, *ze, *mu, *E, *bf, *bm, *be;
    double *B_st, *bm_st;    // where steps 2+3 (k_q_lambda_mult_carry) STORE the new B and beta_mid: the arrays themselves, or -- on an iteration
                             // whose z_mid may be asked for -- the alternate B buffer and z_mid's storage (Ctx::zmid_deferred)
    double *lamc;            // [V][TP] cone multiplier of the last projection (z_mid = lamc/D * pre-image, rebuilt in steps 2+3)
    // DOTS_STEP_CARRY (one GPU, direct solver, pitch <= 128; null otherwise): what the right-hand side and the cone projection of the
    // NEXT iteration gather per corner, stored by steps 2+3 in corner-LIST order (row j = position in cidx; cpos: f*3+k -> j)
    double *cn_sq;           // [3F][2][TP] halves s = 0, 1 of sum_xyz (D (sz/sqrt3 B - beta_mid))^2, column = interval
    double *cn_g;            // [3F][TP]    sum_xyz area * hat * (B - E), column = node
    double *cn_e;            // [3F][TP]    sum_xyz area * hat * E, column = node: the gather of Dual(alpha) (read-back iterations only)
    double *cn_lo;           // [3F] time slab: the s = 1 half of interval t0 - 1 (formed at this slab's first node; summed into send_nsq)
    const int *cpos;         // [3F]
    // halos of a time slab (device arrays, written by dots_slab_unpack / the inverse transform; see SlabHalo in dots_api.hip)
    const double *X_lo;      // [V]  (A + lambda_c - mu) of interval t0 - 1           (right-hand side at node t0)
    const double *lamc_lo;   // [V]  cone multiplier of interval t0 - 1                (steps 2+3 at node t0)
    const double *mu_lo;     // [V]  mu of interval t0 - 1                             (KKT: Dual(alpha), Comp(m, rho o B))
    const double *nsq_hi;    // [V]  s = 1 half of the cone's squared norm for interval t0 + nl - 1, formed by the next slab
    double *phi_hi;          // [V]  phi at node t0 + nl (written by this rank's inverse transform)
    const double *B_hi;      // [3F] B at node t0 + nl                                  (KKT: Comp(rho, f(q)))
    // PCG workspace (node layout)
    double *cg_b, *cg_r, *cg_z, *cg_p0, *cg_p1, *cg_Ap, *cg_x;
    double *partials;        // [MAX_SUMS or NC][n partial blocks]
    double *scal;            // device scalars, see CgScal / sums
    int *flags;
    int cg_ncol;             // columns the PCG works on (T+1 on one GPU; the rank's modes when sharded)
};

// One level of the multigrid hierarchy (device pointers).  A_l = K_l + s M_l on one pattern;
// level 0 has vM == nullptr (M is the diagonal mass, held in dM).
struct MgLevelDev {
    int n, nc;                               // rows of this level / of the next coarser one
    const int *rp, *col;
    const double *vK, *vM, *dK, *dM;
    const int *p_rp, *p_col;                 // P: n x nc
    const double *p_val;
    const int *r_rp, *r_col;                 // R = P^T: nc x n
    const double *r_val;
    const int *ap_rp, *ap_col;               // A P = K P + s M P: n x nc, two value arrays on one pattern
    const double *ap_vK, *ap_vM, *ap_vP;     // ap_vP: P itself on the pattern of A P (zeros where absent)
    double *b, *bt, *t;                      // level vectors [n][TP]: rhs, D^-1 rhs / result, residual (level 0 borrows the PCG's r, z, Ap)
};
struct MgDev {
    int nlev = 0;
    MgLevelDev lv[10]{};
    const double *coarse_inv = nullptr;      // [nL][nL][TP]
    double omega = 2.0 / 3.0;
};

// Direct (multifrontal) solve of the modal problems: the factor of dots_socp_amd/frontal.py on the device.
// F_p = [L_pp^-1 ; A_bs A_ss^-1] of node p starts at F + (foff[p] << tp_shift), entry (i, j, mode) at
// ((i * n_p + j) << tp_shift) + mode: the mode index is fastest, as in every node array.
struct FrontNode {            // a node of the elimination tree as the numeric factorisation sees it (kernels_factor.hip)
    int n, b;                 // separator rows eliminated here / boundary rows
    int k0;                   // elimination index of the first separator row
    int parent;               // parent node (-1: root)
    int64_t foff;             // first (row, col) entry of F_p
    int64_t bdoff;            // first boundary row of this node in bd_vertex
    int c0, c1;               // children (-1: none)
    int64_t ioff;             // first front row in the pull maps
    int64_t soff;             // first entry of the b x b Schur complement
};
// One node of the SWEEPS (kernels_front.hip): an original node, or the nodes of a band of tree heights that hang together
// merged into one block F' = [L'^-1 ; G'] (row stride n; L'^-1 block lower triangular over the members, children first).
struct SweepNode {
    int n, b;                 // columns = separator rows of all members / boundary rows (those of the band's top node)
    int k0;                   // sweep-order index of the first separator row
    int planes;               // update planes of this node that a child writes (the launch reads the band's bucket: 0, 2, 4 or 8)
    int64_t foff;             // first (row, col) entry of F'
    int64_t woff;             // first row of this node's update planes in W, m rows each
    int64_t parent_w;         // first row of the plane this node writes in its parent's W (-1: root)
    int64_t bdoff;            // first boundary row of this node in bd_vertex / cmap
};
// One workgroup of a sweep: the node's record and its block of rows (columns), in ONE 96-byte record so that the
// kernel's dependent-load chain starts with a single (scalar) load instead of descriptor -> node record.
struct FrontWork {
    SweepNode nd;
    int first;                // first row (forward) / first column (backward) of the block
    int lo;                   // forward: first column that can be nonzero in the block's rows; backward: number of row ranges
    int rs[4], re[4];         // backward: separator-row ranges [rs, re) that can be nonzero in the block's columns (rs[0] = first)
    int end;                  // backward: one past the last column of the block's member; forward: > 0 = every row of the block runs over
                              // the columns [lo, end) (a node that stores an explicit inverse), 0 = up to the diagonal
    int pad;
};
// One leaf of the tree when the leaves' band is stored as explicit local inverses (kernels_front.hip: k_front_leaf_fwd / _bwd)
struct LeafWork {
    int k0, n, b, pad;        // first sweep-order index (= device vertex: the leaf is numbered in place), separator / boundary rows
    int64_t soff;             // first entry of S = A_ss^-1 (lower triangle packed by rows) in FrontDev::leafS
    int64_t bdoff;            // first boundary row in bd_vertex / cmap
    int64_t parent_w;         // first row of the plane this leaf writes in its parent's W
    int64_t rowoff;           // first record of its boundary rows in FrontDev::leaf_bd
};
// The leaf's coupling to its boundary, copied once from the CSR of K into one record per row (the sweeps then reach it with ONE load
// behind the leaf's record instead of vertex -> row pointer -> entries): entries in CSR order, so the sums are the CSR path's, bit for bit
constexpr int LEAF_KC = 6, LEAF_KE = 8;
struct LeafBdRow {            // boundary row: where it goes in the parent's plane, its entries inside the leaf (position in the leaf, value; padded with zeros)
    int cm, cnt;
    int u[LEAF_KC];
    double v[LEAF_KC];
};
struct LeafSepRow {           // separator row (indexed by vertex): its entries OUTSIDE the leaf (vertex, value)
    int cnt, pad;
    int u[LEAF_KE];
    double v[LEAF_KE];
};
// One original node inside a merged band node (k_merge_member builds F' from the members' own blocks)
struct MergeMember {
    int n, b;                 // its separator / boundary rows
    int o, c0;                // first own column / first column of its subtree inside the merged node
    int64_t foff;             // its block [L^-1 ; G] in the factor (row stride n)
    int64_t ioff;             // first front row in the pull maps
    int64_t dst;              // first entry of the merged node's F'
    int64_t uoff;             // first entry of U_s (the update operator of its subtree on its boundary; columns from c0)
    int ns;                   // row stride of F' = columns of the merged node
    int ustride;              // row stride of U_s
    int uin;                  // U_s lives in: 0 the factor (its own G rows: no child inside the band), 1 scratch, 2 F' (the band's top node)
    int ch[2];                // its children inside the band by pull map (index of the member record, -1: none or below the band)
    int pad;
};
// Vertex tiles of the right-hand-side / projection launch cut from dots_problem_desc.patch_order (compact patches of the surface)
// with the distinct triangles each tile touches (k_rhs_soc_tiles stages their rows of B, E in LDS).
struct TileDev {
    int n_tiles = 0, VTL = 0, ntri_max = 0;
    const int *vertex = nullptr;     // [n_tiles][VTL] device vertices, -1 = padding
    const int *tri_ptr = nullptr;    // [n_tiles + 1]
    const int *tri = nullptr;        // the distinct triangles of every tile
    const int *c_loc = nullptr;      // [3F] per corner-list entry: position of its triangle in its vertex's tile
};

struct FrontDev {
    int n_nodes = 0, n_levels = 0;        // original nodes; launches per sweep (= bands of tree heights)
    const FrontNode *nodes = nullptr;     // original tree (factorisation)
    const int *vmap = nullptr;            // sweep-order index -> device vertex; nullptr when the device numbering is the sweep order
    const int *bd_vertex = nullptr;       // device vertex of every boundary row
    const int *cmap = nullptr;            // position of every boundary row in the parent's front
    const double *F = nullptr;            // original blocks, then the merged ones
    double *W = nullptr;                  // update planes [rows][TP]; entries no child writes stay zero
    const FrontWork *fwd_desc = nullptr, *bwd_desc = nullptr;             // per workgroup: node record + its block
    const double *leafS = nullptr;        // leaves as explicit inverses: S_p = A_ss^-1 of every leaf, one after the other (n_leaves > 0)
    const LeafWork *leaf_desc = nullptr;
    const LeafBdRow *leaf_bd = nullptr;   // coupling records (nullptr: the leaf kernels walk the CSR)
    const LeafSepRow *leaf_sep = nullptr;
    int n_leaves = 0, leaf_nmax = 0;      // leaves handled by the leaf kernels (0: the band kernels take band 0), their largest n
};

// Layout of the device scalar block used by the PCG (all arrays have NC entries, NC <= 256).
struct CgScalOffsets {
    static constexpr int NCMAX = 256;
    static constexpr int RZ = 0;
    static constexpr int PAP = NCMAX;
    static constexpr int ALPHA = 2 * NCMAX;
    static constexpr int BETA = 3 * NCMAX;
    static constexpr int BREF = 4 * NCMAX;
    static constexpr int BMEAN = 5 * NCMAX;
    static constexpr int SUMS = 6 * NCMAX;     // MAX_SUMS reduction results for KKT / norms
    static constexpr int TOTAL = 6 * NCMAX + 64;
};
// flags: [0..NCMAX) done per column, [NCMAX] iteration counter
constexpr int FLAG_ITERS = CgScalOffsets::NCMAX;
constexpr int FLAG_TOTAL = CgScalOffsets::NCMAX + 8;

// Weighted sums the KKT residuals are combined from (kernels_kkt.hip; solver_socp.py:433-559): vertex sums, then triangle sums
enum {
    V_DPHI2 = 0, V_A2, V_LAM2, V_RESMU2, V_RFST2, V_REND2, V_MU2, V_AUX1_2, V_MUAUX1_2,
    V_COMP_AUX2, V_COMP_RES2, V_CONG_RES2, V_DUALAUX2, N_VSUMS,
    F_DX2 = N_VSUMS, F_B2, F_RESE2, F_E2, F_AUX2_2, F_EAUX2_2, F_AUX5_2, F_AUX5M_2, F_RMID2, N_SUMS
};
constexpr int N_FSUMS = N_SUMS - N_VSUMS;
static_assert(N_SUMS <= MAX_SUMS, "too many reduction slots");
struct KktArgs {
    uint32_t mask;
    double r, sz, cd, cong, ps, ds, bs;   // penalty, scale_factor_z, constant_d, congestion, prim/dual/boundary scale
};
// DOTS_STEP_KKT_SUMS: the sums steps 2+3 can form from their registers while they write the new iterate -- everything the
// conditions Prim(phi, q), Prim(q, z), Dual(beta) and Comp(rho, cong.) need (no gather across rows): per workgroup partial sums
struct KktFused {
    double *part_v = nullptr, *part_f = nullptr;   // [N_VSUMS][nv], [N_FSUMS][nf] (only the fused slots are written)
    int nv = 0, nf = 0;                            // workgroups of the vertex / triangle part of the launch that wrote them
};
constexpr uint32_t KKT_FUSED_MASK = 1u | 2u | 8u | 64u;

struct Ctx;
bool rhs_on_tiles(const Ctx *c);    // the right-hand-side / projection launch runs on patch tiles (k_rhs_soc_tiles)

// ---- launch wrappers implemented in the kernel files (all asynchronous on ctx stream) ----
int launch_soc_projection(Ctx *c, int zmid_mode = 0, bool with_inverse = false);   // 0: write z_mid; 1: write only the cone multiplier; with_inverse: extra workgroups do the modes -> time transform of phi
void preload_alm_kernels();
void preload_kkt_kernels();
void preload_transform_kernels();
int launch_rhs(Ctx *c, bool with_soc = false, double dv = 0.0);   // with_soc (only when rhs_takes_soc): the cone projection rides in the same launch; dv != 0 (only when rhs_divides): a pending penalty division is applied to what is read
bool rhs_divides(const Ctx *c);
bool ql_divides(const Ctx *c, int zmid_mode);
int launch_q_lambda_mult(Ctx *c, int zmid_mode = 0, double dv = 0.0);    // 0: read z_mid; 1: rebuild it from the multiplier and store it; 2: rebuild, do not store; dv != 0 (only when ql_divides): the dual arrays are divided as they are read and written back divided
int materialise_zmid(Ctx *c);                            // z_mid rebuilt from the old B / beta_mid a deferred step kept (no-op otherwise)
int launch_q_lambda_only(Ctx *c);                       // the (q, lambda_c) closed form alone (is_palm's step 0): z_mid is read from memory
int launch_adjust_penalty(Ctx *c, double factor);
int launch_scale_z(Ctx *c, double z_mul, double beta_mul, double sz_new);
int launch_scale_array(Ctx *c, int array_id, double factor);
int launch_to_device_layout(Ctx *c, int array_id, const double *staged);
int launch_from_device_layout(Ctx *c, int array_id, double *staged);
int launch_operator(Ctx *c, int op, double scale, const double *in_staged, double *out_staged);
int launch_calibration(Ctx *c, double *bytes_each_way);
int cg_solve(Ctx *c, dots_step_stats *stats, bool defer_inverse = false);   // defer_inverse: phi is produced by the caller (soc_takes_inverse)
int cg_apply_operator(Ctx *c, const double *x_node, double *y_node);  // y = K x in node layout
int cg_finish_sharded(Ctx *c);                                        // phi of this time slab from all ranks' mode-space solutions (slab.x_recv)
int launch_slab_pack_iteration(Ctx *c);     // halos the neighbours need before the right-hand side / the projection -> slab.send_x, slab.send_nsq
int launch_slab_pack_kkt(Ctx *c);           // halos the neighbours' KKT kernels need -> slab.send_mu, slab.send_b
int cg_bench(Ctx *c, int which, int reps, double *ms, double *bytes);
int64_t cg_partials_needed(const Dev &d);   // doubles of Dev::partials the PCG uses
int front_setup(Ctx *c, const dots_front_desc *desc);
void front_release(Ctx *c);
int front_factorize(Ctx *c, const dots_front_desc *desc, FrontDev &f, const std::vector<FrontNode> &nodes, double *T, const int *grounded_host);
int front_solve(Ctx *c, const double *bhat, double *y, double *x);   // x = A^-1 bhat for every mode of the PCG view (y: scratch)
int mg_vcycle(Ctx *c, const double *r, double *z, double *t0, double *rz_part, int nb, int ept, int vt, int G);  // enqueue z = MG(r); z holds D^-1 r on entry
int kkt_evaluate(Ctx *c, uint32_t mask, double *out);
int kkt_sums(Ctx *c, uint32_t mask, double *sums);                          // the weighted sums of this context's time slab
int kkt_sums_device(Ctx *c, uint32_t mask, double *device_sums);            // the same, left in the caller's device buffer (enqueue only)
int kkt_combine(Ctx *c, uint32_t mask, const double *sums, double *out);    // the residuals from the sums of the whole problem
int penalty_decision_ahead(Ctx *c, uint32_t mask, const double *sums);       // dots_penalty_ahead: the decision + the next iteration's first launch
bool env_int(const char *name, int lo, int hi, int *out);                   // validated integer switch from the environment
int kkt_n_sums();
int objective_evaluate(Ctx *c, double *out);
int objective_sums(Ctx *c, double *sums);
int objective_combine(Ctx *c, const double *sums, double *out);
int norm_square(Ctx *c, int array_id, int part, double *out);
int reduce_partials(Ctx *c, const double *partials, int n_slots, int n_blocks, int first_slot);  // -> scal[SUMS+first_slot+slot]

struct Ctx {
    Dev d{};
    // The PCG's view of the device data.  Identical to `d` on one GPU.  When the time modes are sharded
    // over ranks it has this rank's column count, its own (smaller) pitch, PCG vectors and sigma slice.
    Dev dcg{};
    // Multi-GPU: this context is one TIME SLAB of the problem.  Rank r holds the nodes [r * stride, r * stride + count) of
    // every state array (d: local pitch, t0, nl, ni) AND solves the time modes with the same indices (dcg).
    int shard_begin = 0;    // first node = first time mode of this context
    int shard_count = 0;    // nodes = modes of this context (may be 0 on trailing ranks)
    int shard_stride = 0;   // nodes per rank (same on every rank) = layout of the gathered buffers; 0 = not sharded
    int shard_ranks = 0;    // ranks with at least one node
    Dev dgt{};              // global-time view for the transforms of a slab context: pitch >= T + 1, Q
    dots_slab_buffers slab{};     // exchange buffers (caller's device memory, dots_slab_set_buffers)
    int64_t slab_b_chunk = 0;     // doubles one rank contributes to the right-hand-side all-gather: V * pitch
    int64_t slab_x_chunk = 0;     // ... to the all-gather of the mode-space solution: V * pitch + V (the last interval's cone multipliers)
    int slab_stage = 0;           // next stage dots_slab_stage expects
    int kkt_halo_fresh = 0;       // mu_lo / B_hi belong to the current iterate
    dots_params prm{};
    int device = 0;
    int lap_solver = 0;
    int nnz = 0;
    int64_t bytes = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev[12]{};
    double *stage = nullptr;      // device staging buffer (largest state array)
    int64_t stage_count = 0;
    double *h_pinned = nullptr;   // pinned host scalars
    double *h_mail = nullptr;     // coherent pinned host memory the device writes itself: [0, MAX_SUMS) sums, [MAX_SUMS] sequence number (fetch_sums)
    uint64_t mail_seq = 0;
    int *kkt_counter = nullptr;   // device counter of k_reduce_mail (last workgroup publishes)
    int spin_fetch = 1;           // DOTS_SPIN_FETCH=0: copy + stream synchronise instead of the device-written mailbox (A/B measurements)
    int64_t mail_spins = 20000000;   // host spins on the mailbox before it blocks on the stream (DOTS_MAIL_SPINS)
    int64_t mail_fallbacks = 0;   // evaluations whose sequence number never arrived: sums copied from the device scalars instead
    int mail_test_drop = 0;       // DOTS_MAIL_TEST_DROP=n (tests): every n-th evaluation publishes a wrong sequence number
    int front_rows = 1;           // row-per-lane-group sweep kernels on bands of short rows (DOTS_FRONT_ROWS: 0 never, 1 by rule, 2 wherever they fit)
    int front_xcd = 1;            // DOTS_FRONT_XCD=0: plain work-list order in the sweeps
    int front_leafinv = 1;        // DOTS_FRONT_LEAFINV=0: the leaves keep [L^-1 ; G] blocks like every other node; 2: local inverses, coupling read from the CSR (no records)
    int front_tune = 0;           // DOTS_FRONT_TUNE=1 print the per-band timing table, 2 also apply the fastest choice
    int *h_flags = nullptr;
    int n_partial_blocks = 0;
    int last_cg_iters = 0;
    int cg_stage_lds = 1;         // stage CSR row blocks in LDS (DOTS_CG_STAGE_LDS=0 disables, for A/B measurements)
    MgDev mg{};                   // multigrid preconditioner (nlev == 0: Jacobi only)
    int use_mg = 1;
    int mg_tail_rows = 256;       // levels with at most this many rows run inside the single tail launch (DOTS_MG_TAIL_ROWS)
    int cg_graph_mg = -1;
    int soc_with_rhs = 1;         // DOTS_SOC_WITH_RHS=0: keep the projection after the solve (A/B measurements)
    int kkt_two = 1;              // KKT sums with two nodes per lane (one GPU; DOTS_KKT_TWO=0: one)
    int ql_two = 1;               // steps 2+3 with two nodes per lane (DOTS_QL_TWO=0: one, for A/B measurements)
    int rhs_two = 1;              // right-hand side + projection with two time columns per lane (DOTS_RHS_TWO=0: one)
    int carry_arrays = 1;         // DOTS_CARRY=0: no carried gathers (A/B measurements; dots_front_setup then allocates nothing for them)
    int rhs_tiles = 0;            // DOTS_RHS_TILES: 1 patch tiles with the triangle rows staged in LDS (measured slower), 2 the plain launch on patch tiles (no difference); default 0
    TileDev tiles{};
    // DOTS_STEP_TIMED: phase events of enqueue-only steps, collected later by dots_step_times (no host wait in the loop)
    static constexpr int TIME_SLOTS = 64;
    hipEvent_t tev[TIME_SLOTS][6]{};  // created on first use
    int tkind[TIME_SLOTS]{};      // 0 / 16: a whole dots_step iteration (6 events; 16: the projection rode in the right-hand-side launch), 1 + stage: one dots_slab_stage (events 0 and 1)
    int t_head = 0, t_count = 0;  // ring: oldest slot, slots in flight
    int step_timed = 0;           // dots_step_flags
    int step_skip_zmid = 0;       // dots_step_flags: steps leave z_mid unspecified (never written, rebuilt on the fly)
    int step_palm = 0;            // dots_step_flags: every iteration opens with the (q, lambda_c) closed form (is_palm = True)
    // A penalty update (dots_adjust_penalty) is not carried out at once: the next iteration's kernels divide the five dual arrays as
    // they read them and steps 2+3 write them back divided (one pass over beta_mid saved: solver_socp.py:367-371 is 5 passes in the
    // reference).  Every other entry point that reads or writes those arrays first carries the division out (flush_division).
    double pending_div = 0.0;     // 0: none
    int lazy_div = 1;             // DOTS_LAZY_DIV=0: divide at once (A/B measurements)
    int bm_nt = 0;                // steps 2+3 stream beta_mid with the non-temporal hint (decided by dots_front_setup: see ql2_lane; DOTS_BM_NT = 0 / 1 overrides)
    int step_kkt = 0;             // dots_step_flags: steps 2+3 also form the KKT sums they hold in registers (kkt_fused)
    KktFused kkt_fused{};         // their per-workgroup partial sums (own buffer: d.partials serves the other reductions)
    int64_t kkt_fused_cap_v = 0, kkt_fused_cap_f = 0;   // workgroups the buffers hold per slot
    int kkt_fused_valid = 0;      // ... and they belong to the current iterate and parameters
    int step_carry = 0;           // dots_step_flags: steps 2+3 also store the next iteration's per-corner gathers (cn_sq, cn_g)
    int carry_valid = 0;          // ... and they belong to the current iterate (cleared by every call that changes state or parameters)
    int zmid_stale = 0;           // z_mid does not belong to the current iterate
    // z_mid on demand (one GPU, carry mapping): an iteration after which z_mid MAY be read (residuals read back, possibly the last one) does
    // not store it (18 T F values that are almost never read); instead steps 2+3 write the new beta_mid into z_mid's storage and the new B
    // into an alternate buffer and the pointers are swapped, so that the old B and beta_mid -- the pre-image of the projection, from which
    // z_mid = (lambda / D) * D (s_z/sqrt3 B_old - beta_mid_old) is rebuilt bit for bit (k_rebuild_zmid) -- survive until the next iteration.
    double *B_alt = nullptr;      // [3F][TP]
    int zmid_deferred = 0;        // z_mid's storage holds the OLD beta_mid, B_alt the OLD B: materialise_zmid() before anything reads z_mid
    double zmid_dv = 0.0, zmid_sz = 0.0;   // the penalty division that was pending during that step (0: none); scale_factor_z of that step
    int zmid_defer = 1;           // DOTS_ZMID_DEFER=0: store z_mid on those iterations as before (A/B measurements)
    FrontDev front{};             // multifrontal factor (n_nodes == 0: none)
    int use_front = 0;
    int front_fwd_ptr[66]{}, front_bwd_ptr[66]{};   // workgroup ranges of the tree levels in fwd_desc / bwd_desc
    int front_fwd_rb[65]{}, front_bwd_cb[65]{};     // rows / columns per workgroup on each level
    int front_fwd_nb[65]{}, front_bwd_nb[65]{};     // threads per workgroup on each level (256, or 1024 where a level has few rows)
    int front_fwd_qw[65]{};       // forward launch of a band: -1 the fold kernel (k_front_fwd), >= 0 the row kernel with 2^qw lane groups per row
    int front_fwd_lds[65]{};      // row kernel: columns of w a workgroup stages in LDS (the band's longest block)
    int front_planes[65]{};       // update planes the forward launch of a band reads per node (0, 2, 4 or 8)
    int front_vec2 = 1;           // two modes per lane in the sweeps (DOTS_FRONT_VEC2: 0 never, 1 / 2 wherever the pitch allows, 3 only where bandwidth-bound)
    int rhs_ahead_armed = 0;      // DOTS_STEP_RHS_AHEAD: the next KKT launch is followed by the next iteration's right-hand side
    int rhs_ahead = 0;            // ... which is on the stream and still valid (any call that changes state or parameters clears it); 2: with the
                                  // cone projection, whose results (z_fst, z_end, the cone multiplier) wait in the alternate buffers below
                                  // 3 / 4: started by the penalty decision ahead of the host (dots_penalty_ahead) with the division `ahead_dv` applied as it read and
                                  // the penalty `ahead_r`; becomes 2 when dots_adjust_penalty (3 -> 4) and dots_set_params (4 -> 2) confirm both
    int penalty_armed = 0;        // dots_penalty_ahead: the next evaluation of conditions 0-3 takes the penalty decision itself
    dots_penalty_policy penalty_policy{};
    double ahead_dv = 0.0, ahead_r = 0.0;   // what the launch ahead anticipated
    int64_t penalty_ahead_started = 0, penalty_ahead_confirmed = 0;   // diagnostics (dots_debug_counter 2, 3)
    double ahead_div = 0.0;       // rhs_ahead == 2: the division the launch ahead applied as it read (steps 2+3 of the step that takes it must apply the same)
    double *zf_alt = nullptr, *ze_alt = nullptr, *lamc_alt = nullptr;   // [V][TP] each (one GPU): written ahead, swapped in by the step that takes them
    int front_rb_max = 4;         // most rows (columns) of a node per workgroup (DOTS_FRONT_RB: 1, 2 or 4, for A/B measurements)
    double front_bytes = 0.0;     // factor bytes one solve reads (both sweeps, merged blocks as stored)
    double front_bytes_unmerged = 0.0;   // the same for one launch per tree height (no merged bands)
    int front_heights = 0;        // tree heights of the installed factor
    int front_top_inverse = 0;    // the top band holds explicit inverses: its forward launch writes x, the backward sweep skips it
    void *front_allocs[48]{};
    int n_front_allocs = 0;
    void *mg_allocs[160]{};
    int n_mg_allocs = 0;
    // constants of the KKT normalisation (solver_socp.py:303-313)
    double c_prim_q = 0, c_prim_z = 0, c_dual_alpha = 0, c_dual_beta = 0, c_comp_rho = 0, c_comp_m = 0;
    void *allocs[64]{};
    int n_allocs = 0;
    // hipGraph cache of the PCG iteration body
    hipGraphExec_t cg_graph = nullptr;
    int cg_graph_iters = 0;
    double cg_graph_eps = -1.0;
    double cg_graph_tol = -1.0;
    double *arr(int id) {
        double *t[12] = {d.phi, d.A, d.B, d.lam, d.zf, d.zm, d.ze, d.mu, d.E, d.bf, d.bm, d.be};
        return t[id];
    }
};

// device copy of a host array owned by the factor (released by front_release)
template <typename T>
inline int front_upload(Ctx *c, const T **out, const T *host, int64_t count) {
    void *p = nullptr;
    const size_t bytes = sizeof(T) * (size_t)(count > 1 ? count : 1);
    DOTS_HIP(hipMalloc(&p, bytes));
    if (c->n_front_allocs >= (int)(sizeof(c->front_allocs) / sizeof(c->front_allocs[0]))) {
        (void)hipFree(p);
        set_error("front allocation table full");
        return DOTS_ERR_STATE;
    }
    c->front_allocs[c->n_front_allocs++] = p;
    if (host) DOTS_HIP(hipMemcpyAsync(p, host, sizeof(T) * (size_t)count, hipMemcpyHostToDevice, c->stream));
    else DOTS_HIP(hipMemsetAsync(p, 0, bytes, c->stream));
    DOTS_HIP(hipStreamSynchronize(c->stream));
    *out = (const T *)p;
    return 0;
}

// The time-mode transforms stage a tile of rows and Q (in chunks of <= 32 KB) in LDS (k_time_modes_tile, k_rhs_modes)
// T + 1 >= 64: the transforms are [V x (T+1)] x [(T+1) x (T+1)] fp64 GEMMs worth the matrix cores (k_time_modes_mfma)
inline bool time_modes_mfma_ok(const Dev &d) { return d.TP >= 64 && d.TP <= 256 && d.Qpad != nullptr; }
inline bool time_modes_tile_ok(const Dev &d) { return d.TP <= BLOCK && d.VT >= 1; }
inline int time_modes_chunk(const Dev &d) { return (4096 / d.TP) < (d.T + 1) ? (4096 / d.TP) : (d.T + 1); }    // rows of Q per chunk
inline size_t time_modes_tile_lds(const Dev &d) { return sizeof(double) * ((size_t)time_modes_chunk(d) * d.TP + (size_t)d.VT * (d.TP + 1)); }
// direct solver on one GPU, T + 1 < 64: the inverse time transform of phi rides in the cone-projection launch (the
// projection reads neither phi nor anything the solve writes: steps 1-1 and 1-2 are a separable block)
inline bool soc_takes_inverse(const Ctx *c);
// direct solver on one GPU: the right-hand-side kernel writes the mode-space right-hand side itself
inline bool rhs_writes_modes(const Ctx *c) {
    return c->use_front && c->front.n_nodes > 0 && c->shard_stride == 0 && c->lap_solver == DOTS_LAP_MODAL_PCG && time_modes_tile_ok(c->d);
}

inline bool soc_takes_inverse(const Ctx *c) { return rhs_writes_modes(c) && !time_modes_mfma_ok(c->d); }
// steps 2+3 can form the next iteration's per-corner gathers (k_q_lambda_mult_carry: whole triangles per 192-lane workgroup)
// (one GPU or a time slab; the direct solver's iteration)
inline bool carry_possible(const Ctx *c) {
    return c->d.cn_sq && c->ql_two && c->d.TP >= 4 && c->d.TP <= 128 && c->use_front && c->front.n_nodes > 0 && c->lap_solver == DOTS_LAP_MODAL_PCG &&
           (rhs_writes_modes(c) || c->d.slab);
}
// ... or the projection itself rides in the right-hand-side launch (enqueue-only iterations; TILE_ELEMS threads per tile)
inline bool rhs_takes_soc(const Ctx *c) { return c->soc_with_rhs && rhs_writes_modes(c); }

// a KKT evaluation of the conditions in `mask` reduces the sums the last steps-2+3 launch left (kernels_kkt.hip: kkt_sums) and reads no z_mid
inline bool kkt_takes_fused(const Ctx *c, uint32_t mask) {
    return c->kkt_fused_valid && !(mask & ~(KKT_FUSED_MASK | 4u)) && c->spin_fetch && c->h_mail && c->kkt_counter;
}

int64_t array_count_host(const Dev &d, int array_id);    // elements in the reference layout
int64_t array_count_device(const Dev &d, int array_id);  // elements in the device layout
int array_kind(int array_id);  // 0 node (T+1,V), 1 interval (T,V), 2 triangle (T+1,F,3), 3 corner (T,2,3,F,3)

#ifdef __HIPCC__
// ---- device helpers ---------------------------------------------------------------------
__device__ __forceinline__ int idxV(const Dev &d, int v, int t) { return (v << d.tp_shift) + t; }
__device__ __forceinline__ int64_t idxF(const Dev &d, int f, int c, int t) { return ((int64_t)(f * 3 + c) << d.tp_shift) + t; }
// t: LOCAL interval index (interval - t0), -1 <= t; the entry sits in the column of its node t + s
__device__ __forceinline__ int64_t idxM(const Dev &d, int fk, int s, int c, int t) {
    return ((int64_t)((fk * 2 + s) * 3 + c) << d.tp_shift) + t + s;
}
// two consecutive time columns of a row in one aligned 16-byte word (even first column)
struct D2 { double v[2]; };
__device__ __forceinline__ D2 ld2(const double *p) { const double2 t = *reinterpret_cast<const double2 *>(p); return D2{{t.x, t.y}}; }
__device__ __forceinline__ void st2(double *p, const D2 &x) { *reinterpret_cast<double2 *>(p) = make_double2(x.v[0], x.v[1]); }
// the same with a streaming hint (non-temporal): data written once and read once by the next launch (the carried per-corner sums)
// or not read at all in the loop (z_mid on read-back iterations) should not displace the factor and beta_mid from the caches.
// A/B on one box (-DDOTS_CARRY_NT=0 against the default): knot 9 720-10 020 -> 10 160-10 210 it/s, sphere10k 4 795-4 813 -> 4 936-4 983,
// torus100k 619 -> 631-646, knot63 unchanged (profiles/studies/r04_nontemporal.txt)
#ifndef DOTS_CARRY_NT
#define DOTS_CARRY_NT 1
#endif
typedef double dots_d2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ D2 ld2_nt(const double *p) {
#if DOTS_CARRY_NT
    const dots_d2v t = __builtin_nontemporal_load(reinterpret_cast<const dots_d2v *>(p));
    return D2{{t.x, t.y}};
#else
    return ld2(p);
#endif
}
__device__ __forceinline__ double ld1_nt(const double *p) {
#if DOTS_CARRY_NT
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
__device__ __forceinline__ void st2_nt(double *p, const D2 &x) {
#if DOTS_CARRY_NT
    dots_d2v t;
    t.x = x.v[0];
    t.y = x.v[1];
    __builtin_nontemporal_store(t, reinterpret_cast<dots_d2v *>(p));
#else
    st2(p, x);
#endif
}
// slab predicates for local column t
__device__ __forceinline__ bool first_node(const Dev &d, int t) { return d.t0 + t == 0; }       // global node 0
__device__ __forceinline__ bool last_node(const Dev &d, int t) { return d.t0 + t == d.T; }      // global node T
__device__ __forceinline__ bool has_prev_interval(const Dev &d, int t) { return d.t0 + t > 0; } // interval t - 1 exists
// node t + 1 of a node array x (halo when it belongs to the next slab)
__device__ __forceinline__ double next_node(const Dev &d, const double *x, const double *hi, int64_t row, int t) {
    return (t + 1 < d.nl) ? x[(row << d.tp_shift) + t + 1] : hi[row];
}

// Blocks are dealt round-robin over the 8 XCDs (b and b+8 share an L2).  Give every XCD one
// contiguous range of tiles so that neighbouring tiles, which gather the same rows, share an L2.
__device__ __forceinline__ int xcd_tile(int b, int n_tiles) {
    int per = (n_tiles + 7) >> 3;
    return (b & 7) * per + (b >> 3);
}
inline int xcd_grid(int n_tiles) { return ((n_tiles + 7) / 8) * 8; }

__device__ __forceinline__ double wave_sum(double x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
    return x;
}

// Sum N per-thread values over the workgroup; thread 0 gets the totals.  Fixed order -> deterministic.
template <int N, int NW = 4>
__device__ __forceinline__ void block_sum(double (&v)[N], double *lds /* [N*4] */) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double s = wave_sum(v[i]);
        if (lane == 0) lds[i * 4 + w] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = NW == 4 ? (lds[i * 4] + lds[i * 4 + 1]) + (lds[i * 4 + 2] + lds[i * 4 + 3]) : (lds[i * 4] + lds[i * 4 + 1]) + lds[i * 4 + 2];
    }
}
// The same product on the matrix cores (T + 1 >= 64): one v_mfma_f64_16x16x4_f64 per 16 x 16 output tile and 4
// values of i.  xs holds MROWS = 32 rows ([32][TP + 1], zero padded); wavefront w of the workgroup's NW takes the
// (row tile, column tile) pairs w, w + NW, ...  Operand layout of the instruction (lane l), measured on gfx950
// (profiles/micro/mfma_f64_layout.hip): A[i = l % 16][k = l / 16], B[k = l / 16][j = l % 16],
// D[i = l / 16 + 4 r][j = l % 16] in the r-th accumulator register.
// Qe = zero-padded Q (time -> modes) or Q^T (modes -> time), [TP][TP] row-major: no bounds checks in the loop.
typedef double mfma_f64x4 __attribute__((ext_vector_type(4)));
constexpr int TM_ROWS = 32;
template <int NW>
__device__ __forceinline__ void modes_from_tile_mfma(const Dev &d, const double *__restrict__ Qe, const double *xs, int v0, double *__restrict__ y) {
    const int n = d.T + 1, TP = d.TP, TPp = TP + 1;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int n_jt = TP >> 4;                            // column tiles of 16
    for (int tile = w; tile < 2 * n_jt; tile += NW) {
        const int jt = tile % n_jt, vt = tile / n_jt;
        mfma_f64x4 acc = {0.0, 0.0, 0.0, 0.0};
        const double *__restrict__ bq = Qe + (int64_t)lk * TP + jt * 16 + li;
        const double *a = xs + (vt * 16 + li) * TPp + lk;
        for (int k0 = 0; k0 < TP; k0 += 4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[k0], bq[(int64_t)k0 * TP], acc, 0, 0, 0);
        const int j = jt * 16 + li;
        if (j < n) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int v = v0 + vt * 16 + lk + 4 * r;
                if (v < d.V) y[idxV(d, v, j)] = acc[r];
            }
        }
    }
}

// y[v0 + vl][j] = sum_i Qeff[i][j] xs[vl][i] for the tile staged in xs ([VT][TP + 1], zero padded), with
// Qeff[i][j] = Q[i][j] (FWD: time -> modes) or Q[j][i] (modes -> time), staged through Qs in chunks of IC rows.
// A thread computes up to four outputs that share their Q column.  All threads of the workgroup must call it.
// Only the outputs j in [j0, j0 + jn) are stored, at y[(v << out_shift) + (j - j0)]  (defaults: all of them, pitch TP).
// rows [i0, i0 + ic) of Qeff into Qs.  A caller may stage the first chunk itself BEFORE its own loads of the tile (staged0):
// the Q loads then overlap them instead of adding a memory round trip behind the first barrier.
template <bool FWD, int NB = BLOCK>
__device__ __forceinline__ void stage_q_chunk(const Dev &d, const double *Q, double *Qs, int i0, int ic) {
    const int n = d.T + 1, TP = d.TP;
    for (int e = threadIdx.x; e < ic * TP; e += NB) {
        const int i = i0 + (e >> d.tp_shift), jj = e & (TP - 1);
        Qs[e] = jj < n ? (FWD ? Q[i * n + jj] : Q[jj * n + i]) : 0.0;
    }
}
template <bool FWD, int NB = BLOCK>
__device__ __forceinline__ void modes_from_tile(const Dev &d, const double *Q, const double *xs, double *Qs, int IC, int v0, double *__restrict__ y,
                                                int out_shift = -1, int j0 = 0, int jn = 1 << 30, bool staged0 = false, const int *__restrict__ vids = nullptr) {
    if (out_shift < 0) out_shift = d.tp_shift;
    const int n = d.T + 1, TP = d.TP, TPp = TP + 1, tid = threadIdx.x;
    const int j = tid & (TP - 1), g = tid >> d.tp_shift, G = NB >> d.tp_shift;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int i0 = 0; i0 < n; i0 += IC) {
        const int ic = min(IC, n - i0);
        if (!(staged0 && i0 == 0)) {
            __syncthreads();      // the previous chunk (or the caller's staging of xs) is complete
            stage_q_chunk<FWD, NB>(d, Q, Qs, i0, ic);
        }
        __syncthreads();
        if (j < n && g < d.VT) {
            const double *x0 = xs + g * TPp + i0;
            for (int i = 0; i < ic; ++i) {
                const double q = Qs[(i << d.tp_shift) + j];
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (g + r * G < d.VT) acc[r] += q * x0[r * G * TPp + i];
            }
        }
    }
    if (j < n && j >= j0 && j - j0 < jn) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int vl = g + r * G;
            if (vl >= d.VT) continue;
            const int vv = vids ? vids[vl] : v0 + vl;       // (vids: the tile's own vertex list, -1 = padding)
            if (vv >= 0 && vv < d.V) y[((int64_t)vv << out_shift) + (j - j0)] = acc[r];
        }
    }
}

#endif

}  // namespace dots
