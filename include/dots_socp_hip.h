/*
 * dots_socp_hip.h -- C ABI of libdotsocp_hip.so, the MI355X (gfx950) implementation of the
 * DOTs-SOCP ALM hot path.
 *
 * The reference is pure Python (no FFI exists in it); these entry points are what a ctypes
 * binding placed at the reference's solver plug-in boundary would bind.  Each function names the
 * reference code it replaces (paths relative to the reference root, file:line).
 *
 * Conventions
 *   - every function returns 0 on success, a negative dots_status otherwise; the message of the
 *     last failure is available from dots_last_error() (per thread).
 *   - host pointers are borrowed for the duration of the call only; device memory belongs to the
 *     context.  One host thread per context.
 *   - host-side state arrays use the REFERENCE layouts and fp64:
 *       phi                          (T+1, V)
 *       A, lambda_c, mu, z_fst, z_end, beta_fst, beta_end   (T, V)
 *       B, E                         (T+1, F, 3)
 *       z_mid, beta_mid              (T, 2, 3, F, 3)
 *     the device layout (vertex/triangle-major, time fastest) is internal.
 *     A time-slab context (multi-GPU, dots_problem_desc.slab_*) exchanges its OWN time extent only:
 *       phi (n, V); interval arrays (m, V); B, E (n, F, 3) with n = slab_count nodes, m = min(n, T - slab_begin)
 *       intervals; z_mid, beta_mid (n, 2, 3, F, 3) indexed by the NODE an entry is compared with: entry [j][s] is the
 *       reference's [slab_begin + j - s][s] (entries whose interval does not exist: ignored on upload, zero on download).
 *   - no C++ exceptions, torch types or Python objects cross this boundary.
 */
#ifndef DOTS_SOCP_HIP_H
#define DOTS_SOCP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: only what this header declares is exported */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

#define DOTS_ABI_VERSION 7

typedef struct dots_ctx dots_ctx;

enum dots_status {
    DOTS_OK = 0,
    DOTS_ERR_ARGUMENT = -1,
    DOTS_ERR_HIP = -2,
    DOTS_ERR_NO_DEVICE = -3,
    DOTS_ERR_NOT_CONVERGED = -4,
    DOTS_ERR_STATE = -5,
    DOTS_ERR_MEMORY = -6      /* dots_front_setup: the factor (+ its workspace) does not fit the device memory that is free */
};

/* state arrays, same names as SolutionSocpData (dot_surface_socp/utils/type.py:22-38) */
enum dots_array {
    DOTS_PHI = 0, DOTS_A, DOTS_B, DOTS_LAMBDA_C, DOTS_Z_FST, DOTS_Z_MID, DOTS_Z_END,
    DOTS_MU, DOTS_E, DOTS_BETA_FST, DOTS_BETA_MID, DOTS_BETA_END,
    DOTS_N_ARRAYS
};

/* the seven KKT residuals in the order of solver_socp.py:590-640 */
enum dots_kkt_id {
    DOTS_KKT_PRIM_Q = 0, DOTS_KKT_PRIM_Z, DOTS_KKT_DUAL_ALPHA, DOTS_KKT_DUAL_BETA,
    DOTS_KKT_COMP_RHO_FQ, DOTS_KKT_COMP_M_RHO_B, DOTS_KKT_COMP_CONGESTION,
    DOTS_N_KKT
};

/* which Laplacian solver runs in step 1 (replaces laplacian_inverse_socp.py:11-61) */
enum dots_lap_solver {
    DOTS_LAP_SPACETIME_PCG = 0, /* Jacobi-PCG on the assembled space-time operator            */
    DOTS_LAP_MODAL_PCG = 1      /* time eigen-modes decoupled (DCT): batched shifted-surface PCG, or -- once a factor is
                                   installed with dots_front_setup -- the direct multifrontal sweeps (the default of
                                   the Python driver: lap_solver="modal_direct")                  */
};

/*
 * Problem description: the outputs of the reference's operator assembly
 * (utils/surface_pre_computations_socp.py:11-132, solver_socp.py:102-113,161-192) as flat arrays.
 * The host side (dots_socp_amd/geometry.py) builds them; all index arrays are 0-based int32.
 */
typedef struct dots_problem_desc {
    int32_t abi_version;     /* DOTS_ABI_VERSION */
    int32_t device;          /* HIP device ordinal */
    int32_t n_time;          /* T: number of time intervals (T+1 nodes) */
    int32_t n_vertices;      /* V */
    int32_t n_triangles;     /* F */
    int32_t n_corners;       /* = 3 F, length of the vertex->corner lists */
    int32_t lap_nnz;         /* non-zeros of the surface stiffness matrix */
    int32_t lap_solver;      /* enum dots_lap_solver */

    const int32_t *triangles;    /* [F][3]                                                     */
    const double *hat_grad;      /* [F][3 corners][3 xyz] hat-function gradients (:31-37)       */
    const double *area_tri;      /* [F]                                                        */
    const double *mass_vert;     /* [V]  = (sum of incident triangle areas) / 3  (solver_socp.py:112) */
    const int32_t *corner_ptr;   /* [V+1] vertex -> corner list (CSR)                           */
    const int32_t *corner_idx;   /* [3F]  entries f*3+k, the corners (f,k) with triangles[f][k]==v */
    const int32_t *lap_rowptr;   /* [V+1]  K = -cotangent Laplacian (= G^T diag(area) G), CSR, SPD-semidefinite */
    const int32_t *lap_col;      /* [nnz]                                                       */
    const double *lap_val;       /* [nnz]                                                       */
    const double *mu0;           /* [V] boundary masses (geometry["mu0"], solver_socp.py:267-270) */
    const double *mu1;           /* [V]                                                         */
    const int32_t *perm_vert;    /* [V] device vertex i is caller vertex perm_vert[i]; NULL = identity */
    const int32_t *perm_tri;     /* [F] device triangle i is caller triangle perm_tri[i]; NULL = identity */
    const double *time_modes;    /* [(T+1)*(T+1)] row-major Q[t][a]: orthonormal eigenvectors of the Neumann
                                    time Laplacian (laplacian_inverse_socp.py:15-31); NULL unless MODAL */
    const double *time_eigs;     /* [T+1] eigenvalues sigma_a >= 0 of -L_time;  NULL unless MODAL */
    /* Multi-GPU (one context per rank): TIME SLABS.  This context holds the time nodes [slab_begin, slab_begin +
     * slab_count) of every state array (and the intervals that start at them) and solves the time modes with the same
     * indices; slab_stride = nodes per rank (ceil((T+1)/n_ranks), the same on every rank; slab_count may be smaller, or
     * 0, on trailing ranks).  All zero = single GPU.  See dots_slab_stage. */
    int32_t slab_begin;
    int32_t slab_count;
    int32_t slab_stride;
    int32_t reserved;
    /* Optional (NULL: off): a sequence of all device vertices in which every aligned run of 16 * 2^k entries is a compact
     * patch of the surface (dots_patch_order).  The right-hand-side / cone-projection launch then takes its vertex tiles
     * from this sequence and stages the B, E rows of a tile's distinct triangles through LDS once, instead of gathering
     * them per corner (solver_socp.py:909-921, :997-1017: the two SpMVs with incidence matrices).  Results do not change. */
    const int32_t *patch_order;  /* [V] */
} dots_problem_desc;

/* Scalars the host control logic owns (solver_socp.py:97,318-321 and the kwargs of :25-41). */
typedef struct dots_params {
    double r;              /* penalty                                   */
    double scale_z;        /* scale_factor_z                            */
    double const_d;        /* constant_d                                */
    double norm_d;         /* norm_constant_d (:297,380)                */
    double norm_boundary;  /* norm_boundary (:296)                      */
    double congestion;
    double tau;
    double eps;
    double prim_scale;
    double dual_scale;
    double boundary_scale; /* multiplies the boundary term (-mu0, +mu1)/(r h); 1 unless constant scaling (:357-358) */
    double cg_tol;         /* PCG stops when r^T M^-1 r <= cg_tol^2 * b^T M^-1 b */
    int32_t cg_max_iter;
    int32_t reserved;
} dots_params;

typedef struct dots_step_stats {
    int32_t alm_iterations;     /* iterations performed by this call                     */
    int32_t cg_iterations;      /* PCG iterations summed over them                        */
    int32_t cg_last_iterations; /* PCG iterations of the last one                         */
    int32_t cg_not_converged;   /* number of solves that hit cg_max_iter                  */
    double cg_last_rel_residual;
    double ms_rhs;              /* hipEvent times summed over the call, milliseconds      */
    double ms_laplacian;
    double ms_soc;
    double ms_q_lambda_multiplier;
    double ms_total;
} dots_step_stats;

/* ---- lifecycle ------------------------------------------------------------------------- */
int dots_abi_version(void);
const char *dots_last_error(void);

/* Build the device-resident problem: replaces the setup of solver_socp.py:96-313 (constants,
 * Laplacian operator, zero-initialised state as :239-250 with an empty init_solution). */
int dots_create(const dots_problem_desc *desc, dots_ctx **out);
int dots_destroy(dots_ctx *ctx);
int dots_set_params(dots_ctx *ctx, const dots_params *p);
int dots_get_params(dots_ctx *ctx, dots_params *p);
int dots_sync(dots_ctx *ctx);

/* ---- state transfer (init_solution in, SolutionSocpData out; solver_socp.py:239-250,855-869) */
int dots_upload(dots_ctx *ctx, int array_id, const double *host, int64_t count);
int dots_download(dots_ctx *ctx, int array_id, double *host, int64_t count);
int64_t dots_array_count(dots_ctx *ctx, int array_id);

/* ---- the hot loop ------------------------------------------------------------------------ */
/* n ALM iterations, steps 1-3 of solver_socp.py:674-722 (is_palm = False), device resident.
 * stats == NULL: the iterations are only enqueued on the context's stream (no host wait, nothing timed;
 * meant for the direct solver, which needs no host round trip) -- any later call that returns data waits. */
int dots_step(dots_ctx *ctx, int n_iters, dots_step_stats *stats);

/* Flags for the following dots_step / dots_slab_stage calls.
 * DOTS_STEP_SKIP_Z_MID: z_mid (18*T*F values, an intermediate between the cone projection and steps 2+3) is
 * rebuilt on the fly and not stored: the iterate is the same bit for bit, but z_mid in memory is unspecified
 * afterwards.  dots_kkt(PRIM_Z) and dots_download(Z_MID) then fail with DOTS_ERR_STATE until a step without the
 * flag (or an upload / the projection phase) has produced it again.  The host driver sets it on the iterations
 * after which neither KKT residuals nor the solution are read (solver_socp.py's lazy validator, :766-788). */
#define DOTS_STEP_SKIP_Z_MID 1u
/* DOTS_STEP_PALM: every iteration opens with the (q, lambda_c) closed form alone ("Step 0" of is_palm = True,
 * solver_socp.py:668-672: A, B, lambda_c from the current multipliers, the stored z_mid and grad(phi) of the previous
 * iteration).  It reads z_mid from memory, so it cannot be combined with DOTS_STEP_SKIP_Z_MID (DOTS_ERR_ARGUMENT). */
#define DOTS_STEP_PALM 2u
/* DOTS_STEP_RHS_AHEAD (a hint: ignored without the direct solver or on a time slab; DOTS_ERR_ARGUMENT with DOTS_STEP_PALM): the caller expects the
 * iteration AFTER the next dots_step to start from the state that step leaves (no penalty update, no rescaling in between).
 * The first dots_kkt / dots_kkt_sums call after that step then enqueues the next iteration's right-hand side behind its own
 * kernels, so that the device works while the host waits for the residuals and decides; the following dots_step starts at the
 * solve.  Any call in between that changes the state or the parameters (dots_set_params, dots_upload, dots_adjust_penalty,
 * dots_scale_*, dots_run_phase, ...) drops that right-hand side and the step computes it again: results never depend on the
 * flag.  The flag holds for one dots_step. */
#define DOTS_STEP_RHS_AHEAD 4u
/* DOTS_STEP_TIMED: enqueue-only steps (dots_step / dots_slab_stage with stats == NULL) bracket their phases with events of an
 * internal ring of 64 slots (one per dots_step iteration, one per slab stage) WITHOUT waiting for them; dots_step_times collects
 * the finished slots later, oldest first (the reference's per-step timers, utils/admm_tools.py:244-251, without a host wait in
 * the loop).  A step that finds the ring full is simply not timed. */
#define DOTS_STEP_TIMED 8u
/* DOTS_STEP_CARRY (a hint: ignored without the direct solver, on a time slab, with DOTS_STEP_PALM or a time pitch above 128): the caller
 * expects the NEXT iteration to start from the state this step leaves.  Steps 2+3 then also store, per corner of every triangle,
 * what the next right-hand side (solver_socp.py:983-986: div_x((B - E) area)) and the next cone projection (:997-1017: the squared
 * norm of D (L B - beta_mid)) would gather from B, E and beta_mid -- they hold those values in registers when they write them
 * (:716-722) -- and the next dots_step streams these per-corner sums instead of reading beta_mid a third time.  Any call in
 * between that changes the state or the parameters drops them and the step gathers from the arrays again: results never depend
 * on the flag, bit for bit (both paths form the same sums in the same order). */
#define DOTS_STEP_CARRY 16u
/* DOTS_STEP_KKT_SUMS (a hint: one GPU, ignored with DOTS_STEP_SKIP_Z_MID): the caller will read KKT residuals after this step.  Steps 2+3
 * then also accumulate the weighted sums of Prim(phi, q), Prim(q, z), Dual(beta) and Comp(rho, cong.) (solver_socp.py:433-464,
 * 484-503, 549-559) from the values they hold in registers as they write the new iterate; a following dots_kkt / dots_kkt_sums
 * whose mask holds only these conditions and Dual(alpha) reduces those partial sums (plus one vertex pass for Dual(alpha))
 * instead of reading the state again.  Any call in between that changes the state or the parameters drops them.  The residuals
 * agree with the stand-alone evaluation to rounding (the sums are formed in a different order). */
#define DOTS_STEP_KKT_SUMS 32u
int dots_step_flags(dots_ctx *ctx, uint32_t flags);      /* also apply to dots_slab_stage */
/* The phase times of the timed steps that have finished (wait != 0: of all timed steps, waiting for them), oldest first, at
 * most `capacity`: one dots_step_stats per dots_step iteration (alm_iterations = 1), or one per slab stage (the stage's time in
 * the field of its phase; stage 3 carries alm_iterations = 1). */
int dots_step_times(dots_ctx *ctx, dots_step_stats *out, int capacity, int wait, int *n_out);

/* ---- time slabs (multi-GPU): one ALM iteration in four stages around three exchanges --------------------------------
 * Every operator of the iteration is local in time except nearest-neighbour couplings (solver_socp.py:884, :892-894,
 * :934-940, :955-957), and the Laplacian solve decouples over the time MODES (laplacian_inverse_socp.py:31-41).  Rank r
 * holds the slab of nodes [r s, (r+1) s) of the whole state (s = slab_stride) and solves the modes [r s, (r+1) s):
 *
 *   stage 0   [is_palm: the (q, lambda_c) closed form]; pack what the neighbours need:
 *               send_x   [V]  (A + lambda_c - mu) of this slab's last interval          -> next rank's  recv_x
 *               send_nsq [V]  the s = 1 half of the cone norms of the interval that ends at this slab's first node
 *                             (formed from this slab's B and beta_mid)                    -> previous rank's recv_nsq
 *   (caller)  neighbour exchange (two V-sized messages per slab boundary)
 *   stage 1   right-hand side of this slab's nodes -> b_send = [V][pitch], and the cone projection of its intervals (one launch);
 *             the cone multipliers of the slab's last interval go to the tail [V] of x_send (the next slab's steps 2+3 read them)
 *             -- or in two halves, so that the exchange overlaps the projection:
 *   stage 6     the right-hand side alone;  (caller) starts the all-gather of b_send;
 *   stage 5     the cone projection alone, enqueued behind stage 6 on the context's stream while the all-gather runs on the
 *               caller's (the projection reads nothing the right-hand side, the all-gather or the solve writes: steps 1-1 and
 *               1-2 of solver_socp.py:674-696 minimise a separable block -- the reference runs them on two threads)
 *   (caller)  all-gather of b_send into b_recv = [n_ranks][V * pitch]
 *   stage 2   forward time transform restricted to this rank's modes, solve (sweeps or PCG); x_send = [V][pitch] solution (+ tail)
 *   (caller)  all-gather of x_send into x_recv = [n_ranks][V * pitch + V]
 *   stage 3   inverse time transform for this slab's nodes (+ the next slab's first node, computed redundantly),
 *             steps 2 and 3
 *   stage 4   (only before KKT residuals are evaluated) pack send_mu [V] (mu of the last interval -> next rank) and
 *             send_b [3F] (B of the first node -> previous rank); the caller exchanges them, then dots_kkt_sums
 *
 * Stages must be called in order 0,1,2,3 or 0,6,5,2,3 (DOTS_ERR_STATE otherwise).  With stats == NULL a stage is only enqueued on the
 * context's stream: the caller orders its exchanges against it with dots_stream_wait (no host wait anywhere).
 * All buffers are DEVICE memory owned by the caller (e.g. torch tensors handed to RCCL), registered once with
 * dots_slab_set_buffers; sizes from dots_slab_elems.  Results are bit-identical for every number of ranks, 1 included
 * (the same sums are formed in the same order wherever a value is computed); only the KKT / objective sums differ in
 * rounding (partial sums per slab). */
typedef struct dots_slab_buffers {
    double *send_x, *send_nsq;      /* [V] each: stage 0 output                                              */
    double *recv_x, *recv_nsq;      /* [V] each: from the previous / next rank                               */
    double *b_send, *b_recv;        /* [V * pitch],     [n_ranks][V * pitch]                                 */
    double *x_send, *x_recv;        /* [V * pitch + V], [n_ranks][V * pitch + V]                             */
    double *send_mu, *send_b;       /* [V], [3F]: stage 4 output                                             */
    double *recv_mu, *recv_b;       /* [V], [3F]: from the previous / next rank                              */
} dots_slab_buffers;
enum dots_slab_size { DOTS_SLAB_VERTEX_HALO = 0, DOTS_SLAB_B_CHUNK = 1, DOTS_SLAB_X_CHUNK = 2, DOTS_SLAB_TRIANGLE_HALO = 3 };
int64_t dots_slab_elems(dots_ctx *ctx, int which);      /* doubles; -1 if the context is not a time slab */
int dots_slab_set_buffers(dots_ctx *ctx, const dots_slab_buffers *buffers);
int dots_slab_stage(dots_ctx *ctx, int stage, dots_step_stats *stats);
/* Stream ordering between the context's stream and another HIP stream of the same device (e.g. the one a
 * communication library works on; NULL = the legacy default stream).  ctx_waits = 0: work enqueued on
 * `other_stream` after the call waits for everything enqueued on the context so far; 1: the reverse. */
int dots_stream_wait(dots_ctx *ctx, void *other_stream, int ctx_waits);

/* single phases of one iteration, for per-function parity tests */
enum dots_phase {
    DOTS_PHASE_LAPLACIAN = 0,        /* vanilla_solve_laplacian  solver_socp.py:976-986 + laplacian_inverse_socp.py:52-61 */
    DOTS_PHASE_SOC_PROJECTION = 1,   /* vanilla_solve_proj_soc   solver_socp.py:988-1042 */
    DOTS_PHASE_Q_LAMBDA_MULT = 2,    /* grad_time/grad_space + vanilla_solve_q_lambda :709-714,1044-1065 and the multiplier update :716-722 */
    DOTS_PHASE_Q_LAMBDA = 3          /* vanilla_solve_q_lambda alone (is_palm's step 0, :668-672): no multiplier moves */
};
int dots_run_phase(dots_ctx *ctx, int phase, dots_step_stats *stats);

/* KKT residuals (closures of solver_socp.py:433-559 as wired at :589-643).  mask: bit i set =
 * evaluate condition i.  out[2*i], out[2*i+1] = the two values the reference's wrapper returns
 * (with prim/dual scale, with scale 1); entries of conditions not in mask are left untouched.
 * For conditions 4..6 the second value does not exist in the reference and is set to NaN. */
int dots_kkt(dots_ctx *ctx, uint32_t mask, double *out /* [2*DOTS_N_KKT] */);
/* The same in two halves, for time slabs: the weighted sums of this context's slab (DOTS_KKT_N_SUMS doubles; on a slab
 * the halos of stage 4 must have been exchanged), and the residuals from the sums of the WHOLE problem (the caller adds
 * the slabs' sums: one all-reduce of a small vector, solver_socp.py:433-559). */
#define DOTS_KKT_N_SUMS 24
int dots_kkt_sums(dots_ctx *ctx, uint32_t mask, double *sums /* [DOTS_KKT_N_SUMS] */);
int dots_kkt_combine(dots_ctx *ctx, uint32_t mask, const double *sums, double *out /* [2*DOTS_N_KKT] */);
/* dots_kkt_sums with the result left in DEVICE memory of the caller (DOTS_KKT_N_SUMS doubles, e.g. a torch tensor handed to
 * RCCL's all-reduce): only enqueued on the context's stream, no host wait, no copy -- order the consumer's stream with
 * dots_stream_wait.  Slots of sums the mask does not need (and every slot on a rank without nodes) are zero. */
int dots_kkt_sums_device(dots_ctx *ctx, uint32_t mask, double *device_sums /* [DOTS_KKT_N_SUMS], device memory */);

/* The penalty decision ahead of the host (a hint; one GPU, direct solver).  On the iterations where the reference adapts the penalty
 * (utils/admm_tools.py:30-52) the host reads conditions 0-3, takes the decision (solver_socp.py:806-823, admm_tools.py:54-95), divides
 * the dual arrays and only then starts the next iteration: the device idles meanwhile.  Armed with the policy's numbers, the next
 * dots_kkt / dots_kkt_sums whose mask holds conditions 0-3 takes the SAME decision inside the library as soon as the sums have
 * arrived -- if one of the four conditions fails tol: is_org_kkt |= (max of the unit-scale values < 5 tol); gap = max(prim) / max(dual) of
 * the chosen values; factor from the table (the first threshold that max(gap, 1/gap) exceeds; inverted when gap < 1); r' = clamp(r *
 * factor) -- and starts the next iteration's right-hand side + cone projection with r * (r' / r) and the division by r' / r applied
 * as the arrays are read (results in alternate buffers, as with DOTS_STEP_RHS_AHEAD).  The caller then calls dots_adjust_penalty and
 * dots_set_params as it would anyway: if factor and parameters are the ones anticipated (bit for bit) the next dots_step starts at the
 * solve, otherwise -- or after any other call that changes state -- what was started is dropped.  Results never depend on the hint. */
typedef struct dots_penalty_policy {
    double tol;               /* the conditions pass below tol                                            */
    double r_lower, r_upper;  /* the penalty's clamp (admm_tools.py:25-26)                                 */
    int32_t is_org_kkt;       /* the driver's sticky flag (solver_socp.py:806-808)                         */
    int32_t n_steps;          /* entries of the table, <= 16                                               */
    double threshold[16];     /* descending (admm_tools.py:79-90)                                          */
    double factor[16];
} dots_penalty_policy;
int dots_penalty_ahead(dots_ctx *ctx, const dots_penalty_policy *policy);

/* objective_functional (solver_socp.py:417-431) as called at :773-775/:829-831:
 * out[0] = transportation cost, out[1] = Lagrangian / objective value. */
int dots_objective(dots_ctx *ctx, double *out /* [2] */);
int dots_objective_sums(dots_ctx *ctx, double *sums /* [3] */);
int dots_objective_combine(dots_ctx *ctx, const double *sums /* [3] */, double *out /* [2] */);

/* scaling tools (solver_socp.py:367-395).  These update the arrays only; the caller keeps
 * r / scale_z / const_d in dots_params consistent (dots_set_params). */
int dots_adjust_penalty(dots_ctx *ctx, double factor);             /* adjust_penalty :367-371: 5 dual arrays /= factor */
int dots_scale_z(dots_ctx *ctx, double z_mul, double beta_mul, double scale_z_new);
        /* scale_variable_z :373-395: z *= z_mul, beta *= beta_mul, mu = scale_z_new*(beta_fst-beta_end),
           E = -L^T(beta_mid; scale_z_new) */
int dots_scale_arrays(dots_ctx *ctx, uint32_t array_mask, double factor); /* x *= factor for every array in mask (scale_prim_dual :352-358) */

/* weighted squared norms norm_square_* (solver_socp.py:875-878, :215-218) of one state array,
 * and of dt_phi / dx_phi when array_id is DOTS_PHI with part = 1 / 2.
 * On a time slab: the slab's SHARE of the norm (the sum over its own nodes / intervals / corner entries divided by the
 * GLOBAL averaging count); the caller adds the shares of all slabs (is_constant_scaling, solver_socp.py:324-365).  part = 1
 * reads phi at the next slab's first node as the last solve (dots_slab_stage 3) left it. */
int dots_norm_square(dots_ctx *ctx, int array_id, int part, double *out);

/* ---- standalone operators (rows a4-a6 of SURVEY.md section 8a), host in / host out, for tests */
enum dots_operator {
    DOTS_OP_GRAD_TIME = 0,       /* (T+1,V)      -> (T,V)        solver_socp.py:881-884 */
    DOTS_OP_DIV_TIME,            /* (T,V)        -> (T+1,V)      :886-896 */
    DOTS_OP_GRAD_SPACE,          /* (T+1,V)      -> (T+1,F,3)    :898-907 */
    DOTS_OP_DIV_SPACE,           /* (T+1,F,3)    -> (T+1,V)      :909-921 */
    DOTS_OP_DECOUPLE,            /* (T+1,F,3)    -> (T,2,3,F,3)  :923-942 (scale) */
    DOTS_OP_DECOUPLE_ADJOINT,    /* (T,2,3,F,3)  -> (T+1,F,3)    :944-959 (scale) */
    DOTS_OP_TIME_AVG_ADJOINT,    /* (T,V)        -> (T+1,V)      :961-974 */
    DOTS_OP_LAPLACIAN_APPLY      /* (T+1,V)      -> (T+1,V)   x -> K x, the operator step 1 inverts (sign: K = -Laplacian + eps M) */
};
int dots_apply_operator(dots_ctx *ctx, int op, double scale, const double *in, int64_t n_in, double *out, int64_t n_out);

/* ---- multigrid preconditioner of the modal PCG (optional; Jacobi is used without it) ----------
 * Smoothed-aggregation hierarchy of the surface stiffness matrix, built on the host
 * (dots_socp_amd/multigrid.py).  Level l holds K_l and M_l on one CSR pattern, the prolongation P_l
 * (n_l x n_{l+1}) and its transpose; level 0 is the context's own K and vertex mass (pass NULL for
 * its rowptr/col/val_k/val_m).  coarse_inverse is (K_L + (sigma_a + eps) M_L)^-1 for every mode a,
 * stored [n_L][n_L][n_cols] (pseudo-inverse for a singular mode): it depends on eps, so call
 * dots_mg_setup again when eps changes. */
typedef struct dots_mg_level {
    int32_t n;                 /* rows of this level */
    int32_t nnz;               /* entries of the K/M pattern (0 for level 0) */
    const int32_t *rowptr;     /* [n+1] */
    const int32_t *col;        /* [nnz] */
    const double *val_k;       /* [nnz] */
    const double *val_m;       /* [nnz] */
    const double *diag_k;      /* [n] (level 0: may be NULL, taken from the context) */
    const double *diag_m;      /* [n] */
    int32_t n_coarse;          /* rows of the next level (0 on the coarsest) */
    int32_t p_nnz;
    const int32_t *p_rowptr;   /* [n+1]        P: n x n_coarse */
    const int32_t *p_col;
    const double *p_val;
    const int32_t *r_rowptr;   /* [n_coarse+1] R = P^T */
    const int32_t *r_col;
    const double *r_val;
    int32_t ap_nnz;            /* K P and M P (n x n_coarse) on one pattern: the post-smoothing kernel applies A P */
    int32_t reserved;
    const int32_t *ap_rowptr;  /* [n+1] */
    const int32_t *ap_col;
    const double *ap_val_k;
    const double *ap_val_m;
    const double *ap_val_p;    /* P itself on the pattern of A P (zero where P has no entry) */
} dots_mg_level;

typedef struct dots_mg_desc {
    int32_t n_levels;          /* >= 2 */
    int32_t n_cols;            /* modes the inverse is given for (= T+1 on one GPU) */
    double omega;              /* Jacobi damping of the smoother */
    const dots_mg_level *levels;
    const double *coarse_inverse;
} dots_mg_desc;

int dots_mg_setup(dots_ctx *ctx, const dots_mg_desc *desc);
int dots_mg_enable(dots_ctx *ctx, int on);   /* switch between multigrid (1) and Jacobi (0) preconditioning */

/* ---- direct solve of the modal problems (replaces the T+1 SuperLU factorisations of
 * laplacian_inverse_socp.py:40-61 and their per-iteration triangular solves, :46-60) ----------------
 * Multifrontal Cholesky factor on one nested-dissection tree shared by all modes, built on the host
 * (dots_socp_amd/frontal.py).  Nodes are numbered children-before-parents.  Node p eliminates n[p]
 * separator vertices and touches b[p] boundary vertices of its ancestors; front_idx lists them
 * (separator first) from ioff[p]; its dense block F_p = [L_pp^-1 ; A_bs A_ss^-1], (n+b) x n per mode,
 * starts at row foff[p] of `values` ([n_entries][pitch], mode fastest).  pull0/pull1 (parallel to
 * front_idx) give the position of a front row in the boundary of child 0 / 1 (or -1).  level_nodes
 * lists the nodes by height, level_ptr delimits the heights.  With a factor installed and enabled,
 * step 1 runs the two triangular sweeps instead of the PCG. */
typedef struct dots_front_desc {
    int32_t n_nodes;
    int32_t n_levels;
    int32_t n_modes;             /* modes the factor is given for (= T+1 on one GPU, slab_count on a time slab) */
    int32_t pitch;               /* doubles per entry of values; must equal the context's mode pitch */
    int64_t n_front_rows;        /* length of front_idx, pull0, pull1 */
    int64_t n_entries;           /* rows of values = sum over nodes of (n+b)*n */
    int64_t update_rows;         /* sum of b */
    const int32_t *node_n;
    const int32_t *node_b;
    const int64_t *node_foff;
    const int64_t *node_ioff;
    const int64_t *node_uoff;    /* first row of node p's update vector */
    const int32_t *node_child;   /* [n_nodes][2], -1 = none */
    const int32_t *front_idx;    /* device vertex numbering */
    const int32_t *pull0;
    const int32_t *pull1;
    const int32_t *level_ptr;    /* [n_levels+1] */
    const int32_t *level_nodes;  /* [n_nodes] */
    const double *values;        /* the factor, or NULL: factorise K + (sigma_a + eps) M on the device (eps from dots_params) */
    const int32_t *grounded;     /* [n_modes] 1: the mode's operator is singular (its last root pivot is grounded);
                                    read when values == NULL */
    const int32_t *band_ptr;     /* [n_bands+1] or NULL: tree heights [band_ptr[k], band_ptr[k+1]) are handled by ONE launch
                                    per sweep (0 = band_ptr[0] < ... < band_ptr[n_bands] = n_levels, at most 4 heights per
                                    band): the nodes of a band that hang together are merged into one block, computed on
                                    the device from the factor (csrc/kernels_front.hip).  NULL: one launch per height */
    int32_t n_bands;
    int32_t top_inverse;         /* 1: the nodes of the top band (they have no boundary rows) store the explicit inverse
                                    S^-1 = L'^-T L'^-1 of their merged block: the forward launch of that band writes the
                                    solution itself and the backward sweep starts one band lower (one launch less per solve;
                                    pays on small meshes, where the top band is a few hundred rows) */
} dots_front_desc;

/* DOTS_ERR_MEMORY (nothing allocated, the context stays usable with the PCG): the factor, the copy of the fronts and the Schur
 * complements the numeric factorisation needs beside it, and the per-corner sums of DOTS_STEP_CARRY exceed the free device memory
 * (hipMemGetInfo; DOTS_MEM_BUDGET=<MB> overrides what counts as available).  dots_last_error() names the sizes. */
int dots_front_setup(dots_ctx *ctx, const dots_front_desc *desc);

/* Host-side helpers for dots_front_desc (no device work; the Python reference implementations are in
 * dots_socp_amd/frontal.py).
 * dots_tree_build: geometric nested dissection of the graph (CSR pattern of K, 0-based) of `n_vertices` points
 * `xyz` [V][3]: subsets are cut at the median of their longest bounding-box side (principal axis above 512
 * vertices), the separator is the smaller one-sided vertex boundary of the cut, subsets of <= `leaf` vertices
 * become leaves.  Nodes are numbered children first; copy out with dots_tree_copy: order [V] (vertex eliminated
 * at position k), sep_ptr [n+1], child [n][2], parent [n], height [n].
 * dots_symbolic_build: boundary sets of the tree on the graph given in the numbering `order` refers to:
 * node_b [n], and per front row (separator rows first; dots_symbolic_front_rows of them) front_idx, pull0, pull1. */
/* dots_assemble: the operator assembly of utils/surface_pre_computations_socp.py:11-132 / solver_socp.py:102-113 for the mesh
 * (xyz [V][3], tri [F][3]) on the host: area [F], hat gradients [F][3][3], vertex masses [V], vertex -> corner lists
 * (cptr [V+1], cidx [3F], corners of a vertex ordered by the reference's corner index k F + f) and K = G^T diag(area) G as a
 * sorted CSR (rowptr [V+1], col / val [dots_assemble_nnz]): the arrays dots_problem_desc takes. */
typedef struct dots_mesh_ops dots_mesh_ops;
int dots_assemble(int32_t n_vertices, int32_t n_triangles, const double *xyz, const int32_t *tri, dots_mesh_ops **out);
int64_t dots_assemble_nnz(const dots_mesh_ops *ops);
int dots_assemble_copy(const dots_mesh_ops *ops, double *area, double *hat, double *mass, int32_t *cptr, int32_t *cidx, int32_t *rowptr, int32_t *col, double *val);
void dots_assemble_free(dots_mesh_ops *ops);
/* dots_patch_order: recursive coordinate bisection of the points `xyz` [V][3] (longest side of the bounding box, cut at a
 * multiple of `unit`, leaves of <= unit points, siblings adjacent): order [V]. */
int dots_patch_order(int32_t n_vertices, const double *xyz, int32_t unit, int32_t *order);
typedef struct dots_tree dots_tree;
typedef struct dots_symbolic dots_symbolic;
int dots_tree_build(int32_t n_vertices, const int32_t *indptr, const int32_t *indices, const double *xyz, int32_t leaf, dots_tree **out);
int64_t dots_tree_nodes(const dots_tree *tree);
int dots_tree_copy(const dots_tree *tree, int64_t *order, int64_t *sep_ptr, int32_t *child, int32_t *parent, int32_t *height);
void dots_tree_free(dots_tree *tree);
int dots_symbolic_build(int32_t n_vertices, const int32_t *indptr, const int32_t *indices, int64_t n_nodes, const int64_t *order,
                        const int64_t *sep_ptr, const int32_t *child, dots_symbolic **out);
int64_t dots_symbolic_front_rows(const dots_symbolic *sym);
int dots_symbolic_copy(const dots_symbolic *sym, int32_t *node_b, int32_t *front_idx, int32_t *pull0, int32_t *pull1);
void dots_symbolic_free(dots_symbolic *sym);
int dots_front_enable(dots_ctx *ctx, int on);
/* launches one direct solve takes: 2 x bands of tree heights (one per band and sweep; a band is one height unless
 * dots_front_desc.band_ptr merges heights), minus one with dots_front_desc.top_inverse */
int dots_front_launches(dots_ctx *ctx);
/* out[4]: factor bytes one solve reads with one block per tree node (both sweeps: the algorithmic bytes of the solve),
 * the bytes it reads as installed (merged bands store more), tree heights, bands */
int dots_front_info(dots_ctx *ctx, double *out);
/* the mode pitch `values` must be laid out with (power of two >= the context's mode count, >= 8) */
int dots_front_pitch(dots_ctx *ctx);

/* ---- measurement ------------------------------------------------------------------------- */
/* Launch the dominant kernel (the PCG operator application) `reps` times on the context's stream
 * between two hipEvents and return the average milliseconds per launch and the algorithmic bytes
 * one launch moves (DESIGN.md section "roofline"). */
/* which: 0 PCG operator application, 1 PCG vector update, 2 one multigrid V-cycle, 3 both sweeps of the direct
 * solve, 4 one calibration launch (k_calib_stream: reads and writes *bytes_per_launch bytes each, 8 B per lane)
 * for the PMC traffic counters */
int dots_bench_kernel(dots_ctx *ctx, int which, int reps, double *ms_per_launch, double *bytes_per_launch);

/* diagnostics: which = 0 KKT read-backs whose mailbox sequence number never arrived (the sums were then copied from the
 * device scalars instead: never stale), 1 mailbox hand-overs so far, 2 penalty decisions taken ahead of the host (dots_penalty_ahead), 3 those the
 * caller's own decision then confirmed, 4 tree leaves the sweeps of the direct solve handle as explicit local inverses
 * (0: the leaves' band runs in the band kernels), 5 whether those leaves take their coupling from per-row
 * records (1) or from the CSR of K (0: a row holds more entries than a record, or DOTS_FRONT_LEAFINV=2), 6 whether steps 2+3 stream
 * beta_mid with the non-temporal hint (decided by dots_front_setup from the sizes of factor and state); -1 for an unknown counter */
int64_t dots_debug_counter(dots_ctx *ctx, int which);

/* device memory in use by the context, bytes */
int64_t dots_device_bytes(dots_ctx *ctx);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* DOTS_SOCP_HIP_H */
