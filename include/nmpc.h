/*
 * nmpc.h -- C-ABI of the MI355X (gfx950) batched NMPC solve path.
 *
 * This is the drop-in boundary for the hot path of Atarilab/iterative_learning_NMPC:
 * what the reference reaches through
 *     QuadrupedAcadosSolver.init / .solve            mpc_controller/utils/solver.py:355-429
 *       -> AcadosSolverHelper.solve [contact_tamp, un-vendored] -> acados C-ABI
 *          (<name>_acados_create / _solve / _free, ocp_nlp_*_set / _get; int status)
 * is replaced by the entry points below.  Conventions:
 *   - plain C types only; every pointer marked `dev` is a HIP device pointer (e.g. a torch
 *     tensor's data_ptr()); `host` pointers are ordinary host memory; the caller owns all buffers;
 *   - the library allocates its workspace in nmpc_create() only; no allocation, no
 *     synchronisation in any *_batch call (they are stream-ordered and graph-capturable);
 *   - every function returns 0 on success, a negative NMPC_E_* code on failure;
 *     nmpc_last_error() gives the text.  Per-problem solver status goes to `status[B]`
 *     with acados' codes (0 ok, 1 NaN, 2 max-iter, 3 min-step, 4 QP failure), replacing
 *     the exception -> `diverged` convention of mpc_controller/mpc.py:562-569;
 *   - thread-safety: a handle is not re-entrant (one solve in flight per handle, as the
 *     reference's single worker thread, mpc.py:164,516); distinct handles are independent.
 *     ctypes releases the GIL during calls, so a worker thread may drive a handle.
 *   - layouts are batch-major, stage-major, feature-minor, fp32:
 *        x0[B][nx]  yref[B][N][ny] | [B][ny]  yref_e[B][ny_e]  params[B][N+1][np]
 *        X[B][N+1][nx]  U[B][N][nu]  status[B]  stats[B][4]
 *     (the reference keeps [dim][node] numpy views, solver.py:88-92,169; the Python host
 *      mirror transposes at its boundary).
 */
#ifndef NMPC_H
#define NMPC_H
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- models (mathematics in DESIGN.md section 3) ----
 * Hard limits of the kernels: models 0 and 1 live in one 16x16 tile per matrix (nx, nu <= 12 in the slot layout;
 * nmpc_riccati_batch: nx <= 15, nu <= 16) and N <= 256 (four stages per lane); model 2 in 3x3 / 2x2 tiles with
 * N <= 64 (one stage per lane); rollouts N <= 128.  Violations return NMPC_E_ARG.
 * Environment, read by nmpc_create: NMPC_QP_VARIANT=resident|lean forces one of the two variants of the centroidal QP
 * kernel (by default chosen per call from batch size and horizon; results are bit-identical). */
#define NMPC_MODEL_DOUBLE_INTEGRATOR 0 /* nx 4  nu 2  np 0  ng 4  (BASELINE config 1) */
#define NMPC_MODEL_CENTROIDAL        1 /* nx 12 nu 12 np 16 ng 16 (BASELINE config 2) */
#define NMPC_MODEL_WHOLEBODY         2 /* nx 42 nu 30 np 20 ng 16, ny 90, ny_e 66 (BASELINE configs[2]):
                                        * the problem the reference solves, x = [q18, v18, h6], u = [a18, f12]
                                        * (solver.py:88-92,405-418); per-node params [active(4), peak(4),
                                        * plane_point(4x3)] (solver.py:212-252); cost residuals in the order
                                        * base(12) joint(24) acc(12) swing(4) f_reg(12) contact(12) consist(6),
                                        * terminal base joint swing contact consist (dynamics.py:121-134,
                                        * solver.py:108-141), then the foot-placement rows pos(8) on both;
                                        * model declared in DESIGN.md 3.2.  N <= 64, fp32.  Device arrays of this
                                        * model must be 8 B aligned, params 16 B (any allocator's are; a view that
                                        * starts at a problem boundary keeps it) -- checked, NMPC_E_ARG otherwise. */

/* model parameter vector, nmpc_set_model_params(): */
#define NMPC_MP_DT    0 /* node spacing T/N                                  */
#define NMPC_MP_MASS  1
#define NMPC_MP_IXX   2
#define NMPC_MP_IYY   3
#define NMPC_MP_IZZ   4
#define NMPC_MP_GZ    5 /* gravity along z (negative)                        */
#define NMPC_MP_MU    6 /* friction coefficient (0.8, solver.py:38)          */
#define NMPC_MP_UMAX  7 /* double integrator input box, <= 0 disables        */
#define NMPC_MP_PGAIN 8 /* whole-body: Baumgarte gain of the stance constraint (W_foot_pos_constr_stab, 50) */
#define NMPC_MP_HIPX  9 /* whole-body leg geometry: hip offset x, y; abduction link; thigh; calf            */
#define NMPC_MP_HIPY  10
#define NMPC_MP_LHIP  11
#define NMPC_MP_L1    12
#define NMPC_MP_L2    13
#define NMPC_MP_COUNT 16 /* entries 14, 15 reserved (0) */

/* status codes written to status[B] (acados numbering) */
#define NMPC_STATUS_OK       0
#define NMPC_STATUS_NAN      1
#define NMPC_STATUS_MAXITER  2
#define NMPC_STATUS_MINSTEP  3
#define NMPC_STATUS_QP       4

/* return codes */
#define NMPC_OK            0
#define NMPC_E_ARG        -1 /* bad argument / dimension                      */
#define NMPC_E_HIP        -2 /* HIP runtime error                             */
#define NMPC_E_STATE      -3 /* weights / model parameters not set            */

typedef struct {
    int model_id;  /* NMPC_MODEL_*                                             */
    int N;         /* horizon (number of shooting intervals)                  */
    int B_max;     /* largest batch a *_batch call will be given              */
    int precision; /* 0 = fp32.  Mixed precision (BASELINE configs[4]: bf16 Jacobian,  */
                   /* J'WJ on the bf16 matrix pipe, fp32 Riccati):                     */
                   /* 1 = the Gauss-Newton contraction in bf16 with fp32 accumulation  */
                   /*     -- whole-body: Q~ = Js'Js of the dense residual Jacobian;    */
                   /*     centroidal: the barrier product G'DG | G'v (the only dense   */
                   /*     contraction that model has) --, everything else fp32;        */
                   /* 2 = whole-body only: split bf16, Js = hi + lo, three products;   */
                   /* 3 = whole-body only: three-way split, Js = hi + mid + lo (the 24  */
                   /*     bits of the fp32 Jacobian), six bf16 products, fp32 accumu-   */
                   /*     lation: the recommended configs[4] variant, inside the 1e-5   */
                   /*     bar at the steady-state policy.                               */
                   /* Measured deviations: DESIGN.md 7 (1: ~3e-3, 2: ~1e-4, 3: ~7e-6   */
                   /* after one SQP iteration).                                         */
} nmpc_dims;

/* Dimensions of a model.  Any out pointer may be NULL. */
int nmpc_model_dims(int model_id, int *nx, int *nu, int *np, int *ng);
/* Number of cost residuals of a stage (length of W and of a yref row) and of the terminal node (length of W_e,
 * yref_e): nx + nu and nx for models 0 and 1 (y = [x; u]), 82 and 58 for the whole-body model. */
int nmpc_model_output_dims(int model_id, int *ny, int *ny_e);

/* Replaces <name>_acados_create (+ AcadosSolverHelper.setup, solver.py:68-72). */
int nmpc_create(const nmpc_dims *dims, int device_id, void **handle);
void nmpc_destroy(void *handle);
const char *nmpc_last_error(void *handle); /* handle may be NULL: last create error */
size_t nmpc_workspace_bytes(void *handle);

/* host float[NMPC_MP_COUNT] */
int nmpc_set_model_params(void *handle, const float *mp, int count);

/* Replaces set_cost_weight_constant / set_cost_weight_terminal (solver.py:140-141) and the
 * reg_eps / reg_eps_e constructor arguments (solver.py:53-54).  host W[ny], W_e[ny_e]. */
int nmpc_set_weights(void *handle, const float *W, const float *W_e, float reg, float reg_e);

/* Replaces set_max_iter / set_nlp_tol / set_qp_tol (solver.py:75-79, mpc.py:464-473) and
 * setup(max_qp_iter) (solver.py:71).  max_qp_iter = number of interior-point iterations per
 * SQP iteration (0: inequalities ignored).  nlp_tol <= 0 disables the early exit on
 * max|step| < nlp_tol.  qp_tol is stored for API symmetry; the IPM runs a fixed
 * iteration count.  line_search: 0 full step, 1 backtracking on the l1 merit. */
int nmpc_set_opts(void *handle, int max_sqp_iter, int max_qp_iter, float nlp_tol, float qp_tol,
                  int line_search);

/* Contact patterns of the gait (contact_planner.py:45-118).  The stage sweep has a straight-line
 * body per contact pattern of the four feet; the default kernel carries those of a trot (the two
 * diagonal pairs, four-foot stance, flight) and sends every other pattern through a run-time-mask
 * fallback (same results, ~25 % slower for such stages).  all_patterns = 1 selects the kernel with a
 * static body for all sixteen patterns: full speed for pace, bound, crawl ..., 2.5 % slower for a
 * trot.  The host side picks it from the gait configuration. */
int nmpc_set_contact_patterns(void *handle, int all_patterns);

/* Declared interior-point constants; defaults 10, 0.2, 1, 0.995, 0.1, 1e3. */
int nmpc_set_ipm(void *handle, float mu0, float sigma, float s_min, float gamma, float tau_min,
                 float merit_rho);

/* Replaces QuadrupedAcadosSolver.warm_start_solver (solver.py:290-342): shift the previous
 * primal solution left by `shift` nodes, zero-fill the exposed input tail.  X, U dev in/out. */
int nmpc_shift_warm_start(void *handle, int B, int shift, float *X, float *U, void *stream);

/* Replaces update_solver + solve + parse_sol (solver.py:345-353,396-403): B independent
 * NMPC problems, one per wavefront.  yref_per_stage: 1 = yref[B][N][ny], 0 = yref[B][ny].
 * X/U: in warm start, out solution.  status / stats may be NULL.
 * stats[b] = {cost at the last linearisation, max|step|, step length, SQP iterations}. */
int nmpc_solve_batch(void *handle, int B, const float *x0, const float *yref, int yref_per_stage,
                     const float *yref_e, const float *params, float *X, float *U, int *status,
                     float *stats, void *stream);

/* nmpc_shift_warm_start(shift) followed by nmpc_solve_batch, as ONE operation: the kernels of the
 * first SQP iteration read the previous solution in X, U through the shift's index map, so the
 * warm start of the receding-horizon loop (solver.py:304-322 then :396-403) costs no launch and no
 * pass over memory.  Results are identical to the two calls.  shift = 0 is nmpc_solve_batch. */
int nmpc_shift_solve_batch(void *handle, int B, int shift, const float *x0, const float *yref,
                           int yref_per_stage, const float *yref_e, const float *params, float *X,
                           float *U, int *status, float *stats, void *stream);

/* One Riccati sweep on explicit stage data (the LQ core of the solve), dense row-major dev
 * inputs Q[B][N+1][nx][nx] R[B][N][nu][nu] q[B][N+1][nx] r[B][N][nu] A[B][N][nx][nx]
 * B_[B][N][nx][nu] d[B][N][nx] dx0[B][nx]; outputs dX[B][N+1][nx] dU[B][N][nu] status[B].
 * nx <= 15, nu <= 16.  Uses the handle's workspace: Bsz <= B_max, N == dims.N. */
int nmpc_riccati_batch(void *handle, int Bsz, int nx, int nu, const float *Q, const float *R,
                       const float *q, const float *r, const float *A, const float *B_,
                       const float *d, const float *dx0, float *dX, float *dU, int *status,
                       void *stream);

/* Tracking error of B rollouts against the nominal one
 * (Behavior_Cloning/utils/data_collection_force_perturbation.py:138-156):
 *   err[b][t] = || S[b][t][1:] - S_nom[t][1:] ||_2      (column 0, the gait phase, is skipped)
 * and, if weight != NULL, the sampling weight of Behavior_Cloning/examples/test_train_policy.py:127-134:
 *   weight[b][t] = err > threshold ? ood_weight : 1.
 * Needs no handle (pass NULL) -- stateless. */
int nmpc_tracking_error(void *handle, int B, int T, int ns, const float *S, const float *S_nom,
                        float *err, float *weight, float threshold, float ood_weight, void *stream);

/* Device-resident receding-horizon rollouts of the centroidal model: the reference's simulator-free
 * LocomotionMPC.open_loop (mpc_controller/mpc.py:416-462) for B rollouts from ONE host call -- per
 * replanning step: contact window + base references (mpc.py:210-272, contact_planner.py:121-134),
 * warm-start shift, solve (15 SQP iterations on the first step, mpc.py:464-473), plant = plan,
 * optional base push, reference integration (mpc.py:204-208).  All rollouts share the gait clock. */
typedef struct {
    int n_replans;          /* replanning steps to run                                        */
    int nodes_per_replan;   /* optimisation nodes between two replans (replanning period / dt) */
    int replanning_steps;   /* simulation steps between two replans (mpc.py:113)               */
    int nodes_per_cycle;    /* columns of the contact table                                    */
    int start_node;         /* optimisation node of the first replan                           */
    int first_solve;        /* 1: the first replan is a cold start (no warm start, 15 SQP)     */
    int max_sqp_first;      /* SQP iterations of a cold start (15)                             */
    float nlp_tol_first;    /* its step tolerance (nlp_tol / 10)                               */
    float nlp_tol;          /* steady-state step tolerance (<= 0: none)                        */
    double sim_dt, time_horizon, nom_height, height_offset;
    float push_start, push_duration;   /* s, relative to the first replan; duration 0: no push */
    /* Footsteps (RaiberContactPlanner.get_locations, contact_planner.py:265-322; setup_initial_feet_pos,
     * solver.py:194-210).  0: every foot stays where foot_pos puts it for the whole rollout; 1: a stance foot keeps its
     * position up to its next swing node, every touch-down in the window gets the Raibert target computed from the
     * current base state and command; foot_pos is updated as feet touch down. */
    int footsteps;
    /* 0: one state row per replan; 1: one row per simulation step (replanning_steps rows per replan), the plan
     * up-sampled by cubic Hermite segments as mpc.py:371-414 does (needs nodes_per_replan * dt = replanning_steps * sim_dt) */
    int record_sim_steps;
    float hip_offset[8];               /* Raibert: hip positions in the base frame [4][2], offsets included (mpc.py:80-91) */
    float stance_ratio[4];             /* gait configuration (mpc_gait.py:15-21)                          */
    float nominal_period;
    float foot_size;                   /* z of a planned location (mpc.py:89: 0.0085)                     */
    /* Early termination (the reference's simulator ends a rollout whose robot falls or whose controller diverges, and the
     * data collection discards it and rolls again: DAgger/utils/RolloutMPC.py:424-437,
     * DAgger/example/data_collection_pretrain_omini_vc_policy_1direction_perturbed.py:217-247).  A rollout that raises
     * one of the bits of terminate_mask is frozen from the next replan on: its plant, feet and trajectories stay, its
     * solves are skipped, its remaining rows of S repeat the last recorded one, and failed[b] records the replan that
     * terminated it.  0: nothing terminates (flags are still raised). */
    int terminate_mask;
    float collision_height;            /* base height [m] below which NMPC_ROLLOUT_FLAG_COLLISION is raised [decl: 0.08,
                                        * the trunk on the ground -- the simulator of the reference allows only the feet
                                        * to touch the floor, RolloutMPC.py:404] */
} nmpc_rollout_cfg;
/* bits of failed[b] (sticky over the rollout): the solver's NaN / QP failure, and the reference's unsafe-state
 * predicates on the recorded states (check_unsafe_state_v2, DAgger/utils/Rollout_combined_controller.py:367-431;
 * joint limits have no counterpart in the centroidal plant) */
#define NMPC_ROLLOUT_FLAG_SOLVER        1
#define NMPC_ROLLOUT_FLAG_ROLL          2   /* |roll| > 25 deg                      */
#define NMPC_ROLLOUT_FLAG_PITCH         4   /* |pitch| > 25 deg                     */
#define NMPC_ROLLOUT_FLAG_HEIGHT        8   /* base height outside [0.18, 0.45] m   */
#define NMPC_ROLLOUT_FLAG_VEL_TRACKING 16   /* |v_xy - v_des_xy| > 0.10 m/s         */
#define NMPC_ROLLOUT_FLAG_COLLISION    32   /* base height < cfg.collision_height: the robot lies on the ground [decl] */
#define NMPC_ROLLOUT_FLAG_MASK       0xFF   /* the flag bits of failed[b] ...                                          */
#define NMPC_ROLLOUT_TERM_SHIFT         8   /* ... and above them 1 + the index of the replan that terminated the rollout (0: ran to the end) */
/* gait: dev int8 [4][nodes_per_cycle]; x: dev [B][12] in initial / out final state; v_des, w_des: dev
 * double [B][3] (commands are kept in fp64 like the reference's, so the integrated reference matches); ref_state: dev double [B][12] in/out (the controller's integrated base reference);
 * foot_pos: dev [B][4][3] in/out (out only with cfg.footsteps); push_force: dev [B][3] or NULL; phase: host
 * float[n_replans] recorded gait phase of the per-replan rows; X, U: dev trajectories in/out (warm start of the next
 * call); S: dev [B][n_rows][19], n_rows = n_replans (x replanning_steps with cfg.record_sim_steps), recorded
 * state rows [phase, rdot(3), body rates(3), z, yaw, pitch, roll, base_wrt_feet(8)] -- the sub-vector of the
 * reference's 44-slot row [phase, v(18), q[2:](17), base_wrt_feet(8)] (DAgger/utils/RolloutMPC.py:221) that the
 * centroidal plant has; failed: dev int [B], NMPC_ROLLOUT_FLAG_* bits, sticky (caller zeroes it).
 * Limits: B <= B_max, N <= 128 (contact window in the LDS), model NMPC_MODEL_CENTROIDAL. */
int nmpc_rollout_batch(void *handle, int B, const nmpc_rollout_cfg *cfg, const signed char *gait, float *x,
                       const double *v_des, const double *w_des, double *ref_state, float *foot_pos,
                       const float *push_force, const float *phase, float *X, float *U, float *S,
                       int *status, int *failed, void *stream);

#define NMPC_ROLLOUT_FLAG_JOINT_LIMIT   64   /* whole-body rollouts: a joint outside hip +-70, thigh [25, 115], knee [-155, -60] deg
                                             * (check_unsafe_state_v2, Rollout_combined_controller.py:386-409) */

/* Device-resident rollouts of the WHOLE-BODY model: the reference's own problem through its own loop,
 * LocomotionMPC.open_loop (mpc_controller/mpc.py:416-462), for B rollouts from ONE host call.  Per replanning step:
 * what `optimize` hands to `solver.init` (mpc.py:325-366; solver.py:153-252,355-394 -- contact / peak windows, base
 * references, joint / swing / force references, x0 with its momentum slots, plane points with the stance feet anchored
 * by forward kinematics), the solve with the warm-start shift folded in (first one: 15 SQP iterations from the zero
 * guess, solver.py:386-388), then the plan up-sampled by cubic Hermite segments (mpc.py:388-414) and followed as the
 * plant for `replanning_steps` simulation steps.  All rollouts share the gait clock. */
typedef struct {
    int n_replans;            /* replanning steps to run                                                   */
    int replanning_steps;     /* simulation steps between two replans (mpc.py:113)                         */
    int nodes_per_cycle;      /* columns of the contact / peak tables                                      */
    int first_solve;          /* 1: the first replan is the reference's first solve                        */
    int last_node;            /* first_solve = 0: node of the previous solve (warm-start shift = node - last) */
    int max_sqp_first;        /* 15                                                                        */
    float nlp_tol_first;      /* nlp_tol / 10                                                              */
    float nlp_tol;
    double sim_dt, time_horizon, nom_height, height_offset;
    float step_height;        /* swing-height reference (mpc_gait.py:15-21)                                */
    float push_start, push_duration;
    int record_sim_steps;     /* 0: one row per replan (the state it starts from); 1: one per simulation step */
    int force_reference_gravity;   /* 0: forces regularised to zero (solver.py:128-130); 1: to the stance feet's weight share [decl] */
    float nominal_period;     /* gait period, for the recorded phase                                       */
    int terminate_mask;       /* as nmpc_rollout_cfg                                                       */
    float collision_height;
} nmpc_wb_rollout_cfg;
/* gait, peaks: dev int8 [4][nodes_per_cycle] (contact_planner.py:45-149); nodes: HOST int[n_replans], the optimisation node
 * of each replan (the reference advances it on a float clock, mpc.py:171-186: the host mirror reproduces that clock);
 * q, v: dev [B][18] in initial / out final plant state, Euler layout (q = r, yaw, pitch, roll, joints; v = qdot);
 * v_des, w_des, ref_state as nmpc_rollout_batch; joint_ref: dev [12]; push_force: dev [B][3] or NULL;
 * X [B][N+1][42], U [B][N][30]: dev trajectories in/out; S: dev [B][n_rows][44], rows in the reference's layout
 * [phase, v_mj(18), q_mj[2:](17) = z, quaternion wxyz, joints, base_wrt_feet(8)] (DAgger/utils/RolloutMPC.py:221,
 * dynamics.py:75-98); failed: dev int [B] as nmpc_rollout_batch, plus NMPC_ROLLOUT_FLAG_JOINT_LIMIT.
 * Needs a handle of NMPC_MODEL_WHOLEBODY with line_search = 0; B <= B_max. */
int nmpc_wb_rollout_batch(void *handle, int B, const nmpc_wb_rollout_cfg *cfg, const signed char *gait,
                          const signed char *peaks, const int *nodes, float *q, float *v, const double *v_des,
                          const double *w_des, double *ref_state, const float *joint_ref, const float *push_force,
                          float *X, float *U, float *S, int *status, int *failed, void *stream);

/* Problems to leave out of the following *_batch solves of this handle: flags dev int[B_max] (or NULL: none); a problem
 * with flags[b] & mask != 0 is skipped by every kernel -- its X, U, status, stats stay as they are and it costs no
 * time.  The flags are read when the kernels run (stream order), so the caller may update them between calls without
 * calling this again.  nmpc_rollout_batch uses its own failed[] / terminate_mask for the duration of the call. */
int nmpc_set_skip(void *handle, const int *flags, int mask);

/* Test hook: copy one stage tile of problem b out of the workspace after a solve.
 * which: 0 = A~ = [A d; 0 1], 1 = B~, 2 = K~ = [K kff], 3 = A~ + B~K~ (last sweep).
 * out_host: float[256], the logical 16x16 tile row-major, zero padded.  Synchronises the device. */
int nmpc_debug_read_tile(void *handle, int b, int k, int which, float *out_host);

/* Test hooks of the whole-body kernels: raw floats of problem b's workspace, and the float offsets of its
 * parts for horizon N: out8 = {records, Js images, Q~ images, K~ images, stage arrays, stride, NS, REC}. */
int nmpc_debug_read_workspace(void *handle, int b, size_t offset, size_t count, float *out_host);
int nmpc_debug_wb_layout(int N, size_t *out8);

/* Diagnostic builds (-DNMPC_STAMPS) write cycle counts to dev float[B_max][16]: 8 phases
 * (linearise, IPM update+coefficients, backward, forward, last IPM update, step+write-back, -, -)
 * and 8 segments of the backward stage (tools/phase_shares.py).
 * Production builds ignore the buffer.  NULL detaches it. */
int nmpc_debug_set_buffer(void *handle, float *dev_buffer);

#ifdef __cplusplus
}
#endif
#endif /* NMPC_H */
