/* mpmpc -- C ABI of the MI355X-native batched LTV-MPC QP path.
 *
 * The reference (matssteinweg/Multi-Purpose-MPC) has no FFI layer: its hot path is the Python
 * method MPC.get_control() (src/MPC.py:161-222), which rebuilds the horizon-stacked QP in
 * MPC._init_problem() (src/MPC.py:61-159) and hands it to the third-party OSQP solver
 * (src/MPC.py:158-159,183).  This header is the boundary a maintainer binds with ctypes to move
 * that path onto the GPU for B independent controller instances per call.
 *
 * Conventions
 *   - every array is caller-owned, C-contiguous HOST memory (numpy); the library owns all device
 *     memory behind the opaque handle; one handle = one device = one stream; calls on one handle
 *     are not re-entrant; every call that returns data is synchronous.
 *   - functions return 0 on success, <0 on error; mpmpc_last_error() gives the thread-local text.
 *   - all arithmetic is IEEE float64; nx = 3 states (e_y, e_psi, t), nu = 2 inputs (v, kappa).
 *   - decision vector z = [x_0..x_N (3 each), u_0..u_{N-1} (2 each)], n = 5N+3, as in
 *     src/MPC.py:128-147; constraint rows [3(N+1) dynamics ; 3(N+1) state boxes ; 2N input boxes].
 *   - there is NO CPU fallback: without a HIP device every compute entry point fails.
 */
#ifndef MPMPC_H
#define MPMPC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPMPC_NX 3
#define MPMPC_NU 2
#define MPMPC_MAX_HORIZON 255
#define MPMPC_NUM_FIELDS 27 /* doubles per (instance, stage) of the stage-blocked QP, see below */

/* per-instance status, OSQP's numbering (drives the fallback branch of src/MPC.py:208-216) */
#define MPMPC_SOLVED 1
#define MPMPC_SOLVED_INACCURATE 2
#define MPMPC_MAX_ITER_REACHED (-2)
#define MPMPC_PRIMAL_INFEASIBLE (-3) /* default settings: with a Farkas ray in y; an EMPTY box (lower bound above upper bound,
                                        e.g. the speed cap of src/MPC.py:111-113 below umin[0]) is reported as -3 with a
                                        zero ray - stock OSQP refuses such data at setup */
#define MPMPC_DUAL_INFEASIBLE (-4)
#define MPMPC_UNSOLVED (-10) /* no verdict: the iterate is not finite (NaN / Inf in the inputs) */

/* error codes */
#define MPMPC_OK 0
#define MPMPC_E_ARG (-1)
#define MPMPC_E_HIP (-2)
#define MPMPC_E_STATE (-3)

typedef struct mpmpc_handle_s* mpmpc_handle;

/* Controller constants: what MPC.__init__ stores (src/MPC.py:15-59) plus the model's wheelbase
 * (src/spatial_bicycle_models.py:130) and the path's `circular` flag (src/reference_path.py:96).
 * Q, R, QN hold the DIAGONALS of the reference's weight matrices, Q_offdiag / R_offdiag / QN_offdiag their off-diagonal
 * entries (symmetric; all zero for what src/simulation.py:101-103 builds).  The reference puts the WHOLE Q, R, QN into the
 * Hessian (src/MPC.py:150) while its cost vector uses only diag(Q), diag(R) but the whole QN (src/MPC.py:153-155) - a quirk
 * that is reproduced: with non-diagonal Q or R the minimiser is not the tracking reference even without constraints.
 * A configuration with any off-diagonal weight, like one with bounds on e_psi / t or a cost on t, runs the general kernels,
 * one instance per wavefront (dense 3 x 3 / 2 x 2 stage Hessian blocks); diagonal weights run the reduced-native kernels
 * unchanged.  Q, R, QN must be positive semidefinite.
 * HORIZON: 3 <= N <= MPMPC_MAX_HORIZON (one lane per stage: a wavefront up to N = 63, a workgroup of 2 / 4 wavefronts whose
 * stages talk through LDS beyond that - several times slower per solve; the reference has no
 * upper limit). */
typedef struct {
  int32_t N;          /* horizon, 3 <= N <= MPMPC_MAX_HORIZON (the kappa_pred quirk of MPC.py:86 needs N >= 3); N <= 63: one wavefront
                         (or a part of one) per instance; 64 .. 127 / 128 .. 255: a workgroup of 2 / 4 wavefronts per instance,
                         general kernel (csrc/lane_gpu.hpp: LaneBlock) */
  int32_t max_batch;  /* largest B of any later call */
  int32_t device;     /* HIP device ordinal */
  int32_t circular;   /* ReferencePath.circular */
  double Q[3], R[2], QN[3];
  double xmin[3], xmax[3]; /* StateConstraints 'xmin','xmax' (may be +-inf) */
  double umin[2], umax[2]; /* InputConstraints 'umin','umax' in (v, kappa) */
  double ay_max;           /* MPC.ay_max */
  double wheelbase;        /* model.length */
  double QN_offdiag[3];    /* QN[0][1], QN[0][2], QN[1][2] (= their transposes); all zero for the reference's weights */
  double Q_offdiag[3];     /* Q[0][1], Q[0][2], Q[1][2]: enter the Hessian only (src/MPC.py:150), not the cost vector (153) */
  double R_offdiag[1];     /* R[0][1]: likewise (src/MPC.py:150,155) */
} mpmpc_config;

/* Solver settings: OSQP 0.6.x names and defaults for the ADMM stage (what
 * osqp.OSQP().setup(..., verbose=False) of src/MPC.py:159 runs with), plus the certified polish. */
typedef struct {
  double rho, sigma, alpha;
  double eps_abs, eps_rel, eps_prim_inf, eps_dual_inf;
  int32_t max_iter, check_termination, scaling;
  int32_t adaptive_rho, adaptive_rho_interval;
  double adaptive_rho_tolerance;
  int32_t polish;   /* 0: ADMM only (stock OSQP behaviour); 2: interior-point refine + active-set polish + certificate */
  int32_t ipm_max_iter;  /* interior-point iterations per attempt (30) */
  double ipm_tol, ipm_reg; /* the interior point stops at residual, complementarity < ipm_tol (1e-8: it only has to identify the
                            active set, which it reads off its last step; a failed active-set attempt continues it to
                            ipm_tol x 1e-4); primal / dual regularisation of its Newton systems (1e-8) */
  double as_delta;       /* regularisation of the active-set KKT solve, removed by as_refine refinement steps (1e-10, 5) */
  int32_t as_refine, as_rounds;   /* ...; primal-dual active-set rounds per attempt (4) */
  double cert_tol;       /* KKT certificate (unscaled problem): primal violation, stationarity, complementarity <= 1e-8 */
  int32_t early_polish; /* > 0: try the polish after this many ADMM iterations; instances it cannot certify
                           run the full ADMM (to max_iter / termination / infeasibility) and are polished
                           again.  0: polish only after ADMM has terminated (OSQP's order). */
  int32_t early_scaling; /* Ruiz passes done before the early polish attempt (0 or >= scaling: all of them).  The
                            remaining scaling - early_scaling passes are done before the full ADMM run, which
                            therefore sees exactly OSQP's `scaling` passes. */
  int32_t phase1;        /* 1: an instance the early polish attempt cannot certify is first tested for infeasibility -
                            least-squares phase 1 (every inequality row softened, nothing else in the cost) by the same
                            interior-point code; its multipliers are a Farkas ray, checked with OSQP's own
                            primal-infeasibility criterion (at phase1_eps) -> MPMPC_PRIMAL_INFEASIBLE after ~5-10
                            interior-point iterations instead of hundreds or thousands of ADMM iterations; z then holds
                            the least-violation point, y the ray, resid[0] the largest bound violation of z.  What it
                            cannot decide runs the full ADMM as before.  0: OSQP's ADMM decides infeasibility. */
  double ipm_diverged;   /* the interior point gives up when mu exceeds this multiple of its smallest value so far
                            (multipliers blowing up: infeasible, phase 1 decides) */
  double phase1_theta;   /* start value of phase 1's slacks and multipliers (row space) */
  double phase1_eps;     /* eps of OSQP's primal-infeasibility test when it is applied to phase 1's ray (default 1e-6).
                            eps_prim_inf = 1e-4 is calibrated for ADMM's slowly converging dual steps; the interior-point
                            ray satisfies |A'y| <= 1e-7 |y|, so the test can be sharper: an instance whose corridor
                            cannot be met by a few tenths of a millimetre is still proved infeasible instead of being
                            handed to the ADMM iteration (which calls it "solved" at eps = 1e-3; status 2). */
  int32_t reduce;        /* 1 (default): when the time state carries neither cost nor bound (Q[2] = QN[2] = 0, QN_offdiag
                            without t, xmin[1..2] = -inf, xmax[1..2] = +inf, R[0] > 0 - the reference's own tracking weights,
                            src/simulation.py:101-111) the certified polish and phase 1 solve the REDUCED problem: t enters
                            no other state's dynamics and the speed v drives t alone, so the QP separates into
                            v_k = clip(v_ref_k) in closed form, the roll-forward of t, and the QP in (e_y, e_psi, kappa) with
                            2 x 2 blocks - the same optimum (the KKT certificate is evaluated on the FULL problem) for about
                            half the arithmetic.  0: always the full 3-state polish.  The ADMM stage always runs on the
                            full problem (it reproduces OSQP's iterates). */
  double ipm_start_slack; /* the polish attempt that follows the early_polish ADMM iterations (and the retry from phase 1's */
  double ipm_start_mu;    /* point) starts the interior point CENTRED: slacks max(distance to the bound, ipm_start_slack),
                            multipliers ipm_start_mu / slack (row space of the scaled problem) - after one ADMM iteration
                            the multipliers carry no information and small slacks cost 5-7 blocked steps.  ipm_start_mu = 0:
                            warm start from the ADMM multipliers, as the polish after a full ADMM run always does.
                            Defaults 0.1, 0.01. */
  double ipm_start_dual;  /* > 0: the centred start uses mu0 = max(ipm_start_mu, ipm_start_dual x ipm_start_slack x |P x + q|_inf)
                            (scaled problem, start point x): start multipliers commensurate with the dual residual they
                            have to balance (config 3, time-optimal weights: 12.6 -> 11.4 iterations; the stock weights sit on the
                            ipm_start_mu floor).  0: ipm_start_mu alone.  Default 0.2. */
  double as_add_fraction; /* active-set rounds add only the bounds violated by at least this fraction of the round's worst
                            violation (measured on the scaled variable); 0: every violated bound, the plain primal-dual
                            active-set update; the retry after a failed attempt uses at least 0.5.  Default 0.25. */
  int32_t phase1_accept; /* 1 (default): an instance phase 1 proves infeasible by LESS than OSQP's primal termination tolerance
                            (eps_abs + eps_rel max(|Ax|, |z|) at the least-violation point: millimetres at eps = 1e-3) is not
                            reported infeasible - the reference's OSQP call accepts such a problem as "solved" and the
                            reference drives the plan (src/MPC.py:159,183-206).  Its violated boxes are widened to 1.5 times the
                            least violation, the polish solves that problem, and the plan comes back as MPMPC_SOLVED_INACCURATE
                            with the violation in resid[0] (a usable status: no fallback step).  0: every proven infeasibility
                            is reported as MPMPC_PRIMAL_INFEASIBLE however small the margin (batch sweeps that want the verdict).
                            The test is a MODEL of OSQP's, not a run of it: the least-violation point is the limit of OSQP's
                            ADMM iteration (the minimiser of the sum of the squared scaled violations, dynamics rows weighted a
                            thousand times the box rows: Solver::phase1), max(|Ax|, |z|) is taken at that point; what it
                            cannot know is an instance OSQP abandons at max_iter before either of its tests passes (it then
                            returns its iterate: "solved inaccurate"), or one within 0.1 % of the threshold, where OSQP stops
                            an iterate short of its limit.  Validated on the reference's own laps (golden G6s) and
                            on BASELINE's obstacle batches (tests, bench line); mpmpc.stock_settings() runs the restated OSQP
                            itself. */
  int32_t native;        /* 1 (default): where `reduce` applies and the settings are the defaults of the early attempt
                            (early_polish = 1, ipm_start_mu > 0) the batch launches run the REDUCED-NATIVE kernels: a lane
                            never holds the 3-state problem - v in closed form at load time, own Ruiz pass / start / interior
                            point / active-set rounds / KKT certificate on the (e_y, e_psi, kappa) problem, the roll-forward
                            of t at the store - about half the registers of the general kernels, so two wavefronts share a
                            SIMD.  What they cannot certify (infeasible or very hard instances) goes to the general
                            one-instance-per-wave kernel (phase 1, full OSQP run) exactly like the tail of a packed launch.
                            0: the general kernels only.  The closed loop's warm-started launches run the same kernels
                            (template flag WARM).  Weightings with a terminal cost on the time state and none else on it
                            (QN[2] > 0 = Q[2]: BASELINE config 3) have their own reduced-native kernels
                            (csrc/mpmpc_reduced_t.hpp: the speeds stay in the problem, the time cost is one rank-one term;
                            one instance per wave, cold starts). */
  double native_ipm_tol; /* interior-point tolerance of the reduced-native kernels' FIRST attempt (ipm_tol is the general
                            kernels').  The interior point only has to identify the active set - the active-set rounds
                            and the KKT certificate (cert_tol) make the answer - and on the reduced problem it has done
                            so one iteration earlier than at 1e-8 for most instances: measured, config 2 +8 %, configs 4 / 5
                            +2-3 %, B = 65 536 +4 %.  An attempt whose active-set rounds fail is repeated at a hundred
                            times tighter tolerance, twice if need be.  Default 1e-7. */
  int32_t early_start;   /* start of the general kernels' early attempt (early_polish = 1): 1 = OSQP's first iterate (one
                            factorisation + one KKT solve), 0 (default) = x = 0, no OSQP iterate - measured: config 3 takes
                            11.08 instead of 11.43 interior-point iterations from x = 0 and saves the iterate's 11 us per
                            wave.  The reduced-native kernels always start from x = 0.  iters[.][0] = 1 marks the early
                            attempt either way. */
  double phase1_band;    /* phase1_accept: phase 1 leaves at the first iterate whose multipliers are a valid Farkas ray only while
                            that iterate violates a bound by more than phase1_band times OSQP's primal tolerance (eps_abs +
                            eps_rel x the largest finite bound); inside the band it runs to its converged optimum, because the
                            violation of THAT point decides whether the reference's OSQP call would have returned a plan, and
                            it is up to 2.4 times smaller than the early iterate's.  Default 3: measured on config 5 (65 536
                            instances, emulation) the number of instances that take the other branch than the restated stock
                            OSQP is 67 / 34 / 29 / 26 / 26 / 26 at 1.5 / 2 / 2.5 / 3 / 6 / 1000, and config 4 runs 2.1 / 2.7 /
                            3.2 / 4.3 % slower at 2 / 3 / 4 / 6 than at 0 (profiles/r5/branch_agreement.txt). */
} mpmpc_settings;

const char* mpmpc_version(void);
const char* mpmpc_last_error(void);
int mpmpc_device_count(int32_t* count);
void mpmpc_default_settings(mpmpc_settings* s);

/* replaces MPC.__init__ (src/MPC.py:15-59): allocates device buffers for max_batch instances */
int mpmpc_create(const mpmpc_config* cfg, const mpmpc_settings* settings, mpmpc_handle* out);
int mpmpc_destroy(mpmpc_handle h);
int mpmpc_set_settings(mpmpc_handle h, const mpmpc_settings* settings);
/* Lanes of a 64-lane wavefront given to one QP instance by the solve launches: 0 (default) = chosen from the batch
 * size (one instance per wave up to 1024 instances, then the smallest of 64 / 32 / 16 that holds the N + 1 stages);
 * 64 / 32 / 16 force that packing (parity tests and tuning; every packing returns the same answers).  16 with 17 .. 32
 * stages (16 <= N <= 31) selects the layout with TWO stages per lane - four instances per wavefront - for the batch launches
 * of the reference's own weights (cold starts; every other launch keeps one stage per lane). */
int mpmpc_set_packing(mpmpc_handle h, int32_t lanes_per_instance);
/* Which kernel takes the TAIL of a batch launch - the instances the reduced-native kernel could not certify: infeasible,
 * marginally infeasible and very hard ones.  1 (default) = the reduced-native tail kernel first (phase 1 and one more
 * attempt of the certified polish on the (e_y, e_psi, kappa) problem, two wavefronts per SIMD, two instances per wavefront
 * for horizons up to 31, one for horizons 32 .. 63), the general kernel only on what that leaves; 2 = the same with ONE instance per wavefront (the split
 * layout of the general kernel's phase 1: its answers to rounding, 4e-16); 0 = the general kernel on the whole tail (one
 * wavefront per SIMD).  Same statuses in all three; least-violation points and relaxed plans of 1 agree with those of 2 / 0
 * to ~1e-8 (phase 1 converges along another arithmetic path).  Parity tests, A/B timings. */
int mpmpc_set_tail_kernel(mpmpc_handle h, int32_t reduced_native);

/* replaces the per-stage ReferencePath.get_waypoint() / Waypoint.__sub__ reads of
 * src/MPC.py:93-97 (src/reference_path.py:50-57,356-371): per-waypoint kappa, v_ref and the
 * distance to the next waypoint, uploaded once per path. */
int mpmpc_set_path(mpmpc_handle h, int32_t n_wp, const double* kappa, const double* v_ref,
                   const double* ds_next);

/* replaces ReferencePath.update_path_constraints(wp_id+1, N, ...) of src/MPC.py:116-118 for a
 * static map, where it depends on wp_id only (src/reference_path.py:522-648): row w of the
 * [n_wp x n_cols] tables holds ub / lb for the n_cols waypoints after waypoint w. n_cols >= N. */
int mpmpc_set_corridor(mpmpc_handle h, int32_t n_wp, int32_t n_cols, const double* ub,
                       const double* lb);

/* Device-side corridor generation: the three calls below replace, for a map that
 * changes between steps, the host loop of ReferencePath.update_path_constraints + _compute_free_segments
 * + skimage.draw.line_aa + Map.w2m/m2w (src/reference_path.py:466-648, src/map.py:77-101).
 *   mpmpc_set_map           Map.data (int8, 1 free / 0 occupied, row = y), Map.origin, Map.resolution
 *   mpmpc_set_path_geometry per-waypoint x, y, psi and Waypoint.static_border_cells (world coordinates,
 *                           [n_wp*2] each: upper/left border, lower/right border); needs mpmpc_set_path first
 *   mpmpc_build_corridor    update_path_constraints(w+1, n_cols, min_width, safety_margin) for EVERY start
 *                           waypoint w, written into the handle's corridor table (as mpmpc_set_corridor would)
 *                           and optionally copied out ([n_wp x n_cols]; rows whose first horizon waypoint has no
 *                           free segment - the reference raises there - are NaN).  *bad_rows counts those. */
int mpmpc_set_map(mpmpc_handle h, int32_t height, int32_t width, const int8_t* data, double origin_x,
                  double origin_y, double resolution);
int mpmpc_set_path_geometry(mpmpc_handle h, int32_t n_wp, const double* x, const double* y, const double* psi,
                            const double* border_ub, const double* border_lb);
int mpmpc_build_corridor(mpmpc_handle h, int32_t n_cols, double min_width, double safety_margin, double* ub_out,
                         double* lb_out, int32_t* bad_rows);

/* Closed-loop batched rollout on the device: B cars driven through
 * `while car.s < length: u = mpc.get_control(); car.drive(u)` (src/simulation.py:134-140) without a
 * host round trip per step.  Each step = localise + t2s (src/spatial_bicycle_models.py:183-219,256-279)
 * -> K1 -> K2 -> solution use / infeasibility fallback (src/MPC.py:185-220) + drive
 * (src/spatial_bicycle_models.py:221-244).  Needs mpmpc_set_path, mpmpc_set_path_geometry and a corridor
 * table (mpmpc_set_corridor or mpmpc_build_corridor).
 *   init:  Ts = model.Ts; cum_lengths[n_wp] = cumsum(ReferencePath.segment_lengths); s[B] arc lengths;
 *          pose[B*3] = (x, y, psi); cc0[B*2N] previous plans or NULL for zeros (MPC.__init__).
 *   state: any output may be NULL.  alive: 1 running, 0 lap finished (s >= length), -1 ended by the
 *          reference's exit(1) after N-1 consecutive infeasible steps.
 * The rollout keeps its plans, waypoint ids and states in the handle's batch blocks: mpmpc_upload / mpmpc_solve /
 * mpmpc_assemble on the SAME handle overwrite them, after which mpmpc_rollout_step / _state / _set_counters return
 * MPMPC_E_STATE until mpmpc_rollout_init is called again.  (mpmpc_download and mpmpc_build_corridor are fine.) */
int mpmpc_rollout_init(mpmpc_handle h, int32_t B, double Ts, const double* cum_lengths, const double* s,
                       const double* pose, const double* cc0);
int mpmpc_rollout_step(mpmpc_handle h, int32_t B, int32_t n_steps);
/* MPC.infeasibility_counter of every car (src/MPC.py:206,215), to resume a recorded run: values in [0, N-2].
 * mpmpc_rollout_init starts every car at 0. */
int mpmpc_rollout_set_counters(mpmpc_handle h, int32_t B, const int32_t* counter);
/* Warm start of the closed loop: each step first tries active-set rounds from the active set the previous step
 * certified for the same car, shifted by the waypoints it advanced; what they cannot certify goes through the
 * normal path.  enable: 0 off, 1 on, 2 (default) on where it pays - fleets of more than 1024 cars (several cars per
 * wavefront) and of at most 16; in between a step ends with its slowest car, and one car in fourteen misses its
 * guess.  The reference cold-starts every step (src/MPC.py:158). */
int mpmpc_rollout_warm_start(mpmpc_handle h, int32_t enable);
int mpmpc_rollout_state(mpmpc_handle h, int32_t B, double* s, double* pose, double* cc, int32_t* wp_id,
                        double* x0, double* u_last, int32_t* status, int32_t* counter, int32_t* alive);

/* replaces MPC._init_problem (src/MPC.py:61-155) for B instances: LTV linearisation
 * (src/spatial_bicycle_models.py:391-417) around waypoints wp_id+0..N-1, offsets, speed cap from
 * the previous plan cc_prev (src/MPC.py:86-87,111-113), box bounds, references, cost vectors.
 *   wp_id[B], x0[B*3] (= model.spatial_state), cc_prev[B*2N] (= MPC.current_control),
 *   lb/ub [B*N]: per-instance corridor, or both NULL to use the table of mpmpc_set_corridor.
 * qp_out (host, may be NULL) receives the stage-blocked QP [MPMPC_NUM_FIELDS][B][LD] with
 * LD = mpmpc_stage_ld(N); field order: ds, a10, a20, b20, beq[3], lo[5], hi[5], q[5], p[5]
 * (A_k = [[1,ds,0],[a10,1,0],[a20,0,1]], B_k = [[0,0],[0,ds],[b20,0]] couple stage k to k+1;
 * beq = rhs of equality block k; lo/hi/q/p over (e_y,e_psi,t,v,kappa) of stage k). */
int mpmpc_assemble(mpmpc_handle h, int32_t B, const int32_t* wp_id, const double* x0,
                   const double* cc_prev, const double* lb, const double* ub, double* qp_out);
int32_t mpmpc_stage_ld(int32_t N);

/* replaces _init_problem + osqp setup/solve + solution extraction (src/MPC.py:180-194) for B
 * instances.  Outputs (any may be NULL): z[B*(5N+3)] primal solution (dec.x), u0[B*2] = (v_0,
 * delta_0 = arctan(kappa_0 * wheelbase)) as returned by get_control, status[B], iters[B*2]
 * (OSQP's iteration counter: 0 = certified from the closed loop's warm-start guess, 1 = the early attempt alone - the general
 * kernels start it from OSQP's first iterate, the reduced-native kernels from x = 0 without any OSQP iterate -, more = the ADMM
 * loop ran; interior-point iterations), resid[B*2] (primal, dual residual, unscaled),
 * y[B*(8N+6)] multipliers in the reference's row order. */
int mpmpc_solve(mpmpc_handle h, int32_t B, const int32_t* wp_id, const double* x0,
                const double* cc_prev, const double* lb, const double* ub, double* z, double* u0,
                int32_t* status, int32_t* iters, double* resid, double* y);

/* Split form of mpmpc_solve for callers that keep inputs resident in HBM (benchmarks, closed
 * loops): upload once, launch any number of times (asynchronous on the handle's stream),
 * synchronise, download.  mpmpc_upload returns as soon as the caller's buffers may be reused; the
 * transfer itself is ordered before everything launched after it on the handle's stream. */
int mpmpc_upload(mpmpc_handle h, int32_t B, const int32_t* wp_id, const double* x0,
                 const double* cc_prev, const double* lb, const double* ub);
/* The outputs of a resident launch are defined after the next mpmpc_sync / mpmpc_download on the handle (the only ways to
 * read them), and only for the LAST launch before it.  Resident launches are pipelined inside the handle (by default three
 * launch slots - stream, output block - used in turn: mpmpc_set_pipeline): launch k + 1, k + 2 run beside launch k, whose
 * outputs stay untouched until launch k + 3; mpmpc_sync / mpmpc_download and every other call on the handle wait for both.  The
 * library also uses the freedom the first sentence leaves: the second kernel of a launch (the tail: instances the first kernel could not certify, usually none) is not
 * enqueued while the launches whose outcome the host has seen left no tail; a launch that does leave one has it run inside
 * the next mpmpc_sync / mpmpc_download / mpmpc_upload / mpmpc_set_* call, before that call does anything else. */
int mpmpc_solve_resident(mpmpc_handle h, int32_t B);
/* Do resident launches store the multipliers y (46 % of a solve's output bytes)?  Default 1.  mpmpc_solve decides per
 * call (y == NULL: not stored), the closed-loop rollout never stores them.  mpmpc_download refuses a y the last launch
 * did not produce. */
int mpmpc_set_outputs(mpmpc_handle h, int32_t want_y);
/* Resident launches in flight, 1 .. 8: 3 (default) = pipelined as described above (launch k's outputs stay untouched until
 * launch k + depth); 1 = every launch on one stream and one output block, each waiting for the one before (what every other
 * entry point does anyway).  The HIP runtime spreads streams over GPU_MAX_HW_QUEUES hardware queues (default 4) and streams
 * that share a queue serialise: more than 3 launches in flight pay only in a process that exported GPU_MAX_HW_QUEUES=8 (or
 * more) before its first HIP call (bench.py does).  With depth > 1 the batch launches of the handle pack two / four instances into a wavefront from
 * 128 / 256 instances on (instead of 1 024 / 2 048): the handle is after throughput, several launches fill the chip. */
int mpmpc_set_pipeline(mpmpc_handle h, int32_t depth);
int mpmpc_sync(mpmpc_handle h);
int mpmpc_download(mpmpc_handle h, int32_t B, double* z, double* u0, int32_t* status,
                   int32_t* iters, double* resid, double* y);
/* Zero-copy host path.  mpmpc_solve copies the caller's (pageable) arrays into page-locked staging blocks and the results back
 * out of them - at B = 1 024 those host-side copies are two thirds of the call.  A caller that can build its inputs in place
 * and read the results in place asks for the blocks themselves: mpmpc_staging gives pointers into them, laid out for a batch
 * of B (same shapes as mpmpc_solve's arguments; any out-pointer may be NULL; the pointers stay valid until the handle is
 * destroyed, the layout until a call with another B), mpmpc_solve_staged runs upload + solve + download on them and returns
 * when the outputs are there: u0, status, iters, resid always, z if want_z, z and y if want_y.  with_rows = 0: lb / ub are not
 * read, the corridor table (mpmpc_set_corridor) applies.  Handles whose max_batch needs more than 64 MiB per block have no
 * staging blocks (MPMPC_E_STATE).  The blocks ARE the ones mpmpc_upload / mpmpc_download / mpmpc_solve stage through: a call
 * of those overwrites them, and mpmpc_staging itself first waits for an upload that is still leaving the input block. */
int mpmpc_staging(mpmpc_handle h, int32_t B, int32_t** wp_id, double** x0, double** cc_prev, double** lb, double** ub,
                  double** z, double** u0, int32_t** status, int32_t** iters, double** resid, double** y);
int mpmpc_solve_staged(mpmpc_handle h, int32_t B, int32_t with_rows, int32_t want_z, int32_t want_y);
/* The same call in two halves, for a loop that keeps several handles busy: mpmpc_staged_begin enqueues upload, solve and
 * download on the handle's stream and returns; mpmpc_staged_end waits for them (and runs the rarely needed second kernel of the
 * launch if it turns out to be needed).  Between the two the staging block belongs to the device: do not touch it.  Every other
 * call on the handle ends a begun call first. */
int mpmpc_staged_begin(mpmpc_handle h, int32_t B, int32_t with_rows, int32_t want_z, int32_t want_y);
int mpmpc_staged_end(mpmpc_handle h);
/* one resident pass with HIP events around each kernel on the handle's stream (ms); the launch runs alone on the chip */
int mpmpc_solve_resident_timed(mpmpc_handle h, int32_t B, float* ms_assemble, float* ms_solve);
/* n resident launches issued exactly as mpmpc_solve_resident issues them (double-buffered, see there), each between two
 * HIP events on its own stream: ms_each[n] = duration of every launch with the other slot's launch beside it on the chip,
 * *ms_span (may be NULL) = first start to last end.  What rocprofv3 --kernel-trace reports for the same loop. */
int mpmpc_solve_resident_profile(mpmpc_handle h, int32_t B, int32_t n, float* ms_each, float* ms_span);
/* n launches of the stand-alone assembly kernel K1 (what mpmpc_assemble runs, without its download) back to back on the
 * handle's stream, each between two HIP events: ms_each[n].  mpmpc_solve_resident_timed times K1 right after a solve launch,
 * whose dirty output lines K1's writes then push out of the cache; this is K1 on its own. */
int mpmpc_assemble_resident_timed(mpmpc_handle h, int32_t B, int32_t n, float* ms_each);

/* ---- speed profile (K4): replaces ReferencePath.compute_speed_profile, src/reference_path.py:289-354,
 * the reference's second OSQP call site, for B paths of n + 1 waypoints at once (no handle needed):
 *     min 1/2 |v|^2 - vmax' v   s.t.  a_min <= (v[i+1] - v[i]) / (2 li[i]) <= a_max,  v_min <= v[i] <= vmax[i],
 *     vmax[i] = min(v_max, sqrt(ay_max / (|kappa[i]| + eps)))
 * li [B][n]      distance waypoint i -> i+1            (src/reference_path.py:318)
 * kappa [B][n]   curvature of waypoint i               (src/reference_path.py:320)
 * limits [B][5]  a_min, a_max, v_min, v_max, ay_max    (the Constraints dict, src/reference_path.py:301-307)
 * v [B][n]       <- speed_profile (the caller copies v[n-1] to the last waypoint, src/reference_path.py:351-353)
 * status [B]     <- 1 KKT-certified optimum, 2 interior-point iterate (uncertified), -1 bad input
 * iters [B]      <- interior-point iterations (may be NULL)                                              */
int mpmpc_speed_profile(int32_t device, int32_t B, int32_t n, const double* li, const double* kappa,
                        const double* limits, double eps, double* v, int32_t* status, int32_t* iters);

#ifdef __cplusplus
}
#endif
#endif /* MPMPC_H */
