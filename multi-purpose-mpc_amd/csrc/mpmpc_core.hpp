// Lane-generic core of the batched LTV-MPC QP path (float64).
//
// One lane = one horizon stage k of one QP instance; G consecutive lanes = one instance.
// Every per-stage quantity (3x3 / 3x2 dynamics blocks, bounds, iterates, multipliers, factor
// blocks) lives in that lane's registers; neighbouring stages talk through L::up / L::down,
// instance-wide norms through L::gmax / L::gsum.  The backend L is lane_gpu.hpp (gfx950) in
// the shipped library and lane_emu.hpp (lock-step CPU emulation) in the unit tests.
//
// What is computed (reference file:line it replaces):
//   assemble_stage  src/MPC.py:61-155 + src/spatial_bicycle_models.py:391-417 (one stage)
//   Solver::scale   OSQP Ruiz equilibration of the QP of src/MPC.py:159 (setup)
//   Solver::admm    OSQP ADMM iteration, termination and infeasibility tests (src/MPC.py:183)
//   Solver::ipm / active_set / certificate   certified polish
//
// QP of one instance:  min 1/2 w'Pw + q'w,  w = (x_0..x_N, u_0..u_{N-1}),  P diagonal,
//   equality block k:  -x_k + A_{k-1} x_{k-1} + B_{k-1} u_{k-1} = beq_k   (block 0: -x_0 = -x0)
//   boxes lo <= w <= hi.
// A_k = [[1,ds,0],[a10,1,0],[a20,0,1]], B_k = [[0,0],[0,ds],[b20,0]] are stored sparsely.
#pragma once
#include "mpmpc.h"

#ifndef MPMPC_HD
#define MPMPC_HD inline
#endif
#define MPMPC_UNROLL _Pragma("unroll")
#ifndef MPMPC_TICK_BEGIN          // phase clocks of profiling builds (mpmpc_hip.hip, -DMPMPC_PHASE_CLOCK)
#define MPMPC_TICK_BEGIN(i) ((void)0)
#define MPMPC_TICK_END(i) ((void)0)
#define MPMPC_TICK_COUNT(i) ((void)0)
#endif
#ifndef MPMPC_HOST_DEVICE
#define MPMPC_HOST_DEVICE
#endif
#ifndef MPMPC_COUNT_CONTEXT      // instruction census of the emulation (lane_emu.hpp, -DMPMPC_COUNT_OPS)
#define MPMPC_COUNT_CONTEXT(m) ((void)0)
#define MPMPC_SERIAL_BEGIN() ((void)0)
#define MPMPC_SERIAL_END(nsteps) ((void)0)
#endif

namespace mpmpc {

enum Field { F_DS = 0, F_A10 = 1, F_A20 = 2, F_B20 = 3, F_BEQ = 4, F_LO = 7, F_HI = 12, F_Q = 17, F_P = 22 };

constexpr double INFTY = 1e30, MIN_SCALING = 1e-4, MAX_SCALING = 1e4;
constexpr double RHO_MIN = 1e-6, RHO_MAX = 1e6, RHO_TOL = 1e-4, RHO_EQ_FACTOR = 1e3;
constexpr double INF_BOUND = INFTY * MIN_SCALING;   // a scaled bound beyond this is "infinite"
// The dynamics rows of phase 1 are as soft as OSQP's: its ADMM iteration weights the violation of an equality row
// RHO_EQ_FACTOR = 1000 times that of an inequality row, so its limit point on an infeasible QP trades a little dynamics
// violation (~1e-5) for 0.5 % less box violation - enough to decide instances within that of OSQP's threshold.  Phase 1
// minimises  sum_boxes (scaled violation)^2 + RHO_EQ_FACTOR sum_dynamics (scaled residual)^2:  the equality block of its KKT
// systems carries -1 / RHO_EQ_FACTOR, and its least-violation point leaves the dynamics rows by nu / RHO_EQ_FACTOR.
constexpr double P1_EQ_SOFT = 1.0 / RHO_EQ_FACTOR;

// lanes that hold the N + 1 stages of an instance: a power of two, 16 .. 64 inside a wavefront, 128 / 256 = a workgroup of
// 2 / 4 wavefronts (horizons above 63: lane_gpu.hpp, LaneBlock)
MPMPC_HD int stage_ld(int N) { return N + 1 <= 16 ? 16 : (N + 1 <= 32 ? 32 : (N + 1 <= 64 ? 64 : (N + 1 <= 128 ? 128 : 256))); }

// ------------------------------------------------------------------------------------------
// K1 math: the 27 fields of stage k of one instance.
// ------------------------------------------------------------------------------------------
template <class L>
struct StageIn {
  using R = typename L::real;
  using Mk = typename L::mask;
  R kap, v, ds;        // waypoint wp_id+k      (only used where has_u)
  R kap_p, v_p, ds_p;  // waypoint wp_id+k-1    (only used where !first)
  R x0[3];             // spatial state of the instance
  R cc_a, cc_last;     // previous plan entries cc[3+k] and cc[2N-1]  (src/MPC.py:86-87)
  R lbk, ubk;          // corridor at horizon waypoint k (k >= 1): lb[k-1], ub[k-1]
  Mk first, has_u, terminal;   // k == 0, k < N, k == N
};

// put(f, value) receives the fields one by one, in the order they are formed: K2 collects them in registers (assemble_stage),
// K1 stores each at once (assemble_lane) - 27 values never wait for each other in registers there.
// the curvature-dependent speed cap of src/MPC.py:84,111-113 from the previous plan's entries
template <class L>
MPMPC_HD typename L::real speed_cap(const mpmpc_config& c, const typename L::real& cc_a, const typename L::real& cc_last) {
  using R = typename L::real;
  R kp = tan_(cc_a + cc_last) / R(c.wheelbase);
  R vmax = sqrt_(R(c.ay_max) / (abs_(kp) + R(1e-12)));
  R umax0(c.umax[0]);
  return sel(vmax < umax0, vmax, umax0);
}
// hi_v_pre: the speed cap if the caller has formed it already (K1 does, before it gathers anything else: the tangent is the
// register peak of the stage), else nullptr
template <class L, class Put>
MPMPC_HD void assemble_stage_to(const mpmpc_config& c, const StageIn<L>& in, Put&& put, const typename L::real* hi_v_pre = nullptr) {
  using R = typename L::real;
  const R zero(0.0), one(1.0);
  // linearize(v_ref, kappa_ref, delta_s), same operation order as the reference
  R a10 = (-(in.kap * in.kap)) * in.ds;
  R a20 = ((-in.kap) / in.v) * in.ds;
  R b20 = ((-one) / (in.v * in.v)) * in.ds;
  put(F_DS, sel(in.has_u, in.ds, zero));
  put(F_A10, sel(in.has_u, a10, zero));
  put(F_A20, sel(in.has_u, a20, zero));
  put(F_B20, sel(in.has_u, b20, zero));
  // rhs of equality block k: -x0, or uq_{k-1} = B [v, kappa] - f   (src/MPC.py:107-108)
  R b20p = ((-one) / (in.v_p * in.v_p)) * in.ds_p;
  R f2p = (one / in.v_p) * in.ds_p;
  put(F_BEQ + 0, sel(in.first, -in.x0[0], zero));
  put(F_BEQ + 1, sel(in.first, -in.x0[1], in.ds_p * in.kap_p));
  put(F_BEQ + 2, sel(in.first, -in.x0[2], b20p * in.v_p - f2p));
  // state boxes (src/MPC.py:81-82,119-122)
  put(F_LO + 0, sel(in.first, in.x0[0], in.lbk));
  put(F_HI + 0, sel(in.first, in.x0[0], in.ubk));
  put(F_LO + 1, R(c.xmin[1]));
  put(F_HI + 1, R(c.xmax[1]));
  put(F_LO + 2, R(c.xmin[2]));
  put(F_HI + 2, R(c.xmax[2]));
  // input boxes with the curvature-dependent speed cap (src/MPC.py:84,111-113)
  R hi_v = hi_v_pre ? *hi_v_pre : speed_cap<L>(c, in.cc_a, in.cc_last);
  put(F_LO + 3, sel(in.has_u, R(c.umin[0]), R(-INFTY)));
  put(F_HI + 3, sel(in.has_u, hi_v, R(INFTY)));
  put(F_LO + 4, sel(in.has_u, R(c.umin[1]), R(-INFTY)));
  put(F_HI + 4, sel(in.has_u, R(c.umax[1]), R(INFTY)));
  // cost (src/MPC.py:125,150-155): references are the corridor centre for e_y, (v_ref, kappa_ref) for u
  R xr0 = sel(in.first, zero, (in.lbk + in.ubk) / R(2.0));
  MPMPC_UNROLL
  for (int i = 0; i < 3; ++i) {
    R xr = (i == 0) ? xr0 : zero;
    // terminal stage: -QN . xr with xr = (xr0, 0, 0), i.e. minus the first column of QN times xr0 (src/MPC.py:154)
    const double qn_i0 = i == 0 ? c.QN[0] : c.QN_offdiag[i - 1];
    put(F_Q + i, sel(in.terminal, -(R(qn_i0) * xr0), R(-c.Q[i]) * xr));
    put(F_P + i, sel(in.terminal, R(c.QN[i]), R(c.Q[i])));
  }
  put(F_Q + 3, sel(in.has_u, R(-c.R[0]) * in.v, zero));
  put(F_Q + 4, sel(in.has_u, R(-c.R[1]) * in.kap, zero));
  put(F_P + 3, sel(in.has_u, R(c.R[0]), one));
  put(F_P + 4, sel(in.has_u, R(c.R[1]), one));
}
template <class L>
MPMPC_HD void assemble_stage(const mpmpc_config& c, const StageIn<L>& in, typename L::real out[MPMPC_NUM_FIELDS]) {
  assemble_stage_to<L>(c, in, [&](int f, const typename L::real& v) { out[f] = v; });
}

// Per-path tables uploaded once per handle (device pointers in the library, host pointers in
// the emulation): what ReferencePath.get_waypoint / update_path_constraints provide.
struct PathTables {
  const double* kappa;
  const double* v_ref;
  const double* ds_next;
  int n_wp;
  const double* ub_tab;   // [n_wp x n_cols] or null
  const double* lb_tab;
  int n_cols;
};

// One (instance, stage) of K1: gather the waypoint data, build the fields, store them
// stage-blocked as qp[(field * B + inst) * ld + k] (consecutive lanes -> consecutive addresses).
// (assemble_fields: the 27 fields in registers - what the solve kernel goes on with; assemble_lane: K1, stores them)
template <class L>
MPMPC_HD void gather_stage(const mpmpc_config& c, const PathTables& t, int B, const typename L::ival& inst,
                           const typename L::ival& k, const int* wp_id, const double* x0, const double* cc,
                           const double* lb, const double* ub, StageIn<L>& in, bool with_cc = true) {
  using Mk = typename L::mask;
  using I = typename L::ival;
  const int N = c.N;
  Mk ok = (inst < B) & (k >= 0) & (k <= N);        // (K2 keeps lanes before stage 0: lane_offset)
  in.first = (k == 0);
  in.has_u = ok & (k < N);
  in.terminal = (k == N);
  I wp = L::gatheri(wp_id, inst, ok, 0);
  I ik = wp + k, ip = maxi(wp + k - 1, 0);
  if (c.circular) {
    if (t.n_wp > N) {        // 0 <= wp < n_wp (checked at upload) and k <= N < n_wp: one wrap at most - no integer division
      ik = seli(ik >= t.n_wp, ik - t.n_wp, ik);
      ip = seli(ip >= t.n_wp, ip - t.n_wp, ip);
    } else {
      ik = modi(ik, t.n_wp); ip = modi(ip, t.n_wp);
    }
  } else {
    ik = mini(ik, t.n_wp - 1); ip = mini(ip, t.n_wp - 1);
  }
  in.kap = L::gather(t.kappa, ik, ok, 0.0);
  in.v = L::gather(t.v_ref, ik, ok, 1.0);
  in.ds = L::gather(t.ds_next, ik, ok, 0.0);
  in.kap_p = L::gather(t.kappa, ip, ok, 0.0);
  in.v_p = L::gather(t.v_ref, ip, ok, 1.0);
  in.ds_p = L::gather(t.ds_next, ip, ok, 0.0);
  MPMPC_UNROLL
  for (int i = 0; i < 3; ++i) in.x0[i] = L::gather(x0, inst * 3 + i, ok, 0.0);
  if (with_cc) {
    in.cc_a = L::gather(cc, inst * (2 * N) + k + 3, in.has_u, 0.0);
    in.cc_last = L::gather(cc, inst * (2 * N) + (2 * N - 1), ok, 0.0);
  }
  Mk inner = ok & (k >= 1);
  if (lb != nullptr) {
    in.lbk = L::gather(lb, inst * N + k - 1, inner, 0.0);
    in.ubk = L::gather(ub, inst * N + k - 1, inner, 0.0);
  } else {
    in.lbk = L::gather(t.lb_tab, wp * t.n_cols + k - 1, inner, 0.0);
    in.ubk = L::gather(t.ub_tab, wp * t.n_cols + k - 1, inner, 0.0);
  }
}
template <class L>
MPMPC_HD void assemble_fields(const mpmpc_config& c, const PathTables& t, int B, const typename L::ival& inst,
                              const typename L::ival& k, const int* wp_id, const double* x0, const double* cc,
                              const double* lb, const double* ub, typename L::real* out) {
  StageIn<L> in;
  gather_stage<L>(c, t, B, inst, k, wp_id, x0, cc, lb, ub, in);
  assemble_stage<L>(c, in, out);
}
template <class L>
MPMPC_HD void assemble_lane(const mpmpc_config& c, const PathTables& t, int B, int ld, const typename L::ival& inst,
                            const typename L::ival& k, const int* wp_id, const double* x0, const double* cc,
                            const double* lb, const double* ub, double* qp) {
  using R = typename L::real;
  using I = typename L::ival;
  const typename L::mask ok = (inst < B) & (k >= 0) & (k <= c.N);
  StageIn<L> in;
  // the speed cap first, on its own: its tangent is the register peak of the stage, and nothing else is held while it runs
  in.cc_a = L::gather(cc, inst * (2 * c.N) + k + 3, ok & (k < c.N), 0.0);
  in.cc_last = L::gather(cc, inst * (2 * c.N) + (2 * c.N - 1), ok, 0.0);
  const R hi_v = speed_cap<L>(c, in.cc_a, in.cc_last);
  L::sched_barrier();
  gather_stage<L>(c, t, B, inst, k, wp_id, x0, cc, lb, ub, in, false);
  const I base = inst * ld + k;
  // The row of an instance is written up to the end of the last 128-byte line it touches (zeros behind stage N): N = 30 uses
  // 31 of a row's 32 doubles, and a line that misses its last 8 bytes is a partial write - a read-modify-write in the memory
  // system - for every second line of the output.
  const int kfill = ((c.N + 1 + 15) / 16) * 16 < ld ? ((c.N + 1 + 15) / 16) * 16 : ld;
  const typename L::mask okw = (inst < B) & (k >= 0) & (k < kfill);
  // (one divergent region around all 27 stores: a branch around each would put a wait for the store before it at every join)
  L::when(okw, [&] {
    assemble_stage_to<L>(c, in, [&](int f, const R& v) { L::store(qp, base + f * (B * ld), okw, sel(ok, v, R(0.0))); L::sched_barrier(); }, &hi_v);
  });
}

// ------------------------------------------------------------------------------------------
// K2: the solver.  All state is per lane.
// ------------------------------------------------------------------------------------------
// Solver settings plus the constants derived from them.  A wave-uniform double that the kernel has to
// COMPUTE sits in a VGPR pair for as long as it is live (there is no scalar FP64 unit), while one that
// arrives as a kernel argument is an SGPR pair usable as an operand directly: derive them on the host.
struct SolverParams : mpmpc_settings {
  double one_minus_alpha, inv_ipm_reg, inv_as_delta;
  double eps_abs10, eps_rel10, eps_prim_inf10, eps_dual_inf10;     // OSQP's approximate check: 10 x eps
};
inline SolverParams make_params(const mpmpc_settings& st) {
  SolverParams p;
  static_cast<mpmpc_settings&>(p) = st;
  p.one_minus_alpha = 1.0 - st.alpha;
  p.inv_ipm_reg = 1.0 / st.ipm_reg;
  p.inv_as_delta = 1.0 / st.as_delta;
  p.eps_abs10 = st.eps_abs * 10.0; p.eps_rel10 = st.eps_rel * 10.0;
  p.eps_prim_inf10 = st.eps_prim_inf * 10.0; p.eps_dual_inf10 = st.eps_dual_inf * 10.0;
  return p;
}

// May the certified polish solve the reduced (e_y, e_psi, kappa) problem?  (mpmpc_settings::reduce, layouts table
// in the Solver: the time state must carry neither cost nor bound, the speed must have its own strictly convex cost.)
// Does any weight matrix have an off-diagonal entry?  (src/MPC.py:150 puts the whole Q, R, QN into the Hessian.)  Such a
// configuration runs the general kernels with dense stage blocks (template flag FQ of the Solver), which take the seven
// numbers as one array: QN (01, 02, 12), Q (01, 02, 12), R (01) - weight_offdiag().
inline bool full_weights(const mpmpc_config& c) {
  for (int i = 0; i < 3; ++i)
    if (c.QN_offdiag[i] != 0.0 || c.Q_offdiag[i] != 0.0) return true;
  return c.R_offdiag[0] != 0.0;
}
MPMPC_HOST_DEVICE inline void weight_offdiag(const mpmpc_config& c, double w[7]) {
  for (int i = 0; i < 3; ++i) { w[i] = c.QN_offdiag[i]; w[3 + i] = c.Q_offdiag[i]; }
  w[6] = c.R_offdiag[0];
}
inline bool reducible(const mpmpc_config& c, const mpmpc_settings& st) {
  return st.reduce != 0 && st.polish != 0 && !full_weights(c) && c.Q[2] == 0.0 && c.QN[2] == 0.0 && c.QN_offdiag[0] == 0.0 &&
         c.QN_offdiag[1] == 0.0 && c.QN_offdiag[2] == 0.0 && c.R[0] > 0.0 && !(c.xmin[2] > -INFTY) && !(c.xmax[2] < INFTY) &&
         !(c.xmin[1] > -INFTY) && !(c.xmax[1] < INFTY);      // (and e_psi unbounded: the reduced layouts carry no slack for it)
}

// Are e_psi and t free of bounds (the interior point of the full problem may then skip their slack arithmetic)?
inline bool free_states(const mpmpc_config& c) {
  return !(c.xmin[1] > -INFTY) && !(c.xmax[1] < INFTY) && !(c.xmin[2] > -INFTY) && !(c.xmax[2] < INFTY);
}

// Lane split of the twisted factorisation for G lanes per instance and horizon N (shared by the
// launcher and the emulation): the chains meet at lane C - 1.
inline int lane_split(int G, int N) { return G > 64 ? G / 2 : (G == 16 ? 16 : (G == 32 ? 16 : (N + 1 <= 32 ? 16 : 32))); }
// Stage 0 sits on lane lane_offset of its instance, so that the two chains are equally long for any horizon:
// lanes off .. C-1 climb through stages 0 .. C-1-off, lanes C .. descend through N .. C-off (no second chain: 0).
MPMPC_HOST_DEVICE inline int lane_offset(int G, int C, int N) {
  const int o = C - (N + 2) / 2;
  return (C >= G || o < 0) ? 0 : o;
}

// FQ: FULL WEIGHTS - Q, R, QN are symmetric matrices with off-diagonal entries (src/MPC.py:150 puts the whole matrices into
// the Hessian).  The Hessian stays block diagonal - a 3 x 3 state block (Q on stages 0 .. N - 1, QN on stage N) and a 2 x 2
// input block (R) per stage - so a lane keeps the three off-diagonals of its state block (pod), the one of its input block
// (rod), and the off-diagonals of the INVERSES of the blocks of the current H = P + diagonal terms (hod, hud: dense 3 x 3 /
// 2 x 2 inverses per lane); the Schur complement keeps its block-tridiagonal structure, its blocks A inv(H) A' + B inv(H_u) B'
// and A inv(H) (-I)' are formed with the dense inverses (factor_core), and the products with P and with inv(H) gain terms.
// A block without off-diagonal entries is inverted entry by entry exactly as without the flag.  With FQ = false (the
// reference's own weights are diagonal) none of this code exists in the kernel.
// FREEX: the states e_psi and t are never boxed (xmin[1..2] = -inf, xmax[1..2] = +inf: the reference's own constraints,
// src/simulation.py:110-111) - the interior point of the FULL problem then carries no slack arithmetic for them
// (left out at compile time, like e_psi in the reduced layouts).
// CR: the two elimination chains of the reduced problem's factorisation (2 x 2 blocks, L::split == 16: each chain is one
// row of 16 lanes) are eliminated by CYCLIC REDUCTION IN CHOLESKY FORM - four lane-parallel levels of distance 1, 2, 4, 8
// inside the row instead of 15 dependent steps (factor_cr2 / s_solve_cr2 below).
// RKS: LAY_RED4 only - first of the six cold slots that hold the Sherman-Morrison vector of the current factorisation
// (it is read once per KKT solve: no reason to keep twelve registers for it)
template <class L, bool FQ = false, bool RED = false, bool FREEX = false, bool CR = false, int RKS = 0>
struct Solver {
  static_assert(!(FQ && RED), "the reduced problem needs a diagonal terminal weight");
  static constexpr bool kCR = CR && (L::split == 16 || L::split == 32);
  static constexpr bool kCR32 = kCR && L::split == 32;      // a chain is TWO rows of 16 lanes (see factor_cr2)
  using R = typename L::real;
  using Mk = typename L::mask;
  using I = typename L::ival;
  // An instance-wide condition is wave-wide when the wave holds ONE instance, and the loops below leave as soon as it is false:
  // inside them "update where the instance is still running" needs no select (two v_cndmask per double otherwise).
  MPMPC_HD static R updw(const Mk& m, const R& a, const R& b) { if constexpr (L::per_wave == 1) { (void)m; (void)b; return a; } else return sel(m, a, b); }
  MPMPC_HD static Mk updwb(const Mk& m, const Mk& a, const Mk& b) { if constexpr (L::per_wave == 1) { (void)m; (void)b; return a; } else return selb(m, a, b); }

  // ---- lane context
  int N, n_inst, off_;    // off_: lane of stage 0 inside the group (lane_offset)
  Mk vx, vu, first;      // lane holds a real stage (k <= N), a real input (k < N), k == 0
  Mk down_chain, is_mid, is_end, vxc;   // twisted factorisation: k >= C; chain-layout lanes C-1, 2C-1; chain-layout vx
  // Split layout of the interior-point stage (kSplit: G = 64 and N + 1 <= 32, so lanes 32..63 are free): lane k
  // keeps the three states of stage k, lane k + 32 takes its two inputs (entries v, kappa, -).  sU = upper
  // half, val3 = which of the lane's three entries exist, bU = the B-block numbers (ds, b20) on the upper lanes.
  static constexpr bool kSplit = (L::group == 64 && L::split == 16);
  Mk sU, val3[3];
  R bU[2];
  Mk valid[5];
  Mk live;               // this lane's instance exists
  // ---- where this lane's stage fields live (the unscaled bounds are re-read for the certificate)
  // ---- scaled problem
  R mI[3], a[6], b[2], g[5], p[5], q[5], D[5], Eeq[3], Eb[5], c;
  Mk p1_feasible;        // phase 1 converged to a point that violates nothing: the start of a second polish attempt
  Mk p1_converged;       // phase 1 ended at its converged optimum (not at an earlier iterate that already passed the ray test)
  Mk p1_marginal;        // infeasible by less than OSQP's own primal tolerance: solved on the boxes relaxed by that much (phase1_accept)
  R p1_viol;             // ... and the violation (unscaled) the plan is allowed
  R p1_band;             // phase 1 may stop at the first iterate whose ray passes the Farkas test only while that iterate's
                         // violation exceeds this (unscaled); below it the iteration runs to its converged optimum, whose
                         // violation decides "marginal" (phase1_accept).  0: stop at the first ray.
  R pod[3], hod[3];      // FQ: off-diagonals (01, 02, 12) of the lane's state cost block and of its inv(H) state block
  R rod, hud;            // FQ: off-diagonal (v, kappa) of the lane's input cost block and of its inv(H) input block
  R podS[3];             // FQ, split layout of the interior point: pod on the state lanes, (rod of the stage, 0, 0) on the
                         // input lanes - there entries 0, 1 are (v, kappa), and hod[0] then holds hud
  Mk term;               // this lane holds stage N
  R leq[3], lb[5], ub[5];
  // ---- linear algebra
  R hinv[5], Li[6], Gin[9], Gout[9];          // Li, Gin, Gout in chain layout (see factor)
  // LAY_RED4: the rank-one term of the cost, rk_c = its vector on the entries e_y (0) and v (1) - zero on every other entry
  // and on lanes without a stage; u = inv(K0) [rk_c; 0] of the current factorisation (K0: the KKT matrix without the rank-one
  // term) waits in the cold slots RKS .. RKS + 5 (4 entries, 2 equality rows), rk_g = 1 / (1 + rk_c'u)
  R rk_c[2], rk_g;
  // ---- ADMM state
  R x[5], zeq[3], zb[5], yeq[3], yb[5];
  R rho, rb[5], rbinv[5], rho_eq, rinv_eq;
  // (the last step's x_prev, dy_eq, dy_box live in cold storage, slots 18..30: only the
  //  infeasibility tests read them)
  // ---- results
  I status, iters, ipm_iters, polished;
  I act_bits;            // active set of the certified point: bit j lower, bit 5 + j upper, bit 30 = valid (warm start)
  R pri_res, dua_res;

  // ======================================================================== helpers
  // Commit `nw` where the lane's instance is still running.  With one instance per wavefront
  // (G = 64) the loops exit as soon as that instance stops, so the select is dropped.
  MPMPC_HD static R keep(const Mk& on, const R& nw, const R& old) {
    if constexpr (L::per_wave == 1) return nw; else return sel(on, nw, old);
  }
  MPMPC_HD static I keepi(const Mk& on, const I& nw, const I& old) {
    if constexpr (L::per_wave == 1) return nw; else return seli(on, nw, old);
  }
  MPMPC_HD static R limit(const R& v) {
    R r = sel(v < R(MIN_SCALING), R(1.0), v);
    return sel(r > R(MAX_SCALING), R(MAX_SCALING), r);
  }
  // w = A_k x_k + B_k u_k  (contribution of this stage to equality block k+1)
  MPMPC_HD void couple(const R v[5], R w[3]) const {
    w[0] = fma_(a[1], v[1], a[0] * v[0]);
    w[1] = fma_(b[0], v[4], fma_(a[3], v[1], a[2] * v[0]));
    w[2] = fma_(b[1], v[3], fma_(a[5], v[2], a[4] * v[0]));
  }
  MPMPC_HD void Aeq_mul(const R v[5], R r[3]) const {
    R w[3];
    couple(v, w);
    MPMPC_UNROLL
    for (int i = 0; i < 3; ++i) r[i] = fma_(mI[i], v[i], L::up(w[i]));
  }
  MPMPC_HD void AeqT_mul(const R nu[3], R t[5]) const {
    R nd[3];
    MPMPC_UNROLL
    for (int i = 0; i < 3; ++i) nd[i] = L::down(nu[i]);
    t[0] = fma_(a[4], nd[2], fma_(a[2], nd[1], fma_(a[0], nd[0], mI[0] * nu[0])));
    t[1] = fma_(a[3], nd[1], fma_(a[1], nd[0], mI[1] * nu[1]));
    t[2] = fma_(a[5], nd[2], mI[2] * nu[2]);
    t[3] = b[1] * nd[2];
    t[4] = b[0] * nd[1];
  }
  MPMPC_HD R gmax5(const R v[5]) const {
    R m(0.0);
    MPMPC_UNROLL
    for (int j = 0; j < 5; ++j) m = max_(m, sel(valid[j], abs_(v[j]), R(0.0)));
    return L::gmax(m);
  }
  MPMPC_HD R gmax3(const R v[3]) const {
    R m(0.0);
    MPMPC_UNROLL
    for (int i = 0; i < 3; ++i) m = max_(m, sel(vx, abs_(v[i]), R(0.0)));
    return L::gmax(m);
  }

  // ---- full terminal weight (FQ)
  // v += offdiag(M) w over the three states, M given by its off-diagonals od = (01, 02, 12)
  MPMPC_HD static void od_mul_add(const R od[3], const R* w, R* v) {
    v[0] = fma_(od[1], w[2], fma_(od[0], w[1], v[0]));
    v[1] = fma_(od[2], w[2], fma_(od[0], w[0], v[1]));
    v[2] = fma_(od[2], w[1], fma_(od[1], w[0], v[2]));
  }
  // Hd: diagonal of the Hessian blocks whose inverses h the caller has just formed entry by entry (layout LAY: LAY_FULL = 0,
  // LAY_SPLIT = 1).  Where a block has off-diagonal cost entries it is dense (Hd on the diagonal, pod / rod off it): replace
  // its entries of h by the diagonal of its inverse and keep the inverse's off-diagonals in hod / hud.
  template <int LAY = 0, bool USE = true>
  MPMPC_HD void dense_blocks(const R* Hd, R* h) {
    if constexpr (FQ && USE) {
      const R* po = LAY == 1 ? podS : pod;
      const R zero(0.0);
      const Mk dn = (abs_(po[0]) > zero) | (abs_(po[1]) > zero) | (abs_(po[2]) > zero);
      const R c00 = fma_(Hd[1], Hd[2], -(po[2] * po[2])), c01 = fma_(po[1], po[2], -(po[0] * Hd[2])),
              c02 = fma_(po[0], po[2], -(po[1] * Hd[1])), c11 = fma_(Hd[0], Hd[2], -(po[1] * po[1])),
              c12 = fma_(po[0], po[1], -(Hd[0] * po[2])), c22 = fma_(Hd[0], Hd[1], -(po[0] * po[0]));
      const R idet = R(1.0) / fma_(po[1], c02, fma_(po[0], c01, Hd[0] * c00));
      h[0] = sel(dn, c00 * idet, h[0]); h[1] = sel(dn, c11 * idet, h[1]); h[2] = sel(dn, c22 * idet, h[2]);
      hod[0] = sel(dn, c01 * idet, zero); hod[1] = sel(dn, c02 * idet, zero); hod[2] = sel(dn, c12 * idet, zero);
      if constexpr (LAY == 0) {
        const Mk d2 = abs_(rod) > zero;
        const R idet2 = R(1.0) / fma_(Hd[3], Hd[4], -(rod * rod));
        const R h3 = Hd[4] * idet2, h4 = Hd[3] * idet2;
        h[3] = sel(d2, h3, h[3]); h[4] = sel(d2, h4, h[4]);
        hud = sel(d2, -(rod * idet2), zero);
      }
    } else if constexpr (FQ) {
      hod[0] = hod[1] = hod[2] = R(0.0);
      hud = R(0.0);
    }
  }
  // out += offdiag(P) x  and  t += offdiag(inv H) r  in the layout LAY (FQ only; LAY_FULL: 5 entries, LAY_SPLIT: 3)
  template <int LAY>
  MPMPC_HD void Poff_add(const R* x, R* out) const {
    if constexpr (LAY == 0) {
      od_mul_add(pod, x, out);
      out[3] = fma_(rod, x[4], out[3]); out[4] = fma_(rod, x[3], out[4]);
    } else {
      static_assert(LAY == 1, "full weights run the full problem: LAY_FULL or LAY_SPLIT");
      od_mul_add(podS, x, out);
    }
  }
  template <int LAY>
  MPMPC_HD void Hoff_add(const R* r, R* t) const {
    od_mul_add(hod, r, t);
    if constexpr (LAY == 0) { t[3] = fma_(hud, r[4], t[3]); t[4] = fma_(hud, r[3], t[4]); }
  }

  // ======================================================================== setup
  // field f of this lane's stage, straight from the stage-blocked QP; bounds clipped like OSQP does
  // (the unscaled offsets and bounds are needed again after the Ruiz passes and in the certificate: they wait in
  //  cold storage, slots COLD_RAW .., so that no field is ever re-read from memory)
  static constexpr int COLD_RAW = 42, COLD_COST = 56;
  MPMPC_HD R beq_raw(int i) const { return L::cold_get(COLD_RAW + i); }
  MPMPC_HD R lo_raw(int j) const { return L::cold_get(COLD_RAW + 3 + j); }
  MPMPC_HD R hi_raw(int j) const { return L::cold_get(COLD_RAW + 8 + j); }
  // the 27 stage fields of (inst, k) from a stage-blocked QP in memory (K1's output)
  MPMPC_HD static void fetch_fields(const double* qp, int B, int ld, const I& inst, const I& k, int N_, R* fields) {
    Mk ok = (inst < B) & within_(k, 0, N_);
    I base = inst * ld + k;
    MPMPC_UNROLL
    for (int f = 0; f < MPMPC_NUM_FIELDS; ++f) fields[f] = L::load(qp, base + f * (B * ld), ok, 0.0);
  }

  // fields: the 27 stage fields of this lane's (instance, stage) - assemble_fields, or fetch_fields
  // qn_off: the seven off-diagonal weights (FQ only): QN (01, 02, 12), Q (01, 02, 12), R (01) - mpmpc_config::QN_offdiag ..
  MPMPC_HD void load(const R* fields, int B, const I& inst, const I& k, int N_, const double* qn_off = nullptr) {
    N = N_;
    n_inst = B;
    live = inst < B;
    vx = live & within_(k, 0, N);
    vu = live & within_(k, 0, N - 1);
    first = (k == 0);
    {
      const int C = L::split;
      off_ = lane_offset(L::group, C, N);
      I kl = k + off_;                                  // lane inside the instance's group
      down_chain = (kl >= C);
      is_mid = (kl == C - 1);
      is_end = (kl == 2 * C - 1);
      I kc = seli(down_chain & (kl < 2 * C), kl * (-1) + (3 * C - 1), kl) - off_;      // stage held in chain layout
      vxc = live & within_(kc, 0, N);
      if constexpr (kSplit) {
        sU = (kl >= 32);
        Mk vU = live & within_(kl, 32 + off_, N + 31 + off_);             // stage kl - 32 - off has inputs
        val3[0] = selb(sU, vU, vx); val3[1] = selb(sU, vU, vx); val3[2] = vx & !sU;
      }
    }
    valid[0] = valid[1] = valid[2] = vx;
    valid[3] = valid[4] = vu;
    term = live & (k == N);
    if constexpr (FQ) {
      MPMPC_UNROLL
      for (int i = 0; i < 3; ++i) {
        pod[i] = sel(term, R(qn_off ? qn_off[i] : 0.0), sel(vu, R(qn_off ? qn_off[3 + i] : 0.0), R(0.0)));
        hod[i] = R(0.0); podS[i] = R(0.0);
      }
      rod = sel(vu, R(qn_off ? qn_off[6] : 0.0), R(0.0));
      hud = R(0.0);
    }
    auto fld = [&](int f, double dflt) { return sel(vx, fields[f], R(dflt)); };
    MPMPC_UNROLL
    for (int i = 0; i < 3; ++i) L::cold_put(COLD_RAW + i, fld(F_BEQ + i, 0.0));
    MPMPC_UNROLL
    for (int j = 0; j < 5; ++j) {
      L::cold_put(COLD_RAW + 3 + j, max_(fld(F_LO + j, -INFTY), R(-INFTY)));
      L::cold_put(COLD_RAW + 8 + j, min_(fld(F_HI + j, INFTY), R(INFTY)));
    }
    R ds = fld(F_DS, 0.0), a10 = fld(F_A10, 0.0), a20 = fld(F_A20, 0.0), b20 = fld(F_B20, 0.0);
    R one = sel(vu, R(1.0), R(0.0));
    a[0] = one; a[1] = ds; a[2] = a10; a[3] = one; a[4] = a20; a[5] = one;
    b[0] = ds; b[1] = b20;
    MPMPC_UNROLL
    for (int i = 0; i < 3; ++i) { mI[i] = R(-1.0); Eeq[i] = R(1.0); }
    MPMPC_UNROLL
    for (int j = 0; j < 5; ++j) {
      q[j] = fld(F_Q + j, 0.0);
      p[j] = fld(F_P + j, 1.0);
      g[j] = R(1.0); D[j] = R(1.0); Eb[j] = R(1.0);
    }
    c = R(1.0);
  }

  // OSQP scale_data(): `passes` Ruiz sweeps with cost normalisation, then l, u <- E l, E u.  Resumable:
  // scale(a) followed by scale(b) is scale(a + b) bit for bit; instances outside `on` keep their scaling
  // (their factors are forced to exactly 1).
  MPMPC_HD void scale(int passes, const Mk& on) {
    ruiz(passes, on);
    MPMPC_UNROLL
    for (int i = 0; i < 3; ++i) leq[i] = Eeq[i] * beq_raw(i);
    MPMPC_UNROLL
    for (int j = 0; j < 5; ++j) { lb[j] = Eb[j] * lo_raw(j); ub[j] = Eb[j] * hi_raw(j); }
  }
  // the Ruiz sweeps alone: cost, rows and their scalings (the bounds are scaled by the caller)
  MPMPC_HD void ruiz(int passes, const Mk& on) {
    const R n_total(double(5 * N + 3));
    for (int it = 0; it < passes; ++it) {
      R cn[5], rn[3], r_own[3];
      cn[0] = max_(max_(max_(abs_(p[0]), abs_(mI[0])), max_(abs_(a[0]), abs_(a[2]))), max_(abs_(a[4]), abs_(g[0])));
      cn[1] = max_(max_(abs_(p[1]), abs_(mI[1])), max_(max_(abs_(a[1]), abs_(a[3])), abs_(g[1])));
      cn[2] = max_(max_(abs_(p[2]), abs_(mI[2])), max_(abs_(a[5]), abs_(g[2])));
      if constexpr (FQ) {
        cn[0] = max_(cn[0], max_(abs_(pod[0]), abs_(pod[1])));
        cn[1] = max_(cn[1], max_(abs_(pod[0]), abs_(pod[2])));
        cn[2] = max_(cn[2], max_(abs_(pod[1]), abs_(pod[2])));
      }
      cn[3] = max_(max_(abs_(p[3]), abs_(b[1])), abs_(g[3]));
      cn[4] = max_(max_(abs_(p[4]), abs_(b[0])), abs_(g[4]));
      if constexpr (FQ) { cn[3] = max_(cn[3], abs_(rod)); cn[4] = max_(cn[4], abs_(rod)); }
      r_own[0] = max_(abs_(a[0]), abs_(a[1]));
      r_own[1] = max_(max_(abs_(a[2]), abs_(a[3])), abs_(b[0]));
      r_own[2] = max_(max_(abs_(a[4]), abs_(a[5])), abs_(b[1]));
      MPMPC_UNROLL
      for (int i = 0; i < 3; ++i) rn[i] = max_(abs_(mI[i]), L::up(r_own[i]));
      R Dt[5], Et[3], Etb[5], Etd[3];
      MPMPC_UNROLL
      for (int j = 0; j < 5; ++j) {
        Dt[j] = R(1.0) / sqrt_(limit(cn[j]));
        Etb[j] = R(1.0) / sqrt_(limit(abs_(g[j])));
      }
      MPMPC_UNROLL
      for (int j = 0; j < 5; ++j) { Dt[j] = keep(on, Dt[j], R(1.0)); Etb[j] = keep(on, Etb[j], R(1.0)); }
      MPMPC_UNROLL
      for (int i = 0; i < 3; ++i) { Et[i] = keep(on, R(1.0) / sqrt_(limit(rn[i])), R(1.0)); Etd[i] = L::down(Et[i]); }
      MPMPC_UNROLL
      for (int j = 0; j < 5; ++j) {
        p[j] = (Dt[j] * p[j]) * Dt[j];
        g[j] = (Etb[j] * g[j]) * Dt[j];
        q[j] = Dt[j] * q[j];
        D[j] = D[j] * Dt[j];
        Eb[j] = Eb[j] * Etb[j];
      }
      if constexpr (FQ) {
        pod[0] = (Dt[0] * pod[0]) * Dt[1]; pod[1] = (Dt[0] * pod[1]) * Dt[2]; pod[2] = (Dt[1] * pod[2]) * Dt[2];
        rod = (Dt[3] * rod) * Dt[4];
      }
      MPMPC_UNROLL
      for (int i = 0; i < 3; ++i) { mI[i] = (Et[i] * mI[i]) * Dt[i]; Eeq[i] = Eeq[i] * Et[i]; }
      a[0] = (Etd[0] * a[0]) * Dt[0]; a[1] = (Etd[0] * a[1]) * Dt[1];
      a[2] = (Etd[1] * a[2]) * Dt[0]; a[3] = (Etd[1] * a[3]) * Dt[1];
      a[4] = (Etd[2] * a[4]) * Dt[0]; a[5] = (Etd[2] * a[5]) * Dt[2];
      b[0] = (Etd[1] * b[0]) * Dt[4]; b[1] = (Etd[2] * b[1]) * Dt[3];
      // cost normalisation
      R s(0.0), mq(0.0);
      MPMPC_UNROLL
      for (int j = 0; j < 5; ++j) {
        R colmax = abs_(p[j]);                      // inf-norm of column j of P
        if constexpr (FQ) {
          if (j == 0) colmax = max_(colmax, max_(abs_(pod[0]), abs_(pod[1])));
          if (j == 1) colmax = max_(colmax, max_(abs_(pod[0]), abs_(pod[2])));
          if (j == 2) colmax = max_(colmax, max_(abs_(pod[1]), abs_(pod[2])));
          if (j >= 3) colmax = max_(colmax, abs_(rod));
        }
        s = s + sel(valid[j], colmax, R(0.0));
        mq = max_(mq, sel(valid[j], abs_(q[j]), R(0.0)));
      }
      R ct = L::gsum(s) / n_total;
      R nq = limit(L::gmax(mq));
      ct = keep(on, R(1.0) / limit(max_(ct, nq)), R(1.0));
      MPMPC_UNROLL
      for (int j = 0; j < 5; ++j) { p[j] = p[j] * ct; q[j] = q[j] * ct; }
      if constexpr (FQ) { pod[0] = pod[0] * ct; pod[1] = pod[1] * ct; pod[2] = pod[2] * ct; rod = rod * ct; }
      c = c * ct;
    }
  }

  // OSQP set_rho_vec(): per-row step size by constraint type
  MPMPC_HD void set_rho(const R& rho_new) {
    rho = rho_new;
    rho_eq = R(RHO_EQ_FACTOR) * rho;
    rinv_eq = R(1.0) / rho_eq;
    MPMPC_UNROLL
    for (int j = 0; j < 5; ++j) {
      Mk freerow = (lb[j] < R(-INF_BOUND)) & (ub[j] > R(INF_BOUND));
      Mk eqrow = (ub[j] - lb[j]) < R(RHO_TOL);
      rb[j] = sel(freerow, R(RHO_MIN), sel(eqrow, rho_eq, rho));
      rbinv[j] = R(1.0) / rb[j];
    }
  }

  // Block-tridiagonal Cholesky of S = Aeq diag(h) Aeq' + r I  (3x3 blocks, one per lane), as a
  // TWISTED factorisation: stages 0 .. C-2 are eliminated upwards, stages N .. C downwards, both at
  // the same time, and the two chains meet in stage C-1 (C = L::split).  The serial depth of the
  // factorisation and of each substitution sweep is max(C-1, N-C+1) + 1 steps instead of N + 1.
  // Inside factor() / s_solve() the data lives in "chain layout": the lanes [C, 2C) of the instance
  // are reversed (L::mirror), so that both chains advance by the same one-lane shift L::up and
  // retreat by L::down (as L::cup / L::cdown: zero inflow at the chain ends).  Lane C-1 is the meeting stage ("mid"), lane 2C-1 holds stage C ("end").
  // Per lane, in chain layout:  Li = inv(L_kk) (lower),  Gin = -inv(L_kk) M_in,  Gout = -inv(L_kk)' M_own'
  // where M_in is the coupling block received from the chain predecessor and M_own the one handed on
  // (L_{k+1,k} going up, U_{k-1,k} going down).  The end lane keeps M_own itself in Gout: its only
  // outward neighbour is mid, reached through the two junction steps of s_solve.
  MPMPC_HD int chain_steps() const {
    const int C = L::split;
    int fwd = N + 1 < C - 1 - off_ ? N + 1 : C - 1 - off_, bwd = N - C + off_ + 1;
    return fwd > bwd ? fwd : bwd;
  }
  MPMPC_HD void factor(const R h[5], const R& r) {
    MPMPC_UNROLL
    for (int j = 0; j < 5; ++j) hinv[j] = h[j];
    if constexpr (FQ) factor_core(h, (b[0] * b[0]) * h[4], (b[1] * b[1]) * h[3], r, (b[0] * b[1]) * hud);
    else factor_core(h, (b[0] * b[0]) * h[4], (b[1] * b[1]) * h[3], r);
  }
  // hx = 1/H of the three states; w2b, w5b = b0^2 h_kappa, b1^2 h_v (what the inputs add to A H A' + B H B'); FQ: w4b = b0 b1
  // times the off-diagonal of the inputs' inverse block (rows e_psi, t of B inv(H_u) B')
  MPMPC_HD void factor_core(const R hx[3], const R& w2b, const R& w5b, const R& r, const R& w4b = R(0.0)) {
    const R* h = hx;
    R W[6], T[6], Dg[6], To[9];
    [[maybe_unused]] R T9[9];
    if constexpr (FQ) {
      // dense inverse of the state block: h on the diagonal, hod (01, 02, 12) off it.  AH = A inv(H), row-major 3 x 3
      (void)T;
      const R ah00 = fma_(a[1], hod[0], a[0] * h[0]), ah01 = fma_(a[1], h[1], a[0] * hod[0]), ah02 = fma_(a[1], hod[2], a[0] * hod[1]);
      const R ah10 = fma_(a[3], hod[0], a[2] * h[0]), ah11 = fma_(a[3], h[1], a[2] * hod[0]), ah12 = fma_(a[3], hod[2], a[2] * hod[1]);
      const R ah20 = fma_(a[5], hod[1], a[4] * h[0]), ah21 = fma_(a[5], hod[2], a[4] * hod[0]), ah22 = fma_(a[5], h[2], a[4] * hod[1]);
      W[0] = fma_(ah01, a[1], ah00 * a[0]);
      W[1] = fma_(ah11, a[1], ah10 * a[0]);
      W[2] = fma_(ah11, a[3], ah10 * a[2]) + w2b;
      W[3] = fma_(ah21, a[1], ah20 * a[0]);
      W[4] = fma_(ah21, a[3], ah20 * a[2]) + w4b;
      W[5] = fma_(ah22, a[5], ah20 * a[4]) + w5b;
      T9[0] = ah00 * mI[0]; T9[1] = ah01 * mI[1]; T9[2] = ah02 * mI[2];      // S_{k+1,k} = A inv(H) (-I)': dense
      T9[3] = ah10 * mI[0]; T9[4] = ah11 * mI[1]; T9[5] = ah12 * mI[2];
      T9[6] = ah20 * mI[0]; T9[7] = ah21 * mI[1]; T9[8] = ah22 * mI[2];
    } else {
      R a0h = a[0] * h[0], a2h = a[2] * h[0], a4h = a[4] * h[0], a1h = a[1] * h[1], a3h = a[3] * h[1];
      W[0] = fma_(a[1], a1h, a[0] * a0h);
      W[1] = fma_(a[3], a1h, a[2] * a0h);
      W[2] = fma_(a[3], a3h, a[2] * a2h) + w2b;
      W[3] = a[4] * a0h;
      W[4] = a[4] * a2h;
      W[5] = fma_(a[5] * a[5], h[2], a[4] * a4h) + w5b;
      T[0] = a0h * mI[0]; T[1] = a1h * mI[1];                 // S_{k+1,k} row 0: cols 0,1
      T[2] = a2h * mI[0]; T[3] = a3h * mI[1];                 //           row 1: cols 0,1
      T[4] = a4h * mI[0]; T[5] = (a[5] * h[2]) * mI[2];       //           row 2: cols 0,2
    }
    // diagonal block S_kk (lower: 00 10 11 20 21 22): own -I H -I' + r, plus the predecessor's W
    MPMPC_UNROLL
    for (int i = 0; i < 6; ++i) Dg[i] = L::up(W[i]);
    Dg[0] = Dg[0] + fma_(mI[0] * mI[0], h[0], r);
    Dg[2] = Dg[2] + fma_(mI[1] * mI[1], h[1], r);
    Dg[5] = Dg[5] + fma_(mI[2] * mI[2], h[2], r);
    if constexpr (FQ) {       // -I inv(H_N) -I' of the terminal stage is dense
      Dg[1] = fma_(mI[0] * mI[1], hod[0], Dg[1]);
      Dg[3] = fma_(mI[0] * mI[2], hod[1], Dg[3]);
      Dg[4] = fma_(mI[1] * mI[2], hod[2], Dg[4]);
    }
    // coupling handed on: S_{k+1,k} = T going up, S_{k-1,k} = T_{k-1}' going down, nothing from mid
    if constexpr (FQ) {
      R Tu[9];
      MPMPC_UNROLL
      for (int i = 0; i < 9; ++i) Tu[i] = L::up(T9[i]);
      const R zero(0.0);
      MPMPC_UNROLL
      for (int i = 0; i < 3; ++i) {
        MPMPC_UNROLL
        for (int j = 0; j < 3; ++j) To[3 * i + j] = sel(is_mid, zero, sel(down_chain, Tu[3 * j + i], T9[3 * i + j]));
      }
    } else {
      R Tu[6];
      MPMPC_UNROLL
      for (int i = 0; i < 6; ++i) Tu[i] = L::up(T[i]);
      const R zero(0.0);
      To[0] = sel(down_chain, Tu[0], T[0]); To[1] = sel(down_chain, Tu[2], T[1]); To[2] = sel(down_chain, Tu[4], zero);
      To[3] = sel(down_chain, Tu[1], T[2]); To[4] = sel(down_chain, Tu[3], T[3]); To[5] = zero;
      To[6] = sel(down_chain, zero, T[4]);  To[7] = zero;                         To[8] = sel(down_chain, Tu[5], T[5]);
      MPMPC_UNROLL
      for (int i = 0; i < 9; ++i) To[i] = sel(is_mid, zero, To[i]);
    }
    MPMPC_UNROLL
    for (int i = 0; i < 6; ++i) Dg[i] = L::mirror(Dg[i]);
    MPMPC_UNROLL
    for (int i = 0; i < 9; ++i) if (FQ || (i != 5 && i != 7)) To[i] = L::mirror(To[i]);
    R M[9], Ls[9];
    MPMPC_UNROLL
    for (int i = 0; i < 9; ++i) M[i] = R(0.0);
    const int last = chain_steps();
    auto fstep = [&](bool junction) {
      R Mr[9];
      MPMPC_UNROLL
      for (int i = 0; i < 9; ++i) Mr[i] = L::cup(M[i]);
      // S = Dg - Mr Mr' (lower part), products subtracted inside the FMAs
      R S00 = fma_(-Mr[2], Mr[2], fma_(-Mr[1], Mr[1], fma_(-Mr[0], Mr[0], Dg[0])));
      R S10 = fma_(-Mr[5], Mr[2], fma_(-Mr[4], Mr[1], fma_(-Mr[3], Mr[0], Dg[1])));
      R S11 = fma_(-Mr[5], Mr[5], fma_(-Mr[4], Mr[4], fma_(-Mr[3], Mr[3], Dg[2])));
      R S20 = fma_(-Mr[8], Mr[2], fma_(-Mr[7], Mr[1], fma_(-Mr[6], Mr[0], Dg[3])));
      R S21 = fma_(-Mr[8], Mr[5], fma_(-Mr[7], Mr[4], fma_(-Mr[6], Mr[3], Dg[4])));
      R S22 = fma_(-Mr[8], Mr[8], fma_(-Mr[7], Mr[7], fma_(-Mr[6], Mr[6], Dg[5])));
      if (junction) {
        // junction: both chains have settled; mid also loses the block of the end lane
        R Mx[9];
        MPMPC_UNROLL
        for (int i = 0; i < 9; ++i) Mx[i] = sel(is_mid, L::down(L::mirror(M[i])), R(0.0));
        S00 = fma_(-Mx[2], Mx[2], fma_(-Mx[1], Mx[1], fma_(-Mx[0], Mx[0], S00)));
        S10 = fma_(-Mx[5], Mx[2], fma_(-Mx[4], Mx[1], fma_(-Mx[3], Mx[0], S10)));
        S11 = fma_(-Mx[5], Mx[5], fma_(-Mx[4], Mx[4], fma_(-Mx[3], Mx[3], S11)));
        S20 = fma_(-Mx[8], Mx[2], fma_(-Mx[7], Mx[1], fma_(-Mx[6], Mx[0], S20)));
        S21 = fma_(-Mx[8], Mx[5], fma_(-Mx[7], Mx[4], fma_(-Mx[6], Mx[3], S21)));
        S22 = fma_(-Mx[8], Mx[8], fma_(-Mx[7], Mx[7], fma_(-Mx[6], Mx[6], S22)));
      }
      // 3x3 Cholesky through reciprocal square roots: i_jj = 1 / l_jj
      R i00 = rsqrt_(S00);
      R l10 = S10 * i00, l20 = S20 * i00;
      R i11 = rsqrt_(fma_(-l10, l10, S11));
      R l21 = fma_(-l20, l10, S21) * i11;
      R i22 = rsqrt_(fma_(-l21, l21, fma_(-l20, l20, S22)));
      R i10 = -(l10 * i00) * i11;
      R i21 = -(l21 * i11) * i22;
      R i20 = -(fma_(l21, i10, l20 * i00)) * i22;
      Li[0] = i00; Li[1] = i10; Li[2] = i11; Li[3] = i20; Li[4] = i21; Li[5] = i22;
      MPMPC_UNROLL
      for (int i = 0; i < 9; ++i) Ls[i] = Mr[i];
      // M = To * inv(L_kk)'  -> consumed by the next lane of the chain in the next sweep step
      M[0] = To[0] * i00; M[1] = fma_(To[1], i11, To[0] * i10); M[2] = fma_(To[2], i22, fma_(To[1], i21, To[0] * i20));
      M[3] = To[3] * i00; M[4] = fma_(To[4], i11, To[3] * i10); M[5] = fma_(To[4], i21, To[3] * i20);
      M[6] = To[6] * i00; M[7] = To[6] * i10;                   M[8] = fma_(To[8], i22, To[6] * i20);
      if constexpr (FQ) {       // (dense coupling: the two entries the diagonal-weight blocks do not have)
        M[5] = fma_(To[5], i22, M[5]);
        M[7] = fma_(To[7], i11, M[7]);
        M[8] = fma_(To[7], i21, M[8]);
      }
    };
    {
      MPMPC_SERIAL_BEGIN();
      int s = 0;                                   // two steps per trip (see s_solve), then the junction step
      for (; s + 2 <= last; s += 2) { fstep(false); fstep(false); }
      for (; s < last; ++s) fstep(false);
      fstep(true);
      MPMPC_SERIAL_END(last + 1);
    }
    // recurrence matrices of the two substitution sweeps (stored negated, so a sweep step is 9 FMAs):
    //   inward    y_k  = inv(L_kk) b_k + Gin_k y_pred,        Gin_k  = -inv(L_kk) M_in
    //   outward   nu_k = inv(L_kk)' y_k + Gout_k nu_succ,     Gout_k = -inv(L_kk)' M_own'
    MPMPC_UNROLL
    for (int j = 0; j < 3; ++j) {
      Gin[0 + j] = -(Li[0] * Ls[0 + j]);
      Gin[3 + j] = -fma_(Li[2], Ls[3 + j], Li[1] * Ls[0 + j]);
      Gin[6 + j] = -fma_(Li[5], Ls[6 + j], fma_(Li[4], Ls[3 + j], Li[3] * Ls[0 + j]));
    }
    MPMPC_UNROLL
    for (int j = 0; j < 3; ++j) {                                // Gout[i][j] = -sum_m Li[m][i] * M[j][m]
      R g0 = -fma_(Li[3], M[3 * j + 2], fma_(Li[1], M[3 * j + 1], Li[0] * M[3 * j + 0]));
      R g1 = -fma_(Li[4], M[3 * j + 2], Li[2] * M[3 * j + 1]);
      R g2 = -(Li[5] * M[3 * j + 2]);
      Gout[0 + j] = sel(is_end, M[0 + j], g0);
      Gout[3 + j] = sel(is_end, M[3 + j], g1);
      Gout[6 + j] = sel(is_end, M[6 + j], g2);
    }
  }

  MPMPC_HD void s_solve(const R bv[3], R nu[3]) const {
    // lane-parallel part first, then two sweeps whose serial step is one 3x3 matrix-vector product
    R b0 = sel(vxc, L::mirror(bv[0]), R(0.0)), b1 = sel(vxc, L::mirror(bv[1]), R(0.0)), b2 = sel(vxc, L::mirror(bv[2]), R(0.0));
    R c0 = Li[0] * b0;
    R c1 = fma_(Li[2], b1, Li[1] * b0);
    R c2 = fma_(Li[5], b2, fma_(Li[4], b1, Li[3] * b0));
    const int last = chain_steps();
    R y0(0.0), y1(0.0), y2(0.0);
    // A loop-back branch costs about as much as six of the step's fifteen instructions, and the compiler
    // may not partially unroll a loop of convergent (DPP) operations: four steps per trip by hand.
    auto in_step = [&]() {
      R p0 = L::cup(y0), p1 = L::cup(y1), p2 = L::cup(y2);
      y0 = fma_(Gin[2], p2, fma_(Gin[1], p1, fma_(Gin[0], p0, c0)));
      y1 = fma_(Gin[5], p2, fma_(Gin[4], p1, fma_(Gin[3], p0, c1)));
      y2 = fma_(Gin[8], p2, fma_(Gin[7], p1, fma_(Gin[6], p0, c2)));
    };
    {
      MPMPC_SERIAL_BEGIN();
      int s = 0;
      for (; s + 4 <= last; s += 4) { in_step(); in_step(); in_step(); in_step(); }
      for (; s < last; ++s) in_step();
      MPMPC_SERIAL_END(last);
    }
    {
      // inward junction: the end lane forms M_own y, mid takes it on top of its chain input
      MPMPC_SERIAL_BEGIN();                        // (census: useful on the two lanes of the junction only)
      R t0 = fma_(Gout[2], y2, fma_(Gout[1], y1, Gout[0] * y0));
      R t1 = fma_(Gout[5], y2, fma_(Gout[4], y1, Gout[3] * y0));
      R t2 = fma_(Gout[8], y2, fma_(Gout[7], y1, Gout[6] * y0));
      const R zero(0.0);
      t0 = sel(is_mid, L::down(L::mirror(t0)), zero);
      t1 = sel(is_mid, L::down(L::mirror(t1)), zero);
      t2 = sel(is_mid, L::down(L::mirror(t2)), zero);
      R e0 = c0 - Li[0] * t0;
      R e1 = c1 - fma_(Li[2], t1, Li[1] * t0);
      R e2 = c2 - fma_(Li[5], t2, fma_(Li[4], t1, Li[3] * t0));
      R p0 = L::cup(y0), p1 = L::cup(y1), p2 = L::cup(y2);
      y0 = fma_(Gin[2], p2, fma_(Gin[1], p1, fma_(Gin[0], p0, e0)));
      y1 = fma_(Gin[5], p2, fma_(Gin[4], p1, fma_(Gin[3], p0, e1)));
      y2 = fma_(Gin[8], p2, fma_(Gin[7], p1, fma_(Gin[6], p0, e2)));
      MPMPC_SERIAL_END(N + 1);
    }
    R d0 = fma_(Li[3], y2, fma_(Li[1], y1, Li[0] * y0));
    R d1 = fma_(Li[4], y2, Li[2] * y1);
    R d2 = Li[5] * y2;
    {
      // outward junction: nu of mid is final (it has no successor); the end lane takes it through M_own'
      MPMPC_SERIAL_BEGIN();
      const R zero(0.0);
      R m0 = sel(is_end, L::mirror(L::up(d0)), zero);
      R m1 = sel(is_end, L::mirror(L::up(d1)), zero);
      R m2 = sel(is_end, L::mirror(L::up(d2)), zero);
      R w0 = fma_(Gout[6], m2, fma_(Gout[3], m1, Gout[0] * m0));
      R w1 = fma_(Gout[7], m2, fma_(Gout[4], m1, Gout[1] * m0));
      R w2 = fma_(Gout[8], m2, fma_(Gout[5], m1, Gout[2] * m0));
      d0 = d0 - fma_(Li[3], w2, fma_(Li[1], w1, Li[0] * w0));
      d1 = d1 - fma_(Li[4], w2, Li[2] * w1);
      d2 = d2 - Li[5] * w2;
      MPMPC_SERIAL_END(N + 1);
    }
    R n0(0.0), n1(0.0), n2(0.0);
    auto out_step = [&]() {
      R p0 = L::cdown(n0), p1 = L::cdown(n1), p2 = L::cdown(n2);
      n0 = fma_(Gout[2], p2, fma_(Gout[1], p1, fma_(Gout[0], p0, d0)));
      n1 = fma_(Gout[5], p2, fma_(Gout[4], p1, fma_(Gout[3], p0, d1)));
      n2 = fma_(Gout[8], p2, fma_(Gout[7], p1, fma_(Gout[6], p0, d2)));
    };
    {
      MPMPC_SERIAL_BEGIN();
      int s = 0;
      for (; s + 4 <= last + 1; s += 4) { out_step(); out_step(); out_step(); out_step(); }
      for (; s <= last; ++s) out_step();
      MPMPC_SERIAL_END(last + 1);
    }
    nu[0] = L::mirror(n0); nu[1] = L::mirror(n1); nu[2] = L::mirror(n2);
  }

  // [diag(1/hinv) Aeq'; Aeq -r I] [xt; nu] = [rx; req]
  MPMPC_HD void kkt_solve(const R rx[5], const R req[3], R xt[5], R nu[3]) const {
    R t[5], bv[3], s[5];
    MPMPC_UNROLL
    for (int j = 0; j < 5; ++j) t[j] = hinv[j] * rx[j];
    if constexpr (FQ) Hoff_add<0>(rx, t);
    Aeq_mul(t, bv);
    MPMPC_UNROLL
    for (int i = 0; i < 3; ++i) bv[i] = bv[i] - req[i];
    s_solve(bv, nu);
    AeqT_mul(nu, s);
    MPMPC_UNROLL
    for (int j = 0; j < 5; ++j) { s[j] = rx[j] - s[j]; xt[j] = hinv[j] * s[j]; }
    if constexpr (FQ) Hoff_add<0>(s, xt);
  }

  // ---- the same operators on the split layout (S = true: 3 entries per lane, see kSplit) or the plain one
  // ---- layouts of the certified polish (interior point, active set, phase 1)
  //   LAY_FULL      5 entries per lane: e_y, e_psi, t, v, kappa of the lane's stage; 3 equality rows
  //   LAY_SPLIT     kSplit (G = 64, N + 1 <= 32): lane k keeps the three states, lane k + 32 the two inputs (v, kappa, -)
  //   LAY_RED       REDUCED problem, 3 entries per lane: e_y, e_psi, kappa; 2 equality rows
  //   LAY_REDSPLIT  reduced and split: lane k keeps (e_y, e_psi), lane k + 32 keeps (kappa, -)
  // The reduced problem (template flag RED of the Solver): the time state t enters no other state's dynamics (column 2
  // of A_k is the unit vector) and the speed v drives t alone (column 0 of B_k), so when t carries neither cost nor
  // bound - Q[2] = QN[2] = 0, xmin[2] = -inf, xmax[2] = +inf: the reference's own tracking weights,
  // src/simulation.py:101-103,110-111 - the QP separates into  v_k = clip(v_ref_k, umin, hi_v_k)  in closed form, the
  // roll-forward of t, and the QP in (e_y, e_psi, kappa) with 2 x 2 blocks: the same optimum (the certificate and the
  // tests check the FULL problem's KKT conditions on the reassembled point) for about half the arithmetic.
  //   LAY_RED4      reduced problem PLUS the speed, 4 entries per lane: e_y, e_psi, kappa, v; 2 equality rows; the cost
  //                 carries ONE rank-one term  1/2 (rk_c' x)^2  on top of its diagonal (the terminal cost on the time state,
  //                 t_N being a linear functional of e_y and v: mpmpc_reduced_t.hpp) - every KKT solve is the reduced
  //                 2 x 2-block solve, a diagonal solve for the speeds and a Sherman-Morrison correction
  static constexpr int LAY_FULL = 0, LAY_SPLIT = 1, LAY_RED = 2, LAY_REDSPLIT = 3, LAY_RED4 = 4;
  template <int LAY> static constexpr int EN = LAY == LAY_FULL ? 5 : (LAY == LAY_REDSPLIT ? 2 : (LAY == LAY_RED4 ? 4 : 3));   // entries per lane
  template <int LAY> static constexpr int NR = LAY >= LAY_RED ? 2 : 3;                                 // equality rows per lane
  template <int LAY> static constexpr bool SPL = (LAY == LAY_SPLIT || LAY == LAY_REDSPLIT);
  // stage vector (5 entries) -> layout
  template <int LAY>
  MPMPC_HD void to_lay(const R v[5], R* o) const {
    if constexpr (LAY == LAY_FULL) {
      MPMPC_UNROLL
      for (int j = 0; j < 5; ++j) o[j] = v[j];
    } else if constexpr (LAY == LAY_SPLIT) {
      R t3 = L::from_lower(v[3]), t4 = L::from_lower(v[4]);
      o[0] = sel(sU, t3, v[0]); o[1] = sel(sU, t4, v[1]); o[2] = sel(sU, R(0.0), v[2]);
    } else if constexpr (LAY == LAY_RED) {
      o[0] = v[0]; o[1] = v[1]; o[2] = v[4];
    } else {
      R t4 = L::from_lower(v[4]);
      o[0] = sel(sU, t4, v[0]); o[1] = sel(sU, R(0.0), v[1]);
    }
  }
  // layout -> stage vector on the lanes that hold a stage; the entries a reduced layout does not carry (t, v) keep
  // what o[] holds already
  template <int LAY>
  MPMPC_HD void from_lay(const R* v, R o[5]) const {
    if constexpr (LAY == LAY_FULL) {
      MPMPC_UNROLL
      for (int j = 0; j < 5; ++j) o[j] = v[j];
    } else if constexpr (LAY == LAY_SPLIT) {
      o[0] = v[0]; o[1] = v[1]; o[2] = v[2];
      o[3] = L::from_upper(v[0]); o[4] = L::from_upper(v[1]);
    } else if constexpr (LAY == LAY_RED) {
      o[0] = v[0]; o[1] = v[1]; o[4] = v[2];
    } else {
      o[0] = v[0]; o[1] = v[1]; o[4] = L::from_upper(v[0]);
    }
  }
  // the same for masks (through 0 / 1 values: the exchange between the half-waves moves numbers)
  template <int LAY>
  MPMPC_HD void mask_to_lay(const Mk m[5], Mk* o) const {
    if constexpr (LAY == LAY_FULL) {
      MPMPC_UNROLL
      for (int j = 0; j < 5; ++j) o[j] = m[j];
    } else if constexpr (LAY == LAY_RED) {
      o[0] = m[0]; o[1] = m[1]; o[2] = m[4];
    } else {
      R v[5], w[EN<LAY>];
      MPMPC_UNROLL
      for (int j = 0; j < 5; ++j) v[j] = sel(m[j], R(1.0), R(0.0));
      to_lay<LAY>(v, w);
      MPMPC_UNROLL
      for (int e = 0; e < EN<LAY>; ++e) o[e] = w[e] > R(0.5);
    }
  }
  template <int LAY>
  MPMPC_HD void mask_from_lay(const Mk* m, Mk o[5]) const {
    if constexpr (LAY == LAY_FULL) {
      MPMPC_UNROLL
      for (int j = 0; j < 5; ++j) o[j] = m[j];
    } else if constexpr (LAY == LAY_RED) {
      o[0] = m[0]; o[1] = m[1]; o[4] = m[2];
    } else {
      R w[EN<LAY>], v[5] = {R(0.0), R(0.0), R(0.0), R(0.0), R(0.0)};
      MPMPC_UNROLL
      for (int e = 0; e < EN<LAY>; ++e) w[e] = sel(m[e], R(1.0), R(0.0));
      from_lay<LAY>(w, v);
      MPMPC_UNROLL
      for (int j = 0; j < 5; ++j) o[j] = v[j] > R(0.5);
    }
  }
  // which of the lane's entries exist in the layout
  template <int LAY>
  MPMPC_HD void valid_lay(Mk* vm) const {
    if constexpr (LAY == LAY_FULL) {
      MPMPC_UNROLL
      for (int j = 0; j < 5; ++j) vm[j] = valid[j];
    } else if constexpr (LAY == LAY_SPLIT) {
      vm[0] = val3[0]; vm[1] = val3[1]; vm[2] = val3[2];
    } else if constexpr (LAY == LAY_RED) {
      vm[0] = vx; vm[1] = vx; vm[2] = vu;
    } else {
      vm[0] = val3[0]; vm[1] = val3[2];          // upper lanes: kappa, nothing;  lower lanes: e_y, e_psi
    }
  }

  template <int LAY>
  MPMPC_HD void Aeq_mul_t(const R* v, R* r) const {
    if constexpr (LAY == LAY_FULL) {
      Aeq_mul(v, r);
    } else if constexpr (LAY == LAY_SPLIT) {
      // upper lanes form B u of their stage and hand it to the lower lane, which adds A x
      R c1 = L::from_upper(bU[0] * v[1]), c2 = L::from_upper(bU[1] * v[0]);
      R w[3];
      w[0] = fma_(a[1], v[1], a[0] * v[0]);
      w[1] = fma_(a[3], v[1], a[2] * v[0]) + c1;
      w[2] = fma_(a[5], v[2], a[4] * v[0]) + c2;
      MPMPC_UNROLL
      for (int i = 0; i < 3; ++i) r[i] = fma_(mI[i], v[i], L::up(w[i]));
    } else if constexpr (LAY == LAY_RED || LAY == LAY_RED4) {      // (the speed, entry 3 of LAY_RED4, is in no equality row)
      R w0 = fma_(a[1], v[1], a[0] * v[0]);
      R w1 = fma_(b[0], v[2], fma_(a[3], v[1], a[2] * v[0]));
      r[0] = fma_(mI[0], v[0], L::up(w0));
      r[1] = fma_(mI[1], v[1], L::up(w1));
    } else {
      R c1 = L::from_upper(bU[0] * v[0]);
      R w0 = fma_(a[1], v[1], a[0] * v[0]);
      R w1 = fma_(a[3], v[1], a[2] * v[0]) + c1;
      r[0] = fma_(mI[0], v[0], L::up(w0));
      r[1] = fma_(mI[1], v[1], L::up(w1));
    }
  }
  template <int LAY>
  MPMPC_HD void AeqT_mul_t(const R* nu, R* t) const {
    if constexpr (LAY == LAY_FULL) {
      AeqT_mul(nu, t);
    } else if constexpr (LAY == LAY_SPLIT) {
      R nd[3];
      MPMPC_UNROLL
      for (int i = 0; i < 3; ++i) nd[i] = L::down(nu[i]);
      R u1 = L::from_lower(nd[1]), u2 = L::from_lower(nd[2]);       // the upper lanes need nu of stage k + 1 too
      t[0] = fma_(bU[1], u2, fma_(a[4], nd[2], fma_(a[2], nd[1], fma_(a[0], nd[0], mI[0] * nu[0]))));
      t[1] = fma_(bU[0], u1, fma_(a[3], nd[1], fma_(a[1], nd[0], mI[1] * nu[1])));
      t[2] = fma_(a[5], nd[2], mI[2] * nu[2]);
    } else if constexpr (LAY == LAY_RED || LAY == LAY_RED4) {
      R nd0 = L::down(nu[0]), nd1 = L::down(nu[1]);
      t[0] = fma_(a[2], nd1, fma_(a[0], nd0, mI[0] * nu[0]));
      t[1] = fma_(a[3], nd1, fma_(a[1], nd0, mI[1] * nu[1]));
      t[2] = b[0] * nd1;
      if constexpr (LAY == LAY_RED4) t[3] = R(0.0);
    } else {
      R nd0 = L::down(nu[0]), nd1 = L::down(nu[1]);
      R u1 = L::from_lower(nd1);
      t[0] = fma_(bU[0], u1, fma_(a[2], nd1, fma_(a[0], nd0, mI[0] * nu[0])));       // (bU = 0 on the lower lanes, a = mI = 0 on the upper ones)
      t[1] = fma_(a[3], nd1, fma_(a[1], nd0, mI[1] * nu[1]));
    }
  }
  template <int LAY>
  MPMPC_HD void factor_t(const R* h, const R& r) {
    if constexpr (LAY == LAY_FULL) {
      factor(h, r);
    } else if constexpr (LAY == LAY_SPLIT) {
      MPMPC_UNROLL
      for (int e = 0; e < 3; ++e) hinv[e] = h[e];
      R w2b = L::from_upper((bU[0] * bU[0]) * h[1]), w5b = L::from_upper((bU[1] * bU[1]) * h[0]);
      w2b = sel(sU, R(0.0), w2b); w5b = sel(sU, R(0.0), w5b);
      if constexpr (FQ) {       // (on the input lanes hod[0] is the off-diagonal of the inputs' inverse block: dense_blocks<LAY_SPLIT>)
        const R w4b = sel(sU, R(0.0), L::from_upper((bU[0] * bU[1]) * hod[0]));
        // ... which belongs to the input lanes only: the state lanes' own hod is what factor_core reads
        factor_core(h, w2b, w5b, r, w4b);
      } else factor_core(h, w2b, w5b, r);
    } else if constexpr (LAY == LAY_RED) {
      MPMPC_UNROLL
      for (int e = 0; e < 3; ++e) hinv[e] = h[e];
      factor_core2(h, (b[0] * b[0]) * h[2], r);
    } else if constexpr (LAY == LAY_RED4) {
      MPMPC_UNROLL
      for (int e = 0; e < 4; ++e) hinv[e] = h[e];
      factor_core2(h, (b[0] * b[0]) * h[2], r);
      // Sherman-Morrison: one extra right-hand side per factorisation, u = inv(K0) [rk_c; 0], and 1 / (1 + rk_c'u)
      const R zero(0.0);
      const R rc[4] = {rk_c[0], zero, zero, rk_c[1]}, rq[2] = {zero, zero};
      R u[4], un[2];
      kkt_solve_base<LAY_RED4>(rc, rq, u, un);
      rk_g = R(1.0) / (R(1.0) + L::gsum(fma_(rk_c[1], u[3], rk_c[0] * u[0])));
      MPMPC_UNROLL
      for (int j = 0; j < 4; ++j) L::cold_put(RKS + j, u[j]);
      L::cold_put(RKS + 4, un[0]); L::cold_put(RKS + 5, un[1]);
      L::fence();
    } else {
      hinv[0] = h[0]; hinv[1] = h[1];
      R wb = L::from_upper((bU[0] * bU[0]) * h[0]);
      factor_core2(h, sel(sU, R(0.0), wb), r);
    }
  }
  // [diag(1/hinv) + rank-one, Aeq'; Aeq, -r I] [xt; nu] = [rx; req]  in the layout LAY
  template <int LAY>
  MPMPC_HD void kkt_solve_t(const R* rx, const R* req, R* xt, R* nu) const {
    kkt_solve_base<LAY>(rx, req, xt, nu);
    if constexpr (LAY == LAY_RED4) {
      // inv(K0 + c c') r = s - u (c's) / (1 + c'u)
      const R beta = rk_g * L::gsum(fma_(rk_c[1], xt[3], rk_c[0] * xt[0]));
      MPMPC_UNROLL
      for (int j = 0; j < 4; ++j) xt[j] = fma_(-beta, L::cold_get(RKS + j), xt[j]);
      nu[0] = fma_(-beta, L::cold_get(RKS + 4), nu[0]); nu[1] = fma_(-beta, L::cold_get(RKS + 5), nu[1]);
    }
  }
  // rk_c' x over the instance (LAY_RED4; x in that layout)
  MPMPC_HD R rank_one_dot(const R* x) const { return L::gsum(fma_(rk_c[1], x[3], rk_c[0] * x[0])); }
  template <int LAY>
  MPMPC_HD void kkt_solve_base(const R* rx, const R* req, R* xt, R* nu) const {
    constexpr int E = EN<LAY>, NQ = NR<LAY>;
    R t[E], bv[NQ], s[E];
    MPMPC_UNROLL
    for (int j = 0; j < E; ++j) t[j] = hinv[j] * rx[j];
    if constexpr (FQ) Hoff_add<LAY>(rx, t);
    Aeq_mul_t<LAY>(t, bv);
    MPMPC_UNROLL
    for (int i = 0; i < NQ; ++i) bv[i] = bv[i] - req[i];
    if constexpr (NQ == 3) s_solve(bv, nu); else s_solve2(bv, nu);
    AeqT_mul_t<LAY>(nu, s);
    MPMPC_UNROLL
    for (int j = 0; j < E; ++j) { s[j] = rx[j] - s[j]; xt[j] = hinv[j] * s[j]; }
    if constexpr (FQ) Hoff_add<LAY>(s, xt);
  }

  // ---- the reduced problem's block-tridiagonal Cholesky: factor_core / s_solve with 2 x 2 blocks (rows e_y, e_psi).
  // Same twisted elimination, same chain layout, same junction steps; Li = (i00, i10, i11), Gin / Gout 2 x 2 in the
  // first entries of the member arrays.
  //   A_k = [[a0, a1], [a2, a3]],  B_k = [0; b0]:   W = A H A' + B h_kappa B',   T = S_{k+1,k} = A H (-I)'
  MPMPC_HD void factor_core2(const R hx[2], const R& wb, const R& r) {
    const R* h = hx;
    R W[3], T[4], Dg[3], To[4];
    {
      R a0h = a[0] * h[0], a2h = a[2] * h[0], a1h = a[1] * h[1], a3h = a[3] * h[1];
      W[0] = fma_(a[1], a1h, a[0] * a0h);
      W[1] = fma_(a[3], a1h, a[2] * a0h);
      W[2] = fma_(a[3], a3h, a[2] * a2h) + wb;
      T[0] = a0h * mI[0]; T[1] = a1h * mI[1];
      T[2] = a2h * mI[0]; T[3] = a3h * mI[1];
    }
    MPMPC_UNROLL
    for (int i = 0; i < 3; ++i) Dg[i] = L::up(W[i]);
    Dg[0] = Dg[0] + fma_(mI[0] * mI[0], h[0], r);
    Dg[2] = Dg[2] + fma_(mI[1] * mI[1], h[1], r);
    {
      R Tu[4];
      MPMPC_UNROLL
      for (int i = 0; i < 4; ++i) Tu[i] = L::up(T[i]);
      const R zero(0.0);
      // coupling handed on: S_{k+1,k} = T going up, S_{k-1,k} = T_{k-1}' going down, nothing from mid
      To[0] = sel(down_chain, Tu[0], T[0]); To[1] = sel(down_chain, Tu[2], T[1]);
      To[2] = sel(down_chain, Tu[1], T[2]); To[3] = sel(down_chain, Tu[3], T[3]);
      MPMPC_UNROLL
      for (int i = 0; i < 4; ++i) To[i] = sel(is_mid, zero, To[i]);
    }
    MPMPC_UNROLL
    for (int i = 0; i < 3; ++i) Dg[i] = L::mirror(Dg[i]);
    MPMPC_UNROLL
    for (int i = 0; i < 4; ++i) To[i] = L::mirror(To[i]);
    if constexpr (kCR) { factor_cr2(Dg, To); return; }
    R M[4], Ls[4];
    MPMPC_UNROLL
    for (int i = 0; i < 4; ++i) M[i] = R(0.0);
    const int last = chain_steps();
    auto fstep = [&](bool junction) {
      R Mr[4];
      MPMPC_UNROLL
      for (int i = 0; i < 4; ++i) Mr[i] = L::cup(M[i]);
      R S00 = fma_(-Mr[1], Mr[1], fma_(-Mr[0], Mr[0], Dg[0]));
      R S10 = fma_(-Mr[3], Mr[1], fma_(-Mr[2], Mr[0], Dg[1]));
      R S11 = fma_(-Mr[3], Mr[3], fma_(-Mr[2], Mr[2], Dg[2]));
      if (junction) {
        R Mx[4];
        MPMPC_UNROLL
        for (int i = 0; i < 4; ++i) Mx[i] = sel(is_mid, L::down(L::mirror(M[i])), R(0.0));
        S00 = fma_(-Mx[1], Mx[1], fma_(-Mx[0], Mx[0], S00));
        S10 = fma_(-Mx[3], Mx[1], fma_(-Mx[2], Mx[0], S10));
        S11 = fma_(-Mx[3], Mx[3], fma_(-Mx[2], Mx[2], S11));
      }
      R i00 = rsqrt_(S00);
      R l10 = S10 * i00;
      R i11 = rsqrt_(fma_(-l10, l10, S11));
      R i10 = -(l10 * i00) * i11;
      Li[0] = i00; Li[1] = i10; Li[2] = i11;
      MPMPC_UNROLL
      for (int i = 0; i < 4; ++i) Ls[i] = Mr[i];
      // M = To * inv(L_kk)'
      M[0] = To[0] * i00; M[1] = fma_(To[1], i11, To[0] * i10);
      M[2] = To[2] * i00; M[3] = fma_(To[3], i11, To[2] * i10);
    };
    {
      MPMPC_SERIAL_BEGIN();
      int s = 0;
      for (; s + 2 <= last; s += 2) { fstep(false); fstep(false); }
      for (; s < last; ++s) fstep(false);
      fstep(true);
      MPMPC_SERIAL_END(last + 1);
    }
    //   inward    y_k  = inv(L_kk) b_k + Gin_k y_pred,        Gin_k  = -inv(L_kk) M_in
    //   outward   nu_k = inv(L_kk)' y_k + Gout_k nu_succ,     Gout_k = -inv(L_kk)' M_own'
    MPMPC_UNROLL
    for (int j = 0; j < 2; ++j) {
      Gin[0 + j] = -(Li[0] * Ls[0 + j]);
      Gin[2 + j] = -fma_(Li[2], Ls[2 + j], Li[1] * Ls[0 + j]);
    }
    MPMPC_UNROLL
    for (int j = 0; j < 2; ++j) {                                // Gout[i][j] = -sum_m Li[m][i] * M[j][m]
      R g0 = -fma_(Li[1], M[2 * j + 1], Li[0] * M[2 * j + 0]);
      R g1 = -(Li[2] * M[2 * j + 1]);
      Gout[0 + j] = sel(is_end, M[0 + j], g0);
      Gout[2 + j] = sel(is_end, M[2 + j], g1);
    }
  }
  // ---- cyclic reduction in Cholesky form (kCR).  In chain layout every chain is one row of 16 lanes, position p = 0 .. 15
  // along the chain, position 15 next to the meeting stage (row 0: mid itself; row 1: the end lane).  A Cholesky
  // factorisation may eliminate the stages of an SPD block-tridiagonal matrix in ANY order (a symmetric permutation): level
  // D = 1, 2, 4, 8 eliminates the positions p = 15 - D mod 2D - every second stage of what is left, counted from the row's
  // end - all at once.  Eliminating stage e with the current neighbours a = e - D, b = e + D:
  //     L_e L_e' = D_e,   Ua = inv(L_e) S_ea,   Ub = inv(L_e) S_eb,
  //     D_a -= Ua'Ua,   D_b -= Ub'Ub,   S_ba = -Ub'Ua   (a and b become neighbours at distance 2D),
  // so the blocks stay 2 x 2 and every lane is eliminated exactly once: it keeps inv(L_e) in Li and Ua, Ub in Gin, Gout.
  // After the four levels position 15 of each row holds the Schur complement of its chain; the end lane is eliminated, the
  // meeting stage takes its update (the junction of the sequential scheme), and is factored last.  The data exchanges are
  // in-row DPP shifts by D (one move per dword).  Backward stable like any Cholesky factorisation (it IS one) - unlike the
  // inverse-based parallel cyclic reduction of DESIGN.md 6a.  Depth 4 levels + junction instead of 16 dependent steps.
  // Cm: coupling of the lane's stage with its current LOWER neighbour, S_{p, p - D} (row-major 2 x 2).
  template <int D>
  MPMPC_HD void cr_level(R Dg[3], R Cm[4]) {
    // (written in the order that keeps the fewest blocks alive at once: the kernel lives on a 256-register budget.  inv(L) is
    //  masked ONCE - zero on the lanes that are not eliminated at this level - so that Ua, Ub come out zero there without a
    //  select each, and since every lane is eliminated at exactly one level the kept blocks are ACCUMULATED by exact additions
    //  of those zeros: one v_add_f64 per entry instead of two v_cndmask)
    const Mk E = L::template cr_elim<D>();
    const R zero(0.0);
    // Cholesky of the own block on every lane (used where the lane is eliminated at this level)
    R i00 = rsqrt_(Dg[0]);
    const R l10 = Dg[1] * i00;
    R i11 = rsqrt_(fma_(-l10, l10, Dg[2]));
    R i10 = -(l10 * i00) * i11;
    i00 = sel(E, i00, zero); i10 = sel(E, i10, zero); i11 = sel(E, i11, zero);
    Li[0] = Li[0] + i00; Li[1] = Li[1] + i10; Li[2] = Li[2] + i11;
    // Ub = inv(L) S_eb = inv(L) Cb',  Cb = S_be = the coupling lane e + D holds with its lower neighbour e
    R gb[4];
    {
      R Cb[4], Ub[4];
      MPMPC_UNROLL
      for (int i = 0; i < 4; ++i) Cb[i] = L::template rshl<D>(Cm[i]);
      Ub[0] = i00 * Cb[0]; Ub[1] = i00 * Cb[2];
      Ub[2] = fma_(i11, Cb[1], i10 * Cb[0]); Ub[3] = fma_(i11, Cb[3], i10 * Cb[2]);
      MPMPC_UNROLL
      for (int i = 0; i < 4; ++i) { Gout[i] = Gout[i] + Ub[i]; gb[i] = L::template rshr<D>(Ub[i]); }
    }
    // to the upper neighbour b (lane e + D):  D_b -= Ub'Ub
    Dg[0] = fma_(-gb[2], gb[2], fma_(-gb[0], gb[0], Dg[0]));
    Dg[1] = fma_(-gb[3], gb[2], fma_(-gb[1], gb[0], Dg[1]));
    Dg[2] = fma_(-gb[3], gb[3], fma_(-gb[1], gb[1], Dg[2]));
    // Ua = inv(L) S_ea = inv(L) Cm
    R ga[4];
    {
      R Ua[4];
      Ua[0] = i00 * Cm[0]; Ua[1] = i00 * Cm[1];
      Ua[2] = fma_(i11, Cm[2], i10 * Cm[0]); Ua[3] = fma_(i11, Cm[3], i10 * Cm[1]);
      MPMPC_UNROLL
      for (int i = 0; i < 4; ++i) Gin[i] = Gin[i] + Ua[i];
      // to the lower neighbour a (lane e - D):  D_a -= Ua'Ua
      {
        R fa[4];
        MPMPC_UNROLL
        for (int i = 0; i < 4; ++i) fa[i] = L::template rshl<D>(Ua[i]);
        Dg[0] = fma_(-fa[2], fa[2], fma_(-fa[0], fa[0], Dg[0]));
        Dg[1] = fma_(-fa[3], fa[2], fma_(-fa[1], fa[0], Dg[1]));
        Dg[2] = fma_(-fa[3], fa[3], fma_(-fa[1], fa[1], Dg[2]));
      }
      MPMPC_UNROLL
      for (int i = 0; i < 4; ++i) ga[i] = L::template rshr<D>(Ua[i]);
    }
    // ... and S_ba = -Ub'Ua: the coupling of b with its new lower neighbour a (a lane that survives this level has the
    // eliminated lane p - D below it: its old coupling is consumed)
    Cm[0] = sel(E, Cm[0], -fma_(gb[2], ga[2], gb[0] * ga[0]));
    Cm[1] = sel(E, Cm[1], -fma_(gb[2], ga[3], gb[0] * ga[1]));
    Cm[2] = sel(E, Cm[2], -fma_(gb[3], ga[2], gb[1] * ga[0]));
    Cm[3] = sel(E, Cm[3], -fma_(gb[3], ga[3], gb[1] * ga[1]));
  }
  // Dg: diagonal blocks, To: coupling S_{succ(p), p} with the chain successor, both in chain layout
  MPMPC_HD void factor_cr2(R Dg[3], const R To[4]) {
    const R zero(0.0);
    R Cm[4];
    MPMPC_UNROLL
    for (int i = 0; i < 4; ++i) Cm[i] = L::cup(To[i]);            // S_{p, p-1}: the predecessor's hand-on (zero at the chain heads)
    MPMPC_UNROLL
    for (int i = 0; i < 3; ++i) Li[i] = zero;
    // (position 15 is never eliminated by a level, so its Gout is free until the junction: the end lane's hand-on to the
    //  meeting stage waits there instead of in four more registers)
    MPMPC_UNROLL
    for (int i = 0; i < 4; ++i) { Gin[i] = zero; Gout[i] = sel(is_end, To[i], zero); }
    {
      // (census: every stage is eliminated at exactly ONE of the four levels - its factorisation and the updates it sends
      //  are one level's work - although all lanes execute all four: like one step per stage of a serial sweep)
      MPMPC_SERIAL_BEGIN();
      cr_level<1>(Dg, Cm);
      cr_level<2>(Dg, Cm);
      cr_level<4>(Dg, Cm);
      cr_level<8>(Dg, Cm);
      MPMPC_SERIAL_END(4);
    }
    if constexpr (kCR32) {
      // ---- 32-lane chains: each chain is two rows.  The four levels have eliminated the interiors of all four rows; the
      // first row's survivor X (position 15 of rows 0 / 2) is still coupled with the second row's Y (position 31 of the chain:
      // the meeting stage / the end lane) through the fill S_YX the levels left in Y's Cm, and it has not yet received the
      // updates of the second row's lanes that were eliminated with X as their lower neighbour (positions 0, 1, 3, 7 of rows
      // 1 / 3, one per level - their Ua is in their Gin, and the row shift that carries a level's update stops at the row's
      // edge):  D_X -= sum Ua'Ua, a sum over four lanes of the next row.  Then X is eliminated:  L_X L_X' = D_X,
      // U = inv(L_X) S_XY waits in X's Gout (free: no level eliminates position 15),  D_Y -= U'U.
      MPMPC_SERIAL_BEGIN();
      const Mk spec = L::cr_special(), isX = L::cr_low15(), isY = is_mid | is_end;
      R w0 = sel(spec, fma_(Gin[2], Gin[2], Gin[0] * Gin[0]), zero), w1 = sel(spec, fma_(Gin[3], Gin[2], Gin[1] * Gin[0]), zero),
        w2 = sel(spec, fma_(Gin[3], Gin[3], Gin[1] * Gin[1]), zero);
      // positions 0, 1, 3, 7 summed into position 0 of the row, then one lane down: position 15 of the row below
      w0 = w0 + L::template rshl<1>(w0); w1 = w1 + L::template rshl<1>(w1); w2 = w2 + L::template rshl<1>(w2);
      w0 = w0 + L::template rshl<3>(w0); w1 = w1 + L::template rshl<3>(w1); w2 = w2 + L::template rshl<3>(w2);
      w0 = w0 + L::template rshl<7>(w0); w1 = w1 + L::template rshl<7>(w1); w2 = w2 + L::template rshl<7>(w2);
      Dg[0] = Dg[0] - sel(isX, L::down(w0), zero); Dg[1] = Dg[1] - sel(isX, L::down(w1), zero); Dg[2] = Dg[2] - sel(isX, L::down(w2), zero);
      R i00 = rsqrt_(Dg[0]);
      const R l10 = Dg[1] * i00;
      R i11 = rsqrt_(fma_(-l10, l10, Dg[2]));
      R i10 = -(l10 * i00) * i11;
      i00 = sel(isX, i00, zero); i10 = sel(isX, i10, zero); i11 = sel(isX, i11, zero);
      Li[0] = Li[0] + i00; Li[1] = Li[1] + i10; Li[2] = Li[2] + i11;
      R Cb[4], Ub[4], gb[4];
      MPMPC_UNROLL
      for (int i = 0; i < 4; ++i) Cb[i] = L::from_odd_row(Cm[i]);
      Ub[0] = i00 * Cb[0]; Ub[1] = i00 * Cb[2];
      Ub[2] = fma_(i11, Cb[1], i10 * Cb[0]); Ub[3] = fma_(i11, Cb[3], i10 * Cb[2]);
      MPMPC_UNROLL
      for (int i = 0; i < 4; ++i) { Gout[i] = Gout[i] + Ub[i]; gb[i] = sel(isY, L::from_even_row(Ub[i]), zero); }
      Dg[0] = fma_(-gb[2], gb[2], fma_(-gb[0], gb[0], Dg[0]));
      Dg[1] = fma_(-gb[3], gb[2], fma_(-gb[1], gb[0], Dg[1]));
      Dg[2] = fma_(-gb[3], gb[3], fma_(-gb[1], gb[1], Dg[2]));
      MPMPC_SERIAL_END(N + 1);
    }
    // position 15 of each row: the end lane (row 1) is eliminated, its block M = S_{mid,end} inv(L_end)' goes to the meeting
    // stage (row 0), which is factored last.  (Chains shorter than a row: the positions without a stage carry identity-like
    // blocks and zero couplings, they factor harmlessly.)
    const Mk last = is_mid | is_end;
    MPMPC_SERIAL_BEGIN();                  // (census: the junction is useful on its two lanes only)
    {
      const R i00 = rsqrt_(Dg[0]);
      const R l10 = Dg[1] * i00;
      const R i11 = rsqrt_(fma_(-l10, l10, Dg[2]));
      const R i10 = -(l10 * i00) * i11;
      R M[4];
      M[0] = Gout[0] * i00; M[1] = fma_(Gout[1], i11, Gout[0] * i10);
      M[2] = Gout[2] * i00; M[3] = fma_(Gout[3], i11, Gout[2] * i10);
      Li[0] = sel(is_end, i00, Li[0]); Li[1] = sel(is_end, i10, Li[1]); Li[2] = sel(is_end, i11, Li[2]);
      MPMPC_UNROLL
      for (int i = 0; i < 4; ++i) Gout[i] = sel(is_end, M[i], Gout[i]);          // the end lane keeps M_own (as in the sequential scheme)
      R Mx[4];
      MPMPC_UNROLL
      for (int i = 0; i < 4; ++i) Mx[i] = sel(is_mid, L::down(L::mirror(sel(is_end, M[i], zero))), zero);
      Dg[0] = fma_(-Mx[1], Mx[1], fma_(-Mx[0], Mx[0], Dg[0]));
      Dg[1] = fma_(-Mx[3], Mx[1], fma_(-Mx[2], Mx[0], Dg[1]));
      Dg[2] = fma_(-Mx[3], Mx[3], fma_(-Mx[2], Mx[2], Dg[2]));
    }
    {
      const R i00 = rsqrt_(Dg[0]);
      const R l10 = Dg[1] * i00;
      const R i11 = rsqrt_(fma_(-l10, l10, Dg[2]));
      const R i10 = -(l10 * i00) * i11;
      Li[0] = sel(is_mid, i00, Li[0]); Li[1] = sel(is_mid, i10, Li[1]); Li[2] = sel(is_mid, i11, Li[2]);
    }
    MPMPC_SERIAL_END(N + 1);
    (void)last;
  }
  // forward / backward substitution of one cyclic-reduction level
  template <int D>
  MPMPC_HD void cr_forward(R& b0, R& b1, R& y0, R& y1) const {
    const Mk E = L::template cr_elim<D>();
    const R zero(0.0);
    const R t0 = Li[0] * b0, t1 = fma_(Li[2], b1, Li[1] * b0);            // y = inv(L) b
    const R e0 = sel(E, t0, zero), e1 = sel(E, t1, zero);
    y0 = y0 + e0; y1 = y1 + e1;                                            // (each lane is eliminated once: an exact accumulation)
    // b_a -= Ua' y,  b_b -= Ub' y
    const R pa0 = fma_(Gin[2], e1, Gin[0] * e0), pa1 = fma_(Gin[3], e1, Gin[1] * e0);
    const R pb0 = fma_(Gout[2], e1, Gout[0] * e0), pb1 = fma_(Gout[3], e1, Gout[1] * e0);
    b0 = b0 - L::template rshl<D>(pa0) - L::template rshr<D>(pb0);
    b1 = b1 - L::template rshl<D>(pa1) - L::template rshr<D>(pb1);
  }
  template <int D>
  MPMPC_HD void cr_backward(const R& y0, const R& y1, R& n0, R& n1) const {
    const Mk E = L::template cr_elim<D>();
    // nu_e = inv(L_e)' (y_e - Ua nu_a - Ub nu_b),  nu_a from lane e - D, nu_b from lane e + D
    const R a0 = L::template rshr<D>(n0), a1 = L::template rshr<D>(n1), c0 = L::template rshl<D>(n0), c1 = L::template rshl<D>(n1);
    const R r0 = y0 - fma_(Gin[1], a1, Gin[0] * a0) - fma_(Gout[1], c1, Gout[0] * c0);
    const R r1 = y1 - fma_(Gin[3], a1, Gin[2] * a0) - fma_(Gout[3], c1, Gout[2] * c0);
    n0 = sel(E, fma_(Li[1], r1, Li[0] * r0), n0);
    n1 = sel(E, Li[2] * r1, n1);
  }
  MPMPC_HD void s_solve_cr2(const R bv[2], R nu[2]) const {
    const R zero(0.0);
    R b0 = sel(vxc, L::mirror(bv[0]), zero), b1 = sel(vxc, L::mirror(bv[1]), zero);
    R y0(0.0), y1(0.0);
    {
      MPMPC_SERIAL_BEGIN();
      cr_forward<1>(b0, b1, y0, y1);
      cr_forward<2>(b0, b1, y0, y1);
      cr_forward<4>(b0, b1, y0, y1);
      cr_forward<8>(b0, b1, y0, y1);
      MPMPC_SERIAL_END(4);
    }
    [[maybe_unused]] Mk spec = L::mfalse(), isX = L::mfalse();
    if constexpr (kCR32) {
      // the first rows' survivors X (see factor_cr2):  b_X -= sum Ua'y over the four lanes of the next row that were eliminated
      // against X;  y_X = inv(L_X) b_X;  b_Y -= U'y_X
      MPMPC_SERIAL_BEGIN();
      spec = L::cr_special(); isX = L::cr_low15();
      R c0 = sel(spec, fma_(Gin[2], y1, Gin[0] * y0), zero), c1 = sel(spec, fma_(Gin[3], y1, Gin[1] * y0), zero);
      c0 = c0 + L::template rshl<1>(c0); c1 = c1 + L::template rshl<1>(c1);
      c0 = c0 + L::template rshl<3>(c0); c1 = c1 + L::template rshl<3>(c1);
      c0 = c0 + L::template rshl<7>(c0); c1 = c1 + L::template rshl<7>(c1);
      const R bx0 = b0 - L::down(c0), bx1 = b1 - L::down(c1);
      const R yx0 = sel(isX, Li[0] * bx0, zero), yx1 = sel(isX, fma_(Li[2], bx1, Li[1] * bx0), zero);
      y0 = y0 + yx0; y1 = y1 + yx1;
      const R p0 = fma_(Gout[2], yx1, Gout[0] * yx0), p1 = fma_(Gout[3], yx1, Gout[1] * yx0);      // U'y_X on the X lanes
      const Mk isY = is_mid | is_end;
      b0 = b0 - sel(isY, L::from_even_row(p0), zero); b1 = b1 - sel(isY, L::from_even_row(p1), zero);
      MPMPC_SERIAL_END(N + 1);
    }
    // junction: y_end = inv(L_end) b_end;  b_mid -= M y_end;  y_mid = inv(L_mid) b_mid;  nu_mid = inv(L_mid)' y_mid;
    //           nu_end = inv(L_end)' (y_end - M' nu_mid)
    MPMPC_SERIAL_BEGIN();                                                           // (census: useful on the two lanes of the junction only)
    const R ye0 = Li[0] * b0, ye1 = fma_(Li[2], b1, Li[1] * b0);                   // valid on the end lane (and, pre-update, on mid)
    const R q0 = fma_(Gout[1], ye1, Gout[0] * ye0), q1 = fma_(Gout[3], ye1, Gout[2] * ye0);      // M y_end on the end lane
    const R qm0 = sel(is_mid, L::down(L::mirror(sel(is_end, q0, zero))), zero), qm1 = sel(is_mid, L::down(L::mirror(sel(is_end, q1, zero))), zero);
    const R bm0 = b0 - qm0, bm1 = b1 - qm1;
    const R ym0 = Li[0] * bm0, ym1 = fma_(Li[2], bm1, Li[1] * bm0);
    const R nm0 = fma_(Li[1], ym1, Li[0] * ym0), nm1 = Li[2] * ym1;               // nu of the meeting stage (on mid)
    // to the end lane: M' nu_mid
    const R me0 = sel(is_end, L::mirror(L::up(sel(is_mid, nm0, zero))), zero), me1 = sel(is_end, L::mirror(L::up(sel(is_mid, nm1, zero))), zero);
    const R re0 = ye0 - fma_(Gout[2], me1, Gout[0] * me0), re1 = ye1 - fma_(Gout[3], me1, Gout[1] * me0);
    const R ne0 = fma_(Li[1], re1, Li[0] * re0), ne1 = Li[2] * re1;
    R n0 = sel(is_mid, nm0, sel(is_end, ne0, zero)), n1 = sel(is_mid, nm1, sel(is_end, ne1, zero));
    MPMPC_SERIAL_END(N + 1);
    if constexpr (kCR32) {
      // nu_X = inv(L_X)' (y_X - U nu_Y);  the second rows' lanes that were eliminated against X take  -Ua nu_X  into their y
      // before the levels run backwards (each of them is eliminated at exactly one level)
      MPMPC_SERIAL_BEGIN();
      const R c0 = L::from_odd_row(n0), c1 = L::from_odd_row(n1);
      const R r0 = y0 - fma_(Gout[1], c1, Gout[0] * c0), r1 = y1 - fma_(Gout[3], c1, Gout[2] * c0);
      n0 = sel(isX, fma_(Li[1], r1, Li[0] * r0), n0);
      n1 = sel(isX, Li[2] * r1, n1);
      const R x0 = L::bcast15(n0), x1 = L::bcast15(n1);
      y0 = y0 - sel(spec, fma_(Gin[1], x1, Gin[0] * x0), zero);
      y1 = y1 - sel(spec, fma_(Gin[3], x1, Gin[2] * x0), zero);
      MPMPC_SERIAL_END(N + 1);
    }
    {
      MPMPC_SERIAL_BEGIN();
      cr_backward<8>(y0, y1, n0, n1);
      cr_backward<4>(y0, y1, n0, n1);
      cr_backward<2>(y0, y1, n0, n1);
      cr_backward<1>(y0, y1, n0, n1);
      MPMPC_SERIAL_END(4);
    }
    nu[0] = L::mirror(n0); nu[1] = L::mirror(n1);
  }

  MPMPC_HD void s_solve2(const R bv[2], R nu[2]) const {
    if constexpr (kCR) { s_solve_cr2(bv, nu); return; }
    R b0 = sel(vxc, L::mirror(bv[0]), R(0.0)), b1 = sel(vxc, L::mirror(bv[1]), R(0.0));
    R c0 = Li[0] * b0;
    R c1 = fma_(Li[2], b1, Li[1] * b0);
    const int last = chain_steps();
    R y0(0.0), y1(0.0);
    auto in_step = [&]() {
      R p0 = L::cup(y0), p1 = L::cup(y1);
      y0 = fma_(Gin[1], p1, fma_(Gin[0], p0, c0));
      y1 = fma_(Gin[3], p1, fma_(Gin[2], p0, c1));
    };
    {
      MPMPC_SERIAL_BEGIN();
      int s = 0;
      for (; s + 4 <= last; s += 4) { in_step(); in_step(); in_step(); in_step(); }
      for (; s < last; ++s) in_step();
      MPMPC_SERIAL_END(last);
    }
    {
      // inward junction: the end lane forms M_own y, mid takes it on top of its chain input
      MPMPC_SERIAL_BEGIN();
      R t0 = fma_(Gout[1], y1, Gout[0] * y0);
      R t1 = fma_(Gout[3], y1, Gout[2] * y0);
      const R zero(0.0);
      t0 = sel(is_mid, L::down(L::mirror(t0)), zero);
      t1 = sel(is_mid, L::down(L::mirror(t1)), zero);
      R e0 = c0 - Li[0] * t0;
      R e1 = c1 - fma_(Li[2], t1, Li[1] * t0);
      R p0 = L::cup(y0), p1 = L::cup(y1);
      y0 = fma_(Gin[1], p1, fma_(Gin[0], p0, e0));
      y1 = fma_(Gin[3], p1, fma_(Gin[2], p0, e1));
      MPMPC_SERIAL_END(N + 1);
    }
    R d0 = fma_(Li[1], y1, Li[0] * y0);
    R d1 = Li[2] * y1;
    {
      // outward junction: nu of mid is final; the end lane takes it through M_own'
      MPMPC_SERIAL_BEGIN();
      const R zero(0.0);
      R m0 = sel(is_end, L::mirror(L::up(d0)), zero);
      R m1 = sel(is_end, L::mirror(L::up(d1)), zero);
      R w0 = fma_(Gout[2], m1, Gout[0] * m0);
      R w1 = fma_(Gout[3], m1, Gout[1] * m0);
      d0 = d0 - fma_(Li[1], w1, Li[0] * w0);
      d1 = d1 - Li[2] * w1;
      MPMPC_SERIAL_END(N + 1);
    }
    R n0(0.0), n1(0.0);
    auto out_step = [&]() {
      R p0 = L::cdown(n0), p1 = L::cdown(n1);
      n0 = fma_(Gout[1], p1, fma_(Gout[0], p0, d0));
      n1 = fma_(Gout[3], p1, fma_(Gout[2], p0, d1));
    };
    {
      MPMPC_SERIAL_BEGIN();
      int s = 0;
      for (; s + 4 <= last + 1; s += 4) { out_step(); out_step(); out_step(); out_step(); }
      for (; s <= last; ++s) out_step();
      MPMPC_SERIAL_END(last + 1);
    }
    nu[0] = L::mirror(n0); nu[1] = L::mirror(n1);
  }

  MPMPC_HD void admm_factor(double sigma) {
    R h[5], Hd[5];
    MPMPC_UNROLL
    for (int j = 0; j < 5; ++j) { Hd[j] = p[j] + R(sigma) + (g[j] * g[j]) * rb[j]; h[j] = R(1.0) / Hd[j]; }
    dense_blocks(Hd, h);
    factor(h, rinv_eq);
  }

  // ======================================================================== ADMM (OSQP)
  struct Info {
    R pri, dua, n_z, n_Ax, n_q, n_Aty, n_Px;       // unscaled norms for the termination test
    R s_rp, s_rd, s_z, s_Ax, s_q, s_Aty, s_Px;     // scaled norms for the rho estimate
  };
  MPMPC_HD void info(Info& o) const {
    R Axe[3], Aty[5];
    Aeq_mul(x, Axe);
    AeqT_mul(yeq, Aty);
    R pri(0.0), nz(0.0), nAx(0.0), srp(0.0), sz(0.0), sAx(0.0);
    MPMPC_UNROLL
    for (int i = 0; i < 3; ++i) {
      R ei = R(1.0) / Eeq[i];
      R rp = Axe[i] - zeq[i];
      pri = max_(pri, sel(vx, abs_(ei * rp), R(0.0)));
      nz = max_(nz, sel(vx, abs_(ei * zeq[i]), R(0.0)));
      nAx = max_(nAx, sel(vx, abs_(ei * Axe[i]), R(0.0)));
      srp = max_(srp, sel(vx, abs_(rp), R(0.0)));
      sz = max_(sz, sel(vx, abs_(zeq[i]), R(0.0)));
      sAx = max_(sAx, sel(vx, abs_(Axe[i]), R(0.0)));
    }
    R dua(0.0), nq(0.0), nAty(0.0), nPx(0.0), srd(0.0), sq(0.0), sAty(0.0), sPx(0.0);
    R Pod[5] = {R(0.0), R(0.0), R(0.0), R(0.0), R(0.0)};          // off-diagonal part of P x (FQ)
    if constexpr (FQ) Poff_add<0>(x, Pod);
    MPMPC_UNROLL
    for (int j = 0; j < 5; ++j) {
      R ei = R(1.0) / Eb[j], di = R(1.0) / D[j];
      R Axb = g[j] * x[j];
      R rp = Axb - zb[j];
      pri = max_(pri, sel(valid[j], abs_(ei * rp), R(0.0)));
      nz = max_(nz, sel(valid[j], abs_(ei * zb[j]), R(0.0)));
      nAx = max_(nAx, sel(valid[j], abs_(ei * Axb), R(0.0)));
      srp = max_(srp, sel(valid[j], abs_(rp), R(0.0)));
      sz = max_(sz, sel(valid[j], abs_(zb[j]), R(0.0)));
      sAx = max_(sAx, sel(valid[j], abs_(Axb), R(0.0)));
      R aty = fma_(g[j], yb[j], Aty[j]);
      R Px = p[j] * x[j];
      if constexpr (FQ) Px = Px + Pod[j];
      R rd = Px + q[j] + aty;
      dua = max_(dua, sel(valid[j], abs_(di * rd), R(0.0)));
      nq = max_(nq, sel(valid[j], abs_(di * q[j]), R(0.0)));
      nAty = max_(nAty, sel(valid[j], abs_(di * aty), R(0.0)));
      nPx = max_(nPx, sel(valid[j], abs_(di * Px), R(0.0)));
      srd = max_(srd, sel(valid[j], abs_(rd), R(0.0)));
      sq = max_(sq, sel(valid[j], abs_(q[j]), R(0.0)));
      sAty = max_(sAty, sel(valid[j], abs_(aty), R(0.0)));
      sPx = max_(sPx, sel(valid[j], abs_(Px), R(0.0)));
    }
    R cinv = R(1.0) / c;
    o.pri = L::gmax(pri); o.n_z = L::gmax(nz); o.n_Ax = L::gmax(nAx);
    o.dua = cinv * L::gmax(dua); o.n_q = cinv * L::gmax(nq); o.n_Aty = cinv * L::gmax(nAty); o.n_Px = cinv * L::gmax(nPx);
    o.s_rp = L::gmax(srp); o.s_z = L::gmax(sz); o.s_Ax = L::gmax(sAx);
    o.s_rd = L::gmax(srd); o.s_q = L::gmax(sq); o.s_Aty = L::gmax(sAty); o.s_Px = L::gmax(sPx);
  }

  // unscaled primal residual only (what the early polish attempt wants to know about the ADMM point)
  MPMPC_HD R primal_residual() const {
    R Axe[3], pri(0.0);
    Aeq_mul(x, Axe);
    MPMPC_UNROLL
    for (int i = 0; i < 3; ++i) pri = max_(pri, sel(vx, abs_((R(1.0) / Eeq[i]) * (Axe[i] - zeq[i])), R(0.0)));
    MPMPC_UNROLL
    for (int j = 0; j < 5; ++j) pri = max_(pri, sel(valid[j], abs_((R(1.0) / Eb[j]) * (g[j] * x[j] - zb[j])), R(0.0)));
    return L::gmax(pri);
  }

  // OSQP is_primal_infeasible() on the last dual step
  MPMPC_HD Mk primal_infeasible(double eps) const {
    R nrm, lhs, m;
    farkas_values(nrm, lhs, m);
    return (nrm > R(eps)) & (lhs < R(-eps) * nrm) & (m < R(eps) * nrm);
  }
  // the three numbers of that test for the ray in the cold slots COLD_DYEQ / COLD_DYB: |E dy|_inf, the support
  // u'max(dy,0) + l'min(dy,0), and |inv(D) A'dy|_inf
  MPMPC_HD void farkas_values(R& nrm_out, R& lhs_out, R& m_out) const {
    R nrm(0.0), lhs(0.0), pd[5], dyeq[3];
    MPMPC_UNROLL
    for (int i = 0; i < 3; ++i) dyeq[i] = L::cold_get(COLD_DYEQ + i);
    MPMPC_UNROLL
    for (int i = 0; i < 3; ++i) {
      nrm = max_(nrm, sel(vx, abs_(Eeq[i] * dyeq[i]), R(0.0)));
      lhs = lhs + sel(vx, leq[i] * dyeq[i], R(0.0));      // u*max(dy,0) + l*min(dy,0) with l = u
    }
    MPMPC_UNROLL
    for (int j = 0; j < 5; ++j) {
      Mk lo_inf = lb[j] < R(-INF_BOUND), up_inf = ub[j] > R(INF_BOUND);
      R d = L::cold_get(COLD_DYB + j);
      d = sel(up_inf & lo_inf, R(0.0), sel(up_inf, min_(d, R(0.0)), sel(lo_inf, max_(d, R(0.0)), d)));
      pd[j] = d;
      nrm = max_(nrm, sel(valid[j], abs_(Eb[j] * d), R(0.0)));
      lhs = lhs + sel(valid[j], ub[j] * max_(d, R(0.0)) + lb[j] * min_(d, R(0.0)), R(0.0));
    }
    nrm = L::gmax(nrm);
    lhs = L::gsum(lhs);
    R At[5];
    AeqT_mul(dyeq, At);
    R m(0.0);
    MPMPC_UNROLL
    for (int j = 0; j < 5; ++j) m = max_(m, sel(valid[j], abs_(fma_(g[j], pd[j], At[j]) / D[j]), R(0.0)));
    m = L::gmax(m);
    nrm_out = nrm; lhs_out = lhs; m_out = m;
  }

  // OSQP is_dual_infeasible() on the last primal step
  MPMPC_HD Mk dual_infeasible(double eps) const {
    R dx[5], nrm(0.0), qdx(0.0), pm(0.0);
    MPMPC_UNROLL
    for (int j = 0; j < 5; ++j) dx[j] = x[j] - L::cold_get(COLD_XPREV + j);
    R Pdx[5];
    MPMPC_UNROLL
    for (int j = 0; j < 5; ++j) Pdx[j] = p[j] * dx[j];
    if constexpr (FQ) Poff_add<0>(dx, Pdx);
    MPMPC_UNROLL
    for (int j = 0; j < 5; ++j) {
      nrm = max_(nrm, sel(valid[j], abs_(D[j] * dx[j]), R(0.0)));
      qdx = qdx + sel(valid[j], q[j] * dx[j], R(0.0));
      pm = max_(pm, sel(valid[j], abs_(Pdx[j] / D[j]), R(0.0)));
    }
    nrm = L::gmax(nrm); qdx = L::gsum(qdx); pm = L::gmax(pm);
    R Adx[3];
    Aeq_mul(dx, Adx);
    R thr = R(eps) * nrm;
    Mk bad = L::mfalse();
    MPMPC_UNROLL
    for (int i = 0; i < 3; ++i) bad = bad | (vx & (abs_(Adx[i] / Eeq[i]) > thr));
    MPMPC_UNROLL
    for (int j = 0; j < 5; ++j) {
      R v = (g[j] * dx[j]) / Eb[j];
      Mk lo_inf = lb[j] < R(-INF_BOUND), up_inf = ub[j] > R(INF_BOUND);
      bad = bad | (valid[j] & ((!up_inf & (v > thr)) | (!lo_inf & (v < -thr))));
    }
    bad = L::gany(bad);
    return (nrm > R(eps)) & (qdx < -(c * R(eps)) * nrm) & (pm < (c * R(eps)) * nrm) & !bad;
  }

  MPMPC_HD I check(const Info& o, const SolverParams& st, bool approximate) const {
    const double ea = approximate ? st.eps_abs10 : st.eps_abs, er = approximate ? st.eps_rel10 : st.eps_rel;
    R eps_prim = R(ea) + R(er) * max_(o.n_z, o.n_Ax);
    R eps_dual = R(ea) + R(er) * max_(max_(o.n_q, o.n_Aty), o.n_Px);
    Mk prim_ok = o.pri < eps_prim, dual_ok = o.dua < eps_dual;
    Mk pinf = !prim_ok & primal_infeasible(approximate ? st.eps_prim_inf10 : st.eps_prim_inf);
    Mk dinf = !dual_ok & dual_infeasible(approximate ? st.eps_dual_inf10 : st.eps_dual_inf);
    I stt(MPMPC_UNSOLVED);
    stt = seli(dinf, I(MPMPC_DUAL_INFEASIBLE), stt);
    stt = seli(pinf, I(MPMPC_PRIMAL_INFEASIBLE), stt);
    stt = seli(prim_ok & dual_ok, I(approximate ? MPMPC_SOLVED_INACCURATE : MPMPC_SOLVED), stt);
    return stt;
  }

  // Scalings D, E, the cost diagonal and the previous iterate are only read at termination checks:
  // they live in cold storage between checks.
  static constexpr int COLD_XPREV = 18, COLD_DYEQ = 23, COLD_DYB = 26;
  MPMPC_HD static void put_delta(const Mk& on, int slot, const R& v) {
    if constexpr (L::per_wave == 1) L::cold_put(slot, v); else L::cold_put(slot, sel(on, v, L::cold_get(slot)));
  }
  MPMPC_HD void park_check_data() const {
    MPMPC_UNROLL
    for (int j = 0; j < 5; ++j) { L::cold_put(j, D[j]); L::cold_put(5 + j, Eb[j]); L::cold_put(10 + j, p[j]); }
    MPMPC_UNROLL
    for (int i = 0; i < 3; ++i) L::cold_put(15 + i, Eeq[i]);
    L::fence();
  }
  MPMPC_HD void unpark_check_data() {
    L::fence();
    MPMPC_UNROLL
    for (int j = 0; j < 5; ++j) { D[j] = L::cold_get(j); Eb[j] = L::cold_get(5 + j); p[j] = L::cold_get(10 + j); }
    MPMPC_UNROLL
    for (int i = 0; i < 3; ++i) Eeq[i] = L::cold_get(15 + i);
  }

  // Cold-started ADMM on the instances selected by `which`, at most `limit` iterations.  Instances
  // still running at the limit (only possible when limit < max_iter) keep status UNSOLVED.
  MPMPC_HD void admm(const SolverParams& st, const Mk& which, int limit) {
    const R zero(0.0);
    MPMPC_UNROLL
    for (int j = 0; j < 5; ++j) {
      x[j] = keep(which, zero, x[j]); zb[j] = keep(which, zero, zb[j]); yb[j] = keep(which, zero, yb[j]);
      put_delta(which, COLD_XPREV + j, zero); put_delta(which, COLD_DYB + j, zero);
    }
    MPMPC_UNROLL
    for (int i = 0; i < 3; ++i) {
      zeq[i] = keep(which, zero, zeq[i]); yeq[i] = keep(which, zero, yeq[i]); put_delta(which, COLD_DYEQ + i, zero);
    }
    status = keepi(which, I(MPMPC_UNSOLVED), status);
    iters = keepi(which, I(0), iters);
    ipm_iters = keepi(which, I(0), ipm_iters);
    polished = keepi(which, I(0), polished);
    set_rho(R(st.rho));
    admm_factor(st.sigma);
    const R alpha(st.alpha), oma(st.one_minus_alpha), sigma(st.sigma);
    Mk active = which;
    const bool full = limit >= st.max_iter;
    if (limit > st.max_iter) limit = st.max_iter;
    Info nf;
    park_check_data();
    for (int it = 1; it <= limit; ++it) {
      if (!L::wany(active)) break;
      // ---- one ADMM step (OSQP update_xz_tilde / update_x / update_z / update_y)
      R rx[5], req[3], xt[5], nu[3];
      MPMPC_UNROLL
      for (int j = 0; j < 5; ++j) rx[j] = fma_(g[j], fma_(rb[j], zb[j], -yb[j]), fma_(sigma, x[j], -q[j]));
      MPMPC_UNROLL
      for (int i = 0; i < 3; ++i) req[i] = fma_(-yeq[i], rinv_eq, zeq[i]);
      kkt_solve(rx, req, xt, nu);
      // the last step's (dx, dy) feed the infeasibility tests only: keep them where a check follows
      const bool can_check = st.check_termination > 0 && (it % st.check_termination) == 0;
      const bool can_adapt = st.adaptive_rho && st.adaptive_rho_interval > 0 && (it % st.adaptive_rho_interval) == 0;
      const bool want_delta = can_check || it == limit;
      if (want_delta) {
        MPMPC_UNROLL
        for (int j = 0; j < 5; ++j) put_delta(active, COLD_XPREV + j, x[j]);
      }
      R dyb_n[5], dyeq_n[3];
      MPMPC_UNROLL
      for (int j = 0; j < 5; ++j) {
        R xn = fma_(alpha, xt[j], oma * x[j]);
        R zr = fma_(alpha, g[j] * xt[j], oma * zb[j]);
        R zn = min_(max_(fma_(yb[j], rbinv[j], zr), lb[j]), ub[j]);
        dyb_n[j] = rb[j] * (zr - zn);
        x[j] = keep(active, xn, x[j]);
        zb[j] = keep(active, zn, zb[j]);
        yb[j] = keep(active, yb[j] + dyb_n[j], yb[j]);
      }
      MPMPC_UNROLL
      for (int i = 0; i < 3; ++i) {
        R zt = fma_(nu[i] - yeq[i], rinv_eq, zeq[i]);
        R zr = fma_(alpha, zt, oma * zeq[i]);
        R zn = leq[i];                                   // projection onto [l, l]
        dyeq_n[i] = rho_eq * (zr - zn);
        zeq[i] = keep(active, zn, zeq[i]);
        yeq[i] = keep(active, yeq[i] + dyeq_n[i], yeq[i]);
      }
      if (want_delta) {
        MPMPC_UNROLL
        for (int j = 0; j < 5; ++j) put_delta(active, COLD_DYB + j, dyb_n[j]);
        MPMPC_UNROLL
        for (int i = 0; i < 3; ++i) put_delta(active, COLD_DYEQ + i, dyeq_n[i]);
      }
      iters = keepi(active, I(it), iters);
      // ---- termination
      if (can_check || can_adapt) { unpark_check_data(); info(nf); }
      if (can_check) {
        I stt = check(nf, st, false);
        Mk term = active & (stt != MPMPC_UNSOLVED);
        status = seli(term, stt, status);
        active = active & !term;
      }
      // ---- rho adaptation (OSQP compute_rho_estimate / adapt_rho)
      if (can_adapt) {
        R pr = nf.s_rp / (max_(nf.s_z, nf.s_Ax) + R(1e-10));
        R du = nf.s_rd / (max_(max_(nf.s_q, nf.s_Aty), nf.s_Px) + R(1e-10));
        R est = rho * sqrt_(pr / (du + R(1e-10)));
        est = min_(max_(est, R(RHO_MIN)), R(RHO_MAX));
        Mk upd = active & ((est > rho * R(st.adaptive_rho_tolerance)) | (est < rho / R(st.adaptive_rho_tolerance)));
        if (L::wany(upd)) {
          set_rho(sel(upd, est, rho));
          admm_factor(st.sigma);
        }
      }
    }
    unpark_check_data();
    // ---- ran out of iterations: OSQP's final exact, then approximate, check
    if (full && L::wany(active)) {
      info(nf);
      I s1 = check(nf, st, false);
      I s2 = check(nf, st, true);
      I fin = seli(s1 != MPMPC_UNSOLVED, s1, seli(s2 != MPMPC_UNSOLVED, s2, I(MPMPC_MAX_ITER_REACHED)));
      status = seli(active, fin, status);
    }
    if (full) {
      info(nf);
      pri_res = keep(which, nf.pri, pri_res);
      dua_res = keep(which, nf.dua, dua_res);
    } else {
      // stopped early for a polish attempt: that only asks for the primal residual (warm-start floor);
      // what it cannot certify runs the whole iteration again and gets its residuals there
      pri_res = keep(which, primal_residual(), pri_res);
      dua_res = keep(which, R(0.0), dua_res);
    }
  }

  // The start of the early polish attempt on the reduced problem: OSQP's FIRST iterate from its cold start - the
  // regularised least-squares point KKT^-1 (-q) relaxed by alpha, its projection and the dual step - computed for the
  // (e_y, e_psi, kappa) system with the 2 x 2 factorisation; the speed's own (decoupled) entry in closed form, nothing
  // for the time state.  Same point as admm(st, which, 1) up to the sigma-sized coupling through the time row, for the
  // 2 x 2 instead of the 3 x 3 factorisation.
  // The early attempt WITHOUT an OSQP iterate (mpmpc_settings::early_start = 0, the default): the interior point's centred
  // start from x = 0.  OSQP's first iterate as the start costs one factorisation and one KKT solve and buys nothing: config 3
  // 11.43 -> 11.08 interior-point iterations WITHOUT it (emulation, 256 instances), the reduced problem +0.25.
  // iters = 1 still marks "the early attempt alone".
  MPMPC_HD void zero_start(const SolverParams& st, const Mk& which) {
    status = keepi(which, I(MPMPC_UNSOLVED), status);
    iters = keepi(which, I(1), iters);
    ipm_iters = keepi(which, I(0), ipm_iters);
    polished = keepi(which, I(0), polished);
    set_rho(R(st.rho));
    MPMPC_UNROLL
    for (int j = 0; j < 5; ++j) { x[j] = keep(which, R(0.0), x[j]); zb[j] = keep(which, R(0.0), zb[j]); yb[j] = keep(which, R(0.0), yb[j]); }
    MPMPC_UNROLL
    for (int i = 0; i < 3; ++i) { zeq[i] = keep(which, leq[i], zeq[i]); yeq[i] = keep(which, R(0.0), yeq[i]); }
    pri_res = keep(which, R(1.0), pri_res);
    dua_res = keep(which, R(0.0), dua_res);
  }
  MPMPC_HD void reduced_start(const SolverParams& st, const Mk& which) {
    const R zero(0.0), alpha(st.alpha), sigma(st.sigma);
    status = keepi(which, I(MPMPC_UNSOLVED), status);
    iters = keepi(which, I(1), iters);
    ipm_iters = keepi(which, I(0), ipm_iters);
    polished = keepi(which, I(0), polished);
    set_rho(R(st.rho));
    R h5[5], h3[3], rx[3], req[2] = {zero, zero}, xt[3], nu[2];
    MPMPC_UNROLL
    for (int j = 0; j < 5; ++j) h5[j] = R(1.0) / (p[j] + sigma + (g[j] * g[j]) * rb[j]);
    h3[0] = h5[0]; h3[1] = h5[1]; h3[2] = h5[4];
    factor_t<LAY_RED>(h3, rinv_eq);
    rx[0] = -q[0]; rx[1] = -q[1]; rx[2] = -q[4];
    kkt_solve_t<LAY_RED>(rx, req, xt, nu);
    R xt5[5] = {xt[0], xt[1], zero, h5[3] * (-q[3]), xt[2]};
    MPMPC_UNROLL
    for (int j = 0; j < 5; ++j) {
      const R xn = alpha * xt5[j];
      const R zr = alpha * (g[j] * xt5[j]);
      const R zn = min_(max_(zr, lb[j]), ub[j]);
      x[j] = keep(which, xn, x[j]);
      zb[j] = keep(which, zn, zb[j]);
      yb[j] = keep(which, rb[j] * (zr - zn), yb[j]);
    }
    MPMPC_UNROLL
    for (int i = 0; i < 2; ++i) {
      const R zr = alpha * (nu[i] * rinv_eq);
      zeq[i] = keep(which, leq[i], zeq[i]);
      yeq[i] = keep(which, rho_eq * (zr - leq[i]), yeq[i]);
    }
    zeq[2] = keep(which, leq[2], zeq[2]);
    yeq[2] = keep(which, zero, yeq[2]);
    // unscaled primal residual of the rows the reduced problem has (the floor of the polish's warm start)
    R x3[3] = {x[0], x[1], x[4]}, Ax[2], pri(0.0);
    Aeq_mul_t<LAY_RED>(x3, Ax);
    MPMPC_UNROLL
    for (int i = 0; i < 2; ++i) pri = max_(pri, sel(vx, abs_((R(1.0) / Eeq[i]) * (Ax[i] - zeq[i])), zero));
    MPMPC_UNROLL
    for (int j = 0; j < 5; ++j) if (j != 2) pri = max_(pri, sel(valid[j], abs_((R(1.0) / Eb[j]) * (g[j] * x[j] - zb[j])), zero));
    pri_res = keep(which, L::gmax(pri), pri_res);
    dua_res = keep(which, zero, dua_res);
  }

  // ======================================================================== certified polish
  // Variable-space view of the box rows: g x in [lb, ub]  <=>  x in [lo, hi].
  template <int LAY>
  struct BoxT {
    R lo[EN<LAY>], hi[EN<LAY>];
    Mk Lm[EN<LAY>], Um[EN<LAY>], pin[EN<LAY>];
  };
  using Box = BoxT<LAY_FULL>;
  MPMPC_HD void make_box(Box& bx) const {
    MPMPC_UNROLL
    for (int j = 0; j < 5; ++j) {
      Mk fl = lb[j] > R(-INF_BOUND), fu = ub[j] < R(INF_BOUND);
      Mk pn = fl & fu & ((ub[j] - lb[j]) <= R(1e-12) * max_(R(1.0), abs_(lb[j])));
      bx.lo[j] = lb[j] / g[j];
      bx.hi[j] = ub[j] / g[j];
      bx.pin[j] = pn & valid[j];
      bx.Lm[j] = fl & !pn & valid[j];
      bx.Um[j] = fu & !pn & valid[j];
    }
  }

  // Regularised Mehrotra predictor-corrector, warm started at (xw, nuw, ybw).  Linear systems
  // go through the same block-tridiagonal Schur factorisation as the ADMM step.
  template <int LAY>
  struct IpmT {
    R x[EN<LAY>], nu[NR<LAY>], sl[EN<LAY>], su[EN<LAY>], zl[EN<LAY>], zu[EN<LAY>], pi[EN<LAY>];
    // Active-set indicators of the last step taken (not in phase 1): a bound counts as active when its slack shrinks
    // faster than its multiplier,  ds / s < dz / z  (Tapia's indicators: s+/s -> 0, z+/z -> 1 on an active bound, the
    // other way round on an inactive one).  Unlike "multiplier above slack" it identifies WEAKLY active bounds
    // (multipliers of 1e-7) at mu = 1e-9; with it the first active-set round is the last one for all but ~1 % of the
    // instances (mean 1.01 instead of 1.75 rounds).
    Mk tL[EN<LAY>], tU[EN<LAY>];
  };
  using Ipm = IpmT<LAY_FULL>;
  // pp, qq, vm: cost diagonal, cost vector and validity masks of the lane's entries in the layout S
  //
  // SOFT = true is PHASE 1 (see phase1()): every box entry j with a finite side reads  lo <= x_j + w_j <= hi  with
  // the cost 1/2 om_j w_j^2 and nothing else in the cost (qq is not read; pp carries the WEIGHTS om_j > 0 - phase1() passes
  // the squares of the box rows' scaled entries, which makes the cost OSQP's own metric of a violation, see there).
  // Stationarity in w gives om w = zl - zu, so w is never stored: it shifts the slack residuals, and eliminating its
  // Newton step  dw = ((cu - cl) - th dx) / (om + th),  th = zl / sl + zu / su,  leaves the hard problem's reduced system
  // with  k th = om (1 - k),  k = om / (om + th),  in place of th on the diagonal and  k (cu - cl)  in place of  cu - cl
  // on the right-hand side.  Pinned entries stay hard.
  // The loop also ends for an instance as soon as its multipliers pass the Farkas test in the scaled problem
  // (|A'y| <= eps |y|, support <= -eps |y|): what is asked of phase 1 is a ray, not a converged point.
  template <int LAY, bool SOFT = false>
  MPMPC_HD Mk ipm(const BoxT<LAY>& bx, IpmT<LAY>& s, const R* pp, const R* qq, const Mk* vm, const SolverParams& st,
                  double tol, const Mk& run) {
    constexpr int E = EN<LAY>, NQ = NR<LAY>;
    // In the reduced layouts entry 1 is e_psi (lower lanes) or nothing (upper lanes): never boxed - the reduced
    // polish is only taken when e_psi has no bound (reducible()) - so all of its slack arithmetic is left out at
    // compile time (the loops below are unrolled: boxed(j) is a constant in every copy).
    auto boxed = [](int j) constexpr {
      if (LAY >= LAY_RED) return j != 1;
      if (FREEX) return LAY == LAY_FULL ? (j != 1 && j != 2) : j != 2;      // split: entry 2 is t below, nothing above
      return true;
    };
    MPMPC_COUNT_CONTEXT(SPL<LAY> ? 1 : 0);
    const R reg(st.ipm_reg), ireg(st.inv_ipm_reg), one(1.0), zero(0.0);
    Mk active = run, conv = L::mfalse();
    R cnt(0.0);
    MPMPC_UNROLL
    for (int j = 0; j < E; ++j) if (boxed(j)) cnt = cnt + sel(bx.Lm[j], one, zero) + sel(bx.Um[j], one, zero);
    const R inb = rcp_(max_(L::gsum(cnt), one));      // (its reciprocal once: the complementarity measures below are products)
    I stall(0);
    R mu_min(1e300);
    [[maybe_unused]] R iom[E];        // phase 1: reciprocals of the weights
    if constexpr (SOFT) {
      MPMPC_UNROLL
      for (int j = 0; j < E; ++j) iom[j] = boxed(j) ? rcp_(pp[j]) : one;
    }
    for (int it = 0; it <= st.ipm_max_iter; ++it) {
      // ---- residuals (the slack residuals rl, ru, rpin are cheap functions of the iterate: they are
      //      re-evaluated where needed instead of being carried across the sweeps)
      auto w_of = [&](int j) { return SOFT ? (s.zl[j] - s.zu[j]) * iom[j] : zero; };
      auto rl_of = [&](int j) { return sel(bx.Lm[j], s.x[j] + w_of(j) - bx.lo[j] - s.sl[j], zero); };
      auto ru_of = [&](int j) { return sel(bx.Um[j], bx.hi[j] - s.x[j] - w_of(j) - s.su[j], zero); };
      auto rpin_of = [&](int j) { return sel(bx.pin[j], s.x[j] - bx.lo[j], zero); };
      MPMPC_TICK_BEGIN(10);
      R At[E], rp[NQ], rd[E];
      AeqT_mul_t<LAY>(s.nu, At);
      Aeq_mul_t<LAY>(s.x, rp);
      R res(0.0), msum(0.0);
      [[maybe_unused]] R Pod[E];                               // off-diagonal part of P x (FQ, not in phase 1)
      if constexpr (FQ && !SOFT) {
        MPMPC_UNROLL
        for (int j = 0; j < E; ++j) Pod[j] = zero;
        Poff_add<LAY>(s.x, Pod);
      }
      [[maybe_unused]] R rk_dot(0.0);                          // rank-one part of P x: rk_c (rk_c' x)   (LAY_RED4)
      if constexpr (LAY == LAY_RED4 && !SOFT) rk_dot = rank_one_dot(s.x);
      MPMPC_UNROLL
      for (int i = 0; i < NQ; ++i) {
        rp[i] = rp[i] - leq[i];
        if constexpr (SOFT) rp[i] = fma_(R(-P1_EQ_SOFT), s.nu[i], rp[i]);          // (soft dynamics rows: see P1_EQ_SOFT)
        res = max_(res, sel(vx, abs_(rp[i]), zero));
      }
      MPMPC_UNROLL
      for (int j = 0; j < E; ++j) {
        if constexpr (SOFT) rd[j] = At[j] - s.zl[j] + s.zu[j] + s.pi[j];
        else {
          rd[j] = fma_(pp[j], s.x[j], qq[j]) + At[j] - s.zl[j] + s.zu[j] + s.pi[j];
          if constexpr (FQ) rd[j] = rd[j] + Pod[j];
          if constexpr (LAY == LAY_RED4) { if (j == 0 || j == 3) rd[j] = fma_(rk_c[j == 0 ? 0 : 1], rk_dot, rd[j]); }
        }
        if (!boxed(j)) { res = max_(res, sel(vm[j], abs_(rd[j]), zero)); continue; }
        res = max_(res, sel(vm[j], max_(max_(abs_(rd[j]), abs_(rpin_of(j))), max_(abs_(rl_of(j)), abs_(ru_of(j)))), zero));
        msum = msum + sel(bx.Lm[j], s.sl[j] * s.zl[j], zero) + sel(bx.Um[j], s.su[j] * s.zu[j], zero);
      }
      res = L::gmax(res);
      R mu = L::gsum(msum) * inb;
      // (the residual of a converged iterate sits at ~1e-13 in double precision: the retry at ipm_tol x 1e-4 asks the
      //  complementarity for its tolerance - that is what identifies a weakly active bound - and the residual for 1e-11)
      Mk ok = (res < R(tol > 1e-11 ? tol : 1e-11)) & (mu < R(tol));
      if constexpr (SOFT) p1_converged = selb(active, ok, p1_converged);
      if constexpr (SOFT) {
        // Farkas test on the multipliers y = (nu, zu - zl + pi) in the scaled problem: A'y is the dual residual rd itself
        R ny(0.0), na(0.0), sup(0.0);
        MPMPC_UNROLL
        for (int i = 0; i < NQ; ++i) { ny = max_(ny, sel(vx, abs_(s.nu[i]), zero)); sup = sup + sel(vx, leq[i] * s.nu[i], zero); }
        MPMPC_UNROLL
        for (int j = 0; j < E; ++j) {
          na = max_(na, sel(vm[j], abs_(rd[j]), zero));
          if (!boxed(j)) continue;
          R lam = s.zu[j] - s.zl[j] + s.pi[j];
          ny = max_(ny, sel(vm[j], abs_(lam), zero));
          // hi max(lam, 0) + lo min(lam, 0); an infinite side carries no multiplier (zl / zu are zero there)
          sup = sup + sel(vm[j] & (lam > zero) & (bx.Um[j] | bx.pin[j]), sel(bx.pin[j], bx.lo[j], bx.hi[j]) * lam, zero) +
                sel(vm[j] & (lam < zero) & (bx.Lm[j] | bx.pin[j]), bx.lo[j] * lam, zero);
        }
        ny = L::gmax(ny); na = L::gmax(na); sup = L::gsum(sup);
        const R thr = R(st.phase1_eps) * ny;
        Mk ray = (ny > R(st.phase1_eps)) & (na < thr) & (sup < -thr);
        if (st.phase1_accept) {
          // A ray settles "infeasible" - but whether the instance is MARGINALLY so is decided by the violation of the
          // converged least-violation point, which can be less than half of an early iterate's: the loop leaves at a ray only
          // while the iterate's violation (|w| in unscaled units; qq carries D / om in phase 1) is beyond the band in which
          // that question is open
          R wv(0.0);
          MPMPC_UNROLL
          for (int j = 0; j < E; ++j)
            if (boxed(j)) wv = max_(wv, sel(vm[j], abs_(s.zl[j] - s.zu[j]) * qq[j], zero));
          ray = ray & (L::gmax(wv) > p1_band);
        }
        ok = ok | ray;
      }
      conv = conv | (active & ok);
      active = active & !ok;
      MPMPC_TICK_END(10);
      if (it == st.ipm_max_iter || !L::wany(active)) break;
#ifdef MPMPC_EMU_DEBUG
      std::fprintf(stderr, "  ipm%s it %2d res %.3e mu %.3e mu_min %.3e active %d\n", SOFT ? "(p1)" : "", it, res.v[16], mu.v[16], mu_min.v[16], (int)active.v[16]);
#endif
      if constexpr (!SOFT) {
        // the complementarity measure of a feasible problem falls (nearly) monotonically; on an infeasible one the
        // multipliers blow up within a few iterations (mu jumps by 4-5 orders of magnitude): give up at once, phase 1
        // is what can decide such an instance
        // ... and so is a complementarity measure that has collapsed far below the tolerance while the residual
        // has not moved: the iterate sits on the boundary of an empty set
        active = active & !(mu > R(st.ipm_diverged) * mu_min) & !((mu < R(tol * 1e-3)) & (res > R(1e-5)));
        mu_min = min_(mu_min, mu);
        if (!L::wany(active)) break;
      }
      MPMPC_TICK_COUNT(16);
      MPMPC_TICK_BEGIN(11);
      ipm_iters = seli(active, ipm_iters + I(1), ipm_iters);
      // ---- factor.  Every division by a slack below is a product with its reciprocal, taken once.
      R isl[E], isu[E], h[E];
      MPMPC_UNROLL
      for (int j = 0; j < E; ++j) if (boxed(j)) { isl[j] = rcp_(s.sl[j]); isu[j] = rcp_(s.su[j]); }
      [[maybe_unused]] R kap[E];        // phase 1: k = 1 / (1 + th) of the soft entries, th = zl / sl + zu / su
      auto H_of = [&](int j) {
        if (!boxed(j)) return SOFT ? reg : pp[j] + reg;
        if constexpr (SOFT) {
          kap[j] = rcp_(fma_(sel(bx.Lm[j], s.zl[j] * isl[j], zero) + sel(bx.Um[j], s.zu[j] * isu[j], zero), iom[j], one));
          return fma_(-pp[j], kap[j], reg + pp[j]) + sel(bx.pin[j], ireg, zero);          // k th = om (1 - k)
        } else {
          return pp[j] + reg + sel(bx.Lm[j], s.zl[j] * isl[j], zero) + sel(bx.Um[j], s.zu[j] * isu[j], zero) +
                 sel(bx.pin[j], ireg, zero);
        }
      };
      [[maybe_unused]] R Hd[E];
      MPMPC_UNROLL
      for (int j = 0; j < E; ++j) { const R Hj = H_of(j); if constexpr (FQ) Hd[j] = Hj; h[j] = rcp_(Hj); }
      dense_blocks<LAY, !SOFT>(Hd, h);
      factor_t<LAY>(h, SOFT ? reg + R(P1_EQ_SOFT) : reg);
      MPMPC_TICK_END(11);
      // ---- predictor and corrector share the factorisation.  (No iterative refinement of the directions: over
      //      thousands of instances of every configuration it changed neither an iteration count nor a status -
      //      the answer is made by the active-set solve that follows, which does refine.)
      R dx[E], dnu[NQ], dsl[E], dsu[E], dzl[E], dzu[E], dpi[E];
      R rcl[E], rcu[E];
      MPMPC_UNROLL
      for (int j = 0; j < E; ++j) if (boxed(j)) { rcl[j] = s.sl[j] * s.zl[j]; rcu[j] = s.su[j] * s.zu[j]; }
      R alpha_aff(1.0);
      for (int pass = 0; pass < 2; ++pass) {
        R rhs[E], nreq[NQ];
        [[maybe_unused]] R cul[E];          // phase 1: cu - cl of the entry
        MPMPC_UNROLL
        for (int j = 0; j < E; ++j) {
          if (!boxed(j)) { rhs[j] = -rd[j]; continue; }
          if constexpr (SOFT) {
            cul[j] = sel(bx.Um[j], fma_(s.zu[j], ru_of(j), rcu[j]) * isu[j], zero) -
                     sel(bx.Lm[j], fma_(s.zl[j], rl_of(j), rcl[j]) * isl[j], zero);
            rhs[j] = fma_(kap[j], cul[j], -rd[j]) - sel(bx.pin[j], rpin_of(j) * ireg, zero);
          } else {
            rhs[j] = -rd[j] - sel(bx.Lm[j], fma_(s.zl[j], rl_of(j), rcl[j]) * isl[j], zero) +
                     sel(bx.Um[j], fma_(s.zu[j], ru_of(j), rcu[j]) * isu[j], zero) - sel(bx.pin[j], rpin_of(j) * ireg, zero);
          }
        }
        MPMPC_UNROLL
        for (int i = 0; i < NQ; ++i) nreq[i] = -rp[i];
        MPMPC_TICK_BEGIN(12);
        kkt_solve_t<LAY>(rhs, nreq, dx, dnu);
        if (SOFT && pass == 1) {
          // (the predictor only supplies the centring parameter and the second-order term: not refined)
          // One refinement step against the UN-regularised Newton matrix (phase 1 only: it is rare, and what it is
          // asked for is a clean ray - |A'y| / |y| drops from ~1e-6 to ~1e-9, far below the margin phase1_eps asks
          // of the support; the optimum of the hard problem is made by the refining active-set solve instead).
          R Ad[NQ], Atd[E], r1[E], r2[NQ], ddx[E], ddn[NQ];
          Aeq_mul_t<LAY>(dx, Ad);
          AeqT_mul_t<LAY>(dnu, Atd);
          MPMPC_UNROLL
          for (int j = 0; j < E; ++j) r1[j] = rhs[j] - fma_(rcp_(h[j]) - reg, dx[j], Atd[j]);
          MPMPC_UNROLL
          for (int i = 0; i < NQ; ++i) r2[i] = fma_(R(P1_EQ_SOFT), dnu[i], nreq[i] - Ad[i]);
          kkt_solve_t<LAY>(r1, r2, ddx, ddn);
          MPMPC_UNROLL
          for (int j = 0; j < E; ++j) dx[j] = dx[j] + sel(vm[j], ddx[j], zero);
          MPMPC_UNROLL
          for (int i = 0; i < NQ; ++i) dnu[i] = dnu[i] + sel(vx, ddn[i], zero);
        }
        MPMPC_TICK_END(12);
        // largest step that keeps slacks and multipliers positive: 1 / max(-ds/s, -dz/z)
        R blk(0.0);
        MPMPC_UNROLL
        for (int j = 0; j < E; ++j) {
          if (!boxed(j)) continue;
          R ex = dx[j];                     // step of x + w:  dx + ((cu - cl) - th dx) / (om + th) = k (dx + (cu - cl) / om)
          if constexpr (SOFT) ex = kap[j] * fma_(cul[j], iom[j], dx[j]);
          dsl[j] = sel(bx.Lm[j], ex + rl_of(j), zero);
          dsu[j] = sel(bx.Um[j], -ex + ru_of(j), zero);
          dzl[j] = sel(bx.Lm[j], -fma_(s.zl[j], dsl[j], rcl[j]) * isl[j], zero);
          dzu[j] = sel(bx.Um[j], -fma_(s.zu[j], dsu[j], rcu[j]) * isu[j], zero);
          dpi[j] = sel(bx.pin[j], (rpin_of(j) + dx[j]) * ireg, zero);
          blk = max_(blk, max_(sel(bx.Lm[j], -dsl[j] * isl[j], zero), sel(bx.Um[j], -dsu[j] * isu[j], zero)));
          // (the ratios -dz / z only size the step, which keeps 0.5 % from the boundary anyway: the reciprocal's seed will do)
          blk = max_(blk, max_(sel(bx.Lm[j], -dzl[j] * rcp_fast_(s.zl[j]), zero), sel(bx.Um[j], -dzu[j] * rcp_fast_(s.zu[j]), zero)));
        }
        blk = L::gmax(blk);
        R ratio = sel(blk > zero, rcp_(blk), R(1e300));
        if (pass == 0) {
          alpha_aff = min_(one, ratio);
          R ms(0.0);
          MPMPC_UNROLL
          for (int j = 0; j < E; ++j)
            if (boxed(j))
              ms = ms + sel(bx.Lm[j], fma_(alpha_aff, dsl[j], s.sl[j]) * fma_(alpha_aff, dzl[j], s.zl[j]), zero) +
                   sel(bx.Um[j], fma_(alpha_aff, dsu[j], s.su[j]) * fma_(alpha_aff, dzu[j], s.zu[j]), zero);
          R mu_aff = L::gsum(ms) * inb;
          R sg = mu_aff * rcp_(max_(mu, R(1e-300)));
          sg = sg * sg * sg;
          const R sgmu = sg * mu;
          MPMPC_UNROLL
          for (int j = 0; j < E; ++j) {
            if (!boxed(j)) continue;
            rcl[j] = fma_(dsl[j], dzl[j], fma_(s.sl[j], s.zl[j], -sgmu));
            rcu[j] = fma_(dsu[j], dzu[j], fma_(s.su[j], s.zu[j], -sgmu));
          }
        } else {
          R al = min_(one, R(0.995) * ratio);
          stall = seli(active & (al < R(1e-6)), stall + I(1), I(0));
          MPMPC_UNROLL
          for (int j = 0; j < E; ++j) {
            s.x[j] = sel(active, fma_(al, dx[j], s.x[j]), s.x[j]);
            if (!boxed(j)) continue;
            if constexpr (!SOFT) {
              s.tL[j] = selb(active, bx.Lm[j] & (dsl[j] * s.zl[j] < dzl[j] * s.sl[j]), s.tL[j]);
              s.tU[j] = selb(active, bx.Um[j] & (dsu[j] * s.zu[j] < dzu[j] * s.su[j]), s.tU[j]);
            }
            s.sl[j] = sel(active, fma_(al, dsl[j], s.sl[j]), s.sl[j]);
            s.su[j] = sel(active, fma_(al, dsu[j], s.su[j]), s.su[j]);
            s.zl[j] = sel(active, fma_(al, dzl[j], s.zl[j]), s.zl[j]);
            s.zu[j] = sel(active, fma_(al, dzu[j], s.zu[j]), s.zu[j]);
            s.pi[j] = sel(active, fma_(al, dpi[j], s.pi[j]), s.pi[j]);
          }
          MPMPC_UNROLL
          for (int i = 0; i < NQ; ++i) s.nu[i] = sel(active, fma_(al, dnu[i], s.nu[i]), s.nu[i]);
          active = active & (stall < 3);      // steps collapsed: infeasible or hopelessly degenerate
        }
      }
    }
    return conv;
  }

  // OSQP's polish solve on a given active set, iterated with primal-dual active-set updates.
  // On success (xs, nus, lam) is a KKT point of the scaled problem.  LAY: LAY_FULL, or LAY_RED for the reduced
  // problem (entries e_y, e_psi, kappa; pp, qq: cost diagonal and vector in that layout).
  template <int LAY>
  MPMPC_HD Mk active_set(const BoxT<LAY>& bx, const R* pp, const R* qq, const Mk* vm, Mk* aL, Mk* aU, R* xs, R* nus, R* lam,
                         const SolverParams& st, const Mk& run, double add_fraction) {
    constexpr int E = EN<LAY>, NQ = NR<LAY>;
    const R delta(st.as_delta), idelta(st.inv_as_delta), zero(0.0), one(1.0), tol(1e-9);
    Mk todo = run, okm = L::mfalse();
    for (int rnd = 0; rnd < st.as_rounds; ++rnd) {
      if (!L::wany(todo)) break;
      R bound[E];
      Mk act[E];
      MPMPC_UNROLL
      for (int j = 0; j < E; ++j) act[j] = aL[j] | aU[j] | bx.pin[j];
      {
        R h[E], Hd[E];
        MPMPC_UNROLL
        for (int j = 0; j < E; ++j) {
          Hd[j] = pp[j] + delta + sel(act[j], idelta, zero);
          h[j] = one / Hd[j];
        }
        dense_blocks<LAY>(Hd, h);
        MPMPC_TICK_COUNT(17);
        MPMPC_TICK_BEGIN(13);
        factor_t<LAY>(h, delta);
        MPMPC_TICK_END(13);
      }
      L::fence();          // (a scheduling fence: what follows is formed after the factorisation, not carried through it)
      MPMPC_UNROLL
      for (int j = 0; j < E; ++j) bound[j] = sel(aU[j], bx.hi[j], bx.lo[j]);
      R xn[E], nn[NQ], ln[E];
      MPMPC_UNROLL
      for (int j = 0; j < E; ++j) xn[j] = ln[j] = zero;
      MPMPC_UNROLL
      for (int i = 0; i < NQ; ++i) nn[i] = zero;
      // Packed waves: every instance takes exactly the refinement steps it would take alone - `refine` = the instances whose
      // refinement still runs; a step is committed where it was needed - so that an answer does not depend on which
      // instance shares the wave (the tail kernel's partners come from a list whose order differs from run to run).
      [[maybe_unused]] Mk refine = todo;
      for (int rf = 0; rf <= st.as_refine; ++rf) {
        R At[E], Ax[NQ], rhs[E], r2[NQ], r3[E], dx[E], dnu[NQ];
        AeqT_mul_t<LAY>(nn, At);
        Aeq_mul_t<LAY>(xn, Ax);
        R rs(0.0);                       // KKT residual of the unregularised system at (xn, nn, ln)
        [[maybe_unused]] R Pod[E];
        if constexpr (FQ) {
          MPMPC_UNROLL
          for (int j = 0; j < E; ++j) Pod[j] = zero;
          Poff_add<LAY>(xn, Pod);
        }
        [[maybe_unused]] R rk_dot(0.0);
        if constexpr (LAY == LAY_RED4) rk_dot = rank_one_dot(xn);
        MPMPC_UNROLL
        for (int j = 0; j < E; ++j) {
          R r1 = -qq[j] - pp[j] * xn[j] - At[j] - ln[j];
          if constexpr (FQ) r1 = r1 - Pod[j];
          if constexpr (LAY == LAY_RED4) { if (j == 0 || j == 3) r1 = fma_(-rk_c[j == 0 ? 0 : 1], rk_dot, r1); }
          r3[j] = sel(act[j], bound[j] - xn[j], zero);
          rhs[j] = fma_(r3[j], idelta, r1);
          rs = max_(rs, sel(vm[j], max_(abs_(r1), abs_(r3[j])), zero));
        }
        MPMPC_UNROLL
        for (int i = 0; i < NQ; ++i) { r2[i] = leq[i] - Ax[i]; rs = max_(rs, sel(vx, abs_(r2[i]), zero)); }
        // the point already satisfies the system to rounding level (1e-15) for every instance in the wave: no further solve
        [[maybe_unused]] Mk need = todo;
        if constexpr (L::per_wave == 1) {
          if (rf >= 1 && !L::wany(todo & (L::gmax(rs) > R(1e-15)))) break;
        } else {
          if (rf >= 1) need = refine & (L::gmax(rs) > R(1e-15));
          if (!L::wany(need)) break;
        }
        MPMPC_TICK_COUNT(18);
        MPMPC_TICK_BEGIN(14);
        kkt_solve_t<LAY>(rhs, r2, dx, dnu);
        MPMPC_TICK_END(14);
        R big(0.0);
        MPMPC_UNROLL
        for (int j = 0; j < E; ++j) {
          ln[j] = updw(need, ln[j] + sel(act[j], (dx[j] - r3[j]) * idelta, zero), ln[j]);
          xn[j] = updw(need, xn[j] + dx[j], xn[j]);
          big = max_(big, sel(vm[j], abs_(dx[j]) - R(1e-14) * abs_(xn[j]), zero));
        }
        MPMPC_UNROLL
        for (int i = 0; i < NQ; ++i) nn[i] = updw(need, nn[i] + dnu[i], nn[i]);
        // refinement has converged for every instance in the wave: stop early
        if constexpr (L::per_wave == 1) {
          if (rf >= 1 && !L::wany(todo & (L::gmax(big) > R(1e-15)))) break;
        } else {
          refine = rf < 1 ? need : need & (L::gmax(big) > R(1e-15));
          if (!L::wany(refine)) break;
        }
        // ... or every instance still in the wave has a violation far beyond what refinement can still move
        // (1e-6): this active set is wrong, the next round does not need its exact solution
        // (one instance per wave only: a packed wave would need all its instances to agree, and rarely does)
        if (L::per_wave == 1 && rf >= 1 && rf < st.as_refine) {
          const R far(1e-6);
          Mk clear = L::mfalse();
          MPMPC_UNROLL
          for (int j = 0; j < E; ++j)
            clear = clear | (bx.Lm[j] & !aL[j] & (xn[j] < bx.lo[j] - far)) | (bx.Um[j] & !aU[j] & (xn[j] > bx.hi[j] + far)) |
                    (aL[j] & (ln[j] > far)) | (aU[j] & (ln[j] < -far));
          if (!L::wany(todo & !L::gany(clear))) break;
        }
      }
      Mk anybad = L::mfalse();
      Mk vL[E], vU[E], bL[E], bU_[E];
      R worst(0.0);
      MPMPC_UNROLL
      for (int j = 0; j < E; ++j) {
        vL[j] = bx.Lm[j] & !aL[j] & (xn[j] < bx.lo[j] - tol);
        vU[j] = bx.Um[j] & !aU[j] & (xn[j] > bx.hi[j] + tol);
        bL[j] = aL[j] & (ln[j] > tol);
        bU_[j] = aU[j] & (ln[j] < -tol);
        anybad = anybad | vL[j] | vU[j] | bL[j] | bU_[j];
        worst = max_(worst, max_(sel(vL[j], bx.lo[j] - xn[j], zero), sel(vU[j], xn[j] - bx.hi[j], zero)));
      }
      anybad = L::gany(anybad);
      // Only the violations within as_add_fraction of the worst one enter the active set: the small ones are mostly
      // consequences of the large ones (a missed weakly active bound pushes its neighbours out by a fraction of its own
      // violation), and adding them all at once makes the primal-dual iteration cycle on long horizons.
      {
        const R thr = R(add_fraction) * L::gmax(worst);
        MPMPC_UNROLL
        for (int j = 0; j < E; ++j) {
          vL[j] = vL[j] & !(bx.lo[j] - xn[j] < thr);
          vU[j] = vU[j] & !(xn[j] - bx.hi[j] < thr);
        }
      }
      MPMPC_UNROLL
      for (int j = 0; j < E; ++j) {
        xs[j] = updw(todo, xn[j], xs[j]);
        lam[j] = updw(todo, ln[j], lam[j]);
        Mk nL = (aL[j] & !bL[j]) | vL[j];
        Mk nU = ((aU[j] & !bU_[j]) | vU[j]) & !nL;
        aL[j] = selb(todo & anybad, nL, aL[j]);
        aU[j] = selb(todo & anybad, nU, aU[j]);
      }
      MPMPC_UNROLL
      for (int i = 0; i < NQ; ++i) nus[i] = updw(todo, nn[i], nus[i]);
      okm = okm | (todo & !anybad);
      todo = todo & anybad;
    }
    return okm;
  }

  // ---- reduced problem: the closed-form part.  Given the solution of the (e_y, e_psi, kappa) problem in xs[0], xs[1],
  // xs[4] (scaled), fill in the speed v_k = argmin over its box of its own separable cost (xs[3], with its multiplier
  // lam[3]) and roll the time state forward through its equality rows (xs[2]); the multipliers of the time rows and
  // of the time boxes are zero.  All in the scaled problem, so that certificate() checks the FULL KKT system.
  MPMPC_HD void reduced_complete(const Box& bx, R xs[5], R nus[3], R lam[5]) const {
    const R zero(0.0);
    // v: minimise 1/2 p3 x^2 + q3 x on [lo, hi]  (p3 > 0: the launcher takes the reduced path only then)
    R xv = -q[3] / p[3];
    xv = sel(bx.Um[3] & (xv > bx.hi[3]), bx.hi[3], xv);
    xv = sel((bx.Lm[3] | bx.pin[3]) & (xv < bx.lo[3]), bx.lo[3], xv);
    xv = sel(bx.pin[3], bx.lo[3], xv);
    xs[3] = sel(valid[3], xv, zero);
    lam[3] = sel(valid[3], -fma_(p[3], xs[3], q[3]), zero);
    nus[2] = zero;
    lam[2] = zero;
    // t: row 2 of equality block k:  mI2 t_k + (a4 e_y + a5 t + b1 v)_{k-1} = leq2_k, a forward recurrence along the
    // stages (once per solve: N steps of one fused multiply-add and one lane shift each)
    const R imI = R(1.0) / mI[2], t0 = leq[2] * imI, drive = fma_(b[1], xs[3], a[4] * xs[0]);
    R t = t0;                                     // stage 0; later stages are overwritten step by step
    for (int it = 0; it < N; ++it) {
      const R inflow = L::up(fma_(a[5], t, drive));
      t = sel(first, t0, (leq[2] - inflow) * imI);
    }
    xs[2] = sel(vx, t, zero);
  }

  // KKT certificate in the UNSCALED problem: primal violation, stationarity, complementarity
  MPMPC_HD Mk certificate(const R xs[5], const R nus[3], const R lam[5], double tol, R& prim, R& stat) const {
    R Ax[3], At[5];
    Aeq_mul(xs, Ax);
    AeqT_mul(nus, At);
    R pv(0.0), sv(0.0), cv(0.0);
    R cinv = R(1.0) / c;
    R Pod[5] = {R(0.0), R(0.0), R(0.0), R(0.0), R(0.0)};
    if constexpr (FQ) Poff_add<0>(xs, Pod);
    MPMPC_UNROLL
    for (int i = 0; i < 3; ++i) pv = max_(pv, sel(vx, abs_((Ax[i] - leq[i]) / Eeq[i]), R(0.0)));
    MPMPC_UNROLL
    for (int j = 0; j < 5; ++j) {
      R xu = D[j] * xs[j];
      const R lo0 = lo_raw(j), hi0 = hi_raw(j);
      R viol = max_(max_(lo0 - xu, xu - hi0), R(0.0));
      pv = max_(pv, sel(valid[j], viol, R(0.0)));
      R rd = fma_(p[j], xs[j], q[j]) + At[j] + lam[j];
      if constexpr (FQ) rd = rd + Pod[j];
      sv = max_(sv, sel(valid[j], abs_(rd / D[j]) * cinv, R(0.0)));
      R yu = (lam[j] / D[j]) * cinv;                      // multiplier of the unscaled box row
      Mk fu = hi0 < R(INF_BOUND), fl = lo0 > R(-INF_BOUND);
      R cu = sel(fu, max_(yu, R(0.0)) * abs_(hi0 - xu), sel(yu > R(0.0), R(1e300), R(0.0)));
      R cl = sel(fl, max_(-yu, R(0.0)) * abs_(xu - lo0), sel(yu < R(0.0), R(1e300), R(0.0)));
      cv = max_(cv, sel(valid[j], max_(cu, cl), R(0.0)));
    }
    // a NaN iterate must never pass: compare every entry against a finite bound explicitly
    Mk bad = L::mfalse();
    MPMPC_UNROLL
    for (int j = 0; j < 5; ++j) bad = bad | (valid[j] & !((abs_(xs[j]) < R(1e300)) & (abs_(lam[j]) < R(1e300))));
    MPMPC_UNROLL
    for (int i = 0; i < 3; ++i) bad = bad | (vx & !(abs_(nus[i]) < R(1e300)));
    bad = L::gany(bad);
    prim = L::gmax(pv);
    stat = L::gmax(sv);
    cv = L::gmax(cv);
    return (prim <= R(tol)) & (stat <= R(tol)) & (cv <= R(tol)) & !bad;
  }

  // During the interior-point and active-set loops the scalings, the scaled row bounds and the ADMM
  // point (kept as the fallback answer) are parked in cold storage: slots 0..41.
  MPMPC_HD void stash() const {
    MPMPC_UNROLL
    for (int j = 0; j < 5; ++j) {
      L::cold_put(j, D[j]); L::cold_put(5 + j, Eb[j]); L::cold_put(10 + j, lb[j]); L::cold_put(15 + j, ub[j]);
      L::cold_put(20 + j, x[j]); L::cold_put(25 + j, yb[j]); L::cold_put(37 + j, g[j]);
    }
    MPMPC_UNROLL
    for (int i = 0; i < 3; ++i) { L::cold_put(30 + i, Eeq[i]); L::cold_put(33 + i, yeq[i]); }
    L::cold_put(36, c);
    L::fence();
  }
  MPMPC_HD void unstash() {
    L::fence();
    MPMPC_UNROLL
    for (int j = 0; j < 5; ++j) {
      D[j] = L::cold_get(j); Eb[j] = L::cold_get(5 + j); lb[j] = L::cold_get(10 + j); ub[j] = L::cold_get(15 + j);
      x[j] = L::cold_get(20 + j); yb[j] = L::cold_get(25 + j); g[j] = L::cold_get(37 + j);
    }
    MPMPC_UNROLL
    for (int i = 0; i < 3; ++i) { Eeq[i] = L::cold_get(30 + i); yeq[i] = L::cold_get(33 + i); }
    c = L::cold_get(36);
  }

  MPMPC_HD static I pack_active(const Mk aL[5], const Mk aU[5]) {
    I v(1 << 30);
    MPMPC_UNROLL
    for (int j = 0; j < 5; ++j) v = v + seli(aL[j], I(1 << j), I(0)) + seli(aU[j], I(32 << j), I(0));
    return v;
  }
  // ---- the layouts the polish runs in (RED: template flag of the Solver, see the layout table above)
  static constexpr int LAY_AS = RED ? LAY_RED : LAY_FULL;                                           // active-set rounds
  static constexpr int LAY_IP = RED ? (kSplit ? LAY_REDSPLIT : LAY_RED) : (kSplit ? LAY_SPLIT : LAY_FULL);   // interior point

  // the box, the cost and the validity masks of the lane's entries in layout LAY (from the 5-entry box bx)
  template <int LAY>
  MPMPC_HD void problem_in_layout(const Box& bx, BoxT<LAY>& bi, R* pp, R* qq, Mk* vm) {
    constexpr int E = EN<LAY>;
    const R one(1.0), zero(0.0);
    if constexpr (SPL<LAY>) {
      MPMPC_UNROLL
      for (int i = 0; i < 2; ++i) bU[i] = sel(sU, L::from_lower(b[i]), zero);
      if constexpr (FQ) {       // the cost's off-diagonals in the split layout: the input lanes' entries 0, 1 are (v, kappa)
        podS[0] = sel(sU, L::from_lower(rod), pod[0]); podS[1] = sel(sU, zero, pod[1]); podS[2] = sel(sU, zero, pod[2]);
      }
    }
    to_lay<LAY>(bx.lo, bi.lo); to_lay<LAY>(bx.hi, bi.hi); to_lay<LAY>(p, pp); to_lay<LAY>(q, qq);
    mask_to_lay<LAY>(bx.Lm, bi.Lm); mask_to_lay<LAY>(bx.Um, bi.Um); mask_to_lay<LAY>(bx.pin, bi.pin);
    valid_lay<LAY>(vm);
    // an entry the upper lanes do not have: unit cost keeps its arithmetic finite
    if constexpr (LAY == LAY_SPLIT) pp[2] = sel(sU, one, pp[2]);
    if constexpr (LAY == LAY_REDSPLIT) pp[1] = sel(sU, one, pp[1]);
    (void)E;
  }
  template <int LAY>
  MPMPC_HD void iterate_to_layout(const Ipm& s, IpmT<LAY>& si) const {
    const R one(1.0);
    to_lay<LAY>(s.x, si.x); to_lay<LAY>(s.sl, si.sl); to_lay<LAY>(s.su, si.su);
    to_lay<LAY>(s.zl, si.zl); to_lay<LAY>(s.zu, si.zu); to_lay<LAY>(s.pi, si.pi);
    if constexpr (LAY == LAY_SPLIT) { si.sl[2] = sel(sU, one, si.sl[2]); si.su[2] = sel(sU, one, si.su[2]); }
    if constexpr (LAY == LAY_REDSPLIT) { si.sl[1] = sel(sU, one, si.sl[1]); si.su[1] = sel(sU, one, si.su[1]); }
    MPMPC_UNROLL
    for (int i = 0; i < NR<LAY>; ++i) si.nu[i] = s.nu[i];
  }

  // One active-set attempt from the guess (aL5, aU5) in the 5-entry view, in the layout LAY_AS.  On return (xs, nus,
  // lam) hold the full point (reduced problem: completed by reduced_complete), aL5 / aU5 the final active set.
  MPMPC_HD Mk active_set_full(const Box& bx, Mk aL5[5], Mk aU5[5], R xs[5], R nus[3], R lam[5], const SolverParams& st, double add_fraction,
                              const Mk& run) {
    constexpr int LAY = LAY_AS;
    constexpr int E = EN<LAY>, NQ = NR<LAY>;
    BoxT<LAY> ba;
    R pp[E], qq[E], xa[E], la[E], na[NQ];
    Mk vm[E], aL[E], aU[E];
    problem_in_layout<LAY>(bx, ba, pp, qq, vm);
    mask_to_lay<LAY>(aL5, aL); mask_to_lay<LAY>(aU5, aU);
    to_lay<LAY>(xs, xa); to_lay<LAY>(lam, la);
    MPMPC_UNROLL
    for (int i = 0; i < NQ; ++i) na[i] = nus[i];
    Mk okm = active_set<LAY>(ba, pp, qq, vm, aL, aU, xa, na, la, st, run, add_fraction);
    from_lay<LAY>(xa, xs); from_lay<LAY>(la, lam);
    MPMPC_UNROLL
    for (int i = 0; i < NQ; ++i) nus[i] = na[i];
    mask_from_lay<LAY>(aL, aL5); mask_from_lay<LAY>(aU, aU5);
    if constexpr (RED) {
      reduced_complete(bx, xs, nus, lam);
      // the speed's own activity, for the warm start of the next closed-loop step
      aL5[2] = aU5[2] = L::mfalse();
      aL5[3] = bx.Lm[3] & (lam[3] < R(0.0)) & valid[3];
      aU5[3] = bx.Um[3] & (lam[3] > R(0.0)) & valid[3] & !aL5[3];
    }
    return okm;
  }

  // Warm start (closed loop): `guess` is the active set of the previous step's certified plan, already shifted to
  // this step's stages.  One or two active-set rounds from it usually reproduce the optimum; whatever they
  // cannot certify goes through the normal path.  Runs on the scaled problem, before any ADMM.
  MPMPC_HD void warm_polish(const SolverParams& st, const I& guess, const Mk& run) {
    Box bx;
    make_box(bx);
    Mk aL[5], aU[5];
    MPMPC_UNROLL
    for (int j = 0; j < 5; ++j) {
      aL[j] = bx.Lm[j] & bit_(guess, j);
      aU[j] = bx.Um[j] & bit_(guess, 5 + j) & !aL[j];
    }
    const R zero(0.0);
    R xs[5], nus[3], lam[5];
    MPMPC_UNROLL
    for (int j = 0; j < 5; ++j) xs[j] = lam[j] = zero;
    MPMPC_UNROLL
    for (int i = 0; i < 3; ++i) nus[i] = zero;
    // two rounds at most: a guess that needs more is not worth more than the normal path (the slowest car of
    // the batch decides the step)
    SolverParams sw = st;
    sw.as_rounds = st.as_rounds < 2 ? st.as_rounds : 2;
    Mk okm = active_set_full(bx, aL, aU, xs, nus, lam, sw, st.as_add_fraction, run);
    R prim, stat;
    Mk cert = certificate(xs, nus, lam, st.cert_tol, prim, stat);
    Mk good = run & okm & cert;
    MPMPC_UNROLL
    for (int j = 0; j < 5; ++j) { x[j] = sel(good, xs[j], x[j]); yb[j] = sel(good, lam[j] / g[j], yb[j]); }
    MPMPC_UNROLL
    for (int i = 0; i < 3; ++i) yeq[i] = sel(good, nus[i], yeq[i]);
    pri_res = sel(good, prim, pri_res);
    dua_res = sel(good, stat, dua_res);
    status = seli(good, I(MPMPC_SOLVED), status);
    polished = seli(good, I(1), polished);
    act_bits = seli(good, pack_active(aL, aU), act_bits);
  }

  // `early`: also polish instances whose ADMM was stopped before it terminated (status UNSOLVED);
  // those keep UNSOLVED when the polish cannot certify them, so the caller can resume ADMM.
  MPMPC_HD void polish(const SolverParams& st, bool early) {
    Mk run = live & (polished != 1) &
             ((status == MPMPC_SOLVED) | (status == MPMPC_SOLVED_INACCURATE) | (status == MPMPC_MAX_ITER_REACHED));
    Mk unsolved = live & (status == MPMPC_UNSOLVED);
    if (early) run = run | unsolved;
    if (!L::wany(run)) return;
    Box bx;
    make_box(bx);
    // floor of the warm-started slacks / multipliers: the closer the ADMM point is to feasibility (unscaled
    // primal residual), the more its small slacks can be trusted:  theta = pri_res / 80  in [3e-4, 3e-3]
    const R zero(0.0), one(1.0);
    const R theta = min_(R(3e-3), max_(R(3e-4), pri_res * R(0.0125)));
    Ipm s;
    MPMPC_UNROLL
    for (int i = 0; i < 3; ++i) s.nu[i] = yeq[i];
    MPMPC_UNROLL
    for (int j = 0; j < 5; ++j) {
      R yv = yb[j] * g[j];
      s.x[j] = x[j];
      s.sl[j] = sel(bx.Lm[j], max_(x[j] - bx.lo[j], theta), one);
      s.su[j] = sel(bx.Um[j], max_(bx.hi[j] - x[j], theta), one);
      s.zl[j] = sel(bx.Lm[j], max_(-yv, theta), zero);
      s.zu[j] = sel(bx.Um[j], max_(yv, theta), zero);
      s.pi[j] = sel(bx.pin[j], yv, zero);
    }
    if (early && st.ipm_start_mu > 0.0) {
      // After early_polish (= 1) ADMM iterations the multipliers carry no information and the point violates every
      // speed bound: floors of 3e-3 cost 5-7 blocked steps.  Centred start instead (mpmpc_settings::ipm_start_*, row
      // space of the scaled problem: the slack of row g x is g times the slack of x, its multiplier 1 / g times).
      const R ths(st.ipm_start_slack);
      R mu0(st.ipm_start_mu);
      if (st.ipm_start_dual > 0.0) {
        // ... and multipliers commensurate with the dual residual they will have to balance: mu0 at least
        // ipm_start_dual x slack floor x |P x + q|_inf of the start point
        R rd0(0.0);
        [[maybe_unused]] R Pod[5] = {zero, zero, zero, zero, zero};
        if constexpr (FQ) Poff_add<0>(x, Pod);
        MPMPC_UNROLL
        for (int j = 0; j < 5; ++j) {
          R v = fma_(p[j], x[j], q[j]);
          if constexpr (FQ) v = v + Pod[j];
          rd0 = max_(rd0, sel(valid[j], abs_(v), zero));
        }
        mu0 = max_(mu0, (R(st.ipm_start_dual) * ths) * L::gmax(rd0));
      }
      MPMPC_UNROLL
      for (int i = 0; i < 3; ++i) s.nu[i] = zero;       // (the equality multipliers of that one iteration: worse than none)
      MPMPC_UNROLL
      for (int j = 0; j < 5; ++j) {
        const R fl = ths / g[j];
        s.sl[j] = sel(bx.Lm[j], max_(x[j] - bx.lo[j], fl), one);
        s.su[j] = sel(bx.Um[j], max_(bx.hi[j] - x[j], fl), one);
        s.zl[j] = sel(bx.Lm[j], mu0 / s.sl[j], zero);
        s.zu[j] = sel(bx.Um[j], mu0 / s.su[j], zero);
      }
    }
    // the interior-point stage runs in the split layout where the upper half-wave is free (kSplit), and on the
    // reduced problem where the time state separates (RED)
    constexpr int LAY = LAY_IP;
    constexpr int E = EN<LAY>;
    BoxT<LAY> bi;
    IpmT<LAY> si;
    R pp[E], qq[E];
    Mk vm[E];
    problem_in_layout<LAY>(bx, bi, pp, qq, vm);
    iterate_to_layout<LAY>(s, si);
    MPMPC_UNROLL
    for (int e = 0; e < E; ++e) {       // (before any step: multiplier above slack)
      si.tL[e] = bi.Lm[e] & (si.zl[e] > si.sl[e]);
      si.tU[e] = bi.Um[e] & (si.zu[e] > si.su[e]);
    }
    double tol = st.ipm_tol;
    Mk todo = run;
    for (int attempt = 0; attempt < 2; ++attempt) {
      stash();
      MPMPC_TICK_BEGIN(4);
      Mk conv = ipm<LAY>(bi, si, pp, qq, vm, st, tol, todo);
      MPMPC_TICK_END(4);
      // active-set guess of the interior point: the indicators of its last step (IpmT::tL, tU)
      Mk gL[E], gU[E], aL[5], aU[5];
      MPMPC_UNROLL
      for (int e = 0; e < E; ++e) { gL[e] = bi.Lm[e] & si.tL[e]; gU[e] = bi.Um[e] & si.tU[e]; }
      MPMPC_UNROLL
      for (int j = 0; j < 5; ++j) aL[j] = aU[j] = L::mfalse();
      mask_from_lay<LAY>(gL, aL); mask_from_lay<LAY>(gU, aU);
      MPMPC_UNROLL
      for (int j = 0; j < 5; ++j) { aL[j] = bx.Lm[j] & aL[j]; aU[j] = bx.Um[j] & aU[j] & !aL[j]; }
      R xs[5], nus[3], lam[5];
      MPMPC_UNROLL
      for (int j = 0; j < 5; ++j) { xs[j] = zero; lam[j] = zero; }
      from_lay<LAY>(si.x, xs);
      MPMPC_UNROLL
      for (int i = 0; i < 3; ++i) nus[i] = zero;
      MPMPC_UNROLL
      for (int i = 0; i < NR<LAY>; ++i) nus[i] = si.nu[i];
      MPMPC_TICK_BEGIN(5);
      // (the retry is more careful: only the upper half of the violations enters per round)
      const double frac = attempt == 0 ? st.as_add_fraction : (st.as_add_fraction > 0.5 ? st.as_add_fraction : 0.5);
      Mk okm = active_set_full(bx, aL, aU, xs, nus, lam, st, frac, todo & conv);
      MPMPC_TICK_END(5);
      unstash();
      R prim, stat;
      MPMPC_TICK_BEGIN(6);
      Mk cert = certificate(xs, nus, lam, st.cert_tol, prim, stat);
      MPMPC_TICK_END(6);
      Mk good = todo & conv & okm & cert;
      MPMPC_UNROLL
      for (int j = 0; j < 5; ++j) { x[j] = sel(good, xs[j], x[j]); yb[j] = sel(good, lam[j] / g[j], yb[j]); }
      MPMPC_UNROLL
      for (int i = 0; i < 3; ++i) yeq[i] = sel(good, nus[i], yeq[i]);
      pri_res = sel(good, prim, pri_res);
      dua_res = sel(good, stat, dua_res);
      status = seli(good, I(MPMPC_SOLVED), status);
      polished = seli(good, I(1), polished);
      act_bits = seli(good, pack_active(aL, aU), act_bits);
      todo = todo & conv & !good;       // a diverged interior-point run is not retried
      if (!L::wany(todo)) break;
      tol *= 1e-4;      // a wrong active-set guess means the centring was too loose: tighten it a lot
    }
    // whatever is left could not be certified: keep the ADMM iterate, flag it
    Mk failed = run & (polished != 1);
    if (early) failed = failed & !unsolved;      // an uncertified early attempt is not a verdict
    status = seli(failed, I(MPMPC_SOLVED_INACCURATE), status);
    polished = seli(failed, I(-1), polished);
  }

  // ======================================================================== phase 1
  // Is the instance infeasible?  (st.phase1; runs on what the early polish attempt could not certify, before any full
  // ADMM run.)   min 1/2 |w|^2  s.t.  the dynamics rows and pinned entries as they are,  lo <= x_j + w_j <= hi  on every
  // other entry with a finite side.  Always feasible; optimum 0 iff the QP is feasible; and at its optimum the
  // multipliers y = (nu, zu - zl) satisfy A'y = 0 and  u'max(y,0) + l'min(y,0) = -|w|^2:  a Farkas ray.  The ray is
  // then put to OSQP's own test (primal_infeasible: unscaled norms, at phase1_eps) - a solver-independent verdict,
  // reached in 5-10 interior-point iterations instead of the hundreds or thousands of ADMM iterations OSQP needs.
  // Certified instances: status PRIMAL_INFEASIBLE, x = least-violation point, (yeq, yb) = the ray, pri_res = largest
  // bound violation of x (unscaled).  Everything else is left untouched (status stays UNSOLVED).
  // Reduced problem (RED): the speed boxes are consistent by construction and the time state is free, so feasibility
  // is decided by the (e_y, e_psi, kappa) system alone; its ray has zero entries on the time rows and the speed boxes.
  MPMPC_HD void phase1(const SolverParams& st, const Mk& run) {
    if (!L::wany(run)) return;
    Box bx;
    make_box(bx);
    const R zero(0.0), one(1.0), theta(st.phase1_theta);
    // cold start in row space (x = 0, slacks max(distance to the bound, theta), multipliers theta), expressed in the
    // variable space the iteration works in: s_var = s_row / g, z_var = g z_row
    Ipm s;
    MPMPC_UNROLL
    for (int i = 0; i < 3; ++i) s.nu[i] = zero;
    MPMPC_UNROLL
    for (int j = 0; j < 5; ++j) {
      const R ig = one / g[j];
      s.x[j] = zero;
      s.sl[j] = sel(bx.Lm[j], max_(-lb[j], theta) * ig, one);
      s.su[j] = sel(bx.Um[j], max_(ub[j], theta) * ig, one);
      s.zl[j] = sel(bx.Lm[j], theta * g[j], zero);
      s.zu[j] = sel(bx.Um[j], theta * g[j], zero);
      s.pi[j] = zero;
    }
    constexpr int LAY = LAY_IP;
    constexpr int E = EN<LAY>;
    BoxT<LAY> bi;
    IpmT<LAY> si;
    R pp[E], qq[E];
    Mk vm[E];
    problem_in_layout<LAY>(bx, bi, pp, qq, vm);
    iterate_to_layout<LAY>(s, si);
    {
      // THE METRIC of the least violation (round 5).  What decides whether the reference's OSQP call returns a plan for an
      // infeasible QP is the point its ADMM iteration converges to: the minimiser of  sum_r rho_r (scaled violation of row r)^2
      // (rho on the box rows, a thousand times that on the dynamics rows: practically hard), which OSQP then puts to its
      // primal test  |Ax - z|_inf < eps_abs + eps_rel max(|Ax|, |z|).  Phase 1 minimises that same sum over the box rows: the
      // scaled violation of the box row of entry j is  g_j w_j  (g = the row's scaled entry, w the violation of the scaled
      // variable), hence the weights  om_j = g_j^2.  (Rounds 2 - 4 used om = 1 - unit weight on the violation of the scaled
      // VARIABLE and left at the first iterate with a valid ray; on config 4 that took the other branch than restated stock
      // OSQP on 28 of 8 192 instances, this on 5 - the five OSQP abandons at max_iter: profiles/r5/branch_agreement.txt.
      // More Ruiz passes before phase 1 - OSQP's row scalings after ten passes instead of the early attempt's one - were tried
      // and changed no verdict on configs 4 and 5.)
      R om5[5], omL[E];
      MPMPC_UNROLL
      for (int j = 0; j < 5; ++j) om5[j] = g[j] * g[j];
      to_lay<LAY>(om5, omL);
      MPMPC_UNROLL
      for (int e = 0; e < E; ++e) pp[e] = sel(omL[e] > zero, omL[e], one);
      // ... and what turns the iterate's w = (zl - zu) / om into an unscaled violation: D / om (read in place of the cost vector)
      R dL[E];
      to_lay<LAY>(D, dL);
      MPMPC_UNROLL
      for (int e = 0; e < E; ++e) qq[e] = dL[e] / pp[e];
      // the band below which "marginal" is an open question: phase1_band times OSQP's primal tolerance at the largest finite bound
      R nb(0.0);
      MPMPC_UNROLL
      for (int j = 0; j < 5; ++j) {
        const R lo0 = lo_raw(j), hi0 = hi_raw(j);
        nb = max_(nb, sel(valid[j], max_(sel(lo0 > R(-INF_BOUND), abs_(lo0), zero), sel(hi0 < R(INF_BOUND), abs_(hi0), zero)), zero));
      }
      p1_band = st.phase1_accept ? R(st.phase1_band) * fma_(R(st.eps_rel), L::gmax(nb), R(st.eps_abs)) : zero;
    }
    p1_converged = L::mfalse();
    stash();
    // phase 1 has no use for the cost: it waits in cold storage as well (slots COLD_COST ..)
    MPMPC_UNROLL
    for (int j = 0; j < 5; ++j) { L::cold_put(COLD_COST + j, p[j]); L::cold_put(COLD_COST + 5 + j, q[j]); }
    L::fence();
    MPMPC_TICK_BEGIN(9);
    // (phase 1 converges two digits further than the polish: for an instance infeasible by a tenth of a millimetre the
    //  quantities of the verdict - the ray's support - are themselves at the 1e-9 level)
    // (a looser tolerance under phase1_accept - the marginal instances are not refused any more - was tried: at 1e-8 a few
    //  instances end without a verdict and fall into the ADMM run, config 4 25.7 -> 11.1 M solves/s; not kept)
    ipm<LAY, true>(bi, si, pp, qq, vm, st, st.ipm_tol * 1e-2 < 1e-11 ? st.ipm_tol * 1e-2 : 1e-11, run);
    MPMPC_TICK_END(9);
    L::fence();
    MPMPC_UNROLL
    for (int j = 0; j < 5; ++j) { p[j] = L::cold_get(COLD_COST + j); q[j] = L::cold_get(COLD_COST + 5 + j); }
    // back to five entries per lane: point and ray (lam = zu - zl + pi in variable space, yb = lam / g in row space)
    R xs[5], lam[5], nus[3], l3[E];
    MPMPC_UNROLL
    for (int j = 0; j < 5; ++j) { xs[j] = zero; lam[j] = zero; }
    MPMPC_UNROLL
    for (int e = 0; e < E; ++e) l3[e] = si.zu[e] - si.zl[e] + si.pi[e];
    from_lay<LAY>(l3, lam);
    from_lay<LAY>(si.x, xs);
    MPMPC_UNROLL
    for (int i = 0; i < 3; ++i) nus[i] = zero;
    MPMPC_UNROLL
    for (int i = 0; i < NR<LAY>; ++i) nus[i] = si.nu[i];
    unstash();
    if constexpr (RED) {
      // a point of the full problem: any speed inside its box (its lower end), the time state rolled forward;
      // the ray gets no entry from either (lam[3] must stay zero: reduced_complete would put the cost gradient there)
      reduced_complete(bx, xs, nus, lam);
      lam[3] = zero;
    }
    // OSQP's test reads its ray from the cold slots of the last dual step
    MPMPC_UNROLL
    for (int j = 0; j < 5; ++j) L::cold_put(COLD_DYB + j, sel(valid[j], lam[j] / g[j], zero));
    MPMPC_UNROLL
    for (int i = 0; i < 3; ++i) L::cold_put(COLD_DYEQ + i, sel(vx, nus[i], zero));
    L::fence();
    // Two ways to a verdict.  (A) OSQP's test at phase1_eps - any iterate whose ray passes is a certificate, the loop
    // stops at the first one.  (B) phase 1 ran to its converged optimum and that optimum still violates a bound by
    // more than cert_tol: the least violation is positive, the problem is infeasible however small the margin - taken
    // when the ray's support is negative by at least a hundred times its own residual |A'y| (at convergence the
    // residual is at the 1e-9 level, so this decides instances infeasible by well under a micrometre, which (A) at any
    // fixed eps leaves to hundreds of ADMM iterations that end in "solved inaccurate").
    R f_nrm, f_lhs, f_m, prim, stat;
    farkas_values(f_nrm, f_lhs, f_m);
    certificate(xs, nus, lam, st.cert_tol, prim, stat);
    const R eps1(st.phase1_eps);
    Mk certA = (f_nrm > eps1) & (f_lhs < -eps1 * f_nrm) & (f_m < eps1 * f_nrm);
    Mk certB = p1_converged & (prim > R(st.cert_tol)) & (f_nrm > R(0.0)) & (f_lhs < R(-100.0) * f_m) & (f_lhs < R(0.0));
    Mk cert = run & (certA | certB);
    // MARGINALLY infeasible (phase1_accept, default): the least violation is below the primal tolerance at which the
    // reference's own OSQP call stops and returns a plan - eps_abs + eps_rel max(|Ax|, |z|) at eps = 1e-3, i.e. corridor
    // violations of millimetres (src/MPC.py:159,183 run OSQP at its defaults; the reference then DRIVES that plan instead
    // of taking its fallback branch).  Such an instance is not reported infeasible: every box its least-violation point
    // leaves is widened to 1.5 times that violation, the polish runs once more from that point (pass 1 of run(), like a
    // feasible instance), and the result is returned as SOLVED_INACCURATE with the violation in resid[0].
    p1_marginal = L::mfalse();
    p1_viol = R(0.0);
    if (st.phase1_accept) {
      // max(|Ax|, |z|) of OSQP's test: the identity rows make it the largest entry of the plan - where the corridor cannot
      // be met the steering sits on its bound, so the largest finite box bound of the instance stands for the iterate OSQP
      // would stop at (stock limits: 1e-3 + 1e-3 x 6.47 = 7.5 mm)
      R nAx(0.0);
      MPMPC_UNROLL
      for (int j = 0; j < 5; ++j) {
        const R lo0 = lo_raw(j), hi0 = hi_raw(j);
        R m = abs_(D[j] * xs[j]);
        m = max_(m, sel(lo0 > R(-INF_BOUND), abs_(lo0), R(0.0)));
        m = max_(m, sel(hi0 < R(INF_BOUND), abs_(hi0), R(0.0)));
        nAx = max_(nAx, sel(valid[j], m, R(0.0)));
      }
      nAx = L::gmax(nAx);
      p1_marginal = cert & !(prim > fma_(R(st.eps_rel), nAx, R(st.eps_abs)));
      cert = cert & !p1_marginal;
      if (L::wany(p1_marginal)) {
        p1_viol = sel(p1_marginal, prim, R(0.0));
        MPMPC_UNROLL
        for (int j = 0; j < 5; ++j) {
          const R xu = D[j] * xs[j], lo0 = lo_raw(j), hi0 = hi_raw(j);
          const R wl = sel(p1_marginal & valid[j] & (lo0 > R(-INF_BOUND)), max_(lo0 - xu, R(0.0)), R(0.0));
          const R wh = sel(p1_marginal & valid[j] & (hi0 < R(INF_BOUND)), max_(xu - hi0, R(0.0)), R(0.0));
          const R lo1 = fma_(R(-1.5), wl, lo0), hi1 = fma_(R(1.5), wh, hi0);
          L::cold_put(COLD_RAW + 3 + j, lo1);
          L::cold_put(COLD_RAW + 8 + j, hi1);
          lb[j] = sel(p1_marginal, Eb[j] * lo1, lb[j]);
          ub[j] = sel(p1_marginal, Eb[j] * hi1, ub[j]);
        }
        L::fence();
      }
    }
    // FEASIBLE to tolerance: phase 1 converged and its point violates nothing.  That point - inside every box, well
    // centred by the barrier - is handed back as the start of a second polish attempt (Solver::run, pass 1).
    p1_feasible = run & !cert & ((p1_converged & !(prim > R(st.cert_tol))) | p1_marginal);
    MPMPC_UNROLL
    for (int j = 0; j < 5; ++j) { x[j] = sel(p1_feasible, xs[j], x[j]); yb[j] = sel(p1_feasible, zero, yb[j]); }
    MPMPC_UNROLL
    for (int i = 0; i < 3; ++i) yeq[i] = sel(p1_feasible, zero, yeq[i]);
    pri_res = sel(p1_feasible, R(1.0), pri_res);          // (the polish then floors slacks and multipliers at its largest value)
#ifdef MPMPC_EMU_DEBUG
    std::fprintf(stderr, "phase1: nrm %.3e lhs %.3e m %.3e prim %.3e converged %d A %d B %d ipm_iters %d\n", f_nrm.v[16], f_lhs.v[16], f_m.v[16],
                 prim.v[16], (int)p1_converged.v[16], (int)certA.v[16], (int)certB.v[16], ipm_iters.v[16]);
#endif
    MPMPC_UNROLL
    for (int j = 0; j < 5; ++j) { x[j] = sel(cert, xs[j], x[j]); yb[j] = sel(cert, lam[j] / g[j], yb[j]); }
    MPMPC_UNROLL
    for (int i = 0; i < 3; ++i) yeq[i] = sel(cert, nus[i], yeq[i]);
    pri_res = sel(cert, prim, pri_res);
    dua_res = sel(cert, zero, dua_res);
    status = seli(cert, I(MPMPC_PRIMAL_INFEASIBLE), status);
  }

  // ======================================================================== output
  // z in the reference's ordering, u0 = (v_0, delta_0), multipliers in the reference's row order
  MPMPC_HD void store(const I& inst, const I& k, double wheelbase, double* z, double* u0, int* st_out,
                      int* it_out, double* resid, double* y, int* act = nullptr, int ld = 0) const {
    if (act) L::storei(act, inst * ld + k, vx, act_bits);
    const int n = 5 * N + 3, m = 8 * N + 6;
    R cinv = R(1.0) / c;
    // whole rows are staged per wave and written as consecutive doubles (lane backends: rows())
    if (z) {
      L::rows(z, n, inst, n_inst, [&](auto put) {
        MPMPC_UNROLL
        for (int i = 0; i < 3; ++i) put(k * 3 + i, vx, D[i] * x[i]);
        put(k * 2 + (3 * (N + 1)), vu, D[3] * x[3]);
        put(k * 2 + (3 * (N + 1) + 1), vu, D[4] * x[4]);
      });
    }
    if (y) {
      L::rows(y, m, inst, n_inst, [&](auto put) {
        MPMPC_UNROLL
        for (int i = 0; i < 3; ++i) {
          put(k * 3 + i, vx, (Eeq[i] * yeq[i]) * cinv);
          put(k * 3 + (3 * (N + 1) + i), vx, (Eb[i] * yb[i]) * cinv);
        }
        put(k * 2 + (6 * (N + 1)), vu, (Eb[3] * yb[3]) * cinv);
        put(k * 2 + (6 * (N + 1) + 1), vu, (Eb[4] * yb[4]) * cinv);
      });
    }
    Mk lead = live & first;
    if (u0) {
      L::store(u0, inst * 2, lead, D[3] * x[3]);
      L::store(u0, inst * 2 + 1, lead, atan_((D[4] * x[4]) * R(wheelbase)));   // src/MPC.py:188-189
    }
    if (st_out) L::storei(st_out, inst, lead, status);
    if (it_out) { L::storei(it_out, inst * 2, lead, iters); L::storei(it_out, inst * 2 + 1, lead, ipm_iters); }
    if (resid) { L::store(resid, inst * 2, lead, pri_res); L::store(resid, inst * 2 + 1, lead, dua_res); }
  }

  // mode 0: early attempt + full run; 1: early attempt only (uncertified instances stay UNSOLVED);
  // 2: full run only (the second launch of a packed batch, see mpmpc_solve_kernel)
  // guess: shifted active set of the previous closed-loop step (bit 30 set where there is one), or 0
  // (WARM is a template flag so that the batch kernels do not carry the warm-start code at all)
  // base_ipm: interior-point iterations the instance already spent in an earlier launch (mode 2)
  // P1: carry the phase-1 code (the packed kernels do not: their launches hand what they cannot certify to a
  //     one-instance-per-wave launch, mode 2, and that is where phase 1 runs)
  template <bool WARM = false, bool P1 = true>
  MPMPC_HD void run(const R* fields, int B, const I& inst, const I& k, int N_, const SolverParams& st,
                    int mode = 0, const I& guess = I(0), const I& base_ipm = I(0), const double* qn_off = nullptr) {
    MPMPC_TICK_BEGIN(0);
    load(fields, B, inst, k, N_, qn_off);
    MPMPC_TICK_END(0);
    // The polish does not need a converged ADMM point, only a reasonable one: with early_polish > 0
    // it is first tried after that many iterations, on a problem that has seen early_scaling of the
    // Ruiz passes (stage 0).  What it cannot certify is first asked whether it is feasible at all (phase 1, on the
    // same scaling) and then gets the remaining passes and goes through the full OSQP iteration from a cold start,
    // exactly as without the shortcut, and is polished again (stage 1).  One loop, so admm() / polish() are
    // instantiated once.
    const bool two_stage = st.polish && st.early_polish > 0 && st.early_polish < st.max_iter;
    bool early = two_stage && mode != 2;
    int limit = early ? st.early_polish : st.max_iter;
    int passes = two_stage && st.early_scaling > 0 && st.early_scaling < st.scaling ? st.early_scaling : st.scaling;
    // An EMPTY box - a lower bound above its upper bound, e.g. the curvature-dependent speed cap of src/MPC.py:111-113
    // below umin[0] - makes the QP trivially infeasible.  OSQP refuses such data at setup (the reference would raise
    // there); here the instance is reported infeasible (verdict written at the end of run) and takes no part in the solve:
    // the polish cannot certify it and phase 1's reduced ray has no speed entry, so it would sit through the whole ADMM run
    // and hold its wave for it.
    auto box_gap = [&]() {
      R gap(0.0);
      MPMPC_UNROLL
      for (int j = 0; j < 5; ++j) gap = max_(gap, sel(valid[j], lo_raw(j) - hi_raw(j), R(0.0)));
      return L::gmax(gap);
    };
    // (`live` itself is narrowed for the duration of the solve, so that every stage of it - the polish and phase 1 form
    //  their own masks from it - leaves the instance alone; one mask lives through the solve, the gap is formed again at the end)
    const Mk live_all = live;
    live = live_all & !(box_gap() > R(0.0));
    Mk which = live;
    // (an instance that takes no part keeps a defined point: zero)
    MPMPC_UNROLL
    for (int j = 0; j < 5; ++j) { x[j] = R(0.0); zb[j] = R(0.0); yb[j] = R(0.0); }
    MPMPC_UNROLL
    for (int i = 0; i < 3; ++i) { zeq[i] = R(0.0); yeq[i] = R(0.0); }
    status = I(MPMPC_UNSOLVED); iters = I(0); ipm_iters = I(0); polished = I(0);
    pri_res = R(0.0); dua_res = R(0.0);
    act_bits = I(0);
    Mk warm = L::mfalse();
    if constexpr (WARM) warm = L::gany(live & bit_(guess, 30));
    // Pass 0: the early attempt.  Pass 1 (only for instances phase 1 found FEASIBLE): a second attempt of the polish
    // from phase 1's point - strictly inside every box it can be inside of, well centred - because the warm-started
    // interior point of pass 0 occasionally jams next to a degenerate vertex (slack / multiplier pairs far off the
    // central path; seen once in 12 000 randomised instances, at N = 3): ~10 interior-point iterations instead of the
    // hundreds of ADMM iterations plus a polish that jams again.  Pass 2: the remaining Ruiz passes and the full OSQP run.
    Mk retry = L::mfalse();
    p1_marginal = L::mfalse();
    p1_viol = R(0.0);
    _Pragma("nounroll")
    for (int pass = 0; pass < 3; ++pass) {
      if (pass == 1 && !L::wany(retry)) continue;
      if (pass != 1) {
        MPMPC_TICK_BEGIN(1);
        scale(passes, which);
        MPMPC_TICK_END(1);
        passes = st.scaling - passes;
      }
      if (WARM && pass == 0 && mode != 2 && st.polish && L::wany(warm)) {
        status = I(MPMPC_UNSOLVED); iters = I(0); ipm_iters = I(0); polished = I(0);
        pri_res = R(0.0); dua_res = R(0.0);
        MPMPC_UNROLL
        for (int j = 0; j < 5; ++j) { x[j] = R(0.0); yb[j] = R(0.0); }
        MPMPC_UNROLL
        for (int i = 0; i < 3; ++i) yeq[i] = R(0.0);
        warm_polish(st, guess, warm);
        which = live & (status == MPMPC_UNSOLVED);
        if (!L::wany(which)) break;
      }
      if (pass == 0 && mode == 2 && two_stage) {
        // second launch of a packed batch: the early attempt was made (and failed) in the first one
        status = I(MPMPC_UNSOLVED); iters = I(st.early_polish); ipm_iters = base_ipm; polished = I(0);
        pri_res = R(0.0); dua_res = R(0.0);
        MPMPC_UNROLL
        for (int j = 0; j < 5; ++j) { x[j] = R(0.0); yb[j] = R(0.0); }
        MPMPC_UNROLL
        for (int i = 0; i < 3; ++i) yeq[i] = R(0.0);
      } else {
        if (pass != 1) {
          MPMPC_TICK_BEGIN(2);
          if (early && limit == 1 && st.early_start == 0) zero_start(st, which);
          else if (RED && early && limit == 1) reduced_start(st, which);
          else admm(st, which, limit);
          MPMPC_TICK_END(2);
        }
        const bool attempt = early || pass == 1;      // an attempt leaves what it cannot certify UNSOLVED
        MPMPC_TICK_BEGIN(3);
        if (st.polish) polish(st, attempt);
        MPMPC_TICK_END(3);
        if constexpr (P1) {
          if (pass == 1) {
            // a marginally infeasible instance (phase1_accept) ends here: the optimum over the relaxed boxes, or - if the
            // polish could not certify that - phase 1's least-violation point itself; either way a usable, inaccurate plan
            const Mk mg = live & p1_marginal & ((status == MPMPC_SOLVED) | (status == MPMPC_UNSOLVED));
            status = seli(mg, I(MPMPC_SOLVED_INACCURATE), status);
            pri_res = sel(mg, p1_viol, pri_res);
          }
        }
        which = live & (status == MPMPC_UNSOLVED);
        if (!attempt || mode == 1 || !L::wany(which)) break;
      }
      if (pass == 1) { early = false; limit = st.max_iter; continue; }
      // end of the early stage: is what it could not certify feasible at all?  (still on the early scaling)
      retry = L::mfalse();
      if constexpr (P1) {
        if (st.phase1) {
          phase1(st, which);
          // (what phase 1 certifies keeps the ADMM iteration count of the early attempt it followed)
          iters = seli(which & (status == MPMPC_PRIMAL_INFEASIBLE), I(st.early_polish), iters);
          which = live & (status == MPMPC_UNSOLVED);
          if (!L::wany(which)) break;
          retry = which & p1_feasible;
        }
      }
      if (!L::wany(retry)) { early = false; limit = st.max_iter; }
    }
    // the empty-box verdict: a zero ray (no Farkas ray exists for a single empty interval row) and the width of the gap as
    // its violation
    {
      const Mk empty = live_all & !live;
      live = live_all;
      status = seli(empty, I(MPMPC_PRIMAL_INFEASIBLE), status);
      pri_res = sel(empty, box_gap(), pri_res);
      MPMPC_UNROLL
      for (int j = 0; j < 5; ++j) yb[j] = sel(empty, R(0.0), yb[j]);
      MPMPC_UNROLL
      for (int i = 0; i < 3; ++i) yeq[i] = sel(empty, R(0.0), yeq[i]);
    }
    // non-finite data (a NaN pose, say) must not come back as a "solved" plan: no verdict at all, so
    // that the caller takes its fallback branch (src/MPC.py:208-220) instead of driving NaN controls
    Mk bad = L::mfalse();
    MPMPC_UNROLL
    for (int j = 0; j < 5; ++j) bad = bad | (valid[j] & !(abs_(x[j]) < R(1e300)));
    status = seli(live & L::gany(bad), I(MPMPC_UNSOLVED), status);
  }
};

}  // namespace mpmpc
