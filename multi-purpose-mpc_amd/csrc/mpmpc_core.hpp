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
#include "mpmpc_assemble.hpp"

#ifndef MPMPC_HD
#define MPMPC_HD inline
#endif
#ifndef MPMPC_UNROLL
#define MPMPC_UNROLL _Pragma("unroll")
#endif
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
  // kS2: TWO stages per lane (lane_pair.hpp) - the reduced problem's factorisation is factor_core2_s2 / s_solve_s2 (mpmpc_solver_s2.hpp)
  static constexpr bool kS2 = (L::stages_per_lane == 2);
  static constexpr bool kCR = CR && !kS2 && (L::split == 16 || L::split == 32 || L::split == 64 || L::split == 128);
  static constexpr bool kCR32 = kCR && L::split == 32;      // a chain is TWO rows of 16 lanes (see factor_cr2)
  // ... FOUR rows - a whole wavefront of a 128-lane workgroup - or EIGHT, two wavefronts of a 256-lane one (LaneBlock): the
  // survivors of the rows are eliminated one after the other, kCRrows - 1 steps
  static constexpr bool kCR64 = kCR && (L::split == 64 || L::split == 128);
  static constexpr int kCRrows = L::split / 16;
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
  // NV lane exchanges at once.  Where the lanes of an instance sit in different wavefronts (LaneBlock: horizons above 63) every
  // exchange is a pass through LDS between two workgroup barriers, and the NV values of a step share one pass; on a
  // wavefront these are the plain per-value DPP moves - the same instructions as before.
  // MODE (only where L::staged_sweeps - a chain of the factorisation spans TWO wavefronts, LaneBlock<256>): 0 = the plain
  // exchange; 1 / 2 = the two stages of a staged sweep (staged_sweep below).
  template <int M> struct SweepMode { static constexpr int value = M; };
  template <int NV, int MODE = 0> MPMPC_HD static void cup_n(const R* v, R* o) {
    if constexpr (MODE != 0) L::template chain_shift<NV, -1, MODE>(v, o);
    else if constexpr (L::batched) L::template cupv<NV>(v, o);
    else { MPMPC_UNROLL for (int i = 0; i < NV; ++i) o[i] = L::cup(v[i]); }
  }
  template <int NV, int MODE = 0> MPMPC_HD static void cdown_n(const R* v, R* o) {
    if constexpr (MODE != 0) L::template chain_shift<NV, +1, MODE>(v, o);
    else if constexpr (L::batched) L::template cdownv<NV>(v, o);
    else { MPMPC_UNROLL for (int i = 0; i < NV; ++i) o[i] = L::cdown(v[i]); }
  }
  // A sequential sweep along chains that span two wavefronts (L::staged_sweeps).  In the lock-step form every lane executes
  // every step and every step passes through LDS between two workgroup barriers, although lane k's value is final after k
  // steps and only ONE value per chain crosses the wavefront edge.  Staged: the first wavefront of each chain (in the
  // direction of the sweep) runs its 64 steps alone, shifting in registers, and its edge lane leaves what it hands on in
  // LDS - one more step, which recomputes the final values, leaves the FINAL hand-on there; a barrier; the second wavefront
  // runs the remaining steps with that value flowing in at its edge lane at every step.  Two barriers per sweep instead of
  // two per step, and each wavefront executes half the steps.  Every lane ends with the value of the lock-step form: its
  // last step sees its predecessor's final value either way.  step(SweepMode<m>) must do all its chain shifts through
  // cup_n / cdown_n<NV, m> and nothing else that talks across lanes.
  template <int DIR, class F> MPMPC_HD static void staged_sweep(int steps, F step) {
    // (The chains are right-aligned - they end at the meeting lane - so the wavefront next to the junction is full and the
    //  other one holds the rest: an inward sweep, DIR -1, runs the partial wavefront first and then the full one, 64 steps;
    //  an outward sweep the full one first and then the steps that are left.)
    const bool first = L::sweep_first(DIR);
    const int second = DIR < 0 ? 64 : steps - 64;
    if (first) { for (int s = 0; s < 65; ++s) step(SweepMode<1>{}); }
    L::sync();
    if (!first) { for (int s = 0; s < second; ++s) step(SweepMode<2>{}); }
    L::sync();
  }
  template <int NV> MPMPC_HD static void mirror_n(const R* v, R* o) {
    if constexpr (L::batched) L::template mirrorv<NV>(v, o);
    else { MPMPC_UNROLL for (int i = 0; i < NV; ++i) o[i] = L::mirror(v[i]); }
  }
  template <int NV> MPMPC_HD static void up_n(const R* v, R* o) {
    if constexpr (L::batched) L::template upv<NV>(v, o);
    else { MPMPC_UNROLL for (int i = 0; i < NV; ++i) o[i] = L::up(v[i]); }
  }
  template <int NV> MPMPC_HD static void down_n(const R* v, R* o) {
    if constexpr (L::batched) L::template downv<NV>(v, o);
    else { MPMPC_UNROLL for (int i = 0; i < NV; ++i) o[i] = L::down(v[i]); }
  }
  // w = A_k x_k + B_k u_k  (contribution of this stage to equality block k+1)
  MPMPC_HD void couple(const R v[5], R w[3]) const {
    w[0] = fma_(a[1], v[1], a[0] * v[0]);
    w[1] = fma_(b[0], v[4], fma_(a[3], v[1], a[2] * v[0]));
    w[2] = fma_(b[1], v[3], fma_(a[5], v[2], a[4] * v[0]));
  }
  MPMPC_HD void Aeq_mul(const R v[5], R r[3]) const {
    R w[3], wu[3];
    couple(v, w);
    up_n<3>(w, wu);
    MPMPC_UNROLL
    for (int i = 0; i < 3; ++i) r[i] = fma_(mI[i], v[i], wu[i]);
  }
  MPMPC_HD void AeqT_mul(const R nu[3], R t[5]) const {
    R nd[3];
    down_n<3>(nu, nd);
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

#define MPMPC_SOLVER_BODY 1
#include "mpmpc_solver_linalg.hpp"

#include "mpmpc_solver_s2.hpp"

#include "mpmpc_solver_admm.hpp"

#include "mpmpc_solver_ipm.hpp"

#include "mpmpc_solver_polish.hpp"

#include "mpmpc_solver_phase1.hpp"
#undef MPMPC_SOLVER_BODY

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
