// Reduced-NATIVE solver of the batch path: a lane never holds the 3-state problem.
//
// Where the reduction of mpmpc_core.hpp applies (reducible(): the time state t carries neither cost nor bound, e_psi has
// no bound, R[0] > 0 - the reference's own tracking weights, src/simulation.py:101-111) the QP of src/MPC.py:61-155
// separates EXACTLY into
//     v_k = clip(v_ref_k, umin[0], min(umax[0], vmax_k))     closed form, in the unscaled problem, done at load time
//     t_k                                                      roll-forward through its own equality rows, after the solve
//     the QP in (e_y, e_psi, kappa)                            2 x 2 blocks, 2 equality rows and 3 entries per stage
// The general kernels (Solver::run) assemble, scale and start on the FULL problem and only run the polish on the reduced
// one - they carry the OSQP iteration, phase 1 and the 5-entry problem through the whole solve: 440-512 registers, a
// 120 KB kernel, one wavefront per SIMD.  This solver keeps ONLY the reduced problem: own Ruiz pass on its 3 columns /
// 2 + 3 rows, the interior point from x = 0 (no OSQP iterate: measured, the start from OSQP's first iterate cost more than it saved),
// and the active-set rounds of Solver (ipm<LAY_RED / LAY_REDSPLIT>, active_set<LAY_RED>: the same code), a KKT certificate
// of the reduced problem in unscaled units - the separated parts satisfy their KKT rows by construction: the speed's
// multiplier is -(R0 v + q_v) with the sign its clip gives it, the time rows hold to rounding with zero multipliers, and
// tests/ check every answer against the FULL problem's KKT system with numpy (mpmpc_testlib.kkt_batch).
// What it cannot certify - infeasible and very hard instances - keeps status UNSOLVED: the launcher appends it to the
// tail list, and the reduced-native tail solver (mpmpc_reduced_tail.hpp: phase 1 and one more attempt, same footing) decides
// it; what even that leaves goes to the general one-instance-per-wave kernel (mode 2: phase 1, full OSQP run).
//
// Replaces, per instance: MPC._init_problem + osqp setup/solve (src/MPC.py:61-159,183) for the default settings.
#pragma once
#include "mpmpc_core.hpp"

namespace mpmpc {

// May the batch path run this solver?  (the launcher and the emulation ask the same question)
inline bool reduced_native(const mpmpc_config& c, const mpmpc_settings& st) {
  return st.native != 0 && reducible(c, st) && st.early_polish == 1 && st.max_iter > 1 && st.ipm_start_mu > 0.0 &&
         st.early_scaling >= 0 && st.scaling > 0;
}

// CR: cyclic-reduction elimination of the factorisation's chains (Solver::kCR; on for the shipped kernels)
template <class L, bool CR = true>
struct ReducedSolver : Solver<L, false, true, false, CR> {
  using S = Solver<L, false, true, false, CR>;
  using R = typename L::real;
  using Mk = typename L::mask;
  using I = typename L::ival;
  using S::N; using S::n_inst; using S::off_; using S::vx; using S::vu; using S::first; using S::down_chain; using S::is_mid;
  using S::is_end; using S::vxc; using S::sU; using S::val3; using S::bU; using S::live; using S::a; using S::b; using S::mI;
  using S::leq; using S::status; using S::iters; using S::ipm_iters; using S::polished; using S::pri_res; using S::dua_res;
  static constexpr bool kSplit = S::kSplit;
  static constexpr int LAY_RED = S::LAY_RED, LAY_REDSPLIT = S::LAY_REDSPLIT;
  static constexpr int LAY_IP = kSplit ? LAY_REDSPLIT : LAY_RED;
  static constexpr int EI = kSplit ? 2 : 3;                    // entries per lane in the interior point's layout
  using Box3 = typename S::template BoxT<LAY_RED>;
  using BoxI = typename S::template BoxT<LAY_IP>;
  using IpmI = typename S::template IpmT<LAY_IP>;

  // ---- cold storage (LDS), 512 B per slot: what only setup, the certificate and the store need
  //   C_D, C_E, C_C        scalings D (3 columns), E (2 dynamics rows), cost scaling c
  //   K_LO0 .. K_HI2       box of e_y and kappa in the scaled variable space (lo_raw / D, hi_raw / D)
  //   C_V .. C_BEQ2        the separated parts: v, its multiplier, a20, b20 v, rhs of the time row
  //   C_GAP                width of an empty box (resid[0] of such an instance)
  //   C_G, C_PI            box-row scaling and pin multipliers of the start; later (same slots) C_XS, C_LAM: the certified point
  //   C_NUS                its equality multipliers
  //   K_PP .. K_RP         the packed interior point (ipm3): cost, equality offsets, residuals of the iteration
  //   K_PARK .. +7         the interior point's slacks and bound multipliers while the active-set rounds run (the iteration's
  //                        residual slots are idle then): they are needed again only by a second attempt
  enum { C_D = 0, C_E = 3, C_C = 5, K_LO0 = 6, K_HI0 = 7, K_LO2 = 8, K_HI2 = 9, C_V = 10, C_LAMV = 11, C_A20 = 12, C_BV = 13, C_BEQ2 = 14,
         C_G = 15, C_PI = 18, C_XS = 15, C_LAM = 18, C_NUS = 21, C_GAP = 23,
         K_PP = 24, K_QQ = 27, K_LEQ = 30, K_RD = 32, K_RP = 35, K_PARK = 32, COLD_USED = 40 };
  // LEAN cold storage (a backend with 37 .. 39 slots: the 128-lane workgroup of the pair layout, whose LDS must leave room for a
  // second workgroup on the CU): the last three of the eight parked values go to the slots of the start's box-row scalings
  // (C_G: read before the first interior-point iteration only) instead of three slots of their own, and a point that is not
  // certified is not written over them (commit) - one instance per execution group only
  static constexpr bool kLean = L::cold_slots < COLD_USED;
  static_assert(L::cold_slots >= (kLean ? 37 : COLD_USED) && (!kLean || L::per_wave == 1), "lane backend has too few cold slots");
  MPMPC_HD static constexpr int park_slot(int i) { return (!kLean || i < 5) ? K_PARK + i : C_G + (i - 5); }
  // a scaled bound beyond this is "infinite" (raw infinities are +-1e30, the Ruiz factors stay within [1e-4, 1e4] per pass)
  static constexpr double BOX_INF = 1e20;
  // Attempts: interior point to native_ipm_tol, active-set rounds, certificate; what the rounds cannot settle is taken up
  // again at a 100 x tighter tolerance, twice if need be (1e-7, 1e-9, 1e-11: one to two more iterations each).  (Two attempts
  // at 1e-8 / 1e-12 before: the loose first attempt saves an iteration on most instances, the finer ladder keeps the rare
  // repeat short - B = 4 096: slowest instance 9 -> 8 iterations.)
  static constexpr int RN_ATTEMPTS = 3;
  static constexpr double RN_RETRY = 1e-2;
  static constexpr int RN_IPM_CAP = 16;       // interior-point iterations of the attempt (see run())

  // ---- the reduced problem, scaled; entries e = (e_y, e_psi, kappa)
  R P3[3], Q3[3];
  R x3[3];               // start point (the certified point and its multipliers wait in cold storage: C_XS, C_LAM, C_NUS)
  Mk val[3];             // entry exists: (vx, vx, vu)
  Mk solvable, empty;    // instance exists and no box of it is empty; instance exists and a box is empty

  // ================================================================================ setup: load + Ruiz + cold
  MPMPC_HD void setup(const R* fields, int B, const I& inst, const I& k, int N_, const SolverParams& st) {
    context(B, inst, k, N_);
    setup_problem(fields, st);
  }
  // lane context: which stage this lane holds, where the elimination chains of its instance run
  MPMPC_HD void context(int B, const I& inst, const I& k, int N_) {
    N = N_;
    n_inst = B;
    live = inst < B;
    vx = live & within_(k, 0, N);
    vu = live & within_(k, 0, N - 1);
    first = (k == 0);
    {
      const int C = L::split;
      off_ = lane_offset(L::group, C, N);
      I kl = k + off_;
      down_chain = (kl >= C);
      is_mid = (kl == C - 1);
      is_end = (kl == 2 * C - 1);
      I kc = seli(down_chain & (kl < 2 * C), kl * (-1) + (3 * C - 1), kl) - off_;
      vxc = live & within_(kc, 0, N);
      if constexpr (kSplit) {
        sU = (kl >= 32);
        Mk vU = live & within_(kl, 32 + off_, N + 31 + off_);
        val3[0] = selb(sU, vU, vx); val3[1] = selb(sU, vU, vx); val3[2] = vx & !sU;
      }
    }
    val[0] = vx; val[1] = vx; val[2] = vu;
  }
  MPMPC_HD void setup_problem(const R* fields, const SolverParams& st) {
    auto fld = [&](int f, double dflt) { return sel(vx, fields[f], R(dflt)); };
    const R zero(0.0), onec(1.0);
    // ---- the separated parts, in the UNSCALED problem
    const R lo_v = max_(fld(F_LO + 3, -INFTY), R(-INFTY)), hi_v = min_(fld(F_HI + 3, INFTY), R(INFTY));
    const R p_v = fld(F_P + 3, 1.0), q_v = fld(F_Q + 3, 0.0);
    R v = -q_v / p_v;                                   // minimiser of 1/2 p v^2 + q v  (= v_ref: src/MPC.py:155)
    v = sel((hi_v < R(INF_BOUND)) & (v > hi_v), hi_v, v);
    v = sel((lo_v > R(-INF_BOUND)) & (v < lo_v), lo_v, v);
    v = sel(vu, v, zero);
    L::cold_put(C_V, v);
    L::cold_put(C_LAMV, sel(vu, -fma_(p_v, v, q_v), zero));        // multiplier of the speed's box row
    L::cold_put(C_A20, fld(F_A20, 0.0));
    L::cold_put(C_BV, fld(F_B20, 0.0) * v);
    L::cold_put(C_BEQ2, fld(F_BEQ + 2, 0.0));
    // ---- the (e_y, e_psi, kappa) problem
    const R lo_e = max_(fld(F_LO + 0, -INFTY), R(-INFTY)), hi_e = min_(fld(F_HI + 0, INFTY), R(INFTY));
    const R lo_k = max_(fld(F_LO + 4, -INFTY), R(-INFTY)), hi_k = min_(fld(F_HI + 4, INFTY), R(INFTY));
    {
      // an EMPTY box makes the QP trivially infeasible (Solver::run has the same rule): reported at once, never solved
      R gap = max_(max_(sel(vx, lo_e - hi_e, zero), sel(vu, lo_k - hi_k, zero)), sel(vu, lo_v - hi_v, zero));
      gap = L::gmax(gap);
      empty = live & (gap > zero);
      solvable = live & !empty;
      L::cold_put(C_GAP, gap);
    }
    const R ds = fld(F_DS, 0.0), one = sel(vu, onec, zero);
    a[0] = one; a[1] = ds; a[2] = fld(F_A10, 0.0); a[3] = one;
    b[0] = ds;
    mI[0] = mI[1] = R(-1.0);
    P3[0] = fld(F_P + 0, 1.0); P3[1] = fld(F_P + 1, 1.0); P3[2] = fld(F_P + 4, 1.0);
    Q3[0] = fld(F_Q + 0, 0.0); Q3[1] = fld(F_Q + 1, 0.0); Q3[2] = fld(F_Q + 4, 0.0);
    R D3[3] = {onec, onec, onec}, G3[3] = {onec, onec, onec}, Eb[3] = {onec, onec, onec}, E2[2] = {onec, onec}, c3(1.0);
    // OSQP scale_data() on the reduced problem: Ruiz passes over the columns (e_y, e_psi, kappa) and the rows (2 dynamics
    // rows, 3 box rows), each with the cost normalisation
    const int passes = st.early_scaling > 0 && st.early_scaling < st.scaling ? st.early_scaling : st.scaling;
    const R inv_n_total(1.0 / double(3 * N + 2));
    for (int it = 0; it < passes; ++it) {
      R cn[3], rn[2], r_own[2];
      cn[0] = max_(max_(max_(abs_(P3[0]), abs_(mI[0])), max_(abs_(a[0]), abs_(a[2]))), abs_(G3[0]));
      cn[1] = max_(max_(max_(abs_(P3[1]), abs_(mI[1])), max_(abs_(a[1]), abs_(a[3]))), abs_(G3[1]));
      cn[2] = max_(max_(abs_(P3[2]), abs_(b[0])), abs_(G3[2]));
      r_own[0] = max_(abs_(a[0]), abs_(a[1]));
      r_own[1] = max_(max_(abs_(a[2]), abs_(a[3])), abs_(b[0]));
      R Dt[3], Et[2], Etb[3], Etd[2];
      MPMPC_UNROLL
      for (int i = 0; i < 2; ++i) {
        rn[i] = max_(abs_(mI[i]), L::up(r_own[i]));
        Et[i] = rsqrt_(S::limit(rn[i]));          // (reciprocal square roots and reciprocals, ~1 ulp: 12 - 27 instructions each less than 1 / sqrt, a / b)
        Etd[i] = L::down(Et[i]);
      }
      MPMPC_UNROLL
      for (int e = 0; e < 3; ++e) {
        Dt[e] = rsqrt_(S::limit(cn[e]));
        Etb[e] = rsqrt_(S::limit(abs_(G3[e])));
        P3[e] = (Dt[e] * P3[e]) * Dt[e];
        G3[e] = (Etb[e] * G3[e]) * Dt[e];
        Q3[e] = Dt[e] * Q3[e];
        D3[e] = D3[e] * Dt[e];
        Eb[e] = Eb[e] * Etb[e];
      }
      MPMPC_UNROLL
      for (int i = 0; i < 2; ++i) { mI[i] = (Et[i] * mI[i]) * Dt[i]; E2[i] = E2[i] * Et[i]; }
      a[0] = (Etd[0] * a[0]) * Dt[0]; a[1] = (Etd[0] * a[1]) * Dt[1];
      a[2] = (Etd[1] * a[2]) * Dt[0]; a[3] = (Etd[1] * a[3]) * Dt[1];
      b[0] = (Etd[1] * b[0]) * Dt[2];
      R s(0.0), mq(0.0);
      MPMPC_UNROLL
      for (int e = 0; e < 3; ++e) {
        s = s + sel(val[e], abs_(P3[e]), zero);
        mq = max_(mq, sel(val[e], abs_(Q3[e]), zero));
      }
      R ct = L::gsum(s) * inv_n_total;
      const R nq = S::limit(L::gmax(mq));
      ct = rcp_(S::limit(max_(ct, nq)));
      MPMPC_UNROLL
      for (int e = 0; e < 3; ++e) { P3[e] = P3[e] * ct; Q3[e] = Q3[e] * ct; }
      c3 = c3 * ct;
    }
    leq[0] = E2[0] * fld(F_BEQ + 0, 0.0);
    leq[1] = E2[1] * fld(F_BEQ + 1, 0.0);
    MPMPC_UNROLL
    for (int e = 0; e < 3; ++e) { L::cold_put(C_D + e, D3[e]); L::cold_put(C_G + e, G3[e]); }
    L::cold_put(C_E, E2[0]); L::cold_put(C_E + 1, E2[1]);
    L::cold_put(C_C, c3);
    // box in the scaled variable space:  g x in [lb, ub]  <=>  x in [lo_raw, hi_raw] / D
    {
      const R i0 = rcp_(D3[0]), i2 = rcp_(D3[2]);
      L::cold_put(K_LO0, lo_e * i0); L::cold_put(K_HI0, hi_e * i0); L::cold_put(K_LO2, lo_k * i2); L::cold_put(K_HI2, hi_k * i2);
    }
    // ---- start of the interior point: x = 0, no pin multipliers.  (The first reduced-native build started from OSQP's first
    // iterate of the reduced system - one factorisation + one KKT solve, like Solver::reduced_start: 0.25 interior-point
    // iterations fewer on config 2, none fewer on config 4, for the price of 0.4 iterations; measured without it: config 2
    // 23.8 -> 25.1 M solves/s, config 4 30.1 -> 31.6 M, B = 4 096 45.4 -> 53.6 M)
    MPMPC_UNROLL
    for (int e = 0; e < 3; ++e) {
      x3[e] = zero;
      L::cold_put(C_PI + e, zero);
    }
    L::fence();
  }

  // ================================================================================ layouts
  // 3 entries per lane (e_y, e_psi, kappa)  <->  the interior point's layout (split: lane k keeps (e_y, e_psi), lane
  // k + 32 keeps (kappa, -); `fill` = value of the entry the upper lanes do not have)
  MPMPC_HD void to_ip(const R v[3], R* o, double fill = 0.0) const {
    if constexpr (kSplit) {
      o[0] = sel(sU, L::from_lower(v[2]), v[0]);
      o[1] = sel(sU, R(fill), v[1]);
    } else {
      o[0] = v[0]; o[1] = v[1]; o[2] = v[2];
    }
  }
  MPMPC_HD void from_ip(const R* v, R o[3]) const {
    if constexpr (kSplit) {
      o[0] = v[0]; o[1] = v[1]; o[2] = L::from_upper(v[0]);
    } else {
      o[0] = v[0]; o[1] = v[1]; o[2] = v[2];
    }
  }
  MPMPC_HD void mask_to_ip(const Mk m[3], Mk* o) const {
    if constexpr (kSplit) {
      R v[3], w[2];
      MPMPC_UNROLL
      for (int e = 0; e < 3; ++e) v[e] = sel(m[e], R(1.0), R(0.0));
      to_ip(v, w);
      o[0] = w[0] > R(0.5); o[1] = w[1] > R(0.5);
    } else {
      o[0] = m[0]; o[1] = m[1]; o[2] = m[2];
    }
  }
  MPMPC_HD void mask_from_ip(const Mk* m, Mk o[3]) const {
    if constexpr (kSplit) {
      R w[2] = {sel(m[0], R(1.0), R(0.0)), sel(m[1], R(1.0), R(0.0))}, v[3];
      from_ip(w, v);
      MPMPC_UNROLL
      for (int e = 0; e < 3; ++e) o[e] = v[e] > R(0.5);
    } else {
      o[0] = m[0]; o[1] = m[1]; o[2] = m[2];
    }
  }

  // ================================================================================ interior point, 3 entries per lane
  // Solver::ipm<LAY_RED> (same iteration, same constants, same operation order) written for a small register footprint:
  // two wavefronts share a SIMD only if a wave stays within 256 registers, and the general routine keeps ~390 live
  // around its two KKT solves.  Here the loop invariants (box, cost, equality offsets) and the residuals of the iteration
  // wait in LDS and are re-read where they are used (~40 LDS reads and 5 writes per iteration, 2 % of its instructions);
  // across a factorisation or a KKT solve a lane holds the iterate, the slack reciprocals and the complementarity
  // targets - nothing else.  Entry 1 (e_psi) is never boxed (reducible()).
  // SOFT = true: phase 1 (Solver::ipm<LAY, true>: every boxed entry reads lo <= x + w <= hi with the cost 1/2 om w^2 and nothing
  // else in the cost; om w = zl - zu is never stored; the loop also ends at the first iterate whose multipliers pass the Farkas
  // test).  The weights om = g^2 - OSQP's metric of a violation, Solver::phase1 - are formed from the box-row scalings in the
  // slots C_G where they are used.  Same operations in the same order as the general routine; K_PP / K_QQ are not read then.
  // band_slot (SOFT, phase1_accept): cold slot of the violation beyond which the loop may leave at the first valid ray
  // (Solver::p1_band), -1: leave at any
  template <bool SOFT = false>
  MPMPC_HD Mk ipm3(const Box3& bx, typename S::template IpmT<LAY_RED>& s, const SolverParams& st, double tol, const Mk& run, int band_slot = -1) {
    const R reg(st.ipm_reg), ireg(st.inv_ipm_reg), one(1.0), zero(0.0);
    constexpr int JB[2] = {0, 2};                       // the boxed entries: e_y, kappa
    Mk active = run, conv = L::mfalse();
    // (number of bounds of the instance by ballot + bit count on the scalar unit; its reciprocal once: the complementarity
    //  measures below are products)
    R cnt(0.0);
    MPMPC_UNROLL
    for (int b = 0; b < 2; ++b) cnt = cnt + L::gcount(bx.Lm[JB[b]]) + L::gcount(bx.Um[JB[b]]);
    const R inb = rcp_(max_(cnt, one));
    I stall(0);
    R mu_min(1e300);
    auto lo_of = [&](int b) { return L::cold_get(b == 0 ? K_LO0 : K_LO2); };
    auto hi_of = [&](int b) { return L::cold_get(b == 0 ? K_HI0 : K_HI2); };
    // slack residuals of boxed entry b (cheap functions of the iterate and the box)
    // A side without a bound (infinite, pinned entry, lane without a stage) carries a zero multiplier and - in the products
    // that eliminate the slack steps - a zero in place of its slack reciprocal (isl / isu below): its multiplier step is then
    // exactly zero whatever its slack residual is, and that residual (finite: the clipped infinities are 1e30) needs no mask
    // except where it would enter a norm.  The iterates are bit for bit those of the fully masked form (Solver::ipm<LAY_RED>).
    auto iom_of = [&](int j) { const R gj = L::cold_get(C_G + j); return rcp_(gj * gj); };      // (phase 1 only)
    auto w_of = [&](int j) { return SOFT ? (s.zl[j] - s.zu[j]) * iom_of(j) : zero; };
    auto rl_of = [&](int b) { const int j = JB[b]; return SOFT ? s.x[j] + w_of(j) - lo_of(b) - s.sl[j] : s.x[j] - lo_of(b) - s.sl[j]; };
    auto ru_of = [&](int b) { const int j = JB[b]; return SOFT ? hi_of(b) - s.x[j] - w_of(j) - s.su[j] : hi_of(b) - s.x[j] - s.su[j]; };
    auto rpin_of = [&](int b) { const int j = JB[b]; return sel(bx.pin[j], s.x[j] - lo_of(b), zero); };
    auto rpin_raw = [&](int b) { const int j = JB[b]; return s.x[j] - lo_of(b); };       // (where a select on pin follows anyway)
    for (int it = 0; it <= st.ipm_max_iter; ++it) {
      R mu;
      {
        // ---- residuals -> LDS
        L::fence();
        R At[3], rp[2], rd[3];
        this->template AeqT_mul_t<LAY_RED>(s.nu, At);
        this->template Aeq_mul_t<LAY_RED>(s.x, rp);
        R res(0.0), msum(0.0);
        MPMPC_UNROLL
        for (int i = 0; i < 2; ++i) {
          rp[i] = rp[i] - L::cold_get(K_LEQ + i);
          if constexpr (SOFT) rp[i] = fma_(R(-P1_EQ_SOFT), s.nu[i], rp[i]);          // (soft dynamics rows: mpmpc_core.hpp, P1_EQ_SOFT)
          res = max_(res, sel(vx, abs_(rp[i]), zero));
        }
        MPMPC_UNROLL
        for (int j = 0; j < 3; ++j) {
          if constexpr (SOFT) rd[j] = At[j] - s.zl[j] + s.zu[j] + s.pi[j];
          else rd[j] = fma_(L::cold_get(K_PP + j), s.x[j], L::cold_get(K_QQ + j)) + At[j] - s.zl[j] + s.zu[j] + s.pi[j];
        }
        res = max_(res, sel(val[1], abs_(rd[1]), zero));
        MPMPC_UNROLL
        for (int b = 0; b < 2; ++b) {
          const int j = JB[b];
          res = max_(res, sel(val[j], max_(max_(abs_(rd[j]), abs_(rpin_of(b))), max_(sel(bx.Lm[j], abs_(rl_of(b)), zero), sel(bx.Um[j], abs_(ru_of(b)), zero))), zero));
          // (a side without a bound keeps a finite slack and an exactly zero multiplier - see above: its product is an exact zero)
          msum = msum + s.sl[j] * s.zl[j] + s.su[j] * s.zu[j];
        }
        MPMPC_UNROLL
        for (int j = 0; j < 3; ++j) L::cold_put(K_RD + j, rd[j]);
        L::cold_put(K_RP, rp[0]); L::cold_put(K_RP + 1, rp[1]);
        res = L::gmax(res);
        mu = L::gsum(msum) * inb;
        Mk ok = (res < R(tol > 1e-11 ? tol : 1e-11)) & (mu < R(tol));
        if constexpr (SOFT) {
          S::p1_converged = selb(active, ok, S::p1_converged);
          // Farkas test on the multipliers y = (nu, zu - zl + pi) in the scaled problem: A'y is the dual residual rd itself
          R ny(0.0), na(0.0), sup(0.0);
          MPMPC_UNROLL
          for (int i = 0; i < 2; ++i) { ny = max_(ny, sel(vx, abs_(s.nu[i]), zero)); sup = sup + sel(vx, L::cold_get(K_LEQ + i) * s.nu[i], zero); }
          na = max_(na, sel(val[0], abs_(rd[0]), zero));
          na = max_(na, sel(val[1], abs_(rd[1]), zero));
          na = max_(na, sel(val[2], abs_(rd[2]), zero));
          MPMPC_UNROLL
          for (int b = 0; b < 2; ++b) {
            const int j = JB[b];
            const R lam = s.zu[j] - s.zl[j] + s.pi[j];
            ny = max_(ny, sel(val[j], abs_(lam), zero));
            sup = sup + sel(val[j] & (lam > zero) & (bx.Um[j] | bx.pin[j]), sel(bx.pin[j], lo_of(b), hi_of(b)) * lam, zero) +
                  sel(val[j] & (lam < zero) & (bx.Lm[j] | bx.pin[j]), lo_of(b) * lam, zero);
          }
          ny = L::gmax(ny); na = L::gmax(na); sup = L::gsum(sup);
          const R thr = R(st.phase1_eps) * ny;
          Mk ray = (ny > R(st.phase1_eps)) & (na < thr) & (sup < -thr);
          if (band_slot >= 0) {
            // (Solver::ipm: inside the band the iteration runs to its converged optimum - its violation decides "marginal")
            R wv(0.0);
            MPMPC_UNROLL
            for (int b = 0; b < 2; ++b) {
              const int j = JB[b];
              wv = max_(wv, sel(val[j], abs_(w_of(j)) * L::cold_get(C_D + j), zero));
            }
            ray = ray & (L::gmax(wv) > L::cold_get(band_slot));
          }
          ok = ok | ray;
        }
        conv = conv | (active & ok);
        active = active & !ok;
        if (it == st.ipm_max_iter || !L::wany(active)) break;
        if constexpr (!SOFT) {
          active = active & !(mu > R(st.ipm_diverged) * mu_min) & !((mu < R(tol * 1e-3)) & (res > R(1e-5)));
          mu_min = min_(mu_min, mu);
          if (!L::wany(active)) break;
        }
      }
      ipm_iters = seli(active, ipm_iters + I(1), ipm_iters);
      // ---- factor
      R isl[2], isu[2], rcl[2], rcu[2];
      [[maybe_unused]] R kap[2];
      {
        L::fence();
        R h[3];
        h[1] = SOFT ? rcp_(reg) : rcp_(L::cold_get(K_PP + 1) + reg);
        MPMPC_UNROLL
        for (int b = 0; b < 2; ++b) {
          const int j = JB[b];
          const R il = rcp_(s.sl[j]), iu = rcp_(s.su[j]);
          if constexpr (SOFT) {
            // k = om / (om + th), th = zl / sl + zu / su; the diagonal carries k th (= om (1 - k))
            const R th = sel(bx.Lm[j], s.zl[j] * il, zero) + sel(bx.Um[j], s.zu[j] * iu, zero);
            const R kp = rcp_(fma_(th, iom_of(j), one));
            h[j] = rcp_(fma_(kp, th, reg) + sel(bx.pin[j], ireg, zero));
          } else {
            h[j] = rcp_(L::cold_get(K_PP + j) + reg + sel(bx.Lm[j], s.zl[j] * il, zero) + sel(bx.Um[j], s.zu[j] * iu, zero) +
                        sel(bx.pin[j], ireg, zero));
          }
        }
        this->template factor_t<LAY_RED>(h, SOFT ? reg + R(P1_EQ_SOFT) : reg);
        // (the slack reciprocals and complementarity products are formed AFTER the factorisation: four reciprocals twice are
        //  cheaper than eight registers across the factorisation's levels)
        L::fence();
        MPMPC_UNROLL
        for (int b = 0; b < 2; ++b) {
          const int j = JB[b];
          isl[b] = sel(bx.Lm[j], rcp_(s.sl[j]), zero); isu[b] = sel(bx.Um[j], rcp_(s.su[j]), zero);
          rcl[b] = s.sl[j] * s.zl[j]; rcu[b] = s.su[j] * s.zu[j];
          // (k again, like the reciprocals: two registers less across the factorisation)
          if constexpr (SOFT) kap[b] = rcp_(fma_(s.zl[j] * isl[b] + s.zu[j] * isu[b], iom_of(j), one));
        }
      }
      R alpha_aff(1.0);
      for (int pass = 0; pass < 2; ++pass) {
        R dx[3], dnu[2];
        [[maybe_unused]] R cul[2];          // phase 1: cu - cl of the boxed entries
        {
          L::fence();
          R rhs[3], nreq[2];
          rhs[1] = -L::cold_get(K_RD + 1);
          MPMPC_UNROLL
          for (int b = 0; b < 2; ++b) {
            const int j = JB[b];
            if constexpr (SOFT) {
              cul[b] = fma_(s.zu[j], ru_of(b), rcu[b]) * isu[b] - fma_(s.zl[j], rl_of(b), rcl[b]) * isl[b];
              rhs[j] = fma_(kap[b], cul[b], -L::cold_get(K_RD + j)) - sel(bx.pin[j], rpin_raw(b) * ireg, zero);
            } else {
              rhs[j] = -L::cold_get(K_RD + j) - fma_(s.zl[j], rl_of(b), rcl[b]) * isl[b] + fma_(s.zu[j], ru_of(b), rcu[b]) * isu[b] -
                       sel(bx.pin[j], rpin_raw(b) * ireg, zero);
            }
          }
          nreq[0] = -L::cold_get(K_RP); nreq[1] = -L::cold_get(K_RP + 1);
          this->template kkt_solve_t<LAY_RED>(rhs, nreq, dx, dnu);
          if (SOFT && pass == 1) {
            // one refinement step against the UN-regularised Newton matrix (Solver::ipm: a clean ray is what phase 1 is asked
            // for).  The step waits in the residuals' slots - free until the next iteration - while the second solve runs.
            R Ad[2], Atd[3], r1[3], r2[2];
            this->template Aeq_mul_t<LAY_RED>(dx, Ad);
            this->template AeqT_mul_t<LAY_RED>(dnu, Atd);
            r1[1] = rhs[1] - Atd[1];                                  // (H - reg of the free entry is zero)
            MPMPC_UNROLL
            for (int b = 0; b < 2; ++b) {
              const int j = JB[b];
              // (k th again: th = zl / sl + zu / su from the masked reciprocals)
              r1[j] = rhs[j] - fma_(kap[b] * (s.zl[j] * isl[b] + s.zu[j] * isu[b]) + sel(bx.pin[j], ireg, zero), dx[j], Atd[j]);
            }
            r2[0] = fma_(R(P1_EQ_SOFT), dnu[0], nreq[0] - Ad[0]); r2[1] = fma_(R(P1_EQ_SOFT), dnu[1], nreq[1] - Ad[1]);
            L::fence();
            MPMPC_UNROLL
            for (int j = 0; j < 3; ++j) L::cold_put(K_RD + j, dx[j]);
            L::cold_put(K_RP, dnu[0]); L::cold_put(K_RP + 1, dnu[1]);
            L::fence();
            R ddx[3], ddn[2];
            this->template kkt_solve_t<LAY_RED>(r1, r2, ddx, ddn);
            L::fence();
            MPMPC_UNROLL
            for (int j = 0; j < 3; ++j) dx[j] = L::cold_get(K_RD + j) + sel(val[j], ddx[j], zero);
            dnu[0] = L::cold_get(K_RP) + sel(vx, ddn[0], zero);
            dnu[1] = L::cold_get(K_RP + 1) + sel(vx, ddn[1], zero);
          }
        }
        L::fence();
        // largest step that keeps slacks and multipliers positive: 1 / max(-ds/s, -dz/z)
        R dsl[2], dsu[2], dzl[2], dzu[2], dpi[2];
        R blk(0.0);
        MPMPC_UNROLL
        for (int b = 0; b < 2; ++b) {
          const int j = JB[b];
          const R ex = SOFT ? kap[b] * fma_(cul[b], iom_of(j), dx[j]) : dx[j];          // step of x + w:  k (dx + (cu - cl) / om)
          dsl[b] = ex + rl_of(b);
          dsu[b] = -ex + ru_of(b);
          dzl[b] = -fma_(s.zl[j], dsl[b], rcl[b]) * isl[b];
          dzu[b] = -fma_(s.zu[j], dsu[b], rcu[b]) * isu[b];
          dpi[b] = sel(bx.pin[j], (rpin_raw(b) + dx[j]) * ireg, zero);
          blk = max_(blk, max_(-dsl[b] * isl[b], -dsu[b] * isu[b]));
          // (the ratios -dz / z only size the step, which keeps 0.5 % from the boundary anyway: the reciprocal's seed will do)
          // (a side without a bound: dz = 0 over z = 0 is 0 x inf = NaN, which max_ - v_max_f64, fmax - passes over)
          blk = max_(blk, max_(-dzl[b] * rcp_fast_(s.zl[j]), -dzu[b] * rcp_fast_(s.zu[j])));
        }
        blk = L::gmax(blk);
        const R ratio = sel(blk > zero, rcp_(blk), R(1e300));
        if (pass == 0) {
          alpha_aff = min_(one, ratio);
          R ms(0.0);
          MPMPC_UNROLL
          for (int b = 0; b < 2; ++b) {
            const int j = JB[b];
            ms = ms + fma_(alpha_aff, dsl[b], s.sl[j]) * fma_(alpha_aff, dzl[b], s.zl[j]) +
                 fma_(alpha_aff, dsu[b], s.su[j]) * fma_(alpha_aff, dzu[b], s.zu[j]);
          }
          const R mu_aff = L::gsum(ms) * inb;
          R sg = mu_aff * rcp_(max_(mu, R(1e-300)));
          sg = sg * sg * sg;
          const R sgmu = sg * mu;
          MPMPC_UNROLL
          for (int b = 0; b < 2; ++b) {
            const int j = JB[b];
            rcl[b] = fma_(dsl[b], dzl[b], fma_(s.sl[j], s.zl[j], -sgmu));
            rcu[b] = fma_(dsu[b], dzu[b], fma_(s.su[j], s.zu[j], -sgmu));
          }
        } else {
          const R al = min_(one, R(0.995) * ratio);
          stall = seli(active & (al < R(1e-6)), stall + I(1), I(0));
          s.x[1] = sel(active, fma_(al, dx[1], s.x[1]), s.x[1]);
          MPMPC_UNROLL
          for (int b = 0; b < 2; ++b) {
            const int j = JB[b];
            s.x[j] = sel(active, fma_(al, dx[j], s.x[j]), s.x[j]);
            if constexpr (!SOFT) {
              s.tL[j] = selb(active, bx.Lm[j] & (dsl[b] * s.zl[j] < dzl[b] * s.sl[j]), s.tL[j]);
              s.tU[j] = selb(active, bx.Um[j] & (dsu[b] * s.zu[j] < dzu[b] * s.su[j]), s.tU[j]);
            }
            s.sl[j] = sel(active, fma_(al, dsl[b], s.sl[j]), s.sl[j]);
            s.su[j] = sel(active, fma_(al, dsu[b], s.su[j]), s.su[j]);
            s.zl[j] = sel(active, fma_(al, dzl[b], s.zl[j]), s.zl[j]);
            s.zu[j] = sel(active, fma_(al, dzu[b], s.zu[j]), s.zu[j]);
            s.pi[j] = sel(active, fma_(al, dpi[b], s.pi[j]), s.pi[j]);
          }
          s.nu[0] = sel(active, fma_(al, dnu[0], s.nu[0]), s.nu[0]);
          s.nu[1] = sel(active, fma_(al, dnu[1], s.nu[1]), s.nu[1]);
          active = active & (stall < 3);
        }
      }
    }
    return conv;
  }

  // ================================================================================ certificate (reduced problem, unscaled)
  MPMPC_HD Mk certificate3(const R* pp, const R* qq, const R xs[3], const R nus[2], const R lam[3], double tol, R& prim, R& stat) const {
    const R zero(0.0);
    R Ax[2], At[3];
    this->template Aeq_mul_t<LAY_RED>(xs, Ax);
    this->template AeqT_mul_t<LAY_RED>(nus, At);
    const R cinv = rcp_(L::cold_get(C_C));
    R pv(0.0), sv(0.0), cv(0.0);
    MPMPC_UNROLL
    for (int i = 0; i < 2; ++i) pv = max_(pv, sel(vx, abs_((Ax[i] - leq[i]) * rcp_(L::cold_get(C_E + i))), zero));
    Mk bad = L::mfalse();
    MPMPC_UNROLL
    for (int e = 0; e < 3; ++e) {
      const R De = L::cold_get(C_D + e), iDe = rcp_(De);
      // (the box in scaled units; distances go back to the unscaled problem through D)
      const R lo0 = e == 0 ? L::cold_get(K_LO0) : (e == 2 ? L::cold_get(K_LO2) : R(-INFTY));
      const R hi0 = e == 0 ? L::cold_get(K_HI0) : (e == 2 ? L::cold_get(K_HI2) : R(INFTY));
      const Mk fu = hi0 < R(BOX_INF), fl = lo0 > R(-BOX_INF);
      const R dlo = sel(fl, De * (xs[e] - lo0), R(INFTY)), dhi = sel(fu, De * (hi0 - xs[e]), R(INFTY));   // unscaled distances to the bounds
      pv = max_(pv, sel(val[e], max_(max_(-dlo, -dhi), zero), zero));
      const R rd = fma_(pp[e], xs[e], qq[e]) + At[e] + lam[e];
      sv = max_(sv, sel(val[e], abs_(rd * iDe) * cinv, zero));
      const R yu = (lam[e] * iDe) * cinv;                 // multiplier of the unscaled box row
      const R cu = sel(fu, max_(yu, zero) * abs_(dhi), sel(yu > zero, R(1e300), zero));
      const R cl = sel(fl, max_(-yu, zero) * abs_(dlo), sel(yu < zero, R(1e300), zero));
      cv = max_(cv, sel(val[e], max_(cu, cl), zero));
      bad = bad | (val[e] & !((abs_(xs[e]) < R(1e300)) & (abs_(lam[e]) < R(1e300))));     // a NaN must never pass
    }
    MPMPC_UNROLL
    for (int i = 0; i < 2; ++i) bad = bad | (vx & !(abs_(nus[i]) < R(1e300)));
    bad = L::gany(bad);
    prim = L::gmax(pv);
    stat = L::gmax(sv);
    cv = L::gmax(cv);
    return (prim <= R(tol)) & (stat <= R(tol)) & (cv <= R(tol)) & !bad;
  }

  // The active-set rounds are the register peak of the kernel (their factorisation on top of the caller's state); the boxed
  // entries' slacks and multipliers of the interior point wait in LDS meanwhile.
  MPMPC_HD void park_ip(const IpmI& s) {
    L::fence();
    MPMPC_UNROLL
    for (int b = 0; b < 2; ++b) {
      const int j = kSplit ? b : (b == 0 ? 0 : 2);
      L::cold_put(park_slot(4 * b + 0), s.sl[j]); L::cold_put(park_slot(4 * b + 1), s.su[j]);
      L::cold_put(park_slot(4 * b + 2), s.zl[j]); L::cold_put(park_slot(4 * b + 3), s.zu[j]);
      L::cold_put(K_QQ + b, s.pi[j]);
    }
    // (the packed interior point's copies of cost and offsets are in registers again by now: their slots take the iterate)
    MPMPC_UNROLL
    for (int j = 0; j < EI; ++j) L::cold_put(K_PP + j, s.x[j]);
    L::cold_put(K_LEQ, s.nu[0]); L::cold_put(K_LEQ + 1, s.nu[1]);
    L::fence();
  }
  MPMPC_HD void unpark_ip(IpmI& s) {
    L::fence();
    MPMPC_UNROLL
    for (int b = 0; b < 2; ++b) {
      const int j = kSplit ? b : (b == 0 ? 0 : 2);
      s.sl[j] = L::cold_get(park_slot(4 * b + 0)); s.su[j] = L::cold_get(park_slot(4 * b + 1));
      s.zl[j] = L::cold_get(park_slot(4 * b + 2)); s.zu[j] = L::cold_get(park_slot(4 * b + 3));
      s.pi[j] = L::cold_get(K_QQ + b);
    }
    MPMPC_UNROLL
    for (int j = 0; j < EI; ++j) s.x[j] = L::cold_get(K_PP + j);
    s.nu[0] = L::cold_get(K_LEQ); s.nu[1] = L::cold_get(K_LEQ + 1);
    if constexpr (!kSplit) {
      s.sl[1] = s.su[1] = R(1.0); s.zl[1] = s.zu[1] = s.pi[1] = R(0.0);
      L::fence();
      MPMPC_UNROLL
      for (int e = 0; e < 3; ++e) { L::cold_put(K_PP + e, P3[e]); L::cold_put(K_QQ + e, Q3[e]); }
      L::cold_put(K_LEQ, leq[0]); L::cold_put(K_LEQ + 1, leq[1]);
      L::fence();
    }
  }

  // ================================================================================ the solve
  // the certified point goes to cold storage (the slots of the start's G and pin multipliers, which are done with); with
  // `merge` a packed wave's later commit does not disturb what its partner instance has committed before
  MPMPC_HD void commit(const Mk& good, bool merge, const R xa[3], const R na[2], const R la[3], const R& prim, const R& stat) {
    bool skip = false;
    if constexpr (kLean) skip = !L::wany(good);          // (nothing certified: the slots still hold what a further attempt unparks)
    if (skip) {
    } else if (!merge) {
      MPMPC_UNROLL
      for (int e = 0; e < 3; ++e) { L::cold_put(C_XS + e, xa[e]); L::cold_put(C_LAM + e, la[e]); }
      L::cold_put(C_NUS, na[0]); L::cold_put(C_NUS + 1, na[1]);
    } else {
      MPMPC_UNROLL
      for (int e = 0; e < 3; ++e) {
        L::cold_put(C_XS + e, sel(good, xa[e], L::cold_get(C_XS + e)));
        L::cold_put(C_LAM + e, sel(good, la[e], L::cold_get(C_LAM + e)));
      }
      L::cold_put(C_NUS, sel(good, na[0], L::cold_get(C_NUS))); L::cold_put(C_NUS + 1, sel(good, na[1], L::cold_get(C_NUS + 1)));
    }
    L::fence();
    pri_res = sel(good, prim, pri_res);
    dua_res = sel(good, stat, dua_res);
    status = seli(good, I(MPMPC_SOLVED), status);
    polished = seli(good, I(1), polished);
  }
  // active set of a certified point in the closed loop's format (Solver::pack_active: bit j lower, bit 5 + j upper over
  // (e_y, e_psi, t, v, kappa), bit 30 = valid); the speed's own activity from the sign of its multiplier
  MPMPC_HD I pack_active3(const Mk aL[3], const Mk aU[3]) const {
    const R lam_v = L::cold_get(C_LAMV), zero(0.0);
    I v(1 << 30);
    v = v + seli(aL[0], I(1 << 0), I(0)) + seli(aU[0], I(32 << 0), I(0)) + seli(aL[1], I(1 << 1), I(0)) + seli(aU[1], I(32 << 1), I(0)) +
        seli(aL[2], I(1 << 4), I(0)) + seli(aU[2], I(32 << 4), I(0));
    v = v + seli(vu & (lam_v < zero), I(1 << 3), I(0)) + seli(vu & (lam_v > zero), I(32 << 3), I(0));
    return v;
  }

  // box rows in the scaled variable space:  g x in [lb, ub]  <=>  x in [lo, hi] = [lo_raw, hi_raw] / D  (slots K_LO0 .. K_HI2)
  MPMPC_HD void make_box3(Box3& b3) const {
    const R one(1.0);
    const R lo_s[3] = {L::cold_get(K_LO0), R(-INFTY), L::cold_get(K_LO2)}, hi_s[3] = {L::cold_get(K_HI0), R(INFTY), L::cold_get(K_HI2)};
    MPMPC_UNROLL
    for (int e = 0; e < 3; ++e) {
      const Mk fl = lo_s[e] > R(-BOX_INF), fu = hi_s[e] < R(BOX_INF);
      const Mk pn = fl & fu & ((hi_s[e] - lo_s[e]) <= R(1e-12) * max_(one, abs_(lo_s[e])));
      b3.lo[e] = lo_s[e];
      b3.hi[e] = hi_s[e];
      b3.pin[e] = pn & val[e];
      b3.Lm[e] = fl & !pn & val[e];
      b3.Um[e] = fu & !pn & val[e];
    }
  }

  // The centred start of the interior point at x3 and the attempts from it (interior point, active-set rounds, certificate;
  // see RN_ATTEMPTS).  cap: interior-point iterations an attempt may take; rd0_more: a dual residual of the start the three
  // entries do not show (the tail solver's speed entry), or nullptr; g3: the box-row scalings if the caller holds them in
  // registers (the tail solver: its slots C_G / C_PI already carry a point by then; no pin multipliers in that case), or
  // nullptr.  What is certified is committed; returns what is not.
  template <bool WARM = false>
  MPMPC_HD Mk attempts(Box3& b3, const SolverParams& st, int cap, Mk todo, bool committed, const R* rd0_more = nullptr,
                      const R* g3 = nullptr) {
    const R zero(0.0), one(1.0);
    // ---- centred start of the interior point (Solver::polish, early attempt): slacks max(distance to the bound,
    // ipm_start_slack in row space), multipliers mu0 / slack, no equality multipliers
    R sl[3], su[3], zl[3], zu[3], pi[3];
    {
      const R ths(st.ipm_start_slack);
      R mu0(st.ipm_start_mu);
      if (st.ipm_start_dual > 0.0) {
        R rd0(0.0);
        MPMPC_UNROLL
        for (int e = 0; e < 3; ++e) rd0 = max_(rd0, sel(val[e], abs_(fma_(P3[e], x3[e], Q3[e])), zero));
        R rd0g = L::gmax(rd0);
        if (rd0_more) rd0g = max_(rd0g, *rd0_more);
        mu0 = max_(mu0, (R(st.ipm_start_dual) * ths) * rd0g);
      }
      MPMPC_UNROLL
      for (int e = 0; e < 3; ++e) {
        const R fl = ths * rcp_(g3 ? g3[e] : L::cold_get(C_G + e));
        sl[e] = sel(b3.Lm[e], max_(x3[e] - b3.lo[e], fl), one);
        su[e] = sel(b3.Um[e], max_(b3.hi[e] - x3[e], fl), one);
        zl[e] = sel(b3.Lm[e], mu0 * rcp_(sl[e]), zero);
        zu[e] = sel(b3.Um[e], mu0 * rcp_(su[e]), zero);
        pi[e] = g3 ? zero : L::cold_get(C_PI + e);
      }
    }
    // ---- the interior point's own layout
    BoxI bi;
    IpmI si;
    R pp[EI], qq[EI];
    Mk vm[EI];
    if constexpr (kSplit) {
      bU[0] = sel(sU, L::from_lower(b[0]), zero);
      bU[1] = zero;
      vm[0] = val3[0]; vm[1] = val3[2];
    } else {
      vm[0] = val[0]; vm[1] = val[1]; vm[2] = val[2];
    }
    to_ip(b3.lo, bi.lo); to_ip(b3.hi, bi.hi); to_ip(P3, pp, 1.0); to_ip(Q3, qq);
    mask_to_ip(b3.Lm, bi.Lm); mask_to_ip(b3.Um, bi.Um); mask_to_ip(b3.pin, bi.pin);
    to_ip(x3, si.x); to_ip(sl, si.sl, 1.0); to_ip(su, si.su, 1.0); to_ip(zl, si.zl); to_ip(zu, si.zu); to_ip(pi, si.pi);
    si.nu[0] = si.nu[1] = zero;
    MPMPC_UNROLL
    for (int e = 0; e < EI; ++e) {
      si.tL[e] = bi.Lm[e] & (si.zl[e] > si.sl[e]);
      si.tU[e] = bi.Um[e] & (si.zu[e] > si.su[e]);
    }
    if constexpr (!kSplit) {
      MPMPC_UNROLL
      for (int e = 0; e < 3; ++e) { L::cold_put(K_PP + e, P3[e]); L::cold_put(K_QQ + e, Q3[e]); }
      L::cold_put(K_LEQ, leq[0]); L::cold_put(K_LEQ + 1, leq[1]);
      L::fence();
    }
    // A launch ends with its slowest wave, and a packed wave with its slower instance: feasible instances of this problem
    // family converge in 5 - 11 iterations, so the attempt gives up after RN_IPM_CAP - what is still running by then
    // (marginally infeasible instances: they would use all ipm_max_iter iterations) belongs to the tail launch anyway.
    SolverParams sc = st;
    sc.ipm_max_iter = st.ipm_max_iter < cap ? st.ipm_max_iter : cap;
    double tol = st.native_ipm_tol;
    for (int attempt = 0; attempt < RN_ATTEMPTS; ++attempt) {
      MPMPC_TICK_BEGIN(4);
      Mk conv;
      if constexpr (kSplit) {
        conv = this->template ipm<LAY_IP>(bi, si, pp, qq, vm, sc, tol, todo);
      } else {
        conv = ipm3(bi, si, sc, tol, todo);
        // the packed interior point read its invariants from LDS; what follows takes them from there as well, so that
        // no copy of them had to stay in registers across the loop
        L::fence();
        MPMPC_UNROLL
        for (int e = 0; e < 3; ++e) { P3[e] = L::cold_get(K_PP + e); Q3[e] = L::cold_get(K_QQ + e); }
        leq[0] = L::cold_get(K_LEQ); leq[1] = L::cold_get(K_LEQ + 1);
        b3.lo[0] = bi.lo[0] = L::cold_get(K_LO0); b3.hi[0] = bi.hi[0] = L::cold_get(K_HI0);
        b3.lo[2] = bi.lo[2] = L::cold_get(K_LO2); b3.hi[2] = bi.hi[2] = L::cold_get(K_HI2);
      }
      MPMPC_TICK_END(4);
      // active-set guess: the indicators of the interior point's last step
      Mk gL[EI], gU[EI], aL[3], aU[3];
      MPMPC_UNROLL
      for (int e = 0; e < EI; ++e) { gL[e] = bi.Lm[e] & si.tL[e]; gU[e] = bi.Um[e] & si.tU[e]; }
      mask_from_ip(gL, aL); mask_from_ip(gU, aU);
      MPMPC_UNROLL
      for (int e = 0; e < 3; ++e) { aL[e] = b3.Lm[e] & aL[e]; aU[e] = b3.Um[e] & aU[e] & !aL[e]; }
      // (the rounds overwrite the point on every lane they run on and nothing reads it elsewhere: no copy of the interior
      //  point's iterate has to live through them)
      R xa[3] = {zero, zero, zero}, la[3] = {zero, zero, zero}, na[2] = {zero, zero};
      park_ip(si);
      MPMPC_TICK_BEGIN(5);
      const double frac = attempt == 0 ? st.as_add_fraction : (st.as_add_fraction > 0.5 ? st.as_add_fraction : 0.5);
      const Mk okm = this->template active_set<LAY_RED>(b3, P3, Q3, val, aL, aU, xa, na, la, st, todo & conv, frac);
      MPMPC_TICK_END(5);
      R prim, stat;
      MPMPC_TICK_BEGIN(6);
      const Mk cert = certificate3(P3, Q3, xa, na, la, st.cert_tol, prim, stat);
      MPMPC_TICK_END(6);
      const Mk good = todo & conv & okm & cert;
      commit(good, committed, xa, na, la, prim, stat);
      committed = true;
      if constexpr (WARM) this->act_bits = seli(good, pack_active3(aL, aU), this->act_bits);
      todo = todo & conv & !good;          // a diverged interior-point run is not retried
      if (!L::wany(todo)) break;
      unpark_ip(si);
      if constexpr (kSplit) {          // (re-formed rather than carried through the rounds)
        to_ip(b3.lo, bi.lo); to_ip(b3.hi, bi.hi); to_ip(P3, pp, 1.0); to_ip(Q3, qq);
      }
      tol *= RN_RETRY;
    }
    return todo;
  }

  // WARM (closed loop): `guess` = active set of the previous step's certified plan, already shifted to this step's stages
  // (bit 30 set where there is one).  One or two active-set rounds from it usually reproduce the optimum
  // (Solver::warm_polish); what they cannot certify takes the normal path.
  template <bool WARM = false>
  MPMPC_HD void run(const R* fields, int B, const I& inst, const I& k, int N_, const SolverParams& st, const I& guess = I(0)) {
    MPMPC_TICK_BEGIN(0);
    setup(fields, B, inst, k, N_, st);
    MPMPC_TICK_END(0);
    const R zero(0.0);
    status = I(MPMPC_UNSOLVED);
    iters = I(1);
    ipm_iters = I(0);
    polished = I(0);
    pri_res = dua_res = zero;
    this->act_bits = I(0);
    Mk todo = solvable;
    if (L::wany(todo)) {
      bool committed = false;
      Box3 b3;
      make_box3(b3);
      if constexpr (WARM) {
        const Mk warm = L::gany(todo & bit_(guess, 30));
        if (L::wany(warm)) {
          Mk aL[3], aU[3];
          constexpr int J5[3] = {0, 1, 4};
          MPMPC_UNROLL
          for (int e = 0; e < 3; ++e) {
            aL[e] = b3.Lm[e] & bit_(guess, J5[e]);
            aU[e] = b3.Um[e] & bit_(guess, 5 + J5[e]) & !aL[e];
          }
          R xa[3] = {zero, zero, zero}, la[3] = {zero, zero, zero}, na[2] = {zero, zero};
          SolverParams sw = st;
          sw.as_rounds = st.as_rounds < 2 ? st.as_rounds : 2;      // a guess that needs more is not worth more than the normal path
          const Mk okm = this->template active_set<LAY_RED>(b3, P3, Q3, val, aL, aU, xa, na, la, sw, warm, st.as_add_fraction);
          R prim, stat;
          const Mk cert = certificate3(P3, Q3, xa, na, la, st.cert_tol, prim, stat);
          const Mk good = warm & okm & cert;
          // (merge: the lanes that missed their guess still need the start's G / pin multipliers, which share the slots)
          commit(good, true, xa, na, la, prim, stat);
          committed = true;
          // (bit 29: certified from the guess - "iters = 0" rides in a register that lives through the solve anyway)
          this->act_bits = seli(good, pack_active3(aL, aU) + I(1 << 29), this->act_bits);
          todo = todo & !good;
        }
      }
      if (L::wany(todo)) attempts<WARM>(b3, st, RN_IPM_CAP, todo, committed);
    }
    {
      const Mk warm_hit = bit_(this->act_bits, 29);
      iters = seli(warm_hit, I(0), I(1));
      this->act_bits = seli(warm_hit, this->act_bits - (1 << 29), this->act_bits);
    }
    // empty box: infeasible, zero ray, the width of the gap in resid[0]
    status = seli(empty, I(MPMPC_PRIMAL_INFEASIBLE), status);
    pri_res = sel(empty, L::cold_get(C_GAP), pri_res);
  }

  // ================================================================================ output
  // z in the reference's ordering, u0 = (v_0, delta_0), multipliers in the reference's row order; the separated parts
  // (v, its multiplier, the roll-forward of t) are put together here, in the unscaled problem
  // ANY: the instances of the wave are not consecutive (the packed tail kernel): rows go out one instance at a time
  template <bool ANY = false>
  MPMPC_HD void store(const I& inst, const I& k, double wheelbase, double* z, double* u0, int* st_out, int* it_out,
                      double* resid, double* y, int* act = nullptr, int ld = 0, const Mk* point = nullptr, const Mk* no_lamv = nullptr) const {
    if (act) L::storei(act, inst * ld + k, vx, this->act_bits);
    const int n = 5 * N + 3, m = 8 * N + 6;
    const R zero(0.0);
    // (point / no_lamv: the tail solver's verdicts - a least-violation point with its ray, a plan over relaxed boxes - also
    //  have a point in the slots; the ray carries no speed entry)
    const Mk ok = point ? *point : live & (status == MPMPC_SOLVED);
    const R cinv = rcp_(L::cold_get(C_C));
    R D3[3];
    MPMPC_UNROLL
    for (int e = 0; e < 3; ++e) D3[e] = L::cold_get(C_D + e);
    // the certified point (cold storage; zeros for an instance that has none: tail or empty box)
    R xs[3], lam3[3], nu2[2];
    MPMPC_UNROLL
    for (int e = 0; e < 3; ++e) { xs[e] = sel(ok, L::cold_get(C_XS + e), zero); lam3[e] = sel(ok, L::cold_get(C_LAM + e), zero); }
    nu2[0] = sel(ok, L::cold_get(C_NUS), zero); nu2[1] = sel(ok, L::cold_get(C_NUS + 1), zero);
    const R e_y = D3[0] * xs[0], e_psi = D3[1] * xs[1], kap = D3[2] * xs[2];
    const R v = sel(ok, L::cold_get(C_V), zero), lam_v = sel(no_lamv ? ok & !*no_lamv : ok, L::cold_get(C_LAMV), zero);
    // t: row 2 of equality block k is  -t_k + a20 e_y_{k-1} + t_{k-1} + b20 v_{k-1} = beq2_k  (block 0: -t_0 = -x0[2]),
    // a running sum along the stages
    const R drive = fma_(L::cold_get(C_A20), e_y, L::cold_get(C_BV)), beq2 = L::cold_get(C_BEQ2);
    const R t0 = -beq2;
    R t = t0;
    // (t_k = t_{k-1} + drive_{k-1} - beq2_k: an inclusive prefix sum along the stages, L::gscan - 5 / 6 shifted adds instead of N
    //  dependent steps)
    t = L::gscan(sel(first, t0, sel(vx, L::up(drive) - beq2, R(0.0))));
    t = sel(ok & vx, t, zero);
    const R E2[2] = {L::cold_get(C_E), L::cold_get(C_E + 1)};
    L::fence();
    if (z) {
      auto fill = [&](auto put) {
        put(k * 3, vx, e_y);
        put(k * 3 + 1, vx, e_psi);
        put(k * 3 + 2, vx, t);
        put(k * 2 + (3 * (N + 1)), vu, v);
        put(k * 2 + (3 * (N + 1) + 1), vu, kap);
      };
      if constexpr (ANY) L::rows_any(z, n, inst, n_inst, fill); else L::rows(z, n, inst, n_inst, fill);
    }
    if (y) {
      auto fill = [&](auto put) {
        put(k * 3, vx, (E2[0] * nu2[0]) * cinv);
        put(k * 3 + 1, vx, (E2[1] * nu2[1]) * cinv);
        put(k * 3 + 2, vx, zero);
        put(k * 3 + (3 * (N + 1)), vx, (lam3[0] * rcp_(D3[0])) * cinv);
        put(k * 3 + (3 * (N + 1) + 1), vx, (lam3[1] * rcp_(D3[1])) * cinv);
        put(k * 3 + (3 * (N + 1) + 2), vx, zero);
        put(k * 2 + (6 * (N + 1)), vu, lam_v);
        put(k * 2 + (6 * (N + 1) + 1), vu, (lam3[2] * rcp_(D3[2])) * cinv);
      };
      if constexpr (ANY) L::rows_any(y, m, inst, n_inst, fill); else L::rows(y, m, inst, n_inst, fill);
    }
    const Mk lead = live & first;
    if (u0) {
      L::store(u0, inst * 2, lead, v);
      L::store(u0, inst * 2 + 1, lead, atan_(kap * R(wheelbase)));        // src/MPC.py:188-189
    }
    if (st_out) L::storei(st_out, inst, lead, status);
    if (it_out) { L::storei(it_out, inst * 2, lead, iters); L::storei(it_out, inst * 2 + 1, lead, ipm_iters); }
    if (resid) { L::store(resid, inst * 2, lead, pri_res); L::store(resid, inst * 2 + 1, lead, dua_res); }
  }
};

}  // namespace mpmpc
