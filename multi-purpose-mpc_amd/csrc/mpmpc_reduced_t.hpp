// Reduced-native solver for weightings with a TERMINAL cost on the time state ("time-optimal" weights, README.md:56 of
// the reference: QN[2] > 0; BASELINE config 3) - the batch path of src/MPC.py:61-159,183 for such weights.
//
// The time state t enters no other state's dynamics and only the speed v drives it (mpmpc_reduced.hpp), so t is a running sum
//     t_N = x0[2] + sum_k (a20_k e_y_k + b20_k v_k) - sum_k beq2_k  =  c'(e_y, v) + d
// and when the ONLY cost on t is the terminal  1/2 QN[2] t_N^2  (Q[2] = 0, no bound on t, diagonal QN: src/MPC.py:150-155
// with the reference's xr = 0 on t) the QP of src/MPC.py is EXACTLY
//     min  1/2 x'P x + q'x + 1/2 w (c'x + d)^2      over x = (e_y, e_psi, kappa, v) per stage
//     s.t. the 2 x 2-block dynamics of (e_y, e_psi, kappa)   and the boxes on e_y, kappa, v
// i.e. the reduced problem of mpmpc_reduced.hpp, the speeds as boxed variables with a diagonal cost and no dynamics of their
// own, and ONE rank-one term in the Hessian.  Every KKT solve of the interior point and of the active-set rounds is then the
// reduced 2 x 2-block solve + a diagonal solve for the speeds + a Sherman-Morrison correction (Solver: LAY_RED4 - one extra
// right-hand side per factorisation, one wave reduction per solve): the arithmetic and the register budget of the
// reduced-native kernels (two wavefronts per SIMD) instead of the 3 x 3 chains of the general kernel (490 registers).
// The multipliers of the time rows are all equal to  theta = w t_N  (stationarity in t_k), which is what puts the rank-one
// gradient  theta c  on e_y and v; the store reports them, and tests/ check every answer against the FULL problem's KKT
// system with numpy (mpmpc_testlib.kkt_batch).
// What this solver cannot certify keeps status UNSOLVED and goes to the general kernel's tail launch (mode 2), as in
// mpmpc_reduced.hpp.
#pragma once
#include "mpmpc_reduced.hpp"

namespace mpmpc {

// May the polish work on the (e_y, e_psi, kappa, v) problem with the rank-one time term?
inline bool reducible_tt(const mpmpc_config& c, const mpmpc_settings& st) {
  return st.reduce != 0 && st.polish != 0 && c.Q[2] == 0.0 && c.QN[2] > 0.0 && c.QN_offdiag[0] == 0.0 && c.QN_offdiag[1] == 0.0 &&
         c.QN_offdiag[2] == 0.0 && c.R[0] > 0.0 && !(c.xmin[2] > -INFTY) && !(c.xmax[2] < INFTY) && !(c.xmin[1] > -INFTY) &&
         !(c.xmax[1] < INFTY);
}
inline bool reduced_native_tt(const mpmpc_config& c, const mpmpc_settings& st) {
  return st.native != 0 && reducible_tt(c, st) && st.early_polish == 1 && st.max_iter > 1 && st.ipm_start_mu > 0.0 &&
         st.early_scaling >= 0 && st.scaling > 0;
}

// One instance per wavefront only (G = 64): the cold slots of the certified point are free until the solve ends and serve as
// working storage of the loops (a packed wave would hold its partner instance's committed point there).
template <class L, bool CR = true>
struct ReducedTSolver : Solver<L, false, true, false, CR, 11> {
  static_assert(L::per_wave == 1, "the terminal-time kernels run one instance per wavefront");
  using S = Solver<L, false, true, false, CR, 11>;
  using R = typename L::real;
  using Mk = typename L::mask;
  using I = typename L::ival;
  using S::N; using S::n_inst; using S::off_; using S::vx; using S::vu; using S::first; using S::down_chain; using S::is_mid;
  using S::is_end; using S::vxc; using S::live; using S::a; using S::b; using S::mI; using S::leq; using S::status; using S::iters;
  using S::ipm_iters; using S::polished; using S::pri_res; using S::dua_res; using S::rk_c; using S::term;
  static constexpr int LAY4 = S::LAY_RED4;
  using Box4 = typename S::template BoxT<LAY4>;
  using Ipm4 = typename S::template IpmT<LAY4>;

  // ---- cold storage (LDS), 512 B per slot (40 slots = 20 KB per wave: two wavefronts per SIMD)
  //   C_D (4), C_E (2), C_C   scalings of the 4 columns, the 2 dynamics rows, the cost
  //   C_A20, C_B20, C_BEQ2    the time row: t_k = t_{k-1} + a20 e_y + b20 v - beq2  (also the unscaled rank-one vector)
  //   C_GAP                   width of an empty box
  //   C_G (4)                 box-row scaling of the start; at the end (same slots) C_XS: the certified point,
  //   C_LAM (4), C_NUS (2)    its box and equality multipliers.  Until then these ten slots work for the loops:
  //     S_U (6)               the Sherman-Morrison vector of the current factorisation (Solver: RKS)
  //     S_RD (4)              dual residual of the interior point's iteration (S_RP (2), the primal one: two of the free slots)
  //   K_LO* / K_HI*           boxes of e_y, kappa, v in the scaled variable space
  //   K_PP (4), K_QQ (4), K_LEQ (2)   cost and equality offsets of the interior point's loop
  //   K_PARK .. 39            (= K_LO0 ..: box, cost and offsets are in registers while the active-set rounds run, and three
  //                           free slots) the interior point's iterate during the rounds - only a further attempt reads it again
  enum { C_D = 0, C_E = 4, C_C = 6, C_A20 = 7, C_B20 = 8, C_BEQ2 = 9, C_GAP = 10, C_G = 11, C_XS = 11, C_LAM = 15, C_NUS = 19,
         S_U = 11, S_RD = 17, K_LO0 = 21, K_HI0 = 22, K_LO2 = 23, K_HI2 = 24, K_LO3 = 25, K_HI3 = 26, K_PP = 27, K_QQ = 31, K_LEQ = 35,
         S_RP = 37, S_RES = 37, K_PARK = 21, COLD_USED = 40 };
  static_assert(COLD_USED <= L::cold_slots, "lane backend has too few cold slots");
  static constexpr double BOX_INF = 1e20;
  static constexpr int RN_ATTEMPTS = 3;
  static constexpr double RN_RETRY = 1e-2;
  static constexpr double START_SLACK_FACTOR = 3.0, START_MU_FACTOR = 10.0;
  static constexpr int RN_IPM_CAP = 24;       // (time-optimal weights, N = 50: feasible instances take 8 - 16 iterations)

  R P4[4], Q4[4];
  Mk val[4];             // entry exists: (vx, vx, vu, vu)
  Mk solvable, empty;
  double w_time;         // QN[2]

  // ================================================================================ setup: load + Ruiz + cold
  MPMPC_HD void setup(const R* fields, int B, const I& inst, const I& k, int N_, const SolverParams& st, double qn_time) {
    N = N_;
    n_inst = B;
    w_time = qn_time;
    live = inst < B;
    vx = live & within_(k, 0, N);
    vu = live & within_(k, 0, N - 1);
    {
      const int C = L::split;
      off_ = lane_offset(L::group, C, N);
      I kl = k + off_;
      down_chain = (kl >= C);
      is_mid = (kl == C - 1);
      is_end = (kl == 2 * C - 1);
      I kc = seli(down_chain & (kl < 2 * C), kl * (-1) + (3 * C - 1), kl) - off_;
      vxc = live & within_(kc, 0, N);
    }
    val[0] = vx; val[1] = vx; val[2] = vu; val[3] = vu;
    auto fld = [&](int f, double dflt) { return sel(vx, fields[f], R(dflt)); };
    const R zero(0.0), onec(1.0);
    // (what the Ruiz passes do not touch goes to its cold slot at once - raw, and is put in scaled form afterwards - so that
    //  the passes, the register peak of the kernel's start, do not carry it)
    {
      const R lo_e = max_(fld(F_LO + 0, -INFTY), R(-INFTY)), hi_e = min_(fld(F_HI + 0, INFTY), R(INFTY));
      const R lo_k = max_(fld(F_LO + 4, -INFTY), R(-INFTY)), hi_k = min_(fld(F_HI + 4, INFTY), R(INFTY));
      const R lo_v = max_(fld(F_LO + 3, -INFTY), R(-INFTY)), hi_v = min_(fld(F_HI + 3, INFTY), R(INFTY));
      // an EMPTY box makes the QP trivially infeasible (Solver::run has the same rule): reported at once, never solved
      R gap = max_(max_(sel(vx, lo_e - hi_e, zero), sel(vu, lo_k - hi_k, zero)), sel(vu, lo_v - hi_v, zero));
      gap = L::gmax(gap);
      empty = live & (gap > zero);
      solvable = live & !empty;
      L::cold_put(C_GAP, gap);
      L::cold_put(K_LO0, lo_e); L::cold_put(K_HI0, hi_e); L::cold_put(K_LO2, lo_k); L::cold_put(K_HI2, hi_k);
      L::cold_put(K_LO3, lo_v); L::cold_put(K_HI3, hi_v);
      L::cold_put(K_LEQ, fld(F_BEQ + 0, 0.0)); L::cold_put(K_LEQ + 1, fld(F_BEQ + 1, 0.0));
    }
    const R ds = fld(F_DS, 0.0), one = sel(vu, onec, zero);
    a[0] = one; a[1] = ds; a[2] = fld(F_A10, 0.0); a[3] = one;
    b[0] = ds;
    mI[0] = mI[1] = R(-1.0);
    P4[0] = fld(F_P + 0, 1.0); P4[1] = fld(F_P + 1, 1.0); P4[2] = fld(F_P + 4, 1.0); P4[3] = fld(F_P + 3, 1.0);
    Q4[0] = fld(F_Q + 0, 0.0); Q4[1] = fld(F_Q + 1, 0.0); Q4[2] = fld(F_Q + 4, 0.0); Q4[3] = fld(F_Q + 3, 0.0);
    // ---- the time functional  t_N = c'(e_y, v) + d  (unscaled): c = (a20, b20) of the lane's stage (zero on stage N),
    // d = -sum beq2;  its cost 1/2 w t_N^2 adds  w d c  to the cost vector and  w c c'  to the Hessian
    {
      const R cr0 = sel(vu, fld(F_A20, 0.0), zero), cr1 = sel(vu, fld(F_B20, 0.0), zero);
      const R beq2 = fld(F_BEQ + 2, 0.0);
      L::cold_put(C_A20, cr0); L::cold_put(C_B20, cr1); L::cold_put(C_BEQ2, beq2);
      const R wd = R(w_time) * L::gsum(-beq2);
      Q4[0] = fma_(wd, cr0, Q4[0]);
      Q4[3] = fma_(wd, cr1, Q4[3]);
    }
    L::fence();
    R D4[4] = {onec, onec, onec, onec}, G4[4] = {onec, onec, onec, onec}, E2[2] = {onec, onec}, c4(1.0);
    // OSQP's scale_data() on this problem: Ruiz passes over the columns (e_y, e_psi, kappa, v) and the rows (2 dynamics rows,
    // 4 box rows), each with the cost normalisation (the rank-one term takes no part in the norms: its entries, w c_i c_j,
    // stay below the diagonal's for the path's curvatures and speeds).  Reciprocal square roots by rsqrt_ (1.2 ulp): this
    // kernel's scaling is its own, nothing has to reproduce it bit for bit.
    const int passes = st.early_scaling > 0 && st.early_scaling < st.scaling ? st.early_scaling : st.scaling;
    const R n_total(double(4 * N + 2));
    for (int it = 0; it < passes; ++it) {
      R cn[4], r_own[2];
      cn[0] = max_(max_(max_(abs_(P4[0]), abs_(mI[0])), max_(abs_(a[0]), abs_(a[2]))), abs_(G4[0]));
      cn[1] = max_(max_(max_(abs_(P4[1]), abs_(mI[1])), max_(abs_(a[1]), abs_(a[3]))), abs_(G4[1]));
      cn[2] = max_(max_(abs_(P4[2]), abs_(b[0])), abs_(G4[2]));
      cn[3] = max_(abs_(P4[3]), abs_(G4[3]));
      r_own[0] = max_(abs_(a[0]), abs_(a[1]));
      r_own[1] = max_(max_(abs_(a[2]), abs_(a[3])), abs_(b[0]));
      R Dt[4], Et[2], Etd[2];
      MPMPC_UNROLL
      for (int i = 0; i < 2; ++i) {
        Et[i] = rsqrt_(S::limit(max_(abs_(mI[i]), L::up(r_own[i]))));
        Etd[i] = L::down(Et[i]);
      }
      MPMPC_UNROLL
      for (int e = 0; e < 4; ++e) {
        Dt[e] = rsqrt_(S::limit(cn[e]));
        const R Etb = rsqrt_(S::limit(abs_(G4[e])));
        P4[e] = (Dt[e] * P4[e]) * Dt[e];
        G4[e] = (Etb * G4[e]) * Dt[e];
        Q4[e] = Dt[e] * Q4[e];
        D4[e] = D4[e] * Dt[e];
      }
      MPMPC_UNROLL
      for (int i = 0; i < 2; ++i) { mI[i] = (Et[i] * mI[i]) * Dt[i]; E2[i] = E2[i] * Et[i]; }
      a[0] = (Etd[0] * a[0]) * Dt[0]; a[1] = (Etd[0] * a[1]) * Dt[1];
      a[2] = (Etd[1] * a[2]) * Dt[0]; a[3] = (Etd[1] * a[3]) * Dt[1];
      b[0] = (Etd[1] * b[0]) * Dt[2];
      R s(0.0), mq(0.0);
      MPMPC_UNROLL
      for (int e = 0; e < 4; ++e) {
        s = s + sel(val[e], abs_(P4[e]), zero);
        mq = max_(mq, sel(val[e], abs_(Q4[e]), zero));
      }
      R ct = L::gsum(s) / n_total;
      const R nq = S::limit(L::gmax(mq));
      ct = rcp_(S::limit(max_(ct, nq)));
      MPMPC_UNROLL
      for (int e = 0; e < 4; ++e) { P4[e] = P4[e] * ct; Q4[e] = Q4[e] * ct; }
      c4 = c4 * ct;
    }
    L::fence();
    leq[0] = E2[0] * L::cold_get(K_LEQ);
    leq[1] = E2[1] * L::cold_get(K_LEQ + 1);
    // the rank-one vector in the scaled problem:  1/2 (c4 w) (c' D x)^2  =  1/2 (rk_c' x)^2
    {
      const R sw = sqrt_(c4 * R(w_time));
      rk_c[0] = (sw * D4[0]) * L::cold_get(C_A20);
      rk_c[1] = (sw * D4[3]) * L::cold_get(C_B20);
    }
    MPMPC_UNROLL
    for (int e = 0; e < 4; ++e) { L::cold_put(C_D + e, D4[e]); L::cold_put(C_G + e, G4[e]); }
    L::cold_put(C_E, E2[0]); L::cold_put(C_E + 1, E2[1]);
    L::cold_put(C_C, c4);
    // box in the scaled variable space:  g x in [lb, ub]  <=>  x in [lo_raw, hi_raw] / D
    {
      const R i0 = rcp_(D4[0]), i2 = rcp_(D4[2]), i3 = rcp_(D4[3]);
      L::cold_put(K_LO0, L::cold_get(K_LO0) * i0); L::cold_put(K_HI0, L::cold_get(K_HI0) * i0);
      L::cold_put(K_LO2, L::cold_get(K_LO2) * i2); L::cold_put(K_HI2, L::cold_get(K_HI2) * i2);
      L::cold_put(K_LO3, L::cold_get(K_LO3) * i3); L::cold_put(K_HI3, L::cold_get(K_HI3) * i3);
    }
    L::fence();
  }

  // ================================================================================ interior point, 4 entries per lane
  // The iteration of Solver::ipm (regularised Mehrotra predictor-corrector, same constants and step rules) written like
  // ReducedSolver::ipm3 for a small register footprint: the loop invariants (box, cost, equality offsets), the residuals of the
  // iteration and the Sherman-Morrison vector wait in LDS and are re-read where they are used.  Entry 1 (e_psi) is never boxed
  // (reducible_tt()); entries 0, 2, 3 (e_y, kappa, v) are.  A PINNED entry (lo = hi: e_y of stage 0) is eliminated instead
  // of being held by a proximal term: it starts on its value and its diagonal inverse is zero, so that no step moves it and
  // no multiplier has to be carried for it (s.pi is not used; the active-set rounds compute the pin's multiplier).
  MPMPC_HD Mk ipm4(const Box4& bx, Ipm4& s, const SolverParams& st, double tol, const Mk& run) {
    const R reg(st.ipm_reg), one(1.0), zero(0.0);
    constexpr int NB = 3;
    constexpr int JB[NB] = {0, 2, 3};
    Mk active = run, conv = L::mfalse();
    R cnt(0.0);
    MPMPC_UNROLL
    for (int b = 0; b < NB; ++b) cnt = cnt + L::gcount(bx.Lm[JB[b]]) + L::gcount(bx.Um[JB[b]]);
    const R inb = rcp_(max_(cnt, one));
    I stall(0);
    R mu_min(1e300);
    auto lo_of = [&](int b) { return L::cold_get(b == 0 ? K_LO0 : (b == 1 ? K_LO2 : K_LO3)); };
    auto hi_of = [&](int b) { return L::cold_get(b == 0 ? K_HI0 : (b == 1 ? K_HI2 : K_HI3)); };
    // A side without a bound (infinite, pinned entry, lane without a stage) carries a zero multiplier and - in the products
    // that eliminate the slack steps - a zero in place of its slack reciprocal (isl / isu below): its multiplier step is then
    // exactly zero whatever its slack residual is, and that residual (finite: the clipped infinities are 1e30) needs no mask
    // except where it would enter a norm.  The iterates are bit for bit those of the fully masked form (Solver::ipm).
    auto rl_of = [&](int b) { const int j = JB[b]; return s.x[j] - lo_of(b) - s.sl[j]; };
    auto ru_of = [&](int b) { const int j = JB[b]; return hi_of(b) - s.x[j] - s.su[j]; };
    for (int it = 0; it <= st.ipm_max_iter; ++it) {
      R mu;
      MPMPC_TICK_BEGIN(10);
      {
        // ---- residuals -> LDS
        L::fence();
        R At[4], rd[4], rp[2];
        this->template AeqT_mul_t<LAY4>(s.nu, At);
        this->template Aeq_mul_t<LAY4>(s.x, rp);
        const R rk_dot = this->rank_one_dot(s.x);
        R res(0.0), msum(0.0);
        MPMPC_UNROLL
        for (int i = 0; i < 2; ++i) { rp[i] = rp[i] - L::cold_get(K_LEQ + i); res = max_(res, sel(vx, abs_(rp[i]), zero)); }
        MPMPC_UNROLL
        for (int j = 0; j < 4; ++j) {
          rd[j] = fma_(L::cold_get(K_PP + j), s.x[j], L::cold_get(K_QQ + j)) + At[j] - s.zl[j] + s.zu[j];
          if (j == 0 || j == 3) rd[j] = fma_(rk_c[j == 0 ? 0 : 1], rk_dot, rd[j]);
          if (j != 1) rd[j] = sel(bx.pin[j], zero, rd[j]);          // (a pinned entry has no stationarity row)
        }
        res = max_(res, sel(val[1], abs_(rd[1]), zero));
        MPMPC_UNROLL
        for (int b = 0; b < NB; ++b) {
          const int j = JB[b];
          res = max_(res, sel(val[j], max_(abs_(rd[j]), max_(sel(bx.Lm[j], abs_(rl_of(b)), zero), sel(bx.Um[j], abs_(ru_of(b)), zero))), zero));
          // (a side without a bound keeps a finite slack and an exactly zero multiplier: its product is an exact zero)
          msum = msum + s.sl[j] * s.zl[j] + s.su[j] * s.zu[j];
        }
        MPMPC_UNROLL
        for (int j = 0; j < 4; ++j) L::cold_put(S_RD + j, rd[j]);
        L::cold_put(S_RP, rp[0]); L::cold_put(S_RP + 1, rp[1]);
        res = L::gmax(res);
        mu = L::gsum(msum) * inb;
        const Mk ok = (res < R(tol > 1e-11 ? tol : 1e-11)) & (mu < R(tol));
        conv = conv | (active & ok);
        active = active & !ok;
        MPMPC_TICK_END(10);
        if (it == st.ipm_max_iter || !L::wany(active)) break;
        active = active & !(mu > R(st.ipm_diverged) * mu_min) & !((mu < R(tol * 1e-3)) & (res > R(1e-5)));
        mu_min = min_(mu_min, mu);
        if (!L::wany(active)) break;
      }
      MPMPC_TICK_COUNT(16);
      ipm_iters = seli(active, ipm_iters + I(1), ipm_iters);
      // ---- factor (+ the Sherman-Morrison vector of this factorisation)
      R isl[NB], isu[NB], rcl[NB], rcu[NB];
      MPMPC_TICK_BEGIN(11);
      {
        L::fence();
        R h[4];
        h[1] = rcp_(L::cold_get(K_PP + 1) + reg);
        MPMPC_UNROLL
        for (int b = 0; b < NB; ++b) {
          const int j = JB[b];
          const R il = rcp_(s.sl[j]), iu = rcp_(s.su[j]);
          h[j] = sel(bx.pin[j], zero, rcp_(L::cold_get(K_PP + j) + reg + sel(bx.Lm[j], s.zl[j] * il, zero) + sel(bx.Um[j], s.zu[j] * iu, zero)));
        }
        this->template factor_t<LAY4>(h, reg);
        L::fence();
        MPMPC_UNROLL
        for (int b = 0; b < NB; ++b) {
          const int j = JB[b];
          isl[b] = sel(bx.Lm[j], rcp_(s.sl[j]), zero); isu[b] = sel(bx.Um[j], rcp_(s.su[j]), zero);
          rcl[b] = s.sl[j] * s.zl[j]; rcu[b] = s.su[j] * s.zu[j];
        }
      }
      MPMPC_TICK_END(11);
      R alpha_aff(1.0);
      for (int pass = 0; pass < 2; ++pass) {
        R dx[4], dnu[2];
        {
          L::fence();
          R rhs[4], nreq[2];
          rhs[1] = -L::cold_get(S_RD + 1);
          MPMPC_UNROLL
          for (int b = 0; b < NB; ++b) {
            const int j = JB[b];
            rhs[j] = -L::cold_get(S_RD + j) - fma_(s.zl[j], rl_of(b), rcl[b]) * isl[b] + fma_(s.zu[j], ru_of(b), rcu[b]) * isu[b];
          }
          nreq[0] = -L::cold_get(S_RP); nreq[1] = -L::cold_get(S_RP + 1);
          MPMPC_TICK_BEGIN(12);
          this->template kkt_solve_t<LAY4>(rhs, nreq, dx, dnu);
          MPMPC_TICK_END(12);
        }
        L::fence();
        R dsl[NB], dsu[NB], dzl[NB], dzu[NB];
        R blk(0.0);
        MPMPC_UNROLL
        for (int b = 0; b < NB; ++b) {
          const int j = JB[b];
          dsl[b] = dx[j] + rl_of(b);
          dsu[b] = -dx[j] + ru_of(b);
          dzl[b] = -fma_(s.zl[j], dsl[b], rcl[b]) * isl[b];
          dzu[b] = -fma_(s.zu[j], dsu[b], rcu[b]) * isu[b];
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
          for (int b = 0; b < NB; ++b) {
            const int j = JB[b];
            ms = ms + fma_(alpha_aff, dsl[b], s.sl[j]) * fma_(alpha_aff, dzl[b], s.zl[j]) +
                 fma_(alpha_aff, dsu[b], s.su[j]) * fma_(alpha_aff, dzu[b], s.zu[j]);
          }
          const R mu_aff = L::gsum(ms) * inb;
          R sg = mu_aff * rcp_(max_(mu, R(1e-300)));
          sg = sg * sg * sg;
          const R sgmu = sg * mu;
          MPMPC_UNROLL
          for (int b = 0; b < NB; ++b) {
            const int j = JB[b];
            rcl[b] = fma_(dsl[b], dzl[b], fma_(s.sl[j], s.zl[j], -sgmu));
            rcu[b] = fma_(dsu[b], dzu[b], fma_(s.su[j], s.zu[j], -sgmu));
          }
        } else {
          const R al = min_(one, R(0.995) * ratio);
          stall = seli(active & (al < R(1e-6)), stall + I(1), I(0));
          s.x[1] = sel(active, fma_(al, dx[1], s.x[1]), s.x[1]);
          MPMPC_UNROLL
          for (int b = 0; b < NB; ++b) {
            const int j = JB[b];
            s.x[j] = sel(active, fma_(al, dx[j], s.x[j]), s.x[j]);
            s.tL[j] = selb(active, bx.Lm[j] & (dsl[b] * s.zl[j] < dzl[b] * s.sl[j]), s.tL[j]);
            s.tU[j] = selb(active, bx.Um[j] & (dsu[b] * s.zu[j] < dzu[b] * s.su[j]), s.tU[j]);
            s.sl[j] = sel(active, fma_(al, dsl[b], s.sl[j]), s.sl[j]);
            s.su[j] = sel(active, fma_(al, dsu[b], s.su[j]), s.su[j]);
            s.zl[j] = sel(active, fma_(al, dzl[b], s.zl[j]), s.zl[j]);
            s.zu[j] = sel(active, fma_(al, dzu[b], s.zu[j]), s.zu[j]);
          }
          s.nu[0] = sel(active, fma_(al, dnu[0], s.nu[0]), s.nu[0]);
          s.nu[1] = sel(active, fma_(al, dnu[1], s.nu[1]), s.nu[1]);
          active = active & (stall < 3);
        }
      }
    }
    return conv;
  }

  // ================================================================================ certificate (this problem, unscaled)
  MPMPC_HD Mk certificate4(const Box4& bx, const R* pp, const R* qq, const R xs[4], const R nus[2], const R lam[4], double tol, R& prim, R& stat) const {
    const R zero(0.0);
    R Ax[2], At[4];
    this->template Aeq_mul_t<LAY4>(xs, Ax);
    this->template AeqT_mul_t<LAY4>(nus, At);
    const R rk_dot = this->rank_one_dot(xs);
    const R cinv = R(1.0) / L::cold_get(C_C);
    R pv(0.0), sv(0.0), cv(0.0);
    MPMPC_UNROLL
    for (int i = 0; i < 2; ++i) pv = max_(pv, sel(vx, abs_((Ax[i] - leq[i]) / L::cold_get(C_E + i)), zero));
    Mk bad = L::mfalse();
    MPMPC_UNROLL
    for (int e = 0; e < 4; ++e) {
      const R De = L::cold_get(C_D + e);
      const R lo0 = bx.lo[e], hi0 = bx.hi[e];      // (the box in scaled units; distances go back to the unscaled problem through D)
      const Mk fu = hi0 < R(BOX_INF), fl = lo0 > R(-BOX_INF);
      const R dlo = sel(fl, De * (xs[e] - lo0), R(INFTY)), dhi = sel(fu, De * (hi0 - xs[e]), R(INFTY));   // unscaled distances to the bounds
      pv = max_(pv, sel(val[e], max_(max_(-dlo, -dhi), zero), zero));
      R rd = fma_(pp[e], xs[e], qq[e]) + At[e] + lam[e];
      if (e == 0 || e == 3) rd = fma_(rk_c[e == 0 ? 0 : 1], rk_dot, rd);
      sv = max_(sv, sel(val[e], abs_(rd / De) * cinv, zero));
      const R yu = (lam[e] / De) * cinv;                  // multiplier of the unscaled box row
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

  // While the active-set rounds run (their factorisation and refinement on top of the caller's state), the interior point's
  // iterate waits in LDS, in the slots of the loop invariants (in registers then): only a further attempt reads it again.
  MPMPC_HD void park_ip(const Ipm4& s) {
    L::fence();
    constexpr int JB[3] = {0, 2, 3};
    MPMPC_UNROLL
    for (int j = 0; j < 4; ++j) L::cold_put(K_PARK + j, s.x[j]);
    L::cold_put(K_PARK + 4, s.nu[0]); L::cold_put(K_PARK + 5, s.nu[1]);
    MPMPC_UNROLL
    for (int b = 0; b < 3; ++b) {
      const int j = JB[b];
      L::cold_put(K_PARK + 6 + 4 * b + 0, s.sl[j]); L::cold_put(K_PARK + 6 + 4 * b + 1, s.su[j]);
      L::cold_put(K_PARK + 6 + 4 * b + 2, s.zl[j]); L::cold_put(K_PARK + 6 + 4 * b + 3, s.zu[j]);
    }
    L::fence();
  }
  // ... and back, with the invariants of the loop in their slots again
  MPMPC_HD void unpark_ip(Ipm4& s, const Box4& bx) {
    L::fence();
    constexpr int JB[3] = {0, 2, 3};
    MPMPC_UNROLL
    for (int j = 0; j < 4; ++j) s.x[j] = L::cold_get(K_PARK + j);
    s.nu[0] = L::cold_get(K_PARK + 4); s.nu[1] = L::cold_get(K_PARK + 5);
    MPMPC_UNROLL
    for (int b = 0; b < 3; ++b) {
      const int j = JB[b];
      s.sl[j] = L::cold_get(K_PARK + 6 + 4 * b + 0); s.su[j] = L::cold_get(K_PARK + 6 + 4 * b + 1);
      s.zl[j] = L::cold_get(K_PARK + 6 + 4 * b + 2); s.zu[j] = L::cold_get(K_PARK + 6 + 4 * b + 3);
    }
    s.sl[1] = s.su[1] = R(1.0); s.zl[1] = s.zu[1] = R(0.0);
    L::fence();
    MPMPC_UNROLL
    for (int e = 0; e < 4; ++e) { L::cold_put(K_PP + e, P4[e]); L::cold_put(K_QQ + e, Q4[e]); }
    L::cold_put(K_LEQ, leq[0]); L::cold_put(K_LEQ + 1, leq[1]);
    L::cold_put(K_LO0, bx.lo[0]); L::cold_put(K_HI0, bx.hi[0]);
    L::cold_put(K_LO2, bx.lo[2]); L::cold_put(K_HI2, bx.hi[2]);
    L::cold_put(K_LO3, bx.lo[3]); L::cold_put(K_HI3, bx.hi[3]);
    L::fence();
  }

  // the certified point goes to cold storage (the working slots of the loops, which are done with)
  MPMPC_HD void commit(const Mk& good, const R xa[4], const R na[2], const R la[4], const R& prim, const R& stat) {
    L::fence();
    MPMPC_UNROLL
    for (int e = 0; e < 4; ++e) { L::cold_put(C_XS + e, xa[e]); L::cold_put(C_LAM + e, la[e]); }
    L::cold_put(C_NUS, na[0]); L::cold_put(C_NUS + 1, na[1]);
    L::fence();
    // (the residuals of the certificate wait for the store in cold slots as well - behind the rows the store stages)
    const R zero(0.0);
    L::cold_put(S_RES, sel(good, prim, zero));
    L::cold_put(S_RES + 1, sel(good, stat, zero));
    L::fence();
    status = seli(good, I(MPMPC_SOLVED), I(MPMPC_UNSOLVED));
    polished = seli(good, I(1), I(0));
  }

  // ================================================================================ the solve
  // qn_time: QN[2], the terminal weight of the time state (mpmpc_config)
  MPMPC_HD void run(const R* fields, int B, const I& inst, const I& k, int N_, const SolverParams& st, double qn_time) {
    MPMPC_TICK_BEGIN(0);
    setup(fields, B, inst, k, N_, st, qn_time);
    MPMPC_TICK_END(0);
    const R zero(0.0), one(1.0);
    status = I(MPMPC_UNSOLVED);
    iters = I(1);
    ipm_iters = I(0);
    polished = I(0);
    this->act_bits = I(0);
    L::cold_put(S_RES, zero); L::cold_put(S_RES + 1, zero);
    Mk todo = solvable;
    if (L::wany(todo)) {
      Box4 b4;
      {
        const R lo_s[4] = {L::cold_get(K_LO0), R(-INFTY), L::cold_get(K_LO2), L::cold_get(K_LO3)};
        const R hi_s[4] = {L::cold_get(K_HI0), R(INFTY), L::cold_get(K_HI2), L::cold_get(K_HI3)};
        MPMPC_UNROLL
        for (int e = 0; e < 4; ++e) {
          const Mk fl = lo_s[e] > R(-BOX_INF), fu = hi_s[e] < R(BOX_INF);
          const Mk pn = fl & fu & ((hi_s[e] - lo_s[e]) <= R(1e-12) * max_(one, abs_(lo_s[e])));
          b4.lo[e] = lo_s[e];
          b4.hi[e] = hi_s[e];
          b4.pin[e] = pn & val[e];
          b4.Lm[e] = fl & !pn & val[e];
          b4.Um[e] = fu & !pn & val[e];
        }
      }
      // ---- centred start of the interior point from x = 0 (ReducedSolver::run): slacks max(distance to the bound,
      // ipm_start_slack in row space), multipliers mu0 / slack, no equality multipliers
      Ipm4 si;
      {
        // (the start of this problem family sits further inside than the tracking weights' - the terminal time cost pulls every
        //  speed to its upper bound: three times the slack floor and ten times the complementarity of the settings, which are
        //  tuned on the reference's own weights.  Emulation, 1 024 instances of config 3: 10.23 -> 9.58 iterations; DESIGN.md
        //  6a found the same factor of ten on the general kernel)
        const R ths(START_SLACK_FACTOR * st.ipm_start_slack);
        R mu0(START_MU_FACTOR * st.ipm_start_mu);
        if (st.ipm_start_dual > 0.0) {
          R rd0(0.0);
          MPMPC_UNROLL
          for (int e = 0; e < 4; ++e) rd0 = max_(rd0, sel(val[e], abs_(Q4[e]), zero));
          mu0 = max_(mu0, (R(st.ipm_start_dual) * ths) * L::gmax(rd0));
        }
        // (the loop's invariants go to their slots first: cost and offsets are not needed in registers before the rounds)
        MPMPC_UNROLL
        for (int e = 0; e < 4; ++e) { L::cold_put(K_PP + e, P4[e]); L::cold_put(K_QQ + e, Q4[e]); }
        L::cold_put(K_LEQ, leq[0]); L::cold_put(K_LEQ + 1, leq[1]);
        L::fence();
        MPMPC_UNROLL
        for (int e = 0; e < 4; ++e) {
          const R fl = ths * rcp_(L::cold_get(C_G + e));
          si.x[e] = sel(b4.pin[e], b4.lo[e], zero);          // (a pinned entry sits on its value from the start: ipm4)
          si.sl[e] = sel(b4.Lm[e], max_(-b4.lo[e], fl), one);
          si.su[e] = sel(b4.Um[e], max_(b4.hi[e], fl), one);
          si.zl[e] = sel(b4.Lm[e], mu0 * rcp_(si.sl[e]), zero);
          si.zu[e] = sel(b4.Um[e], mu0 * rcp_(si.su[e]), zero);
          si.tL[e] = b4.Lm[e] & (si.zl[e] > si.sl[e]);
          si.tU[e] = b4.Um[e] & (si.zu[e] > si.su[e]);
        }
        si.nu[0] = si.nu[1] = zero;
      }
      SolverParams sc = st;
      sc.ipm_max_iter = st.ipm_max_iter < RN_IPM_CAP ? st.ipm_max_iter : RN_IPM_CAP;
      double tol = st.native_ipm_tol;
      for (int attempt = 0; attempt < RN_ATTEMPTS; ++attempt) {
        MPMPC_TICK_BEGIN(4);
        const Mk conv = ipm4(b4, si, sc, tol, todo);
        // the interior point read its invariants from LDS; what follows takes them from there as well
        L::fence();
        MPMPC_UNROLL
        for (int e = 0; e < 4; ++e) { P4[e] = L::cold_get(K_PP + e); Q4[e] = L::cold_get(K_QQ + e); }
        leq[0] = L::cold_get(K_LEQ); leq[1] = L::cold_get(K_LEQ + 1);
        b4.lo[0] = L::cold_get(K_LO0); b4.hi[0] = L::cold_get(K_HI0);
        b4.lo[2] = L::cold_get(K_LO2); b4.hi[2] = L::cold_get(K_HI2);
        b4.lo[3] = L::cold_get(K_LO3); b4.hi[3] = L::cold_get(K_HI3);
        MPMPC_TICK_END(4);
        // active-set guess: the indicators of the interior point's last step
        Mk aL[4], aU[4];
        MPMPC_UNROLL
        for (int e = 0; e < 4; ++e) { aL[e] = b4.Lm[e] & si.tL[e]; aU[e] = b4.Um[e] & si.tU[e] & !aL[e]; }
        R xa[4] = {zero, zero, zero, zero}, la[4] = {zero, zero, zero, zero}, na[2] = {zero, zero};
        park_ip(si);
        MPMPC_TICK_BEGIN(5);
        const double frac = attempt == 0 ? st.as_add_fraction : (st.as_add_fraction > 0.5 ? st.as_add_fraction : 0.5);
        const Mk okm = this->template active_set<LAY4>(b4, P4, Q4, val, aL, aU, xa, na, la, st, todo & conv, frac);
        MPMPC_TICK_END(5);
        R prim, stat;
        MPMPC_TICK_BEGIN(6);
        const Mk cert = certificate4(b4, P4, Q4, xa, na, la, st.cert_tol, prim, stat);
        MPMPC_TICK_END(6);
        const Mk good = todo & conv & okm & cert;
        todo = todo & conv & !good;          // a diverged interior-point run is not retried
        if (!L::wany(todo)) {
          // (one instance per wave: the attempts end here, certified or not - only now do the working slots take the point)
          commit(good, xa, na, la, prim, stat);
          break;
        }
        unpark_ip(si, b4);
        tol *= RN_RETRY;
      }
    }
    // empty box: infeasible, zero ray, the width of the gap in resid[0] (store)
    status = seli(empty, I(MPMPC_PRIMAL_INFEASIBLE), status);
  }

  // ================================================================================ output
  // z in the reference's ordering, u0 = (v_0, delta_0), multipliers in the reference's row order; the time state is rolled
  // forward here, in the unscaled problem, and its rows' multipliers are  theta = w t_N  on every stage
  MPMPC_HD void store(const I& inst, const I& k, double wheelbase, double* z, double* u0, int* st_out, int* it_out,
                      double* resid, double* y) const {
    const int n = 5 * N + 3, m = 8 * N + 6;
    const R zero(0.0);
    // (the lane masks of the output are formed again from (inst, k) rather than carried through the solve)
    const Mk live = inst < n_inst, vx = live & within_(k, 0, N), vu = live & within_(k, 0, N - 1), first = (k == 0), term = live & (k == N);
    const Mk ok = live & (status == MPMPC_SOLVED);
    const R cinv = R(1.0) / L::cold_get(C_C);
    R D4[4];
    MPMPC_UNROLL
    for (int e = 0; e < 4; ++e) D4[e] = L::cold_get(C_D + e);
    R xs[4], lam4[4], nu2[2];
    MPMPC_UNROLL
    for (int e = 0; e < 4; ++e) { xs[e] = sel(ok, L::cold_get(C_XS + e), zero); lam4[e] = sel(ok, L::cold_get(C_LAM + e), zero); }
    nu2[0] = sel(ok, L::cold_get(C_NUS), zero); nu2[1] = sel(ok, L::cold_get(C_NUS + 1), zero);
    const R e_y = D4[0] * xs[0], e_psi = D4[1] * xs[1], kap = D4[2] * xs[2], v = D4[3] * xs[3];
    // t: row 2 of equality block k is  -t_k + a20 e_y_{k-1} + t_{k-1} + b20 v_{k-1} = beq2_k  (block 0: -t_0 = -x0[2])
    const R drive = fma_(L::cold_get(C_A20), e_y, L::cold_get(C_B20) * v), beq2 = L::cold_get(C_BEQ2);
    const R t0 = -beq2;
    R t = t0;
    // (t_k = t_{k-1} + drive_{k-1} - beq2_k: an inclusive prefix sum along the stages, L::gscan - 5 / 6 shifted adds instead of N
    //  dependent steps)
    t = L::gscan(sel(first, t0, sel(vx, L::up(drive) - beq2, R(0.0))));
    t = sel(ok & vx, t, zero);
    const R theta = R(w_time) * L::gsum(sel(term, t, zero));      // multiplier of every time row:  w t_N
    const R E2[2] = {L::cold_get(C_E), L::cold_get(C_E + 1)};
    L::fence();
    if (z) {
      L::rows(z, n, inst, n_inst, [&](auto put) {
        put(k * 3, vx, e_y);
        put(k * 3 + 1, vx, e_psi);
        put(k * 3 + 2, vx, t);
        put(k * 2 + (3 * (N + 1)), vu, v);
        put(k * 2 + (3 * (N + 1) + 1), vu, kap);
      });
    }
    if (y) {
      L::rows(y, m, inst, n_inst, [&](auto put) {
        put(k * 3, vx, (E2[0] * nu2[0]) * cinv);
        put(k * 3 + 1, vx, (E2[1] * nu2[1]) * cinv);
        put(k * 3 + 2, vx, theta);
        put(k * 3 + (3 * (N + 1)), vx, (lam4[0] / D4[0]) * cinv);
        put(k * 3 + (3 * (N + 1) + 1), vx, (lam4[1] / D4[1]) * cinv);
        put(k * 3 + (3 * (N + 1) + 2), vx, zero);
        put(k * 2 + (6 * (N + 1)), vu, (lam4[3] / D4[3]) * cinv);
        put(k * 2 + (6 * (N + 1) + 1), vu, (lam4[2] / D4[2]) * cinv);
      });
    }
    const Mk lead = live & first;
    if (u0) {
      L::store(u0, inst * 2, lead, v);
      L::store(u0, inst * 2 + 1, lead, atan_(kap * R(wheelbase)));        // src/MPC.py:188-189
    }
    if (st_out) L::storei(st_out, inst, lead, status);
    if (it_out) { L::storei(it_out, inst * 2, lead, iters); L::storei(it_out, inst * 2 + 1, lead, ipm_iters); }
    if (resid) {
      // (one instance per wave: the rows staged above end below slot 7 - 8 N + 6 <= 510 doubles - and these slots lie behind)
      L::fence();
      const R res0 = L::cold_get(S_RES), res1 = L::cold_get(S_RES + 1), gap = L::cold_get(C_GAP);
      L::store(resid, inst * 2, lead, sel(status == MPMPC_PRIMAL_INFEASIBLE, gap, res0));
      L::store(resid, inst * 2 + 1, lead, res1);
    }
  }
};

}  // namespace mpmpc
