// The TAIL of a reduced-native launch, reduced-native as well: what ReducedSolver could not certify is almost always
// infeasible (a corridor that closes), marginally so (by millimetres: solved on relaxed boxes, mpmpc_settings::phase1_accept) or
// a feasible instance whose interior point ran into the iteration cap.  The general kernel in mode 2 (Solver::run) decides
// all of these with phase 1 and one more attempt of the certified polish, both of which work on the (e_y, e_psi, kappa)
// problem - but it carries the 5-entry problem, the OSQP iteration and their state around them: 458 registers, ONE wave per
// SIMD, and a SIMD that holds a tail wave holds nothing else (config 4: the tail kernel took 39 % of the SIMD time of a
// step for 10.7 % of its instances).
//
// This solver does the same two steps on ReducedSolver's footing: three entries per lane, 40 LDS slots, at most 256
// registers - a tail wave shares its SIMD with another wave, and carries two instances where the horizon fits 32 lanes.
// Phase 1 is Solver::phase1 restated for three entries, on the SAME scaling as the general kernel's (the Ruiz sweeps of the
// FULL problem, Solver::ruiz: the least-violation point of phase 1 is a minimum in the scaled metric, so the scaling is part
// of the answer) and through the same interior point - ReducedSolver::ipm3<SOFT> in the packed layouts (<32,16>, <64,32>),
// Solver::ipm<LAY_REDSPLIT, true> itself in the split layout (<64,16>: the general kernel's answers to the last bits); the
// verdict is OSQP's primal-infeasibility test on the ray, in unscaled terms, as there.  Every verdict is a per-instance mask.
// What it leaves UNSOLVED (nothing, on the BASELINE configurations) goes on to the general kernel.
//
// Replaces, per instance: what osqp.solve() (src/MPC.py:183) answers on an infeasible or marginally infeasible QP - status
// "primal infeasible" after hundreds of ADMM iterations, or a plan that violates the corridor by millimetres.
#pragma once
#include "mpmpc_reduced.hpp"

namespace mpmpc {

// May the tail of a reduced-native launch run this solver first?  (every configuration the first kernel runs, with phase 1 on)
inline bool reduced_native_tail(const mpmpc_config& c, const mpmpc_settings& st) {
  return reduced_native(c, st) && st.phase1 != 0;
}

template <class L, bool CR = true>
struct ReducedTailSolver : ReducedSolver<L, CR> {
  using RS = ReducedSolver<L, CR>;
  using S = typename RS::S;
  using R = typename L::real;
  using Mk = typename L::mask;
  using I = typename L::ival;
  using S::N; using S::vx; using S::vu; using S::first; using S::live; using S::a; using S::b; using S::mI; using S::leq;
  using S::status; using S::iters; using S::ipm_iters; using S::polished; using S::pri_res; using S::dua_res;
  using S::sU; using S::val3; using S::bU;
  using RS::val; using RS::P3; using RS::Q3; using RS::x3; using RS::solvable; using RS::empty;
  using RS::C_D; using RS::C_E; using RS::C_C; using RS::K_LO0; using RS::K_HI0; using RS::K_LO2; using RS::K_HI2; using RS::C_V;
  using RS::C_LAMV; using RS::C_A20; using RS::C_BV; using RS::C_BEQ2; using RS::C_G; using RS::C_PI; using RS::C_XS; using RS::C_LAM;
  using RS::C_NUS; using RS::C_GAP;
  using Box3 = typename RS::Box3;
  using BoxI = typename RS::BoxI;
  using IpmI = typename RS::IpmI;
  static constexpr int EI = RS::EI;
  // ---- cold slots of its own, in places nothing else uses until the verdict is out: the last three slots of the parking
  //      area (the packed interior point's residuals end below them) and the slots of the start's pin multipliers (there are
  //      none here: attempts() is given the box-row scalings in registers and then reads neither)
  //   T_RAW0 .. 3    unscaled box of e_y and kappa (the verdict and the relaxation work in the unscaled problem)
  //   T_NAX          largest finite bound / speed of the instance (OSQP's max(|Ax|, |z|), see Solver::phase1)
  //   T_RD0          dual residual of the speed entry at the start of the attempt (scaled problem)
  //   T_MARK         C_GAP's slot (the gap of an empty box is kept in a register here): the violation of a MARGINAL instance on
  //                  its lanes, zero otherwise - read back after the attempt, so that no mask and no scalar of the verdict
  //                  lives through it
  // The least-violation point of a verdict (infeasible: with its ray; marginal: with zero multipliers, in case the attempt
  // fails) goes to the slots of the certified point BEFORE the attempt, which then commits in merge mode: what it certifies
  // replaces it, what it does not leaves it there.
  enum { T_RAW0 = 37, T_RAW1 = 38, T_RAW2 = 39, T_RAW3 = RS::C_PI, T_NAX = RS::C_PI + 1, T_RD0 = RS::C_PI + 2, T_MARK = RS::C_GAP,
         T_BAND = RS::C_NUS };      // (the slots of the certified point's equality multipliers: written when a verdict is out)
  static_assert(RS::K_RP + 2 <= 37, "the packed interior point's residual slots must end below T_RAW0");
  MPMPC_HD static constexpr int t_raw(int i) { return i == 0 ? T_RAW0 : (i == 1 ? T_RAW1 : (i == 2 ? T_RAW2 : T_RAW3)); }

  // (No mask of this solver lives through the attempt: "marginal" is read back from T_MARK, and the store tells a ray -
  //  status PRIMAL_INFEASIBLE of a non-empty box - and a bare least-violation point - SOLVED_INACCURATE that no attempt
  //  certified - from the status and the `polished` flag.)

  // ================================================================================ setup: the FULL problem's scaling
  MPMPC_HD void setup_full(const R* fields, int B, const I& inst, const I& k, int N_, const SolverParams& st) {
    RS::context(B, inst, k, N_);
    auto fld = [&](int f, double dflt) { return sel(vx, fields[f], R(dflt)); };
    const R zero(0.0), onec(1.0);
    // ---- the separated parts, in the UNSCALED problem (as ReducedSolver::setup_problem)
    const R lo_v = max_(fld(F_LO + 3, -INFTY), R(-INFTY)), hi_v = min_(fld(F_HI + 3, INFTY), R(INFTY));
    const R p_v = fld(F_P + 3, 1.0), q_v = fld(F_Q + 3, 0.0);
    R v = -q_v / p_v;
    v = sel((hi_v < R(INF_BOUND)) & (v > hi_v), hi_v, v);
    v = sel((lo_v > R(-INF_BOUND)) & (v < lo_v), lo_v, v);
    v = sel(vu, v, zero);
    L::cold_put(C_V, v);
    L::cold_put(C_LAMV, sel(vu, -fma_(p_v, v, q_v), zero));
    L::cold_put(C_A20, fld(F_A20, 0.0));
    L::cold_put(C_BV, fld(F_B20, 0.0) * v);
    L::cold_put(C_BEQ2, fld(F_BEQ + 2, 0.0));
    const R lo_e = max_(fld(F_LO + 0, -INFTY), R(-INFTY)), hi_e = min_(fld(F_HI + 0, INFTY), R(INFTY));
    const R lo_k = max_(fld(F_LO + 4, -INFTY), R(-INFTY)), hi_k = min_(fld(F_HI + 4, INFTY), R(INFTY));
    {
      R gap = max_(max_(sel(vx, lo_e - hi_e, zero), sel(vu, lo_k - hi_k, zero)), sel(vu, lo_v - hi_v, zero));
      gap = L::gmax(gap);
      empty = live & (gap > zero);
      solvable = live & !empty;
      pri_res = gap;          // (kept in a register here: the slot C_GAP serves as T_MARK)
      L::cold_put(T_MARK, zero);
    }
    {
      auto fin = [&](const R& bnd) { return sel(abs_(bnd) < R(INF_BOUND), abs_(bnd), zero); };
      R m = sel(vx, max_(fin(lo_e), fin(hi_e)), zero);
      m = max_(m, sel(vu, max_(max_(fin(lo_v), fin(hi_v)), max_(abs_(v), max_(fin(lo_k), fin(hi_k)))), zero));
      L::cold_put(T_NAX, L::gmax(m));
      // (Solver::p1_band: the violation beyond which phase 1 may leave at the first valid ray; from the finite BOUNDS alone,
      //  as the general kernel forms it)
      R mb = sel(vx, max_(fin(lo_e), fin(hi_e)), zero);
      mb = max_(mb, sel(vu, max_(max_(fin(lo_v), fin(hi_v)), max_(fin(lo_k), fin(hi_k))), zero));
      L::cold_put(T_BAND, st.phase1_accept ? R(st.phase1_band) * fma_(R(st.eps_rel), L::gmax(mb), R(st.eps_abs)) : zero);
    }
    // ---- the FULL problem (5 entries, 3 equality rows per stage) as Solver::load sets it up, through Solver::ruiz
    const R ds = fld(F_DS, 0.0), one = sel(vu, onec, zero);
    S::valid[0] = S::valid[1] = S::valid[2] = vx;
    S::valid[3] = S::valid[4] = vu;
    a[0] = one; a[1] = ds; a[2] = fld(F_A10, 0.0); a[3] = one; a[4] = fld(F_A20, 0.0); a[5] = one;
    b[0] = ds; b[1] = fld(F_B20, 0.0);
    MPMPC_UNROLL
    for (int i = 0; i < 3; ++i) { mI[i] = R(-1.0); S::Eeq[i] = onec; }
    MPMPC_UNROLL
    for (int j = 0; j < 5; ++j) {
      S::q[j] = fld(F_Q + j, 0.0);
      S::p[j] = fld(F_P + j, 1.0);
      S::g[j] = onec; S::D[j] = onec; S::Eb[j] = onec;
    }
    S::c = onec;
    const int passes = st.early_scaling > 0 && st.early_scaling < st.scaling ? st.early_scaling : st.scaling;
    S::ruiz(passes, live);
    // ---- what the reduced problem keeps of it
    constexpr int J5[3] = {0, 1, 4};
    MPMPC_UNROLL
    for (int e = 0; e < 3; ++e) {
      P3[e] = S::p[J5[e]]; Q3[e] = S::q[J5[e]];
      L::cold_put(C_D + e, S::D[J5[e]]);
      L::cold_put(C_G + e, S::g[J5[e]]);
      x3[e] = zero;
    }
    L::cold_put(T_RAW0, lo_e); L::cold_put(T_RAW1, hi_e); L::cold_put(T_RAW2, lo_k); L::cold_put(T_RAW3, hi_k);
    leq[0] = S::Eeq[0] * fld(F_BEQ + 0, 0.0);
    leq[1] = S::Eeq[1] * fld(F_BEQ + 1, 0.0);
    L::cold_put(C_E, S::Eeq[0]); L::cold_put(C_E + 1, S::Eeq[1]);
    L::cold_put(C_C, S::c);
    // box in the scaled variable space, formed like Solver::make_box does: (Eb lo_raw) / g
    L::cold_put(K_LO0, (S::Eb[0] * lo_e) / S::g[0]); L::cold_put(K_HI0, (S::Eb[0] * hi_e) / S::g[0]);
    L::cold_put(K_LO2, (S::Eb[4] * lo_k) / S::g[4]); L::cold_put(K_HI2, (S::Eb[4] * hi_k) / S::g[4]);
    // |P x + q| of the speed entry at its closed-form value (Solver::polish: the centred start's multipliers are sized by the
    // dual residual of ALL entries of the start point)
    L::cold_put(T_RD0, L::gmax(sel(vu, abs_(fma_(S::p[3], v / S::D[3], S::q[3])), zero)));
    L::fence();
  }

  // largest violation of the box and of the dynamics rows by the point xs (scaled), in the unscaled problem (Solver::certificate's `prim`)
  MPMPC_HD R violation(const R xs[3]) const {
    const R zero(0.0);
    R Ax[2], pv(0.0);
    this->template Aeq_mul_t<RS::LAY_RED>(xs, Ax);
    MPMPC_UNROLL
    for (int i = 0; i < 2; ++i) pv = max_(pv, sel(vx, abs_((Ax[i] - leq[i]) / L::cold_get(C_E + i)), zero));
    MPMPC_UNROLL
    for (int b2 = 0; b2 < 2; ++b2) {
      const int e = b2 == 0 ? 0 : 2;
      const R xu = L::cold_get(C_D + e) * xs[e];
      pv = max_(pv, sel(val[e], max_(max_(L::cold_get(t_raw(2 * b2)) - xu, xu - L::cold_get(t_raw(2 * b2 + 1))), zero), zero));
    }
    return L::gmax(pv);
  }

  // ================================================================================ the tail solve
  // base_ipm: interior-point iterations the first launch spent on the instance
  MPMPC_HD void run(const R* fields, int B, const I& inst, const I& k, int N_, const SolverParams& st, const I& base_ipm) {
    MPMPC_TICK_BEGIN(0);
    setup_full(fields, B, inst, k, N_, st);
    MPMPC_TICK_END(0);
    const R zero(0.0), one(1.0);
    // (an empty box - never in a tail list, the first launch reports it - is infeasible with a zero ray and the width of the
    //  gap as its violation; nothing below touches such an instance)
    status = seli(empty, I(MPMPC_PRIMAL_INFEASIBLE), I(MPMPC_UNSOLVED));
    ipm_iters = base_ipm;
    polished = I(0);
    pri_res = sel(empty, pri_res, zero);
    dua_res = zero;
    this->act_bits = I(0);
    const Mk todo = solvable;
    if (L::wany(todo)) {
      Box3 b3;
      RS::make_box3(b3);
      // ---- phase 1 (Solver::phase1):  min 1/2 |w|^2  s.t. the dynamics rows and pinned entries as they are,
      //      lo <= x_j + w_j <= hi  on every other entry with a finite side.  Cold start in row space (x = 0, slacks
      //      max(distance to the bound, theta), multipliers theta), expressed in the variable space of the iteration.
      R xs[3], lam[3], nus[2];
      {
        const R theta(st.phase1_theta);
        R sl[3], su[3], zl[3], zu[3];
        MPMPC_UNROLL
        for (int e = 0; e < 3; ++e) {
          const R ge = L::cold_get(C_G + e), ig = one / ge;
          sl[e] = sel(b3.Lm[e], max_(-(b3.lo[e] * ge), theta) * ig, one);
          su[e] = sel(b3.Um[e], max_(b3.hi[e] * ge, theta) * ig, one);
          zl[e] = sel(b3.Lm[e], theta * ge, zero);
          zu[e] = sel(b3.Um[e], theta * ge, zero);
        }
        BoxI bi;
        IpmI si;
        Mk vm[EI];
        if constexpr (RS::kSplit) {
          bU[0] = sel(sU, L::from_lower(b[0]), zero);
          bU[1] = zero;
          vm[0] = val3[0]; vm[1] = val3[2];
        } else {
          vm[0] = val[0]; vm[1] = val[1]; vm[2] = val[2];
        }
        RS::to_ip(b3.lo, bi.lo); RS::to_ip(b3.hi, bi.hi);
        RS::mask_to_ip(b3.Lm, bi.Lm); RS::mask_to_ip(b3.Um, bi.Um); RS::mask_to_ip(b3.pin, bi.pin);
        const R z3[3] = {zero, zero, zero};
        RS::to_ip(z3, si.x); RS::to_ip(sl, si.sl, 1.0); RS::to_ip(su, si.su, 1.0); RS::to_ip(zl, si.zl); RS::to_ip(zu, si.zu); RS::to_ip(z3, si.pi);
        si.nu[0] = si.nu[1] = zero;
        MPMPC_UNROLL
        for (int e = 0; e < EI; ++e) si.tL[e] = si.tU[e] = L::mfalse();
        S::p1_converged = L::mfalse();
        MPMPC_TICK_BEGIN(9);
        // (two digits further than the polish: Solver::phase1)
        const double tol1 = st.ipm_tol * 1e-2 < 1e-11 ? st.ipm_tol * 1e-2 : 1e-11;
        if constexpr (RS::kSplit) {
          // (the weights of phase 1's metric, Solver::phase1: squares of the box rows' scaled entries)
          R om3[3], omI[EI];
          MPMPC_UNROLL
          for (int e = 0; e < 3; ++e) { const R ge = L::cold_get(C_G + e); om3[e] = ge * ge; }
          RS::to_ip(om3, omI, 1.0);
          // ... and D / om: the iterate's w in unscaled units (Solver::ipm reads it in place of the cost vector)
          R d3[3], dI[EI];
          MPMPC_UNROLL
          for (int e = 0; e < 3; ++e) d3[e] = L::cold_get(C_D + e) / om3[e];
          RS::to_ip(d3, dI, 1.0);
          S::p1_band = L::cold_get(T_BAND);
          this->template ipm<RS::LAY_IP, true>(bi, si, omI, dI, vm, st, tol1, todo);
        } else {
          // the packed interior point reads the equality offsets from LDS; the cost - not read in phase 1 - waits in its own
          // slots there meanwhile instead of in twelve registers
          MPMPC_UNROLL
          for (int e = 0; e < 3; ++e) { L::cold_put(RS::K_PP + e, P3[e]); L::cold_put(RS::K_QQ + e, Q3[e]); }
          L::cold_put(RS::K_LEQ, leq[0]); L::cold_put(RS::K_LEQ + 1, leq[1]);
          L::fence();
          RS::template ipm3<true>(bi, si, st, tol1, todo, st.phase1_accept ? (int)T_BAND : -1);
          L::fence();
          MPMPC_UNROLL
          for (int e = 0; e < 3; ++e) { P3[e] = L::cold_get(RS::K_PP + e); Q3[e] = L::cold_get(RS::K_QQ + e); }
          leq[0] = L::cold_get(RS::K_LEQ); leq[1] = L::cold_get(RS::K_LEQ + 1);
          b3.lo[0] = L::cold_get(K_LO0); b3.hi[0] = L::cold_get(K_HI0); b3.lo[2] = L::cold_get(K_LO2); b3.hi[2] = L::cold_get(K_HI2);
        }
        MPMPC_TICK_END(9);
        L::fence();
        // point and ray (lam = zu - zl + pi in variable space)
        R l2[EI];
        MPMPC_UNROLL
        for (int e = 0; e < EI; ++e) l2[e] = si.zu[e] - si.zl[e] + si.pi[e];
        RS::from_ip(l2, lam);
        RS::from_ip(si.x, xs);
        nus[0] = si.nu[0]; nus[1] = si.nu[1];
        MPMPC_UNROLL
        for (int e = 0; e < 3; ++e) { xs[e] = sel(val[e], xs[e], zero); lam[e] = sel(val[e], lam[e], zero); }
        lam[1] = zero;          // (e_psi has no box row)
        nus[0] = sel(vx, nus[0], zero); nus[1] = sel(vx, nus[1], zero);
      }
      // ---- the verdict.  OSQP's is_primal_infeasible() on the ray, unscaled norms (Solver::farkas_values): |E dy|_inf, the
      //      support u'max(dy, 0) + l'min(dy, 0), |inv(D) A'dy|_inf; the speed boxes and the time rows carry no ray entry.
      R f_nrm(0.0), f_lhs(0.0), f_m(0.0);
      {
        R At[3];
        this->template AeqT_mul_t<RS::LAY_RED>(nus, At);
        MPMPC_UNROLL
        for (int i = 0; i < 2; ++i) {
          f_nrm = max_(f_nrm, sel(vx, abs_(L::cold_get(C_E + i) * nus[i]), zero));
          f_lhs = f_lhs + sel(vx, leq[i] * nus[i], zero);
        }
        MPMPC_UNROLL
        for (int e = 0; e < 3; ++e) {
          const R De = L::cold_get(C_D + e);
          if (e != 1) {
            // (row space: the ray entry is lam / g, its bounds g lo, g hi; an infinite side takes no part)
            const Mk fl = b3.lo[e] > R(-RS::BOX_INF), fu = b3.hi[e] < R(RS::BOX_INF);
            R d = lam[e];
            d = sel(!fu & !fl, zero, sel(!fu, min_(d, zero), sel(!fl, max_(d, zero), d)));
            f_nrm = max_(f_nrm, sel(val[e], abs_(d / De), zero));
            f_lhs = f_lhs + sel(val[e], sel(fu, b3.hi[e], zero) * max_(d, zero) + sel(fl, b3.lo[e], zero) * min_(d, zero), zero);
            f_m = max_(f_m, sel(val[e], abs_((d + At[e]) / De), zero));
          } else {
            f_m = max_(f_m, sel(val[e], abs_(At[e] / De), zero));
          }
        }
        f_nrm = L::gmax(f_nrm); f_lhs = L::gsum(f_lhs); f_m = L::gmax(f_m);
      }
      const R prim = violation(xs);
      const R eps1(st.phase1_eps);
      const Mk certA = (f_nrm > eps1) & (f_lhs < -eps1 * f_nrm) & (f_m < eps1 * f_nrm);
      const Mk certB = S::p1_converged & (prim > R(st.cert_tol)) & (f_nrm > zero) & (f_lhs < R(-100.0) * f_m) & (f_lhs < zero);
      Mk cert = todo & (certA | certB);
      // ---- marginally infeasible (phase1_accept): below the primal tolerance at which the reference's own OSQP call returns
      //      a plan.  max(|Ax|, |z|): the largest entry of the least-violation point or finite bound of the instance.
      Mk marginal = L::mfalse();
      if (st.phase1_accept && L::wany(cert)) {
        const R e_y = L::cold_get(C_D) * xs[0], e_psi = L::cold_get(C_D + 1) * xs[1], kap = L::cold_get(C_D + 2) * xs[2];
        const R drive = fma_(L::cold_get(C_A20), e_y, L::cold_get(C_BV)), beq2 = L::cold_get(C_BEQ2);
        const R t0 = -beq2;
        R t = t0;
        t = L::gscan(sel(first, t0, sel(vx, L::up(drive) - beq2, zero)));
        R m = sel(vx, max_(max_(abs_(e_y), abs_(e_psi)), abs_(t)), zero);
        m = max_(m, sel(vu, abs_(kap), zero));
        const R nAx = max_(L::gmax(m), L::cold_get(T_NAX));
        marginal = cert & !(prim > fma_(R(st.eps_rel), nAx, R(st.eps_abs)));
        cert = cert & !marginal;
        if (L::wany(marginal)) {
          // every box the least-violation point leaves is widened to 1.5 times that violation
          MPMPC_UNROLL
          for (int b2 = 0; b2 < 2; ++b2) {
            const int e = b2 == 0 ? 0 : 2;
            const R De = L::cold_get(C_D + e), ge = L::cold_get(C_G + e);
            const R xu = De * xs[e], lo0 = L::cold_get(t_raw(2 * b2)), hi0 = L::cold_get(t_raw(2 * b2 + 1));
            const R wl = sel(val[e] & (lo0 > R(-INF_BOUND)), max_(lo0 - xu, zero), zero);
            const R wh = sel(val[e] & (hi0 < R(INF_BOUND)), max_(xu - hi0, zero), zero);
            const R lo1 = fma_(R(-1.5), wl, lo0), hi1 = fma_(R(1.5), wh, hi0);
            // (Eb = g / D: the general kernel keeps Eb itself; the quotient differs from it in the last bit at most)
            const R Ebe = ge / De;
            // (merged: a partner instance of the wave keeps its boxes to the bit)
            L::cold_put(e == 0 ? K_LO0 : K_LO2, sel(marginal, (Ebe * lo1) / ge, L::cold_get(e == 0 ? K_LO0 : K_LO2)));
            L::cold_put(e == 0 ? K_HI0 : K_HI2, sel(marginal, (Ebe * hi1) / ge, L::cold_get(e == 0 ? K_HI0 : K_HI2)));
          }
          L::cold_put(T_MARK, sel(marginal, prim, zero));
          L::fence();
          RS::make_box3(b3);
        }
      }
      // ---- feasible to tolerance (phase 1 converged, its point violates nothing) or marginal: one more attempt of the
      //      certified polish, from phase 1's point - inside every box it can be inside of, well centred
      const Mk retry = todo & !cert & ((S::p1_converged & !(prim > R(st.cert_tol))) | marginal);
      // the box-row scalings of the attempt's start: the slots they live in take the verdict's point now
      R g3[3];
      MPMPC_UNROLL
      for (int e = 0; e < 3; ++e) g3[e] = L::cold_get(C_G + e);
      const R rd0v = L::cold_get(T_RD0);          // (its slot is one of the point's as well)
      L::fence();
      // ---- the point of a verdict: least-violation point, for "infeasible" with the ray; for a marginal instance with zero
      //      multipliers - what it ends with if the attempt cannot certify the optimum over the relaxed boxes
      {
        const Mk pre = cert | marginal;
        MPMPC_UNROLL
        for (int e = 0; e < 3; ++e) {
          L::cold_put(C_XS + e, sel(pre, xs[e], L::cold_get(C_XS + e)));
          L::cold_put(C_LAM + e, sel(pre, sel(cert, lam[e], zero), L::cold_get(C_LAM + e)));
        }
        L::cold_put(C_NUS, sel(pre, sel(cert, nus[0], zero), L::cold_get(C_NUS)));
        L::cold_put(C_NUS + 1, sel(pre, sel(cert, nus[1], zero), L::cold_get(C_NUS + 1)));
        L::fence();
        pri_res = sel(cert, prim, pri_res);
        dua_res = sel(cert, zero, dua_res);
        status = seli(cert, I(MPMPC_PRIMAL_INFEASIBLE), status);
      }
      if (L::wany(retry)) {
        MPMPC_UNROLL
        for (int e = 0; e < 3; ++e) x3[e] = sel(retry, xs[e], zero);
        // (merge mode: what the attempt certifies replaces the point in the slots, what it does not leaves it there)
        (void)RS::template attempts<false>(b3, st, st.ipm_max_iter, retry, true, &rd0v, g3);
        // a marginal instance ends here: the optimum over the relaxed boxes, or - if the attempt could not certify that -
        // phase 1's least-violation point itself; either way a usable, inaccurate plan
        L::fence();
        const R viol = L::gmax(L::cold_get(T_MARK));        // > 0: the instance was marginal
        const Mk mg = (viol > zero) & ((status == MPMPC_SOLVED) | (status == MPMPC_UNSOLVED));
        status = seli(mg, I(MPMPC_SOLVED_INACCURATE), status);
        pri_res = sel(mg, viol, pri_res);
      }
    }
    iters = I(st.early_polish);      // (the ADMM iteration of the first launch's early attempt: Solver::run, mode 2)
  }

  MPMPC_HD void store(const I& inst, const I& k, double wheelbase, double* z, double* u0, int* st_out, int* it_out,
                      double* resid, double* y) const {
    const Mk ray = (status == MPMPC_PRIMAL_INFEASIBLE) & !empty;
    const Mk point = live & ((status == MPMPC_SOLVED) | (status == MPMPC_SOLVED_INACCURATE) | ray);
    const Mk no_lamv = ray | ((status == MPMPC_SOLVED_INACCURATE) & (polished != 1));
    RS::template store<(L::per_wave > 1)>(inst, k, wheelbase, z, u0, st_out, it_out, resid, y, nullptr, 0, &point, &no_lamv);
  }
};

}  // namespace mpmpc
