// PART OF struct mpmpc::Solver (mpmpc_core.hpp) - phase 1: is the instance infeasible?  (least violation in OSQP's metric, Farkas ray, marginal verdict).
// This file is included INSIDE the class body; it is not a header of its own.
#ifndef MPMPC_SOLVER_BODY
#error "include mpmpc_core.hpp"
#endif
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
