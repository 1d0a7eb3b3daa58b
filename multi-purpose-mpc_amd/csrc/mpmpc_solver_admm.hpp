// PART OF struct mpmpc::Solver (mpmpc_core.hpp) - the restated OSQP ADMM iteration: termination and infeasibility tests, rho adaptation, the starts of the
// early attempt.
// This file is included INSIDE the class body; it is not a header of its own.
#ifndef MPMPC_SOLVER_BODY
#error "include mpmpc_core.hpp"
#endif
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
