// PART OF struct mpmpc::Solver (mpmpc_core.hpp) - the certified polish: primal-dual active-set rounds, the reduced problem's closed-form part, the KKT
// certificate, the layouts the polish runs in, warm start and the polish driver.
// This file is included INSIDE the class body; it is not a header of its own.
#ifndef MPMPC_SOLVER_BODY
#error "include mpmpc_core.hpp"
#endif
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
