// PART OF struct mpmpc::Solver (mpmpc_core.hpp) - the regularised Mehrotra interior point (and, SOFT, phase 1's least-violation iteration).
// This file is included INSIDE the class body; it is not a header of its own.
#ifndef MPMPC_SOLVER_BODY
#error "include mpmpc_core.hpp"
#endif
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
          const R th = sel(bx.Lm[j], s.zl[j] * isl[j], zero) + sel(bx.Um[j], s.zu[j] * isu[j], zero);
          kap[j] = rcp_(fma_(th, iom[j], one));
          return fma_(kap[j], th, reg) + sel(bx.pin[j], ireg, zero);          // k th (= om (1 - k): the weights themselves are not needed)
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
