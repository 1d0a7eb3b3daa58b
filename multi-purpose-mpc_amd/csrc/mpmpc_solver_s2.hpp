// PART OF struct mpmpc::Solver (mpmpc_core.hpp) - the reduced problem's block-tridiagonal Cholesky for the layout with TWO
// STAGES PER LANE (lane_pair.hpp: a lane holds the stages 2p, 2p + 1 of its instance; an instance of up to 32 stages is one
// chain of 16 lanes = one DPP row, four instances per wavefront; one of 65 .. 128 stages a chain of 64 lanes, one wavefront).
// This file is included INSIDE the class body; it is not a header of its own.
#ifndef MPMPC_SOLVER_BODY
#error "include mpmpc_core.hpp"
#endif
  // The same cyclic reduction in Cholesky form as factor_cr2 - a symmetric permutation of the stages, blocks stay 2 x 2 - with
  // ONE MORE LEVEL in front that costs no lane exchange to speak of and leaves no lane idle:
  //   level H   every lane eliminates its EVEN stage e = 2p against a = 2p - 1 (the odd stage of the lane below: one shifted
  //             operand, S_ea) and b = 2p + 1 (its own odd stage):  L_e L_e' = D_e,  Ua = inv(L_e) S_ea,  Ub = inv(L_e) S_eb,
  //             D_b -= Ub'Ub in the lane,  D_a -= Ua'Ua one lane down (three shifted sums),  S_ba = -Ub'Ua in the lane: the
  //             odd stages are a block-tridiagonal system of their own, one stage per lane;
  //   levels D = 1, 2, 4, 8 on those survivors, on SCALAR values of the lanes underneath (L::L1) - what a wavefront of the
  //             one-stage layout does for two instances it does here for four;
  //   chains of FOUR rows (64 lanes per instance: horizons 64 .. 127 in one wavefront): the survivors of the rows are eliminated one
  //             after the other exactly as in factor_cr2's kCR64 part (step r: the survivor X of row r against the survivor Y of
  //             row r + 1 and that row's lanes 0, 1, 3, 7, whose lower neighbour X was when they were eliminated);
  //   the last position (the last stage, or the identity-like block behind a shorter horizon) is factored last.
  // Per lane: component 0 of Li / Gin / Gout = inv(L_e), Ua, Ub of the even stage, component 1 = the survivor's blocks of the
  // level that eliminated it.  Horizons below 31 leave identity-like blocks with zero couplings at the end of the chain; they
  // factor harmlessly (as in the one-stage layout).
  // ---- one cross-lane level on scalars (cr_level, for a single chain of 16 lanes: position 7 has nobody below it)
  template <class LL, int D, class T>
  MPMPC_HD static void s2_level(T Dg[3], T Cm[4], T Li1[3], T Gin1[4], T Gout1[4]) {
    const auto E = LL::template cr_elim<D>();
    const T zero(0.0);
    T i00 = rsqrt_(Dg[0]);
    const T l10 = Dg[1] * i00;
    T i11 = rsqrt_(fma_(-l10, l10, Dg[2]));
    T i10 = -(l10 * i00) * i11;
    i00 = sel(E, i00, zero); i10 = sel(E, i10, zero); i11 = sel(E, i11, zero);
    Li1[0] = Li1[0] + i00; Li1[1] = Li1[1] + i10; Li1[2] = Li1[2] + i11;
    T gb[4];
    {
      T Cb[4], Ub[4];
      MPMPC_UNROLL
      for (int i = 0; i < 4; ++i) Cb[i] = LL::template rshl<D>(Cm[i]);
      Ub[0] = i00 * Cb[0]; Ub[1] = i00 * Cb[2];
      Ub[2] = fma_(i11, Cb[1], i10 * Cb[0]); Ub[3] = fma_(i11, Cb[3], i10 * Cb[2]);
      MPMPC_UNROLL
      for (int i = 0; i < 4; ++i) { Gout1[i] = Gout1[i] + Ub[i]; gb[i] = LL::template rshr<D>(Ub[i]); }
    }
    Dg[0] = fma_(-gb[2], gb[2], fma_(-gb[0], gb[0], Dg[0]));
    Dg[1] = fma_(-gb[3], gb[2], fma_(-gb[1], gb[0], Dg[1]));
    Dg[2] = fma_(-gb[3], gb[3], fma_(-gb[1], gb[1], Dg[2]));
    // (the last level, D = 8, eliminates position 7: nobody of its row is below it - in a one-row chain its coupling is an exact
    //  zero, in a chain of several rows the survivor of the row below is, and takes its update in s2_rows_factor)
    if constexpr (D != 8 || LL::group > 16) {
      T ga[4];
      {
        T Ua[4];
        Ua[0] = i00 * Cm[0]; Ua[1] = i00 * Cm[1];
        Ua[2] = fma_(i11, Cm[2], i10 * Cm[0]); Ua[3] = fma_(i11, Cm[3], i10 * Cm[1]);
        MPMPC_UNROLL
        for (int i = 0; i < 4; ++i) Gin1[i] = Gin1[i] + Ua[i];
        // to the lower neighbour: D_a -= Ua'Ua (the three sums are formed where Ua is and travel, not the four entries)
        if constexpr (D != 8) {
          const T w0 = fma_(Ua[2], Ua[2], Ua[0] * Ua[0]), w1 = fma_(Ua[3], Ua[2], Ua[1] * Ua[0]), w2 = fma_(Ua[3], Ua[3], Ua[1] * Ua[1]);
          Dg[0] = Dg[0] - LL::template rshl<D>(w0); Dg[1] = Dg[1] - LL::template rshl<D>(w1); Dg[2] = Dg[2] - LL::template rshl<D>(w2);
        }
        MPMPC_UNROLL
        for (int i = 0; i < 4; ++i) ga[i] = LL::template rshr<D>(Ua[i]);
      }
      Cm[0] = sel(E, Cm[0], -fma_(gb[2], ga[2], gb[0] * ga[0]));
      Cm[1] = sel(E, Cm[1], -fma_(gb[2], ga[3], gb[0] * ga[1]));
      Cm[2] = sel(E, Cm[2], -fma_(gb[3], ga[2], gb[1] * ga[0]));
      Cm[3] = sel(E, Cm[3], -fma_(gb[3], ga[3], gb[1] * ga[1]));
    }
  }
  // ---- chains of several rows: the survivors in turn (factor_cr2, kCR64 - on scalars of the lanes underneath)
  template <class LL, class T>
  MPMPC_HD static void s2_rows_factor(T Dg[3], const T Cm[4], T Li1[3], const T Gin1[4], T Gout1[4]) {
    constexpr int ROWS = LL::group / 16;
    const T zero(0.0);
    MPMPC_UNROLL
    for (int r = 0; r < ROWS - 1; ++r) {
      const auto spec = LL::cr64_special(r), isX = LL::cr64_x(r), isY = LL::cr64_x(r + 1);
      T w0 = sel(spec, fma_(Gin1[2], Gin1[2], Gin1[0] * Gin1[0]), zero), w1 = sel(spec, fma_(Gin1[3], Gin1[2], Gin1[1] * Gin1[0]), zero),
        w2 = sel(spec, fma_(Gin1[3], Gin1[3], Gin1[1] * Gin1[1]), zero);
      w0 = w0 + LL::template rshl<1>(w0); w1 = w1 + LL::template rshl<1>(w1); w2 = w2 + LL::template rshl<1>(w2);
      w0 = w0 + LL::template rshl<3>(w0); w1 = w1 + LL::template rshl<3>(w1); w2 = w2 + LL::template rshl<3>(w2);
      w0 = w0 + LL::template rshl<7>(w0); w1 = w1 + LL::template rshl<7>(w1); w2 = w2 + LL::template rshl<7>(w2);
      {
        const T wv[3] = {w0, w1, w2};
        T wd[3];
        LL::template cr_down<3>(r, wv, wd);
        Dg[0] = Dg[0] - sel(isX, wd[0], zero); Dg[1] = Dg[1] - sel(isX, wd[1], zero); Dg[2] = Dg[2] - sel(isX, wd[2], zero);
      }
      T i00 = rsqrt_(Dg[0]);
      const T l10 = Dg[1] * i00;
      T i11 = rsqrt_(fma_(-l10, l10, Dg[2]));
      T i10 = -(l10 * i00) * i11;
      i00 = sel(isX, i00, zero); i10 = sel(isX, i10, zero); i11 = sel(isX, i11, zero);
      Li1[0] = Li1[0] + i00; Li1[1] = Li1[1] + i10; Li1[2] = Li1[2] + i11;
      T Cb[4], Ub[4], gb[4];
      LL::template cr_pull<4>(r, Cm, Cb);
      Ub[0] = i00 * Cb[0]; Ub[1] = i00 * Cb[2];
      Ub[2] = fma_(i11, Cb[1], i10 * Cb[0]); Ub[3] = fma_(i11, Cb[3], i10 * Cb[2]);
      LL::template cr_push<4>(r, Ub, gb);
      MPMPC_UNROLL
      for (int i = 0; i < 4; ++i) { Gout1[i] = Gout1[i] + Ub[i]; gb[i] = sel(isY, gb[i], zero); }
      Dg[0] = fma_(-gb[2], gb[2], fma_(-gb[0], gb[0], Dg[0]));
      Dg[1] = fma_(-gb[3], gb[2], fma_(-gb[1], gb[0], Dg[1]));
      Dg[2] = fma_(-gb[3], gb[3], fma_(-gb[1], gb[1], Dg[2]));
    }
  }
  template <class LL, class T>
  MPMPC_HD static void s2_rows_forward(T& b0, T& b1, T& y0, T& y1, const T Li1[3], const T Gin1[4], const T Gout1[4]) {
    constexpr int ROWS = LL::group / 16;
    const T zero(0.0);
    MPMPC_UNROLL
    for (int r = 0; r < ROWS - 1; ++r) {
      const auto spec = LL::cr64_special(r), isX = LL::cr64_x(r), isY = LL::cr64_x(r + 1);
      T c0 = sel(spec, fma_(Gin1[2], y1, Gin1[0] * y0), zero), c1 = sel(spec, fma_(Gin1[3], y1, Gin1[1] * y0), zero);
      c0 = c0 + LL::template rshl<1>(c0); c1 = c1 + LL::template rshl<1>(c1);
      c0 = c0 + LL::template rshl<3>(c0); c1 = c1 + LL::template rshl<3>(c1);
      c0 = c0 + LL::template rshl<7>(c0); c1 = c1 + LL::template rshl<7>(c1);
      const T cv[2] = {c0, c1};
      T cd[2];
      LL::template cr_down<2>(r, cv, cd);
      const T bx0 = b0 - cd[0], bx1 = b1 - cd[1];
      const T yx0 = sel(isX, Li1[0] * bx0, zero), yx1 = sel(isX, fma_(Li1[2], bx1, Li1[1] * bx0), zero);
      y0 = y0 + yx0; y1 = y1 + yx1;
      const T pv[2] = {fma_(Gout1[2], yx1, Gout1[0] * yx0), fma_(Gout1[3], yx1, Gout1[1] * yx0)};
      T pp_[2];
      LL::template cr_push<2>(r, pv, pp_);
      b0 = b0 - sel(isY, pp_[0], zero); b1 = b1 - sel(isY, pp_[1], zero);
    }
  }
  template <class LL, class T>
  MPMPC_HD static void s2_rows_backward(T& y0, T& y1, T& n0, T& n1, const T Li1[3], const T Gin1[4], const T Gout1[4]) {
    constexpr int ROWS = LL::group / 16;
    const T zero(0.0);
    MPMPC_UNROLL
    for (int r = ROWS - 2; r >= 0; --r) {
      const auto spec = LL::cr64_special(r), isX = LL::cr64_x(r);
      const T nv_[2] = {n0, n1};
      T cn_[2];
      LL::template cr_pull<2>(r, nv_, cn_);
      const T c0 = cn_[0], c1 = cn_[1];
      const T r0 = y0 - fma_(Gout1[1], c1, Gout1[0] * c0), r1 = y1 - fma_(Gout1[3], c1, Gout1[2] * c0);
      n0 = sel(isX, fma_(Li1[1], r1, Li1[0] * r0), n0);
      n1 = sel(isX, Li1[2] * r1, n1);
      const T nx_[2] = {n0, n1};
      T xb_[2];
      LL::template cr_bcast<2>(r, nx_, xb_);
      y0 = y0 - sel(spec, fma_(Gin1[1], xb_[1], Gin1[0] * xb_[0]), zero);
      y1 = y1 - sel(spec, fma_(Gin1[3], xb_[1], Gin1[2] * xb_[0]), zero);
    }
  }
  MPMPC_HD void factor_core2_s2(const R hx[2], const R& wb, const R& r) {
    using LL = typename L::L1;
    using T = typename LL::real;
    const R* h = hx;
    R W[3], Tc[4], Dg[3];
    {
      R a0h = a[0] * h[0], a2h = a[2] * h[0], a1h = a[1] * h[1], a3h = a[3] * h[1];
      W[0] = fma_(a[1], a1h, a[0] * a0h);
      W[1] = fma_(a[3], a1h, a[2] * a0h);
      W[2] = fma_(a[3], a3h, a[2] * a2h) + wb;
      Tc[0] = a0h * mI[0]; Tc[1] = a1h * mI[1];          // S_{s+1, s}: the coupling a stage hands up the chain (zero from stage N on)
      Tc[2] = a2h * mI[0]; Tc[3] = a3h * mI[1];
    }
    up_n<3>(W, Dg);
    Dg[0] = Dg[0] + fma_(mI[0] * mI[0], h[0], r);
    Dg[2] = Dg[2] + fma_(mI[1] * mI[1], h[1], r);
    T D1[3], C1[4], Li1[3], Gin1[4], Gout1[4];
    {
      MPMPC_SERIAL_BEGIN();          // (census: level H eliminates one stage per lane - all of its work is useful)
      // ---- level H, inside the lanes
      const T i00 = rsqrt_(Dg[0].v[0]);
      const T l10 = Dg[1].v[0] * i00;
      const T i11 = rsqrt_(fma_(-l10, l10, Dg[2].v[0]));
      const T i10 = -(l10 * i00) * i11;
      T Cm[4], Ua[4], Ub[4];
      {
        const T t1[4] = {Tc[0].v[1], Tc[1].v[1], Tc[2].v[1], Tc[3].v[1]};
        L::template up1n<4>(t1, Cm);                                 // S_{e, a}: what the odd stage below hands up
      }
      Ub[0] = i00 * Tc[0].v[0]; Ub[1] = i00 * Tc[2].v[0];                         // inv(L_e) S_be'
      Ub[2] = fma_(i11, Tc[1].v[0], i10 * Tc[0].v[0]); Ub[3] = fma_(i11, Tc[3].v[0], i10 * Tc[2].v[0]);
      Ua[0] = i00 * Cm[0]; Ua[1] = i00 * Cm[1];
      Ua[2] = fma_(i11, Cm[2], i10 * Cm[0]); Ua[3] = fma_(i11, Cm[3], i10 * Cm[1]);
      D1[0] = fma_(-Ub[2], Ub[2], fma_(-Ub[0], Ub[0], Dg[0].v[1]));
      D1[1] = fma_(-Ub[3], Ub[2], fma_(-Ub[1], Ub[0], Dg[1].v[1]));
      D1[2] = fma_(-Ub[3], Ub[3], fma_(-Ub[1], Ub[1], Dg[2].v[1]));
      const T w0 = fma_(Ua[2], Ua[2], Ua[0] * Ua[0]), w1 = fma_(Ua[3], Ua[2], Ua[1] * Ua[0]), w2 = fma_(Ua[3], Ua[3], Ua[1] * Ua[1]);
      {
        const T wv[3] = {w0, w1, w2};
        T wd[3];
        L::template down1n<3>(wv, wd);
        D1[0] = D1[0] - wd[0]; D1[1] = D1[1] - wd[1]; D1[2] = D1[2] - wd[2];
      }
      C1[0] = -fma_(Ub[2], Ua[2], Ub[0] * Ua[0]);
      C1[1] = -fma_(Ub[2], Ua[3], Ub[0] * Ua[1]);
      C1[2] = -fma_(Ub[3], Ua[2], Ub[1] * Ua[0]);
      C1[3] = -fma_(Ub[3], Ua[3], Ub[1] * Ua[1]);
      Li[0].v[0] = i00; Li[1].v[0] = i10; Li[2].v[0] = i11;
      MPMPC_UNROLL
      for (int i = 0; i < 4; ++i) { Gin[i].v[0] = Ua[i]; Gout[i].v[0] = Ub[i]; }
      MPMPC_SERIAL_END(1);
    }
    {
      // ---- the survivors, one per lane: four cross-lane levels (census: each is eliminated at exactly one of them)
      MPMPC_SERIAL_BEGIN();
      const T zero(0.0);
      MPMPC_UNROLL
      for (int i = 0; i < 3; ++i) Li1[i] = zero;
      MPMPC_UNROLL
      for (int i = 0; i < 4; ++i) Gin1[i] = Gout1[i] = zero;
      s2_level<LL, 1>(D1, C1, Li1, Gin1, Gout1);
      s2_level<LL, 2>(D1, C1, Li1, Gin1, Gout1);
      s2_level<LL, 4>(D1, C1, Li1, Gin1, Gout1);
      s2_level<LL, 8>(D1, C1, Li1, Gin1, Gout1);
      MPMPC_SERIAL_END(4);
    }
    if constexpr (LL::group > 16) {
      MPMPC_SERIAL_BEGIN();
      s2_rows_factor<LL>(D1, C1, Li1, Gin1, Gout1);
      MPMPC_SERIAL_END(N + 1);
    }
    {
      MPMPC_SERIAL_BEGIN();          // (census: useful on one lane)
      const auto last = is_mid.v[1];
      const T i00 = rsqrt_(D1[0]);
      const T l10 = D1[1] * i00;
      const T i11 = rsqrt_(fma_(-l10, l10, D1[2]));
      const T i10 = -(l10 * i00) * i11;
      Li[0].v[1] = sel(last, i00, Li1[0]); Li[1].v[1] = sel(last, i10, Li1[1]); Li[2].v[1] = sel(last, i11, Li1[2]);
      MPMPC_UNROLL
      for (int i = 0; i < 4; ++i) { Gin[i].v[1] = Gin1[i]; Gout[i].v[1] = Gout1[i]; }
      MPMPC_SERIAL_END(N + 1);
    }
  }
  template <class LL, int D, class T>
  MPMPC_HD static void s2_forward(T& b0, T& b1, T& y0, T& y1, const T Li1[3], const T Gin1[4], const T Gout1[4]) {
    const auto E = LL::template cr_elim<D>();
    const T zero(0.0);
    const T t0 = Li1[0] * b0, t1 = fma_(Li1[2], b1, Li1[1] * b0);
    const T e0 = sel(E, t0, zero), e1 = sel(E, t1, zero);
    y0 = y0 + e0; y1 = y1 + e1;
    const T pb0 = fma_(Gout1[2], e1, Gout1[0] * e0), pb1 = fma_(Gout1[3], e1, Gout1[1] * e0);
    if constexpr (D == 8) {
      b0 = b0 - LL::template rshr<D>(pb0);
      b1 = b1 - LL::template rshr<D>(pb1);
    } else {
      const T pa0 = fma_(Gin1[2], e1, Gin1[0] * e0), pa1 = fma_(Gin1[3], e1, Gin1[1] * e0);
      b0 = b0 - LL::template rshl<D>(pa0) - LL::template rshr<D>(pb0);
      b1 = b1 - LL::template rshl<D>(pa1) - LL::template rshr<D>(pb1);
    }
  }
  template <class LL, int D, class T>
  MPMPC_HD static void s2_backward(const T& y0, const T& y1, T& n0, T& n1, const T Li1[3], const T Gin1[4], const T Gout1[4]) {
    const auto E = LL::template cr_elim<D>();
    const T c0 = LL::template rshl<D>(n0), c1 = LL::template rshl<D>(n1);
    [[maybe_unused]] T a0(0.0), a1(0.0);
    if constexpr (D != 8) { a0 = LL::template rshr<D>(n0); a1 = LL::template rshr<D>(n1); }
    auto level = [&] {
      T r0, r1;
      if constexpr (D == 8) {
        r0 = fma_(-Gout1[1], c1, fma_(-Gout1[0], c0, y0));
        r1 = fma_(-Gout1[3], c1, fma_(-Gout1[2], c0, y1));
      } else {
        r0 = fma_(-Gout1[1], c1, fma_(-Gout1[0], c0, fma_(-Gin1[1], a1, fma_(-Gin1[0], a0, y0))));
        r1 = fma_(-Gout1[3], c1, fma_(-Gout1[2], c0, fma_(-Gin1[3], a1, fma_(-Gin1[2], a0, y1))));
      }
      n0 = sel(E, fma_(Li1[1], r1, Li1[0] * r0), n0);
      n1 = sel(E, Li1[2] * r1, n1);
    };
    LL::when(E, level);
  }
  MPMPC_HD void s_solve_s2(const R bv[2], R nu[2]) const {
    using LL = typename L::L1;
    using T = typename LL::real;
    const R zero2(0.0);
    const T zero(0.0);
    const R b0 = sel(vx, bv[0], zero2), b1 = sel(vx, bv[1], zero2);
    T Li1[3], Gin1[4], Gout1[4];
    MPMPC_UNROLL
    for (int i = 0; i < 3; ++i) Li1[i] = Li[i].v[1];
    MPMPC_UNROLL
    for (int i = 0; i < 4; ++i) { Gin1[i] = Gin[i].v[1]; Gout1[i] = Gout[i].v[1]; }
    // ---- level H forward: y_e = inv(L_e) b_e;  b_b -= Ub'y_e (in the lane),  b_a -= Ua'y_e (one lane down)
    T c0, c1, t0, t1;
    {
      MPMPC_SERIAL_BEGIN();
      t0 = Li[0].v[0] * b0.v[0]; t1 = fma_(Li[2].v[0], b1.v[0], Li[1].v[0] * b0.v[0]);
      const T pb0 = fma_(Gout[2].v[0], t1, Gout[0].v[0] * t0), pb1 = fma_(Gout[3].v[0], t1, Gout[1].v[0] * t0);
      const T pa0 = fma_(Gin[2].v[0], t1, Gin[0].v[0] * t0), pa1 = fma_(Gin[3].v[0], t1, Gin[1].v[0] * t0);
      const T pv[2] = {pa0, pa1};
      T pd[2];
      L::template down1n<2>(pv, pd);
      c0 = b0.v[1] - pb0 - pd[0];
      c1 = b1.v[1] - pb1 - pd[1];
      MPMPC_SERIAL_END(1);
    }
    T y0(0.0), y1(0.0);
    {
      MPMPC_SERIAL_BEGIN();
      s2_forward<LL, 1>(c0, c1, y0, y1, Li1, Gin1, Gout1);
      s2_forward<LL, 2>(c0, c1, y0, y1, Li1, Gin1, Gout1);
      s2_forward<LL, 4>(c0, c1, y0, y1, Li1, Gin1, Gout1);
      s2_forward<LL, 8>(c0, c1, y0, y1, Li1, Gin1, Gout1);
      MPMPC_SERIAL_END(4);
    }
    if constexpr (LL::group > 16) {
      MPMPC_SERIAL_BEGIN();
      s2_rows_forward<LL>(c0, c1, y0, y1, Li1, Gin1, Gout1);
      MPMPC_SERIAL_END(N + 1);
    }
    T n0, n1;
    {
      // position 15: y = inv(L) b, nu = inv(L)'y
      MPMPC_SERIAL_BEGIN();
      const auto last = is_mid.v[1];
      const T ye0 = Li1[0] * c0, ye1 = fma_(Li1[2], c1, Li1[1] * c0);
      n0 = sel(last, fma_(Li1[1], ye1, Li1[0] * ye0), zero);
      n1 = sel(last, Li1[2] * ye1, zero);
      MPMPC_SERIAL_END(N + 1);
    }
    if constexpr (LL::group > 16) {
      MPMPC_SERIAL_BEGIN();
      s2_rows_backward<LL>(y0, y1, n0, n1, Li1, Gin1, Gout1);
      MPMPC_SERIAL_END(N + 1);
    }
    {
      MPMPC_SERIAL_BEGIN();
      s2_backward<LL, 8>(y0, y1, n0, n1, Li1, Gin1, Gout1);
      s2_backward<LL, 4>(y0, y1, n0, n1, Li1, Gin1, Gout1);
      s2_backward<LL, 2>(y0, y1, n0, n1, Li1, Gin1, Gout1);
      s2_backward<LL, 1>(y0, y1, n0, n1, Li1, Gin1, Gout1);
      MPMPC_SERIAL_END(4);
    }
    {
      // ---- level H backward: nu_e = inv(L_e)' (y_e - Ua nu_a - Ub nu_b),  nu_a from the lane below, nu_b in the lane
      MPMPC_SERIAL_BEGIN();
      const T nv[2] = {n0, n1};
      T na[2];
      L::template up1n<2>(nv, na);
      const T a0 = na[0], a1 = na[1];
      const T r0 = fma_(-Gout[1].v[0], n1, fma_(-Gout[0].v[0], n0, fma_(-Gin[1].v[0], a1, fma_(-Gin[0].v[0], a0, t0))));
      const T r1 = fma_(-Gout[3].v[0], n1, fma_(-Gout[2].v[0], n0, fma_(-Gin[3].v[0], a1, fma_(-Gin[2].v[0], a0, t1))));
      nu[0] = R(fma_(Li[1].v[0], r1, Li[0].v[0] * r0), n0);
      nu[1] = R(Li[2].v[0] * r1, n1);
      MPMPC_SERIAL_END(1);
    }
  }
