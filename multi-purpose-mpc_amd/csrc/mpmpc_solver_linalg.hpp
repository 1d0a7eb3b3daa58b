// PART OF struct mpmpc::Solver (mpmpc_core.hpp) - the linear algebra: twisted block-tridiagonal Cholesky of the Schur complement (3 x 3 and 2 x 2 blocks), its
// cyclic-reduction form, the substitution sweeps, the KKT solves and the operators of the polish's layouts.
// This file is included INSIDE the class body; it is not a header of its own.
#ifndef MPMPC_SOLVER_BODY
#error "include mpmpc_core.hpp"
#endif
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
  // ---- the two moves of a junction step: the value the END lane of the descending chain holds, on the meeting lane (zero
  // elsewhere), and back.  In chain layout that is lane 2C - 1 -> lane C - 1 of the instance: the general form goes through
  // mirror + a one-lane shift (two moves and two selects more per dword); a wavefront backend does it with ONE row / half
  // swap (v_permlane16_swap / v_permlane32_swap: LaneGpu::end_to_mid).  Same values on every lane either way.
  MPMPC_HD R end_to_mid(const R& q) const {
    const R zero(0.0);
    if constexpr (L::junction_moves) return sel(is_mid, L::end_to_mid(q), zero);
    else return sel(is_mid, L::down(L::mirror(sel(is_end, q, zero))), zero);
  }
  MPMPC_HD R mid_to_end(const R& v) const {
    const R zero(0.0);
    if constexpr (L::junction_moves) return sel(is_end, L::mid_to_end(v), zero);
    else return sel(is_end, L::mirror(L::up(sel(is_mid, v, zero))), zero);
  }
  // ... NV values at once: where an exchange passes workgroup barriers (L::batched) the values share one pass
  template <int NV> MPMPC_HD void end_to_mid_n(const R* q, R* o) const {
    if constexpr (L::batched) {
      const R zero(0.0);
      R e[NV], m[NV], d[NV];
      MPMPC_UNROLL
      for (int i = 0; i < NV; ++i) e[i] = sel(is_end, q[i], zero);
      mirror_n<NV>(e, m);
      down_n<NV>(m, d);
      MPMPC_UNROLL
      for (int i = 0; i < NV; ++i) o[i] = sel(is_mid, d[i], zero);
    } else {
      MPMPC_UNROLL
      for (int i = 0; i < NV; ++i) o[i] = end_to_mid(q[i]);
    }
  }
  template <int NV> MPMPC_HD void mid_to_end_n(const R* v, R* o) const {
    if constexpr (L::batched) {
      const R zero(0.0);
      R m[NV], u[NV];
      MPMPC_UNROLL
      for (int i = 0; i < NV; ++i) m[i] = sel(is_mid, v[i], zero);
      up_n<NV>(m, u);
      R w[NV];
      mirror_n<NV>(u, w);
      MPMPC_UNROLL
      for (int i = 0; i < NV; ++i) o[i] = sel(is_end, w[i], zero);
    } else {
      MPMPC_UNROLL
      for (int i = 0; i < NV; ++i) o[i] = mid_to_end(v[i]);
    }
  }
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
    auto fstep = [&](auto mode, bool junction) {
      R Mr[9];
      cup_n<9, decltype(mode)::value>(M, Mr);
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
        end_to_mid_n<9>(M, Mx);
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
      if constexpr (L::staged_sweeps) {
        staged_sweep<-1>(last, [&](auto mode) { fstep(mode, false); });
      } else {
        int s = 0;                                   // two steps per trip (see s_solve), then the junction step
        for (; s + 2 <= last; s += 2) { fstep(SweepMode<0>{}, false); fstep(SweepMode<0>{}, false); }
        for (; s < last; ++s) fstep(SweepMode<0>{}, false);
      }
      fstep(SweepMode<0>{}, true);
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
    auto in_step = [&](auto mode) {
      const R yv[3] = {y0, y1, y2};
      R pv[3];
      cup_n<3, decltype(mode)::value>(yv, pv);
      const R p0 = pv[0], p1 = pv[1], p2 = pv[2];
      y0 = fma_(Gin[2], p2, fma_(Gin[1], p1, fma_(Gin[0], p0, c0)));
      y1 = fma_(Gin[5], p2, fma_(Gin[4], p1, fma_(Gin[3], p0, c1)));
      y2 = fma_(Gin[8], p2, fma_(Gin[7], p1, fma_(Gin[6], p0, c2)));
    };
    {
      MPMPC_SERIAL_BEGIN();
      if constexpr (L::staged_sweeps) {
        staged_sweep<-1>(last, in_step);
      } else {
        const SweepMode<0> m0{};
        int s = 0;
        for (; s + 4 <= last; s += 4) { in_step(m0); in_step(m0); in_step(m0); in_step(m0); }
        for (; s < last; ++s) in_step(m0);
      }
      MPMPC_SERIAL_END(last);
    }
    {
      // inward junction: the end lane forms M_own y, mid takes it on top of its chain input
      MPMPC_SERIAL_BEGIN();                        // (census: useful on the two lanes of the junction only)
      R t0 = fma_(Gout[2], y2, fma_(Gout[1], y1, Gout[0] * y0));
      R t1 = fma_(Gout[5], y2, fma_(Gout[4], y1, Gout[3] * y0));
      R t2 = fma_(Gout[8], y2, fma_(Gout[7], y1, Gout[6] * y0));
      const R zero(0.0);
      { const R tv[3] = {t0, t1, t2}; R to_[3]; end_to_mid_n<3>(tv, to_); t0 = to_[0]; t1 = to_[1]; t2 = to_[2]; }
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
      R m0, m1, m2;
      { const R dv[3] = {d0, d1, d2}; R mo_[3]; mid_to_end_n<3>(dv, mo_); m0 = mo_[0]; m1 = mo_[1]; m2 = mo_[2]; }
      R w0 = fma_(Gout[6], m2, fma_(Gout[3], m1, Gout[0] * m0));
      R w1 = fma_(Gout[7], m2, fma_(Gout[4], m1, Gout[1] * m0));
      R w2 = fma_(Gout[8], m2, fma_(Gout[5], m1, Gout[2] * m0));
      d0 = d0 - fma_(Li[3], w2, fma_(Li[1], w1, Li[0] * w0));
      d1 = d1 - fma_(Li[4], w2, Li[2] * w1);
      d2 = d2 - Li[5] * w2;
      MPMPC_SERIAL_END(N + 1);
    }
    R n0(0.0), n1(0.0), n2(0.0);
    auto out_step = [&](auto mode) {
      const R nv[3] = {n0, n1, n2};
      R pv[3];
      cdown_n<3, decltype(mode)::value>(nv, pv);
      const R p0 = pv[0], p1 = pv[1], p2 = pv[2];
      n0 = fma_(Gout[2], p2, fma_(Gout[1], p1, fma_(Gout[0], p0, d0)));
      n1 = fma_(Gout[5], p2, fma_(Gout[4], p1, fma_(Gout[3], p0, d1)));
      n2 = fma_(Gout[8], p2, fma_(Gout[7], p1, fma_(Gout[6], p0, d2)));
    };
    {
      MPMPC_SERIAL_BEGIN();
      if constexpr (L::staged_sweeps) {
        staged_sweep<+1>(last + 1, out_step);
      } else {
        const SweepMode<0> m0{};
        int s = 0;
        for (; s + 4 <= last + 1; s += 4) { out_step(m0); out_step(m0); out_step(m0); out_step(m0); }
        for (; s <= last; ++s) out_step(m0);
      }
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
      const R w[2] = {fma_(a[1], v[1], a[0] * v[0]), fma_(b[0], v[2], fma_(a[3], v[1], a[2] * v[0]))};
      R wu[2];
      up_n<2>(w, wu);
      r[0] = fma_(mI[0], v[0], wu[0]);
      r[1] = fma_(mI[1], v[1], wu[1]);
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
      R nd_[2];
      down_n<2>(nu, nd_);
      const R nd0 = nd_[0], nd1 = nd_[1];
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
    if constexpr (kS2) { factor_core2_s2(hx, wb, r); return; }          // two stages per lane: mpmpc_solver_s2.hpp
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
    up_n<3>(W, Dg);
    Dg[0] = Dg[0] + fma_(mI[0] * mI[0], h[0], r);
    Dg[2] = Dg[2] + fma_(mI[1] * mI[1], h[1], r);
    {
      R Tu[4];
      up_n<4>(T, Tu);
      const R zero(0.0);
      // coupling handed on: S_{k+1,k} = T going up, S_{k-1,k} = T_{k-1}' going down, nothing from mid
      To[0] = sel(down_chain, Tu[0], T[0]); To[1] = sel(down_chain, Tu[2], T[1]);
      To[2] = sel(down_chain, Tu[1], T[2]); To[3] = sel(down_chain, Tu[3], T[3]);
      MPMPC_UNROLL
      for (int i = 0; i < 4; ++i) To[i] = sel(is_mid, zero, To[i]);
    }
    { R t_[3]; mirror_n<3>(Dg, t_); MPMPC_UNROLL for (int i = 0; i < 3; ++i) Dg[i] = t_[i]; }
    { R t_[4]; mirror_n<4>(To, t_); MPMPC_UNROLL for (int i = 0; i < 4; ++i) To[i] = t_[i]; }
    if constexpr (kCR) { factor_cr2(Dg, To); return; }
    R M[4], Ls[4];
    MPMPC_UNROLL
    for (int i = 0; i < 4; ++i) M[i] = R(0.0);
    const int last = chain_steps();
    auto fstep = [&](auto mode, bool junction) {
      R Mr[4];
      cup_n<4, decltype(mode)::value>(M, Mr);
      R S00 = fma_(-Mr[1], Mr[1], fma_(-Mr[0], Mr[0], Dg[0]));
      R S10 = fma_(-Mr[3], Mr[1], fma_(-Mr[2], Mr[0], Dg[1]));
      R S11 = fma_(-Mr[3], Mr[3], fma_(-Mr[2], Mr[2], Dg[2]));
      if (junction) {
        R Mx[4];
        end_to_mid_n<4>(M, Mx);
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
      if constexpr (L::staged_sweeps) {
        staged_sweep<-1>(last, [&](auto mode) { fstep(mode, false); });
      } else {
        int s = 0;
        for (; s + 2 <= last; s += 2) { fstep(SweepMode<0>{}, false); fstep(SweepMode<0>{}, false); }
        for (; s < last; ++s) fstep(SweepMode<0>{}, false);
      }
      fstep(SweepMode<0>{}, true);
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
    // (The LAST level, D = 8, eliminates position 7, whose lower neighbour would be position -1: nothing of a row receives
    //  what it sends downwards - the row shifts have zero inflow - so D_a needs no update; and in a 16-lane chain its coupling
    //  Cm is an exact zero (the chain head's hand-on is zero and every fill S_ba below it is a product with that zero), so
    //  Ua, its share of Gin and of the new couplings are exact zeros too and are not formed: 16 moves and some thirty
    //  operations per factorisation.  In a 32-lane chain position 7 of the SECOND row has the first row's survivor below it:
    //  its Ua stays - in Gin, and in the fill S_YX it leaves with position 15.)
    constexpr bool kLast = (D == 8), kUa = !kLast || kCR32 || kCR64;
    if constexpr (kUa) {
      R ga[4];
      {
        R Ua[4];
        Ua[0] = i00 * Cm[0]; Ua[1] = i00 * Cm[1];
        Ua[2] = fma_(i11, Cm[2], i10 * Cm[0]); Ua[3] = fma_(i11, Cm[3], i10 * Cm[1]);
        MPMPC_UNROLL
        for (int i = 0; i < 4; ++i) Gin[i] = Gin[i] + Ua[i];
        // to the lower neighbour a (lane e - D):  D_a -= Ua'Ua
        if constexpr (!kLast) {
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
    if constexpr (kCR64) {
      // ---- chains of FOUR rows (a wavefront of a 128-lane workgroup) or EIGHT (two wavefronts of a 256-lane one).  The same
      // step as above, kCRrows - 1 times in turn: step r eliminates the survivor X of row r (position 15) against the survivor
      // Y of row r + 1 - which is the X of the next step, and after the last one the lane of the junction.  The moves are by
      // whole rows: inside the wavefront, or - the one step of an eight-row chain that crosses its two wavefronts - through LDS
      // (L::cr_pull / cr_push / cr_down / cr_bcast).
      MPMPC_SERIAL_BEGIN();
      MPMPC_UNROLL
      for (int r = 0; r < kCRrows - 1; ++r) {
        const Mk spec = L::cr64_special(r), isX = L::cr64_x(r), isY = L::cr64_x(r + 1);
        R w0 = sel(spec, fma_(Gin[2], Gin[2], Gin[0] * Gin[0]), zero), w1 = sel(spec, fma_(Gin[3], Gin[2], Gin[1] * Gin[0]), zero),
          w2 = sel(spec, fma_(Gin[3], Gin[3], Gin[1] * Gin[1]), zero);
        w0 = w0 + L::template rshl<1>(w0); w1 = w1 + L::template rshl<1>(w1); w2 = w2 + L::template rshl<1>(w2);
        w0 = w0 + L::template rshl<3>(w0); w1 = w1 + L::template rshl<3>(w1); w2 = w2 + L::template rshl<3>(w2);
        w0 = w0 + L::template rshl<7>(w0); w1 = w1 + L::template rshl<7>(w1); w2 = w2 + L::template rshl<7>(w2);
        {
          const R wv[3] = {w0, w1, w2};
          R wd[3];
          L::template cr_down<3>(r, wv, wd);
          Dg[0] = Dg[0] - sel(isX, wd[0], zero); Dg[1] = Dg[1] - sel(isX, wd[1], zero); Dg[2] = Dg[2] - sel(isX, wd[2], zero);
        }
        R i00 = rsqrt_(Dg[0]);
        const R l10 = Dg[1] * i00;
        R i11 = rsqrt_(fma_(-l10, l10, Dg[2]));
        R i10 = -(l10 * i00) * i11;
        i00 = sel(isX, i00, zero); i10 = sel(isX, i10, zero); i11 = sel(isX, i11, zero);
        Li[0] = Li[0] + i00; Li[1] = Li[1] + i10; Li[2] = Li[2] + i11;
        R Cb[4], Ub[4], gb[4];
        L::template cr_pull<4>(r, Cm, Cb);
        Ub[0] = i00 * Cb[0]; Ub[1] = i00 * Cb[2];
        Ub[2] = fma_(i11, Cb[1], i10 * Cb[0]); Ub[3] = fma_(i11, Cb[3], i10 * Cb[2]);
        L::template cr_push<4>(r, Ub, gb);
        MPMPC_UNROLL
        for (int i = 0; i < 4; ++i) { Gout[i] = Gout[i] + Ub[i]; gb[i] = sel(isY, gb[i], zero); }
        Dg[0] = fma_(-gb[2], gb[2], fma_(-gb[0], gb[0], Dg[0]));
        Dg[1] = fma_(-gb[3], gb[2], fma_(-gb[1], gb[0], Dg[1]));
        Dg[2] = fma_(-gb[3], gb[3], fma_(-gb[1], gb[1], Dg[2]));
      }
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
      end_to_mid_n<4>(M, Mx);
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
    // b_a -= Ua' y,  b_b -= Ub' y  (the last level, D = 8, has nobody below position 7 inside its row: cr_level)
    const R pb0 = fma_(Gout[2], e1, Gout[0] * e0), pb1 = fma_(Gout[3], e1, Gout[1] * e0);
    if constexpr (D == 8) {
      b0 = b0 - L::template rshr<D>(pb0);
      b1 = b1 - L::template rshr<D>(pb1);
    } else {
      const R pa0 = fma_(Gin[2], e1, Gin[0] * e0), pa1 = fma_(Gin[3], e1, Gin[1] * e0);
      b0 = b0 - L::template rshl<D>(pa0) - L::template rshr<D>(pb0);
      b1 = b1 - L::template rshl<D>(pa1) - L::template rshr<D>(pb1);
    }
  }
  template <int D>
  MPMPC_HD void cr_backward(const R& y0, const R& y1, R& n0, R& n1) const {
    const Mk E = L::template cr_elim<D>();
    // nu_e = inv(L_e)' (y_e - Ua nu_a - Ub nu_b),  nu_a from lane e - D, nu_b from lane e + D  (the last level, D = 8: nothing
    // flows in from below position 7 - nu_a is the shift's zero inflow there)
    // (The exchanges stay OUTSIDE the region below: a DPP move reads nothing from a lane the execution mask has switched off.)
    const R c0 = L::template rshl<D>(n0), c1 = L::template rshl<D>(n1);
    [[maybe_unused]] R a0(0.0), a1(0.0);
    if constexpr (D != 8) { a0 = L::template rshr<D>(n0); a1 = L::template rshr<D>(n1); }
    // (the level's lanes as ONE region: on the device only the lanes of E execute it, the selects inside fold away and nu is
    //  written in place under the execution mask; the lock-step emulation runs the body on every lane and the selects do the
    //  masking - the same values either way.  Not for the rank-one kernels (RKS): there it costs five registers too many.)
    auto level = [&] {
      R r0, r1;
      if constexpr (D == 8) {
        r0 = fma_(-Gout[1], c1, fma_(-Gout[0], c0, y0));
        r1 = fma_(-Gout[3], c1, fma_(-Gout[2], c0, y1));
      } else {
        r0 = fma_(-Gout[1], c1, fma_(-Gout[0], c0, fma_(-Gin[1], a1, fma_(-Gin[0], a0, y0))));
        r1 = fma_(-Gout[3], c1, fma_(-Gout[2], c0, fma_(-Gin[3], a1, fma_(-Gin[2], a0, y1))));
      }
      n0 = sel(E, fma_(Li[1], r1, Li[0] * r0), n0);
      n1 = sel(E, Li[2] * r1, n1);
    };
    if constexpr (RKS == 0) L::when(E, level); else level();
  }
  MPMPC_HD void s_solve_cr2(const R bv[2], R nu[2]) const {
    const R zero(0.0);
    R bm_[2];
    mirror_n<2>(bv, bm_);
    R b0 = sel(vxc, bm_[0], zero), b1 = sel(vxc, bm_[1], zero);
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
    if constexpr (kCR64) {
      MPMPC_SERIAL_BEGIN();
      MPMPC_UNROLL
      for (int r = 0; r < kCRrows - 1; ++r) {           // (factor_cr2: the survivors of the rows in turn)
        const Mk spec64 = L::cr64_special(r), isX64 = L::cr64_x(r), isY64 = L::cr64_x(r + 1);
        R c0 = sel(spec64, fma_(Gin[2], y1, Gin[0] * y0), zero), c1 = sel(spec64, fma_(Gin[3], y1, Gin[1] * y0), zero);
        c0 = c0 + L::template rshl<1>(c0); c1 = c1 + L::template rshl<1>(c1);
        c0 = c0 + L::template rshl<3>(c0); c1 = c1 + L::template rshl<3>(c1);
        c0 = c0 + L::template rshl<7>(c0); c1 = c1 + L::template rshl<7>(c1);
        const R cv[2] = {c0, c1};
        R cd[2];
        L::template cr_down<2>(r, cv, cd);
        const R bx0 = b0 - cd[0], bx1 = b1 - cd[1];
        const R yx0 = sel(isX64, Li[0] * bx0, zero), yx1 = sel(isX64, fma_(Li[2], bx1, Li[1] * bx0), zero);
        y0 = y0 + yx0; y1 = y1 + yx1;
        const R pv[2] = {fma_(Gout[2], yx1, Gout[0] * yx0), fma_(Gout[3], yx1, Gout[1] * yx0)};
        R pp_[2];
        L::template cr_push<2>(r, pv, pp_);
        b0 = b0 - sel(isY64, pp_[0], zero); b1 = b1 - sel(isY64, pp_[1], zero);
      }
      MPMPC_SERIAL_END(N + 1);
    }
    // junction: y_end = inv(L_end) b_end;  b_mid -= M y_end;  y_mid = inv(L_mid) b_mid;  nu_mid = inv(L_mid)' y_mid;
    //           nu_end = inv(L_end)' (y_end - M' nu_mid)
    MPMPC_SERIAL_BEGIN();                                                           // (census: useful on the two lanes of the junction only)
    const R ye0 = Li[0] * b0, ye1 = fma_(Li[2], b1, Li[1] * b0);                   // valid on the end lane (and, pre-update, on mid)
    const R q0 = fma_(Gout[1], ye1, Gout[0] * ye0), q1 = fma_(Gout[3], ye1, Gout[2] * ye0);      // M y_end on the end lane
    R qm0, qm1;
    { const R qv[2] = {q0, q1}; R qo_[2]; end_to_mid_n<2>(qv, qo_); qm0 = qo_[0]; qm1 = qo_[1]; }
    const R bm0 = b0 - qm0, bm1 = b1 - qm1;
    const R ym0 = Li[0] * bm0, ym1 = fma_(Li[2], bm1, Li[1] * bm0);
    const R nm0 = fma_(Li[1], ym1, Li[0] * ym0), nm1 = Li[2] * ym1;               // nu of the meeting stage (on mid)
    // to the end lane: M' nu_mid
    R me0, me1;
    { const R nv2[2] = {nm0, nm1}; R mo2_[2]; mid_to_end_n<2>(nv2, mo2_); me0 = mo2_[0]; me1 = mo2_[1]; }
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
    if constexpr (kCR64) {
      MPMPC_SERIAL_BEGIN();
      MPMPC_UNROLL
      for (int r = kCRrows - 2; r >= 0; --r) {          // (the survivors backwards: the last row's first - its Y is the junction lane)
        const Mk spec64 = L::cr64_special(r), isX64 = L::cr64_x(r);
        const R nv_[2] = {n0, n1};
        R cn_[2];
        L::template cr_pull<2>(r, nv_, cn_);
        const R c0 = cn_[0], c1 = cn_[1];
        const R r0 = y0 - fma_(Gout[1], c1, Gout[0] * c0), r1 = y1 - fma_(Gout[3], c1, Gout[2] * c0);
        n0 = sel(isX64, fma_(Li[1], r1, Li[0] * r0), n0);
        n1 = sel(isX64, Li[2] * r1, n1);
        const R nx_[2] = {n0, n1};
        R xb_[2];
        L::template cr_bcast<2>(r, nx_, xb_);
        const R x0 = xb_[0], x1 = xb_[1];
        y0 = y0 - sel(spec64, fma_(Gin[1], x1, Gin[0] * x0), zero);
        y1 = y1 - sel(spec64, fma_(Gin[3], x1, Gin[2] * x0), zero);
      }
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
    { const R nn_[2] = {n0, n1}; mirror_n<2>(nn_, nu); }
  }

  MPMPC_HD void s_solve2(const R bv[2], R nu[2]) const {
    if constexpr (kS2) { s_solve_s2(bv, nu); return; }
    if constexpr (kCR) { s_solve_cr2(bv, nu); return; }
    R b0 = sel(vxc, L::mirror(bv[0]), R(0.0)), b1 = sel(vxc, L::mirror(bv[1]), R(0.0));
    R c0 = Li[0] * b0;
    R c1 = fma_(Li[2], b1, Li[1] * b0);
    const int last = chain_steps();
    R y0(0.0), y1(0.0);
    auto in_step = [&](auto mode) {
      const R yv[2] = {y0, y1};
      R pv[2];
      cup_n<2, decltype(mode)::value>(yv, pv);
      const R p0 = pv[0], p1 = pv[1];
      y0 = fma_(Gin[1], p1, fma_(Gin[0], p0, c0));
      y1 = fma_(Gin[3], p1, fma_(Gin[2], p0, c1));
    };
    {
      MPMPC_SERIAL_BEGIN();
      if constexpr (L::staged_sweeps) {
        staged_sweep<-1>(last, in_step);
      } else {
        const SweepMode<0> m0{};
        int s = 0;
        for (; s + 4 <= last; s += 4) { in_step(m0); in_step(m0); in_step(m0); in_step(m0); }
        for (; s < last; ++s) in_step(m0);
      }
      MPMPC_SERIAL_END(last);
    }
    {
      // inward junction: the end lane forms M_own y, mid takes it on top of its chain input
      MPMPC_SERIAL_BEGIN();
      R t0 = fma_(Gout[1], y1, Gout[0] * y0);
      R t1 = fma_(Gout[3], y1, Gout[2] * y0);
      const R zero(0.0);
      { const R tv[2] = {t0, t1}; R to_[2]; end_to_mid_n<2>(tv, to_); t0 = to_[0]; t1 = to_[1]; }
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
      R m0, m1;
      { const R dv[2] = {d0, d1}; R mo_[2]; mid_to_end_n<2>(dv, mo_); m0 = mo_[0]; m1 = mo_[1]; }
      R w0 = fma_(Gout[2], m1, Gout[0] * m0);
      R w1 = fma_(Gout[3], m1, Gout[1] * m0);
      d0 = d0 - fma_(Li[1], w1, Li[0] * w0);
      d1 = d1 - Li[2] * w1;
      MPMPC_SERIAL_END(N + 1);
    }
    R n0(0.0), n1(0.0);
    auto out_step = [&](auto mode) {
      const R nv[2] = {n0, n1};
      R pv[2];
      cdown_n<2, decltype(mode)::value>(nv, pv);
      const R p0 = pv[0], p1 = pv[1];
      n0 = fma_(Gout[1], p1, fma_(Gout[0], p0, d0));
      n1 = fma_(Gout[3], p1, fma_(Gout[2], p0, d1));
    };
    {
      MPMPC_SERIAL_BEGIN();
      if constexpr (L::staged_sweeps) {
        staged_sweep<+1>(last + 1, out_step);
      } else {
        const SweepMode<0> m0{};
        int s = 0;
        for (; s + 4 <= last + 1; s += 4) { out_step(m0); out_step(m0); out_step(m0); out_step(m0); }
        for (; s <= last; ++s) out_step(m0);
      }
      MPMPC_SERIAL_END(last + 1);
    }
    { const R nn_[2] = {n0, n1}; mirror_n<2>(nn_, nu); }
  }

  MPMPC_HD void admm_factor(double sigma) {
    R h[5], Hd[5];
    MPMPC_UNROLL
    for (int j = 0; j < 5; ++j) { Hd[j] = p[j] + R(sigma) + (g[j] * g[j]) * rb[j]; h[j] = R(1.0) / Hd[j]; }
    dense_blocks(Hd, h);
    factor(h, rinv_eq);
  }
