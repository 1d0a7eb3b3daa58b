// TWO horizon stages per lane ("S = 2"): a lane backend on top of a one-stage-per-lane backend (lane_gpu.hpp on the device,
// lane_emu.hpp in the tests).  A lane of the base backend holds the stages 2p (component 0) and 2p + 1 (component 1) of its
// instance, so an instance of up to 2 G stages takes G lanes: N + 1 <= 32 in 16 lanes - FOUR instances per wavefront - where
// the one-stage layout packs two.  The same lane-generic solver code (mpmpc_core.hpp, mpmpc_reduced.hpp) runs on it; every
// per-stage value is a pair, every elementwise operation two instructions (the work per stage is what it was), and
//   * a one-stage shift along the horizon moves ONE component across lanes - the other one changes places inside the lane:
//     half the DPP moves per stage;
//   * an instance-wide reduction combines the two components in the lane and runs ONE butterfly for both: a wavefront's
//     reductions serve four instances instead of two;
//   * the cyclic reduction of the factorisation eliminates the even stages INSIDE the lanes (no lane is idle, no select, one
//     shifted operand) and runs its four cross-lane levels on the survivors - one per lane - for four instances at once
//     (Solver::factor_s2 / s_solve_s2).
// Price: twice the per-lane state - one wavefront per SIMD (up to 512 registers, 80 LDS slots = 40 KB per wave) instead of
// two; the independent instruction streams of the lane's two stages stand in for the second wave.
// VERDICT r5 "next round" item 1; replaces nothing of the reference - it is a layout of src/MPC.py:61-159's arithmetic.
#pragma once

#ifndef MPMPC_HD
#define MPMPC_HD inline
#endif

namespace mpmpc {

#ifdef MPMPC_LANE_EMU          // lane_emu.hpp: a "scalar" is a vector over the emulated lanes
using R1 = VD; using M1 = VB; using I1 = VI;
#else
using R1 = double; using M1 = bool; using I1 = int;
#endif

struct D2 {
  R1 v[2];
  MPMPC_HD D2() {}
  MPMPC_HD D2(double s) { v[0] = R1(s); v[1] = R1(s); }
  MPMPC_HD D2(const R1& a, const R1& b) { v[0] = a; v[1] = b; }
};
struct B2 {
  M1 v[2];
  MPMPC_HD B2() {}
  MPMPC_HD B2(const M1& a, const M1& b) { v[0] = a; v[1] = b; }
};
struct I2 {
  I1 v[2];
  MPMPC_HD I2() {}
  MPMPC_HD I2(int s) { v[0] = I1(s); v[1] = I1(s); }
  MPMPC_HD I2(const I1& a, const I1& b) { v[0] = a; v[1] = b; }
};

#define MPMPC_P2_BIN(op)                                                                                  \
  MPMPC_HD D2 operator op(const D2& a, const D2& b) { return D2(a.v[0] op b.v[0], a.v[1] op b.v[1]); }   \
  MPMPC_HD D2 operator op(const D2& a, double b) { return D2(a.v[0] op b, a.v[1] op b); }                 \
  MPMPC_HD D2 operator op(double a, const D2& b) { return D2(a op b.v[0], a op b.v[1]); }
MPMPC_P2_BIN(+) MPMPC_P2_BIN(-) MPMPC_P2_BIN(*) MPMPC_P2_BIN(/)
#undef MPMPC_P2_BIN
MPMPC_HD D2 operator-(const D2& a) { return D2(-a.v[0], -a.v[1]); }
MPMPC_HD D2& operator+=(D2& a, const D2& b) { a.v[0] += b.v[0]; a.v[1] += b.v[1]; return a; }
MPMPC_HD D2& operator-=(D2& a, const D2& b) { a.v[0] -= b.v[0]; a.v[1] -= b.v[1]; return a; }
MPMPC_HD D2& operator*=(D2& a, const D2& b) { a.v[0] *= b.v[0]; a.v[1] *= b.v[1]; return a; }
#define MPMPC_P2_CMP(op)                                                                                  \
  MPMPC_HD B2 operator op(const D2& a, const D2& b) { return B2(a.v[0] op b.v[0], a.v[1] op b.v[1]); }   \
  MPMPC_HD B2 operator op(const D2& a, double b) { return B2(a.v[0] op b, a.v[1] op b); }
MPMPC_P2_CMP(<) MPMPC_P2_CMP(>) MPMPC_P2_CMP(<=) MPMPC_P2_CMP(>=)
#undef MPMPC_P2_CMP
#define MPMPC_P2_ICMP(op) \
  MPMPC_HD B2 operator op(const I2& a, int b) { return B2(a.v[0] op b, a.v[1] op b); }
MPMPC_P2_ICMP(<) MPMPC_P2_ICMP(>) MPMPC_P2_ICMP(<=) MPMPC_P2_ICMP(>=) MPMPC_P2_ICMP(==) MPMPC_P2_ICMP(!=)
#undef MPMPC_P2_ICMP
MPMPC_HD B2 operator&(const B2& a, const B2& b) { return B2(a.v[0] & b.v[0], a.v[1] & b.v[1]); }
MPMPC_HD B2 operator|(const B2& a, const B2& b) { return B2(a.v[0] | b.v[0], a.v[1] | b.v[1]); }
MPMPC_HD B2 operator!(const B2& a) { return B2(!a.v[0], !a.v[1]); }
MPMPC_HD I2 operator+(const I2& a, const I2& b) { return I2(a.v[0] + b.v[0], a.v[1] + b.v[1]); }
MPMPC_HD I2 operator+(const I2& a, int b) { return I2(a.v[0] + b, a.v[1] + b); }
MPMPC_HD I2 operator-(const I2& a, int b) { return I2(a.v[0] - b, a.v[1] - b); }
MPMPC_HD I2 operator*(const I2& a, int b) { return I2(a.v[0] * b, a.v[1] * b); }
MPMPC_HD I2 modi(const I2& a, int m) { return I2(modi(a.v[0], m), modi(a.v[1], m)); }
MPMPC_HD I2 mini(const I2& a, int b) { return I2(mini(a.v[0], b), mini(a.v[1], b)); }
MPMPC_HD I2 maxi(const I2& a, int b) { return I2(maxi(a.v[0], b), maxi(a.v[1], b)); }

MPMPC_HD D2 fma_(const D2& a, const D2& b, const D2& c) { return D2(fma_(a.v[0], b.v[0], c.v[0]), fma_(a.v[1], b.v[1], c.v[1])); }
#define MPMPC_P2_UN(f) MPMPC_HD D2 f(const D2& a) { return D2(f(a.v[0]), f(a.v[1])); }
MPMPC_P2_UN(sqrt_) MPMPC_P2_UN(rcp_) MPMPC_P2_UN(rcp_fast_) MPMPC_P2_UN(rsqrt_) MPMPC_P2_UN(abs_) MPMPC_P2_UN(tan_) MPMPC_P2_UN(atan_)
#undef MPMPC_P2_UN
MPMPC_HD D2 max_(const D2& a, const D2& b) { return D2(max_(a.v[0], b.v[0]), max_(a.v[1], b.v[1])); }
MPMPC_HD D2 min_(const D2& a, const D2& b) { return D2(min_(a.v[0], b.v[0]), min_(a.v[1], b.v[1])); }
MPMPC_HD D2 sel(const B2& m, const D2& a, const D2& b) { return D2(sel(m.v[0], a.v[0], b.v[0]), sel(m.v[1], a.v[1], b.v[1])); }
MPMPC_HD I2 seli(const B2& m, const I2& a, const I2& b) { return I2(seli(m.v[0], a.v[0], b.v[0]), seli(m.v[1], a.v[1], b.v[1])); }
MPMPC_HD B2 selb(const B2& m, const B2& a, const B2& b) { return B2(selb(m.v[0], a.v[0], b.v[0]), selb(m.v[1], a.v[1], b.v[1])); }
MPMPC_HD B2 within_(const I2& v, int lo, int hi) { return B2(within_(v.v[0], lo, hi), within_(v.v[1], lo, hi)); }
MPMPC_HD B2 bit_(const I2& v, int b) { return B2(bit_(v.v[0], b), bit_(v.v[1], b)); }

// Base: a one-stage-per-lane backend whose instance is ONE chain of G lanes (no twist): a DPP row (G = C = 16: horizons
// 16 .. 31, four instances per wavefront) or a whole wavefront of four rows (G = C = 64: horizons 64 .. 127 in ONE wavefront -
// no LDS exchange, no workgroup barrier - where the one-stage layout needs a workgroup of two)
template <class Base>
struct LanePair {
  // (... or a workgroup of two wavefronts, G = C = 128: horizons 128 .. 255, the lanes' exchanges through LDS - LaneBlock)
  static_assert(Base::split == Base::group && (Base::group == 16 || Base::group == 64 || Base::group == 128),
                "two stages per lane: one chain of 16, 64 or 128 lanes per instance");
  using L1 = Base;                    // the lanes underneath: what the cross-lane levels of the cyclic reduction run on
  using real = D2;
  using mask = B2;
  using ival = I2;
  static constexpr int stages_per_lane = 2;
  static constexpr int group = 2 * Base::group;        // STAGES per instance
  static constexpr int split = group;                  // one chain, no twist: stage 0 on lane 0, component 0
  static constexpr int per_wave = Base::per_wave;
  static constexpr bool batched = Base::batched;       // (a workgroup base: the values of a step share one pass through LDS)
  static constexpr bool junction_moves = false;
  static constexpr bool staged_sweeps = false;
  static constexpr int cold_slots = Base::cold_slots / 2;

  static MPMPC_HD I2 lane_id() { const I1 l = Base::lane_id(); return I2(l, l); }
  static MPMPC_HD I2 stage() { const I1 s = Base::stage() * 2; return I2(s, s + 1); }
  static MPMPC_HD I2 slot() { const I1 s = Base::slot(); return I2(s, s); }
#ifndef MPMPC_LANE_EMU
  static MPMPC_HD I2 stage_again() { const int s = Base::stage_again() * 2; return I2(s, s + 1); }
  static MPMPC_HD I2 slot_again() { const int s = Base::slot_again(); return I2(s, s); }
#endif
  static MPMPC_HD B2 mtrue() { return B2(Base::mtrue(), Base::mtrue()); }
  static MPMPC_HD B2 mfalse() { return B2(Base::mfalse(), Base::mfalse()); }

  // the previous / next STAGE's value: stage 2p takes stage 2p - 1 from the lane below (a row shift: zero inflow at stage 0),
  // stage 2p + 1 takes stage 2p from its own lane - and the other way round
  // (the lanes' own one-lane shift with zero inflow at the ends of the instance: the row shift where an instance is a row)
  static MPMPC_HD R1 up1(const R1& a) { if constexpr (Base::group == 16) return Base::template rshr<1>(a); else return Base::up(a); }
  static MPMPC_HD R1 down1(const R1& a) { if constexpr (Base::group == 16) return Base::template rshl<1>(a); else return Base::down(a); }
  static MPMPC_HD D2 up(const D2& a) { return D2(up1(a.v[1]), a.v[0]); }
  static MPMPC_HD D2 down(const D2& a) { return D2(a.v[1], down1(a.v[0])); }
  // ... NV values at once (where the base batches its exchanges, they share ONE pass)
  template <int NV> static MPMPC_HD void up1n(const R1* v, R1* o) {
    if constexpr (Base::batched) Base::template upv<NV>(v, o);
    else { for (int i = 0; i < NV; ++i) o[i] = up1(v[i]); }
  }
  template <int NV> static MPMPC_HD void down1n(const R1* v, R1* o) {
    if constexpr (Base::batched) Base::template downv<NV>(v, o);
    else { for (int i = 0; i < NV; ++i) o[i] = down1(v[i]); }
  }
  template <int NV> static MPMPC_HD void upv(const D2* v, D2* o) {
    R1 t[NV], u[NV];
    for (int i = 0; i < NV; ++i) t[i] = v[i].v[1];
    up1n<NV>(t, u);
    for (int i = 0; i < NV; ++i) o[i] = D2(u[i], v[i].v[0]);
  }
  template <int NV> static MPMPC_HD void downv(const D2* v, D2* o) {
    R1 t[NV], u[NV];
    for (int i = 0; i < NV; ++i) t[i] = v[i].v[0];
    down1n<NV>(t, u);
    for (int i = 0; i < NV; ++i) o[i] = D2(v[i].v[1], u[i]);
  }
  template <int NV> static MPMPC_HD void cupv(const D2* v, D2* o) { upv<NV>(v, o); }
  template <int NV> static MPMPC_HD void cdownv(const D2* v, D2* o) { downv<NV>(v, o); }
  template <int NV> static MPMPC_HD void mirrorv(const D2* v, D2* o) { for (int i = 0; i < NV; ++i) o[i] = v[i]; }
  // (one chain in stage order: the chain layout is the stage layout)
  static MPMPC_HD D2 mirror(const D2& a) { return a; }
  static MPMPC_HD D2 cup(const D2& a) { return up(a); }
  static MPMPC_HD D2 cdown(const D2& a) { return down(a); }

  // instance-wide reductions: the lane's two stages first, then the base's butterfly - once for both
  static MPMPC_HD D2 gmax(const D2& a) { const R1 r = Base::gmax(max_(a.v[0], a.v[1])); return D2(r, r); }
  static MPMPC_HD D2 gmin(const D2& a) { const R1 r = Base::gmin(min_(a.v[0], a.v[1])); return D2(r, r); }
  static MPMPC_HD D2 gsum(const D2& a) { const R1 r = Base::gsum(a.v[0] + a.v[1]); return D2(r, r); }
  static MPMPC_HD B2 gany(const B2& m) { const M1 r = Base::gany(m.v[0] | m.v[1]); return B2(r, r); }
  static MPMPC_HD bool wany(const B2& m) { return Base::wany(m.v[0] | m.v[1]); }
  static MPMPC_HD D2 gcount(const B2& m) { const R1 r = Base::gcount(m.v[0]) + Base::gcount(m.v[1]); return D2(r, r); }
  // inclusive prefix sum along the stages: the lane totals are scanned across the lanes, the lanes below add to both stages
  static MPMPC_HD D2 gscan(const D2& a) {
    const R1 t = a.v[0] + a.v[1];
    const R1 below = up1(Base::gscan(t));
    return D2(below + a.v[0], below + t);
  }

  static MPMPC_HD void cold_put(int slot, const D2& a) { Base::cold_put(2 * slot, a.v[0]); Base::cold_put(2 * slot + 1, a.v[1]); }
  static MPMPC_HD D2 cold_get(int slot) { return D2(Base::cold_get(2 * slot), Base::cold_get(2 * slot + 1)); }
  static MPMPC_HD void fence() { Base::fence(); }
  static MPMPC_HD void sched_barrier() { Base::sched_barrier(); }

  template <class F>
  static MPMPC_HD void rows(double* dst, int rowlen, const I2& inst, int n_inst, F fill) {
    Base::rows(dst, rowlen, inst.v[0], n_inst, [&](auto put1) {
      fill([&](const I2& idx, const B2& ok, const D2& v) { put1(idx.v[0], ok.v[0], v.v[0]); put1(idx.v[1], ok.v[1], v.v[1]); });
    });
  }
  template <class F>
  static MPMPC_HD void rows_any(double* dst, int rowlen, const I2& inst, int n_inst, F fill) {
    Base::rows_any(dst, rowlen, inst.v[0], n_inst, [&](auto put1) {
      fill([&](const I2& idx, const B2& ok, const D2& v) { put1(idx.v[0], ok.v[0], v.v[0]); put1(idx.v[1], ok.v[1], v.v[1]); });
    });
  }
  static MPMPC_HD D2 load(const double* p, const I2& idx, const B2& ok, double dflt) {
    return D2(Base::load(p, idx.v[0], ok.v[0], dflt), Base::load(p, idx.v[1], ok.v[1], dflt));
  }
  static MPMPC_HD I2 loadi(const int* p, const I2& idx, const B2& ok, int dflt) {
    return I2(Base::loadi(p, idx.v[0], ok.v[0], dflt), Base::loadi(p, idx.v[1], ok.v[1], dflt));
  }
  static MPMPC_HD D2 gather(const double* p, const I2& idx, const B2& ok, double dflt) {
    return D2(Base::gather(p, idx.v[0], ok.v[0], dflt), Base::gather(p, idx.v[1], ok.v[1], dflt));
  }
  static MPMPC_HD I2 gatheri(const int* p, const I2& idx, const B2& ok, int dflt) {
    return I2(Base::gatheri(p, idx.v[0], ok.v[0], dflt), Base::gatheri(p, idx.v[1], ok.v[1], dflt));
  }
  static MPMPC_HD void store(double* p, const I2& idx, const B2& ok, const D2& a) {
    Base::store(p, idx.v[0], ok.v[0], a.v[0]); Base::store(p, idx.v[1], ok.v[1], a.v[1]);
  }
  static MPMPC_HD void storei(int* p, const I2& idx, const B2& ok, const I2& a) {
    Base::storei(p, idx.v[0], ok.v[0], a.v[0]); Base::storei(p, idx.v[1], ok.v[1], a.v[1]);
  }
  template <class F>
  static MPMPC_HD void when(const B2& ok, F f) { Base::when(ok.v[0] | ok.v[1], f); }
};

}  // namespace mpmpc
