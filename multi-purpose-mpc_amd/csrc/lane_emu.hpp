// Host-side emulation of one 64-lane wavefront, used ONLY by the CPU unit tests
// (tests/emul/): the solver core in mpmpc_core.hpp is written against an abstract
// "lane backend" so the very same source that hipcc compiles for gfx950 (lane = one
// double) can be executed in lock-step on the CPU (lane value = 64 doubles) and checked
// against the oracle without a GPU.  This file is not part of the shipped library.
#pragma once
#include <cmath>
#include <cstdint>
#define MPMPC_LANE_EMU 1          // (lane_pair.hpp: a "scalar" of the lanes underneath is a vector over the emulated lanes)

namespace mpmpc {

// width of the emulated execution group: one wavefront (64), or - horizons above 63, tests/emul/emul_wide.cpp - a workgroup of
// 2 / 4 wavefronts whose lanes exchange through LDS on the device (lane_gpu.hpp: LaneBlock); -DMPMPC_EMU_W=128 / 256
#ifndef MPMPC_EMU_W
#define MPMPC_EMU_W 64
#endif
constexpr int EMU_W = MPMPC_EMU_W;

// Optional instruction census (tests/emul built with -DMPMPC_COUNT_OPS): wave-level FP64
// instructions by class, used to state the algorithmic flop count of a solve in DESIGN.md.
#ifdef MPMPC_COUNT_OPS
// Counters by execution context (mpmpc_core.hpp marks the contexts): 0 = lane-parallel code, one stage per lane;
// 1 = lane-parallel code in the split layout of the interior point (states on lanes 0..31, inputs on 32..63);
// 2 = inside a serial sweep of the twisted factorisation / substitution (every step is executed by all lanes, but
// each stage's step does useful work exactly once per sweep).  serial_useful accumulates, per sweep, the flops of
// ONE step (section flops / steps): multiplied by the N + 1 stages that is the structure-exploiting flop count.
struct OpCount { long long fma, addmul, div, sqrt, cmpsel, shift, reduce; };
struct OpCensus {
  OpCount c[3];
  int mode = 0;
  double serial_useful = 0.0;
  long long mark = 0;
};
inline OpCensus& op_census() { static OpCensus s{}; return s; }
inline OpCount& op_count() { return op_census().c[op_census().mode]; }
inline long long op_flops(const OpCount& c) { return 2 * c.fma + c.addmul + c.div + c.sqrt; }
#define MPMPC_OP(f) (++op_count().f)
struct OpModeScope {          // RAII: contexts nest (a sweep inside the split interior point)
  int prev;
  explicit OpModeScope(int m) : prev(op_census().mode) { op_census().mode = m; }
  ~OpModeScope() { op_census().mode = prev; }
};
#define MPMPC_COUNT_CONTEXT(m) OpModeScope mpmpc_count_scope_(m)
#define MPMPC_SERIAL_BEGIN() OpModeScope mpmpc_serial_scope_(2); op_census().mark = op_flops(op_census().c[2])
#define MPMPC_SERIAL_END(nsteps) op_census().serial_useful += double(op_flops(op_census().c[2]) - op_census().mark) / double((nsteps) > 0 ? (nsteps) : 1)
#else
#define MPMPC_OP(f) ((void)0)
#endif

struct VB { bool v[EMU_W]; };
struct VI {
  int v[EMU_W];
  VI() {}
  VI(int s) { for (int i = 0; i < EMU_W; ++i) v[i] = s; }
};
struct VD {
  double v[EMU_W];
  VD() {}
  VD(double s) { for (int i = 0; i < EMU_W; ++i) v[i] = s; }
};

#define MPMPC_EMU_BIN(op)                                                                       \
  inline VD operator op(const VD& a, const VD& b) { MPMPC_OP(cls); VD r; for (int i = 0; i < EMU_W; ++i) r.v[i] = a.v[i] op b.v[i]; return r; } \
  inline VD operator op(const VD& a, double b) { MPMPC_OP(cls); VD r; for (int i = 0; i < EMU_W; ++i) r.v[i] = a.v[i] op b; return r; }       \
  inline VD operator op(double a, const VD& b) { MPMPC_OP(cls); VD r; for (int i = 0; i < EMU_W; ++i) r.v[i] = a op b.v[i]; return r; }
#define cls addmul
MPMPC_EMU_BIN(+) MPMPC_EMU_BIN(-) MPMPC_EMU_BIN(*)
#undef cls
#define cls div
MPMPC_EMU_BIN(/)
#undef cls
#undef MPMPC_EMU_BIN
inline VD operator-(const VD& a) { VD r; for (int i = 0; i < EMU_W; ++i) r.v[i] = -a.v[i]; return r; }
inline VD& operator+=(VD& a, const VD& b) { for (int i = 0; i < EMU_W; ++i) a.v[i] += b.v[i]; return a; }
inline VD& operator-=(VD& a, const VD& b) { for (int i = 0; i < EMU_W; ++i) a.v[i] -= b.v[i]; return a; }
inline VD& operator*=(VD& a, const VD& b) { for (int i = 0; i < EMU_W; ++i) a.v[i] *= b.v[i]; return a; }

#define MPMPC_EMU_CMP(op)                                                                       \
  inline VB operator op(const VD& a, const VD& b) { VB r; for (int i = 0; i < EMU_W; ++i) r.v[i] = a.v[i] op b.v[i]; return r; } \
  inline VB operator op(const VD& a, double b) { VB r; for (int i = 0; i < EMU_W; ++i) r.v[i] = a.v[i] op b; return r; }
MPMPC_EMU_CMP(<) MPMPC_EMU_CMP(>) MPMPC_EMU_CMP(<=) MPMPC_EMU_CMP(>=)
#undef MPMPC_EMU_CMP
#define MPMPC_EMU_ICMP(op)                                                                      \
  inline VB operator op(const VI& a, int b) { VB r; for (int i = 0; i < EMU_W; ++i) r.v[i] = a.v[i] op b; return r; }
MPMPC_EMU_ICMP(<) MPMPC_EMU_ICMP(>) MPMPC_EMU_ICMP(<=) MPMPC_EMU_ICMP(>=) MPMPC_EMU_ICMP(==) MPMPC_EMU_ICMP(!=)
#undef MPMPC_EMU_ICMP
inline VB operator&(const VB& a, const VB& b) { VB r; for (int i = 0; i < EMU_W; ++i) r.v[i] = a.v[i] && b.v[i]; return r; }
inline VB operator|(const VB& a, const VB& b) { VB r; for (int i = 0; i < EMU_W; ++i) r.v[i] = a.v[i] || b.v[i]; return r; }
inline VB operator!(const VB& a) { VB r; for (int i = 0; i < EMU_W; ++i) r.v[i] = !a.v[i]; return r; }
inline VI operator+(const VI& a, const VI& b) { VI r; for (int i = 0; i < EMU_W; ++i) r.v[i] = a.v[i] + b.v[i]; return r; }
inline VI operator+(const VI& a, int b) { VI r; for (int i = 0; i < EMU_W; ++i) r.v[i] = a.v[i] + b; return r; }
inline VI operator-(const VI& a, int b) { VI r; for (int i = 0; i < EMU_W; ++i) r.v[i] = a.v[i] - b; return r; }
inline VI modi(const VI& a, int m) { VI r; for (int i = 0; i < EMU_W; ++i) r.v[i] = a.v[i] % m; return r; }
inline VI mini(const VI& a, int b) { VI r; for (int i = 0; i < EMU_W; ++i) r.v[i] = a.v[i] < b ? a.v[i] : b; return r; }
inline VI maxi(const VI& a, int b) { VI r; for (int i = 0; i < EMU_W; ++i) r.v[i] = a.v[i] > b ? a.v[i] : b; return r; }
inline VI operator*(const VI& a, int b) { VI r; for (int i = 0; i < EMU_W; ++i) r.v[i] = a.v[i] * b; return r; }

inline VD fma_(const VD& a, const VD& b, const VD& c) { MPMPC_OP(fma); VD r; for (int i = 0; i < EMU_W; ++i) r.v[i] = std::fma(a.v[i], b.v[i], c.v[i]); return r; }
inline VD sqrt_(const VD& a) { MPMPC_OP(sqrt); VD r; for (int i = 0; i < EMU_W; ++i) r.v[i] = std::sqrt(a.v[i]); return r; }
inline VD rcp_(const VD& a) { MPMPC_OP(div); VD r; for (int i = 0; i < EMU_W; ++i) r.v[i] = 1.0 / a.v[i]; return r; }
inline VD rcp_fast_(const VD& a) { MPMPC_OP(div); VD r; for (int i = 0; i < EMU_W; ++i) r.v[i] = 1.0 / a.v[i]; return r; }
inline VD rsqrt_(const VD& a) { MPMPC_OP(sqrt); VD r; for (int i = 0; i < EMU_W; ++i) r.v[i] = 1.0 / std::sqrt(a.v[i]); return r; }
inline VD abs_(const VD& a) { VD r; for (int i = 0; i < EMU_W; ++i) r.v[i] = std::fabs(a.v[i]); return r; }
inline VD max_(const VD& a, const VD& b) { MPMPC_OP(cmpsel); VD r; for (int i = 0; i < EMU_W; ++i) r.v[i] = std::fmax(a.v[i], b.v[i]); return r; }
inline VD min_(const VD& a, const VD& b) { MPMPC_OP(cmpsel); VD r; for (int i = 0; i < EMU_W; ++i) r.v[i] = std::fmin(a.v[i], b.v[i]); return r; }
inline VD tan_(const VD& a) { VD r; for (int i = 0; i < EMU_W; ++i) r.v[i] = std::tan(a.v[i]); return r; }
inline VD atan_(const VD& a) { VD r; for (int i = 0; i < EMU_W; ++i) r.v[i] = std::atan(a.v[i]); return r; }
inline VD sel(const VB& m, const VD& a, const VD& b) { MPMPC_OP(cmpsel); VD r; for (int i = 0; i < EMU_W; ++i) r.v[i] = m.v[i] ? a.v[i] : b.v[i]; return r; }
inline VI seli(const VB& m, const VI& a, const VI& b) { VI r; for (int i = 0; i < EMU_W; ++i) r.v[i] = m.v[i] ? a.v[i] : b.v[i]; return r; }
inline VB within_(const VI& v, int lo, int hi) { VB r; for (int i = 0; i < EMU_W; ++i) r.v[i] = v.v[i] >= lo && v.v[i] <= hi; return r; }
inline VB bit_(const VI& v, int b) { VB r; for (int i = 0; i < EMU_W; ++i) r.v[i] = ((v.v[i] >> b) & 1) != 0; return r; }
inline VB selb(const VB& m, const VB& a, const VB& b) { VB r; for (int i = 0; i < EMU_W; ++i) r.v[i] = m.v[i] ? a.v[i] : b.v[i]; return r; }

// G = lanes per instance (16, 32 or 64; 128 / 256 in the wide builds); one emulated wave carries EMU_W / G instances.
// C = split of the twisted factorisation (mpmpc_core.hpp, factor): lanes [C, 2C) of an instance are
// reversed in chain layout; C == G means no second chain.
// SLOTS: cold slots (the device backends' LDS budget: 66 the general solver, 2 x 40 the pair layout, 2 x 37 its lean form)
template <int G, int C = G / 2, int SLOTS = 80>
struct LaneEmu {
  static constexpr int split = C;
  using real = VD;
  using mask = VB;
  using ival = VI;
  static constexpr int group = G;
  static constexpr int per_wave = EMU_W / G;
  static constexpr bool batched = false;
  static constexpr bool junction_moves = false;
  static constexpr bool staged_sweeps = false;
  static constexpr int stages_per_lane = 1;          // (2: lane_pair.hpp)

  static VI lane_id() { VI r; for (int i = 0; i < EMU_W; ++i) r.v[i] = i; return r; }
  static VI stage() { VI r; for (int i = 0; i < EMU_W; ++i) r.v[i] = i % G; return r; }
  static VI slot() { VI r; for (int i = 0; i < EMU_W; ++i) r.v[i] = i / G; return r; }
  static VB mtrue() { VB r; for (int i = 0; i < EMU_W; ++i) r.v[i] = true; return r; }
  static VB mfalse() { VB r; for (int i = 0; i < EMU_W; ++i) r.v[i] = false; return r; }

  // value of the previous / next stage's lane (0.0 at the ends of an instance)
  static VD up(const VD& a) { MPMPC_OP(shift); VD r; for (int i = 0; i < EMU_W; ++i) r.v[i] = (i % G == 0) ? 0.0 : a.v[i - 1]; return r; }
  static VD down(const VD& a) { MPMPC_OP(shift); VD r; for (int i = 0; i < EMU_W; ++i) r.v[i] = (i % G == G - 1) ? 0.0 : a.v[i + 1]; return r; }

  // chain layout: reversal of the lanes [C, 2C) and one-lane shifts with zero inflow at the chain ends
  static VD mirror(const VD& a) {
    MPMPC_OP(shift);
    VD r;
    for (int i = 0; i < EMU_W; ++i) { int l = i % G; r.v[i] = (l >= C && l < 2 * C) ? a.v[i - l + 3 * C - 1 - l] : a.v[i]; }
    return r;
  }
  // (exactly what the device does: with C = 16 row shifts, every row of 16 lanes zero-filled, also the rows beyond
  //  2C; with C = 32 plain wavefront shifts, so lane 32 / lane 31 do receive their neighbour's value - which the
  //  zero coupling block of the meeting stage multiplies away, see lane_gpu.hpp)
  static VD cup(const VD& a) { MPMPC_OP(shift); VD r; for (int i = 0; i < EMU_W; ++i) { int l = i % G; r.v[i] = (l == 0 || (l == C && C != 32) || (C == 16 && l % 16 == 0)) ? 0.0 : a.v[i - 1]; } return r; }
  static VD cdown(const VD& a) { MPMPC_OP(shift); VD r; for (int i = 0; i < EMU_W; ++i) { int l = i % G; r.v[i] = ((l == C - 1 && C != 32) || l == 2 * C - 1 || l == G - 1 || (C == 16 && l % 16 == 15)) ? 0.0 : a.v[i + 1]; } return r; }

  // shifts by D lanes inside each row of 16 lanes, zero inflow (see lane_gpu.hpp)
  template <int D>
  static VD rshr(const VD& a) { MPMPC_OP(shift); VD r; for (int i = 0; i < EMU_W; ++i) r.v[i] = (i % 16 >= D) ? a.v[i - D] : 0.0; return r; }
  template <int D>
  static VD rshl(const VD& a) { MPMPC_OP(shift); VD r; for (int i = 0; i < EMU_W; ++i) r.v[i] = (i % 16 + D < 16) ? a.v[i + D] : 0.0; return r; }
  template <int D>
  static VB cr_elim() { VB r; for (int i = 0; i < EMU_W; ++i) r.v[i] = (((i % 16) + D + 1) & (2 * D - 1)) == 0; return r; }

  // row-pair exchanges and lane roles of the 32-lane chains' cyclic reduction (see lane_gpu.hpp)
  static VD from_even_row(const VD& a) { MPMPC_OP(shift); VD r; for (int i = 0; i < EMU_W; ++i) r.v[i] = a.v[i & ~16]; return r; }
  static VD from_odd_row(const VD& a) { MPMPC_OP(shift); VD r; for (int i = 0; i < EMU_W; ++i) r.v[i] = a.v[i | 16]; return r; }
  static VD bcast31(const VD& a) { MPMPC_OP(shift); VD r; for (int i = 0; i < EMU_W; ++i) r.v[i] = (i & 32) ? a.v[(i & ~63) | 31] : 0.0; return r; }
  // inclusive prefix sum along the lanes of an instance, in the device's order of additions (lane_gpu.hpp: gscan)
  static VD gscan(VD a) {
    a = a + rshr<1>(a); a = a + rshr<2>(a); a = a + rshr<4>(a); a = a + rshr<8>(a);
    if constexpr (G >= 32) a = a + bcast15(a);
    if constexpr (G >= 64) a = a + bcast31(a);
    if constexpr (G > 64) {       // a workgroup of 2 / 4 wavefronts: the totals of the wavefronts below, one after the other (LaneBlock::gscan)
      VD off;
      for (int i = 0; i < EMU_W; ++i) {
        double o = 0.0;
        for (int j = 0; j < (i % G) / 64; ++j) o = o + a.v[(i - i % G) + 64 * j + 63];
        off.v[i] = o;
      }
      MPMPC_OP(shift);
      a = a + off;
    }
    return a;
  }
  static VD bcast15(const VD& a) { MPMPC_OP(shift); VD r; for (int i = 0; i < EMU_W; ++i) r.v[i] = (i & 16) ? a.v[(i & ~31) | 15] : 0.0; return r; }
  static VB cr_low15() { VB r; for (int i = 0; i < EMU_W; ++i) r.v[i] = (i & 31) == 15; return r; }
  static VB cr_special() { VB r; for (int i = 0; i < EMU_W; ++i) { const int p = i & 15; r.v[i] = (i & 16) != 0 && ((p & (p + 1)) == 0) && p != 15; } return r; }

  // ---- chains of four / eight rows (a 128- / 256-lane workgroup; lane_gpu.hpp: LaneBlock, Solver::kCR64): position of a lane in
  // its chain = lane % C; step r works on the survivor X of row r, the survivor Y of row r + 1 and that row's lanes 0, 1, 3, 7
  static VB cr64_x(int r) { VB m; for (int i = 0; i < EMU_W; ++i) m.v[i] = (i % C) == 16 * r + 15; return m; }
  static VB cr64_special(int r) { VB m; for (int i = 0; i < EMU_W; ++i) { const int p = i & 15; m.v[i] = ((i % C) >> 4) == r + 1 && ((p & (p + 1)) == 0) && p != 15; } return m; }
  // pull: every lane gets v of the same position one row up the chain; push: one row down; down: of the next lane; bcast: of
  // position 15 of the row below (what a lane without such a source gets is not used: 0 here)
  template <int NV> static void cr_pull(int, const VD* v, VD* o) { MPMPC_OP(shift); for (int k = 0; k < NV; ++k) for (int i = 0; i < EMU_W; ++i) o[k].v[i] = ((i % C) + 16 < C) ? v[k].v[i + 16] : 0.0; }
  template <int NV> static void cr_push(int, const VD* v, VD* o) { MPMPC_OP(shift); for (int k = 0; k < NV; ++k) for (int i = 0; i < EMU_W; ++i) o[k].v[i] = ((i % C) >= 16) ? v[k].v[i - 16] : 0.0; }
  template <int NV> static void cr_down(int, const VD* v, VD* o) { MPMPC_OP(shift); for (int k = 0; k < NV; ++k) for (int i = 0; i < EMU_W; ++i) o[k].v[i] = ((i % C) != C - 1) ? v[k].v[i + 1] : 0.0; }
  template <int NV> static void cr_bcast(int, const VD* v, VD* o) { MPMPC_OP(shift); for (int k = 0; k < NV; ++k) for (int i = 0; i < EMU_W; ++i) o[k].v[i] = ((i % C) >= 16) ? v[k].v[((i & ~15) - 16) | 15] : 0.0; }

  // half-wave exchange (see lane_gpu.hpp)
  static VD from_upper(const VD& a) { MPMPC_OP(shift); VD r; for (int i = 0; i < EMU_W; ++i) r.v[i] = a.v[i | 32]; return r; }
  static VD from_lower(const VD& a) { MPMPC_OP(shift); VD r; for (int i = 0; i < EMU_W; ++i) r.v[i] = a.v[i & 31]; return r; }

  // butterfly all-reduce inside an instance's lanes; same association order as the GPU backend
  template <class F>
  static VD bfly(VD a, F f) {
    MPMPC_OP(reduce);
    for (int off = 1; off < G; off <<= 1) {
      VD t;
      for (int i = 0; i < EMU_W; ++i) t.v[i] = a.v[i ^ off];
      for (int i = 0; i < EMU_W; ++i) a.v[i] = f(a.v[i], t.v[i]);
    }
    return a;
  }
  static VD gmax(const VD& a) { return bfly(a, [](double x, double y) { return std::fmax(x, y); }); }
  static VD gmin(const VD& a) { return bfly(a, [](double x, double y) { return std::fmin(x, y); }); }
  static VD gsum(const VD& a) { return bfly(a, [](double x, double y) { return x + y; }); }
  static VB gany(const VB& m) {
    VB r;
    for (int g = 0; g < per_wave; ++g) {
      bool any = false;
      for (int i = 0; i < G; ++i) any = any || m.v[g * G + i];
      for (int i = 0; i < G; ++i) r.v[g * G + i] = any;
    }
    return r;
  }
  static VD gcount(const VB& m) {
    VD r;
    for (int g = 0; g < per_wave; ++g) {
      int c = 0;
      for (int i = 0; i < G; ++i) c += m.v[g * G + i] ? 1 : 0;
      for (int i = 0; i < G; ++i) r.v[g * G + i] = double(c);
    }
    return r;
  }
  static bool wany(const VB& m) { for (int i = 0; i < EMU_W; ++i) if (m.v[i]) return true; return false; }

  // "cold" per-lane storage (LDS on the GPU) for values that are only needed at termination checks
  // and in the certificate, so that they do not occupy registers inside the iteration loops
  static constexpr int cold_slots = SLOTS;
  static VD* cold() { static VD buf[cold_slots]; return buf; }
  static void cold_put(int slot, const VD& a) { cold()[slot] = a; }
  static VD cold_get(int slot) { return cold()[slot]; }
  static void fence() {}
  static void sched_barrier() {}

  // output rows: the device stages them in LDS and writes them coalesced; same memory image here
  template <class F>
  static void rows(double* dst, int rowlen, const VI& inst, int /*n_inst*/, F fill) {
    fill([&](const VI& idx, const VB& ok, const VD& v) { store(dst, inst * rowlen + idx, ok, v); });
  }

  template <class F>
  static void rows_any(double* dst, int rowlen, const VI& inst, int n_inst, F fill) { rows(dst, rowlen, inst, n_inst, fill); }

  static VD load(const double* p, const VI& idx, const VB& ok, double dflt) {
    VD r; for (int i = 0; i < EMU_W; ++i) r.v[i] = ok.v[i] ? p[idx.v[i]] : dflt; return r;
  }
  static VI loadi(const int* p, const VI& idx, const VB& ok, int dflt) {
    VI r; for (int i = 0; i < EMU_W; ++i) r.v[i] = ok.v[i] ? p[idx.v[i]] : dflt; return r;
  }
  template <class F>
  static void when(const VB& /*ok*/, F f) { f(); }          // (every store inside carries its own mask)
  static VD gather(const double* p, const VI& idx, const VB& ok, double dflt) { return load(p, idx, ok, dflt); }
  static VI gatheri(const int* p, const VI& idx, const VB& ok, int dflt) { return loadi(p, idx, ok, dflt); }
  static void store(double* p, const VI& idx, const VB& ok, const VD& a) {
    for (int i = 0; i < EMU_W; ++i) if (ok.v[i]) p[idx.v[i]] = a.v[i];
  }
  static void storei(int* p, const VI& idx, const VB& ok, const VI& a) {
    for (int i = 0; i < EMU_W; ++i) if (ok.v[i]) p[idx.v[i]] = a.v[i];
  }
};

}  // namespace mpmpc
