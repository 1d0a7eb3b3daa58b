// gfx950 lane backend for mpmpc_core.hpp: one lane = one horizon stage of one QP,
// G consecutive lanes of a 64-wide wavefront = one QP instance (64/G instances per wave).
// Neighbour exchange along the horizon is a DPP wavefront shift (no LDS, no memory);
// instance-wide norms are butterfly reductions over the G lanes.
#pragma once
#include <hip/hip_runtime.h>

namespace mpmpc {

__device__ __forceinline__ double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ double sqrt_(double a) { return __builtin_sqrt(a); }
// 1/sqrt(a): hardware seed (v_rsq_f64, measured relative error 5e-8) + one cubic Newton step in FMA form:
// 1.24 ulp, the same as with a further quadratic step (profiles/micro/rsq_accuracy.hip)
__device__ __forceinline__ double rsqrt_(double a) {
  double r = __builtin_amdgcn_rsq(a);
  double e = __builtin_fma(-(a * r), r, 1.0);
  return __builtin_fma(r * e, __builtin_fma(0.375, e, 0.5), r);
}
// 1/a: hardware seed (v_rcp_f64, 5e-8) + one cubic step r (1 + e + e^2): 1.0 ulp; a product with it replaces
// the full IEEE division sequence where the last bit does not matter (interior-point iteration)
__device__ __forceinline__ double rcp_(double a) {
  double r = __builtin_amdgcn_rcp(a);
  double e = __builtin_fma(-a, r, 1.0);
  return __builtin_fma(r, __builtin_fma(e, e, e), r);
}
// 1/a by the hardware seed alone (relative error 5e-8): where the quotient only sizes a step
__device__ __forceinline__ double rcp_fast_(double a) { return __builtin_amdgcn_rcp(a); }
__device__ __forceinline__ double abs_(double a) { return __builtin_fabs(a); }
__device__ __forceinline__ double max_(double a, double b) { return __builtin_fmax(a, b); }   // v_max_f64
__device__ __forceinline__ double min_(double a, double b) { return __builtin_fmin(a, b); }   // v_min_f64
// v_max_f64 / v_min_f64 as ONE instruction, for the steps of a wave reduction: through __builtin_fmax the compiler (IEEE mode)
// puts a canonicalising v_max x, x, x in front of every operand it cannot prove quiet - and a value that has come through a
// DPP move is one.  The instruction itself quiets what it returns and returns the other operand for a NaN.
__device__ __forceinline__ double max_raw_(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ double min_raw_(double a, double b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ double tan_(double a) { return ::tan(a); }
__device__ __forceinline__ double atan_(double a) { return ::atan(a); }
__device__ __forceinline__ double sel(bool m, double a, double b) { return m ? a : b; }
__device__ __forceinline__ int seli(bool m, int a, int b) { return m ? a : b; }
__device__ __forceinline__ int modi(int a, int m) { return a % m; }
__device__ __forceinline__ int mini(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int maxi(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ bool selb(bool m, bool a, bool b) { return m ? a : b; }
__device__ __forceinline__ bool bit_(int v, int b) { return ((v >> b) & 1) != 0; }
__device__ __forceinline__ bool within_(int v, int lo, int hi) { return (unsigned)(v - lo) <= (unsigned)(hi - lo); }   // lo <= v <= hi

// DPP controls (GFX9)
constexpr int DPP_ROW_SHL1 = 0x101;    // lane i <- lane i+1 inside each row of 16 lanes
constexpr int DPP_ROW_SHR1 = 0x111;    // lane i <- lane i-1 inside each row of 16 lanes
constexpr int DPP_WAVE_SHL1 = 0x130;   // lane i <- lane i+1 over the whole wavefront
constexpr int DPP_WAVE_SHR1 = 0x138;   // lane i <- lane i-1 over the whole wavefront
constexpr int DPP_ROW_MIRROR = 0x140;  // reverse the 16 lanes of each row
constexpr int DPP_ROW_HALF_MIRROR = 0x141;   // reverse each half row of 8 lanes

template <int CTRL>
__device__ __forceinline__ double dpp_shift(double a) {
  int lo = __double2loint(a), hi = __double2hiint(a);
  // old = 0 and bound_ctrl = true: the lane with no source reads 0
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
// lanes selected by ROWS / BANKS take the permuted value, all others keep `keep`
template <int CTRL, int ROWS, int BANKS, bool BOUND>
__device__ __forceinline__ double dpp_merge(double keep, double a) {
  int lo = __builtin_amdgcn_update_dpp(__double2loint(keep), __double2loint(a), CTRL, ROWS, BANKS, BOUND);
  int hi = __builtin_amdgcn_update_dpp(__double2hiint(keep), __double2hiint(a), CTRL, ROWS, BANKS, BOUND);
  return __hiloint2double(hi, lo);
}

// G lanes per instance; C = split of the twisted factorisation (mpmpc_core.hpp, factor): in chain
// layout the lanes [C, 2C) of an instance are reversed.  Supported: <64,16> (N + 1 <= 32), <64,32>,
// <32,16>, <16,16> (no second chain).
// SLOTS: 512-byte slots of per-wave LDS ("cold" storage + staging of the output rows): 66 for the general solver (33 KB: four
// waves per CU, one per SIMD), 40 for the reduced-native one (20 KB: eight waves per CU, two per SIMD, exactly the 160 KB).
template <int G, int C = G / 2, int SLOTS = 66>
struct LaneGpu {
  // (<64, 64>: ONE chain of four rows - the lanes underneath the two-stages-per-lane layout of horizons 64 .. 127, lane_pair.hpp)
  static_assert((G == 64 && (C == 16 || C == 32 || C == 64)) || (G == 32 && C == 16) || (G == 16 && C == 16), "unsupported lane split");
  static constexpr int split = C;
  using real = double;
  using mask = bool;
  using ival = int;
  static constexpr int group = G;
  static constexpr int per_wave = 64 / G;
  static constexpr bool batched = false;       // lane exchanges are register moves: nothing to batch (Solver::cup_n)
  static constexpr int stages_per_lane = 1;    // (2: lane_pair.hpp on top of LaneGpu<16, 16>)
  static constexpr bool staged_sweeps = false; // a chain never spans two wavefronts here (Solver::staged_sweep)

  static __device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
  static __device__ __forceinline__ int stage() { return (threadIdx.x & 63) % G; }
  static __device__ __forceinline__ int slot() { return (threadIdx.x & 63) / G; }
  // The lane's number formed anew (blocks are one wavefront): a kernel that needs stage() / slot() once at its start and
  // once at its very end asks again there instead of carrying a register through everything in between.  (The opaque
  // zero keeps the compiler from merging this with the first computation.)
  static __device__ __forceinline__ int lane_again() {
    int z;
    asm volatile("s_mov_b32 %0, 0" : "=s"(z));
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, z));
  }
  static __device__ __forceinline__ int stage_again() { return lane_again() % G; }
  static __device__ __forceinline__ int slot_again() { return lane_again() / G; }
  static __device__ __forceinline__ bool mtrue() { return true; }
  static __device__ __forceinline__ bool mfalse() { return false; }

  static __device__ __forceinline__ double up(double a) {
    double r = dpp_shift<DPP_WAVE_SHR1>(a);
    if (G < 64) r = (stage() == 0) ? 0.0 : r;
    return r;
  }
  static __device__ __forceinline__ double down(double a) {
    double r = dpp_shift<DPP_WAVE_SHL1>(a);
    if (G < 64) r = (stage() == G - 1) ? 0.0 : r;
    return r;
  }
  // ---- chain layout of the twisted factorisation
  static __device__ __forceinline__ double mirror(double a) {
    if constexpr (C == G) {
      return a;
    } else if constexpr (C == 16) {
      return dpp_merge<DPP_ROW_MIRROR, (G == 64 ? 0x2 : 0xA), 0xf, false>(a, a);     // rows 1 (and 3)
    } else {                                                                          // <64,32>: lanes 32..63
      const int src = lane_id() < 32 ? lane_id() : 95 - lane_id();
      return __shfl(a, src, 64);
    }
  }
  // one-lane shifts along the chains; lanes 0 and C (cup), C-1 and 2C-1 (cdown) of an instance read 0
  static __device__ __forceinline__ double cup(double a) {
    if constexpr (C == G) {
      return up(a);
    } else if constexpr (C == 16) {
      return dpp_shift<DPP_ROW_SHR1>(a);                 // chains are rows: the row shift zero-fills
    } else {
      // <64,32>: lane 32 (head of the descending chain) receives lane 31's value instead of 0.  That is harmless:
      // lane 31 is the meeting stage, whose outgoing block is exactly zero (To = 0 in factor), so lane 32's
      // recurrence matrix Gin = -Li M_in is zero and whatever arrives in the sweeps is multiplied by it.
      return dpp_shift<DPP_WAVE_SHR1>(a);
    }
  }
  static __device__ __forceinline__ double cdown(double a) {
    if constexpr (C == G) {
      return down(a);
    } else if constexpr (C == 16) {
      return dpp_shift<DPP_ROW_SHL1>(a);
    } else {
      // <64,32>: lane 31 (the meeting stage) receives lane 32's value instead of 0; its Gout is zero (M_own = 0)
      return dpp_shift<DPP_WAVE_SHL1>(a);
    }
  }

  // ---- shifts by D lanes inside each row of 16 lanes, zero inflow (the cyclic-reduction levels of the chain factorisation:
  //      one DPP move per dword, like the one-lane shifts)
  template <int D>
  static __device__ __forceinline__ double rshr(double a) { return dpp_shift<0x110 + D>(a); }      // lane i <- lane i - D
  template <int D>
  static __device__ __forceinline__ double rshl(double a) { return dpp_shift<0x100 + D>(a); }      // lane i <- lane i + D
  // position p of the lane in its row is eliminated at the cyclic-reduction level of distance D (1, 2, 4, 8) iff
  // p = 15 - D mod 2D: level by level the odd positions counted from the row's END go, position 15 survives them all
  template <int D>
  static __device__ __forceinline__ bool cr_elim() { return (((threadIdx.x & 15) + D + 1) & (2 * D - 1)) == 0; }

  // ---- row-pair exchanges of the 32-lane chains' cyclic reduction (a chain is two rows of 16 lanes: rows 0|1, rows 2|3)
  // from_odd_row(a): every lane gets a of the same position in the ODD row of its pair; from_even_row(a): ... in the EVEN
  // row.  v_permlane16_swap on (a, a) produces both at once, one instruction per dword.
  static __device__ __forceinline__ double from_even_row(double a) {
    auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(a), false, false);
    auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(a), false, false);
    return __hiloint2double((int)hi[0], (int)lo[0]);
  }
  static __device__ __forceinline__ double from_odd_row(double a) {
    auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(a), false, false);
    auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(a), false, false);
    return __hiloint2double((int)hi[1], (int)lo[1]);
  }
  // bcast15(a): the lanes of the odd rows get a of lane 15 of the row below (DPP row_bcast:15, rows 1 and 3), all others 0
  static __device__ __forceinline__ double bcast15(double a) {
    int lo = __builtin_amdgcn_update_dpp(0, __double2loint(a), 0x142, 0xA, 0xf, false);
    int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(a), 0x142, 0xA, 0xf, false);
    return __hiloint2double(hi, lo);
  }
  // bcast31(a): lanes 32 .. 63 get a of lane 31 (DPP row_bcast:31), all others 0
  static __device__ __forceinline__ double bcast31(double a) {
    int lo = __builtin_amdgcn_update_dpp(0, __double2loint(a), 0x143, 0xC, 0xf, false);
    int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(a), 0x143, 0xC, 0xf, false);
    return __hiloint2double(hi, lo);
  }
  // ---- junction of the two chains (Solver::end_to_mid / mid_to_end): lane C - 1 of an instance gets a of its lane 2C - 1, and
  // back.  What the other lanes get is not used (the caller masks).  C = 16: up the rows with row_bcast:15 (one move per
  // dword); down them there is no single move, so row 1 is reversed and the wavefront shifted by one lane (two moves per
  // dword - without the pre-mask, the zero inflow and the second select of the general form).  C = 32: the half swap / row_bcast:31.
  static constexpr bool junction_moves = (C != G);
  static __device__ __forceinline__ double end_to_mid(double a) {
    if constexpr (C == 16) return dpp_shift<DPP_WAVE_SHL1>(mirror(a)); else return from_upper(a);
  }
  static __device__ __forceinline__ double mid_to_end(double a) {
    if constexpr (C == 16) return bcast15(a); else return bcast31(a);
  }
  // Inclusive prefix sum along the lanes of an instance (lane order): four shifted adds inside the rows of 16, then the
  // total of the row below (G >= 32) and of the half below (G = 64) - 5 / 6 steps instead of G - 1 dependent ones.
  static __device__ __forceinline__ double gscan(double a) {
    a += rshr<1>(a); a += rshr<2>(a); a += rshr<4>(a); a += rshr<8>(a);
    if constexpr (G >= 32) a += bcast15(a);
    if constexpr (G == 64) a += bcast31(a);
    return a;
  }
  // lane roles of that scheme: the first row's survivor of each chain (position 15 of rows 0 and 2), and the lanes of the
  // second rows whose lower neighbour, at the level that eliminates them, is that survivor (positions 0, 1, 3, 7 of rows 1, 3)
  static __device__ __forceinline__ bool cr_low15() { return (threadIdx.x & 31) == 15; }
  static __device__ __forceinline__ bool cr_special() {
    const int p = threadIdx.x & 15;
    return (threadIdx.x & 16) != 0 && ((p & (p + 1)) == 0) && p != 15;      // p = 2^m - 1: 0, 1, 3, 7
  }

  // ---- a chain of FOUR rows inside the wavefront (<64, 64>; Solver::s2_rows_*): step r works on the survivor X of row r
  // (position 15), the survivor Y of row r + 1 and that row's lanes 0, 1, 3, 7 - the moves of LaneBlock's chains, none of which
  // leaves the wavefront here
  static __device__ __forceinline__ bool cr64_x(int r) { return (int)(threadIdx.x & 63) == 16 * r + 15; }
  static __device__ __forceinline__ bool cr64_special(int r) {
    const int p = threadIdx.x & 15;
    return (int)((threadIdx.x & 63) >> 4) == r + 1 && ((p & (p + 1)) == 0) && p != 15;
  }
  template <int NV> static __device__ __forceinline__ void cr_pull(int, const double* v, double* o) {
#pragma unroll
    for (int i = 0; i < NV; ++i) o[i] = __shfl(v[i], ((int)threadIdx.x + 16) & 63, 64);
  }
  template <int NV> static __device__ __forceinline__ void cr_push(int, const double* v, double* o) {
#pragma unroll
    for (int i = 0; i < NV; ++i) o[i] = __shfl(v[i], ((int)threadIdx.x - 16) & 63, 64);
  }
  template <int NV> static __device__ __forceinline__ void cr_down(int, const double* v, double* o) {
#pragma unroll
    for (int i = 0; i < NV; ++i) o[i] = dpp_shift<DPP_WAVE_SHL1>(v[i]);
  }
  template <int NV> static __device__ __forceinline__ void cr_bcast(int, const double* v, double* o) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {        // row_bcast:15 into rows 1 .. 3
      int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v[i]), 0x142, 0xE, 0xf, false);
      int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v[i]), 0x142, 0xE, 0xf, false);
      o[i] = __hiloint2double(hi, lo);
    }
  }

  // ---- half-wave exchange (G = 64, N + 1 <= 32: lanes 32..63 carry the inputs of the stage on lane - 32)
  // from_upper(a): every lane gets a of lane | 32;  from_lower(a): every lane gets a of lane & 31.
  // v_permlane32_swap on (a, a) produces both at once, one instruction per dword, no LDS.
  static __device__ __forceinline__ double from_upper(double a) {
    auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(a), false, false);
    auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(a), false, false);
    return __hiloint2double((int)hi[1], (int)lo[1]);
  }
  static __device__ __forceinline__ double from_lower(double a) {
    auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(a), false, false);
    auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(a), false, false);
    return __hiloint2double((int)hi[0], (int)lo[0]);
  }

  // Instance-wide all-reduce without LDS: a butterfly whose partner permutations are DPP modifiers inside a
  // row (quad_perm xor 1 / xor 2, row_half_mirror, row_mirror: after the first two steps the quads - then the
  // half rows - are uniform, so any lane of the other half is as good as the xor partner and the sums come out
  // bit-identical to the xor butterfly) and the gfx950 row / half-wave swaps above it.  With __shfl_xor every
  // step was a ds_bpermute round trip, ~100 cycles of exposed latency at one wave per SIMD.
  template <class F>
  static __device__ __forceinline__ double bfly(double a, F f) {
    a = f(a, dpp_shift<0xB1>(a));                       // quad_perm [1,0,3,2]
    a = f(a, dpp_shift<0x4E>(a));                       // quad_perm [2,3,0,1]
    a = f(a, dpp_shift<DPP_ROW_HALF_MIRROR>(a));
    a = f(a, dpp_shift<DPP_ROW_MIRROR>(a));
    if constexpr (G >= 32) {                            // rows 0|1 and 2|3
      auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(a), false, false);
      auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(a), false, false);
      a = f(__hiloint2double((int)hi[0], (int)lo[0]), __hiloint2double((int)hi[1], (int)lo[1]));
    }
    if constexpr (G == 64) a = f(from_lower(a), from_upper(a));
    return a;
  }
  static __device__ __forceinline__ double gmax(double a) { return bfly(a, [](double x, double y) { return max_raw_(x, y); }); }
  static __device__ __forceinline__ double gmin(double a) { return bfly(a, [](double x, double y) { return min_raw_(x, y); }); }
  static __device__ __forceinline__ double gsum(double a) { return bfly(a, [](double x, double y) { return x + y; }); }
  static __device__ __forceinline__ bool gany(bool m) {
    unsigned long long b = __ballot(m);
    if (G == 64) return b != 0ull;
    unsigned long long mine = (b >> (slot() * G)) & ((1ull << (G & 63)) - 1ull);
    return mine != 0ull;
  }
  static __device__ __forceinline__ bool wany(bool m) { return __ballot(m) != 0ull; }
  // number of lanes of the instance on which m holds (scalar unit: ballot + bit count, no lane arithmetic)
  static __device__ __forceinline__ double gcount(bool m) {
    unsigned long long b = __ballot(m);
    if (G < 64) b = (b >> (slot() * G)) & ((1ull << (G & 63)) - 1ull);
    return double(__popcll(b));
  }

  // "cold" per-lane storage in LDS (one wavefront per block, slot-major: a wave access is 64
  // consecutive doubles, conflict free) for values only needed at termination checks and in the
  // certificate, so that they do not occupy registers - or worse, scratch - inside the loops
  static constexpr int cold_slots = SLOTS;       // x 512 B per wave
  static __device__ __forceinline__ double* cold() {
    __shared__ double buf[cold_slots * 64];
    return buf;
  }
  static __device__ __forceinline__ void cold_put(int slot, double a) { cold()[slot * 64 + lane_id()] = a; }
  static __device__ __forceinline__ double cold_get(int slot) { return cold()[slot * 64 + lane_id()]; }
  // compiler-level fence: values parked before it are re-read after it, not kept in registers
  static __device__ __forceinline__ void fence() { asm volatile("" ::: "memory"); }
  // scheduling barrier: the instruction scheduler moves nothing across it (K1: every field is stored when it is formed)
  static __device__ __forceinline__ void sched_barrier() { __builtin_amdgcn_sched_barrier(0); }

  // Output rows (rowlen doubles per instance, instance-major in dst): the lanes drop their entries
  // into the wave's LDS buffer at their place in the row, then the wave writes the 64/G rows it owns
  // as one run of consecutive doubles.  fill(put) calls put(index in row, lane active, value).
  template <class F>
  static __device__ __forceinline__ void rows(double* dst, int rowlen, int inst, int n_inst, F fill) {
    double* buf = cold();
    __syncthreads();                              // one wave per block: orders the LDS accesses only
    // (rows are written at the very end of a kernel: the lane number is formed again here, so that nothing derived from
    //  threadIdx has to stay in a register through the solve for it)
    const int lane = lane_again(), sl = lane / G;
    const int mine = sl * rowlen;
    fill([&](int idx, bool ok, double v) { if (ok) buf[mine + idx] = v; });
    __syncthreads();
    const int inst0 = inst - sl;                  // first instance of this wave
    const int cnt = (n_inst - inst0 < per_wave ? n_inst - inst0 : per_wave) * rowlen;
    double* out = dst + (long)inst0 * rowlen;
    for (int i = lane; i < cnt; i += 64) out[i] = buf[i];
  }

  // The same for a wave whose instances are NOT consecutive (the packed tail kernel takes them from a list): each of the
  // 64 / G rows goes to its own instance's place.
  template <class F>
  static __device__ __forceinline__ void rows_any(double* dst, int rowlen, int inst, int n_inst, F fill) {
    double* buf = cold();
    __syncthreads();
    const int lane = lane_again(), sl = lane / G;
    const int mine = sl * rowlen;
    fill([&](int idx, bool ok, double v) { if (ok) buf[mine + idx] = v; });
    __syncthreads();
    for (int s = 0; s < per_wave; ++s) {
      const int inst_s = __shfl(inst, s * G);     // the instance of the wave's s-th group of lanes (wave-uniform)
      if (inst_s >= n_inst) continue;
      double* out = dst + (long)inst_s * rowlen;
      for (int i = lane; i < rowlen; i += 64) out[i] = buf[s * rowlen + i];
    }
  }

  static __device__ __forceinline__ double load(const double* p, int idx, bool ok, double dflt) {
    return ok ? p[idx] : dflt;
  }
  static __device__ __forceinline__ int loadi(const int* p, int idx, bool ok, int dflt) {
    return ok ? p[idx] : dflt;
  }
  // run f on the lanes where ok holds, as ONE divergent region (the masked stores inside it need no branch of their own)
  template <class F>
  static __device__ __forceinline__ void when(bool ok, F f) { if (ok) f(); }
  // the same reads without a branch around each: a lane that has nothing to read reads element 0 and drops it, so that
  // the loads of a gather go out back to back and are waited for once (K1 is made of such gathers; p[0] must exist)
  static __device__ __forceinline__ double gather(const double* p, int idx, bool ok, double dflt) {
    const double v = p[ok ? idx : 0];
    return ok ? v : dflt;
  }
  static __device__ __forceinline__ int gatheri(const int* p, int idx, bool ok, int dflt) {
    const int v = p[ok ? idx : 0];
    return ok ? v : dflt;
  }
  static __device__ __forceinline__ void store(double* p, int idx, bool ok, double a) {
    if (ok) p[idx] = a;
  }
  static __device__ __forceinline__ void storei(int* p, int idx, bool ok, int a) {
    if (ok) p[idx] = a;
  }
};

// ----------------------------------------------------------------------------------------------------------------
// Horizons above 63: one instance = one WORKGROUP of G = 128 / 256 threads (2 / 4 wavefronts), still one lane per stage.
// The same lane code (mpmpc_core.hpp: Solver) runs on it; what changes is how lanes talk: a neighbour may sit in another
// wavefront, so a lane exchange in STAGE order goes through an LDS buffer between two workgroup barriers, and an instance-wide
// reduction is the in-wave DPP butterfly followed by a combination of the 2 / 4 wave results that every lane forms in the
// same order (the result is bit-identical on all lanes - it decides block-uniform branches - and associates like the xor
// butterfly of the emulation, lane_emu.hpp: pairs of waves first).  Chains of the twisted factorisation: lanes [0, G/2) climb,
// lanes [G/2, G) descend (C = G / 2), exactly as for <64, 32>.  The one-lane shifts ALONG the chains - what the sequential
// sweeps are made of - avoid the barriers: at G = 128 a chain is a wavefront and they are DPP shifts in registers; at G = 256 a
// chain spans two wavefronts and the sweep is staged wavefront by wavefront (chain_shift; Solver::staged_sweep).  The reduced
// solvers (2 x 2 blocks) do without the sweeps: their chains of four / eight rows are factored by cyclic reduction (rshl / rshr /
// cr64_* / cr_pull .. below; Solver::kCR64).  No split layout, no packing; one wavefront per SIMD for the general solver - the
// price of a horizon the reference allows (src/MPC.py:73-74 has no limit) and a 64-lane wavefront does not hold.
// LDS per workgroup: SLOTS cold slots of G doubles (also the staging of the output rows) + nine exchange rows + the
// reduction scratch: 77 KB at G = 128 (two workgroups per CU), 154 KB at G = 256, passed as DYNAMIC shared memory (above the 64 KB static limit).
// CH: lanes of a chain of the reduced solvers' cyclic reduction - G / 2 (two chains that meet in the middle: the twisted
// factorisation of the one-stage layout) or G (ONE chain in stage order: the lanes underneath the two-stages-per-lane layout of
// horizons 128 .. 255, lane_pair.hpp).  XR: exchange rows - the widest batch of values one pass through LDS carries.
template <int G, int SLOTS = 66, int CH = G / 2, int XR = 9>
struct LaneBlock {
  static_assert(G == 128 || G == 256, "a workgroup of 2 or 4 wavefronts");
  static_assert(CH == G / 2 || CH == G, "two chains that meet in the middle, or one");
  static constexpr int split = CH;
  static constexpr int C = G / 2;
  using real = double;
  using mask = bool;
  using ival = int;
  static constexpr int group = G;
  static constexpr int per_wave = 1;          // ONE instance per execution group: instance-wide conditions are group-wide
  static constexpr int waves = G / 64;
  static constexpr int cold_slots = SLOTS;
  static constexpr bool batched = true;       // several values of a step share one pass through LDS (Solver::cup_n)
  static constexpr int stages_per_lane = 1;
  static constexpr bool junction_moves = false;
  static constexpr bool staged_sweeps = (G == 256);   // a chain spans two wavefronts: Solver::staged_sweep
  static constexpr int xrows = XR;            // exchange rows: the widest batch is the 3 x 3 block of a factorisation step (general solver)
  // ... + 8 doubles of reduction scratch + the edge values of a staged sweep (xrows per chain)
  static constexpr size_t lds_bytes = sizeof(double) * ((size_t)(SLOTS + xrows) * G + 8 + 2 * xrows);

  static __device__ __forceinline__ int lane_id() { return threadIdx.x; }
  static __device__ __forceinline__ int stage() { return threadIdx.x; }
  static __device__ __forceinline__ int slot() { return 0; }
  static __device__ __forceinline__ int lane_again() { return threadIdx.x; }
  static __device__ __forceinline__ int stage_again() { return threadIdx.x; }
  static __device__ __forceinline__ int slot_again() { return 0; }
  static __device__ __forceinline__ bool mtrue() { return true; }
  static __device__ __forceinline__ bool mfalse() { return false; }

  static __device__ __forceinline__ double* lds() {
    extern __shared__ double mpmpc_block_lds[];
    return mpmpc_block_lds;
  }
  static __device__ __forceinline__ double* cold() { return lds(); }
  static __device__ __forceinline__ double* xrow() { return lds() + (size_t)SLOTS * G; }
  static __device__ __forceinline__ double* rrow() { return lds() + (size_t)(SLOTS + xrows) * G; }
  static __device__ __forceinline__ double* erow() { return rrow() + 8; }
  static __device__ __forceinline__ void sync() { __syncthreads(); }
  // ---- staged sweeps (G = 256; Solver::staged_sweep).  Chain layout: chain 0 = wavefronts 0 | 1, chain 1 = wavefronts 2 | 3.
  // sweep_first(DIR): this lane's wavefront is the first of its chain in the direction of the sweep (DIR -1: values flow
  // towards higher lanes - the even wavefronts; DIR +1: the odd ones).
  static __device__ __forceinline__ bool sweep_first(int dir) { return ((threadIdx.x >> 6) & 1) == (dir < 0 ? 0 : 1); }
  // chain_shift<NV, DIR, MODE>: the one-lane shift along the chains inside the wavefront (DPP, zero inflow).  MODE 1 (first
  // wavefront): its edge lane also leaves v - what the next lane of the chain would take - in LDS.  MODE 2 (second
  // wavefront): its edge lane takes that instead of the zero.  No barrier: the sweep places them.
  // OFF: first of the NV edge slots this shift uses.  A step of a staged sweep that shifts TWICE must give its shifts disjoint slot
  // ranges - with the same range the second would overwrite what the first handed on before the other wavefront has read it
  // (the sweeps of the solver shift once per step: OFF = 0).
  template <int NV, int DIR, int MODE, int OFF = 0>
  static __device__ __forceinline__ void chain_shift(const double* v, double* o) {
    static_assert(G == 256 && OFF >= 0 && OFF + NV <= xrows && (MODE == 1 || MODE == 2), "staged sweeps: chains of two wavefronts, edge slots within a chain's row");
    const int lane = threadIdx.x & 63, chain = threadIdx.x >> 7;
    double* e = erow() + chain * xrows + OFF;
#pragma unroll
    for (int i = 0; i < NV; ++i) o[i] = DIR < 0 ? dpp_shift<DPP_WAVE_SHR1>(v[i]) : dpp_shift<DPP_WAVE_SHL1>(v[i]);
    const bool edge_out = DIR < 0 ? lane == 63 : lane == 0, edge_in = DIR < 0 ? lane == 0 : lane == 63;
    if constexpr (MODE == 1) {
      if (edge_out) {
#pragma unroll
        for (int i = 0; i < NV; ++i) e[i] = v[i];
      }
    } else {
      if (edge_in) {
#pragma unroll
        for (int i = 0; i < NV; ++i) o[i] = e[i];
      }
    }
  }

  // the value lane `src` holds (src outside [0, G): 0) - every lane of the workgroup must call it
  static __device__ __forceinline__ double take(double a, int src) {
    double* x = xrow();
    x[threadIdx.x] = a;
    __syncthreads();
    const double r = (unsigned)src < (unsigned)G ? x[src] : 0.0;
    __syncthreads();          // (the row is free again before anyone writes the next exchange into it)
    return r;
  }
  // ... NV values at once: one barrier pair for all of them
  template <int NV>
  static __device__ __forceinline__ void takev(const double* v, double* o, int src) {
    static_assert(NV <= xrows, "widen the exchange rows");
    double* x = xrow();
#pragma unroll
    for (int i = 0; i < NV; ++i) x[i * G + threadIdx.x] = v[i];
    __syncthreads();
    const bool ok = (unsigned)src < (unsigned)G;
#pragma unroll
    for (int i = 0; i < NV; ++i) o[i] = ok ? x[i * G + src] : 0.0;
    __syncthreads();
  }
  template <int NV> static __device__ __forceinline__ void upv(const double* v, double* o) { takev<NV>(v, o, (int)threadIdx.x - 1); }
  template <int NV> static __device__ __forceinline__ void downv(const double* v, double* o) { takev<NV>(v, o, (int)threadIdx.x + 1); }
  // G = 128: a chain of the twisted factorisation IS a wavefront (lanes [0, 64) climb, lanes [64, 128) descend), so the
  // one-lane shifts along the chains - what the sequential sweeps of a factorisation or a solve are made of, some hundred per
  // interior-point iteration - are wavefront shifts in registers (DPP, zero inflow at lane 0 / lane 63 = the chain heads) and
  // need no barrier at all.  G = 256: a chain spans two wavefronts, the shift goes through LDS.
  template <int NV> static __device__ __forceinline__ void cupv(const double* v, double* o) {
    if constexpr (G == 128) {
#pragma unroll
      for (int i = 0; i < NV; ++i) o[i] = dpp_shift<DPP_WAVE_SHR1>(v[i]);
    } else {
      const int t = threadIdx.x;
      takev<NV>(v, o, (t == 0 || t == C) ? -1 : t - 1);
    }
  }
  template <int NV> static __device__ __forceinline__ void cdownv(const double* v, double* o) {
    if constexpr (G == 128) {
#pragma unroll
      for (int i = 0; i < NV; ++i) o[i] = dpp_shift<DPP_WAVE_SHL1>(v[i]);
    } else {
      const int t = threadIdx.x;
      takev<NV>(v, o, (t == C - 1 || t == 2 * C - 1) ? -1 : t + 1);
    }
  }
  static __device__ __forceinline__ double up(double a) { return take(a, (int)threadIdx.x - 1); }
  static __device__ __forceinline__ double down(double a) { return take(a, (int)threadIdx.x + 1); }
  // chain layout: lanes [C, 2C) reversed; one-lane shifts along the chains with zero inflow at the chain heads
  static __device__ __forceinline__ double mirror(double a) {
    const int t = threadIdx.x;
    if constexpr (G == 128) {
      // (the second wavefront reverses itself: a lane permutation inside the wave, ds_bpermute, no barrier)
      if (t >= 64) a = __shfl(a, 127 - t, 64);
      return a;
    } else {
      return take(a, t < C ? t : 3 * C - 1 - t);
    }
  }
  template <int NV> static __device__ __forceinline__ void mirrorv(const double* v, double* o) {
    if constexpr (G == 128) {
#pragma unroll
      for (int i = 0; i < NV; ++i) o[i] = mirror(v[i]);
    } else {
      const int t = threadIdx.x;
      takev<NV>(v, o, t < C ? t : 3 * C - 1 - t);
    }
  }
  static __device__ __forceinline__ double cup(double a) {
    if constexpr (G == 128) return dpp_shift<DPP_WAVE_SHR1>(a);
    const int t = threadIdx.x;
    return take(a, (t == 0 || t == C) ? -1 : t - 1);
  }
  static __device__ __forceinline__ double cdown(double a) {
    if constexpr (G == 128) return dpp_shift<DPP_WAVE_SHL1>(a);
    const int t = threadIdx.x;
    return take(a, (t == C - 1 || t == 2 * C - 1) ? -1 : t + 1);
  }

  // ---- cyclic reduction of the chains (G = 128: a chain is a wavefront of four rows; G = 256: two wavefronts, eight rows;
  // Solver::kCR64).  The in-row shifts and masks of the levels are the wavefront backend's; the survivors of the rows meet
  // through moves by whole rows.
  template <int D> static __device__ __forceinline__ double rshr(double a) { return dpp_shift<0x110 + D>(a); }
  template <int D> static __device__ __forceinline__ double rshl(double a) { return dpp_shift<0x100 + D>(a); }
  template <int D> static __device__ __forceinline__ bool cr_elim() { return (((threadIdx.x & 15) + D + 1) & (2 * D - 1)) == 0; }
  // position of a lane in its chain = lane % C; step r works on the survivor X of row r (position 15), the survivor Y of row
  // r + 1 and that row's lanes 0, 1, 3, 7.  At G = 256 (chains of eight rows over two wavefronts) step 3 crosses the wavefronts.
  static __device__ __forceinline__ bool cr64_x(int r) { return (int)(threadIdx.x & (CH - 1)) == 16 * r + 15; }
  static __device__ __forceinline__ bool cr64_special(int r) {
    const int p = threadIdx.x & 15;
    return (int)((threadIdx.x & (CH - 1)) >> 4) == r + 1 && ((p & (p + 1)) == 0) && p != 15;
  }
  // (a chain of more than four rows spans wavefronts: the step from the last row of one to the first row of the next crosses)
  static __device__ __forceinline__ bool cr_crosses(int r) { return CH > 64 && (r & 3) == 3; }
  // pull: every lane gets v of the same position one row up the chain (lane + 16); push: one row down (lane - 16); down: of
  // the next lane; bcast: of position 15 of the row below.  Inside a wavefront a lane permutation / DPP move per value; the
  // crossing step batches its values through the exchange rows (what a lane without such a source gets is not used).
  template <int NV> static __device__ __forceinline__ void cr_pull(int r, const double* v, double* o) {
    if (cr_crosses(r)) { takev<NV>(v, o, (int)threadIdx.x + 16); return; }
#pragma unroll
    for (int i = 0; i < NV; ++i) o[i] = __shfl(v[i], ((int)threadIdx.x + 16) & 63, 64);
  }
  template <int NV> static __device__ __forceinline__ void cr_push(int r, const double* v, double* o) {
    if (cr_crosses(r)) { takev<NV>(v, o, (int)threadIdx.x - 16); return; }
#pragma unroll
    for (int i = 0; i < NV; ++i) o[i] = __shfl(v[i], ((int)threadIdx.x - 16) & 63, 64);
  }
  template <int NV> static __device__ __forceinline__ void cr_down(int r, const double* v, double* o) {
    if (cr_crosses(r)) { takev<NV>(v, o, (int)threadIdx.x + 1); return; }
#pragma unroll
    for (int i = 0; i < NV; ++i) o[i] = dpp_shift<DPP_WAVE_SHL1>(v[i]);
  }
  template <int NV> static __device__ __forceinline__ void cr_bcast(int r, const double* v, double* o) {
    if (cr_crosses(r)) { takev<NV>(v, o, (((int)threadIdx.x & ~15) - 16) | 15); return; }
#pragma unroll
    for (int i = 0; i < NV; ++i) {        // row_bcast:15 into rows 1 .. 3 of the wavefront
      int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v[i]), 0x142, 0xE, 0xf, false);
      int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v[i]), 0x142, 0xE, 0xf, false);
      o[i] = __hiloint2double(hi, lo);
    }
  }

  template <class F>
  static __device__ __forceinline__ double reduce(double a, F f) {
    a = LaneGpu<64, 32, 1>::bfly(a, f);          // wave-uniform
    double* r = rrow();
    if ((threadIdx.x & 63) == 0) r[threadIdx.x >> 6] = a;
    __syncthreads();
    double t = f(r[0], r[1]);
    if constexpr (waves == 4) t = f(t, f(r[2], r[3]));
    __syncthreads();
    return t;
  }
  static __device__ __forceinline__ double gmax(double a) { return reduce(a, [](double x, double y) { return max_(x, y); }); }
  static __device__ __forceinline__ double gmin(double a) { return reduce(a, [](double x, double y) { return min_(x, y); }); }
  static __device__ __forceinline__ double gsum(double a) { return reduce(a, [](double x, double y) { return x + y; }); }
  // inclusive prefix sum along the lanes of the workgroup: the 64-lane scan of the wavefront backend, then the totals of the
  // wavefronts below, added one after the other (lane_emu.hpp: gscan adds in the same order)
  static __device__ __forceinline__ double gscan(double a) {
    a = LaneGpu<64, 32, 1>::gscan(a);
    double* r = rrow();
    if ((threadIdx.x & 63) == 63) r[threadIdx.x >> 6] = a;
    __syncthreads();
    double off = 0.0;
    const int w = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < waves - 1; ++j) off = (j < w) ? off + r[j] : off;
    __syncthreads();
    return a + off;
  }
  static __device__ __forceinline__ bool gany(bool m) { return __syncthreads_or(m ? 1 : 0) != 0; }
  static __device__ __forceinline__ bool wany(bool m) { return __syncthreads_or(m ? 1 : 0) != 0; }
  static __device__ __forceinline__ double gcount(bool m) { return double(__syncthreads_count(m ? 1 : 0)); }

  static __device__ __forceinline__ void cold_put(int slot, double a) { cold()[slot * G + threadIdx.x] = a; }
  static __device__ __forceinline__ double cold_get(int slot) { return cold()[slot * G + threadIdx.x]; }
  static __device__ __forceinline__ void fence() { asm volatile("" ::: "memory"); }
  static __device__ __forceinline__ void sched_barrier() { __builtin_amdgcn_sched_barrier(0); }

  // output row of the workgroup's instance: staged in LDS (the cold slots: the solve is over), written as one run
  template <class F>
  static __device__ __forceinline__ void rows(double* dst, int rowlen, int inst, int n_inst, F fill) {
    double* buf = cold();
    __syncthreads();
    fill([&](int idx, bool ok, double v) { if (ok) buf[idx] = v; });
    __syncthreads();
    if (inst < n_inst) {
      double* out = dst + (long)inst * rowlen;
      for (int i = threadIdx.x; i < rowlen; i += G) out[i] = buf[i];
    }
  }
  template <class F>
  static __device__ __forceinline__ void rows_any(double* dst, int rowlen, int inst, int n_inst, F fill) { rows(dst, rowlen, inst, n_inst, fill); }

  static __device__ __forceinline__ double load(const double* p, int idx, bool ok, double dflt) { return ok ? p[idx] : dflt; }
  static __device__ __forceinline__ int loadi(const int* p, int idx, bool ok, int dflt) { return ok ? p[idx] : dflt; }
  template <class F>
  static __device__ __forceinline__ void when(bool ok, F f) { if (ok) f(); }
  static __device__ __forceinline__ double gather(const double* p, int idx, bool ok, double dflt) {
    const double v = p[ok ? idx : 0];
    return ok ? v : dflt;
  }
  static __device__ __forceinline__ int gatheri(const int* p, int idx, bool ok, int dflt) {
    const int v = p[ok ? idx : 0];
    return ok ? v : dflt;
  }
  static __device__ __forceinline__ void store(double* p, int idx, bool ok, double a) { if (ok) p[idx] = a; }
  static __device__ __forceinline__ void storei(int* p, int idx, bool ok, int a) { if (ok) p[idx] = a; }
};

}  // namespace mpmpc
