// libmpmpc.so: HIP kernels for gfx950 (MI355X) + the C ABI of include/mpmpc.h.
//
//   K1  mpmpc_assemble_kernel   one thread per (instance, stage): gathers waypoint data from
//                               LDS-staged path tables, builds the stage's dynamics blocks,
//                               offsets, bounds and cost terms, stores them stage-blocked
//                               ([field][instance][stage], unit-stride across lanes).  HBM bound.
//   K2  mpmpc_solve_kernel      the general solve kernel, one 64-lane wavefront per instance, lane = horizon stage;
//                               Ruiz scaling, OSQP ADMM, certified polish, phase 1, all state in VGPRs,
//                               stage coupling by DPP shifts, norms by wavefront reductions.
//                               FP64-VALU issue bound; builds its QP per lane with
//                               K1's code and keeps it in registers: HBM sees the batch inputs and the solution.
//   K2r mpmpc_reduced_kernel    the batch path of the reference's own weights: the (e_y, e_psi, kappa) problem only, 1 / 2 / 4
//                               instances per wavefront, <= 256 registers and 20 KB of LDS: two wavefronts per SIMD
//   K2t mpmpc_reduced_t_kernel  its twin for a terminal cost on the time state (rank-one term, Sherman-Morrison)
//   K2p mpmpc_reduced_tail_kernel  the tail of a K2r launch (infeasible / marginal / capped instances) on the same footing:
//                               phase 1 and one more attempt; K2 takes what it leaves
//   K2b / K2rb                  horizons 64 .. 255: the general and the reduced-native solver on a WORKGROUP of 2 / 4 wavefronts
//                               per instance (lane_gpu.hpp: LaneBlock)
//   K0 / K3 / K4                corridor tables from the map, closed-loop rollout, speed profile (see below).
//
// No CPU path exists in this library: every compute entry point needs a HIP device.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#define MPMPC_HD __device__ __forceinline__
#define MPMPC_HOST_DEVICE __host__ __device__
#ifdef MPMPC_PHASE_CLOCK
// profiling builds only (profiles/phases.sh): per-wave time of the phases of K2 in 10 ns ticks
__device__ long long g_phase[4096 * 32];
#define MPMPC_TICK_BEGIN(i) do { if ((threadIdx.x & 63) == 0 && blockIdx.x < 4096) g_phase[blockIdx.x * 32 + (i)] -= wall_clock64(); } while (0)
#define MPMPC_TICK_END(i) do { if ((threadIdx.x & 63) == 0 && blockIdx.x < 4096) g_phase[blockIdx.x * 32 + (i)] += wall_clock64(); } while (0)
#define MPMPC_TICK_COUNT(i) do { if ((threadIdx.x & 63) == 0 && blockIdx.x < 4096) g_phase[blockIdx.x * 32 + (i)] += 1; } while (0)
#endif
#include "lane_gpu.hpp"
#include "lane_pair.hpp"
#include "mpmpc_core.hpp"
#include "mpmpc_reduced.hpp"
#include "mpmpc_reduced_t.hpp"
#include "mpmpc_reduced_tail.hpp"
#include "corridor_core.hpp"
#include "rollout_core.hpp"
#include "speed_core.hpp"

using namespace mpmpc;

// ------------------------------------------------------------------------------------ kernels
constexpr int K1_THREADS = 256;
constexpr int K1_LDS_WP = 1024;   // path tables of up to this many waypoints are staged in LDS

// NT: non-temporal stores - the default (MPMPC_K1_NT=0: plain stores).  Where the kernel stands (profiles/r4/k1_occupancy.txt,
// same-box medians): 71 - 76 % of the HBM peak at B = 65 536 (550 MB per launch), 59 - 62 % at B = 8 192.  Two things it took:
// rows written to the end of their last 128-byte line (N = 30 fills 31 of a row's 32 doubles; with partial lines the kernel
// reached 39 %, and non-temporal stores were SLOWER than plain ones), and then non-temporal stores (plain ones: 56 - 58 %).
// That is the rate of the store pattern alone (profiles/micro/k1_pattern.hip: 77 %).  Off the solve path: the solve launches
// build their QP in registers with the same code.
// (-DMPMPC_K1_WAVES=w: occupancy experiments, profiles/k1_occupancy.py)
#ifdef MPMPC_K1_WAVES
#define MPMPC_K1_OCC __attribute__((amdgpu_waves_per_eu(MPMPC_K1_WAVES, MPMPC_K1_WAVES)))
#else
#define MPMPC_K1_OCC
#endif
template <bool NT>
__global__ __launch_bounds__(K1_THREADS) MPMPC_K1_OCC void mpmpc_assemble_kernel(
    mpmpc_config cfg, PathTables tab, int B, int ld, const int* __restrict__ wp_id, const double* __restrict__ x0,
    const double* __restrict__ cc, const double* __restrict__ lb, const double* __restrict__ ub,
    double* __restrict__ qp) {
  // (dynamic LDS: 3 x n_wp doubles when the tables are staged - a 200-waypoint lap takes 4.8 KB of a block, not 24)
  extern __shared__ double s_tab[];
  double *s_kappa = s_tab, *s_vref = s_tab + tab.n_wp, *s_ds = s_tab + 2 * tab.n_wp;
  PathTables t = tab;
  if (tab.n_wp <= K1_LDS_WP) {   // block-uniform
    for (int i = threadIdx.x; i < tab.n_wp; i += K1_THREADS) {
      s_kappa[i] = tab.kappa[i];
      s_vref[i] = tab.v_ref[i];
      s_ds[i] = tab.ds_next[i];
    }
    __syncthreads();
    t.kappa = s_kappa;
    t.v_ref = s_vref;
    t.ds_next = s_ds;
  }
  // one thread per (instance, stage) pair, stage fastest: a wavefront writes 64 consecutive doubles of one field.
  // NO grid-stride loop: around a loop the compiler keeps every constant of the stage (the tangent's polynomial, the
  // configuration's doubles) in registers - 109 of them, four waves per SIMD; without it 48 and eight waves per SIMD,
  // and what bounds this kernel is how many waves have their stores in flight (profiles/micro/k1_pattern.hip: the store
  // pattern alone reaches 77 % of the HBM peak)
  using L = LaneGpu<64>;
  const int total = B * ld;
  const int g = blockIdx.x * K1_THREADS + threadIdx.x;
  if (g < total) {
    const int inst = g / ld, k = g - inst * ld;
    if constexpr (!NT) {
      assemble_lane<L>(cfg, t, B, ld, inst, k, wp_id, x0, cc, lb, ub, qp);
    } else {
      double f0[MPMPC_NUM_FIELDS];
      assemble_fields<L>(cfg, t, B, inst, k, wp_id, x0, cc, lb, ub, f0);
      // (rows written to the end of their last 128-byte line, like assemble_lane)
      const int kfill = ((cfg.N + 1 + 15) / 16) * 16 < ld ? ((cfg.N + 1 + 15) / 16) * 16 : ld;
      if (inst < B && k < kfill) {
        double* base = qp + (size_t)inst * ld + k;
        MPMPC_UNROLL
        for (int f = 0; f < MPMPC_NUM_FIELDS; ++f) __builtin_nontemporal_store(k <= cfg.N ? f0[f] : 0.0, base + f * ((size_t)B * ld));
      }
    }
  }
}

// mode 0: the whole solve (early polish attempt, then the full OSQP run of what it could not certify).
// mode 1: the early attempt only; the ids of uncertified instances are appended to tail[1..], tail[0] counts.
// mode 2: the full run only, on the instances listed in tail (one per wave).
// A packed launch (2 or 4 instances per wave) uses modes 1 + 2: the few instances that need hundreds or
// thousands of ADMM iterations then run one per wave, on the faster G = 64 code, instead of holding a
// packed wave (and its finished partner lanes) for the whole tail.
// The launch assembles its own QP: every lane builds the 27 fields of its (instance, stage) with K1's code
// (assemble_fields) and goes on with them in registers - the stage-blocked QP is never written to or read from
// memory on this path (mpmpc_assemble, the parity / debug export, is what runs K1 proper).
struct AssembleIn {
  PathTables tab;
  const int* wp_id;
  const double *x0, *cc, *lb, *ub;
};
// VAR: 0 = the full problem with diagonal weights, 1 = full weights (FQ: Q, R, QN with off-diagonal entries), 2 = reduced polish (RED),
//      3 = the full problem whose states e_psi and t have no bounds (FREEX; one-instance-per-wave kernels only)
template <int G, int C, bool WARM, int VAR = 0>
__global__ __launch_bounds__(64) void mpmpc_solve_kernel(mpmpc_config cfg, SolverParams st, int B, int ld,
                                                         AssembleIn ain, double* __restrict__ z,
                                                         double* __restrict__ u0, int* __restrict__ status,
                                                         int* __restrict__ iters, double* __restrict__ resid,
                                                         double* __restrict__ y, int mode, int* __restrict__ tail,
                                                         int* __restrict__ act, const int* __restrict__ shift,
                                                         int* __restrict__ tail_reset) {
  using L = LaneGpu<G, C>;
  int inst = blockIdx.x * L::per_wave + L::slot();
  if (mode == 2) {
    // the list the NEXT launch on this handle will fill starts empty (two lists, used in turn: no memset between launches)
    if (blockIdx.x == 0 && threadIdx.x == 0) *tail_reset = 0;
    if ((int)blockIdx.x >= tail[0]) return;       // wave-uniform: G = 64 here
    inst = tail[1 + blockIdx.x];
  }
  const int k = L::stage() - lane_offset(G, C, cfg.N);      // stage of this lane (negative / > N: none)
  // closed loop: the active set the previous step certified for this car, moved on by the waypoints it advanced
  int guess = 0;
  if (WARM && act && shift && inst < B && k >= 0 && k <= cfg.N) {
    int kk = k + shift[inst];
    guess = act[inst * ld + (kk > cfg.N ? cfg.N : kk)];
  }
  MPMPC_TICK_BEGIN(8);
  double fields[MPMPC_NUM_FIELDS];
  assemble_fields<L>(cfg, ain.tab, B, inst, k, ain.wp_id, ain.x0, ain.cc, ain.lb, ain.ub, fields);
  Solver<L, VAR == 1, VAR == 2, VAR == 3, VAR == 2> s;   // (the reduced variant factors its 16-lane chains by cyclic reduction, like the reduced-native kernels)
  // (second launch of a packed batch: the interior-point iterations the first launch spent on this instance)
  const int base_ipm = (mode == 2 && iters) ? iters[inst * 2 + 1] : 0;
  static_assert(G == 64, "the general kernels run one instance per wave (only the reduced-native kernels pack)");
  double woff[7];
  weight_offdiag(cfg, woff);
  s.template run<WARM, true>(fields, B, inst, k, cfg.N, st, mode, guess, base_ipm, VAR == 1 ? woff : nullptr);
  MPMPC_TICK_BEGIN(7);
  s.store(inst, k, cfg.wheelbase, z, u0, status, iters, resid, y, WARM ? act : nullptr, ld);
  MPMPC_TICK_END(7);
  MPMPC_TICK_END(8);
  if (mode == 1 && k == 0 && inst < B && s.status == MPMPC_UNSOLVED) tail[1 + atomicAdd(tail, 1)] = inst;
}

// K2b: the general solve kernel for horizons above 63 - one instance per WORKGROUP of G = 128 / 256 threads (2 / 4
// wavefronts), one lane per stage as before; the lanes of different wavefronts talk through LDS (lane_gpu.hpp: LaneBlock).
// Same Solver code, the whole solve in one launch (mode 0), cold starts.  VAR as for mpmpc_solve_kernel: 0 = the full problem,
// 1 = full weight matrices, 2 = the polish on the (e_y, e_psi, kappa) problem where the time state separates (the reference's
// own weights: 2 x 2 blocks in the sweeps instead of 3 x 3, three entries per lane instead of five, fewer iterations).
template <int G, int VAR>
__global__ __launch_bounds__(G) void mpmpc_solve_block_kernel(mpmpc_config cfg, SolverParams st, int B, AssembleIn ain,
                                                              double* __restrict__ z, double* __restrict__ u0,
                                                              int* __restrict__ status, int* __restrict__ iters,
                                                              double* __restrict__ resid, double* __restrict__ y,
                                                              const int* __restrict__ tail) {
  using L = LaneBlock<G>;
  // tail: the instances the reduced-native workgroup kernel (below) could not certify - one workgroup each, straight to
  // phase 1 and the full iteration (mode 2, as for the wavefront kernels); null: every instance, the whole solve
  if (tail && (int)blockIdx.x >= tail[0]) return;
  const int inst = tail ? tail[1 + blockIdx.x] : (int)blockIdx.x;
  const int k = L::stage() - lane_offset(G, G / 2, cfg.N);
  double fields[MPMPC_NUM_FIELDS];
  assemble_fields<L>(cfg, ain.tab, B, inst, k, ain.wp_id, ain.x0, ain.cc, ain.lb, ain.ub, fields);
  // (the reduced variant factors its chains of four / eight rows by cyclic reduction: Solver::kCR64)
  Solver<L, VAR == 1, VAR == 2, false, VAR == 2> s;
  double woff[7];
  weight_offdiag(cfg, woff);
  const int base_ipm = (tail && iters) ? iters[inst * 2 + 1] : 0;       // (the interior-point iterations the first kernel spent on it)
  s.template run<false, true>(fields, B, inst, k, cfg.N, st, tail ? 2 : 0, 0, base_ipm, VAR == 1 ? woff : nullptr);
  s.store(inst, k, cfg.wheelbase, z, u0, status, iters, resid, y, nullptr, 0);
}

// K2rb: the reduced-native solver (mpmpc_reduced.hpp: the (e_y, e_psi, kappa) problem from the start - own scaling, interior
// point from x = 0, active-set rounds, certificate) on a workgroup, for the reference's own weights at horizons above 63.
// What it cannot certify is listed in `tail` for mpmpc_solve_block_kernel<G, 2>.
// (40 cold slots like K2r instead of the general solver's 66: 50 KB of LDS per workgroup at G = 128 - three workgroups per CU)
constexpr int RNB_SLOTS = 40;
template <int G>
__global__ __launch_bounds__(G) void mpmpc_reduced_block_kernel(mpmpc_config cfg, SolverParams st, int B, AssembleIn ain,
                                                                double* __restrict__ z, double* __restrict__ u0,
                                                                int* __restrict__ status, int* __restrict__ iters,
                                                                double* __restrict__ resid, double* __restrict__ y,
                                                                int* __restrict__ tail) {
  using L = LaneBlock<G, RNB_SLOTS>;
  const int inst = blockIdx.x;
  const int k = L::stage() - lane_offset(G, G / 2, cfg.N);
  double fields[MPMPC_NUM_FIELDS];
  assemble_fields<L>(cfg, ain.tab, B, inst, k, ain.wp_id, ain.x0, ain.cc, ain.lb, ain.ub, fields);
  ReducedSolver<L> s;
  s.template run<false>(fields, B, inst, k, cfg.N, st, 0);
  s.store(inst, k, cfg.wheelbase, z, u0, status, iters, resid, y, nullptr, 0);
  if (k == 0 && inst < B && s.status == MPMPC_UNSOLVED) tail[1 + atomicAdd(tail, 1)] = inst;
}

// K2r: the reduced-native solve kernel (mpmpc_reduced.hpp) - the batch path of every configuration whose time state
// separates (the reference's own weights).  One launch assembles (K1's code, in registers), solves and stores; the ids of
// the instances it cannot certify are appended to tail[1..] (tail[0] counts) for the general kernel in mode 2.
// 40 LDS slots (20 KB) and at most 256 registers: two wavefronts per SIMD.
constexpr int RN_SLOTS = 40;
// WARM (closed loop): act [B x ld] holds the active sets the previous step certified, shift [B] the waypoints each car has
// advanced since; the kernel starts from them and leaves this step's sets in act.
template <int G, int C, bool WARM>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void mpmpc_reduced_kernel(mpmpc_config cfg, SolverParams st, int B, int ld, AssembleIn ain,
                                                           double* __restrict__ z, double* __restrict__ u0,
                                                           int* __restrict__ status, int* __restrict__ iters,
                                                           double* __restrict__ resid, double* __restrict__ y,
                                                           int* __restrict__ tail, int* __restrict__ act,
                                                           const int* __restrict__ shift, int* __restrict__ tail_reset,
                                                           unsigned* __restrict__ tail_flag, unsigned seq, int* __restrict__ tail2_reset) {
  using L = LaneGpu<G, C, RN_SLOTS>;
  // (the list of the NEXT launch is emptied here as well: the tail launch, which does it too, may be deferred - see
  //  launch_solve; and so is the list the reduced-native tail kernel of THIS launch appends to)
  if (blockIdx.x == 0 && threadIdx.x == 0) { *tail_reset = 0; *tail2_reset = 0; }
  const int inst = blockIdx.x * L::per_wave + L::slot();
  const int k = L::stage() - lane_offset(G, C, cfg.N);
  int guess = 0;
  if (WARM && inst < B && k >= 0 && k <= cfg.N) {
    const int kk = k + shift[inst];
    guess = act[inst * ld + (kk > cfg.N ? cfg.N : kk)];
  }
  MPMPC_TICK_BEGIN(8);
  double fields[MPMPC_NUM_FIELDS];
  assemble_fields<L>(cfg, ain.tab, B, inst, k, ain.wp_id, ain.x0, ain.cc, ain.lb, ain.ub, fields);
  ReducedSolver<L> s;
  s.template run<WARM>(fields, B, inst, k, cfg.N, st, guess);
  MPMPC_TICK_BEGIN(7);
  // (instance and stage are formed again rather than kept in registers through the solve)
  const int inst_o = blockIdx.x * L::per_wave + L::slot_again();
  const int k_o = L::stage_again() - lane_offset(G, C, cfg.N);
  s.store(inst_o, k_o, cfg.wheelbase, z, u0, status, iters, resid, y, WARM ? act : nullptr, ld);
  MPMPC_TICK_END(7);
  MPMPC_TICK_END(8);
  if (k_o == 0 && inst_o < B && s.status == MPMPC_UNSOLVED) {
    tail[1 + atomicAdd(tail, 1)] = inst_o;
    // "this launch left a tail": the launch's sequence number, in host memory the device writes through (read by the host
    // after the stream has drained: observe_tail)
    __hip_atomic_store(tail_flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// K2r2: K2r with TWO STAGES PER LANE (lane_pair.hpp): an instance of 17 .. 32 stages takes 16 lanes, a
// wavefront carries FOUR.  The same ReducedSolver on the pair backend: every elementwise operation is issued for both stages of
// the lane (the work per stage is what it was), a stage-order shift moves one of the two components across lanes, a wave
// reduction serves four instances, and the cyclic reduction eliminates the even stages inside the lanes before its four
// cross-lane levels run on the survivors (mpmpc_solver_s2.hpp).  Twice the state per lane: up to 512 registers and 80 LDS slots
// (40 KB), ONE wavefront per SIMD - the two independent stages of a lane stand in for the second wave.  Cold starts only (the
// closed loop keeps K2r); same launch contract as K2r, same tail lists.
// GB = 64: an instance of 65 .. 128 stages (horizons 64 .. 127) in ONE wavefront - a chain of four rows, no LDS exchange and
// no workgroup barrier, where the one-stage layout takes a workgroup of two wavefronts (K2rb); what it cannot certify goes to
// mpmpc_solve_block_kernel<128, 2> through the same list.
constexpr int RN2_SLOTS = 80;
template <int GB>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void mpmpc_reduced_pair_kernel(mpmpc_config cfg, SolverParams st, int B, int ld, AssembleIn ain,
                                                           double* __restrict__ z, double* __restrict__ u0,
                                                           int* __restrict__ status, int* __restrict__ iters,
                                                           double* __restrict__ resid, double* __restrict__ y,
                                                           int* __restrict__ tail, int* __restrict__ tail_reset,
                                                           unsigned* __restrict__ tail_flag, unsigned seq, int* __restrict__ tail2_reset) {
  using L = LanePair<LaneGpu<GB, GB, RN2_SLOTS>>;
  if (blockIdx.x == 0 && threadIdx.x == 0) { *tail_reset = 0; *tail2_reset = 0; }
  const I2 inst = L::slot() + (int)(blockIdx.x * L::per_wave);
  const I2 k = L::stage();          // (one chain, stage 0 on lane 0: no lane offset)
  MPMPC_TICK_BEGIN(8);
  D2 fields[MPMPC_NUM_FIELDS];
  assemble_fields<L>(cfg, ain.tab, B, inst, k, ain.wp_id, ain.x0, ain.cc, ain.lb, ain.ub, fields);
  ReducedSolver<L> s;
  s.template run<false>(fields, B, inst, k, cfg.N, st);
  MPMPC_TICK_BEGIN(7);
  const I2 inst_o = L::slot_again() + (int)(blockIdx.x * L::per_wave);
  const I2 k_o = L::stage_again();
  s.store(inst_o, k_o, cfg.wheelbase, z, u0, status, iters, resid, y, nullptr, ld);
  MPMPC_TICK_END(7);
  MPMPC_TICK_END(8);
  if (k_o.v[0] == 0 && inst_o.v[0] < B && s.status.v[0] == MPMPC_UNSOLVED) {
    tail[1 + atomicAdd(tail, 1)] = inst_o.v[0];
    __hip_atomic_store(tail_flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// K2rb2: horizons 128 .. 255 with two stages per lane - the instance on a workgroup of TWO wavefronts (128 lanes, one chain of
// eight rows; the lanes of the two wavefronts talk through LDS: LaneBlock<128, ., 128>) instead of four with one stage per lane.
// Lean LDS - 37 pair slots (ReducedSolver::kLean) + 4 exchange rows = 80 000 B - so that TWO workgroups share a CU: four
// wavefronts per CU, one per SIMD, and two instances per CU in flight where K2rb<256> (100 KB, 298 registers) holds one.
constexpr int RNB2_SLOTS = 74, RNB2_XR = 4;
using LanePairBlock = LanePair<LaneBlock<128, RNB2_SLOTS, 128, RNB2_XR>>;
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(1, 1))) void mpmpc_reduced_pair_block_kernel(mpmpc_config cfg, SolverParams st, int B, AssembleIn ain,
                                                           double* __restrict__ z, double* __restrict__ u0,
                                                           int* __restrict__ status, int* __restrict__ iters,
                                                           double* __restrict__ resid, double* __restrict__ y,
                                                           int* __restrict__ tail) {
  using L = LanePairBlock;
  const I2 inst((int)blockIdx.x);
  const I2 k = L::stage();
  D2 fields[MPMPC_NUM_FIELDS];
  assemble_fields<L>(cfg, ain.tab, B, inst, k, ain.wp_id, ain.x0, ain.cc, ain.lb, ain.ub, fields);
  ReducedSolver<L> s;
  static_assert(ReducedSolver<L>::kLean, "37 pair slots: the lean cold storage");
  s.template run<false>(fields, B, inst, k, cfg.N, st);
  s.store(inst, k, cfg.wheelbase, z, u0, status, iters, resid, y, nullptr, 0);
  if (k.v[0] == 0 && inst.v[0] < B && s.status.v[0] == MPMPC_UNSOLVED) tail[1 + atomicAdd(tail, 1)] = inst.v[0];
}

// ... its tail (K2p's solver) and its twin for a terminal cost on the time state (K2t's solver) on the same workgroup layout, with
// the FULL cold storage (80 slots + 4 exchange rows = 86 KB: one workgroup per CU - the tail is a tenth of a batch, and the
// terminal-time weights ran the general 3-state solver on FOUR wavefronts before: 0.19 M solves/s at N = 150)
using LanePairBlockFull = LanePair<LaneBlock<128, RN2_SLOTS, 128, RNB2_XR>>;
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(1, 1))) void mpmpc_reduced_tail_pair_block_kernel(mpmpc_config cfg, SolverParams st, int B, AssembleIn ain,
                                                           double* __restrict__ z, double* __restrict__ u0,
                                                           int* __restrict__ status, int* __restrict__ iters,
                                                           double* __restrict__ resid, double* __restrict__ y,
                                                           const int* __restrict__ tail, int* __restrict__ tail2) {
  using L = LanePairBlockFull;
  if ((int)blockIdx.x >= tail[0]) return;
  const int in0 = tail[1 + blockIdx.x];
  const I2 inst(in0);
  const I2 k = L::stage();
  D2 fields[MPMPC_NUM_FIELDS];
  assemble_fields<L>(cfg, ain.tab, B, inst, k, ain.wp_id, ain.x0, ain.cc, ain.lb, ain.ub, fields);
  ReducedTailSolver<L> s;
  const int base = iters ? iters[in0 * 2 + 1] : 0;
  s.run(fields, B, inst, k, cfg.N, st, I2(base));
  s.store(inst, k, cfg.wheelbase, z, u0, status, iters, resid, y);
  if (k.v[0] == 0 && s.status.v[0] == MPMPC_UNSOLVED) tail2[1 + atomicAdd(tail2, 1)] = in0;
}
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(1, 1))) void mpmpc_reduced_t_pair_block_kernel(mpmpc_config cfg, SolverParams st, int B, AssembleIn ain,
                                                           double* __restrict__ z, double* __restrict__ u0,
                                                           int* __restrict__ status, int* __restrict__ iters,
                                                           double* __restrict__ resid, double* __restrict__ y,
                                                           int* __restrict__ tail) {
  using L = LanePairBlockFull;
  const I2 inst((int)blockIdx.x);
  const I2 k = L::stage();
  D2 fields[MPMPC_NUM_FIELDS];
  assemble_fields<L>(cfg, ain.tab, B, inst, k, ain.wp_id, ain.x0, ain.cc, ain.lb, ain.ub, fields);
  ReducedTSolver<L> s;
  s.run(fields, B, inst, k, cfg.N, st, cfg.QN[2]);
  s.store(inst, k, cfg.wheelbase, z, u0, status, iters, resid, y);
  if (k.v[0] == 0 && inst.v[0] < B && s.status.v[0] == MPMPC_UNSOLVED) tail[1 + atomicAdd(tail, 1)] = inst.v[0];
}

// K2p2: the reduced-native TAIL solver (mpmpc_reduced_tail.hpp: phase 1 / Farkas ray / relaxed plan, one more attempt) with two
// stages per lane - the tail of K2r2<64> at horizons 64 .. 127, one instance per wavefront, where the general solver on a workgroup
// used to take all of it (40 % of an obstacle-course step at N = 100); what it leaves is listed in tail2 for that kernel.
template <int GB>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void mpmpc_reduced_tail_pair_kernel(mpmpc_config cfg, SolverParams st, int B, AssembleIn ain,
                                                           double* __restrict__ z, double* __restrict__ u0,
                                                           int* __restrict__ status, int* __restrict__ iters,
                                                           double* __restrict__ resid, double* __restrict__ y,
                                                           const int* __restrict__ tail, int* __restrict__ tail2) {
  using L = LanePair<LaneGpu<GB, GB, RN2_SLOTS>>;
  static_assert(L::per_wave == 1, "one instance per wavefront");
  if ((int)blockIdx.x >= tail[0]) return;
  const int in0 = tail[1 + blockIdx.x];
  const I2 inst(in0, in0);
  const I2 k = L::stage();
  D2 fields[MPMPC_NUM_FIELDS];
  assemble_fields<L>(cfg, ain.tab, B, inst, k, ain.wp_id, ain.x0, ain.cc, ain.lb, ain.ub, fields);
  ReducedTailSolver<L> s;
  const int base = iters ? iters[in0 * 2 + 1] : 0;
  s.run(fields, B, inst, k, cfg.N, st, I2(base));
  const I2 k_o = L::stage_again();
  s.store(inst, k_o, cfg.wheelbase, z, u0, status, iters, resid, y);
  if (k_o.v[0] == 0 && s.status.v[0] == MPMPC_UNSOLVED) tail2[1 + atomicAdd(tail2, 1)] = in0;
}

// K2t2: K2t with two stages per lane, for horizons 64 .. 127 - ONE wavefront per instance where such weights used to run the
// general 3-state solver on a workgroup of two (mpmpc_solve_block_kernel<128, 0>); 502 registers, no scratch.
template <int GB>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void mpmpc_reduced_t_pair_kernel(mpmpc_config cfg, SolverParams st, int B, AssembleIn ain,
                                                           double* __restrict__ z, double* __restrict__ u0,
                                                           int* __restrict__ status, int* __restrict__ iters,
                                                           double* __restrict__ resid, double* __restrict__ y,
                                                           int* __restrict__ tail) {
  using L = LanePair<LaneGpu<GB, GB, RN2_SLOTS>>;
  const I2 inst = L::slot() + (int)(blockIdx.x * L::per_wave);
  const I2 k = L::stage();
  D2 fields[MPMPC_NUM_FIELDS];
  assemble_fields<L>(cfg, ain.tab, B, inst, k, ain.wp_id, ain.x0, ain.cc, ain.lb, ain.ub, fields);
  ReducedTSolver<L> s;
  s.run(fields, B, inst, k, cfg.N, st, cfg.QN[2]);
  const I2 inst_o = L::slot_again() + (int)(blockIdx.x * L::per_wave);
  const I2 k_o = L::stage_again();
  s.store(inst_o, k_o, cfg.wheelbase, z, u0, status, iters, resid, y);
  if (k_o.v[0] == 0 && inst_o.v[0] < B && s.status.v[0] == MPMPC_UNSOLVED) tail[1 + atomicAdd(tail, 1)] = inst_o.v[0];
}

// K2t: the reduced-native kernel of the weightings with a TERMINAL cost on the time state (mpmpc_reduced_t.hpp; BASELINE
// config 3): the (e_y, e_psi, kappa) problem plus the speeds plus one rank-one term, Sherman-Morrison on the 2 x 2-block
// solves.  Same launch contract as K2r (cold starts only: a warm-started closed loop of such weights starts cold here).
template <int G, int C>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void mpmpc_reduced_t_kernel(mpmpc_config cfg, SolverParams st, int B, AssembleIn ain,
                                                           double* __restrict__ z, double* __restrict__ u0,
                                                           int* __restrict__ status, int* __restrict__ iters,
                                                           double* __restrict__ resid, double* __restrict__ y,
                                                           int* __restrict__ tail, int* __restrict__ tail_reset,
                                                           unsigned* __restrict__ tail_flag, unsigned seq) {
  using L = LaneGpu<G, C, RN_SLOTS>;
  if (blockIdx.x == 0 && threadIdx.x == 0) *tail_reset = 0;
  const int inst = blockIdx.x * L::per_wave + L::slot();
  const int k = L::stage() - lane_offset(G, C, cfg.N);
  MPMPC_TICK_BEGIN(8);
  double fields[MPMPC_NUM_FIELDS];
  assemble_fields<L>(cfg, ain.tab, B, inst, k, ain.wp_id, ain.x0, ain.cc, ain.lb, ain.ub, fields);
  ReducedTSolver<L> s;
  s.run(fields, B, inst, k, cfg.N, st, cfg.QN[2]);
  MPMPC_TICK_BEGIN(7);
  const int inst_o = blockIdx.x * L::per_wave + L::slot_again();
  const int k_o = L::stage_again() - lane_offset(G, C, cfg.N);
  s.store(inst_o, k_o, cfg.wheelbase, z, u0, status, iters, resid, y);
  MPMPC_TICK_END(7);
  MPMPC_TICK_END(8);
  if (k_o == 0 && inst_o < B && s.status == MPMPC_UNSOLVED) {
    tail[1 + atomicAdd(tail, 1)] = inst_o;
    __hip_atomic_store(tail_flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// K2p: the reduced-native TAIL kernel (mpmpc_reduced_tail.hpp): phase 1 and one more attempt of the certified polish on
// the instances the reduced-native launch listed in `tail` - two per wave (<32,16>, the default: 243 registers, no scratch),
// one per wave in the split layout (<64,16>) or, for horizons 32 .. 63, one per wave with a lane per stage (<64,32>: 235
// registers, no scratch) - within the same 256 registers and 40 LDS slots: a tail wave shares its SIMD.  What it leaves UNSOLVED is appended to tail2 for the general kernel (mode 2).
template <int G, int C>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void mpmpc_reduced_tail_kernel(mpmpc_config cfg, SolverParams st, int B, AssembleIn ain,
                                                           double* __restrict__ z, double* __restrict__ u0,
                                                           int* __restrict__ status, int* __restrict__ iters,
                                                           double* __restrict__ resid, double* __restrict__ y,
                                                           const int* __restrict__ tail, int* __restrict__ tail_reset,
                                                           int* __restrict__ tail2, unsigned* __restrict__ tail2_flag, unsigned seq) {
  using L = LaneGpu<G, C, RN_SLOTS>;
  if (blockIdx.x == 0 && threadIdx.x == 0) *tail_reset = 0;
  if ((int)blockIdx.x * L::per_wave >= tail[0]) return;
  // (the instances of a wave: consecutive entries of the list; the last wave of a packed launch may carry B = "none")
  const int e_ = blockIdx.x * L::per_wave + L::slot();
  const int inst = e_ < tail[0] ? tail[1 + e_] : B;
  const int k = L::stage() - lane_offset(G, C, cfg.N);
  MPMPC_TICK_BEGIN(8);
  double fields[MPMPC_NUM_FIELDS];
  assemble_fields<L>(cfg, ain.tab, B, inst, k, ain.wp_id, ain.x0, ain.cc, ain.lb, ain.ub, fields);
  ReducedTailSolver<L> s;
  s.run(fields, B, inst, k, cfg.N, st, (iters && inst < B) ? iters[inst * 2 + 1] : 0);
  MPMPC_TICK_BEGIN(7);
  const int e_o = blockIdx.x * L::per_wave + L::slot_again();
  const int inst_o = e_o < tail[0] ? tail[1 + e_o] : B;
  const int k_o = L::stage_again() - lane_offset(G, C, cfg.N);
  s.store(inst_o, k_o, cfg.wheelbase, z, u0, status, iters, resid, y);
  MPMPC_TICK_END(7);
  MPMPC_TICK_END(8);
  if (k_o == 0 && inst_o < B && s.status == MPMPC_UNSOLVED) {
    tail2[1 + atomicAdd(tail2, 1)] = inst_o;
    __hip_atomic_store(tail2_flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// K4: speed profile.  The kernel of choice is mpmpc_speed_profile_wave_kernel below (one wavefront per path); these
// two run the same code one thread per path (a serial interior-point / active-set run over a scalar tridiagonal
// system, speed_core.hpp) for paths too long for the LDS: workspace path-minor in HBM, so that the threads of a
// wave touch consecutive addresses, or in LDS for a few paths.
__global__ __launch_bounds__(64) void mpmpc_speed_profile_kernel(int B, int n, const double* __restrict__ li,
                                                                 const double* __restrict__ kappa,
                                                                 const double* __restrict__ limits, double eps,
                                                                 double* __restrict__ work, double* __restrict__ v,
                                                                 int* __restrict__ status, int* __restrict__ iters) {
  const int p = blockIdx.x * 64 + threadIdx.x;
  if (p >= B) return;
  SpWork W{work + p, n, B};
  SpLimits lim{limits[5 * p], limits[5 * p + 1], limits[5 * p + 2], limits[5 * p + 3], limits[5 * p + 4]};
  int it = 0;
  status[p] = sp_solve(n, li + (long)p * n, kappa + (long)p * n, 1, lim, eps, W, v + (long)p * n, 1, &it);
  iters[p] = it;
}

// Few paths (the usual case is one): one path per block with the workspace in LDS, so that the serial
// run is bound by LDS latency instead of HBM latency.
__global__ __launch_bounds__(64) void mpmpc_speed_profile_lds_kernel(int B, int n, const double* __restrict__ li,
                                                                     const double* __restrict__ kappa,
                                                                     const double* __restrict__ limits, double eps,
                                                                     double* __restrict__ v, int* __restrict__ status,
                                                                     int* __restrict__ iters) {
  extern __shared__ double sp_lds[];
  const int p = blockIdx.x;
  if (threadIdx.x != 0 || p >= B) return;
  SpWork W{sp_lds, n, 1};
  SpLimits lim{limits[5 * p], limits[5 * p + 1], limits[5 * p + 2], limits[5 * p + 3], limits[5 * p + 4]};
  int it = 0;
  status[p] = sp_solve(n, li + (long)p * n, kappa + (long)p * n, 1, lim, eps, W, v + (long)p * n, 1, &it);
  iters[p] = it;
}

// One WAVEFRONT per path: the elementwise loops of sp_solve_t run strided over the 64 lanes, reductions are
// shuffles, and the tridiagonal systems are solved by parallel cyclic reduction in LDS - "factor" keeps the two
// multipliers of every row and level (ceil(log2 n) levels) and the final reciprocal diagonal, "solve" applies them to
// a right-hand side.  Everything (25 work arrays + 25 n doubles of cyclic-reduction state) lives in LDS.
struct SpWave {
  double* x;          // cyclic-reduction state behind the SP_ARRAYS work arrays
  int cap;            // row capacity (n of the launch)
  int levels = 0;
  __device__ int first() const { return threadIdx.x; }
  __device__ int step() const { return 64; }
  __device__ void sync() const { __syncthreads(); }
  __device__ double rmax(double v) const {
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
  }
  __device__ double rsum(double v) const {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
  }
  __device__ bool any(bool b) const { __syncthreads(); return __ballot(b) != 0ull; }
  // layout of x: A0 B0 C0 A1 B1 C1 (double-buffered rows), D0 D1, BINV, then K1[l], K2[l] per level
  __device__ double* arr(int a) const { return x + (long)a * cap; }
  __device__ void tri_factor(const SpWork& W, int n) const {
    const int lane = threadIdx.x;
    double *a0 = arr(0), *b0 = arr(1), *c0 = arr(2), *a1 = arr(3), *b1 = arr(4), *c1 = arr(5);
    for (int i = lane; i < n; i += 64) {
      a0[i] = i > 0 ? W(SP_ME, i - 1) : 0.0;
      b0[i] = W(SP_MD, i);
      c0[i] = i + 1 < n ? W(SP_ME, i) : 0.0;
    }
    __syncthreads();
    int l = 0;
    for (int s = 1; s < n; s <<= 1, ++l) {
      double *k1 = arr(9 + 2 * l), *k2 = arr(10 + 2 * l);
      for (int i = lane; i < n; i += 64) {
        const bool lo = i - s >= 0, hi = i + s < n;
        const double m1 = lo ? a0[i] / b0[i - s] : 0.0, m2 = hi ? c0[i] / b0[i + s] : 0.0;
        k1[i] = m1; k2[i] = m2;
        a1[i] = lo ? -a0[i - s] * m1 : 0.0;
        c1[i] = hi ? -c0[i + s] * m2 : 0.0;
        b1[i] = b0[i] - (lo ? c0[i - s] * m1 : 0.0) - (hi ? a0[i + s] * m2 : 0.0);
      }
      __syncthreads();
      double* t;
      t = a0; a0 = a1; a1 = t; t = b0; b0 = b1; b1 = t; t = c0; c0 = c1; c1 = t;
    }
    double* binv = arr(8);
    for (int i = lane; i < n; i += 64) binv[i] = 1.0 / b0[i];
    __syncthreads();
  }
  __device__ void tri_solve(const SpWork& W, int n) const {
    const int lane = threadIdx.x;
    double *d0 = arr(6), *d1 = arr(7);
    for (int i = lane; i < n; i += 64) d0[i] = W(SP_RHS, i);
    __syncthreads();
    int l = 0;
    for (int s = 1; s < n; s <<= 1, ++l) {
      const double *k1 = arr(9 + 2 * l), *k2 = arr(10 + 2 * l);
      for (int i = lane; i < n; i += 64)
        d1[i] = d0[i] - (i - s >= 0 ? d0[i - s] * k1[i] : 0.0) - (i + s < n ? d0[i + s] * k2[i] : 0.0);
      __syncthreads();
      double* t = d0; d0 = d1; d1 = t;
    }
    const double* binv = arr(8);
    for (int i = lane; i < n; i += 64) W(SP_DX, i) = d0[i] * binv[i];
    __syncthreads();
  }
};
__host__ __device__ inline int sp_wave_arrays(int n) {      // doubles of SpWave state per row
  int levels = 0;
  for (int s = 1; s < n; s <<= 1) ++levels;
  return 9 + 2 * levels;
}
__global__ __launch_bounds__(64) void mpmpc_speed_profile_wave_kernel(int B, int n, const double* __restrict__ li,
                                                                      const double* __restrict__ kappa,
                                                                      const double* __restrict__ limits, double eps,
                                                                      double* __restrict__ v, int* __restrict__ status,
                                                                      int* __restrict__ iters) {
  extern __shared__ double sp_lds[];
  const int p = blockIdx.x;
  if (p >= B) return;
  SpWork W{sp_lds, n, 1};
  SpWave pol{sp_lds + (long)SP_ARRAYS * n, n};
  SpLimits lim{limits[5 * p], limits[5 * p + 1], limits[5 * p + 2], limits[5 * p + 3], limits[5 * p + 4]};
  int it = 0;
  const int st = sp_solve_t(pol, n, li + (long)p * n, kappa + (long)p * n, 1, lim, eps, W, v + (long)p * n, 1, &it);
  if (threadIdx.x == 0) { status[p] = st; iters[p] = it; }
}

// K0a: free segments of every waypoint's border line, one WAVEFRONT per waypoint.  Lane 0 walks Zingl's anti-aliased
// line (a float32 recurrence in skimage's exact cell order: inherently serial, ~95 cells on Sim_Track) and leaves the
// cells in LDS; all 64 lanes then fetch the occupancies (one memory round trip for the whole line instead of one per
// cell; cells outside the grid count as occupied); lane 0 runs the run-length state machine over the LDS copy and, for
// a waypoint with at most one free segment, computes its bounds once (cor_forced) instead of once per start waypoint
// and column in K0b.  A line longer than COR_CELL_CAP cells or with more than COR_MAXSEG free segments raises *err
// (1 / 2) - mpmpc_build_corridor then fails instead of working with a truncated list.
__global__ __launch_bounds__(64) void mpmpc_free_segments_kernel(MapView map, PathGeom g, const double* __restrict__ bub,
                                                                 const double* __restrict__ blb, double min_width,
                                                                 double safety_margin, double* __restrict__ segs,
                                                                 int* __restrict__ nseg, double* __restrict__ wpc,
                                                                 int* __restrict__ err) {
  __shared__ int cells[COR_CELL_CAP];
  __shared__ unsigned char occ[COR_CELL_CAP];
  __shared__ int s_n;
  __shared__ double seg[4 * COR_MAXSEG];      // lane 0's segment list (in LDS: a per-lane array would live in scratch)
  const int i = blockIdx.x, lane = threadIdx.x;
  int ux, uy, lx, ly;
  cor_w2m(map, bub[2 * i], bub[2 * i + 1], ux, uy);
  cor_w2m(map, blb[2 * i], blb[2 * i + 1], lx, ly);
  if (lane == 0) s_n = cor_line_cells(ux, uy, lx, ly, cells, COR_CELL_CAP);
  __syncthreads();
  const int n = s_n;
  int cnt;
  if (n > COR_CELL_CAP) {
    cnt = COR_E_CELLS;
  } else {
    for (int k = lane; k < n; k += 64) {
      int x, y;
      cor_unpack_cell(cells[k], x, y);
      occ[k] = cor_cell_free(map, x, y) ? 1 : 0;
    }
    __syncthreads();
    if (lane != 0) return;
    cnt = cor_scan_cells(map, ux, uy, lx, ly, min_width, n, [&](int c, int& x, int& y) { cor_unpack_cell(cells[c], x, y); },
                         [&](int c) { return occ[c] != 0; }, seg);
    for (int k = 0; k < 4 * COR_MAXSEG; ++k) segs[(long)i * 4 * COR_MAXSEG + k] = (cnt > 0 && k < 4 * cnt) ? seg[k] : 0.0;
  }
  if (lane != 0) return;
  if (cnt < 0) { atomicMax(err, cnt == COR_E_CELLS ? 1 : 2); cnt = 0; }
  nseg[i] = cnt;
  if (cnt <= 1) cor_forced(g, segs, nseg, i, safety_margin, wpc + (long)i * COR_WPC);
}

// K0b: horizon walk for every start waypoint w (table row w = update_path_constraints(w + 1, ...)).
// K0b: one thread per (start waypoint, column): cor_select_one replays only the short run of multi-segment waypoints
// right before its column, everything else is a row of K0a's per-waypoint table.
__global__ __launch_bounds__(256) void mpmpc_corridor_select_kernel(PathGeom g, const double* __restrict__ segs,
                                                                    const int* __restrict__ nseg, int n_cols,
                                                                    double safety_margin, double* __restrict__ ub_tab,
                                                                    double* __restrict__ lb_tab, int* __restrict__ bad,
                                                                    const double* __restrict__ wpc) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int w = t / n_cols, n = t - w * n_cols;
  if (w >= g.n_wp) return;
  double ub, lb;
  if (!cor_select_one(g, segs, nseg, w + 1, n, safety_margin, wpc, &ub, &lb)) {
    ub = lb = __builtin_nan("");
    if (n == 0) atomicAdd(bad, 1);
  }
  ub_tab[(long)w * n_cols + n] = ub;
  lb_tab[(long)w * n_cols + n] = lb;
}

// K3a: where is each car on the path, and what is its path-relative state (one thread per car)
__global__ __launch_bounds__(256) void mpmpc_localise_kernel(int B, int n_wp, const double* __restrict__ cum,
                                                             const double* __restrict__ gx, const double* __restrict__ gy,
                                                             const double* __restrict__ gpsi, const double* __restrict__ s,
                                                             const double* __restrict__ pose, int* __restrict__ alive,
                                                             int* __restrict__ wp_id, double* __restrict__ x0,
                                                             int* __restrict__ shift) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B || alive[i] != 1) return;
  const int wp = ro_current_waypoint(cum, n_wp, s[i]);
  if (wp < 0) { alive[i] = 0; return; }              // lap finished
  if (shift) {                                        // waypoints advanced since the last step (warm start)
    int d = wp - wp_id[i];
    if (d < 0) d += n_wp;
    shift[i] = d;
  }
  wp_id[i] = wp;
  ro_t2s(pose[3 * i], pose[3 * i + 1], pose[3 * i + 2], gx[wp], gy[wp], gpsi[wp], x0 + 3 * i);
}

// K3b: use the solution (or the fallback plan), drive the plant one step (one thread per car)
__global__ __launch_bounds__(256) void mpmpc_advance_kernel(int B, int N, double L, double Ts, const double* __restrict__ kappa,
                                                            const int* __restrict__ wp_id, const double* __restrict__ x0,
                                                            const int* __restrict__ status, const double* __restrict__ z,
                                                            double* __restrict__ cc, int* __restrict__ counter,
                                                            int* __restrict__ alive, double* __restrict__ pose,
                                                            double* __restrict__ s, double* __restrict__ u_last) {
  // one thread per (car, plan entry): the N arctangents of a car's new plan are independent; the thread of entry 0
  // then drives the car (it reads only plan entries it wrote itself, or - fallback - entries nobody writes)
  const int g = blockIdx.x * 256 + threadIdx.x;
  const int i = g / N, k = g - i * N;
  if (i >= B || alive[i] != 1) return;
  const int n = 5 * N + 3;
  const int st = status[i];
  if (ro_usable(st)) ro_plan_entry(N, L, z + (long)i * n, cc + (long)i * 2 * N, k);
  if (k == 0 && !ro_drive(N, L, Ts, st, cc + (long)i * 2 * N, counter + i, x0 + 3 * i, kappa[wp_id[i]], pose + 3 * i,
                          s + i, u_last + 2 * i))
    alive[i] = -1;
}

// ------------------------------------------------------------------------------------ host side
static thread_local std::string g_err;
static int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess)                                                                          \
      return fail(MPMPC_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));                 \
  } while (0)

struct mpmpc_handle_s {
  mpmpc_config cfg;
  mpmpc_settings st;
  int ld = 0, n = 0, m = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
  hipEvent_t ev_in = nullptr;      // the last staged upload has left the pinned buffer
  bool in_flight = false;
  // tables
  double *kappa = nullptr, *v_ref = nullptr, *ds_next = nullptr, *ub_tab = nullptr, *lb_tab = nullptr;
  int n_wp = 0, n_cols = 0;
  // corridor generation on the device: grid, per-waypoint geometry, per-waypoint free segments
  int8_t* map = nullptr;
  int map_h = 0, map_w = 0;
  double map_ox = 0, map_oy = 0, map_res = 0;
  double *gx = nullptr, *gy = nullptr, *gpsi = nullptr, *bub = nullptr, *blb = nullptr, *segs = nullptr, *gtrig = nullptr;
  int *nseg = nullptr, *bad = nullptr;      // bad[0]: start waypoints without a free segment, bad[1]: K0a overflow code
  int geom_n = 0;
  std::vector<double> host_bub, host_blb;   // border points, kept to validate them against the map of the moment
  // closed-loop rollout state
  double *ro_cum = nullptr, *ro_s = nullptr, *ro_pose = nullptr, *ro_u = nullptr;
  int *ro_counter = nullptr, *ro_alive = nullptr;
  int *ro_act = nullptr, *ro_shift = nullptr;      // warm start: certified active sets [B x ld], waypoints advanced [B]
  int ro_warm = 2;      // 0 off, 1 on, 2 where it pays (see launch_solve)
  double ro_Ts = 0;
  int ro_B = 0;
  bool ro_valid = false;      // the batch blocks still hold the rollout's plans / waypoint ids / states
  // per-batch inputs
  int* wp_id = nullptr;
  double *x0 = nullptr, *cc = nullptr, *lb = nullptr, *ub = nullptr;
  bool have_rows = false;     // per-instance corridor rows uploaded (else: table)
  int uploaded = 0;
  // the inputs of a batch lie back to back in ONE device block (wp_id, x0, cc, lb, ub, laid out for the batch size
  // of the last upload), and so do the outputs (u0, resid, status, iters, z, y): an upload or a download through
  // the pinned staging buffers is then a single copy instead of five or six - what a single-car call is made of
  char *in_block = nullptr, *out_block = nullptr;
  int laid_out = 0;
  // stage-blocked QP and outputs
  double *qp = nullptr, *z = nullptr, *u0 = nullptr, *resid = nullptr, *y = nullptr;
  int *status = nullptr, *iters = nullptr;
  // pinned host staging for mpmpc_upload / mpmpc_download (small copies from pageable memory are
  // synchronous and slow; nullptr above STAGE_LIMIT bytes: large batches amortise the direct path)
  char *stage_in = nullptr, *stage_out = nullptr;
  size_t stage_in_bytes = 0, stage_out_bytes = 0;
  // ... above that limit an upload goes through a small page-locked bounce buffer in two halves (allocated by the first such
  // upload): the device copy of one half runs while the host fills the other, and the call returns when the caller's buffers
  // have been read - not when the device has them.  (A copy from pageable memory is synchronous: one process feeding several
  // devices - sharded.py - would otherwise serve them one after the other)
  char* bounce = nullptr;
  hipEvent_t ev_bounce[2] = {nullptr, nullptr};
  bool bounce_busy[2] = {false, false};
  int bounce_turn = 0;
  // instances the early pass of a packed (2 or 4 per wave) launch could not certify: [0] = count, [1..] = ids
  int* tail = nullptr;
  int tail_flip = 0;          // which of the two lists the next reduced-native launch fills
  // Deferred tail (batch launches of the reduced-native kernels, not the closed loop).  Launching the tail kernel costs
  // 4.8 us even when its list is empty - 10 % of a 1 024-instance step - so when the last launch whose outcome the host
  // has SEEN left no tail, the next launches do not enqueue one: the reduced-native kernel stamps its sequence number into
  // tail_flag (host memory) if it appends anything, and every entry point through which results or state can be observed
  // or changed (sync, download, upload, set_*, rollout, timed launches) first drains the stream, looks at the stamp and
  // runs the tail of the LAST launch if it left one (observe_tail).  A launch that is followed by another launch without
  // such a call in between has unobservable results either way (same output buffers).
  unsigned* tail_flag = nullptr;      // hipHostMalloc'ed, device-visible
  unsigned seq = 0;                   // sequence number of the last reduced-native launch
  bool tail_expect_empty = false;     // the last observed launch left no tail
  bool pend = false;                  // the last launch's tail launch was not enqueued
  bool tail_ran_late = false;         // the last observe_tail had to run a deferred tail
  // Second level (reduced_native_tail configurations): the tail goes to the reduced-native tail kernel first, which appends
  // what IT leaves to a third list (tail + 2 (max_batch + 1)) and stamps tail_flag[1]; the general kernel on that list is
  // deferred by the same rule.
  bool tail2_expect_empty = false, pend2 = false;
  size_t staged_bytes = 0;            // mpmpc_staged_begin without its mpmpc_staged_end: bytes of the output block on their way back
  int pend_B = 0;
  bool pend_y = false;
  int *pend_cur = nullptr, *pend_next = nullptr;
  // ---- further launch slots (pipelined resident launches, mpmpc_solve_resident): each its own stream, output block, tail
  // lists and deferred-tail state.  The members above are those of the slot of the LAST launch; `alt` holds the other slots',
  // and a resident launch swaps the next one in before it goes out, so that launch k + 1 .. k + pipeline - 1 run beside launch k
  // (each on its own stream, writing its own outputs) and everything else in this file keeps working on "the last launch"
  // unchanged.
  struct Slot {
    hipStream_t stream = nullptr;
    char* out_block = nullptr;
    double *z = nullptr, *u0 = nullptr, *resid = nullptr, *y = nullptr;
    int *status = nullptr, *iters = nullptr;
    int* tail = nullptr;
    int tail_flip = 0;
    unsigned* tail_flag = nullptr;
    unsigned seq = 0;
    bool tail_expect_empty = false, pend = false, tail_ran_late = false, tail2_expect_empty = false, pend2 = false;
    int pend_B = 0;
    bool pend_y = false;
    int *pend_cur = nullptr, *pend_next = nullptr;
    bool y_valid = false;
    bool busy = false;      // launches in flight on this slot's stream that nothing has waited for yet
  };
  static constexpr int MAX_PIPELINE = 8;
  Slot alt[MAX_PIPELINE - 1];      // the other slots; alt[i] exists (is allocated) for i < n_alt
  int n_alt = 0;
  int ring = 0;             // alt slot the next resident launch swaps with: 0 .. pipeline - 2 in turn, which takes the launches
                            // through all `pipeline` slots round robin
  bool busy = false;        // ... the same for the slot of the last launch
  int pipeline = 3;         // mpmpc_set_pipeline: resident launches in flight (1 = one slot only).  Three by default: the HIP
                            // runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4, one of them taken), and two
                            // streams on one queue serialise - measured, config 2: depth 2 / 3 / 4 = 38 / 49 / 36 M solves/s with
                            // the default, 38 / 49 / 61.5 M with GPU_MAX_HW_QUEUES=8 (profiles/r4/depth_sweep.txt)
  hipEvent_t ev_order = nullptr;
  bool order_pending = false;      // asynchronous work other than resident solves is queued on `stream`: the next resident launch
                                   // on the OTHER stream has to wait for it (uploads, closed-loop steps: they write what solves read)
  int force_lanes = 0;      // mpmpc_set_packing: 0 = chosen from the batch size
  // mpmpc_set_tail_kernel (MPMPC_LEAN_TAIL=0 in the environment: off from the start, for A/B timings of whole programs)
  bool lean_tail = !(std::getenv("MPMPC_LEAN_TAIL") && std::atoi(std::getenv("MPMPC_LEAN_TAIL")) == 0);
  bool lean_tail_single = std::getenv("MPMPC_LEAN_TAIL") && std::atoi(std::getenv("MPMPC_LEAN_TAIL")) == 2;      // ONE tail instance per wave
  bool staged_async = true; // mpmpc_staged_begin was called as such (not as the first half of mpmpc_solve_staged)
  bool resident_y = true;   // mpmpc_set_outputs: do resident launches store the multipliers y (46 % of the output bytes)?
  bool y_valid = false;     // the last solve launch stored y
};

static int host_stage_ld(int N) { return N + 1 <= 16 ? 16 : (N + 1 <= 32 ? 32 : (N + 1 <= 64 ? 64 : (N + 1 <= 128 ? 128 : 256))); }

static int observe_tail(mpmpc_handle h);
static void swap_slots(mpmpc_handle h, int i) {
  auto& a = h->alt[i];
  std::swap(h->stream, a.stream); std::swap(h->out_block, a.out_block);
  std::swap(h->z, a.z); std::swap(h->u0, a.u0); std::swap(h->resid, a.resid); std::swap(h->y, a.y);
  std::swap(h->status, a.status); std::swap(h->iters, a.iters);
  std::swap(h->tail, a.tail); std::swap(h->tail_flip, a.tail_flip); std::swap(h->tail_flag, a.tail_flag); std::swap(h->seq, a.seq);
  std::swap(h->tail_expect_empty, a.tail_expect_empty); std::swap(h->pend, a.pend); std::swap(h->tail_ran_late, a.tail_ran_late);
  std::swap(h->tail2_expect_empty, a.tail2_expect_empty); std::swap(h->pend2, a.pend2);
  std::swap(h->pend_B, a.pend_B); std::swap(h->pend_y, a.pend_y); std::swap(h->pend_cur, a.pend_cur); std::swap(h->pend_next, a.pend_next);
  std::swap(h->y_valid, a.y_valid); std::swap(h->busy, a.busy);
}
// The other slots' launches are drained and their deferred tails, if any, are run (the slot of the last launch is left as it
// is: what follows on its stream is ordered behind it anyway).
static int settle_other_slot(mpmpc_handle h) {
  for (int i = 0; i < h->n_alt; ++i) {
    if (!h->alt[i].busy && !h->alt[i].pend && !h->alt[i].pend2) continue;
    swap_slots(h, i);
    const int rc = observe_tail(h);
    swap_slots(h, i);
    if (rc) return rc;
  }
  return MPMPC_OK;
}
// first statement of every entry point that reads results or changes what a launch in flight - or a deferred tail launch -
// works on
#define MPMPC_SETTLE(h)                                      \
  do {                                                       \
    if ((h)->staged_bytes) {                                 \
      if (int rc_ = mpmpc_staged_end(h)) return rc_;         \
    }                                                        \
    if (int rc_ = settle_other_slot(h)) return rc_;          \
    if ((h)->pend || (h)->pend2) {                           \
      if (int rc_ = observe_tail(h)) return rc_;             \
    }                                                        \
    (h)->order_pending = true;                               \
  } while (0)

static int check_settings(const mpmpc_settings* s) {
  if (!s) return fail(MPMPC_E_ARG, "settings is NULL");
  if (!(s->rho > 0) || !(s->sigma > 0) || !(s->alpha > 0 && s->alpha < 2))
    return fail(MPMPC_E_ARG, "need rho > 0, sigma > 0, 0 < alpha < 2");
  if (s->early_polish < 0) return fail(MPMPC_E_ARG, "early_polish must be >= 0");
  if (s->early_scaling < 0) return fail(MPMPC_E_ARG, "early_scaling must be >= 0");
  if (s->max_iter < 0 || s->check_termination < 0 || s->scaling < 0 || s->ipm_max_iter < 0 || s->as_rounds < 0 ||
      s->as_refine < 0)
    return fail(MPMPC_E_ARG, "iteration counts must be non-negative");
  if (s->polish != 0 && s->polish != 2) return fail(MPMPC_E_ARG, "polish must be 0 or 2");
  if (s->polish == 2 && (!(s->ipm_reg > 0) || !(s->as_delta > 0) || !(s->ipm_tol > 0)))
    return fail(MPMPC_E_ARG, "polish needs ipm_reg, ipm_tol, as_delta > 0");
  if (s->phase1 != 0 && s->phase1 != 1) return fail(MPMPC_E_ARG, "phase1 must be 0 or 1");
  if (s->reduce != 0 && s->reduce != 1) return fail(MPMPC_E_ARG, "reduce must be 0 or 1");
  if (s->native != 0 && s->native != 1) return fail(MPMPC_E_ARG, "native must be 0 or 1");
  if (!(s->native_ipm_tol > 0)) return fail(MPMPC_E_ARG, "need native_ipm_tol > 0");
  if (s->early_start != 0 && s->early_start != 1) return fail(MPMPC_E_ARG, "early_start must be 0 or 1");
  if (s->phase1_accept != 0 && s->phase1_accept != 1) return fail(MPMPC_E_ARG, "phase1_accept must be 0 or 1");
  if (!(s->phase1_band >= 0)) return fail(MPMPC_E_ARG, "phase1_band must be >= 0");
  if (!(s->ipm_start_dual >= 0)) return fail(MPMPC_E_ARG, "need ipm_start_dual >= 0");
  if (!(s->ipm_start_mu >= 0) || !(s->ipm_start_slack > 0)) return fail(MPMPC_E_ARG, "need ipm_start_mu >= 0, ipm_start_slack > 0");
  if (!(s->as_add_fraction >= 0) || !(s->as_add_fraction <= 1)) return fail(MPMPC_E_ARG, "need 0 <= as_add_fraction <= 1");
  if (!(s->ipm_diverged > 1) || !(s->phase1_theta > 0) || !(s->phase1_eps > 0))
    return fail(MPMPC_E_ARG, "need ipm_diverged > 1, phase1_theta > 0, phase1_eps > 0");
  return MPMPC_OK;
}

static int grow_slots(mpmpc_handle h, int want);
static int upload_through_bounce(mpmpc_handle h, void* dst, const void* src, size_t bytes);
static int launch_assemble(mpmpc_handle h, int B);
// how a solve launch was asked for (decides the packing of the reduced-native kernels, see launch_solve)
enum LaunchKind { LAUNCH_SINGLE = 0, LAUNCH_PIPELINED = 1 };
static int launch_solve(mpmpc_handle h, int B, bool closed_loop = false, bool want_y = true, int tail_only = 0, LaunchKind kind = LAUNCH_SINGLE);

extern "C" {

#ifndef MPMPC_SRC_HASH
#define MPMPC_SRC_HASH "unhashed"
#endif
const char* mpmpc_version(void) { return "mpmpc 0.2.0 (gfx950, float64, src " MPMPC_SRC_HASH ")"; }
const char* mpmpc_last_error(void) { return g_err.c_str(); }

int mpmpc_device_count(int32_t* count) {
  if (!count) return fail(MPMPC_E_ARG, "count is NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *count = 0;
    return fail(MPMPC_E_HIP, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
  }
  *count = n;
  return MPMPC_OK;
}

void mpmpc_default_settings(mpmpc_settings* s) {
  if (!s) return;
  s->rho = 0.1; s->sigma = 1e-6; s->alpha = 1.6;
  s->eps_abs = 1e-3; s->eps_rel = 1e-3; s->eps_prim_inf = 1e-4; s->eps_dual_inf = 1e-4;
  s->max_iter = 4000; s->check_termination = 25; s->scaling = 10;
  s->adaptive_rho = 1; s->adaptive_rho_interval = 50; s->adaptive_rho_tolerance = 5.0;
  s->polish = 2; s->ipm_max_iter = 30; s->ipm_tol = 1e-8; s->ipm_reg = 1e-8;
  s->as_delta = 1e-10; s->as_refine = 5; s->as_rounds = 4; s->cert_tol = 1e-8;
  s->early_polish = 1;
  s->early_scaling = 1;
  s->phase1 = 1;
  s->ipm_diverged = 1e2;
  s->phase1_theta = 1.0;
  s->phase1_eps = 1e-6;
  s->reduce = 1;
  s->ipm_start_slack = 0.1;
  s->ipm_start_mu = 0.01;
  s->ipm_start_dual = 0.2;
  s->as_add_fraction = 0.25;
  s->phase1_accept = 1;
  s->native = 1;
  s->native_ipm_tol = 1e-7;
  s->early_start = 0;
  s->phase1_band = 3.0;
}

int32_t mpmpc_stage_ld(int32_t N) { return host_stage_ld(N); }

static size_t pad8(size_t b) { return (b + 7) & ~size_t(7); }
struct BlockLayout {
  size_t wp_id, x0, cc, lb, ub, in_end_cc, in_end;          // byte offsets in in_block
  size_t u0, resid, status, iters, z, y, out_end_z, out_end;  // byte offsets in out_block
};
static BlockLayout block_layout(int N, int B) {
  BlockLayout L;
  const size_t d = sizeof(double), n = 5 * (size_t)N + 3, m = 8 * (size_t)N + 6, b = (size_t)B;
  L.wp_id = 0;
  L.x0 = pad8(sizeof(int) * b);
  L.cc = L.x0 + d * 3 * b;
  L.lb = L.in_end_cc = L.cc + d * 2 * N * b;
  L.ub = L.lb + d * N * b;
  L.in_end = L.ub + d * N * b;
  L.u0 = 0;
  L.resid = L.u0 + d * 2 * b;
  L.status = L.resid + d * 2 * b;
  L.iters = L.status + pad8(sizeof(int) * b);
  L.z = L.iters + pad8(sizeof(int) * 2 * b);
  L.y = L.out_end_z = L.z + d * n * b;
  L.out_end = L.y + d * m * b;
  return L;
}
static void lay_out(mpmpc_handle h, int B) {
  const BlockLayout L = block_layout(h->cfg.N, B);
  h->wp_id = (int*)(h->in_block + L.wp_id);
  h->x0 = (double*)(h->in_block + L.x0);
  h->cc = (double*)(h->in_block + L.cc);
  h->lb = (double*)(h->in_block + L.lb);
  h->ub = (double*)(h->in_block + L.ub);
  h->u0 = (double*)(h->out_block + L.u0);
  h->resid = (double*)(h->out_block + L.resid);
  h->status = (int*)(h->out_block + L.status);
  h->iters = (int*)(h->out_block + L.iters);
  h->z = (double*)(h->out_block + L.z);
  h->y = (double*)(h->out_block + L.y);
  for (int i = 0; i < h->n_alt; ++i) {
    auto& a = h->alt[i];
    a.u0 = (double*)(a.out_block + L.u0);
    a.resid = (double*)(a.out_block + L.resid);
    a.status = (int*)(a.out_block + L.status);
    a.iters = (int*)(a.out_block + L.iters);
    a.z = (double*)(a.out_block + L.z);
    a.y = (double*)(a.out_block + L.y);
  }
  h->laid_out = B;
}

// launch slots beyond the first: stream, output block (laid out like the first), tail lists, tail flag
static int grow_slots(mpmpc_handle h, int want) {
  const BlockLayout lay = block_layout(h->cfg.N, h->cfg.max_batch);
  while (h->n_alt < want) {
    auto& a = h->alt[h->n_alt];
    const size_t nt = 3 * ((size_t)h->cfg.max_batch + 1);
    if (hipMalloc((void**)&a.out_block, lay.out_end) != hipSuccess || hipMalloc((void**)&a.tail, nt * sizeof(int)) != hipSuccess ||
        hipMemset(a.tail, 0, nt * sizeof(int)) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void**>(&a.tail_flag), 2 * sizeof(unsigned), hipHostMallocDefault) != hipSuccess ||
        hipStreamCreate(&a.stream) != hipSuccess)
      return fail(MPMPC_E_HIP, "allocation of a launch slot failed (stream / output block / tail list)");
    a.tail_flag[0] = a.tail_flag[1] = 0u;
    h->n_alt += 1;
  }
  lay_out(h, h->laid_out ? h->laid_out : h->cfg.max_batch);
  return MPMPC_OK;
}

int mpmpc_destroy(mpmpc_handle h) {
  if (!h) return MPMPC_OK;
  (void)hipSetDevice(h->cfg.device);
  void* ptrs[] = {h->kappa, h->v_ref, h->ds_next, h->ub_tab, h->lb_tab, h->in_block, h->out_block,
                  h->qp,    h->map, h->gx,  h->gy,
                  h->gpsi,  h->bub,   h->blb,     h->segs,   h->nseg,   h->bad,    h->ro_cum, h->ro_s, h->ro_pose, h->gtrig,
                  h->ro_u,  h->ro_counter, h->ro_alive, h->tail, h->ro_act, h->ro_shift};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  for (auto& a : h->alt) {
    if (a.out_block) (void)hipFree(a.out_block);
    if (a.tail) (void)hipFree(a.tail);
    if (a.tail_flag) (void)hipHostFree(a.tail_flag);
    if (a.stream) (void)hipStreamDestroy(a.stream);
  }
  if (h->ev_order) (void)hipEventDestroy(h->ev_order);
  if (h->bounce) (void)hipHostFree(h->bounce);
  for (auto& e : h->ev_bounce)
    if (e) (void)hipEventDestroy(e);
  if (h->stage_in) (void)hipHostFree(h->stage_in);
  if (h->stage_out) (void)hipHostFree(h->stage_out);
  if (h->tail_flag) (void)hipHostFree(h->tail_flag);
  for (auto& e : h->ev)
    if (e) (void)hipEventDestroy(e);
  if (h->ev_in) (void)hipEventDestroy(h->ev_in);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return MPMPC_OK;
}

int mpmpc_create(const mpmpc_config* cfg, const mpmpc_settings* settings, mpmpc_handle* out) {
  if (!cfg || !out) return fail(MPMPC_E_ARG, "cfg/out is NULL");
  *out = nullptr;
  if (cfg->N < 3 || cfg->N > MPMPC_MAX_HORIZON)
    return fail(MPMPC_E_ARG, "horizon N must satisfy 3 <= N <= " + std::to_string(MPMPC_MAX_HORIZON));
  if (cfg->max_batch < 1) return fail(MPMPC_E_ARG, "max_batch must be >= 1");
  if (!(cfg->wheelbase > 0)) return fail(MPMPC_E_ARG, "wheelbase must be > 0");
  for (int i = 0; i < 3; ++i)
    if (!(cfg->Q[i] >= 0) || !(cfg->QN[i] >= 0)) return fail(MPMPC_E_ARG, "Q, QN diagonals must be >= 0");
  {
    // Q, QN positive semidefinite: principal minors of the symmetric 3 x 3 (with a rounding allowance)
    auto psd3 = [](const double* dg, const double* od) {
      const double a = dg[0], b = dg[1], c = dg[2], d = od[0], e = od[1], f = od[2];
      if (!std::isfinite(d) || !std::isfinite(e) || !std::isfinite(f)) return false;
      const double tol = 1e-12 * (1.0 + a * a + b * b + c * c);
      const double m2a = a * b - d * d, m2b = a * c - e * e, m2c = b * c - f * f;
      const double det = a * (b * c - f * f) - d * (d * c - f * e) + e * (d * f - b * e);
      return !(m2a < -tol || m2b < -tol || m2c < -tol || det < -tol * (1.0 + a + b + c));
    };
    if (!psd3(cfg->QN, cfg->QN_offdiag)) return fail(MPMPC_E_ARG, "QN must be finite and positive semidefinite");
    if (!psd3(cfg->Q, cfg->Q_offdiag)) return fail(MPMPC_E_ARG, "Q must be finite and positive semidefinite");
    const double r = cfg->R_offdiag[0];
    if (!std::isfinite(r) || cfg->R[0] * cfg->R[1] - r * r < -1e-12 * (1.0 + cfg->R[0] * cfg->R[0] + cfg->R[1] * cfg->R[1]))
      return fail(MPMPC_E_ARG, "R must be finite and positive semidefinite");
  }
  for (int i = 0; i < 2; ++i) {
    if (!(cfg->R[i] >= 0)) return fail(MPMPC_E_ARG, "R diagonal must be >= 0");
    if (!std::isfinite(cfg->umin[i]) || !std::isfinite(cfg->umax[i]) || cfg->umin[i] > cfg->umax[i])
      return fail(MPMPC_E_ARG, "input bounds must be finite with umin <= umax");
  }
  mpmpc_settings st;
  if (settings) st = *settings; else mpmpc_default_settings(&st);
  if (int rc = check_settings(&st)) return rc;
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (cfg->device < 0 || cfg->device >= ndev)
    return fail(MPMPC_E_HIP, "no such HIP device (this library has no CPU fallback)");
  HIP_TRY(hipSetDevice(cfg->device));
  mpmpc_handle h = new (std::nothrow) mpmpc_handle_s();
  if (!h) return fail(MPMPC_E_STATE, "out of host memory");
  h->cfg = *cfg;
  h->st = st;
  h->ld = host_stage_ld(cfg->N);
  h->n = 5 * cfg->N + 3;
  h->m = 8 * cfg->N + 6;
  const size_t B = (size_t)cfg->max_batch;
#define ALLOC(ptr, count)                                                                \
  do {                                                                                   \
    hipError_t e_ = hipMalloc((void**)&(ptr), (count) * sizeof(*(ptr)));                 \
    if (e_ != hipSuccess) {                                                              \
      mpmpc_destroy(h);                                                                  \
      return fail(MPMPC_E_HIP, std::string("hipMalloc " #ptr ": ") + hipGetErrorString(e_)); \
    }                                                                                    \
  } while (0)
  const BlockLayout lay = block_layout(cfg->N, cfg->max_batch);
  ALLOC(h->in_block, lay.in_end);
  ALLOC(h->out_block, lay.out_end);
  lay_out(h, cfg->max_batch);
  ALLOC(h->qp, (size_t)MPMPC_NUM_FIELDS * B * h->ld);
  ALLOC(h->tail, 3 * (B + 1));
  (void)hipMemset(h->tail, 0, 3 * (B + 1) * sizeof(int));
  if (hipHostMalloc(reinterpret_cast<void**>(&h->tail_flag), 2 * sizeof(unsigned), hipHostMallocDefault) != hipSuccess) {
    h->tail_flag = nullptr;
    mpmpc_destroy(h);
    return fail(MPMPC_E_HIP, "hipHostMalloc tail_flag");
  }
  h->tail_flag[0] = h->tail_flag[1] = 0u;
#undef ALLOC
  {
    const size_t STAGE_LIMIT = 64u << 20;
    const size_t in_b = lay.in_end, out_b = lay.out_end;
    if (in_b <= STAGE_LIMIT && hipHostMalloc((void**)&h->stage_in, in_b, hipHostMallocDefault) == hipSuccess) h->stage_in_bytes = in_b;
    else h->stage_in = nullptr;
    if (out_b <= STAGE_LIMIT && hipHostMalloc((void**)&h->stage_out, out_b, hipHostMallocDefault) == hipSuccess) h->stage_out_bytes = out_b;
    else h->stage_out = nullptr;
  }
  hipError_t e = hipStreamCreate(&h->stream);
  for (int i = 0; i < 3 && e == hipSuccess; ++i) e = hipEventCreate(&h->ev[i]);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_in, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_order, hipEventDisableTiming);
  if (e != hipSuccess) {
    mpmpc_destroy(h);
    return fail(MPMPC_E_HIP, std::string("stream/event creation: ") + hipGetErrorString(e));
  }
  // (the further launch slots of pipelined resident launches are allocated by the first such launch: next_slot)
  *out = h;
  return MPMPC_OK;
}

int mpmpc_set_packing(mpmpc_handle h, int32_t lanes_per_instance) {
  if (!h) return fail(MPMPC_E_ARG, "handle is NULL");
  MPMPC_SETTLE(h);
  const int g = lanes_per_instance;
  if (g != 0 && g != 16 && g != 32 && g != 64 && !(g == 128 && h->cfg.N + 1 > 64) && !(g == 256 && h->cfg.N + 1 > 128))
    return fail(MPMPC_E_ARG, "lanes_per_instance must be 0 (auto), 16, 32 or 64 (128 / 256: the one-stage workgroup kernels of horizons 64 .. 127 / 128 .. 255)");
  // (16 lanes for 17 .. 32 stages: TWO stages per lane, four instances per wavefront - the reduced-native batch kernel only)
  // (64 / 128 at 65 .. 128 stages: two stages per lane in one wavefront - the default there - or the workgroup kernel)
  const bool two = (g == 16 && h->cfg.N + 1 > 16 && h->cfg.N + 1 <= 32) || ((g == 64 || g == 128) && h->cfg.N + 1 > 64 && h->cfg.N + 1 <= 128) ||
                   ((g == 128 || g == 256) && h->cfg.N + 1 > 128);
  if (g != 0 && h->cfg.N + 1 > g && !two) return fail(MPMPC_E_ARG, "lanes_per_instance must hold the N + 1 stages of an instance, or half of them at 16 (horizons above 63 take a workgroup: only 0)");
  h->force_lanes = g;
  return MPMPC_OK;
}

int mpmpc_set_tail_kernel(mpmpc_handle h, int32_t reduced_native) {
  if (!h) return fail(MPMPC_E_ARG, "handle is NULL");
  MPMPC_SETTLE(h);
  if (reduced_native < 0 || reduced_native > 2) return fail(MPMPC_E_ARG, "reduced_native must be 0, 1 or 2");
  h->lean_tail = reduced_native != 0;
  h->lean_tail_single = reduced_native == 2;
  return MPMPC_OK;
}

int mpmpc_set_settings(mpmpc_handle h, const mpmpc_settings* settings) {
  if (!h) return fail(MPMPC_E_ARG, "handle is NULL");
  MPMPC_SETTLE(h);
  if (int rc = check_settings(settings)) return rc;
  h->st = *settings;
  return MPMPC_OK;
}

static int upload_table(mpmpc_handle h, double** dst, const double* src, size_t count) {
  if (*dst) { HIP_TRY(hipFree(*dst)); *dst = nullptr; }
  HIP_TRY(hipMalloc((void**)dst, count * sizeof(double)));
  HIP_TRY(hipMemcpyAsync(*dst, src, count * sizeof(double), hipMemcpyHostToDevice, h->stream));
  return MPMPC_OK;
}

int mpmpc_set_path(mpmpc_handle h, int32_t n_wp, const double* kappa, const double* v_ref, const double* ds_next) {
  if (!h || !kappa || !v_ref || !ds_next) return fail(MPMPC_E_ARG, "NULL argument");
  MPMPC_SETTLE(h);
  if (n_wp < 2) return fail(MPMPC_E_ARG, "need at least 2 waypoints");
  for (int i = 0; i < n_wp; ++i)
    if (!(v_ref[i] > 0) || !std::isfinite(kappa[i]) || !(ds_next[i] >= 0))
      return fail(MPMPC_E_ARG, "path table: need v_ref > 0, finite kappa, ds_next >= 0");
  HIP_TRY(hipSetDevice(h->cfg.device));
  if (int rc = upload_table(h, &h->kappa, kappa, n_wp)) return rc;
  if (int rc = upload_table(h, &h->v_ref, v_ref, n_wp)) return rc;
  if (int rc = upload_table(h, &h->ds_next, ds_next, n_wp)) return rc;
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (h->n_wp != n_wp) { h->n_cols = 0; }   // a corridor table of another path is void
  h->n_wp = n_wp;
  return MPMPC_OK;
}

int mpmpc_set_corridor(mpmpc_handle h, int32_t n_wp, int32_t n_cols, const double* ub, const double* lb) {
  if (!h || !ub || !lb) return fail(MPMPC_E_ARG, "NULL argument");
  MPMPC_SETTLE(h);
  if (h->n_wp == 0 || n_wp != h->n_wp) return fail(MPMPC_E_STATE, "set the path first; n_wp must match it");
  if (n_cols < h->cfg.N) return fail(MPMPC_E_ARG, "corridor table needs n_cols >= N");
  HIP_TRY(hipSetDevice(h->cfg.device));
  if (int rc = upload_table(h, &h->ub_tab, ub, (size_t)n_wp * n_cols)) return rc;
  if (int rc = upload_table(h, &h->lb_tab, lb, (size_t)n_wp * n_cols)) return rc;
  HIP_TRY(hipStreamSynchronize(h->stream));
  h->n_cols = n_cols;
  return MPMPC_OK;
}

int mpmpc_set_map(mpmpc_handle h, int32_t height, int32_t width, const int8_t* data, double origin_x,
                  double origin_y, double resolution) {
  if (!h || !data) return fail(MPMPC_E_ARG, "NULL argument");
  MPMPC_SETTLE(h);
  if (height < 1 || width < 1 || !(resolution > 0)) return fail(MPMPC_E_ARG, "map needs positive size and resolution");
  if (height > COR_MAX_SIDE || width > COR_MAX_SIDE) return fail(MPMPC_E_ARG, "map sides are limited to 65534 cells");
  HIP_TRY(hipSetDevice(h->cfg.device));
  if (h->map) { HIP_TRY(hipFree(h->map)); h->map = nullptr; }
  HIP_TRY(hipMalloc((void**)&h->map, (size_t)height * width));
  HIP_TRY(hipMemcpyAsync(h->map, data, (size_t)height * width, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  h->map_h = height; h->map_w = width; h->map_ox = origin_x; h->map_oy = origin_y; h->map_res = resolution;
  return MPMPC_OK;
}

int mpmpc_set_path_geometry(mpmpc_handle h, int32_t n_wp, const double* x, const double* y, const double* psi,
                            const double* border_ub, const double* border_lb) {
  if (!h || !x || !y || !psi || !border_ub || !border_lb) return fail(MPMPC_E_ARG, "NULL argument");
  MPMPC_SETTLE(h);
  if (h->n_wp == 0 || n_wp != h->n_wp) return fail(MPMPC_E_STATE, "set the path first; n_wp must match it");
  HIP_TRY(hipSetDevice(h->cfg.device));
  if (int rc = upload_table(h, &h->gx, x, n_wp)) return rc;
  if (int rc = upload_table(h, &h->gy, y, n_wp)) return rc;
  if (int rc = upload_table(h, &h->gpsi, psi, n_wp)) return rc;
  if (int rc = upload_table(h, &h->bub, border_ub, 2 * (size_t)n_wp)) return rc;
  if (int rc = upload_table(h, &h->blb, border_lb, 2 * (size_t)n_wp)) return rc;
  {
    // everything of a waypoint that needs libm, from the HOST's libm: the device tables then hold no device-libm
    // result (bit-exact against the reference's tables, golden G3)
    std::vector<double> trig((size_t)n_wp * COR_TRIG);
    for (int i = 0; i < n_wp; ++i) cor_trig_row(psi[i], trig.data() + (size_t)i * COR_TRIG);
    if (int rc = upload_table(h, &h->gtrig, trig.data(), trig.size())) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));      // `trig` leaves scope
  }
  h->host_bub.assign(border_ub, border_ub + 2 * (size_t)n_wp);
  h->host_blb.assign(border_lb, border_lb + 2 * (size_t)n_wp);
  if (h->segs) { HIP_TRY(hipFree(h->segs)); h->segs = nullptr; }
  if (h->nseg) { HIP_TRY(hipFree(h->nseg)); h->nseg = nullptr; }
  if (!h->bad) HIP_TRY(hipMalloc((void**)&h->bad, 2 * sizeof(int)));
  HIP_TRY(hipMalloc((void**)&h->segs, sizeof(double) * (4 * COR_MAXSEG + COR_WPC) * (size_t)n_wp));   // + cor_forced rows
  HIP_TRY(hipMalloc((void**)&h->nseg, sizeof(int) * (size_t)n_wp));
  HIP_TRY(hipStreamSynchronize(h->stream));
  h->geom_n = n_wp;
  return MPMPC_OK;
}

int mpmpc_build_corridor(mpmpc_handle h, int32_t n_cols, double min_width, double safety_margin, double* ub_out,
                         double* lb_out, int32_t* bad_rows) {
  if (!h) return fail(MPMPC_E_ARG, "handle is NULL");
  MPMPC_SETTLE(h);
  if (!h->map || h->geom_n == 0 || h->geom_n != h->n_wp)
    return fail(MPMPC_E_STATE, "needs mpmpc_set_path, mpmpc_set_map and mpmpc_set_path_geometry first");
  if (n_cols < h->cfg.N) return fail(MPMPC_E_ARG, "corridor table needs n_cols >= N");
  HIP_TRY(hipSetDevice(h->cfg.device));
  const int n = h->n_wp;
  if (h->n_cols != n_cols || !h->ub_tab) {
    if (h->ub_tab) { HIP_TRY(hipFree(h->ub_tab)); h->ub_tab = nullptr; }
    if (h->lb_tab) { HIP_TRY(hipFree(h->lb_tab)); h->lb_tab = nullptr; }
    HIP_TRY(hipMalloc((void**)&h->ub_tab, sizeof(double) * (size_t)n * n_cols));
    HIP_TRY(hipMalloc((void**)&h->lb_tab, sizeof(double) * (size_t)n * n_cols));
  }
  MapView mv{h->map, h->map_h, h->map_w, h->map_ox, h->map_oy, h->map_res};
  PathGeom pg{h->gx, h->gy, h->gpsi, h->ds_next, n, h->cfg.circular, h->gtrig};
  // the border cells of every waypoint must lie on the map of the moment (numpy indexing in the reference raises
  // IndexError past the upper edges and silently wraps around below zero: here both are an error)
  for (int i = 0; i < n; ++i) {
    int cx[2], cy[2];
    cor_w2m(mv, h->host_bub[2 * i], h->host_bub[2 * i + 1], cx[0], cy[0]);
    cor_w2m(mv, h->host_blb[2 * i], h->host_blb[2 * i + 1], cx[1], cy[1]);
    for (int e = 0; e < 2; ++e)
      if (cx[e] < 0 || cx[e] >= h->map_w || cy[e] < 0 || cy[e] >= h->map_h)
        return fail(MPMPC_E_ARG, "border cell of waypoint " + std::to_string(i) + " lies outside the map");
  }
  HIP_TRY(hipMemsetAsync(h->bad, 0, 2 * sizeof(int), h->stream));
  double* wpc = h->segs + (size_t)4 * COR_MAXSEG * n;
  hipLaunchKernelGGL(mpmpc_free_segments_kernel, dim3(n), dim3(64), 0, h->stream, mv, pg, h->bub, h->blb, min_width,
                     safety_margin, h->segs, h->nseg, wpc, h->bad + 1);
  hipLaunchKernelGGL(mpmpc_corridor_select_kernel, dim3((n * n_cols + 255) / 256), dim3(256), 0, h->stream, pg, h->segs,
                     h->nseg, n_cols, safety_margin, h->ub_tab, h->lb_tab, h->bad, wpc);
  HIP_TRY(hipGetLastError());
  int bad2[2] = {0, 0};
  HIP_TRY(hipMemcpyAsync(bad2, h->bad, 2 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  if (ub_out) HIP_TRY(hipMemcpyAsync(ub_out, h->ub_tab, sizeof(double) * (size_t)n * n_cols, hipMemcpyDeviceToHost, h->stream));
  if (lb_out) HIP_TRY(hipMemcpyAsync(lb_out, h->lb_tab, sizeof(double) * (size_t)n * n_cols, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (bad2[1] != 0) {
    h->n_cols = 0;        // the table is not usable
    return fail(MPMPC_E_ARG, bad2[1] == 1 ? "a waypoint's border line has more than 1024 cells (COR_CELL_CAP)"
                                           : "a waypoint's border line has more than 8 free segments (COR_MAXSEG)");
  }
  const int bad = bad2[0];
  if (bad_rows) *bad_rows = bad;
  h->n_cols = n_cols;
  return MPMPC_OK;
}

int mpmpc_rollout_init(mpmpc_handle h, int32_t B, double Ts, const double* cum_lengths, const double* s,
                       const double* pose, const double* cc0) {
  if (!h || !cum_lengths || !s || !pose) return fail(MPMPC_E_ARG, "NULL argument");
  MPMPC_SETTLE(h);
  if (B < 1 || B > h->cfg.max_batch) return fail(MPMPC_E_ARG, "B must be in [1, max_batch]");
  if (!(Ts > 0)) return fail(MPMPC_E_ARG, "Ts must be > 0");
  if (h->n_wp == 0 || h->geom_n != h->n_wp) return fail(MPMPC_E_STATE, "needs mpmpc_set_path and mpmpc_set_path_geometry");
  if (h->n_cols == 0) return fail(MPMPC_E_STATE, "needs a corridor table (mpmpc_set_corridor / mpmpc_build_corridor)");
  HIP_TRY(hipSetDevice(h->cfg.device));
  const size_t mb = (size_t)h->cfg.max_batch;
  if (!h->ro_s) {
    HIP_TRY(hipMalloc((void**)&h->ro_s, sizeof(double) * mb));
    HIP_TRY(hipMalloc((void**)&h->ro_pose, sizeof(double) * 3 * mb));
    HIP_TRY(hipMalloc((void**)&h->ro_u, sizeof(double) * 2 * mb));
    HIP_TRY(hipMalloc((void**)&h->ro_counter, sizeof(int) * mb));
    HIP_TRY(hipMalloc((void**)&h->ro_alive, sizeof(int) * mb));
    HIP_TRY(hipMalloc((void**)&h->ro_act, sizeof(int) * mb * h->ld));
    HIP_TRY(hipMalloc((void**)&h->ro_shift, sizeof(int) * mb));
  }
  if (int rc = upload_table(h, &h->ro_cum, cum_lengths, h->n_wp)) return rc;
  const int N = h->cfg.N;
  if (B != h->laid_out) lay_out(h, B);
  HIP_TRY(hipMemcpyAsync(h->ro_s, s, sizeof(double) * B, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipMemcpyAsync(h->ro_pose, pose, sizeof(double) * 3 * B, hipMemcpyHostToDevice, h->stream));
  if (cc0) HIP_TRY(hipMemcpyAsync(h->cc, cc0, sizeof(double) * 2 * N * B, hipMemcpyHostToDevice, h->stream));
  else HIP_TRY(hipMemsetAsync(h->cc, 0, sizeof(double) * 2 * N * B, h->stream));
  HIP_TRY(hipMemsetAsync(h->ro_counter, 0, sizeof(int) * B, h->stream));
  HIP_TRY(hipMemsetAsync(h->ro_act, 0, sizeof(int) * (size_t)B * h->ld, h->stream));     // no guess yet
  HIP_TRY(hipMemsetAsync(h->ro_shift, 0, sizeof(int) * B, h->stream));
  HIP_TRY(hipMemsetAsync(h->ro_u, 0, sizeof(double) * 2 * B, h->stream));
  HIP_TRY(hipMemsetAsync(h->wp_id, 0, sizeof(int) * B, h->stream));
  HIP_TRY(hipMemsetAsync(h->x0, 0, sizeof(double) * 3 * B, h->stream));
  HIP_TRY(hipMemsetAsync(h->status, 0, sizeof(int) * B, h->stream));
  std::string ones(sizeof(int) * (size_t)B, '\0');
  for (int i = 0; i < B; ++i) ((int*)ones.data())[i] = 1;
  HIP_TRY(hipMemcpyAsync(h->ro_alive, ones.data(), sizeof(int) * B, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  h->ro_Ts = Ts;
  h->ro_B = B;
  h->ro_valid = true;
  h->have_rows = false;       // the corridor comes from the table
  h->uploaded = B;
  return MPMPC_OK;
}

int mpmpc_rollout_set_counters(mpmpc_handle h, int32_t B, const int32_t* counter) {
  if (!h || !counter) return fail(MPMPC_E_ARG, "NULL argument");
  MPMPC_SETTLE(h);
  if (B < 1 || B > h->ro_B || !h->ro_valid) return fail(MPMPC_E_STATE, "call mpmpc_rollout_init for at least B cars first");
  for (int i = 0; i < B; ++i)
    if (counter[i] < 0 || counter[i] >= h->cfg.N - 1) return fail(MPMPC_E_ARG, "infeasibility counters must be in [0, N - 2]");
  HIP_TRY(hipSetDevice(h->cfg.device));
  HIP_TRY(hipMemcpyAsync(h->ro_counter, counter, sizeof(int) * B, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return MPMPC_OK;
}

int mpmpc_rollout_step(mpmpc_handle h, int32_t B, int32_t n_steps) {
  if (!h) return fail(MPMPC_E_ARG, "handle is NULL");
  MPMPC_SETTLE(h);
  if (B < 1 || B > h->ro_B) return fail(MPMPC_E_STATE, "call mpmpc_rollout_init for at least B cars first");
  if (!h->ro_valid)
    return fail(MPMPC_E_STATE, "the rollout's state was overwritten by an upload / solve / assemble on this handle: "
                               "call mpmpc_rollout_init again (or use a second handle for single solves)");
  if (n_steps < 0) return fail(MPMPC_E_ARG, "n_steps must be >= 0");
  HIP_TRY(hipSetDevice(h->cfg.device));
  const int blocks = (B + 255) / 256;
  for (int t = 0; t < n_steps; ++t) {
    hipLaunchKernelGGL(mpmpc_localise_kernel, dim3(blocks), dim3(256), 0, h->stream, B, h->n_wp, h->ro_cum, h->gx, h->gy,
                       h->gpsi, h->ro_s, h->ro_pose, h->ro_alive, h->wp_id, h->x0, h->ro_shift);
    if (int rc = launch_solve(h, B, true, false)) return rc;      // (the plant step reads z and the status only)
    hipLaunchKernelGGL(mpmpc_advance_kernel, dim3((B * h->cfg.N + 255) / 256), dim3(256), 0, h->stream, B, h->cfg.N, h->cfg.wheelbase, h->ro_Ts,
                       h->kappa, h->wp_id, h->x0, h->status, h->z, h->cc, h->ro_counter, h->ro_alive, h->ro_pose, h->ro_s,
                       h->ro_u);
  }
  HIP_TRY(hipGetLastError());
  return MPMPC_OK;
}

int mpmpc_rollout_warm_start(mpmpc_handle h, int32_t enable) {
  if (!h) return fail(MPMPC_E_ARG, "handle is NULL");
  MPMPC_SETTLE(h);
  h->ro_warm = enable < 0 ? 0 : (enable > 2 ? 2 : enable);
  return MPMPC_OK;
}

int mpmpc_rollout_state(mpmpc_handle h, int32_t B, double* s, double* pose, double* cc, int32_t* wp_id, double* x0,
                        double* u_last, int32_t* status, int32_t* counter, int32_t* alive) {
  if (!h) return fail(MPMPC_E_ARG, "handle is NULL");
  MPMPC_SETTLE(h);
  if (B < 1 || B > h->ro_B) return fail(MPMPC_E_STATE, "call mpmpc_rollout_init for at least B cars first");
  if (!h->ro_valid)
    return fail(MPMPC_E_STATE, "the rollout's state was overwritten by an upload / solve / assemble on this handle: "
                               "call mpmpc_rollout_init again");
  HIP_TRY(hipSetDevice(h->cfg.device));
  const int N = h->cfg.N;
#define PULL(dst, src, bytes) if (dst) HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->stream))
  PULL(s, h->ro_s, sizeof(double) * B);
  PULL(pose, h->ro_pose, sizeof(double) * 3 * B);
  PULL(cc, h->cc, sizeof(double) * 2 * N * B);
  PULL(wp_id, h->wp_id, sizeof(int) * B);
  PULL(x0, h->x0, sizeof(double) * 3 * B);
  PULL(u_last, h->ro_u, sizeof(double) * 2 * B);
  PULL(status, h->status, sizeof(int) * B);
  PULL(counter, h->ro_counter, sizeof(int) * B);
  PULL(alive, h->ro_alive, sizeof(int) * B);
#undef PULL
  HIP_TRY(hipStreamSynchronize(h->stream));
  return MPMPC_OK;
}

int mpmpc_upload(mpmpc_handle h, int32_t B, const int32_t* wp_id, const double* x0, const double* cc_prev,
                 const double* lb, const double* ub) {
  if (!h || !wp_id || !x0 || !cc_prev) return fail(MPMPC_E_ARG, "NULL argument");
  MPMPC_SETTLE(h);
  if (B < 1 || B > h->cfg.max_batch) return fail(MPMPC_E_ARG, "B must be in [1, max_batch]");
  if (h->n_wp == 0) return fail(MPMPC_E_STATE, "no path set (mpmpc_set_path)");
  if ((lb == nullptr) != (ub == nullptr)) return fail(MPMPC_E_ARG, "lb and ub must both be given or both NULL");
  if (!lb && h->n_cols == 0) return fail(MPMPC_E_STATE, "no corridor rows given and no corridor table set");
  const int N = h->cfg.N;
  for (int i = 0; i < B; ++i) {
    if (wp_id[i] < 0 || wp_id[i] >= h->n_wp) return fail(MPMPC_E_ARG, "wp_id out of range");
    // an open path ends the run when the horizon passes its last waypoint (src/reference_path.py:367-369)
    if (!h->cfg.circular && wp_id[i] + N >= h->n_wp) return fail(MPMPC_E_ARG, "Reached end of path!");
  }
  HIP_TRY(hipSetDevice(h->cfg.device));
  h->ro_valid = false;        // the batch blocks are about to be re-laid out and overwritten
  if (B != h->laid_out) lay_out(h, B);
  const BlockLayout L = block_layout(N, B);
  if (h->stage_in) {
    // through pinned staging: the batch is gathered in the device block's own layout and goes over in one copy.
    // The caller's buffers are free as soon as they are gathered, so nothing waits for the copy here - only the
    // NEXT upload does, before it overwrites the staging buffer.
    if (h->in_flight) HIP_TRY(hipEventSynchronize(h->ev_in));
    char* st = h->stage_in;
    std::memcpy(st + L.wp_id, wp_id, sizeof(int) * B);
    std::memcpy(st + L.x0, x0, sizeof(double) * 3 * B);
    std::memcpy(st + L.cc, cc_prev, sizeof(double) * 2 * N * B);
    if (lb) {
      std::memcpy(st + L.lb, lb, sizeof(double) * N * B);
      std::memcpy(st + L.ub, ub, sizeof(double) * N * B);
    }
    HIP_TRY(hipMemcpyAsync(h->in_block, st, lb ? L.in_end : L.in_end_cc, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipEventRecord(h->ev_in, h->stream));
    h->in_flight = true;
  } else {
    // (no staging block of this size: through the bounce buffer - asynchronous on the device's side, the caller's buffers are
    //  free when the call returns)
    if (int rc = upload_through_bounce(h, h->wp_id, wp_id, sizeof(int) * B)) return rc;
    if (int rc = upload_through_bounce(h, h->x0, x0, sizeof(double) * 3 * B)) return rc;
    if (int rc = upload_through_bounce(h, h->cc, cc_prev, sizeof(double) * 2 * N * B)) return rc;
    if (lb) {
      if (int rc = upload_through_bounce(h, h->lb, lb, sizeof(double) * N * B)) return rc;
      if (int rc = upload_through_bounce(h, h->ub, ub, sizeof(double) * N * B)) return rc;
    }
  }
  h->have_rows = lb != nullptr;
  h->uploaded = B;
  return MPMPC_OK;
}

constexpr size_t BOUNCE_HALF = 8u << 20;
static int upload_through_bounce(mpmpc_handle h, void* dst, const void* src, size_t bytes) {
  if (!h->bounce) {
    HIP_TRY(hipHostMalloc((void**)&h->bounce, 2 * BOUNCE_HALF, hipHostMallocDefault));
    for (auto& e : h->ev_bounce) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  for (size_t off = 0; off < bytes; off += BOUNCE_HALF) {
    const int k = h->bounce_turn;
    if (h->bounce_busy[k]) HIP_TRY(hipEventSynchronize(h->ev_bounce[k]));      // its last copy has left this half
    const size_t n = bytes - off < BOUNCE_HALF ? bytes - off : BOUNCE_HALF;
    char* half = h->bounce + (size_t)k * BOUNCE_HALF;
    std::memcpy(half, static_cast<const char*>(src) + off, n);
    HIP_TRY(hipMemcpyAsync(static_cast<char*>(dst) + off, half, n, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipEventRecord(h->ev_bounce[k], h->stream));
    h->bounce_busy[k] = true;
    h->bounce_turn = 1 - k;
  }
  return MPMPC_OK;
}

static int launch_assemble(mpmpc_handle h, int B) {
  PathTables t{h->kappa, h->v_ref, h->ds_next, h->n_wp, h->ub_tab, h->lb_tab, h->n_cols};
  const int total = B * h->ld;
  const int blocks = (total + K1_THREADS - 1) / K1_THREADS;
  // (tuning knob of profiles/k1_timing.py: MPMPC_K1_NT = 0 forces plain stores)
  static const int force_nt = std::getenv("MPMPC_K1_NT") ? std::atoi(std::getenv("MPMPC_K1_NT")) : -1;
  const bool nt = force_nt != 0;
  const size_t k1_lds = h->n_wp <= K1_LDS_WP ? 3 * sizeof(double) * (size_t)h->n_wp : 0;
#define K1_LAUNCH(V)                                                                                                       \
  hipLaunchKernelGGL(mpmpc_assemble_kernel<V>, dim3(blocks), dim3(K1_THREADS), k1_lds, h->stream, h->cfg, t, B, h->ld, h->wp_id, \
                     h->x0, h->cc, h->have_rows ? h->lb : nullptr, h->have_rows ? h->ub : nullptr, h->qp)
  if (nt) K1_LAUNCH(true);
  else K1_LAUNCH(false);
#undef K1_LAUNCH
  HIP_TRY(hipGetLastError());
  return MPMPC_OK;
}

}  // extern "C"  (the dispatch functions below are templates)

// ---------------------------------------------------------------------------------------------------- launch dispatch
// What a solve launch hands to its kernels, and ONE small function per kernel family that turns it into the launch of an
// instantiation.  launch_solve below decides WHICH kernels run (policy: packing, warm start, tail deferral); the tables after
// the families map that decision - (lanes per instance, chain split, variant, warm) - to the instantiation (VERDICT r5 item 8:
// the six nested launch macros this replaces multiplied with every new layout).
struct SolveLaunch {
  mpmpc_handle h;
  SolverParams prm;
  AssembleIn ain;
  int B;
  double* y_out;
  int *tail_cur, *tail_next, *tail2;      // the list this launch fills / the next launch's (emptied here) / what the tail kernel leaves
  int* warm_act;
  const int* warm_shift;
};
// general kernel, one instance per wave: the whole solve (mode 0) or the tail of a reduced-native launch (mode 2, list `tail`)
template <int C, bool WARM, int VAR>
static void go_general(const SolveLaunch& a, int mode, int blocks, int* tail) {
  mpmpc_handle h = a.h;
  hipLaunchKernelGGL((mpmpc_solve_kernel<64, C, WARM, VAR>), dim3(blocks), dim3(64), 0, h->stream, h->cfg, a.prm, a.B, h->ld, a.ain, h->z, h->u0,
                     h->status, h->iters, h->resid, a.y_out, mode, tail, WARM ? a.warm_act : nullptr, WARM ? a.warm_shift : nullptr, a.tail_next);
}
using GeneralFn = void (*)(const SolveLaunch&, int, int, int*);
// [C == 32][warm][VAR: 0 full problem, 1 full weights, 2 reduced polish, 3 free e_psi / t]
#define MPMPC_GENERAL_ROW(C, W) {go_general<C, W, 0>, go_general<C, W, 1>, go_general<C, W, 2>, go_general<C, W, 3>}
static const GeneralFn kGeneral[2][2][4] = {{MPMPC_GENERAL_ROW(16, false), MPMPC_GENERAL_ROW(16, true)},
                                            {MPMPC_GENERAL_ROW(32, false), MPMPC_GENERAL_ROW(32, true)}};
#undef MPMPC_GENERAL_ROW
// (knob of the occupancy experiment, profiles/r3/occupancy.txt: MPMPC_RN_OCC=1 pads every block of the one-stage reduced-native
//  kernels with 20 KB of unused dynamic LDS - 40 KB per wave, four waves per CU, ONE per SIMD - so that the same code object can
//  be timed at one and at two waves per SIMD)
static int rn_pad() {
  static const int pad = (std::getenv("MPMPC_RN_OCC") && std::atoi(std::getenv("MPMPC_RN_OCC")) == 1) ? 20 * 1024 : 0;
  return pad;
}
// K2r: the reduced-native batch kernel, 1 / 2 / 4 instances per wave
template <int G, int C, bool WARM>
static void go_reduced(const SolveLaunch& a, int blocks) {
  mpmpc_handle h = a.h;
  hipLaunchKernelGGL((mpmpc_reduced_kernel<G, C, WARM>), dim3(blocks), dim3(64), rn_pad(), h->stream, h->cfg, a.prm, a.B, h->ld, a.ain, h->z, h->u0,
                     h->status, h->iters, h->resid, a.y_out, a.tail_cur, a.warm_act, a.warm_shift, a.tail_next, h->tail_flag, h->seq, a.tail2);
}
using ReducedFn = void (*)(const SolveLaunch&, int);
// [layout: <64,16>, <64,32>, <32,16>, <16,16>][warm]
static const ReducedFn kReduced[4][2] = {{go_reduced<64, 16, false>, go_reduced<64, 16, true>}, {go_reduced<64, 32, false>, go_reduced<64, 32, true>},
                                         {go_reduced<32, 16, false>, go_reduced<32, 16, true>}, {go_reduced<16, 16, false>, go_reduced<16, 16, true>}};
static int reduced_layout(int G, int C) { return G == 64 ? (C == 16 ? 0 : 1) : (G == 32 ? 2 : 3); }
// K2t: its twin for a terminal cost on the time state (one instance per wave)
template <int C>
static void go_reduced_t(const SolveLaunch& a, int blocks) {
  mpmpc_handle h = a.h;
  hipLaunchKernelGGL((mpmpc_reduced_t_kernel<64, C>), dim3(blocks), dim3(64), rn_pad(), h->stream, h->cfg, a.prm, a.B, a.ain, h->z, h->u0, h->status,
                     h->iters, h->resid, a.y_out, a.tail_cur, a.tail_next, h->tail_flag, h->seq);
}
// K2p: the reduced-native tail kernel on the list `tail_cur`; what it leaves goes to tail2
template <int G, int C>
static void go_reduced_tail(const SolveLaunch& a, int blocks) {
  mpmpc_handle h = a.h;
  hipLaunchKernelGGL((mpmpc_reduced_tail_kernel<G, C>), dim3(blocks), dim3(64), rn_pad(), h->stream, h->cfg, a.prm, a.B, a.ain, h->z, h->u0, h->status,
                     h->iters, h->resid, a.y_out, a.tail_cur, a.tail_next, a.tail2, h->tail_flag + 1, h->seq);
}
// K2r2 / K2t2 / K2p2: two stages per lane (lane_pair.hpp) - GB lanes per instance
template <int GB>
static void go_pair(const SolveLaunch& a, int blocks) {
  mpmpc_handle h = a.h;
  hipLaunchKernelGGL(mpmpc_reduced_pair_kernel<GB>, dim3(blocks), dim3(64), 0, h->stream, h->cfg, a.prm, a.B, h->ld, a.ain, h->z, h->u0, h->status,
                     h->iters, h->resid, a.y_out, a.tail_cur, a.tail_next, h->tail_flag, h->seq, a.tail2);
}
template <int GB>
static void go_pair_t(const SolveLaunch& a) {
  mpmpc_handle h = a.h;
  hipLaunchKernelGGL(mpmpc_reduced_t_pair_kernel<GB>, dim3(a.B), dim3(64), 0, h->stream, h->cfg, a.prm, a.B, a.ain, h->z, h->u0, h->status, h->iters,
                     h->resid, a.y_out, a.tail_cur);
}
template <int GB>
static void go_pair_tail(const SolveLaunch& a) {
  mpmpc_handle h = a.h;
  hipLaunchKernelGGL(mpmpc_reduced_tail_pair_kernel<GB>, dim3(a.B), dim3(64), 0, h->stream, h->cfg, a.prm, a.B, a.ain, h->z, h->u0, h->status, h->iters,
                     h->resid, a.y_out, a.tail_cur, a.tail_next);
}
// K2b / K2rb: one instance per WORKGROUP of G = 128 / 256 lanes.  The kernels need more dynamic LDS than the default limit: the
// attribute is set once per function AND DEVICE (one process may drive several - sharded.py, bench.py --single-process - and the
// attribute belongs to the function on the device that is current; two threads may both set it once: the call is idempotent)
constexpr int MAX_DEVICES = 64;
static int device_slot(mpmpc_handle h) { return h->cfg.device >= 0 && h->cfg.device < MAX_DEVICES ? h->cfg.device : 0; }
template <int G, int VAR>
static int go_block(const SolveLaunch& a, const int* tail) {
  mpmpc_handle h = a.h;
  static std::atomic<bool> attr_set[MAX_DEVICES];
  if (!attr_set[device_slot(h)]) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&mpmpc_solve_block_kernel<G, VAR>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)LaneBlock<G>::lds_bytes));
    attr_set[device_slot(h)] = true;
  }
  hipLaunchKernelGGL((mpmpc_solve_block_kernel<G, VAR>), dim3(a.B), dim3(G), LaneBlock<G>::lds_bytes, h->stream, h->cfg, a.prm, a.B, a.ain, h->z, h->u0,
                     h->status, h->iters, h->resid, a.y_out, tail);
  return MPMPC_OK;
}
using BlockFn = int (*)(const SolveLaunch&, const int*);
static const BlockFn kBlock[2][3] = {{go_block<128, 0>, go_block<128, 1>, go_block<128, 2>}, {go_block<256, 0>, go_block<256, 1>, go_block<256, 2>}};
template <int G>
static int go_rblock(const SolveLaunch& a) {
  mpmpc_handle h = a.h;
  using LB = LaneBlock<G, RNB_SLOTS>;
  static std::atomic<bool> attr_set[MAX_DEVICES];
  if (!attr_set[device_slot(h)]) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&mpmpc_reduced_block_kernel<G>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LB::lds_bytes));
    attr_set[device_slot(h)] = true;
  }
  hipLaunchKernelGGL((mpmpc_reduced_block_kernel<G>), dim3(a.B), dim3(G), LB::lds_bytes, h->stream, h->cfg, a.prm, a.B, a.ain, h->z, h->u0, h->status, h->iters,
                     h->resid, a.y_out, a.tail_cur);
  return MPMPC_OK;
}

static int go_pair_block(const SolveLaunch& a) {
  mpmpc_handle h = a.h;
  using LB = LaneBlock<128, RNB2_SLOTS, 128, RNB2_XR>;
  static std::atomic<bool> attr_set[MAX_DEVICES];
  if (!attr_set[device_slot(h)]) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&mpmpc_reduced_pair_block_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LB::lds_bytes));
    attr_set[device_slot(h)] = true;
  }
  hipLaunchKernelGGL(mpmpc_reduced_pair_block_kernel, dim3(a.B), dim3(128), LB::lds_bytes, h->stream, h->cfg, a.prm, a.B, a.ain, h->z, h->u0, h->status, h->iters,
                     h->resid, a.y_out, a.tail_cur);
  return MPMPC_OK;
}

static int go_pair_block_tail(const SolveLaunch& a) {
  mpmpc_handle h = a.h;
  using LB = LaneBlock<128, RN2_SLOTS, 128, RNB2_XR>;
  static std::atomic<bool> attr_set[MAX_DEVICES];
  if (!attr_set[device_slot(h)]) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&mpmpc_reduced_tail_pair_block_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LB::lds_bytes));
    attr_set[device_slot(h)] = true;
  }
  hipLaunchKernelGGL(mpmpc_reduced_tail_pair_block_kernel, dim3(a.B), dim3(128), LB::lds_bytes, h->stream, h->cfg, a.prm, a.B, a.ain, h->z, h->u0, h->status,
                     h->iters, h->resid, a.y_out, a.tail_cur, a.tail_next);
  return MPMPC_OK;
}
static int go_pair_block_t(const SolveLaunch& a) {
  mpmpc_handle h = a.h;
  using LB = LaneBlock<128, RN2_SLOTS, 128, RNB2_XR>;
  static std::atomic<bool> attr_set[MAX_DEVICES];
  if (!attr_set[device_slot(h)]) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&mpmpc_reduced_t_pair_block_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LB::lds_bytes));
    attr_set[device_slot(h)] = true;
  }
  hipLaunchKernelGGL(mpmpc_reduced_t_pair_block_kernel, dim3(a.B), dim3(128), LB::lds_bytes, h->stream, h->cfg, a.prm, a.B, a.ain, h->z, h->u0, h->status,
                     h->iters, h->resid, a.y_out, a.tail_cur);
  return MPMPC_OK;
}

// Horizons above 63.  65 .. 128 stages of the reference's own weights (or of a terminal cost on the time state): TWO stages per
// lane, the whole instance in ONE wavefront (K2r2<64> / K2t2<64>), its tail to K2p2<64>, what that leaves to the general solver on
// a workgroup; mpmpc_set_packing(h, 128) keeps round 5's workgroup kernels.  Everything else - longer horizons, full weights,
// bounded e_psi / t - one instance per workgroup of 2 / 4 wavefronts (K2rb in front where the reduction applies).  No packing, no
// deferred tail, cold starts in the closed loop too.
static int launch_long_horizon(mpmpc_handle h, SolveLaunch& a, int tail_only) {
  if (tail_only) return MPMPC_OK;
  const int N = h->cfg.N;
  h->pend = h->pend2 = false;
  const bool fullqn = full_weights(h->cfg);
  const int var = fullqn ? 1 : (reducible(h->cfg, h->st) ? 2 : 0);
  const bool one_wave = N + 1 <= 128 && h->force_lanes != 128;
  int* list1 = h->tail;
  int* list2 = h->tail + ((size_t)h->cfg.max_batch + 1);
  int* list3 = list2 + ((size_t)h->cfg.max_batch + 1);
  const int* tail_blk = nullptr;          // the list the general workgroup kernel works on (null: every instance, the whole solve)
  a.tail_cur = list1;
  if (!fullqn && reduced_native(h->cfg, h->st)) {
    HIP_TRY(hipMemsetAsync(list1, 0, sizeof(int), h->stream));
    tail_blk = list1;
    if (one_wave) {
      // (the kernel empties "the lists of the next launch" - here the two lists behind the one it fills; the tail kernels below
      //  always run, so the host-side flag it stamps is not consulted)
      a.tail_next = list2; a.tail2 = list3;
      go_pair<64>(a, a.B);
      if (h->lean_tail && reduced_native_tail(h->cfg, h->st)) {
        a.tail_next = list2;          // K2p2 reads list 1, fills list 2
        go_pair_tail<64>(a);
        tail_blk = list2;
      }
    } else if (N + 1 > 128 && h->force_lanes != 256) {
      // 129 .. 256 stages: two stages per lane on a workgroup of two wavefronts (K2rb2; mpmpc_set_packing(h, 256): K2rb<256>), its
      // tail to the tail solver on the same layout (the list it fills is emptied first: no kernel of this sequence does it)
      if (int rc = go_pair_block(a)) return rc;
      if (h->lean_tail && reduced_native_tail(h->cfg, h->st)) {
        HIP_TRY(hipMemsetAsync(list2, 0, sizeof(int), h->stream));
        a.tail_next = list2;
        if (int rc = go_pair_block_tail(a)) return rc;
        tail_blk = list2;
      }
    } else if (int rc = (N + 1 <= 128 ? go_rblock<128>(a) : go_rblock<256>(a))) return rc;
  } else if (!fullqn && reduced_native_tt(h->cfg, h->st) && (one_wave || (N + 1 > 128 && h->force_lanes != 256))) {
    // a terminal cost on the time state: K2t's solver with two stages per lane - one wavefront per instance up to 128 stages, a
    // workgroup of two above - and the general workgroup kernel on what it lists
    HIP_TRY(hipMemsetAsync(list1, 0, sizeof(int), h->stream));
    if (one_wave) go_pair_t<64>(a);
    else if (int rc = go_pair_block_t(a)) return rc;
    tail_blk = list1;
  }
  if (int rc = kBlock[N + 1 <= 128 ? 0 : 1][var](a, tail_blk)) return rc;
  HIP_TRY(hipGetLastError());
  return MPMPC_OK;
}

// tail_only: 1 = the deferred tail launches of the last reduced-native launch (observe_tail), nothing else; 2 = of those,
// only the general kernel on what the reduced-native tail kernel left
static int launch_solve(mpmpc_handle h, int B, bool closed_loop, bool want_y, int tail_only, LaunchKind kind) {
  const int N = h->cfg.N;
  h->y_valid = want_y;
  SolveLaunch a{};
  a.h = h;
  a.B = B;
  a.prm = make_params(h->st);
  a.y_out = want_y ? h->y : nullptr;        // nobody wants the multipliers: the kernel skips their stores
  a.ain.tab = PathTables{h->kappa, h->v_ref, h->ds_next, h->n_wp, h->ub_tab, h->lb_tab, h->n_cols};
  a.ain.wp_id = h->wp_id; a.ain.x0 = h->x0; a.ain.cc = h->cc;
  a.ain.lb = h->have_rows ? h->lb : nullptr; a.ain.ub = h->have_rows ? h->ub : nullptr;
  if (N + 1 > 64) return launch_long_horizon(h, a, tail_only);
  // Full weights (Q, R or QN with off-diagonal entries), bounds on e_psi / t or a cost on t rule the reduction out:
  // such configurations run the general kernels, one instance per wave.
  const bool fullqn = full_weights(h->cfg);
  const bool red = reducible(h->cfg, h->st);      // the polish may work on the (e_y, e_psi, kappa) problem
  const bool freex = !fullqn && !red && free_states(h->cfg);
  const int var = fullqn ? 1 : (red ? 2 : (freex ? 3 : 0));
  // The reduced-native kernels (mpmpc_reduced.hpp) take the batch path of every configuration they apply to - cold and
  // warm-started - and only they pack several instances into a wave; the general kernel then sees their tail.
  const bool rnt = reduced_native_tt(h->cfg, h->st);      // ... or their twin for a terminal cost on the time state
  const bool rn = rnt || reduced_native(h->cfg, h->st);
  // lanes per instance: one instance per wave while there are no more instances than SIMDs (1024); beyond that the
  // smallest power of two holding N + 1 stages, so that a wave carries 2 or 4 instances and a SIMD two such waves
  int G = 64;
  bool two = false;
  if (rn && !rnt) {      // (the terminal-time kernels run one instance per wave)
    // A launch that is one of several in flight - a pipelined resident launch (mpmpc_solve_resident with mpmpc_set_pipeline > 1:
    // the default) or the begun half of a split host-buffer call (mpmpc_staged_begin: the caller keeps several handles busy) -
    // is after throughput, and a packed wave does two or four instances for about the instructions of one: such launches pack
    // from 128 instances on - the launches in flight fill the chip, not the waves of one launch (B = 1 024, four launches in
    // flight: 36.5 M solves/s with one instance per wave, measured 41.4 M with two launches of the packed kernel, DESIGN.md
    // section 4).  A launch that is waited for on its own - mpmpc_solve, mpmpc_solve_staged, the timed launch, a handle held at
    // one launch in flight - and the closed loop (one launch per step, each waiting for the one before) keep one instance per
    // wave up to the chip's 1 024 SIMDs: the latency of a single launch is 43 us against 52 (the decision used to
    // follow the handle's pipeline depth alone, so single calls paid the packed kernel's latency).  The choice depends on the
    // entry point, the handle's settings and the batch size only: a given call gives the same bits every time - and every
    // packing returns the same bits anyway (tests).
    const bool throughput = kind == LAUNCH_PIPELINED && h->pipeline > 1 && !closed_loop;
    if (N + 1 <= 32 && B > (throughput ? 128 : 1024)) G = 32;
    if (N + 1 <= 16 && B > (throughput ? 256 : 2048)) G = 16;
    if (h->force_lanes && N + 1 <= h->force_lanes) G = h->force_lanes;      // mpmpc_set_packing
    // TWO stages per lane (K2r2<16>): 17 .. 32 stages in 16 lanes, four instances per wavefront; cold starts only.  Measured
    // slower than <32,16> at two waves per SIMD: never the automatic choice.
    if (h->force_lanes == 16 && N + 1 > 16 && N + 1 <= 32 && !closed_loop) { G = 32; two = true; }
  }
  // closed loop: the previous step's active sets as a first guess.  A launch with one instance per wave ends with
  // its slowest car, and with more than a handful of cars one of them always misses its guess (hit rate 91-93 %
  // per car and step: the miss pays for the attempt AND the normal path, 1024 cars -7 %), so "auto" warm-starts the
  // packed launches (x1.5 at 8192 cars) and the very small fleets (x1.7 at 8 cars) only.
  const bool warm = closed_loop && !rnt && (h->ro_warm == 1 || (h->ro_warm == 2 && (G < 64 || B <= 16)));
  a.warm_act = warm ? h->ro_act : nullptr;
  a.warm_shift = warm ? h->ro_shift : nullptr;
  const int per = two ? 4 : 64 / G;
  const int blocks = (B + per - 1) / per;
  const int C = lane_split(G, N);        // where the two elimination chains of the factorisation meet
  const int C64 = lane_split(64, N);     // ... of the one-instance-per-wave kernels that take a tail
  // tail lists ([0] = count, [1..] = instance ids), two of them used in turn: the tail launch of this step empties the
  // list of the next one, so that no memset has to sit between the launches of consecutive steps
  a.tail_cur = h->tail + (size_t)h->tail_flip * (h->cfg.max_batch + 1);
  a.tail_next = h->tail + (size_t)(1 - h->tail_flip) * (h->cfg.max_batch + 1);
  a.tail2 = h->tail + 2 * ((size_t)h->cfg.max_batch + 1);      // what the reduced-native tail kernel leaves to the general one
  if (tail_only) { a.tail_cur = h->pend_cur; a.tail_next = h->pend_next; }
  else if (rn) { h->tail_flip = 1 - h->tail_flip; h->seq += 1; }
  if (!rn) {
    // the general kernel, one instance per wave, the whole solve
    kGeneral[C == 32][warm][var](a, 0, blocks, a.tail_cur);
    HIP_TRY(hipGetLastError());
    return MPMPC_OK;
  }
  // The tail of a batch launch goes to the reduced-native tail kernel first (K2p: two waves per SIMD instead of one).  The
  // closed loop keeps the general kernel for the whole tail: its step would pay for a third launch every time.
  const bool lean = !rnt && !closed_loop && h->lean_tail && reduced_native_tail(h->cfg, h->st);
  if (!tail_only) {
    if (rnt) (C == 16 ? go_reduced_t<16> : go_reduced_t<32>)(a, blocks);
    else if (two) go_pair<16>(a, blocks);
    else kReduced[reduced_layout(G, C)][warm](a, blocks);
  }
  // The tail is short (infeasible / very hard instances).  One block per instance of the batch: blocks beyond the
  // list's length return at once (an all-empty launch takes 4.8 us at 1 024 blocks, 15 us at 65 536: rocprofv3 kernel
  // trace - which is why it is not enqueued when no tail is expected, below; a grid-stride loop over the list around the solver costs
  // the general kernels 70 registers and puts 148-544 B of scratch into kernels that have none: measured on the code
  // object, not kept).  Its instances carry no guess for the next closed-loop step (act stays 0 from the first launch).
  // Deferred (see the handle): no tail launch is enqueued while the launches the host has seen leave none; the closed
  // loop consumes its results on the device and always launches it.
  h->pend = !tail_only && !closed_loop && h->tail_expect_empty;
  h->pend2 = false;
  if (h->pend) {
    h->pend_B = B; h->pend_y = want_y; h->pend_cur = a.tail_cur; h->pend_next = a.tail_next;
  } else if (lean) {
    if (tail_only != 2) {
      // (the list's order differs from run to run - atomic appends - and with it the two instances that share a wave of the
      //  packed form: the solver's arithmetic does not depend on the partner, Solver::active_set)
      if (C64 == 32) go_reduced_tail<64, 32>(a, B);          // horizons 32 .. 63: one lane per stage, one instance per wave
      else if (!h->lean_tail_single) go_reduced_tail<32, 16>(a, (B + 1) / 2);
      else go_reduced_tail<64, 16>(a, B);
    }
    // ... and the general kernel on what that left: deferred like the tail itself while the launches seen leave nothing
    h->pend2 = tail_only != 2 && h->tail2_expect_empty;
    if (h->pend2) {
      h->pend_B = B; h->pend_y = want_y; h->pend_cur = a.tail_cur; h->pend_next = a.tail_next;
    } else kGeneral[C64 == 32][0][var](a, 2, B, a.tail2);
  } else kGeneral[C64 == 32][0][var](a, 2, B, a.tail_cur);
  HIP_TRY(hipGetLastError());
  return MPMPC_OK;
}

extern "C" {

// Drain the stream, see whether the last reduced-native launch left a tail, run it if its launch was deferred.
static int observe_tail(mpmpc_handle h) {
  HIP_TRY(hipSetDevice(h->cfg.device));
  HIP_TRY(hipStreamSynchronize(h->stream));
  h->busy = false;
  h->order_pending = false;
  if (h->seq == 0) return MPMPC_OK;
  const bool left = *static_cast<volatile unsigned*>(h->tail_flag) == h->seq;
  h->tail_ran_late = false;
  if (h->pend) {
    h->pend = false;
    if (left) {
      h->tail_ran_late = true;
      if (int rc = launch_solve(h, h->pend_B, false, h->pend_y, 1)) return rc;
      HIP_TRY(hipStreamSynchronize(h->stream));
    }
  }
  h->tail_expect_empty = !left;
  // (second level: read after a tail that ran late has run)
  const bool left2 = static_cast<volatile unsigned*>(h->tail_flag)[1] == h->seq;
  if (h->pend2) {
    h->pend2 = false;
    if (left2) {
      h->tail_ran_late = true;
      if (int rc = launch_solve(h, h->pend_B, false, h->pend_y, 2)) return rc;
      HIP_TRY(hipStreamSynchronize(h->stream));
    }
  }
  h->tail2_expect_empty = !left2;
  return MPMPC_OK;
}

// Resident launches are DOUBLE-BUFFERED: launch k + 1 goes to the other slot (stream, output block, tail lists) and runs
// beside launch k - a batch launch fills one of the two wave slots of a SIMD (B <= 1 024) or leaves SIMDs idle while its last
// waves and its tail launch run, and the next batch takes what is idle.  The results of launch k stay where they are until
// launch k + 2; mpmpc_download / mpmpc_sync and every other call on the handle refer to the LAST launch and wait for both.
// The next launch slot of a pipelined handle becomes "the slot of the last launch".  What the current stream still has queued
// that solves read (an upload, closed-loop steps) must be done before ANY slot's next launch starts: an event goes to all
// streams.  Shared by mpmpc_solve_resident and mpmpc_solve_resident_profile (the profile loop used to swap slots
// without it, so an upload followed by it could be overtaken).  The further slots are allocated on first use: a handle that
// never launches resident batches (single calls, host-buffer streams, the closed loop) holds one output block, not three.
static int next_slot(mpmpc_handle h) {
  if (h->pipeline > 1) {
    if (h->n_alt < h->pipeline - 1) {
      if (int rc = grow_slots(h, h->pipeline - 1)) return rc;
    }
    const bool order = h->order_pending;
    if (order) HIP_TRY(hipEventRecord(h->ev_order, h->stream));
    swap_slots(h, h->ring);
    h->ring = (h->ring + 1) % (h->pipeline - 1);
    if (order) {
      HIP_TRY(hipStreamWaitEvent(h->stream, h->ev_order, 0));
      for (int i = 0; i < h->pipeline - 1; ++i) HIP_TRY(hipStreamWaitEvent(h->alt[i].stream, h->ev_order, 0));
    }
    h->order_pending = false;
  }
  h->busy = true;
  return MPMPC_OK;
}

int mpmpc_solve_resident(mpmpc_handle h, int32_t B) {
  if (!h) return fail(MPMPC_E_ARG, "handle is NULL");
  if (h->staged_bytes) {
    if (int rc = mpmpc_staged_end(h)) return rc;
  }
  if (B < 1 || B > h->uploaded) return fail(MPMPC_E_STATE, "B exceeds the uploaded batch");
  HIP_TRY(hipSetDevice(h->cfg.device));
  if (int rc = next_slot(h)) return rc;
  return launch_solve(h, B, false, h->resident_y, 0, LAUNCH_PIPELINED);      // one launch: the assembly runs inside K2
}

int mpmpc_set_pipeline(mpmpc_handle h, int32_t depth) {
  if (!h) return fail(MPMPC_E_ARG, "handle is NULL");
  MPMPC_SETTLE(h);
  if (depth < 1 || depth > mpmpc_handle_s::MAX_PIPELINE) return fail(MPMPC_E_ARG, "pipeline depth must be in [1, 8]");
  h->pipeline = depth;
  h->ring = 0;
  return MPMPC_OK;
}

int mpmpc_set_outputs(mpmpc_handle h, int32_t want_y) {
  if (!h) return fail(MPMPC_E_ARG, "handle is NULL");
  MPMPC_SETTLE(h);
  h->resident_y = want_y != 0;
  return MPMPC_OK;
}

int mpmpc_solve_resident_timed(mpmpc_handle h, int32_t B, float* ms_assemble, float* ms_solve) {
  if (!h) return fail(MPMPC_E_ARG, "handle is NULL");
  MPMPC_SETTLE(h);          // (a timed launch runs alone: nothing of an earlier launch is left on the chip)
  if (B < 1 || B > h->uploaded) return fail(MPMPC_E_STATE, "B exceeds the uploaded batch");
  HIP_TRY(hipSetDevice(h->cfg.device));
  HIP_TRY(hipEventRecord(h->ev[0], h->stream));
  if (int rc = launch_assemble(h, B)) return rc;
  HIP_TRY(hipEventRecord(h->ev[1], h->stream));
  if (int rc = launch_solve(h, B, false, h->resident_y)) return rc;
  HIP_TRY(hipEventRecord(h->ev[2], h->stream));
  HIP_TRY(hipEventSynchronize(h->ev[2]));
  float a = 0.f, s = 0.f;
  HIP_TRY(hipEventElapsedTime(&a, h->ev[0], h->ev[1]));
  HIP_TRY(hipEventElapsedTime(&s, h->ev[1], h->ev[2]));
  // a deferred tail this launch turns out to need runs now, and its time counts
  HIP_TRY(hipEventRecord(h->ev[1], h->stream));
  if (int rc = observe_tail(h)) return rc;
  if (h->tail_ran_late) {
    HIP_TRY(hipEventRecord(h->ev[2], h->stream));
    HIP_TRY(hipEventSynchronize(h->ev[2]));
    float t = 0.f;
    HIP_TRY(hipEventElapsedTime(&t, h->ev[1], h->ev[2]));
    s += t;
  }
  if (ms_assemble) *ms_assemble = a;
  if (ms_solve) *ms_solve = s;
  return MPMPC_OK;
}

// n resident launches exactly as mpmpc_solve_resident issues them (double-buffered unless mpmpc_set_pipeline(h, 1)), each
// bracketed by HIP events on the stream it goes to: ms_each[i] = duration of launch i (its kernels, with whatever else is on
// the chip beside it - the other slot's launch), *ms_span = first start to last end.  For bench.py's roofline line: rocprofv3's
// per-kernel durations of the same loop must agree with ms_each.
int mpmpc_solve_resident_profile(mpmpc_handle h, int32_t B, int32_t n, float* ms_each, float* ms_span) {
  if (!h || !ms_each) return fail(MPMPC_E_ARG, "NULL argument");
  if (n < 1 || n > 4096) return fail(MPMPC_E_ARG, "n must be in [1, 4096]");
  MPMPC_SETTLE(h);
  if (B < 1 || B > h->uploaded) return fail(MPMPC_E_STATE, "B exceeds the uploaded batch");
  HIP_TRY(hipSetDevice(h->cfg.device));
  std::vector<hipEvent_t> ev(2 * (size_t)n, nullptr);
  int rc = MPMPC_OK;
  for (auto& e : ev)
    if (hipEventCreate(&e) != hipSuccess) { rc = fail(MPMPC_E_HIP, "hipEventCreate"); break; }
  for (int i = 0; i < n && rc == MPMPC_OK; ++i) {
    if ((rc = next_slot(h)) != MPMPC_OK) break;
    if (hipEventRecord(ev[2 * i], h->stream) != hipSuccess) { rc = fail(MPMPC_E_HIP, "hipEventRecord"); break; }
    rc = launch_solve(h, B, false, h->resident_y, 0, LAUNCH_PIPELINED);
    if (rc == MPMPC_OK && hipEventRecord(ev[2 * i + 1], h->stream) != hipSuccess) rc = fail(MPMPC_E_HIP, "hipEventRecord");
  }
  if (rc == MPMPC_OK) rc = settle_other_slot(h);
  if (rc == MPMPC_OK) rc = observe_tail(h);
  if (rc == MPMPC_OK) {
    for (int i = 0; i < n; ++i)
      if (hipEventElapsedTime(&ms_each[i], ev[2 * i], ev[2 * i + 1]) != hipSuccess) { rc = fail(MPMPC_E_HIP, "hipEventElapsedTime"); break; }
    if (rc == MPMPC_OK && ms_span) {
      // (the last END is the later of the last two launches' - they run side by side)
      float last = 0.f;
      for (int i = n - 1; i >= 0 && i >= n - h->pipeline; --i) {
        float a = 0.f;
        (void)hipEventElapsedTime(&a, ev[0], ev[2 * i + 1]);
        if (a > last) last = a;
      }
      *ms_span = last;
    }
  }
  for (auto& e : ev)
    if (e) (void)hipEventDestroy(e);
  return rc;
}

int mpmpc_assemble_resident_timed(mpmpc_handle h, int32_t B, int32_t n, float* ms_each) {
  if (!h || !ms_each) return fail(MPMPC_E_ARG, "NULL argument");
  if (n < 1 || n > 4096) return fail(MPMPC_E_ARG, "n must be in [1, 4096]");
  MPMPC_SETTLE(h);
  if (B < 1 || B > h->uploaded) return fail(MPMPC_E_STATE, "B exceeds the uploaded batch");
  HIP_TRY(hipSetDevice(h->cfg.device));
  std::vector<hipEvent_t> ev((size_t)n + 1, nullptr);
  int rc = MPMPC_OK;
  for (auto& e : ev)
    if (hipEventCreate(&e) != hipSuccess) { rc = fail(MPMPC_E_HIP, "hipEventCreate"); break; }
  if (rc == MPMPC_OK && hipEventRecord(ev[0], h->stream) != hipSuccess) rc = fail(MPMPC_E_HIP, "hipEventRecord");
  for (int i = 0; i < n && rc == MPMPC_OK; ++i) {
    rc = launch_assemble(h, B);
    if (rc == MPMPC_OK && hipEventRecord(ev[i + 1], h->stream) != hipSuccess) rc = fail(MPMPC_E_HIP, "hipEventRecord");
  }
  if (rc == MPMPC_OK && hipStreamSynchronize(h->stream) != hipSuccess) rc = fail(MPMPC_E_HIP, "hipStreamSynchronize");
  for (int i = 0; i < n && rc == MPMPC_OK; ++i)
    if (hipEventElapsedTime(&ms_each[i], ev[i], ev[i + 1]) != hipSuccess) rc = fail(MPMPC_E_HIP, "hipEventElapsedTime");
  for (auto& e : ev)
    if (e) (void)hipEventDestroy(e);
  return rc;
}

// device scratch of mpmpc_speed_profile, kept between calls (a call is a set-up step, and without this its cost was
// allocation: 3.5 of 4 ms for a single path).  One scratch per device, each behind its own mutex: callers on
// different devices neither serialise nor free each other's block.
struct SpScratch {
  std::mutex mu;
  size_t bytes = 0;
  char* block = nullptr;
  hipStream_t stream = nullptr;
};
constexpr int SP_MAX_DEVICES = 64;
static SpScratch g_sp_dev[SP_MAX_DEVICES];

int mpmpc_speed_profile(int32_t device, int32_t B, int32_t n, const double* li, const double* kappa,
                        const double* limits, double eps, double* v, int32_t* status, int32_t* iters) {
  if (B < 1 || n < 2) return fail(MPMPC_E_ARG, "speed profile needs B >= 1 paths of n >= 2 segments");
  if (!li || !kappa || !limits || !v || !status) return fail(MPMPC_E_ARG, "li, kappa, limits, v, status must not be NULL");
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return fail(MPMPC_E_HIP, "no such HIP device");
  HIP_TRY(hipSetDevice(device));
  const size_t vec = sizeof(double) * (size_t)B * n;
  const size_t lds = sizeof(double) * SP_ARRAYS * (size_t)n;
  const size_t lds_wave = sizeof(double) * (SP_ARRAYS + sp_wave_arrays(n)) * (size_t)n;
  const bool wave = lds_wave <= 156 * 1024, small = !wave && B <= 256 && lds <= 64 * 1024;
  // one block: li, kappa, v, limits, status, iters (+ the HBM workspace of the thread-per-path kernel)
  const size_t o_li = 0, o_kappa = vec, o_v = 2 * vec, o_lim = 3 * vec, o_status = o_lim + pad8(sizeof(double) * 5 * B),
               o_iters = o_status + pad8(sizeof(int) * (size_t)B), o_work = o_iters + pad8(sizeof(int) * (size_t)B),
               need = o_work + ((wave || small) ? 0 : vec * SP_ARRAYS);
  if (device >= SP_MAX_DEVICES) return fail(MPMPC_E_ARG, "device ordinal beyond the speed-profile scratch table");
  SpScratch& g_sp = g_sp_dev[device];
  std::lock_guard<std::mutex> lock(g_sp.mu);
  if (g_sp.bytes < need) {
    if (g_sp.block) (void)hipFree(g_sp.block);
    g_sp.block = nullptr; g_sp.bytes = 0;
    HIP_TRY(hipMalloc((void**)&g_sp.block, need));
    g_sp.bytes = need;
  }
  if (!g_sp.stream) HIP_TRY(hipStreamCreate(&g_sp.stream));
  hipStream_t stream = g_sp.stream;
  char* blk = g_sp.block;
  double *d_li = (double*)(blk + o_li), *d_kappa = (double*)(blk + o_kappa), *d_v = (double*)(blk + o_v),
         *d_lim = (double*)(blk + o_lim), *d_work = (double*)(blk + o_work);
  int *d_status = (int*)(blk + o_status), *d_iters = (int*)(blk + o_iters);
  HIP_TRY(hipMemcpyAsync(d_li, li, vec, hipMemcpyHostToDevice, stream));
  HIP_TRY(hipMemcpyAsync(d_kappa, kappa, vec, hipMemcpyHostToDevice, stream));
  HIP_TRY(hipMemcpyAsync(d_lim, limits, sizeof(double) * 5 * B, hipMemcpyHostToDevice, stream));
  if (wave) {
    // one wavefront per path (any batch size): lane-parallel arithmetic, cyclic reduction for the tridiagonal solves
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&mpmpc_speed_profile_wave_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_wave));
    hipLaunchKernelGGL(mpmpc_speed_profile_wave_kernel, dim3(B), dim3(64), lds_wave, stream, B, n, d_li, d_kappa, d_lim,
                       eps, d_v, d_status, d_iters);
  } else if (small) {
    hipLaunchKernelGGL(mpmpc_speed_profile_lds_kernel, dim3(B), dim3(64), lds, stream, B, n, d_li, d_kappa, d_lim, eps,
                       d_v, d_status, d_iters);
  } else {
    hipLaunchKernelGGL(mpmpc_speed_profile_kernel, dim3((B + 63) / 64), dim3(64), 0, stream, B, n, d_li, d_kappa, d_lim,
                       eps, d_work, d_v, d_status, d_iters);
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(v, d_v, vec, hipMemcpyDeviceToHost, stream));
  HIP_TRY(hipMemcpyAsync(status, d_status, sizeof(int) * B, hipMemcpyDeviceToHost, stream));
  if (iters) HIP_TRY(hipMemcpyAsync(iters, d_iters, sizeof(int) * B, hipMemcpyDeviceToHost, stream));
  HIP_TRY(hipStreamSynchronize(stream));
  return MPMPC_OK;
}

int mpmpc_sync(mpmpc_handle h) {
  if (!h) return fail(MPMPC_E_ARG, "handle is NULL");
  if (h->staged_bytes) return mpmpc_staged_end(h);
  if (int rc = settle_other_slot(h)) return rc;
  return observe_tail(h);
}

int mpmpc_download(mpmpc_handle h, int32_t B, double* z, double* u0, int32_t* status, int32_t* iters,
                   double* resid, double* y) {
  if (!h) return fail(MPMPC_E_ARG, "handle is NULL");
  if (h->staged_bytes) {
    if (int rc = mpmpc_staged_end(h)) return rc;
  }
  if (int rc = settle_other_slot(h)) return rc;
  if (int rc = observe_tail(h)) return rc;        // (also how the single-call path learns that its launches leave no tail)
  if (B < 1 || B > h->uploaded) return fail(MPMPC_E_STATE, "B exceeds the uploaded batch");
  if (y && !h->y_valid)
    return fail(MPMPC_E_STATE, "the last solve launch did not store the multipliers (mpmpc_set_outputs(h, 0) / closed loop)");
  HIP_TRY(hipSetDevice(h->cfg.device));
  if (h->stage_out && B == h->laid_out) {
    // the whole batch was laid out for this B: one copy of the output block (without y, z when they are not wanted)
    const BlockLayout L = block_layout(h->cfg.N, B);
    char* st = h->stage_out;
    HIP_TRY(hipMemcpyAsync(st, h->out_block, y ? L.out_end : (z ? L.out_end_z : L.z), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (z) std::memcpy(z, st + L.z, sizeof(double) * h->n * B);
    if (u0) std::memcpy(u0, st + L.u0, sizeof(double) * 2 * B);
    if (status) std::memcpy(status, st + L.status, sizeof(int) * B);
    if (iters) std::memcpy(iters, st + L.iters, sizeof(int) * 2 * B);
    if (resid) std::memcpy(resid, st + L.resid, sizeof(double) * 2 * B);
    if (y) std::memcpy(y, st + L.y, sizeof(double) * h->m * B);
    return MPMPC_OK;
  }
  struct Pull { void* user; const char* staged; size_t bytes; } pulls[6];
  int np = 0;
  char* stage = h->stage_out;
  auto pull = [&](void* user, const void* dev, size_t bytes) -> hipError_t {
    if (!user) return hipSuccess;
    void* to = user;
    if (stage) { to = stage; pulls[np++] = Pull{user, stage, bytes}; stage += (bytes + 7) & ~size_t(7); }
    return hipMemcpyAsync(to, dev, bytes, hipMemcpyDeviceToHost, h->stream);
  };
  HIP_TRY(pull(z, h->z, sizeof(double) * h->n * B));
  HIP_TRY(pull(u0, h->u0, sizeof(double) * 2 * B));
  HIP_TRY(pull(status, h->status, sizeof(int) * B));
  HIP_TRY(pull(iters, h->iters, sizeof(int) * 2 * B));
  HIP_TRY(pull(resid, h->resid, sizeof(double) * 2 * B));
  HIP_TRY(pull(y, h->y, sizeof(double) * h->m * B));
  HIP_TRY(hipStreamSynchronize(h->stream));
  for (int i = 0; i < np; ++i) std::memcpy(pulls[i].user, pulls[i].staged, pulls[i].bytes);
  return MPMPC_OK;
}

// ---- zero-copy host path: the caller works in the handle's own page-locked staging blocks
int mpmpc_staging(mpmpc_handle h, int32_t B, int32_t** wp_id, double** x0, double** cc_prev, double** lb, double** ub,
                  double** z, double** u0, int32_t** status, int32_t** iters, double** resid, double** y) {
  if (!h) return fail(MPMPC_E_ARG, "handle is NULL");
  if (B < 1 || B > h->cfg.max_batch) return fail(MPMPC_E_ARG, "B must be in [1, max_batch]");
  if (!h->stage_in || !h->stage_out) return fail(MPMPC_E_STATE, "this handle has no staging blocks (max_batch above the staging limit)");
  // The blocks are the ones mpmpc_upload / mpmpc_download stage through: an upload still on its way out of the input block
  // must have left it before the caller may write there, and a begun staged call owns both blocks
  MPMPC_SETTLE(h);
  if (h->in_flight) { HIP_TRY(hipEventSynchronize(h->ev_in)); h->in_flight = false; }
  const BlockLayout L = block_layout(h->cfg.N, B);
  char *si = h->stage_in, *so = h->stage_out;
  if (wp_id) *wp_id = reinterpret_cast<int32_t*>(si + L.wp_id);
  if (x0) *x0 = reinterpret_cast<double*>(si + L.x0);
  if (cc_prev) *cc_prev = reinterpret_cast<double*>(si + L.cc);
  if (lb) *lb = reinterpret_cast<double*>(si + L.lb);
  if (ub) *ub = reinterpret_cast<double*>(si + L.ub);
  if (u0) *u0 = reinterpret_cast<double*>(so + L.u0);
  if (resid) *resid = reinterpret_cast<double*>(so + L.resid);
  if (status) *status = reinterpret_cast<int32_t*>(so + L.status);
  if (iters) *iters = reinterpret_cast<int32_t*>(so + L.iters);
  if (z) *z = reinterpret_cast<double*>(so + L.z);
  if (y) *y = reinterpret_cast<double*>(so + L.y);
  return MPMPC_OK;
}

int mpmpc_staged_end(mpmpc_handle h) {
  if (!h) return fail(MPMPC_E_ARG, "handle is NULL");
  if (!h->staged_bytes) return MPMPC_OK;
  const size_t out_bytes = h->staged_bytes;
  if (int rc = settle_other_slot(h)) return rc;
  if (int rc = observe_tail(h)) return rc;          // drains the stream; a launch that left a deferred tail has it run now
  h->staged_bytes = 0;                              // (only now: an error above leaves the call begun)
  if (h->tail_ran_late) {
    HIP_TRY(hipMemcpyAsync(h->stage_out, h->out_block, out_bytes, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
  }
  return MPMPC_OK;
}

int mpmpc_solve_staged(mpmpc_handle h, int32_t B, int32_t with_rows, int32_t want_z, int32_t want_y) {
  if (!h) return fail(MPMPC_E_ARG, "handle is NULL");
  h->staged_async = false;          // one call, waited for at once: the packing of a single launch
  const int rc = mpmpc_staged_begin(h, B, with_rows, want_z, want_y);
  h->staged_async = true;
  if (rc) return rc;
  return mpmpc_staged_end(h);
}

int mpmpc_staged_begin(mpmpc_handle h, int32_t B, int32_t with_rows, int32_t want_z, int32_t want_y) {
  if (!h) return fail(MPMPC_E_ARG, "handle is NULL");
  if (h->staged_bytes) return fail(MPMPC_E_STATE, "mpmpc_staged_begin: the previous one has not been ended");
  MPMPC_SETTLE(h);
  if (B < 1 || B > h->cfg.max_batch) return fail(MPMPC_E_ARG, "B must be in [1, max_batch]");
  if (!h->stage_in || !h->stage_out) return fail(MPMPC_E_STATE, "this handle has no staging blocks (max_batch above the staging limit)");
  if (h->n_wp == 0) return fail(MPMPC_E_STATE, "no path set (mpmpc_set_path)");
  if (!with_rows && h->n_cols == 0) return fail(MPMPC_E_STATE, "no corridor rows given and no corridor table set");
  const int N = h->cfg.N;
  const BlockLayout L = block_layout(N, B);
  const int32_t* wp = reinterpret_cast<const int32_t*>(h->stage_in + L.wp_id);
  for (int i = 0; i < B; ++i) {
    if (wp[i] < 0 || wp[i] >= h->n_wp) return fail(MPMPC_E_ARG, "wp_id out of range");
    if (!h->cfg.circular && wp[i] + N >= h->n_wp) return fail(MPMPC_E_ARG, "Reached end of path!");
  }
  HIP_TRY(hipSetDevice(h->cfg.device));
  if (h->in_flight) { HIP_TRY(hipEventSynchronize(h->ev_in)); h->in_flight = false; }
  h->ro_valid = false;
  if (B != h->laid_out) lay_out(h, B);
  HIP_TRY(hipMemcpyAsync(h->in_block, h->stage_in, with_rows ? L.in_end : L.in_end_cc, hipMemcpyHostToDevice, h->stream));
  h->have_rows = with_rows != 0;
  h->uploaded = B;
  if (int rc = launch_solve(h, B, false, want_y != 0, 0, h->staged_async ? LAUNCH_PIPELINED : LAUNCH_SINGLE)) return rc;
  const size_t out_bytes = want_y ? L.out_end : (want_z ? L.out_end_z : L.z);
  HIP_TRY(hipMemcpyAsync(h->stage_out, h->out_block, out_bytes, hipMemcpyDeviceToHost, h->stream));
  h->staged_bytes = out_bytes;
  return MPMPC_OK;
}

int mpmpc_assemble(mpmpc_handle h, int32_t B, const int32_t* wp_id, const double* x0, const double* cc_prev,
                   const double* lb, const double* ub, double* qp_out) {
  if (int rc = mpmpc_upload(h, B, wp_id, x0, cc_prev, lb, ub)) return rc;
  if (int rc = launch_assemble(h, B)) return rc;
  if (qp_out) {
    // device layout is [field][B][ld]; rows of B*ld doubles per field are contiguous
    HIP_TRY(hipMemcpyAsync(qp_out, h->qp, sizeof(double) * MPMPC_NUM_FIELDS * (size_t)B * h->ld,
                           hipMemcpyDeviceToHost, h->stream));
  }
  HIP_TRY(hipStreamSynchronize(h->stream));
  return MPMPC_OK;
}

int mpmpc_solve(mpmpc_handle h, int32_t B, const int32_t* wp_id, const double* x0, const double* cc_prev,
                const double* lb, const double* ub, double* z, double* u0, int32_t* status, int32_t* iters,
                double* resid, double* y) {
  if (int rc = mpmpc_upload(h, B, wp_id, x0, cc_prev, lb, ub)) return rc;
  HIP_TRY(hipSetDevice(h->cfg.device));
  if (int rc = launch_solve(h, B, false, y != nullptr)) return rc;      // no y asked for: none stored
  return mpmpc_download(h, B, z, u0, status, iters, resid, y);
}

}  // extern "C"

#ifdef MPMPC_PHASE_CLOCK
extern "C" int mpmpc_debug_phase(long long* out, int reset) {
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase), sizeof(long long) * 4096 * 32) != hipSuccess) return -1;
  if (reset) {
    static long long zeros[4096 * 32];
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_phase), zeros, sizeof(zeros)) != hipSuccess) return -1;
  }
  return 0;
}
#endif
