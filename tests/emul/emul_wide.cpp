// CPU lock-step emulation of the general solve kernel for horizons above 63 (TEST INFRASTRUCTURE ONLY).
// On the device such an instance takes a WORKGROUP of 2 / 4 wavefronts - one lane per stage, neighbours across the wavefront
// boundaries through LDS (lane_gpu.hpp: LaneBlock<128> / LaneBlock<256>, mpmpc_solve_block_kernel); here the same lane code
// (mpmpc_core.hpp: Solver) runs on an emulated execution group of MPMPC_EMU_W = 128 / 256 lanes.  Built twice
// (tests/emul/Makefile: libmpmpc_emul_w128.so, libmpmpc_emul_w256.so).  Never loaded by the product.
#include <cstring>
#include <vector>
#define MPMPC_TICK_BEGIN(i) ((void)0)
#define MPMPC_TICK_END(i) ((void)0)
#define MPMPC_TICK_COUNT(i) ((void)0)
#include "lane_emu.hpp"
#include "lane_pair.hpp"
#include "mpmpc_core.hpp"
#include "mpmpc_reduced.hpp"
#include "mpmpc_reduced_t.hpp"
#include "mpmpc_reduced_tail.hpp"

using namespace mpmpc;
static_assert(EMU_W == 128 || EMU_W == 256, "build with -DMPMPC_EMU_W=128 or 256");

// (a chain of the 128-lane workgroup is a wavefront: the reduced variant factors it by cyclic reduction, as on the device)
static_assert(EMU_W != 128 || Solver<LaneEmu<EMU_W, EMU_W / 2>, false, true, false, true>::kCR64, "cyclic reduction of 64-lane chains");
static_assert(EMU_W != 256 || Solver<LaneEmu<EMU_W, EMU_W / 2>, false, true, false, true>::kCRrows == 8, "256 lanes: chains of eight rows");

template <int VAR>
static void solve_wide(const mpmpc_config* cfg, const mpmpc_settings* st, const double* qp, int B, double* z, double* u0,
                       int* status, int* iters, double* resid, double* y) {
  constexpr int G = EMU_W, C = EMU_W / 2;
  using L = LaneEmu<G, C>;
  const int ld = stage_ld(cfg->N);
  for (int w = 0; w < B; ++w) {
    VI inst = L::slot() + w;
    VI k = L::stage() - lane_offset(G, C, cfg->N);
    using S = Solver<L, VAR == 1, VAR == 2, false, VAR == 2>;      // (as mpmpc_solve_block_kernel: cyclic reduction where a chain is a wavefront)
    S s;
    double woff7[7];
    weight_offdiag(*cfg, woff7);
    typename L::real fields[MPMPC_NUM_FIELDS];
    S::fetch_fields(qp, B, ld, inst, k, cfg->N, fields);
    s.template run<false, true>(fields, B, inst, k, cfg->N, make_params(*st), 0, VI(0), VI(0), VAR == 1 ? woff7 : nullptr);
    s.store(inst, k, cfg->wheelbase, z, u0, status, iters, resid, y, nullptr, ld);
  }
}

// the launcher's sequence for the reference's own weights at the default settings (reduced_native): the reduced-native solver
// on the workgroup first (mpmpc_reduced_block_kernel), then the general one - straight to phase 1 and the full iteration, mode 2 -
// on what that could not certify
static void solve_wide_native(const mpmpc_config* cfg, const mpmpc_settings* st, const double* qp, int B, double* z, double* u0,
                              int* status, int* iters, double* resid, double* y) {
  constexpr int G = EMU_W, C = EMU_W / 2;
  using L = LaneEmu<G, C>;
  const int ld = stage_ld(cfg->N);
  std::vector<int> tail;
  for (int w = 0; w < B; ++w) {
    VI inst = L::slot() + w;
    VI k = L::stage() - lane_offset(G, C, cfg->N);
    ReducedSolver<L> s;
    typename L::real fields[MPMPC_NUM_FIELDS];
    ReducedSolver<L>::fetch_fields(qp, B, ld, inst, k, cfg->N, fields);
    s.template run<false>(fields, B, inst, k, cfg->N, make_params(*st), VI(0));
    s.store(inst, k, cfg->wheelbase, z, u0, status, iters, resid, y, nullptr, ld);
    for (int i = 0; i < EMU_W; ++i)
      if (k.v[i] == 0 && inst.v[i] < B && s.status.v[i] == MPMPC_UNSOLVED) tail.push_back(inst.v[i]);
  }
  for (int id : tail) {
    VI inst = L::slot() + id;
    VI k = L::stage() - lane_offset(G, C, cfg->N);
    using S = Solver<L, false, true, false, true>;
    S s;
    typename L::real fields[MPMPC_NUM_FIELDS];
    S::fetch_fields(qp, B, ld, inst, k, cfg->N, fields);
    s.template run<false, true>(fields, B, inst, k, cfg->N, make_params(*st), 2, VI(0), VI(iters[id * 2 + 1]), nullptr);
    s.store(inst, k, cfg->wheelbase, z, u0, status, iters, resid, y, nullptr, ld);
  }
}

// the tail alone: mpmpc_solve_block_kernel<G, VAR> on the listed instances (mode 2: phase 1 and the full iteration) - what follows
// the reduced-native kernels with two stages per lane (horizons 64 .. 127 in one wavefront: emul.cpp, solve_rn2<64> / solve_rnt2<64>)
template <int VAR>
static void solve_wide_tail(const mpmpc_config* cfg, const mpmpc_settings* st, const double* qp, int B, double* z, double* u0,
                            int* status, int* iters, double* resid, double* y, const int* ids, int n_ids) {
  constexpr int G = EMU_W, C = EMU_W / 2;
  using L = LaneEmu<G, C>;
  const int ld = stage_ld(cfg->N);
  for (int j = 0; j < n_ids; ++j) {
    const int id = ids[j];
    VI inst = L::slot() + id;
    VI k = L::stage() - lane_offset(G, C, cfg->N);
    using S = Solver<L, false, VAR == 2, false, VAR == 2>;
    S s;
    typename L::real fields[MPMPC_NUM_FIELDS];
    S::fetch_fields(qp, B, ld, inst, k, cfg->N, fields);
    s.template run<false, true>(fields, B, inst, k, cfg->N, make_params(*st), 2, VI(0), VI(iters[id * 2 + 1]), nullptr);
    s.store(inst, k, cfg->wheelbase, z, u0, status, iters, resid, y, nullptr, ld);
  }
}
extern "C" int emuw_solve_tail(const mpmpc_config* cfg, const mpmpc_settings* st, const double* qp, int B, double* z, double* u0,
                               int* status, int* iters, double* resid, double* y, const int* ids, int n_ids) {
  if (stage_ld(cfg->N) != EMU_W || full_weights(*cfg)) return -1;
  if (reducible(*cfg, *st)) solve_wide_tail<2>(cfg, st, qp, B, z, u0, status, iters, resid, y, ids, n_ids);
  else solve_wide_tail<0>(cfg, st, qp, B, z, u0, status, iters, resid, y, ids, n_ids);      // (the terminal-time kernel's tail: the full problem)
  return 0;
}

// Horizons 128 .. 255 with TWO stages per lane: the instance on an emulated workgroup of 128 lanes - one chain of eight rows over
// two wavefronts (mpmpc_reduced_pair_block_kernel; the 128-lane build only).  ids <- the instances it leaves UNSOLVED (*n of them):
// they go to the general solver on 256 lanes (the 256-lane build's emuw_solve_tail).
extern "C" int emuw_solve_rn_pair(const mpmpc_config* cfg, const mpmpc_settings* st, const double* qp, int B, double* z, double* u0,
                                  int* status, int* iters, double* resid, double* y, int* ids, int* n) {
#if MPMPC_EMU_W != 128
  (void)cfg; (void)st; (void)qp; (void)B; (void)z; (void)u0; (void)status; (void)iters; (void)resid; (void)y; (void)ids; (void)n;
  return -1;
#else
  {
    if (stage_ld(cfg->N) != 256 || full_weights(*cfg) || !reduced_native(*cfg, *st)) return -1;
    using L = LanePair<LaneEmu<128, 128, 74>>;          // (37 pair slots: the lean cold storage of the device kernel, ReducedSolver::kLean)
    static_assert(ReducedSolver<L>::kLean, "lean cold storage");
    const int ld = stage_ld(cfg->N);
    *n = 0;
    for (int w = 0; w < B; ++w) {
      const I2 inst = L::slot() + w;
      const I2 k = L::stage();
      ReducedSolver<L> s;
      typename L::real fields[MPMPC_NUM_FIELDS];
      ReducedSolver<L>::fetch_fields(qp, B, ld, inst, k, cfg->N, fields);
      s.template run<false>(fields, B, inst, k, cfg->N, make_params(*st));
      s.store(inst, k, cfg->wheelbase, z, u0, status, iters, resid, y, nullptr, ld);
      if (s.status.v[0].v[0] == MPMPC_UNSOLVED) ids[(*n)++] = w;
    }
    return 0;
  }
#endif
}

// ... its twin for a terminal cost on the time state (mpmpc_reduced_t_pair_block_kernel; full cold storage: one workgroup per CU)
extern "C" int emuw_solve_rnt_pair(const mpmpc_config* cfg, const mpmpc_settings* st, const double* qp, int B, double* z, double* u0,
                                   int* status, int* iters, double* resid, double* y, int* ids, int* n) {
#if MPMPC_EMU_W != 128
  (void)cfg; (void)st; (void)qp; (void)B; (void)z; (void)u0; (void)status; (void)iters; (void)resid; (void)y; (void)ids; (void)n;
  return -1;
#else
  if (stage_ld(cfg->N) != 256 || full_weights(*cfg) || !reduced_native_tt(*cfg, *st)) return -1;
  using L = LanePair<LaneEmu<128, 128>>;
  const int ld = stage_ld(cfg->N);
  *n = 0;
  for (int w = 0; w < B; ++w) {
    const I2 inst = L::slot() + w;
    const I2 k = L::stage();
    ReducedTSolver<L> s;
    typename L::real fields[MPMPC_NUM_FIELDS];
    ReducedTSolver<L>::fetch_fields(qp, B, ld, inst, k, cfg->N, fields);
    s.run(fields, B, inst, k, cfg->N, make_params(*st), cfg->QN[2]);
    s.store(inst, k, cfg->wheelbase, z, u0, status, iters, resid, y);
    if (s.status.v[0].v[0] == MPMPC_UNSOLVED) ids[(*n)++] = w;
  }
  return 0;
#endif
}
// ... and the reduced-native TAIL solver on the same workgroup, on a list of instances (mpmpc_reduced_tail_pair_block_kernel);
// ids2 <- what it leaves
extern "C" int emuw_solve_rn_tail_pair(const mpmpc_config* cfg, const mpmpc_settings* st, const double* qp, int B, double* z, double* u0,
                                       int* status, int* iters, double* resid, double* y, const int* ids, int n_ids, int* ids2, int* n2) {
#if MPMPC_EMU_W != 128
  (void)cfg; (void)st; (void)qp; (void)B; (void)z; (void)u0; (void)status; (void)iters; (void)resid; (void)y; (void)ids; (void)n_ids; (void)ids2; (void)n2;
  return -1;
#else
  if (stage_ld(cfg->N) != 256 || !reduced_native_tail(*cfg, *st)) return -1;
  using L = LanePair<LaneEmu<128, 128>>;
  const int ld = stage_ld(cfg->N);
  *n2 = 0;
  for (int j = 0; j < n_ids; ++j) {
    const int id = ids[j];
    const I2 inst = I2(id), base = I2(iters[id * 2 + 1]);
    const I2 k = L::stage();
    ReducedTailSolver<L> s;
    typename L::real fields[MPMPC_NUM_FIELDS];
    ReducedTailSolver<L>::fetch_fields(qp, B, ld, inst, k, cfg->N, fields);
    s.run(fields, B, inst, k, cfg->N, make_params(*st), base);
    s.store(inst, k, cfg->wheelbase, z, u0, status, iters, resid, y);
    if (s.status.v[0].v[0] == MPMPC_UNSOLVED) ids2[(*n2)++] = id;
  }
  return 0;
#endif
}

// the kernel the launcher picks for a horizon above 63: the general solver, one instance per workgroup; full weights
// where a weight matrix has off-diagonal entries, the reduced polish where the time state separates.  -1: the horizon does not belong to this width.
extern "C" int emuw_width() { return EMU_W; }
extern "C" int emuw_solve(const mpmpc_config* cfg, const mpmpc_settings* st, const double* qp, int B, double* z, double* u0,
                          int* status, int* iters, double* resid, double* y) {
  if (stage_ld(cfg->N) != EMU_W) return -1;
  if (!full_weights(*cfg) && reduced_native(*cfg, *st)) { solve_wide_native(cfg, st, qp, B, z, u0, status, iters, resid, y); return 0; }
  if (full_weights(*cfg)) solve_wide<1>(cfg, st, qp, B, z, u0, status, iters, resid, y);
  else if (reducible(*cfg, *st)) solve_wide<2>(cfg, st, qp, B, z, u0, status, iters, resid, y);      // (the launcher's choice: mpmpc_hip.hip, launch_solve)
  else solve_wide<0>(cfg, st, qp, B, z, u0, status, iters, resid, y);
  return 0;
}
