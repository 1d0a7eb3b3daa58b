// CPU lock-step emulation of the HIP kernels (TEST INFRASTRUCTURE ONLY).
// Instantiates the lane-generic core of multi-purpose-mpc_amd/csrc/mpmpc_core.hpp with the
// 64-lane emulated wavefront of lane_emu.hpp so the kernel algorithm can be checked against
// the oracle in the GPU-less authoring container.  Never loaded by the product.
#include <cstring>
#include <initializer_list>
// event counters of the solver code (the phase-clock build of the library counts the same events per wave):
// 16 interior-point iterations, 17 active-set rounds, 18 active-set KKT solves - per emulated wave
static long long g_emu_count[32];
#define MPMPC_TICK_BEGIN(i) ((void)0)
#define MPMPC_TICK_END(i) ((void)0)
#define MPMPC_TICK_COUNT(i) (++g_emu_count[i])
#include "lane_emu.hpp"
#include "lane_pair.hpp"
#include "mpmpc_core.hpp"
#include "mpmpc_reduced.hpp"
#include "mpmpc_reduced_t.hpp"
#include "mpmpc_reduced_tail.hpp"
#include "corridor_core.hpp"
#include "rollout_core.hpp"
#include <limits>
#include <vector>

using namespace mpmpc;

// mode / tail as in mpmpc_solve_kernel: mode 1 appends the instances it leaves UNSOLVED to tail[1..] (tail[0] counts),
// mode 2 runs one wave per listed instance
template <int G, int C, bool FQ = false, bool RED = false, bool FREEX = false>
static void solve_g(const mpmpc_config* cfg, const mpmpc_settings* st, const double* qp, int B, double* z,
                    double* u0, int* status, int* iters, double* resid, double* y, const int* guess = nullptr,
                    int* act = nullptr, int mode = 0, int* tail = nullptr) {
  using L = LaneEmu<G, C>;
  const int ld = stage_ld(cfg->N);
  const int per = L::per_wave;
  const int waves = mode == 2 ? tail[0] : (B + per - 1) / per;
  for (int w = 0; w < waves; ++w) {
    VI inst = L::slot() + w * per;
    if (mode == 2) inst = VI(tail[1 + w]);
    VI k = L::stage() - lane_offset(G, C, cfg->N);
    VI gs, base(0);
    for (int i = 0; i < EMU_W; ++i) {
      const int in = inst.v[i], kk = k.v[i];
      gs.v[i] = (guess && in < B && kk >= 0 && kk <= cfg->N) ? guess[in * ld + kk] : 0;
      if (mode == 2) base.v[i] = iters[in * 2 + 1];
    }
    Solver<L, FQ, RED, FREEX, RED> s;
    double woff7[7];
    weight_offdiag(*cfg, woff7);
    const double* woff = FQ ? woff7 : nullptr;
    typename L::real fields[MPMPC_NUM_FIELDS];
    Solver<L, FQ, RED, FREEX, RED>::fetch_fields(qp, B, ld, inst, k, cfg->N, fields);
    // (like the device: the packed kernels carry no phase-1 code when they run as the first of two launches)
    if (guess) s.template run<true>(fields, B, inst, k, cfg->N, make_params(*st), mode, gs, base, woff);
    else if (mode == 1) s.template run<false, (G == 64)>(fields, B, inst, k, cfg->N, make_params(*st), mode, VI(0), base, woff);
    else s.template run<false, true>(fields, B, inst, k, cfg->N, make_params(*st), mode, VI(0), base, woff);
    s.store(inst, k, cfg->wheelbase, z, u0, status, iters, resid, y, act, ld);
    if (mode == 1)
      for (int i = 0; i < EMU_W; ++i)
        if (k.v[i] == 0 && inst.v[i] < B && s.status.v[i] == MPMPC_UNSOLVED) tail[1 + tail[0]++] = inst.v[i];
  }
}

// mpmpc_reduced_kernel: the reduced-native solver; instances it leaves UNSOLVED are appended to tail[1..]
template <int G, int C, bool CR = true>
static void solve_rn(const mpmpc_config* cfg, const mpmpc_settings* st, const double* qp, int B, double* z, double* u0,
                     int* status, int* iters, double* resid, double* y, int* tail, const int* guess = nullptr, int* act = nullptr) {
  using L = LaneEmu<G, C>;
  const int ld = stage_ld(cfg->N);
  const int per = L::per_wave;
  for (int w = 0; w < (B + per - 1) / per; ++w) {
    VI inst = L::slot() + w * per;
    VI k = L::stage() - lane_offset(G, C, cfg->N);
    VI gs(0);
    for (int i = 0; i < EMU_W; ++i) {
      const int in = inst.v[i], kk = k.v[i];
      gs.v[i] = (guess && in < B && kk >= 0 && kk <= cfg->N) ? guess[in * ld + kk] : 0;
    }
    ReducedSolver<L, CR> s;
    typename L::real fields[MPMPC_NUM_FIELDS];
    ReducedSolver<L, CR>::fetch_fields(qp, B, ld, inst, k, cfg->N, fields);
    if (guess) s.template run<true>(fields, B, inst, k, cfg->N, make_params(*st), gs);
    else s.template run<false>(fields, B, inst, k, cfg->N, make_params(*st));
    s.store(inst, k, cfg->wheelbase, z, u0, status, iters, resid, y, act, ld);
    for (int i = 0; i < EMU_W; ++i)
      if (k.v[i] == 0 && inst.v[i] < B && s.status.v[i] == MPMPC_UNSOLVED) tail[1 + tail[0]++] = inst.v[i];
  }
}
// mpmpc_reduced_pair_kernel: the same solver with TWO stages per lane (lane_pair.hpp) - 16 lanes per instance for N + 1 <= 32,
// four instances per emulated wave
// (GB = 64: horizons 64 .. 127 in ONE emulated wavefront - on the one-stage layout they take a workgroup of two, emul_wide.cpp)
template <int GB>
static void solve_rn2(const mpmpc_config* cfg, const mpmpc_settings* st, const double* qp, int B, double* z, double* u0,
                      int* status, int* iters, double* resid, double* y, int* tail) {
  using L = LanePair<LaneEmu<GB, GB>>;
  const int ld = stage_ld(cfg->N);
  const int per = L::per_wave;
  for (int w = 0; w < (B + per - 1) / per; ++w) {
    const I2 inst = L::slot() + w * per;
    const I2 k = L::stage();
    ReducedSolver<L> s;
    typename L::real fields[MPMPC_NUM_FIELDS];
    ReducedSolver<L>::fetch_fields(qp, B, ld, inst, k, cfg->N, fields);
    s.template run<false>(fields, B, inst, k, cfg->N, make_params(*st));
    s.store(inst, k, cfg->wheelbase, z, u0, status, iters, resid, y, nullptr, ld);
    for (int i = 0; i < EMU_W; ++i)
      if (k.v[0].v[i] == 0 && inst.v[0].v[i] < B && s.status.v[0].v[i] == MPMPC_UNSOLVED) tail[1 + tail[0]++] = inst.v[0].v[i];
  }
}
// mpmpc_reduced_t_kernel: the reduced-native solver of the weightings with a terminal cost on the time state
template <int G, int C, bool CR = true>
static void solve_rnt(const mpmpc_config* cfg, const mpmpc_settings* st, const double* qp, int B, double* z, double* u0,
                      int* status, int* iters, double* resid, double* y, int* tail) {
  using L = LaneEmu<G, C>;
  const int ld = stage_ld(cfg->N);
  const int per = L::per_wave;
  for (int w = 0; w < (B + per - 1) / per; ++w) {
    VI inst = L::slot() + w * per;
    VI k = L::stage() - lane_offset(G, C, cfg->N);
    ReducedTSolver<L, CR> s;
    typename L::real fields[MPMPC_NUM_FIELDS];
    ReducedTSolver<L, CR>::fetch_fields(qp, B, ld, inst, k, cfg->N, fields);
    s.run(fields, B, inst, k, cfg->N, make_params(*st), cfg->QN[2]);
    s.store(inst, k, cfg->wheelbase, z, u0, status, iters, resid, y);
    for (int i = 0; i < EMU_W; ++i)
      if (k.v[i] == 0 && inst.v[i] < B && s.status.v[i] == MPMPC_UNSOLVED) tail[1 + tail[0]++] = inst.v[i];
  }
}
// ... with two stages per lane: horizons 64 .. 127 in one emulated wavefront (mpmpc_reduced_t_pair_kernel<64>)
template <int GB>
static void solve_rnt2(const mpmpc_config* cfg, const mpmpc_settings* st, const double* qp, int B, double* z, double* u0,
                       int* status, int* iters, double* resid, double* y, int* tail) {
  using L = LanePair<LaneEmu<GB, GB>>;
  const int ld = stage_ld(cfg->N);
  for (int w = 0; w < B; ++w) {
    const I2 inst = L::slot() + w;
    const I2 k = L::stage();
    ReducedTSolver<L> s;
    typename L::real fields[MPMPC_NUM_FIELDS];
    ReducedTSolver<L>::fetch_fields(qp, B, ld, inst, k, cfg->N, fields);
    s.run(fields, B, inst, k, cfg->N, make_params(*st), cfg->QN[2]);
    s.store(inst, k, cfg->wheelbase, z, u0, status, iters, resid, y);
    for (int i = 0; i < EMU_W; ++i)
      if (k.v[0].v[i] == 0 && inst.v[0].v[i] < B && s.status.v[0].v[i] == MPMPC_UNSOLVED) tail[1 + tail[0]++] = inst.v[0].v[i];
  }
}
// mpmpc_reduced_tail_kernel: the reduced-native tail solver on the instances listed in tail; what it leaves UNSOLVED is
// appended to tail2[1..]
static int g_emu_lean_tail = 1;          // emu_set_lean_tail, like mpmpc_set_tail_kernel: 0 = the general kernel takes the whole
                                         // tail (as before round 4); 1 = the tail solver, TWO instances per wave (32 lanes each,
                                         // three entries per lane: the device's default); 2 = the tail solver, one instance per wave
template <int G, int C>
static void solve_rn_tail(const mpmpc_config* cfg, const mpmpc_settings* st, const double* qp, int B, double* z, double* u0,
                          int* status, int* iters, double* resid, double* y, const int* tail, int* tail2) {
  using L = LaneEmu<G, C>;
  const int ld = stage_ld(cfg->N);
  const int per = L::per_wave;
  for (int w = 0; w < (tail[0] + per - 1) / per; ++w) {
    // (the instances of a wave: consecutive entries of the list; a wave of the last, partly filled group carries B = "none")
    VI slot = L::slot(), inst, base(0);
    for (int i = 0; i < EMU_W; ++i) {
      const int e = w * per + slot.v[i];
      inst.v[i] = e < tail[0] ? tail[1 + e] : B;
      base.v[i] = inst.v[i] < B ? iters[inst.v[i] * 2 + 1] : 0;
    }
    VI k = L::stage() - lane_offset(G, C, cfg->N);
    ReducedTailSolver<L> s;
    typename L::real fields[MPMPC_NUM_FIELDS];
    ReducedTailSolver<L>::fetch_fields(qp, B, ld, inst, k, cfg->N, fields);
    s.run(fields, B, inst, k, cfg->N, make_params(*st), base);
    s.store(inst, k, cfg->wheelbase, z, u0, status, iters, resid, y);
    for (int i = 0; i < EMU_W; ++i)
      if (k.v[i] == 0 && inst.v[i] < B && s.status.v[i] == MPMPC_UNSOLVED) tail2[1 + tail2[0]++] = inst.v[i];
  }
}
// ... with two stages per lane (mpmpc_reduced_tail_pair_kernel<64>: horizons 64 .. 127, one instance per emulated wavefront)
template <int GB>
static void solve_rn_tail2(const mpmpc_config* cfg, const mpmpc_settings* st, const double* qp, int B, double* z, double* u0,
                           int* status, int* iters, double* resid, double* y, const int* tail, int* tail2) {
  using L = LanePair<LaneEmu<GB, GB>>;
  static_assert(L::per_wave == 1, "one instance per wavefront");
  const int ld = stage_ld(cfg->N);
  for (int w = 0; w < tail[0]; ++w) {
    const int id = tail[1 + w];
    const I2 inst = I2(id), base = I2(iters[id * 2 + 1]);
    const I2 k = L::stage();
    ReducedTailSolver<L> s;
    typename L::real fields[MPMPC_NUM_FIELDS];
    ReducedTailSolver<L>::fetch_fields(qp, B, ld, inst, k, cfg->N, fields);
    s.run(fields, B, inst, k, cfg->N, make_params(*st), base);
    s.store(inst, k, cfg->wheelbase, z, u0, status, iters, resid, y);
    for (int i = 0; i < EMU_W; ++i)
      if (k.v[0].v[i] == 0 && s.status.v[0].v[i] == MPMPC_UNSOLVED) tail2[1 + tail2[0]++] = id;
  }
}
// the reduced-native tail solver (pair layout) on a list of instances; ids2 <- what it leaves (n2 of them)
extern "C" int emu_solve_rn_tail_pair(const mpmpc_config* cfg, const mpmpc_settings* st, const double* qp, int B, double* z, double* u0,
                                      int* status, int* iters, double* resid, double* y, const int* ids, int n_ids, int* ids2, int* n2) {
  if (cfg->N + 1 <= 64 || cfg->N + 1 > 128 || !reduced_native_tail(*cfg, *st)) return -1;
  std::vector<int> tail(n_ids + 1), tail2(n_ids + 1, 0);
  tail[0] = n_ids;
  for (int i = 0; i < n_ids; ++i) tail[1 + i] = ids[i];
  solve_rn_tail2<64>(cfg, st, qp, B, z, u0, status, iters, resid, y, tail.data(), tail2.data());
  *n2 = tail2[0];
  for (int i = 0; i < tail2[0]; ++i) ids2[i] = tail2[1 + i];
  return 0;
}
extern "C" void emu_set_lean_tail(int on) { g_emu_lean_tail = on; }
extern "C" int emu_reduced_native_tail(const mpmpc_config* cfg, const mpmpc_settings* st) { return reduced_native_tail(*cfg, *st) ? 1 : 0; }
static int g_emu_tail2 = 0;              // instances the last emu_solve_launch's reduced-native tail solver left to the general kernel
extern "C" int emu_last_tail2() { return g_emu_tail2; }

static int solve_rnt_g(int G, const mpmpc_config* cfg, const mpmpc_settings* st, const double* qp, int B, double* z, double* u0,
                       int* status, int* iters, double* resid, double* y, int* tail) {
  const int C = lane_split(64, cfg->N);
  // (one instance per wave whatever packing the caller asked for: the launcher does the same)
  (void)G;
  if (cfg->N + 1 > 64) { if (cfg->N + 1 > 128) return -1; solve_rnt2<64>(cfg, st, qp, B, z, u0, status, iters, resid, y, tail); return 0; }
  if (C == 16 && lane_split(64, cfg->N) == 16) solve_rnt<64, 16>(cfg, st, qp, B, z, u0, status, iters, resid, y, tail);
  else solve_rnt<64, 32>(cfg, st, qp, B, z, u0, status, iters, resid, y, tail);
  return 0;
}
static int solve_rn_g(int G, const mpmpc_config* cfg, const mpmpc_settings* st, const double* qp, int B, double* z, double* u0,
                      int* status, int* iters, double* resid, double* y, int* tail, const int* guess = nullptr, int* act = nullptr) {
  const int C = lane_split(G, cfg->N);
  // (16 lanes for more than 16 stages: two stages per lane - cold starts only, like the launcher)
  if (G == 16 && cfg->N + 1 > 16) { if (guess || cfg->N + 1 > 32) return -1; solve_rn2<16>(cfg, st, qp, B, z, u0, status, iters, resid, y, tail); return 0; }
  if (G == 64 && cfg->N + 1 > 64) { if (guess || cfg->N + 1 > 128) return -1; solve_rn2<64>(cfg, st, qp, B, z, u0, status, iters, resid, y, tail); return 0; }
  if (G == 64 && C == 16) solve_rn<64, 16>(cfg, st, qp, B, z, u0, status, iters, resid, y, tail, guess, act);
  else if (G == 64) solve_rn<64, 32>(cfg, st, qp, B, z, u0, status, iters, resid, y, tail, guess, act);
  else if (G == 32) solve_rn<32, 16>(cfg, st, qp, B, z, u0, status, iters, resid, y, tail, guess, act);
  else if (G == 16) solve_rn<16, 16>(cfg, st, qp, B, z, u0, status, iters, resid, y, tail, guess, act);
  else return -1;
  return 0;
}

// the variant the launcher would pick: reduced polish where the configuration allows it
#define SOLVE_G(GG, CC, ...)                                                          \
  do {                                                                                \
    if (reducible(*cfg, *st)) solve_g<GG, CC, false, true>(__VA_ARGS__);              \
    else if (GG == 64 && free_states(*cfg)) solve_g<GG, CC, false, false, (GG == 64)>(__VA_ARGS__);   \
    else solve_g<GG, CC>(__VA_ARGS__);                                                \
  } while (0)

extern "C" int emu_solve(const mpmpc_config* cfg, const mpmpc_settings* st, int G, const double* qp, int B,
                         double* z, double* u0, int* status, int* iters, double* resid, double* y) {
  if (cfg->N + 1 > G) return -1;
  const int C = lane_split(G, cfg->N);       // same variant as the launcher picks
  const bool fullqn = full_weights(*cfg);          // Q, R or QN with off-diagonal entries
  if (fullqn && G != 64) return -1;          // the launcher gives such instances a wave each
  if (fullqn && C == 16) solve_g<64, 16, true>(cfg, st, qp, B, z, u0, status, iters, resid, y);
  else if (fullqn) solve_g<64, 32, true>(cfg, st, qp, B, z, u0, status, iters, resid, y);
  else if (G == 64 && C == 16) SOLVE_G(64, 16, cfg, st, qp, B, z, u0, status, iters, resid, y);
  else if (G == 64) SOLVE_G(64, 32, cfg, st, qp, B, z, u0, status, iters, resid, y);
  else if (G == 32) SOLVE_G(32, 16, cfg, st, qp, B, z, u0, status, iters, resid, y);
  else if (G == 16) SOLVE_G(16, 16, cfg, st, qp, B, z, u0, status, iters, resid, y);
  else return -1;
  return 0;
}

// what launch_solve does with a packed batch (G < 64 and an early polish attempt): the packed kernel in mode 1, then
// the <64, C> kernel in mode 2 on the instances it left unsolved (phase 1, full ADMM run); otherwise one launch.
extern "C" int emu_solve_launch(const mpmpc_config* cfg, const mpmpc_settings* st, int G, const double* qp, int B,
                                double* z, double* u0, int* status, int* iters, double* resid, double* y, int* n_tail) {
  if (cfg->N + 1 > G && !(G == 16 && cfg->N + 1 <= 32 && reduced_native(*cfg, *st))) return -1;      // (two stages per lane: solve_rn2)
  const bool early = st->polish && st->early_polish > 0 && st->early_polish < st->max_iter;
  if (n_tail) *n_tail = 0;
  std::vector<int> tail(B + 1, 0);
  if (reduced_native(*cfg, *st) || reduced_native_tt(*cfg, *st)) {
    // the reduced-native kernel for the whole batch (any packing), then the general kernel on its tail
    if (reduced_native_tt(*cfg, *st) ? solve_rnt_g(G, cfg, st, qp, B, z, u0, status, iters, resid, y, tail.data())
                                     : solve_rn_g(G, cfg, st, qp, B, z, u0, status, iters, resid, y, tail.data())) return -1;
    if (n_tail) *n_tail = tail[0];
    g_emu_tail2 = tail[0];
    if (g_emu_lean_tail && !reduced_native_tt(*cfg, *st) && reduced_native_tail(*cfg, *st)) {
      // the reduced-native tail solver first; the general kernel on what that leaves
      std::vector<int> tail2(B + 1, 0);
      if (lane_split(64, cfg->N) == 32) solve_rn_tail<64, 32>(cfg, st, qp, B, z, u0, status, iters, resid, y, tail.data(), tail2.data());
      else if (g_emu_lean_tail != 2) solve_rn_tail<32, 16>(cfg, st, qp, B, z, u0, status, iters, resid, y, tail.data(), tail2.data());
      else solve_rn_tail<64, 16>(cfg, st, qp, B, z, u0, status, iters, resid, y, tail.data(), tail2.data());
      tail.swap(tail2);
      g_emu_tail2 = tail[0];
    }
    if (lane_split(64, cfg->N) == 16) SOLVE_G(64, 16, cfg, st, qp, B, z, u0, status, iters, resid, y, nullptr, nullptr, 2, tail.data());
    else SOLVE_G(64, 32, cfg, st, qp, B, z, u0, status, iters, resid, y, nullptr, nullptr, 2, tail.data());
    return 0;
  }
  if (G == 64 || !early) return emu_solve(cfg, st, G, qp, B, z, u0, status, iters, resid, y);
  if (G == 32) SOLVE_G(32, 16, cfg, st, qp, B, z, u0, status, iters, resid, y, nullptr, nullptr, 1, tail.data());
  else if (G == 16) SOLVE_G(16, 16, cfg, st, qp, B, z, u0, status, iters, resid, y, nullptr, nullptr, 1, tail.data());
  else return -1;
  if (n_tail) *n_tail = tail[0];
  if (lane_split(64, cfg->N) == 16) SOLVE_G(64, 16, cfg, st, qp, B, z, u0, status, iters, resid, y, nullptr, nullptr, 2, tail.data());
  else SOLVE_G(64, 32, cfg, st, qp, B, z, u0, status, iters, resid, y, nullptr, nullptr, 2, tail.data());
  return 0;
}

// the reduced-native kernel alone: what it cannot certify stays UNSOLVED and is counted in *n_tail
extern "C" int emu_solve_rn(const mpmpc_config* cfg, const mpmpc_settings* st, int G, const double* qp, int B,
                            double* z, double* u0, int* status, int* iters, double* resid, double* y, int* n_tail) {
  if (cfg->N + 1 > G && !((G == 16 || G == 64) && cfg->N + 1 <= 2 * G && (reducible(*cfg, *st) || (G == 64 && reducible_tt(*cfg, *st))))) return -1;
  std::vector<int> tail(B + 1, 0);
  if (reducible_tt(*cfg, *st)) {
    if (solve_rnt_g(G, cfg, st, qp, B, z, u0, status, iters, resid, y, tail.data())) return -1;
    if (n_tail) *n_tail = tail[0];
    return 0;
  }
  if (!reducible(*cfg, *st)) return -1;
  if (solve_rn_g(G, cfg, st, qp, B, z, u0, status, iters, resid, y, tail.data())) return -1;
  if (n_tail) *n_tail = tail[0];
  return 0;
}
// the same kernel with the SEQUENTIAL elimination of the chains (the cyclic-reduction form is what ships): A/B in the tests
extern "C" int emu_solve_rn_sequential(const mpmpc_config* cfg, const mpmpc_settings* st, int G, const double* qp, int B,
                                       double* z, double* u0, int* status, int* iters, double* resid, double* y, int* n_tail) {
  if (cfg->N + 1 > G) return -1;
  std::vector<int> tail(B + 1, 0);
  const int C = lane_split(G, cfg->N);
  if (reducible_tt(*cfg, *st)) {          // the terminal-time kernels (one instance per wave)
    if (lane_split(64, cfg->N) == 16) solve_rnt<64, 16, false>(cfg, st, qp, B, z, u0, status, iters, resid, y, tail.data());
    else solve_rnt<64, 32, false>(cfg, st, qp, B, z, u0, status, iters, resid, y, tail.data());
    if (n_tail) *n_tail = tail[0];
    return 0;
  }
  if (!reducible(*cfg, *st)) return -1;
  if (G == 64 && C == 32) solve_rn<64, 32, false>(cfg, st, qp, B, z, u0, status, iters, resid, y, tail.data());
  else if (G == 64 && C == 16) solve_rn<64, 16, false>(cfg, st, qp, B, z, u0, status, iters, resid, y, tail.data());
  else if (G == 32) solve_rn<32, 16, false>(cfg, st, qp, B, z, u0, status, iters, resid, y, tail.data());
  else if (G == 16) solve_rn<16, 16, false>(cfg, st, qp, B, z, u0, status, iters, resid, y, tail.data());
  else return -1;
  if (n_tail) *n_tail = tail[0];
  return 0;
}
extern "C" int emu_reduced_native(const mpmpc_config* cfg, const mpmpc_settings* st) { return reduced_native(*cfg, *st) ? 1 : 0; }
extern "C" int emu_reduced_native_tt(const mpmpc_config* cfg, const mpmpc_settings* st) { return reduced_native_tt(*cfg, *st) ? 1 : 0; }

// the closed-loop variant: `guess` [B x ld] = active sets to start from (bit 30 = valid), `act` [B x ld] <- the
// active sets of the certified points (what mpmpc_solve_kernel<..., true> reads and writes in a rollout)
extern "C" int emu_solve_warm(const mpmpc_config* cfg, const mpmpc_settings* st, int G, const double* qp, int B,
                              const int* guess, double* z, double* u0, int* status, int* iters, double* resid,
                              double* y, int* act) {
  if (cfg->N + 1 > G) return -1;
  const int C = lane_split(G, cfg->N);
  if (reduced_native(*cfg, *st)) {
    // what the launcher runs in a warm-started closed-loop step: the reduced-native kernel with the guesses, then the
    // general kernel (no guess) on its tail
    std::vector<int> tail(B + 1, 0);
    if (solve_rn_g(G, cfg, st, qp, B, z, u0, status, iters, resid, y, tail.data(), guess, act)) return -1;
    if (lane_split(64, cfg->N) == 16) SOLVE_G(64, 16, cfg, st, qp, B, z, u0, status, iters, resid, y, nullptr, nullptr, 2, tail.data());
    else SOLVE_G(64, 32, cfg, st, qp, B, z, u0, status, iters, resid, y, nullptr, nullptr, 2, tail.data());
    return 0;
  }
  if (G != 64) return -1;          // the general kernels run one instance per wave
  if (G == 64 && C == 16) SOLVE_G(64, 16, cfg, st, qp, B, z, u0, status, iters, resid, y, guess, act);
  else if (G == 64) SOLVE_G(64, 32, cfg, st, qp, B, z, u0, status, iters, resid, y, guess, act);
  else if (G == 32) SOLVE_G(32, 16, cfg, st, qp, B, z, u0, status, iters, resid, y, guess, act);
  else if (G == 16) SOLVE_G(16, 16, cfg, st, qp, B, z, u0, status, iters, resid, y, guess, act);
  else return -1;
  return 0;
}

extern "C" int emu_assemble(const mpmpc_config* cfg, int n_wp, const double* kappa, const double* v_ref,
                            const double* ds_next, int n_cols, const double* ub_tab, const double* lb_tab, int B,
                            const int* wp_id, const double* x0, const double* cc, const double* lb,
                            const double* ub, double* qp) {
  using L = LaneEmu<64>;
  PathTables t{kappa, v_ref, ds_next, n_wp, ub_tab, lb_tab, n_cols};
  const int ld = stage_ld(cfg->N);
  const int total = B * ld;
  for (int t0 = 0; t0 < total; t0 += EMU_W) {
    VI inst, k;
    for (int i = 0; i < EMU_W; ++i) { inst.v[i] = (t0 + i) / ld; k.v[i] = (t0 + i) % ld; }
    assemble_lane<L>(*cfg, t, B, ld, inst, k, wp_id, x0, cc, lb, ub, qp);
  }
  return 0;
}

extern "C" int emu_stage_ld(int N) { return stage_ld(N); }
extern "C" void emu_event_counts(long long* out32, int reset) {
  for (int i = 0; i < 32; ++i) { out32[i] = g_emu_count[i]; if (reset) g_emu_count[i] = 0; }
}

// instruction census of everything executed since the last reset (only with -DMPMPC_COUNT_OPS)
// out7: wave instructions by class, all contexts together
extern "C" int emu_op_count(long long* out7, int reset) {
#ifdef MPMPC_COUNT_OPS
  OpCensus& s = op_census();
  for (int i = 0; i < 7; ++i) out7[i] = 0;
  for (int m = 0; m < 3; ++m) {
    const OpCount& c = s.c[m];
    out7[0] += c.fma; out7[1] += c.addmul; out7[2] += c.div; out7[3] += c.sqrt; out7[4] += c.cmpsel; out7[5] += c.shift; out7[6] += c.reduce;
  }
  if (reset) s = OpCensus{};
  return 1;
#else
  (void)out7; (void)reset;
  return 0;
#endif
}
// out4: FP64 flops (FMA = 2) of the wave instructions by context - lane-parallel, lane-parallel in the split layout,
// inside serial sweeps (as executed) - and the flops of ONE step per serial sweep (what a stage needs of it)
extern "C" int emu_op_flops(double* out4, int reset) {
#ifdef MPMPC_COUNT_OPS
  OpCensus& s = op_census();
  out4[0] = double(op_flops(s.c[0])); out4[1] = double(op_flops(s.c[1])); out4[2] = double(op_flops(s.c[2]));
  out4[3] = s.serial_useful;
  if (reset) s = OpCensus{};
  return 1;
#else
  (void)out4; (void)reset;
  return 0;
#endif
}

// Structure-exploiting flop count of the two linear-algebra pieces every iteration is made of, measured on the
// counting build: one factor() and one kkt_solve() of a <64,16> instance of horizon N with unit data.
// out4: factor lane-parallel, factor one-serial-step, kkt_solve lane-parallel, kkt_solve one-step-per-sweep (summed)
extern "C" int emu_census_pieces(int N, double* out4) {
#ifdef MPMPC_COUNT_OPS
  using L = LaneEmu<64, 16>;
  if (N + 1 > 32) return 0;
  Solver<L> s;
  typename L::real fields[MPMPC_NUM_FIELDS];
  for (int f = 0; f < MPMPC_NUM_FIELDS; ++f) fields[f] = VD(f == F_DS ? 0.05 : (f >= F_P ? 1.0 : (f >= F_HI && f < F_Q ? 1.0 : (f >= F_LO && f < F_HI ? -1.0 : 0.01))));
  VI inst(0), k = L::stage() - lane_offset(64, 16, N);
  s.load(fields, 1, inst, k, N);
  VD h[5], rx[5], req[3], xt[5], nu[3];
  for (int j = 0; j < 5; ++j) { h[j] = VD(0.5); rx[j] = VD(1.0); }
  for (int i = 0; i < 3; ++i) req[i] = VD(1.0);
  double o[4];
  op_census() = OpCensus{};
  s.factor(h, VD(1e-3));
  out4[0] = double(op_flops(op_census().c[0])); out4[1] = op_census().serial_useful;
  op_census() = OpCensus{};
  s.kkt_solve(rx, req, xt, nu);
  out4[2] = double(op_flops(op_census().c[0])); out4[3] = op_census().serial_useful;
  op_census() = OpCensus{};
  (void)o;
  return 1;
#else
  (void)N; (void)out4;
  return 0;
#endif
}

// returns the number of start waypoints without a free segment (>= 0), or: -3000 a border cell outside the map,
// -3001 more than COR_MAXSEG free segments on a line (what mpmpc_build_corridor reports as errors), -1000 the two
// forms of the column walk disagree
extern "C" int emu_corridor(int height, int width, const int8_t* data, double ox, double oy, double res, int n_wp,
                            const double* x, const double* y, const double* psi, const double* ds_next, int circular,
                            const double* bub, const double* blb, int n_cols, double min_width, double safety_margin,
                            double* ub_tab, double* lb_tab, int* nseg_out) {
  MapView mv{data, height, width, ox, oy, res};
  std::vector<double> trig((size_t)n_wp * COR_TRIG);
  for (int i = 0; i < n_wp; ++i) cor_trig_row(psi[i], trig.data() + (size_t)i * COR_TRIG);
  PathGeom pg{x, y, psi, ds_next, n_wp, circular, trig.data()};
  for (int i = 0; i < n_wp; ++i) {
    int cx, cy;
    cor_w2m(mv, bub[2 * i], bub[2 * i + 1], cx, cy);
    if (cx < 0 || cx >= width || cy < 0 || cy >= height) return -3000;
    cor_w2m(mv, blb[2 * i], blb[2 * i + 1], cx, cy);
    if (cx < 0 || cx >= width || cy < 0 || cy >= height) return -3000;
  }
  double* segs = new double[(size_t)n_wp * 4 * COR_MAXSEG]();
  int* nseg = new int[n_wp];
  for (int i = 0; i < n_wp; ++i) {
    nseg[i] = cor_free_segments(mv, bub[2 * i], bub[2 * i + 1], blb[2 * i], blb[2 * i + 1], min_width, segs + (size_t)i * 4 * COR_MAXSEG);
    if (nseg[i] < 0) { delete[] segs; delete[] nseg; return -3001; }
    // the staged form the device runs (cells of the line as packed 16-bit pairs, then the scan over the copy) must
    // give the same bits
    {
      int cells[COR_CELL_CAP];
      double seg2[4 * COR_MAXSEG] = {0};
      int ux, uy, lx, ly;
      cor_w2m(mv, bub[2 * i], bub[2 * i + 1], ux, uy);
      cor_w2m(mv, blb[2 * i], blb[2 * i + 1], lx, ly);
      const int n = cor_line_cells(ux, uy, lx, ly, cells, COR_CELL_CAP);
      if (n <= COR_CELL_CAP) {
        const int c2 = cor_scan_cells(mv, ux, uy, lx, ly, min_width, n, [&](int c, int& cx, int& cy) { cor_unpack_cell(cells[c], cx, cy); },
                                      [&](int c) { int cx, cy; cor_unpack_cell(cells[c], cx, cy); return cor_cell_free(mv, cx, cy); }, seg2);
        if (c2 != nseg[i] || std::memcmp(seg2, segs + (size_t)i * 4 * COR_MAXSEG, sizeof(double) * 4 * c2) != 0) {
          delete[] segs; delete[] nseg;
          return -2000;
        }
      }
    }
  }
  double* wpc = new double[(size_t)n_wp * COR_WPC]();
  for (int i = 0; i < n_wp; ++i)
    if (nseg[i] <= 1) cor_forced(pg, segs, nseg, i, safety_margin, wpc + (size_t)i * COR_WPC);
  int bad = 0;
  for (int w = 0; w < n_wp; ++w) {
    if (!cor_select(pg, segs, nseg, w + 1, n_cols, safety_margin, ub_tab + (size_t)w * n_cols, lb_tab + (size_t)w * n_cols, wpc)) {
      ++bad;
      for (int n = 0; n < n_cols; ++n) ub_tab[(size_t)w * n_cols + n] = lb_tab[(size_t)w * n_cols + n] = std::numeric_limits<double>::quiet_NaN();
      continue;
    }
    // the per-column form the device runs (one thread per (start waypoint, column)) must give the same bits
    for (int n = 0; n < n_cols; ++n) {
      double ub, lb;
      if (!cor_select_one(pg, segs, nseg, w + 1, n, safety_margin, wpc, &ub, &lb) || ub != ub_tab[(size_t)w * n_cols + n] ||
          lb != lb_tab[(size_t)w * n_cols + n]) {
        delete[] segs; delete[] nseg; delete[] wpc;
        return -1000;
      }
    }
  }
  if (nseg_out) for (int i = 0; i < n_wp; ++i) nseg_out[i] = nseg[i];
  delete[] segs; delete[] nseg; delete[] wpc;
  return bad;
}

// host runs of the per-car rollout code that the K3 kernels execute per thread
extern "C" int emu_localise(int n_wp, const double* cum, const double* gx, const double* gy, const double* gpsi, double s,
                            const double* pose, double* x0) {
  int wp = ro_current_waypoint(cum, n_wp, s);
  if (wp >= 0) ro_t2s(pose[0], pose[1], pose[2], gx[wp], gy[wp], gpsi[wp], x0);
  return wp;
}
extern "C" int emu_advance(int N, double L, double Ts, int status, const double* z, double* cc, int* counter,
                           const double* x0, double kappa_wp, double* pose, double* s, double* u_out) {
  return ro_advance(N, L, Ts, status, z, cc, counter, x0, kappa_wp, pose, s, u_out) ? 1 : 0;
}

// host run of the speed-profile code that mpmpc_speed_profile_kernel executes per thread
#include "speed_core.hpp"
#include <cstring>
#include <vector>
extern "C" int emu_speed_profile(int n, const double* li, const double* kappa, const double* lim5, double eps,
                                 double* v, int* iters) {
  std::vector<double> ws((size_t)SP_ARRAYS * n);
  SpWork W{ws.data(), n, 1};
  SpLimits lim{lim5[0], lim5[1], lim5[2], lim5[3], lim5[4]};
  return sp_solve(n, li, kappa, 1, lim, eps, W, v, 1, iters);
}
