// What is SQ_WAIT_ANY made of in the solve kernels?  Micro-kernels that isolate one candidate each, run at ONE wave
// per SIMD (1024 blocks x 40 KB of LDS: four blocks per CU) and at TWO (2048 x 20 KB), timed with wall_clock64 and -
// under `rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU
// SQ_INSTS_VALU SQ_INSTS_SALU` - measured with the same counters as the real kernel (profiles/collect_wait.sh).
//   fma_indep     8 independent FP64 FMA chains                      (issue bound: the floor)
//   fma_dep       one dependent FP64 FMA chain                       (result latency beyond the 4-cycle issue slot)
//   mul_add_dep   dependent v_mul_f64 / v_add_f64 / v_max_f64 mix    (the non-FMA arithmetic of the lane code)
//   dpp_dep       fma -> v_mov_b32_dpp x2 -> fma ...                 (a lane shift between dependent FMAs: the sweeps)
//   sweep         the real substitution step: 6 DPP moves + 9 FMAs, 2 x 2 blocks (s_solve2's in_step), straight line
//   sweep_loop1   the same, one step per loop trip                   (a taken branch every 10 instructions)
//   sweep_loop4   the same, four steps per trip                      (what the library ships)
//   trans_dep     v_rsq_f64 + cubic Newton, dependent                (the Cholesky pivots)
//   sel_dep       dependent v_cndmask pairs                          (select(active, new, old))
//   lds_rt        ds_write_b64 + ds_read_b64 + wait, dependent       (cold storage round trip)
// hipcc --offload-arch=gfx950 -O3 -o dep_wait dep_wait.hip ;  ./dep_wait [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int CTRL>
__device__ __forceinline__ double dpp(double a) {
  int lo = __double2loint(a), hi = __double2hiint(a);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double rsq64(double a) {
  double y = __builtin_amdgcn_rsq(a);
  double e = __builtin_fma(-(a * y), y, 1.0);
  return __builtin_fma(y * e, __builtin_fma(0.375, e, 0.5), y);
}
#define FMA __builtin_fma

template <int LDS_KB> __device__ __forceinline__ double* pad() {
  __shared__ double buf[LDS_KB * 128];
  return buf;
}
#define PROLOGUE                                                             \
  double* lds = pad<LDS_KB>();                                               \
  lds[threadIdx.x] = seed;                                                   \
  double x = seed + threadIdx.x * 1e-6;                                      \
  const double ca = 0.999999, cb = 1e-7;                                     \
  long long t0 = wall_clock64();
#define EPILOGUE(val)                                                        \
  long long t1 = wall_clock64();                                             \
  out[blockIdx.x * 64 + threadIdx.x] = (val) + lds[63 - threadIdx.x];        \
  if (threadIdx.x == 0 && blockIdx.x == 0) out[1 << 20] = (double)(t1 - t0);

template <int LDS_KB> __global__ __launch_bounds__(64) void fma_indep(double* out, int iters, double seed) {
  PROLOGUE
  double y0 = x, y1 = x + 1, y2 = x + 2, y3 = x + 3, y4 = x + 4, y5 = x + 5, y6 = x + 6, y7 = x + 7;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      y0 = FMA(y0, ca, cb); y1 = FMA(y1, ca, cb); y2 = FMA(y2, ca, cb); y3 = FMA(y3, ca, cb);
      y4 = FMA(y4, ca, cb); y5 = FMA(y5, ca, cb); y6 = FMA(y6, ca, cb); y7 = FMA(y7, ca, cb);
    }
  }
  EPILOGUE(y0 + y1 + y2 + y3 + y4 + y5 + y6 + y7)
}
template <int LDS_KB> __global__ __launch_bounds__(64) void fma_dep(double* out, int iters, double seed) {
  PROLOGUE
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 64; ++u) x = FMA(x, ca, cb);
  }
  EPILOGUE(x)
}
template <int LDS_KB> __global__ __launch_bounds__(64) void mul_add_dep(double* out, int iters, double seed) {
  PROLOGUE
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) { x = x * ca; x = x + cb; x = __builtin_fmax(x, cb); x = x * ca; }
  }
  EPILOGUE(x)
}
template <int LDS_KB> __global__ __launch_bounds__(64) void dpp_dep(double* out, int iters, double seed) {
  PROLOGUE
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 32; ++u) x = FMA(dpp<0x111>(x), ca, cb);      // row_shr:1
  }
  EPILOGUE(x)
}
// the substitution step of the reduced problem (mpmpc_core.hpp: s_solve2, in_step): 4 DPP dwords + 4 FMAs
#define SWEEP_STEP                                                 \
  {                                                                \
    double p0 = dpp<0x111>(y0), p1 = dpp<0x111>(y1);               \
    y0 = FMA(g1, p1, FMA(g0, p0, c0));                             \
    y1 = FMA(g3, p1, FMA(g2, p0, c1));                             \
  }
template <int LDS_KB> __global__ __launch_bounds__(64) void sweep(double* out, int iters, double seed) {
  PROLOGUE
  double y0 = x, y1 = x * 0.5;
  const double g0 = 0.3 + seed * 1e-9, g1 = -0.2, g2 = 0.1, g3 = 0.4, c0 = 1e-3, c1 = 2e-3;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) SWEEP_STEP
  }
  EPILOGUE(y0 + y1)
}
template <int LDS_KB, int PER_TRIP> __global__ __launch_bounds__(64) void sweep_loop(double* out, int iters, double seed) {
  PROLOGUE
  double y0 = x, y1 = x * 0.5;
  const double g0 = 0.3 + seed * 1e-9, g1 = -0.2, g2 = 0.1, g3 = 0.4, c0 = 1e-3, c1 = 2e-3;
  const int trips = iters * 16 / PER_TRIP;
#pragma unroll 1
  for (int i = 0; i < trips; ++i) {
#pragma unroll
    for (int u = 0; u < PER_TRIP; ++u) SWEEP_STEP
  }
  EPILOGUE(y0 + y1)
}
template <int LDS_KB> __global__ __launch_bounds__(64) void trans_dep(double* out, int iters, double seed) {
  PROLOGUE
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) x = FMA(rsq64(x), 0.5, 0.75);
  }
  EPILOGUE(x)
}
template <int LDS_KB> __global__ __launch_bounds__(64) void sel_dep(double* out, int iters, double seed) {
  PROLOGUE
  double o = x + 1.0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 32; ++u) { double n = FMA(x, ca, cb); x = (n > o) ? n : x; o = o * ca; }
  }
  EPILOGUE(x + o)
}
template <int LDS_KB> __global__ __launch_bounds__(64) void lds_rt(double* out, int iters, double seed) {
  PROLOGUE
  volatile double* v = lds + 64;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) { v[threadIdx.x] = x; x = FMA(v[threadIdx.x ^ 1], ca, cb); }
  }
  EPILOGUE(x)
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;
  double* d;
  (void)hipMalloc(&d, sizeof(double) * ((1 << 20) + 8));
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto run = [&](const char* name, auto launch, int waves_per_simd, double insts_per_iter) {
    for (int rep = 0; rep < 3; ++rep) {          // the last repetition is reported (clocks ramped)
      (void)hipEventRecord(e0);
      launch();
      (void)hipEventRecord(e1);
      (void)hipDeviceSynchronize();
    }
    float ms = 0;
    double ticks = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(&ticks, d + (1 << 20), sizeof(double), hipMemcpyDeviceToHost);
    printf("%-14s %d wave/SIMD  kernel %8.3f ms  wave 0: %7.2f ns per instruction (%.0f instructions per iteration)\n", name,
           waves_per_simd, ms, ticks * 10.0 / iters / insts_per_iter, insts_per_iter);
  };
#define BOTH(name, kern, ipi)                                                                   \
  run(name, [&] { kern<40><<<1024, 64>>>(d, iters, 1.0); }, 1, ipi);                          \
  run(name, [&] { kern<20><<<2048, 64>>>(d, iters, 1.0); }, 2, ipi);
  BOTH("fma_indep", fma_indep, 64.0)
  BOTH("fma_dep", fma_dep, 64.0)
  BOTH("mul_add_dep", mul_add_dep, 64.0)
  BOTH("dpp_dep", dpp_dep, 96.0)
  BOTH("sweep", sweep, 128.0)
  run("sweep_loop1", [&] { sweep_loop<40, 1><<<1024, 64>>>(d, iters, 1.0); }, 1, 128.0);
  run("sweep_loop1", [&] { sweep_loop<20, 1><<<2048, 64>>>(d, iters, 1.0); }, 2, 128.0);
  run("sweep_loop4", [&] { sweep_loop<40, 4><<<1024, 64>>>(d, iters, 1.0); }, 1, 128.0);
  run("sweep_loop4", [&] { sweep_loop<20, 4><<<2048, 64>>>(d, iters, 1.0); }, 2, 128.0);
  BOTH("trans_dep", trans_dep, 8.0 * 6.0)
  BOTH("sel_dep", sel_dep, 32.0 * 5.0)
  BOTH("lds_rt", lds_rt, 12.0)
  return 0;
}
