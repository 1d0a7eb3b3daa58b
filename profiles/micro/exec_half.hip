// Micro-benchmarks behind two statements of DESIGN.md:
//  (1) a wave64 whose upper lanes are masked off issues FP64 VALU instructions no faster than a full wave;
//  (2) cycles per wave instruction (s_memtime around the loop) for FP64 FMA, FP32 FMA and a DPP move,
//      one wave per SIMD (1024 blocks of 64).
//   hipcc --offload-arch=gfx950 -O3 -o exec_half exec_half.hip && ./exec_half
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KIND>
__global__ __launch_bounds__(64) void k(double* out, long long* cyc, int iters, int active) {
  if ((int)(threadIdx.x & 63) >= active) return;
  double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float f0 = threadIdx.x, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7;
  int m0 = threadIdx.x, m1 = m0 + 1, m2 = m0 + 2, m3 = m0 + 3, m4 = m0 + 4, m5 = m0 + 5, m6 = m0 + 6, m7 = m0 + 7;
  const double b = 1.0000001, c = 1e-9;
  const float bf = 1.0000001f, cf = 1e-9f;
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    if (KIND == 6) {       // the FP64 body four times per loop trip: amortises the loop-back branch
      _Pragma("unroll") for (int r = 0; r < 4; ++r) {
        a0 = __builtin_fma(a0, b, c); a1 = __builtin_fma(a1, b, c); a2 = __builtin_fma(a2, b, c); a3 = __builtin_fma(a3, b, c);
        a4 = __builtin_fma(a4, b, c); a5 = __builtin_fma(a5, b, c); a6 = __builtin_fma(a6, b, c); a7 = __builtin_fma(a7, b, c);
      }
    } else if (KIND == 0) {
      a0 = __builtin_fma(a0, b, c); a1 = __builtin_fma(a1, b, c); a2 = __builtin_fma(a2, b, c); a3 = __builtin_fma(a3, b, c);
      a4 = __builtin_fma(a4, b, c); a5 = __builtin_fma(a5, b, c); a6 = __builtin_fma(a6, b, c); a7 = __builtin_fma(a7, b, c);
    } else if (KIND == 1) {
      f0 = __builtin_fmaf(f0, bf, cf); f1 = __builtin_fmaf(f1, bf, cf); f2 = __builtin_fmaf(f2, bf, cf); f3 = __builtin_fmaf(f3, bf, cf);
      f4 = __builtin_fmaf(f4, bf, cf); f5 = __builtin_fmaf(f5, bf, cf); f6 = __builtin_fmaf(f6, bf, cf); f7 = __builtin_fmaf(f7, bf, cf);
    } else if (KIND == 3) {       // integer adds
      m0 += i; m1 += m0; m2 += i; m3 += m2; m4 += i; m5 += m4; m6 += i; m7 += m6;
    } else if (KIND == 4) {       // 4 FP64 FMAs interleaved with 4 DPP moves
      a0 = __builtin_fma(a0, b, c); m0 = __builtin_amdgcn_update_dpp(0, m0, 0x111, 0xf, 0xf, true);
      a1 = __builtin_fma(a1, b, c); m1 = __builtin_amdgcn_update_dpp(0, m1, 0x111, 0xf, 0xf, true);
      a2 = __builtin_fma(a2, b, c); m2 = __builtin_amdgcn_update_dpp(0, m2, 0x111, 0xf, 0xf, true);
      a3 = __builtin_fma(a3, b, c); m3 = __builtin_amdgcn_update_dpp(0, m3, 0x111, 0xf, 0xf, true);
    } else if (KIND == 5) {       // FP64 add and mul instead of FMA
      a0 = a0 * b; a1 = a1 + c; a2 = a2 * b; a3 = a3 + c; a4 = a4 * b; a5 = a5 + c; a6 = a6 * b; a7 = a7 + c;
    } else {
      m0 = __builtin_amdgcn_update_dpp(0, m0, 0x111, 0xf, 0xf, true); m1 = __builtin_amdgcn_update_dpp(0, m1, 0x111, 0xf, 0xf, true);
      m2 = __builtin_amdgcn_update_dpp(0, m2, 0x111, 0xf, 0xf, true); m3 = __builtin_amdgcn_update_dpp(0, m3, 0x111, 0xf, 0xf, true);
      m4 = __builtin_amdgcn_update_dpp(0, m4, 0x111, 0xf, 0xf, true); m5 = __builtin_amdgcn_update_dpp(0, m5, 0x111, 0xf, 0xf, true);
      m6 = __builtin_amdgcn_update_dpp(0, m6, 0x111, 0xf, 0xf, true); m7 = __builtin_amdgcn_update_dpp(0, m7, 0x111, 0xf, 0xf, true);
    }
  }
  long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + m0 + m1 + m2 + m3 + m4 + m5 + m6 + m7;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int KIND>
void run(const char* name, double* d, long long* c, int active) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 100000;
  k<KIND><<<1024, 64>>>(d, c, 1000, active); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); k<KIND><<<1024, 64>>>(d, c, iters, active); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  long long cyc; (void)hipMemcpy(&cyc, c, sizeof(cyc), hipMemcpyDeviceToHost);
  printf("%-9s active lanes %2d: %.3f ms, %.2f ns and %.2f counter ticks per wave instruction\n", name, active, ms,
         ms * 1e6 / (8.0 * iters), (double)cyc / (8.0 * iters));
}
int main() {
  double* d; long long* c;
  (void)hipMalloc(&d, sizeof(double) * 64 * 1024); (void)hipMalloc(&c, sizeof(long long));
  for (int active : {64, 32, 16}) run<0>("fma f64", d, c, active);
  run<1>("fma f32", d, c, 64);
  run<2>("dpp mov", d, c, 64);
  run<3>("int add", d, c, 64);
  run<4>("f64+dpp", d, c, 64);
  run<5>("f64 add/mul", d, c, 64);
  run<6>("fma f64 x4 (ticks and ns are per 4 instructions)", d, c, 64);
  return 0;
}
