// Micro-benchmark: does a wave64 whose upper 32 lanes are masked off (EXEC = low half) issue FP64 VALU
// instructions faster than a full wave?  One wave per SIMD (1024 blocks of 64), 8 independent FMA chains.
//   hipcc --offload-arch=gfx950 -O3 -o exec_half exec_half.hip && ./exec_half
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(64) void k(double* out, int iters, int active) {
  if ((int)(threadIdx.x & 63) >= active) return;
  double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const double b = 1.0000001, c = 1e-9;
  for (int i = 0; i < iters; ++i) {
    a0 = __builtin_fma(a0, b, c); a1 = __builtin_fma(a1, b, c); a2 = __builtin_fma(a2, b, c); a3 = __builtin_fma(a3, b, c);
    a4 = __builtin_fma(a4, b, c); a5 = __builtin_fma(a5, b, c); a6 = __builtin_fma(a6, b, c); a7 = __builtin_fma(a7, b, c);
  }
  out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
int main() {
  double* d; hipMalloc(&d, sizeof(double) * 64 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int active : {64, 48, 32, 16}) {
    k<<<1024, 64>>>(d, 1000, active); hipDeviceSynchronize();
    hipEventRecord(e0); k<<<1024, 64>>>(d, 100000, active); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("active lanes %2d: %.3f ms for 800000 FMAs per lane -> %.2f ns per wave instruction\n", active, ms, ms * 1e6 / 800000.0);
  }
  return 0;
}
