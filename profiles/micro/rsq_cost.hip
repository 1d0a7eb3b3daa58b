// Cost of the reciprocal-square-root seed inside a dependent chain (one wave per SIMD, like K2's factor loop):
// v_rsq_f64 against v_cvt_f32_f64 + v_rsq_f32 + v_cvt_f64_f32, each followed by the cubic Newton step, plus the
// accuracy of the f32-seeded variant.   hipcc --offload-arch=gfx950 -O3 -o rsq_cost rsq_cost.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
__device__ __forceinline__ double rsq64(double a) {
  double y = __builtin_amdgcn_rsq(a);
  double e = __builtin_fma(-(a * y), y, 1.0);
  return __builtin_fma(y * e, __builtin_fma(0.375, e, 0.5), y);
}
__device__ __forceinline__ double rsq32(double a) {
  double y = (double)__builtin_amdgcn_rsqf((float)a);
  double e = __builtin_fma(-(a * y), y, 1.0);
  return __builtin_fma(y * e, __builtin_fma(0.375, e, 0.5), y);
}
template <int V>
__global__ void chain(double* out, int iters, double seed) {
  double x = seed + threadIdx.x * 1e-3;
  long long t0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
    // dependent chain: x -> rsq(x) -> x (kept near 1)
    double r = V == 0 ? rsq64(x) : rsq32(x);
    x = __builtin_fma(r, 0.5, 0.75);
  }
  long long t1 = wall_clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = x;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[1 << 20] = (double)(t1 - t0);
}
template <int V>
__global__ void indep(double* out, int iters, double seed) {
  double x0 = seed + threadIdx.x * 1e-3, x1 = x0 + 0.1, x2 = x0 + 0.2, x3 = x0 + 0.3;
  long long t0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
    double r0 = V == 0 ? rsq64(x0) : rsq32(x0), r1 = V == 0 ? rsq64(x1) : rsq32(x1);
    double r2 = V == 0 ? rsq64(x2) : rsq32(x2), r3 = V == 0 ? rsq64(x3) : rsq32(x3);
    x0 = __builtin_fma(r0, 0.5, 0.75); x1 = __builtin_fma(r1, 0.5, 0.75); x2 = __builtin_fma(r2, 0.5, 0.75); x3 = __builtin_fma(r3, 0.5, 0.75);
  }
  long long t1 = wall_clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[1 << 20] = (double)(t1 - t0);
}
__global__ void acc(const double* x, double* out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { out[2 * i] = rsq64(x[i]); out[2 * i + 1] = rsq32(x[i]); }
}
int main(int argc, char** argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 1024;      // 1024 = one wave per SIMD; 2048 / 4096: two / four
  double* d;
  (void)hipMalloc(&d, sizeof(double) * ((1 << 20) + 8));
  printf("%d waves of 64 lanes\n", blocks);
  const int iters = 20000;
  double ticks;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto report = [&](const char* name, int per) {
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("[kernel %.3f ms] ", ms);
    (void)hipMemcpy(&ticks, d + (1 << 20), sizeof(double), hipMemcpyDeviceToHost);
    printf("%-44s %7.1f ns per rsqrt\n", name, ticks * 10.0 / iters / per);
  };
  (void)hipEventRecord(e0); chain<0><<<blocks, 64>>>(d, iters, 1.0); report("v_rsq_f64 + cubic, dependent chain", 1);
  (void)hipEventRecord(e0); chain<1><<<blocks, 64>>>(d, iters, 1.0); report("cvt + v_rsq_f32 + cvt + cubic, dependent", 1);
  (void)hipEventRecord(e0); indep<0><<<blocks, 64>>>(d, iters, 1.0); report("v_rsq_f64 + cubic, 4 independent", 4);
  (void)hipEventRecord(e0); indep<1><<<blocks, 64>>>(d, iters, 1.0); report("cvt + v_rsq_f32 + cvt + cubic, 4 independent", 4);
  const int n = 1 << 19;
  std::vector<double> x(n), out(2 * n);
  std::mt19937_64 g(1);
  std::uniform_real_distribution<double> u(-35.0, 40.0), m(1.0, 2.0);
  for (auto& v : x) v = std::ldexp(m(g), (int)u(g));
  double* dx;
  (void)hipMalloc(&dx, sizeof(double) * n);
  (void)hipMemcpy(dx, x.data(), sizeof(double) * n, hipMemcpyHostToDevice);
  acc<<<n / 256, 256>>>(dx, d, n);
  (void)hipMemcpy(out.data(), d, sizeof(double) * 2 * n, hipMemcpyDeviceToHost);
  for (int j = 0; j < 2; ++j) {
    long double worst = 0;
    for (int i = 0; i < n; ++i) {
      long double ref = 1.0L / sqrtl((long double)x[i]);
      long double err = fabsl(((long double)out[2 * i + j] - ref) / ref);
      if (err > worst) worst = err;
    }
    printf("%-44s max relative error %.3Le (%.2Lf ulp)\n", j ? "f32 seed + cubic (x in 2^-35 .. 2^40)" : "f64 seed + cubic", worst, worst / 1.1102230246251565e-16L);
  }
  return 0;
}
