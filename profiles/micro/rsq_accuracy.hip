// Accuracy of the hardware seeds v_rsq_f64 / v_rcp_f64 and of the Newton variants built on them, against
// correctly rounded 1/sqrt(x) and 1/x (host long double):  hipcc --offload-arch=gfx950 -O3 -o rsq_accuracy rsq_accuracy.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
__global__ void k(const double* x, double* out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double a = x[i];
  double r0 = __builtin_amdgcn_rsq(a);
  double e = __builtin_fma(-(a * r0), r0, 1.0);
  double r1 = __builtin_fma(r0 * e, __builtin_fma(0.375, e, 0.5), r0);       // one cubic step
  e = __builtin_fma(-(a * r1), r1, 1.0);
  double r2 = __builtin_fma(r1 * e, 0.5, r1);                                  // + one quadratic step
  double c0 = __builtin_amdgcn_rcp(a);
  double f = __builtin_fma(-a, c0, 1.0);
  double c1 = __builtin_fma(c0, f, c0);                                        // one quadratic step
  f = __builtin_fma(-a, c1, 1.0);
  double c2 = __builtin_fma(c1, f, c1);                                        // two quadratic steps
  f = __builtin_fma(-a, c0, 1.0);
  double c3 = __builtin_fma(c0, __builtin_fma(f, f, f), c0);                   // one cubic step
  out[7 * i + 0] = r0; out[7 * i + 1] = r1; out[7 * i + 2] = r2; out[7 * i + 3] = c0; out[7 * i + 4] = c1; out[7 * i + 5] = c2; out[7 * i + 6] = c3;
}
int main() {
  const int n = 1 << 20;
  std::vector<double> x(n), out(7 * n);
  std::mt19937_64 g(1);
  std::uniform_real_distribution<double> u(-60.0, 60.0), m(1.0, 2.0);
  for (auto& v : x) v = std::ldexp(m(g), (int)u(g));
  double *dx, *dout;
  (void)hipMalloc(&dx, sizeof(double) * n); (void)hipMalloc(&dout, sizeof(double) * 7 * n);
  (void)hipMemcpy(dx, x.data(), sizeof(double) * n, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(dx, dout, n);
  (void)hipMemcpy(out.data(), dout, sizeof(double) * 7 * n, hipMemcpyDeviceToHost);
  const char* names[7] = {"v_rsq_f64 seed", "rsq + cubic", "rsq + cubic + quadratic", "v_rcp_f64 seed", "rcp + 1 Newton", "rcp + 2 Newton", "rcp + cubic"};
  for (int j = 0; j < 7; ++j) {
    long double worst = 0;
    for (int i = 0; i < n; ++i) {
      long double ref = j < 3 ? 1.0L / sqrtl((long double)x[i]) : 1.0L / (long double)x[i];
      long double err = fabsl(((long double)out[7 * i + j] - ref) / ref);
      if (err > worst) worst = err;
    }
    printf("%-26s max relative error %.3Le  (%.2Lf ulp of 2^-53)\n", names[j], worst, worst / 1.1102230246251565e-16L);
  }
  return 0;
}
