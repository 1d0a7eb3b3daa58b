// The STORE PATTERN of K1 alone (no assembly arithmetic, no table gathers): one thread per (instance, stage), 27 planes of
// B x 32 doubles, a wavefront writes 64 consecutive doubles of one plane per store - against a plain fill of the same bytes.
//   hipcc --offload-arch=gfx950 -O3 -o k1_pattern k1_pattern.hip && ./k1_pattern
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
constexpr int F = 27;
// variant 0: K1's pattern (8-byte stores, one pair per thread)
__global__ __launch_bounds__(256) void planes8(double* __restrict__ out, size_t P, double v) {
  const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
#pragma unroll
  for (int f = 0; f < F; ++f) out[f * P + p] = v + f;
}
// variant 1: the same bytes, 16-byte stores (two consecutive pairs per thread)
__global__ __launch_bounds__(256) void planes16(double* __restrict__ out, size_t P, double v) {
  const size_t p = ((size_t)blockIdx.x * 256 + threadIdx.x) * 2;
  if (p >= P) return;
#pragma unroll
  for (int f = 0; f < F; ++f) *reinterpret_cast<double2*>(out + f * P + p) = make_double2(v + f, v - f);
}
// variant 2: a block walks ONE plane at a time over a larger chunk (4 x 256 pairs): 8 KB runs per plane and block
__global__ __launch_bounds__(256) void planes8x4(double* __restrict__ out, size_t P, double v) {
  const size_t base = (size_t)blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
  for (int f = 0; f < F; ++f)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (base + j * 256 < P) out[f * P + base + j * 256] = v + f;
}
// variant 3: plain fill of the same number of bytes, 8-byte stores, grid-stride
__global__ __launch_bounds__(256) void fill8(double* __restrict__ out, size_t n, double v) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = v;
}
// variant 4: K1's pattern with a read of two input planes per pair (the corridor rows lb / ub)
__global__ __launch_bounds__(256) void planes8_in(double* __restrict__ out, const double* __restrict__ in, size_t P, double v) {
  const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  const double a = in[p], b = in[P + p];
#pragma unroll
  for (int f = 0; f < F; ++f) out[f * P + p] = a * f + b;
}
template <class Fn> static void time_it(const char* name, size_t bytes, Fn fn) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  std::vector<float> ms;
  for (int i = 0; i < 40; ++i) { hipEventRecord(e0); fn(); hipEventRecord(e1); hipEventSynchronize(e1); float t; hipEventElapsedTime(&t, e0, e1); if (i >= 5) ms.push_back(t); }
  std::sort(ms.begin(), ms.end());
  std::printf("%-44s min %.4f med %.4f ms -> %.3f / %.3f of 8 TB/s\n", name, ms.front(), ms[ms.size() / 2], bytes / (ms.front() * 1e-3) / 8e12, bytes / (ms[ms.size() / 2] * 1e-3) / 8e12);
}
int main() {
  for (int B : {8192, 65536}) {
    const size_t P = (size_t)B * 32, n = P * F;
    double *out, *in; hipMalloc(&out, n * 8); hipMalloc(&in, 2 * P * 8); hipMemset(in, 0, 2 * P * 8);
    std::printf("B = %d (%.0f MB written)\n", B, n * 8 / 1e6);
    const int blocks = (int)((P + 255) / 256);
    time_it("planes, 8-byte stores (K1's pattern)", n * 8, [&] { planes8<<<blocks, 256>>>(out, P, 1.0); });
    time_it("planes, 16-byte stores", n * 8, [&] { planes16<<<(blocks + 1) / 2, 256>>>(out, P, 1.0); });
    time_it("planes, 8-byte stores, 4 chunks per block", n * 8, [&] { planes8x4<<<(blocks + 3) / 4, 256>>>(out, P, 1.0); });
    time_it("planes, 8-byte stores + 2 input planes", n * 8 + 2 * P * 8, [&] { planes8_in<<<blocks, 256>>>(out, in, P, 1.0); });
    time_it("plain fill, 8-byte stores, 8192 blocks", n * 8, [&] { fill8<<<8192, 256>>>(out, n, 1.0); });
    time_it("plain fill, 8-byte stores, 65536 blocks", n * 8, [&] { fill8<<<65536, 256>>>(out, n, 1.0); });
    time_it("hipMemsetAsync", n * 8, [&] { hipMemsetAsync(out, 0, n * 8, 0); });
    hipFree(out); hipFree(in);
  }
  return 0;
}
