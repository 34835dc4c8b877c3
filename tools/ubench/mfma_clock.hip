// Microbenchmark: sustained fp32 MFMA rate and shader clock of the whole chip as a function of the operand data
// (zeros / constant / random, changing every instruction): is the 157.3 TFLOP/s (2.4 GHz) figure reachable with dense data?
//   hipcc --offload-arch=gfx950 -O3 -o mfma_clock tools/ubench/mfma_clock.hip && ./mfma_clock
// 256 blocks x 512 threads (2 waves per SIMD), 8 independent accumulators per wave, ~60 ms per run.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters, int mode) {
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a[8], b[8];
  unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  for (int i = 0; i < 8; ++i) {
    s = s * 1664525u + 1013904223u; const float ra = ((int)(s >> 8) - (1 << 23)) * (1.f / (1 << 23));
    s = s * 1664525u + 1013904223u; const float rb = ((int)(s >> 8) - (1 << 23)) * (1.f / (1 << 23));
    a[i] = mode == 0 ? 0.f : mode == 1 ? 1.0f : ra;                 // 0 zeros, 1 constant, 2 random, 3 random, half of them zero (post-ReLU)
    b[i] = mode == 0 ? 0.f : mode == 1 ? 0.5f : rb * 0.05f;
    if (mode == 3 && (s & 0x100)) a[i] = 0.f;
  }
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it += 8) {  // operand pairs rotate with compile-time indices (no VALU work in the loop)
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int m = 0; m < 8; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], b[(m + u) & 7], acc[m], 0, 0, 0);
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float r = 0.f;
  for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][7];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  float* out; long long* cyc;
  (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 256 * 8);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 100000;  // x 8 MFMAs x 64 cycles x 2 waves per SIMD = 102 M cycles
  const char* names[4] = {"zeros", "constant", "random", "random, half zero"};
  for (int mode = 0; mode < 4; ++mode) {
    k<<<256, 512>>>(out, cyc, 1000, mode); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); k<<<256, 512>>>(out, cyc, iters, mode); (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(256); (void)hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    double c = 0; for (auto v : h) c += (double)v; c /= 256;
    const double flops = 256.0 * 8 * iters * 8 * 2.0 * 32 * 32 * 2;
    printf("%-18s %8.2f ms  %7.1f TFLOP/s  s_memtime ticks %.3e (%.1f MHz)  MFMA cycles / tick %.2f\n", names[mode], ms, flops / ms / 1e9, c,
           c / ms / 1e3, 2.0 * iters * 8 * 64 / c);
  }
  return 0;
}
