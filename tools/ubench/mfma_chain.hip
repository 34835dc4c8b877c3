// Microbenchmark: fp32 MFMA throughput (wall clock, whole chip) as a function of the number of INDEPENDENT accumulators a
// wave cycles through and of the waves per SIMD: how long is the accumulate-to-accumulate dependency of
// v_mfma_f32_32x32x2_f32?   hipcc --offload-arch=gfx950 -O3 -o mfma_chain tools/ubench/mfma_chain.hip && ./mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(512) void k(float* out, int iters, int waves_active) {
  if ((int)(threadIdx.x >> 6) >= waves_active) return;
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const float a = threadIdx.x * 1e-3f, b = 0.25f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 16; ++m) acc[m % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m % NACC], 0, 0, 0);
  }
  float r = 0.f;
  for (int i = 0; i < NACC; ++i) r += acc[i][0] + acc[i][9];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int NACC>
void run(float* out, int waves) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 20000;
  k<NACC><<<256, 512>>>(out, 200, waves); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); k<NACC><<<256, 512>>>(out, iters, waves); (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double n = 256.0 * waves * iters * 16;
  printf("accumulators %d  waves/SIMD %d : %7.1f TFLOP/s   %.1f ns per MFMA per SIMD\n", NACC, waves / 4, n * 4096 / ms / 1e9,
         ms * 1e6 / (n / 1024));
}
int main() {
  float* out; (void)hipMalloc(&out, 256 * 512 * 4);
  run<1>(out, 4); run<2>(out, 4); run<4>(out, 4); run<8>(out, 4);
  run<1>(out, 8); run<2>(out, 8); run<4>(out, 8); run<8>(out, 8);
  return 0;
}
