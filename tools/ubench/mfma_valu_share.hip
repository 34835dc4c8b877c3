// Microbenchmark: how a SIMD shares its issue between an MFMA wave and VALU waves (the consumer / producer split of
// conv_bf16_ws_kernel).  One workgroup per CU: waves 0-3 run back-to-back v_mfma_f32_32x32x16_bf16 (4 independent accumulators),
// the next NV waves run the BatchNorm + ReLU staging arithmetic of the producers on registers (per 8 values: 4 shl, 4 and,
// 4 v_pk_fma_f32, 4 v_cvt_pk_bf16_f32, 4 v_pk_max_i16).  Reported: cycles per MFMA of wave 0 and cycles per VALU instruction of
// the first VALU wave, alone and together.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_valu_share.hip -o ab/mfma_valu_share
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// mode bit 0: MFMA waves active, bit 1: VALU waves active
__global__ __launch_bounds__(1024, 1) void share_kernel(float* out, int iters, int viters, int mode, unsigned long long* cyc) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (wave < 4) {
    if (!(mode & 1)) return;
    f32x16 acc[4] = {};
    s16x8 fa = {1, 2, 3, 4, 5, 6, 7, (short)lane}, fb = {2, 3, 4, 5, 6, 7, 8, (short)tid};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 18; ++k) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[3], 0, 0, 0);
      }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 1024 + tid] = s;
    if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    return;
  }
  if (!(mode & 2)) return;
  // VALU waves: 6 items of 4 words each per "stage" (as stage_halo), values kept live through a running xor
  uint32_t w[6][4];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) w[i][e] = 0x3f803f80u + tid * 77 + i * 4 + e;
  f32x2 sc = {1.0001f, 0.9999f}, sh = {0.001f, -0.001f};
  uint32_t keep = 0;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < viters; ++it) {
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const f32x2 x = {__builtin_bit_cast(float, w[i][e] << 16), __builtin_bit_cast(float, w[i][e] & 0xffff0000u)};
        f32x2 z;
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(z) : "v"(x), "v"(sc), "v"(sh));
        uint32_t p;
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p) : "v"(z[0]), "v"(z[1]));
        uint32_t q;
        asm volatile("v_pk_max_i16 %0, %1, 0" : "=v"(q) : "v"(p));
        w[i][e] = q + it;   // (one more add: 6 instructions per word)
        keep ^= q;
      }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 1024 + tid] = __builtin_bit_cast(float, keep);
  if (tid == 256 && blockIdx.x == 0) cyc[1] = t1 - t0;
}

static void run(const char* name, int nv, int mode, int vmul = 1) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 16); hipMemset(cyc, 0, 16);
  const int iters = 2000, threads = 256 + 64 * nv;
  hipLaunchKernelGGL(share_kernel, dim3(256), dim3(threads), 0, 0, out, 10, 10, mode, cyc);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(share_kernel, dim3(256), dim3(threads), 0, 0, out, iters, iters * vmul, mode, cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c[2]; hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost);
  // VALU instructions per iteration per wave: 24 words x 7 (shl, and, pk_fma, cvt, max, add, xor)
  printf("%-44s %d VALU waves: %.3f ms; %.1f cycles per MFMA; %.2f cycles per VALU instruction (VALU wave)\n", name, nv, ms,
         c[0] / (iters * 72.0), c[1] / (iters * vmul * 24.0 * 7.0));
  hipFree(out); hipFree(cyc);
}

int main() {
  run("MFMA waves alone", 4, 1);
  run("VALU waves alone", 4, 2);
  run("VALU waves alone", 8, 2);
  run("VALU waves alone", 12, 2);
  // (the VALU figure is valid when the MFMA waves outlast the VALU waves: x1; the MFMA figure when the VALU waves outlast them: x8)
  run("MFMA + VALU (VALU x1: read VALU)", 4, 3, 1);
  run("MFMA + VALU (VALU x8: read MFMA)", 4, 3, 8);
  run("MFMA + VALU (VALU x1: read VALU)", 8, 3, 1);
  run("MFMA + VALU (VALU x8: read MFMA)", 8, 3, 8);
  run("MFMA + VALU (VALU x1: read VALU)", 12, 3, 1);
  run("MFMA + VALU (VALU x8: read MFMA)", 12, 3, 8);
  return 0;
}
