// Microbenchmark: do VALU / LDS instructions overlap with MFMAs of the same wave / the partner wave of the SIMD?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_overlap tools/ubench/mfma_overlap.hip && /tmp/mfma_overlap
// Fillers are inline-asm v_fma_f32 on private registers (8 independent chains), pinned between the MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define FILL1(R) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(R) : "v"(y), "v"(x));
#define LDSRD(R, OFF) asm volatile("ds_read_b128 %0, %1 offset:" #OFF : "=v"(R) : "v"(laddr));

// BF: 0 = v_mfma_f32_32x32x2_f32 (64 cycles), 1 = v_mfma_f32_32x32x16_bf16 (32 cycles); PER = fillers after EACH MFMA
// (PER < 0: -PER * 8 fillers grouped after every 8 MFMAs); KIND 0 = v_fma_f32, 1 = ds_read_b128
template <int BF, int PER, int KIND>
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters, int waves_active) {
  __shared__ float lds[4096];
  const int wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i;
  __syncthreads();
  if (wave >= waves_active) return;
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a8, b8;
  for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(threadIdx.x * 1e-3f + i); b8[i] = (__bf16)1.0f; }
  float x = threadIdx.x * 1e-3f, y = 1.0001f;
  float v0 = x, v1 = x + 1, v2 = x + 2, v3 = x + 3, v4 = x + 4, v5 = x + 5, v6 = x + 6, v7 = x + 7;
  f32x4 l0, l1, l2, l3;
  const unsigned laddr = (threadIdx.x & 63) * 16;
  l0 = l1 = l2 = l3 = f32x4{0.f, 0.f, 0.f, 0.f};
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      if (BF) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc[m & 3], 0, 0, 0);
      else acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[m & 3], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (PER > 0 && KIND == 0) {
        if (PER >= 1) FILL1(v0) if (PER >= 2) FILL1(v1) if (PER >= 3) FILL1(v2) if (PER >= 4) FILL1(v3)
        if (PER >= 5) FILL1(v4) if (PER >= 6) FILL1(v5) if (PER >= 7) FILL1(v6) if (PER >= 8) FILL1(v7)
        if (PER >= 9) FILL1(v0) if (PER >= 10) FILL1(v1) if (PER >= 11) FILL1(v2) if (PER >= 12) FILL1(v3)
        __builtin_amdgcn_sched_barrier(0);
      }
      if (PER > 0 && KIND == 1) {
        if (PER >= 1) LDSRD(l0, 0) if (PER >= 2) LDSRD(l1, 1024) if (PER >= 3) LDSRD(l2, 2048) if (PER >= 4) LDSRD(l3, 3072)
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (PER < 0) {
#pragma unroll
      for (int j = 0; j < -PER; ++j) { FILL1(v0) FILL1(v1) FILL1(v2) FILL1(v3) FILL1(v4) FILL1(v5) FILL1(v6) FILL1(v7) }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (KIND == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 + l0[0] + l1[1] + l2[2] + l3[3];
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][5];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int BF, int PER, int KIND>
void run(int waves_active) {
  float* out; long long* cyc;
  (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 256 * 8 * 8);
  const int iters = 2000;
  for (int rep = 0; rep < 2; ++rep) { k<BF, PER, KIND><<<256, 512>>>(out, cyc, iters, waves_active); (void)hipDeviceSynchronize(); }
  std::vector<long long> h(256 * 8);
  (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  double mx = 0;
  for (int i = 0; i < waves_active; ++i) mx = std::max(mx, (double)h[8 * 7 + i]);
  printf("%s MFMA x8 | %s per MFMA %3d %s | waves/SIMD %d : %7.1f cycles / 8 MFMAs\n", BF ? "bf16 32x32x16" : "f32  32x32x2 ",
         KIND ? "ds_read_b128" : "v_fma_f32   ", PER < 0 ? -PER : PER, PER < 0 ? "(grouped x8)" : "            ", waves_active / 4, mx / iters);
  (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
  run<0, 0, 0>(4); run<0, 0, 0>(8);
  run<0, 2, 0>(4); run<0, 4, 0>(4); run<0, 8, 0>(4); run<0, 12, 0>(4); run<0, -4, 0>(4); run<0, -8, 0>(4);
  run<0, 4, 0>(8); run<0, 8, 0>(8); run<0, -8, 0>(8);
  run<0, 2, 1>(4); run<0, 4, 1>(4); run<0, 4, 1>(8);
  run<1, 0, 0>(4); run<1, 0, 0>(8);
  run<1, 2, 0>(4); run<1, 4, 0>(4); run<1, 6, 0>(4); run<1, 8, 0>(4); run<1, -4, 0>(4);
  run<1, 4, 0>(8); run<1, 8, 0>(8);
  run<1, 2, 1>(4); run<1, 4, 1>(4); run<1, 4, 1>(8);
  return 0;
}
