// Microbenchmark: issue cost of the vector instructions sem_ce_kernel is made of, per wave instruction, with 1 / 2 / 4 waves per
// SIMD: v_fma_f32, v_pk_fma_f32, v_pk_mul_f32, v_exp_f32, v_add_f32 with a DPP row shift, v_fma_f32 with a scalar operand, and the
// mixes "2 fma + exp + add" (one class of the planned lanes = (x, class) form) and "4 pk + pk_add + 2 exp + pk_add" (pass A today).
// 8 independent chains per kind, 64 instructions per loop trip.  Reported: s_memtime ticks per instruction of the oldest and
// the youngest wave of workgroup 0 (the SIMD arbitrates oldest first), and wall time per instruction issued by a SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o ab/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int KIND>
__global__ __launch_bounds__(1024, 1) void rate_kernel(float* out, int iters, unsigned long long* cyc, float sc) {
  const int tid = threadIdx.x;
  float a[8];
  f32x2 p[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = 0.001f * (float)(tid + i);
    p[i] = f32x2{0.002f * (float)(tid + i), 0.003f * (float)(tid - i)};
  }
  const float m = 0.999f, q = 0.0001f;
  const f32x2 m2 = {0.999f, 0.998f}, q2 = {0.0001f, 0.0002f};
  float sacc = 0.f;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      if (KIND == 0) {
#define X(I) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[I]) : "v"(m), "v"(q));
        REP8(X)
#undef X
      } else if (KIND == 1) {
#define X(I) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[I]) : "v"(m2), "v"(q2));
        REP8(X)
#undef X
      } else if (KIND == 2) {
#define X(I) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[I]) : "v"(m2));
        REP8(X)
#undef X
      } else if (KIND == 3) {
#define X(I) asm volatile("v_exp_f32 %0, %0" : "+v"(a[I]));
        REP8(X)
#undef X
      } else if (KIND == 4) {
#define X(I) asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[I]));
        REP8(X)
#undef X
      } else if (KIND == 5) {
#define X(I) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[I]) : "s"(sc), "v"(q));
        REP8(X)
#undef X
      } else if (KIND == 6) {  // per element: 2 fma, exp, add  (8 instructions = 2 elements per X)
#define X(I) asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_exp_f32 %0, %0\n s_nop 0\n v_add_f32 %1, %1, %0" \
                          : "+v"(a[I]), "+v"(sacc) : "v"(m), "v"(q));
        REP8(X)
#undef X
      } else if (KIND == 7) {  // pass A today, per class pair: 4 pk, pk_add, 2 exp, nop, pk_add
#define X(I) asm volatile("v_pk_mul_f32 %0, %0, %2\n v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %0, %0, %2, %3\n" \
                          "v_pk_add_f32 %0, %0, %3\n v_exp_f32 %1, %1\n v_exp_f32 %1, %1\n s_nop 0\n v_pk_add_f32 %0, %0, %3" \
                          : "+v"(p[I]), "+v"(a[I]) : "v"(m2), "v"(q2));
        REP8(X)
#undef X
      } else if (KIND == 8) {
#define X(I) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[I]) : "v"(m));
        REP8(X)
#undef X
      } else if (KIND == 9) {
#define X(I) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[I]));
        REP8(X)
#undef X
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = sacc;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i] + p[i][0] + p[i][1];
  out[blockIdx.x * 1024 + tid] = s;
  if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
  if (tid == (int)blockDim.x - 64 && blockIdx.x == 0) cyc[1] = t1 - t0;   // the youngest wave of the workgroup
}

template <int KIND>
static void run(const char* name, int per_x, float* out, unsigned long long* cyc) {
  const int iters = 2000;
  for (int waves_per_simd = 1; waves_per_simd <= 4; waves_per_simd *= 2) {
    const int threads = 256 * waves_per_simd;
    hipLaunchKernelGGL((rate_kernel<KIND>), dim3(256), dim3(threads), 0, 0, out, iters, cyc, 0.999f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((rate_kernel<KIND>), dim3(256), dim3(threads), 0, 0, out, iters, cyc, 0.999f);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c[2] = {0, 0};
    hipMemcpy(c, cyc, sizeof(c), hipMemcpyDeviceToHost);
    const double n = (double)iters * 64 * per_x;
    printf("%-38s waves/SIMD %d: oldest wave %6.2f, youngest %6.2f cycles per instruction; %7.3f ns per instruction issued by a SIMD\n", name,
           waves_per_simd, (double)c[0] / n, (double)c[1] / n, (double)ms * 1e6 / (n * waves_per_simd));
  }
}

int main() {
  float* out;
  unsigned long long* cyc;
  hipMalloc(&out, 256 * 1024 * sizeof(float));
  hipMalloc(&cyc, 16);
  run<0>("v_fma_f32", 1, out, cyc);
  run<8>("v_mul_f32", 1, out, cyc);
  run<5>("v_fma_f32 (scalar operand)", 1, out, cyc);
  run<1>("v_pk_fma_f32", 1, out, cyc);
  run<2>("v_pk_mul_f32", 1, out, cyc);
  run<3>("v_exp_f32", 1, out, cyc);
  run<9>("v_rcp_f32", 1, out, cyc);
  run<4>("v_add_f32 dpp row_shr:1", 1, out, cyc);
  run<6>("2 fma + exp + nop + add (5 instr)", 5, out, cyc);
  run<7>("4 pk + pk_add + 2 exp + nop + pk_add", 9, out, cyc);
  return 0;
}
