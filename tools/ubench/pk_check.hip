// Checks the semantics of the inline-asm packed fp32 helpers used by the Winograd kernels (v_pk_add_f32 with neg
// modifiers, v_pk_fma_f32, v_pk_mul_f32) against scalar arithmetic.   hipcc --offload-arch=gfx950 -O3 pk_check.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_add(f32x2 x, f32x2 y) { f32x2 d; asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y)); return d; }
__device__ __forceinline__ f32x2 pk_sub(f32x2 x, f32x2 y) { f32x2 d; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(x), "v"(y)); return d; }
__device__ __forceinline__ f32x2 pk_fma(f32x2 x, f32x2 y, f32x2 z) { f32x2 d; asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(x), "v"(y), "v"(z)); return d; }
__device__ __forceinline__ f32x2 pk_mul(f32x2 x, f32x2 y) { f32x2 d; asm("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y)); return d; }
__global__ void k(const float* in, float* out) {
  const int t = threadIdx.x;
  f32x2 a = {in[t * 6 + 0], in[t * 6 + 1]}, b = {in[t * 6 + 2], in[t * 6 + 3]}, c = {in[t * 6 + 4], in[t * 6 + 5]};
  f32x2 r0 = pk_add(a, b), r1 = pk_sub(a, b), r2 = pk_fma(a, b, c), r3 = pk_mul(a, b);
  float* o = out + t * 8;
  o[0] = r0[0]; o[1] = r0[1]; o[2] = r1[0]; o[3] = r1[1]; o[4] = r2[0]; o[5] = r2[1]; o[6] = r3[0]; o[7] = r3[1];
}
int main() {
  float h[64 * 6], o[64 * 8], *di, *dout;
  for (int i = 0; i < 64 * 6; ++i) h[i] = (float)((i * 37 % 101) - 50) / 7.f;
  hipMalloc(&di, sizeof(h)); hipMalloc(&dout, sizeof(o));
  hipMemcpy(di, h, sizeof(h), hipMemcpyHostToDevice);
  k<<<1, 64>>>(di, dout);
  hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int t = 0; t < 64; ++t) {
    const float* a = h + t * 6; const float* b = a + 2; const float* c = a + 4;
    float e[8] = {a[0] + b[0], a[1] + b[1], a[0] - b[0], a[1] - b[1], fmaf(a[0], b[0], c[0]), fmaf(a[1], b[1], c[1]), a[0] * b[0], a[1] * b[1]};
    for (int j = 0; j < 8; ++j) if (e[j] != o[t * 8 + j]) { if (bad < 8) printf("lane %d out %d: got %g want %g\n", t, j, o[t * 8 + j], e[j]); ++bad; }
  }
  printf("pk_check: %d mismatches\n", bad);
  return bad != 0;
}
