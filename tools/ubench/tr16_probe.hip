// Probe of ds_read_b64_tr_b16 (gfx950): which LDS element lands in which (lane, element) slot, and a check of the
// v_mfma_f32_32x32x16_bf16 operand / result layouts used by csrc/conv_bf16.hip.h.
// Build: hipcc --offload-arch=gfx950 -O2 tools/ubench/tr16_probe.hip -o gpurun_out/tr16_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

__global__ void tr_probe(uint16_t* out, int stride_bytes) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[8192];
  const int l = threadIdx.x;
  for (int i = l; i < 8192; i += 64) lds[i] = (uint16_t)i;  // element id = its index
  __syncthreads();
  // lane address: 16-lane group q = l>>4, lane i = l&15: row (i>>2) of a [4][16] block, 4 contiguous elements at col 4*(i&3)
  const int i = l & 15, q = l >> 4;
  // LDS byte offset of the array (address space 3 pointer value); deriving the address from it keeps the stores alive
  const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint16_t*)lds;
  const uint32_t addr = base + (uint32_t)((i >> 2) * stride_bytes + (i & 3) * 8 + q * 32);  // bytes; group q = columns 16q..16q+15
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  out[l * 4 + 0] = (uint16_t)(v[0] & 0xffff);
  out[l * 4 + 1] = (uint16_t)(v[0] >> 16);
  out[l * 4 + 2] = (uint16_t)(v[1] & 0xffff);
  out[l * 4 + 3] = (uint16_t)(v[1] >> 16);
}

static inline uint16_t f2bf(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  u += 0x7fff + ((u >> 16) & 1);
  return (uint16_t)(u >> 16);
}

// D[i][j] = sum_k A[i][k] B[k][j], A 32x16, B 16x32 (row-major in global); operand layout under test:
// lane l: A row i = l & 31, k = 8 (l >> 5) .. + 7; B col j = l & 31, same k; D: col = l & 31, row = (r & 3) + 8 (r >> 2) + 4 (l >> 5)
__global__ void mfma_probe(const uint16_t* A, const uint16_t* B, float* D) {
  const int l = threadIdx.x, i = l & 31, g = l >> 5;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) {
    a[e] = (short)A[i * 16 + 8 * g + e];
    b[e] = (short)B[(8 * g + e) * 32 + i];
  }
  f32x16 c = {};
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * g) * 32 + i] = c[r];
}

int main() {
  uint16_t* d_out;
  hipMalloc(&d_out, 64 * 4 * 2);
  for (int stride : {32, 64, 144}) {
    hipLaunchKernelGGL(tr_probe, dim3(1), dim3(64), 0, 0, d_out, stride);
    std::vector<uint16_t> h(256);
    hipMemcpy(h.data(), d_out, 512, hipMemcpyDeviceToHost);
    printf("stride %d bytes (%d elements per row): lane -> 4 element ids\n", stride, stride / 2);
    for (int l = 0; l < 64; ++l) {
      printf("  l%02d:", l);
      for (int e = 0; e < 4; ++e) {
        const int id = h[l * 4 + e];
        printf(" (r%d,c%d)", id / (stride / 2), id % (stride / 2));
      }
      if ((l & 3) == 3) printf("\n");
    }
  }
  // MFMA layout check
  std::vector<uint16_t> A(32 * 16), B(16 * 32);
  std::vector<float> Af(32 * 16), Bf(16 * 32), ref(32 * 32, 0.f), D(32 * 32);
  for (int i = 0; i < 32 * 16; ++i) { Af[i] = (float)((i * 7) % 13 - 6); A[i] = f2bf(Af[i]); }
  for (int i = 0; i < 16 * 32; ++i) { Bf[i] = (float)((i * 5) % 11 - 5) * 0.5f; B[i] = f2bf(Bf[i]); }
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) for (int k = 0; k < 16; ++k) ref[i * 32 + j] += Af[i * 16 + k] * Bf[k * 32 + j];
  uint16_t *dA, *dB; float* dD;
  hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dD, D.size() * 4);
  hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(mfma_probe, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
  double err = 0;
  for (int i = 0; i < 1024; ++i) err = fmax(err, fabs(D[i] - ref[i]));
  printf("mfma_f32_32x32x16_bf16 layout check: max abs err %.3g (%s)\n", err, err < 1e-3 ? "OK" : "MISMATCH");
  return 0;
}
