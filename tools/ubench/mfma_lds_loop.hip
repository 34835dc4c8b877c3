// Microbenchmark of the conv_bf16_kernel inner loop: per k-step 2 weight fragments (lane-linear ds_read_b128) + 2 pixel fragments
// (ds_read_b128 at the halo addressing: slot stride 80 B, lane -> pixel mapping of cb_lane_pixel) feeding 4 v_mfma_f32_32x32x16_bf16.
// Variants: 0 = MFMAs only; 1 = + linear reads for all four fragments; 2 = the kernel's addressing; 3 = 2 with 8 waves per CU.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_lds_loop.hip -o ab/mfma_lds_loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void lane_pixel(int j, int& r, int& c) {
  int k;
  if (j < 4) { r = 0; k = j; } else if (j < 12) { r = 0; k = j + 4; } else if (j < 16) { r = 0; k = j - 8; }
  else if (j < 20) { r = 1; k = j - 16; } else if (j < 28) { r = 1; k = j - 12; } else { r = 1; k = j - 24; }
  c = (k - 2 * r) & 15;
}

template <int VAR>
__global__ __launch_bounds__(256, 2) void loop_kernel(float* out, int iters, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lj = lane & 31, lg = lane >> 5;
  for (int i = tid; i < 62784 / 4; i += 256) reinterpret_cast<uint32_t*>(sm)[i] = 0x3c003c00u + i;
  __syncthreads();
  unsigned char* sH = sm;
  unsigned char* sW = sm + 25920;
  int pr, pc;
  lane_pixel(lj, pr, pc);
  int boff[2];
  for (int nt = 0; nt < 2; ++nt) boff[nt] = VAR == 1 ? (wave * 2 + nt) * 1024 + lane * 16 : ((4 * wave + 2 * nt + pr) * 18 + pc) * 80 + lg * 16;
  const int aoff = lane * 16;
  f32x16 acc[2][2] = {};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    s16x8 fa[2][2], fb[2][2];
    auto fetch = [&](int step, s16x8 (&qa)[2], s16x8 (&qb)[2]) {
      const int tap = step >> 1, ks = step & 1, dy = tap / 3, dx = tap % 3;
      if (VAR == 0) { qa[0] = qa[1] = qb[0] = qb[1] = s16x8{1, 2, 3, 4, 5, 6, 7, (short)step}; return; }
      qa[0] = *reinterpret_cast<const s16x8*>(sW + aoff + ((tap * 2 + ks) * 2 + 0) * 1024);
      qa[1] = *reinterpret_cast<const s16x8*>(sW + aoff + ((tap * 2 + ks) * 2 + 1) * 1024);
      const int toff = VAR == 1 ? (step & 1) * 8192 : (dy * 18 + dx) * 80 + ks * 32;
      qb[0] = *reinterpret_cast<const s16x8*>(sH + boff[0] + toff);
      qb[1] = *reinterpret_cast<const s16x8*>(sH + boff[1] + toff);
    };
    fetch(0, fa[0], fb[0]);
#pragma unroll
    for (int step = 0; step < 18; ++step) {
      const int cur = step & 1;
      if (step + 1 < 18) fetch(step + 1, fa[cur ^ 1], fb[cur ^ 1]);
      __builtin_amdgcn_sched_barrier(0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][0], fb[cur][0], acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][0], fb[cur][1], acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][1], fb[cur][0], acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][1], fb[cur][1], acc[1][1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[blockIdx.x * 256 + tid] = s;
  if (tid == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int VAR>
static void run(const char* name, int grid) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 8);
  auto k = loop_kernel<VAR>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 62784);
  const int iters = 2000;
  hipLaunchKernelGGL(k, dim3(grid), dim3(256), 62784, 0, out, 10, cyc);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k, dim3(grid), dim3(256), 62784, 0, out, iters, cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double mfma = (double)grid * 4 * iters * 72;
  printf("%-38s grid %4d: %.3f ms, %.1f TF/s, %.1f wave-clock cycles per MFMA per wave\n", name, grid, ms, mfma * 32768.0 / ms / 1e9, (double)c / (iters * 72.0));
}

// sustained form: `mfma_lds_loop <variant 0|1|2> <seconds>` keeps launching one variant (grid 256) so that rocm-smi can sample the
// socket power / shader clock of "MFMAs alone" against "MFMAs + their LDS operand reads" (tools/dbg/power_ubench.sh)
template <int VAR>
static void sustain(double seconds) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 8);
  auto k = loop_kernel<VAR>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 62784);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  double total_ms = 0; long launches = 0;
  while (total_ms < seconds * 1e3) {
    hipEventRecord(e0);
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k, dim3(256), dim3(256), 62784, 0, out, 2000, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    total_ms += ms; launches += 50;
  }
  const double mfma = 256.0 * 4 * 2000 * 72 * launches;
  printf("variant %d sustained %.1f s: %.1f TF/s\n", VAR, total_ms / 1e3, mfma * 32768.0 / total_ms / 1e9);
}

int main(int argc, char** argv) {
  if (argc >= 3) {
    const int v = atoi(argv[1]); const double sec = atof(argv[2]);
    if (v == 0) sustain<0>(sec); else if (v == 1) sustain<1>(sec); else sustain<2>(sec);
    return 0;
  }
  run<0>("MFMAs only (register operands)", 256);
  run<0>("MFMAs only (register operands)", 512);
  run<1>("+ 4 linear ds_read_b128 per step", 256);
  run<1>("+ 4 linear ds_read_b128 per step", 512);
  run<2>("kernel addressing (halo, stride 80 B)", 256);
  run<2>("kernel addressing (halo, stride 80 B)", 512);
  return 0;
}
