// Stand-alone timing of the segmentation-loss kernels (semantic-superpoint_amd/csrc/sem_kernels.hip.h) at the training shape:
// 32 x 240x320 pixels, 133 classes, channel stride 136.  Builds in seconds, so that ablations (-DSEMX_ABL=n) and variants can be
// compared in one GPU call:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -I semantic-superpoint_amd/csrc tools/ubench/sem_ce_bench.hip -o ab/sem_ce_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "pk_math.hip.h"
#include "sem_kernels.hip.h"
using namespace sspk;

__global__ void prep(StepAccum* acc, double cnt) {
  acc->sem_sum[0] = 0.0;
  acc->sem_cnt[0] = cnt;
  acc->coef_sem = 1.f;
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 20;
  const int B = 32, Hc = 30, Wc = 40, H = 240, W = 320, C = 133, cs = 136;
  const size_t ncell = (size_t)B * Hc * Wc, npx = (size_t)B * H * W;
  std::vector<float> hs(ncell * cs, 0.f);
  std::vector<int64_t> hl(npx);
  unsigned long long r = 12345;
  auto rnd = [&]() { r = r * 6364136223846793005ull + 1442695040888963407ull; return (unsigned)(r >> 33); };
  for (size_t i = 0; i < ncell; ++i)
    for (int c = 0; c < C; ++c) hs[i * cs + c] = ((float)(rnd() % 20001) / 10000.f - 1.f) * 5.f;
  for (int kind = 0; kind < 2; ++kind) {
    for (int n = 0; n < B; ++n)
      for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x)
          hl[((size_t)n * H + y) * W + x] = kind == 0 ? (int64_t)(((n * 7 + (y / 24) * 14 + x / 24) * 2654435761u >> 8) % C) : (int64_t)(rnd() % C);
    float *sout, *dsout;
    int64_t* lab;
    StepAccum* acc;
    hipMalloc(&sout, hs.size() * 4);
    hipMalloc(&dsout, hs.size() * 4);
    hipMalloc(&lab, npx * 8);
    hipMalloc(&acc, sizeof(StepAccum));
    hipMemcpy(sout, hs.data(), hs.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(lab, hl.data(), npx * 8, hipMemcpyHostToDevice);
    const long ntile = (long)B * (Hc + 1) * (Wc + 1);
    for (int variant = 0; variant < 4; ++variant) {  // 0: old train, 1: old forward, 2: xc train, 3: xc forward
      auto launch = [&]() {
        if (variant == 0) hipLaunchKernelGGL((sem_ce_kernel<3>), dim3(4096), dim3(256), 0, 0, sout, lab, dsout, acc, 0, B, Hc, Wc, H, W, C, cs);
        if (variant == 1) hipLaunchKernelGGL((sem_ce_kernel<1>), dim3(4096), dim3(256), 0, 0, sout, lab, (float*)nullptr, acc, 0, B, Hc, Wc, H, W, C, cs);
        const SemXcGeom geom = sem_xc_geom(B, Hc, Wc, 512);
        const int grid = B * geom.row_groups * geom.x_splits;
        if (variant == 2) hipLaunchKernelGGL((sem_ce_xc_kernel<3, 9>), dim3(grid), dim3(256), 0, 0, sout, lab, dsout, acc, 0, B, Hc, Wc, C, cs, geom);
        if (variant == 3) hipLaunchKernelGGL((sem_ce_xc_kernel<1, 9>), dim3(grid), dim3(256), 0, 0, sout, lab, (float*)nullptr, acc, 0, B, Hc, Wc, C, cs, geom);
      };
      hipMemset(dsout, 0, hs.size() * 4);
      hipLaunchKernelGGL(prep, dim3(1), dim3(1), 0, 0, acc, (double)npx);
      launch();
      hipDeviceSynchronize();
      StepAccum ha;
      hipMemcpy(&ha, acc, sizeof(ha), hipMemcpyDeviceToHost);
      std::vector<float> hd(hs.size());
      hipMemcpy(hd.data(), dsout, hs.size() * 4, hipMemcpyDeviceToHost);
      double sabs = 0;
      for (float v : hd) sabs += v < 0 ? -v : v;
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      hipEventRecord(e0, 0);
      for (int i = 0; i < reps; ++i) launch();
      hipEventRecord(e1, 0);
      hipDeviceSynchronize();
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      const char* names[4] = {"pixels-then-classes train", "pixels-then-classes forward", "(x, class) lanes train", "(x, class) lanes forward"};
      printf("%-8s %-28s %8.1f us per launch   loss %.6f  sum|d| %.6f   (%ld tiles)\n", kind ? "noise" : "segments", names[variant],
             ms * 1e3 / reps, ha.sem_sum[0] / (double)npx, sabs, ntile);
    }
    hipFree(sout); hipFree(dsout); hipFree(lab); hipFree(acc);
  }
  return 0;
}
