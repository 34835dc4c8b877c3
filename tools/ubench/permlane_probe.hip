// Lane map of v_permlane32_swap / v_permlane16_swap (gfx950) as the builtins return them: prints, for each result, which
// (operand, lane) every lane received.  build: hipcc --offload-arch=gfx950 -O3 tools/ubench/permlane_probe.hip -o ab/permlane_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* o) {
  const unsigned a = threadIdx.x, b = 100 + threadIdx.x;
  const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  const auto q = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  o[threadIdx.x] = r[0]; o[64 + threadIdx.x] = r[1]; o[128 + threadIdx.x] = q[0]; o[192 + threadIdx.x] = q[1];
}
int main() {
  unsigned *d, h[256];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[4] = {"permlane32_swap [0]", "permlane32_swap [1]", "permlane16_swap [0]", "permlane16_swap [1]"};
  for (int r = 0; r < 4; ++r) {
    printf("%s (a = lane, b = 100 + lane):", names[r]);
    for (int l = 0; l < 64; l += 8) printf(" %u", h[r * 64 + l]);
    printf("\n");
  }
  return 0;
}
