// Bit-reproducible accumulation (opt-in: SSP_DETERMINISTIC=1 or ssp_set_deterministic(1) before ssp_bind).
//
// The path accumulates in two kinds of places whose order of commits varies from run to run:
//   (a) fp64 accumulators fed by floating-point atomics - BatchNorm statistics and backward sums (NREP replicas each), loss sums;
//   (b) fp32 tensors fed by fp32 atomics - descriptor / segmentation gradients scattered from sampled points, bias gradients, the
//       first layer's weight gradient.
// (a): every addend is first rounded to a multiple of a fixed quantum q = 2^-k.  Sums of multiples of q are exact in fp64 while
//      |sum| < 2^53 q, and exact additions commute: the accumulator no longer depends on the order.  Readers are untouched.  (Past
//      the bound the sum merely becomes order-dependent again; the quanta below leave 2^31 .. 2^10 of head room for their class.)
// (b): 24 bits are too few for that trick.  Registered fp32 targets get a 64-bit fixed-point shadow (2^-40 units): the atomics go
//      to the shadow as integer adds (associative), det_fold_kernel adds the shadow into the fp32 tensor and clears it.
// Default mode (flag 0): the plain atomics, one predictable branch per flush.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sspk {

__device__ int g_det;   // 0 = plain atomics (default)

constexpr int DET_MAX_REGIONS = 16;   // (five per bound handle with gradients)
struct DetRegion {
  const float* lo;
  const float* hi;
  long long* shadow;
};
__device__ DetRegion g_det_region[DET_MAX_REGIONS];
__device__ int g_det_nregion;

// quanta (as 2^k and 2^-k) per accumulator class
constexpr double DET_K_STATS = 4194304.0, DET_Q_STATS = 1.0 / 4194304.0;                       // 2^22: |sum| < 2.1e9
constexpr double DET_K_GRAD = 8796093022208.0, DET_Q_GRAD = 1.0 / 8796093022208.0;             // 2^43: |sum| < 1024
constexpr double DET_K_LOSS = 67108864.0, DET_Q_LOSS = 1.0 / 67108864.0;                      // 2^26: |sum| < 1.3e8
constexpr double DET_K_SHADOW = 1099511627776.0, DET_Q_SHADOW = 1.0 / 1099511627776.0;        // 2^40: |sum| < 8.4e6

__device__ __forceinline__ void acc_add_q(double* p, double v, double k, double q) {
  if (g_det) v = rint(v * k) * q;
  unsafeAtomicAdd(p, v);
}
__device__ __forceinline__ void acc_add_stats(double* p, double v) { acc_add_q(p, v, DET_K_STATS, DET_Q_STATS); }
__device__ __forceinline__ void acc_add_grad(double* p, double v) { acc_add_q(p, v, DET_K_GRAD, DET_Q_GRAD); }
__device__ __forceinline__ void acc_add_loss(double* p, double v) { acc_add_q(p, v, DET_K_LOSS, DET_Q_LOSS); }
// the convolution epilogues accumulate forward statistics (sums of activations) or, as a data gradient with a fused
// BatchNorm-backward reduction, sums of activation gradients through the same code: the quantum follows the use
__device__ __forceinline__ void acc_add_stats_or_grad(double* p, double v, bool grad) {
  if (grad) acc_add_grad(p, v); else acc_add_stats(p, v);
}

// fp32 target: plain atomic, or the fixed-point shadow of the registered region that holds p
__device__ __forceinline__ void facc_add(float* p, float v) {
  if (g_det) {
    const int n = g_det_nregion;
    for (int i = 0; i < n; ++i) {
      const DetRegion r = g_det_region[i];
      if (p >= r.lo && p < r.hi) {
        // (a non-finite contribution has no fixed-point image - __double2ll_rn would turn it into 0 or a saturated value and hide
        // a diverged gradient: it goes to the tensor itself, where the fold's addition keeps it)
        if (isfinite(v)) {
          atomicAdd(reinterpret_cast<unsigned long long*>(r.shadow + (p - r.lo)), (unsigned long long)__double2ll_rn((double)v * DET_K_SHADOW));
          return;
        }
        break;
      }
    }
  }
  atomicAdd(p, v);
}

// A scatter target resolved ONCE per thread (the kernels that scatter many values into one tensor: descriptor and segmentation
// gradients): the region lookup of facc_add per atomic was most of the mode's cost (desc_match_kernel 0.27 -> 1.2 ms).
struct DetTarget {
  float* lo;
  long long* shadow;   // nullptr: plain fp32 atomics
};
__device__ __forceinline__ DetTarget det_resolve(float* base) {
  DetTarget t = {base, nullptr};
  if (g_det && base != nullptr) {
    const int n = g_det_nregion;
    for (int i = 0; i < n; ++i) {
      const DetRegion r = g_det_region[i];
      if (base >= r.lo && base < r.hi) { t.lo = const_cast<float*>(r.lo); t.shadow = r.shadow; break; }
    }
  }
  return t;
}
__device__ __forceinline__ void facc_add(const DetTarget& t, float* p, float v) {
  if (t.shadow != nullptr && isfinite(v))   // (non-finite: see facc_add above)
    atomicAdd(reinterpret_cast<unsigned long long*>(t.shadow + (p - t.lo)), (unsigned long long)__double2ll_rn((double)v * DET_K_SHADOW));
  else
    atomicAdd(p, v);
}

// dst[i] += shadow[i] 2^-40; shadow[i] = 0   (one launch per registered region, after the kernels that scatter into it)
__global__ void det_fold_kernel(long long* __restrict__ shadow, float* __restrict__ dst, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long long s = shadow[i];
    if (s != 0) {
      dst[i] += (float)((double)s * DET_Q_SHADOW);
      shadow[i] = 0;
    }
  }
}

}  // namespace sspk
