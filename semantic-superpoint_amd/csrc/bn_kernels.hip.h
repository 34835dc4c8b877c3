// HBM-bound kernels around the convolutions: first-layer direct conv (1->64), BatchNorm statistics
// finalisation, BatchNorm(+ReLU(+max-pool)) backward, head epilogues.  NHWC fp32, float4 per lane.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "conv_mfma.hip.h"

namespace sspk {

// ---- 4 consecutive channels of an NHWC tensor of fp32 (T = float) or bf16 (T = uint16_t: the bf16 path) elements ----
template <typename T>
__device__ __forceinline__ float4 ld4(const T* p) {
  if constexpr (sizeof(T) == 4) {
    return *reinterpret_cast<const float4*>(p);
  } else {
    const u32x2 v = *reinterpret_cast<const u32x2*>(p);
    return make_float4(bf16_lo(v[0]), bf16_hi(v[0]), bf16_lo(v[1]), bf16_hi(v[1]));
  }
}
template <typename T>
__device__ __forceinline__ void st4(T* p, float4 v) {
  if constexpr (sizeof(T) == 4) {
    *reinterpret_cast<float4*>(p) = v;
  } else {
    u32x2 o;
    o[0] = pack_bf16(v.x, v.y);
    o[1] = pack_bf16(v.z, v.w);
    *reinterpret_cast<u32x2*>(p) = o;
  }
}

// ---- block reduction helper: per-thread `NV` floats, threads with the same (tid % nq) are summed ----
// smem must hold blockDim.x * NV floats.  Result valid for tid < nq (returned in v[]).
template <int NV>
__device__ __forceinline__ void reduce_by_column(float (&v)[NV], float* smem, int nq) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < NV; ++i) smem[tid * NV + i] = v[i];
  __syncthreads();
  if (tid < nq) {
    const int rows = blockDim.x / nq;
    for (int r = 1; r < rows; ++r)
#pragma unroll
      for (int i = 0; i < NV; ++i) v[i] += smem[(r * nq + tid) * NV + i];
  }
}

// ------------------------------------------------------------------------------------------------
// First layer: Conv2d(1, 64, 3, padding=1) on the grayscale image (models/unet_parts.py:14, in_ch = 1).
// K = 9: HBM-bound (writes 256 B per pixel).  16 lanes per pixel, float4 of channels per lane.
// Also accumulates the BatchNorm sums.
// ------------------------------------------------------------------------------------------------
// Row-based work split (a block owns image rows, its 16 pixel lanes walk along x): no per-pixel integer division,
// the row tests are wave-uniform.  grid: l0_grid(N * H), block 256.
constexpr int L0_UNROLL = 4;        // pixels in flight per thread, pass 1
constexpr int L0_UNROLL_APPLY = 2;  // pass 2
static inline int l0_grid(long rows) {
  const long per = (rows + 1023) / 1024;
  return (int)((rows + per - 1) / per);
}

// the 3x3 neighbourhood of pixel (row, ox) of a one-channel image, zero-padded.  Every load is unconditional from
// a clamped (always valid) address and zeroed afterwards: `cond ? load : 0` compiles to one exec-masked branch and
// one s_waitcnt per tap, which serialises the nine latencies (measured: 306 us vs 150 us per view for pass 2).
__device__ __forceinline__ void l0_taps(const float* __restrict__ xr, int ox, bool up, bool down, int W, float (&xv)[9]) {
  const bool l = ox > 0, r = ox < W - 1;
  const int xl = l ? ox - 1 : ox, xq = r ? ox + 1 : ox;
  const int ru = up ? -W : 0, rd = down ? W : 0;
  const float t0 = xr[ru + xl], t1 = xr[ru + ox], t2 = xr[ru + xq];
  const float t3 = xr[xl], t4 = xr[ox], t5 = xr[xq];
  const float t6 = xr[rd + xl], t7 = xr[rd + ox], t8 = xr[rd + xq];
  xv[0] = (up && l) ? t0 : 0.f;
  xv[1] = up ? t1 : 0.f;
  xv[2] = (up && r) ? t2 : 0.f;
  xv[3] = l ? t3 : 0.f;
  xv[4] = t4;
  xv[5] = r ? t5 : 0.f;
  xv[6] = (down && l) ? t6 : 0.f;
  xv[7] = down ? t7 : 0.f;
  xv[8] = (down && r) ? t8 : 0.f;
}

// The same neighbourhood from an LDS image of the three image rows around `row` ([3][W + 2] floats, zero borders and
// zero rows outside the image): 9 LDS reads at base + constant offsets instead of 9 clamped global loads and 9 selects.
// l0_stage_rows fills the image of one row (all 256 threads); the caller alternates two images, so ONE barrier per row
// (after the fill) is enough: a wave that fills image b again has passed the barrier of image b ^ 1, i.e. every wave has
// finished reading b.
constexpr int L0_MAX_W = 2046;  // 2 x 3 x (W + 2) floats of dynamic LDS <= 48 KB
static inline size_t l0_lds_bytes(int W) { return (size_t)2 * 3 * (W + 2) * sizeof(float); }
__device__ __forceinline__ void l0_stage_rows(float* __restrict__ img, const float* __restrict__ xr, bool up, bool down, int W) {
  const int Wp = W + 2;
  for (int c = threadIdx.x; c < Wp; c += 256) {
    const bool in = c >= 1 && c <= W;
    const int cc = in ? c - 1 : 0;
    const float t0 = xr[(up ? -W : 0) + cc], t1 = xr[cc], t2 = xr[(down ? W : 0) + cc];
    img[c] = (in && up) ? t0 : 0.f;
    img[Wp + c] = in ? t1 : 0.f;
    img[2 * Wp + c] = (in && down) ? t2 : 0.f;
  }
}
__device__ __forceinline__ void l0_taps_lds(const float* __restrict__ img, int ox, int W, float (&xv)[9]) {
  const int Wp = W + 2;
  const float* p = img + ox;  // column ox - 1 of the image = LDS column ox
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) xv[r * 3 + c] = p[r * Wp + c];
}

// four output channels of the first layer in registers (two channel PAIRS: every fma of the layer is one half of a
// v_pk_fma_f32, the tap value broadcast to both halves); y4() is THE fma chain of the layer (forward and the
// recomputation in its BatchNorm backward must agree bit for bit, or the ReLU mask would differ)
struct L0Taps {
  f32x2 p[5];  // taps 0..8 in pairs (2k, 2k + 1); p[4].y unused
  __device__ __forceinline__ void set(const float (&xv)[9]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) p[k] = f32x2{xv[2 * k], xv[2 * k + 1]};
    p[4] = f32x2{xv[8], 0.f};
  }
};
struct L0Conv {
  f32x2 w[2][9], b[2];  // [channel pair][tap]
  __device__ __forceinline__ void load(const float* __restrict__ wt, const float* __restrict__ bias, int c0) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      b[h] = f32x2{bias[c0 + 2 * h], bias[c0 + 2 * h + 1]};
#pragma unroll
      for (int t = 0; t < 9; ++t) w[h][t] = f32x2{wt[(c0 + 2 * h) * 9 + t], wt[(c0 + 2 * h + 1) * 9 + t]};
    }
  }
  // y[c0 + 2h], y[c0 + 2h + 1] = bias + sum_t x[t] * w[.][t], taps in ascending order
  __device__ __forceinline__ f32x2 y2(int h, const L0Taps& x) const {
    f32x2 o = b[h];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      o = pk_fma_bcast<false>(x.p[k], w[h][2 * k], o);
      o = pk_fma_bcast<true>(x.p[k], w[h][2 * k + 1], o);
    }
    return pk_fma_bcast<false>(x.p[4], w[h][8], o);
  }
};

template <typename TO = float>  // TO = uint16_t: bf16 output (conv algorithm 12); the statistics are those of the STORED values
__global__ __launch_bounds__(256) void conv0_direct_kernel(const float* __restrict__ x0, const float* __restrict__ x1,
                                                           const float* __restrict__ w, const float* __restrict__ bias,
                                                           TO* __restrict__ out0, TO* __restrict__ out1,
                                                           double* __restrict__ stats0, double* __restrict__ stats1, int N,
                                                           int H, int W) {
  // blockIdx.y = view of the pair
  const float* __restrict__ x = blockIdx.y ? x1 : x0;
  TO* __restrict__ out = blockIdx.y ? out1 : out0;
  double* __restrict__ stats = blockIdx.y ? stats1 : stats0;
  __shared__ float red[256 * 8];
  const int tid = threadIdx.x;
  const int q = tid & 15;      // channel quad
  const int pl = tid >> 4;     // pixel lane 0..15
  L0Conv cv;
  cv.load(w, bias, q * 4);
  f32x2 s01 = {0.f, 0.f}, s23 = s01, q01 = s01, q23 = s01;  // BatchNorm sums / sums of squares of the channel pairs
  const int rows = N * H;
  extern __shared__ float l0_rows[];
  int par = 0;
  for (int row = blockIdx.x; row < rows; row += gridDim.x, par ^= 1) {
    const int oy = row % H;
    const bool up = oy > 0, down = oy < H - 1;
    const float* xr = x + (size_t)row * W;
    float* const img = l0_rows + par * 3 * (W + 2);
    l0_stage_rows(img, xr, up, down, W);
    __syncthreads();
    TO* orow = out + (size_t)row * W * 64 + q * 4;
    for (int ox = pl; ox < W; ox += 16) {
      float v[9];
      l0_taps_lds(img, ox, W, v);
      L0Taps xt;
      xt.set(v);
      f32x2 o01 = cv.y2(0, xt), o23 = cv.y2(1, xt);
      if constexpr (sizeof(TO) == 4) {
        *reinterpret_cast<f32x4*>(orow + (size_t)ox * 64) = cat2(o01, o23);
      } else {
        u32x2 pk;
        pk[0] = pack_bf16(o01[0], o01[1]);
        pk[1] = pack_bf16(o23[0], o23[1]);
        *reinterpret_cast<u32x2*>(orow + (size_t)ox * 64) = pk;
        o01 = f32x2{bf16_lo(pk[0]), bf16_hi(pk[0])};
        o23 = f32x2{bf16_lo(pk[1]), bf16_hi(pk[1])};
      }
      s01 = pk_add(s01, o01); s23 = pk_add(s23, o23);
      q01 = pk_fma(o01, o01, q01); q23 = pk_fma(o23, o23, q23);
    }
  }
  float acc[8] = {s01[0], s01[1], s23[0], s23[1], q01[0], q01[1], q23[0], q23[1]};
  if (stats != nullptr) {
    reduce_by_column<8>(acc, red, 16);
    if (tid < 16) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        double* st = stats + (size_t)(blockIdx.x % NREP) * 128;
        acc_add_stats(st + tid * 4 + c, (double)acc[c]);
        acc_add_stats(st + 64 + tid * 4 + c, (double)acc[4 + c]);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// BatchNorm statistics -> per-channel affine (nn.BatchNorm2d defaults eps 1e-5, momentum 0.1).
// train: batch statistics from the fp64 sums + running-stat update; eval: running statistics.
// ------------------------------------------------------------------------------------------------
struct BnLayer {
  const double* stats;  // [NREP][2C] sum, sumsq
  const float* gamma;
  const float* beta;
  float* running_mean;
  float* running_var;
  float* scale;   // gamma * invstd
  float* shift;   // beta - mean * scale
  float* mean;
  float* invstd;
  int C;
  double count;
};

// Sum of the NREP (= 32) replicas of a pair of fp64 accumulators, one replica per lane of a 32-lane group: one load
// latency + 5 shuffle steps instead of 32 dependent iterations (these one-block kernels sit on the critical path
// between two convolutions: 11.7 -> ~5 us each, 24 of them per step).
static_assert(NREP == 32, "replica reduction is written for 32 lanes per channel");
__device__ __forceinline__ void replica_sums(const double* __restrict__ sums, int C, int c, int r, double& s1, double& s2) {
  s1 = sums[(size_t)r * 2 * C + c];
  s2 = sums[(size_t)r * 2 * C + C + c];
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) {
    s1 += __shfl_xor(s1, o);
    s2 += __shfl_xor(s2, o);
  }
}

__device__ __forceinline__ void bn_finalize_body(const BnLayer& L0, const BnLayer& L1, int nviews, int train, int64_t* nbt) {
  // 32 lanes per channel (replica r each); lane 0 of the group writes.  grid: ceil(C * 32 / blockDim)
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if ((t >> 5) >= ((L0.C + 7) & ~7)) return;   // (whole 256-thread workgroups past the layer: a launch shared with a wider layer)
  const int c = min(t >> 5, L0.C - 1), r = t & 31;  // clamped: every lane takes part in the shuffles
  const bool writer = r == 0 && (t >> 5) < L0.C;
  for (int v = 0; v < nviews; ++v) {  // view 0 then view 1: the running statistics are updated in the reference's order
    const BnLayer& L = v ? L1 : L0;
    double mean, var;
    if (train) {
      double s1, s2;
      replica_sums(L.stats, L.C, c, r, s1, s2);
      mean = s1 / L.count;
      var = s2 / L.count - mean * mean;
      if (var < 0) var = 0;
      const double unbiased = L.count > 1 ? var * L.count / (L.count - 1) : var;
      if (writer) {
        L.running_mean[c] = (float)(0.9 * (double)L.running_mean[c] + 0.1 * mean);
        L.running_var[c] = (float)(0.9 * (double)L.running_var[c] + 0.1 * unbiased);
        if (c == 0 && nbt != nullptr) *nbt += 1;
      }
    } else {
      mean = L.running_mean[c];
      var = L.running_var[c];
    }
    if (writer) {
      const float invstd = (float)(1.0 / sqrt(var + 1e-5));
      const float sc = L.gamma[c] * invstd;
      L.mean[c] = (float)mean;
      L.invstd[c] = invstd;
      L.scale[c] = sc;
      L.shift[c] = L.beta[c] - (float)mean * sc;
    }
  }
}
__global__ void bn_finalize_kernel(const BnLayer L0, const BnLayer L1, int nviews, int train, int64_t* nbt) {
  bn_finalize_body(L0, L1, nviews, train, nbt);
}
// up to three layers whose statistics are complete at the same point of the forward pass (the pointwise heads of one grouped launch;
// the three 3x3 heads, which all read layer 7): blockIdx.y = layer
struct BnFinJobs {
  BnLayer L0[3], L1[3];
  int64_t* nbt[3];
  int n;
};
__global__ void bn_finalize_multi_kernel(const BnFinJobs J, int nviews, int train) {
  const int j = blockIdx.y;
  bn_finalize_body(J.L0[j], J.L1[j], nviews, train, J.nbt[j]);
}

// MaxPool2d(2)(ReLU(BN(y))) materialised once per pooled layer boundary (layers 1, 3, 5): the three consumers
// (forward conv, weight gradient of the next layer) then read a 4x smaller, already activated tensor through their
// prefetched raw-input path instead of staging 4 loads per pixel synchronously (85-97 TF -> ~125 TF on those
// launches).  y: [N,H,W,C] raw conv output; out: [N,H/2,W/2,C].  C % 4 == 0.
__global__ __launch_bounds__(256) void bn_relu_pool_kernel(const float* __restrict__ y0, const float* __restrict__ scale0,
                                                           const float* __restrict__ shift0, float* __restrict__ out0,
                                                           const float* __restrict__ y1, const float* __restrict__ scale1,
                                                           const float* __restrict__ shift1, float* __restrict__ out1,
                                                           int N, int H, int W, int C) {
  // blockIdx.y = view of the pair
  const float* __restrict__ y = blockIdx.y ? y1 : y0;
  const float* __restrict__ scale = blockIdx.y ? scale1 : scale0;
  const float* __restrict__ shift = blockIdx.y ? shift1 : shift0;
  float* __restrict__ out = blockIdx.y ? out1 : out0;
  const int nq = C >> 2;
  const long total = (long)N * (H / 2) * (W / 2) * nq;
  const int Wo = W / 2, Ho = H / 2;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int q = (int)(idx % nq);
    const long p = idx / nq;
    const int ox = (int)(p % Wo), oy = (int)((p / Wo) % Ho), n = (int)(p / ((long)Wo * Ho));
    const float4 sc = *reinterpret_cast<const float4*>(scale + q * 4);
    const float4 sh = *reinterpret_cast<const float4*>(shift + q * 4);
    const size_t base = ((size_t)(n * H + 2 * oy) * W + 2 * ox) * C + q * 4;
    const float4 a = xform<1>(*reinterpret_cast<const float4*>(y + base), sc, sh);
    const float4 b = xform<1>(*reinterpret_cast<const float4*>(y + base + C), sc, sh);
    const float4 c = xform<1>(*reinterpret_cast<const float4*>(y + base + (size_t)W * C), sc, sh);
    const float4 d = xform<1>(*reinterpret_cast<const float4*>(y + base + (size_t)W * C + C), sc, sh);
    *reinterpret_cast<float4*>(out + (size_t)p * C + q * 4) = max4(max4(a, b), max4(c, d));
  }
}

// ------------------------------------------------------------------------------------------------
// BatchNorm (+ReLU (+2x2 max-pool)) backward, two passes over (dOut, Y):
//   pass 1 (reduce): S1 = sum dZ, S2 = sum dZ * xhat            (fp64 atomics, [2C])
//   pass 2 (apply) : dY = gamma*invstd * (dZ - S1/n - xhat*S2/n)    and   db_conv += sum dY (~0)
// where z = y*scale+shift, dZ = dA * [z > 0] (RELU) and dA is dOut routed to the first arg-max of each
// 2x2 window (POOL; torch max_pool2d backward semantics).  dgamma = S2, dbeta = S1.
// Layout: Y [N,H,W,cs] (H,W = full resolution of this layer), dOut [N,H/2,W/2,dcs] when POOL.
// ------------------------------------------------------------------------------------------------
struct BnBwdArgs {
  const float* y;
  const float* dout;
  float* dy;
  const float* scale;
  const float* shift;
  const float* mean;
  const float* invstd;
  const float* gamma;
  double* sums;      // [NREP][2C] pass-1 accumulators
  const float* k12;  // [2C] S1/n, S2/n reduced over the replicas by bn_bwd_sums_kernel (pass 2 input)
  float* dbias;      // conv bias gradient (accumulated) or nullptr
  const float* x;      // layer 0 only: the one-channel input image (y0 is recomputed from it)
  const float* apool;  // pooled layers: maxpool(relu(bn(y))) materialised by the forward
  const float* beta;   // pooled layers (pass 1 from apool)
  int pool_fix;        // pass 1 came from the data-gradient conv's epilogue (pooled layer): channels with gamma == 0 get
                       // their S2 from a scan over Y in bn_bwd_sums_kernel (xhat is not recoverable from apool there)
  int N, H, W, C;
  int y_cs, y_co, d_cs, d_co, dy_cs, dy_co;
  double count;
};

// T = float, or uint16_t for the bf16 path (y, dout and dy are bf16 tensors behind the float* fields; offsets in elements)
template <bool RELU, bool POOL, bool APPLY, typename T = float>
__global__ __launch_bounds__(256) void bn_bwd_kernel(const BnBwdArgs a0, const BnBwdArgs a1, const BnBwdArgs b0, const BnBwdArgs b1) {
  // the two views of a pair ride one launch (blockIdx.y 0, 1), and so may a second layer of the same template form (2, 3: the two
  // pointwise heads' BatchNorm - a small launch costs ~13 us of the step whatever it does); callers with one layer pass it twice
  const BnBwdArgs& a = blockIdx.y == 0 ? a0 : blockIdx.y == 1 ? a1 : blockIdx.y == 2 ? b0 : b1;
  const T* const t_y = reinterpret_cast<const T*>(a.y);
  const T* const t_dout = reinterpret_cast<const T*>(a.dout);
  T* const t_dy = reinterpret_cast<T*>(a.dy);
  __shared__ float red[256 * 8];
  const int tid = threadIdx.x;
  const int nq = (a.C + 3) / 4;                 // channel quads
  const int rows = 256 / nq;                    // pixel lanes per block
  const int q = tid % nq, pl = tid / nq;
  const bool active = pl < rows;
  const int c0 = q * 4;
  const int Ho = POOL ? a.H / 2 : a.H, Wo = POOL ? a.W / 2 : a.W;  // resolution of dOut
  const long npix = (long)a.N * Ho * Wo;
  float4 sc = make_float4(0, 0, 0, 0), sh = sc, mu = sc, is = sc, k1 = sc, k2 = sc, gs = sc;
  bool cv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) cv[i] = (c0 + i) < a.C;
  auto ldp = [&](const float* p) {
    float4 v = make_float4(0, 0, 0, 0);
    if (cv[0]) v.x = p[c0];
    if (cv[1]) v.y = p[c0 + 1];
    if (cv[2]) v.z = p[c0 + 2];
    if (cv[3]) v.w = p[c0 + 3];
    return v;
  };
  if (active) {
    sc = ldp(a.scale);
    sh = ldp(a.shift);
    mu = ldp(a.mean);
    is = ldp(a.invstd);
    if (APPLY) {
      const float4 g = ldp(a.gamma);
      float s1[4], s2[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        s1[i] = cv[i] ? a.k12[c0 + i] : 0.f;
        s2[i] = cv[i] ? a.k12[a.C + c0 + i] : 0.f;
      }
      k1 = make_float4(s1[0], s1[1], s1[2], s1[3]);
      k2 = make_float4(s2[0], s2[1], s2[2], s2[3]);
      gs = make_float4(g.x * is.x, g.y * is.y, g.z * is.z, g.w * is.w);
    }
  }
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, shv[4] = {sh.x, sh.y, sh.z, sh.w};
  const float muv[4] = {mu.x, mu.y, mu.z, mu.w}, isv[4] = {is.x, is.y, is.z, is.w};
  const float k1v[4] = {k1.x, k1.y, k1.z, k1.w}, k2v[4] = {k2.x, k2.y, k2.z, k2.w};
  const float gsv[4] = {gs.x, gs.y, gs.z, gs.w};
  if (active) {
    for (long p = (long)blockIdx.x * rows + pl; p < npix; p += (long)gridDim.x * rows) {
      const int ox = (int)(p % Wo);
      const int oy = (int)((p / Wo) % Ho);
      const int n = (int)(p / ((long)Wo * Ho));
      const float4 d4 = ld4<T>(t_dout + (size_t)p * a.d_cs + a.d_co + c0);
      const float dv[4] = {d4.x, d4.y, d4.z, d4.w};
      if (!POOL) {
        const size_t yo = (size_t)p * a.y_cs + a.y_co + c0;
        const float4 y4 = ld4<T>(t_y + yo);
        const float yv[4] = {y4.x, y4.y, y4.z, y4.w};
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float z = fmaf(yv[i], scv[i], shv[i]);
          const float dz = (!RELU || z > 0.f) ? dv[i] : 0.f;
          const float xh = (yv[i] - muv[i]) * isv[i];
          if (!APPLY) {
            acc[i] += dz;
            acc[4 + i] += dz * xh;
          } else {
            o[i] = gsv[i] * (dz - k1v[i] - xh * k2v[i]);
            acc[i] += o[i];
          }
        }
        if (APPLY) st4<T>(t_dy + (size_t)p * a.dy_cs + a.dy_co + c0, make_float4(o[0], o[1], o[2], o[3]));
      } else {
        float yv[4][4], o[4][4];
        size_t yo[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int yy = 2 * oy + (k >> 1), xx = 2 * ox + (k & 1);
          yo[k] = ((size_t)(n * a.H + yy) * a.W + xx);
          const float4 y4 = ld4<T>(t_y + yo[k] * a.y_cs + a.y_co + c0);
          yv[k][0] = y4.x; yv[k][1] = y4.y; yv[k][2] = y4.z; yv[k][3] = y4.w;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          // first maximum of relu(z) in window scan order (torch: strictly greater replaces)
          float best = -1.f;
          int bk = 0;
          float zk[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            zk[k] = fmaf(yv[k][i], scv[i], shv[i]);
            const float av = fmaxf(zk[k], 0.f);
            if (av > best) {
              best = av;
              bk = k;
            }
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float dz = (k == bk && zk[k] > 0.f) ? dv[i] : 0.f;
            const float xh = (yv[k][i] - muv[i]) * isv[i];
            if (!APPLY) {
              acc[i] += dz;
              acc[4 + i] += dz * xh;
            } else {
              o[k][i] = gsv[i] * (dz - k1v[i] - xh * k2v[i]);
              acc[i] += o[k][i];
            }
          }
        }
        if (APPLY) {
#pragma unroll
          for (int k = 0; k < 4; ++k) st4<T>(t_dy + yo[k] * a.dy_cs + a.dy_co + c0, make_float4(o[k][0], o[k][1], o[k][2], o[k][3]));
        }
      }
    }
  }
  if (!active) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  }
  // reduce over the block's pixel lanes
  {
#pragma unroll
    for (int i = 0; i < 8; ++i) red[tid * 8 + i] = acc[i];
    __syncthreads();
    if (tid < nq) {
      for (int r = 1; r < rows; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] += red[(r * nq + tid) * 8 + i];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (!cv[i]) continue;
        if (!APPLY) {
          double* sm = a.sums + (size_t)(blockIdx.x % NREP) * 2 * a.C;
          acc_add_grad(sm + c0 + i, (double)acc[i]);
          acc_add_grad(sm + a.C + c0 + i, (double)acc[4 + i]);
        }
      }
    }
  }
}

// Pass 1 of a POOLED layer from the pooled activation the forward materialised (apool = maxpool(relu(z)), 1/4 of
// the pixels) instead of Y: the only window position with dZ != 0 is the arg-max, where z == apool > 0, so
// S1 = sum dOut*[apool > 0] and S2 = sum dOut*[apool > 0]*xhat with xhat = (z - beta)/gamma (z = gamma*xhat + beta).
// Reads 2 x 1/4 tensors instead of 1/4 + 1.  Channels with gamma == 0 (xhat not recoverable from z) take the
// window scan over Y like bn_bwd_kernel<true, true, false>.  C % 4 == 0.
__global__ __launch_bounds__(256) void bn_bwd_reduce_pool_kernel(const BnBwdArgs a0, const BnBwdArgs a1,
                                                                 const float* __restrict__ beta) {
  const BnBwdArgs& a = blockIdx.y ? a1 : a0;
  const float* __restrict__ apool = a.apool;
  __shared__ float red[256 * 8];
  const int tid = threadIdx.x;
  const int nq = a.C >> 2, rows = 256 / nq;
  const int q = tid % nq, pl = tid / nq;
  const int c0 = q * 4;
  const int Ho = a.H / 2, Wo = a.W / 2;
  const long npix = (long)a.N * Ho * Wo;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (pl < rows) {
    float bev[4], igv[4], scv[4], shv[4], muv[4], isv[4];
    bool degenerate = false;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float g = a.gamma[c0 + i];
      bev[i] = beta[c0 + i];
      igv[i] = g != 0.f ? 1.f / g : 0.f;
      degenerate |= g == 0.f;
      scv[i] = a.scale[c0 + i]; shv[i] = a.shift[c0 + i]; muv[i] = a.mean[c0 + i]; isv[i] = a.invstd[c0 + i];
    }
    for (long p = (long)blockIdx.x * rows + pl; p < npix; p += (long)gridDim.x * rows) {
      const float4 d4 = *reinterpret_cast<const float4*>(a.dout + (size_t)p * a.d_cs + a.d_co + c0);
      const float4 z4 = *reinterpret_cast<const float4*>(apool + (size_t)p * a.C + c0);
      const float dv[4] = {d4.x, d4.y, d4.z, d4.w}, zv[4] = {z4.x, z4.y, z4.z, z4.w};
      if (!degenerate) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float dz = zv[i] > 0.f ? dv[i] : 0.f;
          acc[i] += dz;
          acc[4 + i] += dz * ((zv[i] - bev[i]) * igv[i]);
        }
      } else {
        const int ox = (int)(p % Wo), oy = (int)((p / Wo) % Ho), n = (int)(p / ((long)Wo * Ho));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float best = -1.f, yb = 0.f;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const size_t yo = ((size_t)(n * a.H + 2 * oy + (k >> 1)) * a.W + 2 * ox + (k & 1));
            const float yv = a.y[yo * a.y_cs + a.y_co + c0 + i];
            const float av = fmaxf(fmaf(yv, scv[i], shv[i]), 0.f);
            if (av > best) { best = av; yb = yv; }
          }
          const float dz = best > 0.f ? dv[i] : 0.f;
          acc[i] += dz;
          acc[4 + i] += dz * ((yb - muv[i]) * isv[i]);
        }
      }
    }
  }
  reduce_by_column<8>(acc, red, nq);
  if (tid < nq) {
    double* sm = a.sums + (size_t)(blockIdx.x % NREP) * 2 * a.C;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      acc_add_grad(sm + c0 + i, (double)acc[i]);
      acc_add_grad(sm + a.C + c0 + i, (double)acc[4 + i]);
    }
  }
}

// Layer 0 (Conv2d(1,64,3) + BN + ReLU).  Both passes RECOMPUTE the layer's conv output y0 from the one-channel
// input image (L0Conv::y, 9 fma per value) instead of reading it back: 629 MB less HBM traffic per pass and view at
// B = 32.  Row-based like conv0_direct_kernel; L0_UNROLL pixels per thread are loaded before the arithmetic.
// pass 1: S1 = sum dZ, S2 = sum dZ * xhat over one view (reads dOut only)
// TD = uint16_t (bf16 path): dOut is a bf16 tensor and the recomputed y0 is rounded to bf16 like the stored activation was
template <typename TD = float>
__global__ __launch_bounds__(256) void bn_bwd_reduce_l0_kernel(const BnBwdArgs a0, const BnBwdArgs a1,
                                                               const float* __restrict__ w0,
                                                               const float* __restrict__ b0) {
  const BnBwdArgs& a = blockIdx.y ? a1 : a0;
  const float* __restrict__ x = a.x;
  __shared__ float red[256 * 8];
  const int tid = threadIdx.x;
  const int q = tid & 15, pl = tid >> 4;
  const int c0 = q * 4;
  L0Conv cv;
  cv.load(w0, b0, c0);
  // per channel pair: z = y scale + shift; xhat = y invstd - mean invstd
  f32x2 sc2[2], sh2[2], is2[2], nm2[2], s1[2], s2[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int c = c0 + 2 * h;
    sc2[h] = f32x2{a.scale[c], a.scale[c + 1]}; sh2[h] = f32x2{a.shift[c], a.shift[c + 1]};
    is2[h] = f32x2{a.invstd[c], a.invstd[c + 1]};
    nm2[h] = f32x2{-a.mean[c] * a.invstd[c], -a.mean[c + 1] * a.invstd[c + 1]};
    s1[h] = f32x2{0.f, 0.f}; s2[h] = s1[h];
  }
  const int rows = a.N * a.H, W = a.W;
  extern __shared__ float l0_rows[];
  int par = 0;
  for (int row = blockIdx.x; row < rows; row += gridDim.x, par ^= 1) {
    const int oy = row % a.H;
    const bool up = oy > 0, down = oy < a.H - 1;
    const float* xr = x + (size_t)row * W;
    float* const img = l0_rows + par * 3 * (W + 2);
    l0_stage_rows(img, xr, up, down, W);
    __syncthreads();
    const TD* drow = reinterpret_cast<const TD*>(a.dout) + (size_t)row * W * 64 + c0;
    for (int ox0 = pl; ox0 < W; ox0 += 16 * L0_UNROLL) {
      float4 d4[L0_UNROLL];
      float xv[L0_UNROLL][9];
#pragma unroll
      for (int j = 0; j < L0_UNROLL; ++j) {
        const bool ok = ox0 + 16 * j < W;
        const int ox = ok ? ox0 + 16 * j : W - 1;  // clamped: the loads stay branch-free
        const float4 d = ld4<TD>(drow + (size_t)ox * 64);
        d4[j] = ok ? d : make_float4(0.f, 0.f, 0.f, 0.f);  // dZ == 0 past the end of the row
        l0_taps_lds(img, ox, W, xv[j]);
      }
#pragma unroll
      for (int j = 0; j < L0_UNROLL; ++j) {
        const float dv[4] = {d4[j].x, d4[j].y, d4[j].z, d4[j].w};
        L0Taps xt;
        xt.set(xv[j]);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          f32x2 yv = cv.y2(h, xt);
          if constexpr (sizeof(TD) == 2) {
            const uint32_t pk = pack_bf16(yv[0], yv[1]);
            yv = f32x2{bf16_lo(pk), bf16_hi(pk)};
          }
          const f32x2 z = pk_fma(yv, sc2[h], sh2[h]);
          const f32x2 dz = {z[0] > 0.f ? dv[2 * h] : 0.f, z[1] > 0.f ? dv[2 * h + 1] : 0.f};
          s1[h] = pk_add(s1[h], dz);
          s2[h] = pk_fma(dz, pk_fma(yv, is2[h], nm2[h]), s2[h]);
        }
      }
    }
  }
  float acc[8] = {s1[0][0], s1[0][1], s1[1][0], s1[1][1], s2[0][0], s2[0][1], s2[1][0], s2[1][1]};
  reduce_by_column<8>(acc, red, 16);
  if (tid < 16) {
    double* sm = a.sums + (size_t)(blockIdx.x % NREP) * 128;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      acc_add_grad(sm + c0 + i, (double)acc[i]);
      acc_add_grad(sm + 64 + c0 + i, (double)acc[4 + i]);
    }
  }
}

// pass 2 FUSED with the first layer's weight gradient.  dY0 is consumed in registers (dW0[co][tap] += dY0[p][co] *
// x[p+tap]) and never written: the input image needs no data gradient, so nothing else reads dY0.
template <typename TD = float>
__global__ __launch_bounds__(256) void bn_bwd_apply_l0_kernel(const BnBwdArgs a0, const BnBwdArgs a1,
                                                              const float* __restrict__ w0, const float* __restrict__ b0,
                                                              float* __restrict__ dw) {
  const BnBwdArgs& a = blockIdx.y ? a1 : a0;
  const float* __restrict__ x = a.x;
  __shared__ float red[256 * 10];
  const int tid = threadIdx.x;
  const int q = tid & 15, pl = tid >> 4;  // channel quad, pixel lane
  const int c0 = q * 4;
  L0Conv cv;
  cv.load(w0, b0, c0);
  // per channel pair: z = y scale + shift; xhat = y invstd - mean invstd; dY = gs (dZ - k1 - xhat k2)
  f32x2 sc2[2], sh2[2], is2[2], nm2[2], nk1[2], nk2[2], gs2[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int c = c0 + 2 * h;
    sc2[h] = f32x2{a.scale[c], a.scale[c + 1]}; sh2[h] = f32x2{a.shift[c], a.shift[c + 1]};
    is2[h] = f32x2{a.invstd[c], a.invstd[c + 1]};
    nm2[h] = f32x2{-a.mean[c] * a.invstd[c], -a.mean[c + 1] * a.invstd[c + 1]};
    nk1[h] = f32x2{-a.k12[c], -a.k12[c + 1]};
    nk2[h] = f32x2{-a.k12[64 + c], -a.k12[64 + c + 1]};
    gs2[h] = f32x2{a.gamma[c] * a.invstd[c], a.gamma[c + 1] * a.invstd[c + 1]};
  }
  f32x2 wacc[2][9];  // dW0 of the two channel pairs
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int t = 0; t < 9; ++t) wacc[h][t] = f32x2{0.f, 0.f};
  const int rows = a.N * a.H, W = a.W;
  extern __shared__ float l0_rows[];
  int par = 0;
  for (int row = blockIdx.x; row < rows; row += gridDim.x, par ^= 1) {
    const int oy = row % a.H;
    const bool up = oy > 0, down = oy < a.H - 1;
    const float* xr = x + (size_t)row * W;
    float* const img = l0_rows + par * 3 * (W + 2);
    l0_stage_rows(img, xr, up, down, W);
    __syncthreads();
    const TD* drow = reinterpret_cast<const TD*>(a.dout) + (size_t)row * W * 64 + c0;
    for (int ox0 = pl; ox0 < W; ox0 += 16 * L0_UNROLL_APPLY) {
      float4 d4[L0_UNROLL_APPLY];
      float xv[L0_UNROLL_APPLY][9];
      bool ok[L0_UNROLL_APPLY];
#pragma unroll
      for (int j = 0; j < L0_UNROLL_APPLY; ++j) {
        ok[j] = ox0 + 16 * j < W;
        const int ox = ok[j] ? ox0 + 16 * j : W - 1;  // clamped: the loads stay branch-free
        d4[j] = ld4<TD>(drow + (size_t)ox * 64);
        l0_taps_lds(img, ox, W, xv[j]);
      }
#pragma unroll
      for (int j = 0; j < L0_UNROLL_APPLY; ++j) {
        const float dv[4] = {d4[j].x, d4[j].y, d4[j].z, d4[j].w};
        const float okf = ok[j] ? 1.f : 0.f;  // a clamped (repeated) pixel past the end of the row contributes nothing
        L0Taps xt;
        xt.set(xv[j]);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          f32x2 yv = cv.y2(h, xt);
          if constexpr (sizeof(TD) == 2) {
            const uint32_t pk = pack_bf16(yv[0], yv[1]);
            yv = f32x2{bf16_lo(pk), bf16_hi(pk)};
          }
          const f32x2 z = pk_fma(yv, sc2[h], sh2[h]);
          const f32x2 dz = {z[0] > 0.f ? dv[2 * h] : 0.f, z[1] > 0.f ? dv[2 * h + 1] : 0.f};
          const f32x2 xh = pk_fma(yv, is2[h], nm2[h]);
          const f32x2 g = pk_mul(pk_mul(gs2[h], f32x2{okf, okf}), pk_fma(xh, nk2[h], pk_add(dz, nk1[h])));
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            wacc[h][2 * k] = pk_fma_bcast<false>(xt.p[k], g, wacc[h][2 * k]);
            wacc[h][2 * k + 1] = pk_fma_bcast<true>(xt.p[k], g, wacc[h][2 * k + 1]);
          }
          wacc[h][8] = pk_fma_bcast<false>(xt.p[4], g, wacc[h][8]);
        }
      }
    }
  }
  // block reduction over the 16 pixel lanes, one channel of the quad at a time: [256][10] floats
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 9; ++t) red[tid * 10 + t] = wacc[i >> 1][t][i & 1];
    __syncthreads();
    if (tid < 160) {  // 16 quads x 10 slots
      const int qq = tid / 10, t = tid - qq * 10;
      if (t < 9) {
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += red[(r * 16 + qq) * 10 + t];
        facc_add(dw + (qq * 4 + i) * 9 + t, sum);
      }
    }
  }
}

// Between the two passes: reduce the NREP replicas of the fp64 sums ONCE (one thread per channel) into
// k12 = {S1/n, S2/n} for pass 2, and accumulate dgamma += S2, dbeta += S1.  (Letting every pass-2 thread sum the
// 32 replicas itself cost a fixed ~110 us per launch: 2.6 ms per step in the first profiles.)
template <typename T = float>  // element type of y / dout (the pool_fix scan is the only place that reads them)
__device__ __forceinline__ void bn_bwd_sums_body(const BnBwdArgs& a0, const BnBwdArgs& a1, int nviews, float* __restrict__ dgamma,
                                                 float* __restrict__ dbeta) {
  // 32 lanes per channel (replica r each); lane 0 of the group writes.  grid: ceil(C * 32 / blockDim)
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if ((t >> 5) >= ((a0.C + 7) & ~7)) return;   // (whole 256-thread workgroups past the layer: a launch shared with a wider layer)
  const int c = min(t >> 5, a0.C - 1), r = t & 31;
  const bool writer = r == 0 && (t >> 5) < a0.C;
  for (int v = 0; v < nviews; ++v) {  // the views accumulate into the same gradients: one after the other in this thread
    const BnBwdArgs& a = v ? a1 : a0;
    const int C = a.C;
    double s1, s2;
    replica_sums(a.sums, C, c, r, s1, s2);
    if (a.pool_fix && a.gamma[c] == 0.f) {
      // degenerate channel of a pooled layer: z == beta everywhere, the first window element is the arg-max
      double part = 0.0;
      if (a.beta[c] > 0.f) {
        const int Ho = a.H / 2, Wo = a.W / 2;
        const long npool = (long)a.N * Ho * Wo;
        for (long p = r; p < npool; p += 32) {
          const int ox = (int)(p % Wo), oy = (int)((p / Wo) % Ho), n = (int)(p / ((long)Wo * Ho));
          const size_t yi = ((size_t)(n * a.H + 2 * oy) * a.W + 2 * ox) * a.y_cs + a.y_co + c, di = (size_t)p * a.d_cs + a.d_co + c;
          float y0, dv;
          if constexpr (sizeof(T) == 4) { y0 = a.y[yi]; dv = a.dout[di]; }
          else {
            y0 = bf16_lo(reinterpret_cast<const uint16_t*>(a.y)[yi]);
            dv = bf16_lo(reinterpret_cast<const uint16_t*>(a.dout)[di]);
          }
          part += (double)(dv * ((y0 - a.mean[c]) * a.invstd[c]));
        }
      }
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) part += __shfl_xor(part, o);
      s2 = part;
    }
    if (!writer) continue;
    const float k1 = (float)(s1 / a.count), k2 = (float)(s2 / a.count);
    float* k12 = const_cast<float*>(a.k12);
    k12[c] = k1;
    k12[C + c] = k2;
    if (dbeta != nullptr) dbeta[c] += (float)s1;
    if (dgamma != nullptr) dgamma[c] += (float)s2;
    // Gradient of the conv bias that feeds this BatchNorm: sum_p dY = gamma*invstd*(S1 - n*k1 - k2*sum xhat) == 0 in
    // exact arithmetic (the reference's autograd produces rounding noise here).  It is
    // evaluated from the sums (what is left is the fp32 rounding of k1) instead of by one float atomic per channel
    // and block in pass 2, which cost ~110 us per launch through same-address contention.
    if (a.dbias != nullptr) a.dbias[c] += a.gamma[c] * a.invstd[c] * (float)(s1 - a.count * (double)k1);
  }
}
template <typename T = float>
__global__ void bn_bwd_sums_kernel(const BnBwdArgs a0, const BnBwdArgs a1, int nviews, float* __restrict__ dgamma,
                                   float* __restrict__ dbeta) {
  bn_bwd_sums_body<T>(a0, a1, nviews, dgamma, dbeta);
}
// up to three layers whose pass 1 is complete at the same point of the backward pass (the 3x3 heads): blockIdx.y = layer
struct BnSumsJobs {
  BnBwdArgs a0[3], a1[3];
  float* dgamma[3];
  float* dbeta[3];
  int n;
};
template <typename T = float>
__global__ void bn_bwd_sums_multi_kernel(const BnSumsJobs J, int nviews) {
  const int j = blockIdx.y;
  bn_bwd_sums_body<T>(J.a0[j], J.a1[j], nviews, J.dgamma[j], J.dbeta[j]);
}

// ------------------------------------------------------------------------------------------------
// Head epilogues.
// semi = bnPb(convPb(.)) : NHWC raw [cells][cs] -> NCHW [N,65,Hc,Wc] (API output only).
// desc = bnDb(convDb(.)) / ||.||_2 : one wave per cell (256 channels = 4 per lane).
// ------------------------------------------------------------------------------------------------
__global__ void nhwc_affine_to_nchw_kernel(const float* __restrict__ y, const float* __restrict__ scale,
                                           const float* __restrict__ shift, float* __restrict__ out, int N, int HW,
                                           int C, int cs, int co) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;  // over N*C*HW, HW fastest
  if (idx >= (long)N * C * HW) return;
  const int p = (int)(idx % HW);
  const int c = (int)((idx / HW) % C);
  const int n = (int)(idx / ((long)HW * C));
  float v = y[((size_t)n * HW + p) * cs + co + c];
  if (scale != nullptr) v = fmaf(v, scale[c], shift[c]);
  out[idx] = v;
}

__global__ void nchw_to_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out, int N, int HW, int C, int cs,
                                    int co) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;  // over N*HW*C, C fastest
  if (idx >= (long)N * C * HW) return;
  const int c = (int)(idx % C);
  const int p = (int)((idx / C) % HW);
  const int n = (int)(idx / ((long)HW * C));
  out[((size_t)n * HW + p) * cs + co + c] = in[((size_t)n * C + c) * HW + p];
}

// desc: raw YDb [cells][cs] -> normalised desc [cells][256] and 1/norm [cells]
// (models/SuperPointNet_gauss2.py:64-65, no epsilon).
// blockIdx.y = view of the pair (one launch for both: a small launch costs ~13 us of the step whatever it does, PERF_LOG round 6
// section 4); the second pointer set may be omitted (gridDim.y == 1).
__global__ __launch_bounds__(256) void desc_normalize_kernel(const float* __restrict__ y0, const float* __restrict__ scale0,
                                                             const float* __restrict__ shift0, float* __restrict__ desc0,
                                                             float* __restrict__ inv_norm0, float* __restrict__ zero_out0, int ncells,
                                                             int cs, int co, const float* __restrict__ y1 = nullptr,
                                                             const float* __restrict__ scale1 = nullptr,
                                                             const float* __restrict__ shift1 = nullptr, float* __restrict__ desc1 = nullptr,
                                                             float* __restrict__ inv_norm1 = nullptr, float* __restrict__ zero_out1 = nullptr) {
  const bool v1 = blockIdx.y != 0;
  const float* __restrict__ y = v1 ? y1 : y0;
  const float* __restrict__ scale = v1 ? scale1 : scale0;
  const float* __restrict__ shift = v1 ? shift1 : shift0;
  float* __restrict__ desc = v1 ? desc1 : desc0;
  float* __restrict__ inv_norm = v1 ? inv_norm1 : inv_norm0;
  float* __restrict__ zero_out = v1 ? zero_out1 : zero_out0;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (wave >= ncells) return;
  // (training step with the sparse descriptor loss: d(desc) [cells][256], the scatter target of the loss kernels, starts at zero)
  if (zero_out != nullptr) *reinterpret_cast<float4*>(zero_out + (size_t)wave * 256 + lane * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 v = *reinterpret_cast<const float4*>(y + (size_t)wave * cs + co + lane * 4);
  const float4 sc = *reinterpret_cast<const float4*>(scale + lane * 4);
  const float4 sh = *reinterpret_cast<const float4*>(shift + lane * 4);
  float4 r = make_float4(fmaf(v.x, sc.x, sh.x), fmaf(v.y, sc.y, sh.y), fmaf(v.z, sc.z, sh.z), fmaf(v.w, sc.w, sh.w));
  float s = r.x * r.x + r.y * r.y + r.z * r.z + r.w * r.w;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float inv = 1.f / sqrtf(s);
  *reinterpret_cast<float4*>(desc + (size_t)wave * 256 + lane * 4) = make_float4(r.x * inv, r.y * inv, r.z * inv, r.w * inv);
  if (lane == 0) inv_norm[wave] = inv;
}

// backward of the L2 normalisation: d_raw = (d - desc * <desc, d>) * inv_norm   (in place on d)
__global__ __launch_bounds__(256) void desc_normalize_bwd_kernel(const float* __restrict__ desc0,
                                                                 const float* __restrict__ inv_norm0,
                                                                 float* __restrict__ d0, int ncells,
                                                                 const float* __restrict__ desc1 = nullptr,
                                                                 const float* __restrict__ inv_norm1 = nullptr,
                                                                 float* __restrict__ d1 = nullptr) {   // (blockIdx.y = view)
  const bool v1 = blockIdx.y != 0;
  const float* __restrict__ desc = v1 ? desc1 : desc0;
  const float* __restrict__ inv_norm = v1 ? inv_norm1 : inv_norm0;
  float* __restrict__ d = v1 ? d1 : d0;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (wave >= ncells) return;
  const float4 n = *reinterpret_cast<const float4*>(desc + (size_t)wave * 256 + lane * 4);
  float4 g = *reinterpret_cast<const float4*>(d + (size_t)wave * 256 + lane * 4);
  float s = n.x * g.x + n.y * g.y + n.z * g.z + n.w * g.w;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float inv = inv_norm[wave];
  g = make_float4((g.x - n.x * s) * inv, (g.y - n.y * s) * inv, (g.z - n.z * s) * inv, (g.w - n.w * s) * inv);
  *reinterpret_cast<float4*>(d + (size_t)wave * 256 + lane * 4) = g;
}

// Adam (torch.optim.Adam defaults: betas (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad)
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long n, float lr, float bc1, float bc2_sqrt) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float gi = g[i];
  const float mi = 0.9f * m[i] + 0.1f * gi;
  const float vi = 0.999f * v[i] + 0.001f * gi * gi;
  m[i] = mi;
  v[i] = vi;
  const float denom = sqrtf(vi) / bc2_sqrt + 1e-8f;
  p[i] -= (lr / bc1) * (mi / denom);
}

// the same step on g * gscale (data parallel: gscale = 1 / world after the all-reduce SUM), leaving g untouched
__global__ void adam_scaled_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                   float* __restrict__ v, long n, float lr, float bc1, float bc2_sqrt, float gscale) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float gi = g[i] * gscale;
  const float mi = 0.9f * m[i] + 0.1f * gi;
  const float vi = 0.999f * v[i] + 0.001f * gi * gi;
  m[i] = mi;
  v[i] = vi;
  const float denom = sqrtf(vi) / bc2_sqrt + 1e-8f;
  p[i] -= (lr / bc1) * (mi / denom);
}

}  // namespace sspk
