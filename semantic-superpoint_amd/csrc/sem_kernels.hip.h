// Segmentation head loss (fused bilinear upsample + log-softmax + NLL, never materialising the
// [B,133,H,W] logits) and the device-side sparse-loss index sampler.
//   reference: models/SuperPointNet_gauss2_ssmall.py:87-91 (F.interpolate bilinear, align_corners=False),
//              Train_model_heatmap_all.py:181-193 (CrossEntropyLoss(ignore_index=133)),
//              utils/loss_functions/sparse_loss.py:184-246, correspondence_finder.py:29-34 (sampling).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "loss_kernels.hip.h"
#include "pair_kernels.hip.h"

namespace sspk {

// PyTorch area_pixel_compute_source_index (align_corners=False, non-cubic): clamp below at 0.
__device__ __forceinline__ void up_src(int dst, int in_size, int out_size, int& i0, int& i1, float& l1) {
  const float scale = (float)in_size / (float)out_size;
  float s = scale * ((float)dst + 0.5f) - 0.5f;
  if (s < 0.f) s = 0.f;
  i0 = (int)s;
  i1 = i0 + ((i0 < in_size - 1) ? 1 : 0);
  l1 = s - (float)i0;
}

// Number of non-ignored labels (CrossEntropyLoss(ignore_index=C) averages over them): sem_cnt[view] += count, both views in
// one launch (blockIdx.y).  Two labels per 16-byte load, four loads in flight per thread: the kernel sits on the critical path
// in front of sem_ce_kernel and was latency-bound (40 us per view for 20 MB with one 8-byte load per thread and trip).
// zero0 / zero1 (optional): nzero floats per view set to 0 on the way - d(convSout), the scatter target of the loss kernel (16-byte
// aligned, nzero % 4 == 0: [cells][channel stride], stride % 4 == 0).
__global__ __launch_bounds__(256) void sem_count_kernel(const int64_t* __restrict__ labels0, const int64_t* __restrict__ labels1,
                                                        long n, int C, StepAccum* __restrict__ acc, float* __restrict__ zero0 = nullptr,
                                                        float* __restrict__ zero1 = nullptr, long nzero = 0) {
  __shared__ float red[4];
  const int view = blockIdx.y;
  const int64_t* __restrict__ labels = view ? labels1 : labels0;
  if (float* const z = view ? zero1 : zero0) {
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q * 4 < nzero; q += (long)gridDim.x * blockDim.x)
      *reinterpret_cast<f32x4*>(z + q * 4) = zero4;
  }
  typedef long long ll2 __attribute__((ext_vector_type(2)));
  float cnt = 0.f;
  const long stride = (long)gridDim.x * blockDim.x;
  const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long npair = ((reinterpret_cast<size_t>(labels) & 15) == 0) ? n >> 1 : 0;  // (0 <= label < C; C = ignore_index; anything
  long i = tid;                                                                     // else is ignored too: sem_ce_kernel)
  for (; i + 3 * stride < npair; i += 4 * stride) {
    ll2 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const ll2*>(labels + 2 * (i + u * stride));
#pragma unroll
    for (int u = 0; u < 4; ++u)
      cnt += (((uint64_t)v[u][0] < (uint64_t)C) ? 1.f : 0.f) + (((uint64_t)v[u][1] < (uint64_t)C) ? 1.f : 0.f);
  }
  for (; i < npair; i += stride) {
    const ll2 v = *reinterpret_cast<const ll2*>(labels + 2 * i);
    cnt += (((uint64_t)v[0] < (uint64_t)C) ? 1.f : 0.f) + (((uint64_t)v[1] < (uint64_t)C) ? 1.f : 0.f);
  }
  for (long k = 2 * npair + tid; k < n; k += stride) cnt += ((uint64_t)labels[k] < (uint64_t)C) ? 1.f : 0.f;
  cnt = wave_sum(cnt);
  const float tot = block_sum_of_waves(cnt, red);
  if (threadIdx.x == 0) unsafeAtomicAdd(&acc->sem_cnt[view], (double)tot);
}

// Fused bilinear upsample (align_corners=False) + log-softmax + NLL (+ gradient), one wave per 8x8 pixel tile
// shifted by (4,4): all 64 pixels of such a tile interpolate between the same 4 source cells.  The kernel is bound by
// vector instructions (C = 134 classes x 64 pixels x ~10 operations per tile), so both passes run on the packed fp32
// instructions, in the base-2 domain (weights and shift pre-multiplied by log2 e: exp(l - m) = exp2(l' - m')):
//   pass A (lanes = PIXELS): the wave walks the classes TWO at a time; the 4 corner logits of a class pair are
//     wave-uniform (scalar 8-byte loads), every lane interpolates its own pixel and accumulates sum_c exp(l_c - m); the
//     label logit is gathered afterwards.  The shift m is not the exact maximum but the bound
//     sum_k w_k max_c corner_k[c] >= max_c l_c (the weights are >= 0 and sum to 1), which costs 4 wave reductions
//     per TILE instead of 2 per pixel; log-sum-exp is shift invariant.
//   pass B (lanes = CLASSES, c = lane + 64 j, j < 3; gradient only): the wave walks the pixels TWO at a time (pixel
//     pair = the halves of the packed registers; weights, shift and g / sum come back from LDS as broadcast reads);
//     d(convSout) of the 4 corners accumulates in 2 x 12 registers per lane.  The -g [c == label] term does not go
//     through the class lanes at all: every pixel adds -g w_k to an LDS histogram over (corner, class), which joins the
//     accumulators in the flush (12 atomics per lane and tile).  sem_cnt[view] must be final before a BWD launch.
// MODE bit 0: accumulate the loss sum, bit 1: accumulate d(convSout); the training step does both in ONE pass (3).
__device__ __forceinline__ float lane_bcast(float v, int src_lane) {  // src_lane wave-uniform
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src_lane));
}

// x.lo / x.hi broadcast * y (both halves) [+ z], y wave-uniform in a scalar register pair
template <bool BCAST_HI>
__device__ __forceinline__ f32x2 pk_fma_bcast_sy(f32x2 x, f32x2 y, f32x2 z) {
  f32x2 d;
  if (BCAST_HI) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(x), "s"(y), "v"(z));
  else asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(d) : "v"(x), "s"(y), "v"(z));
  return d;
}
template <bool BCAST_HI>
__device__ __forceinline__ f32x2 pk_mul_bcast_sy(f32x2 x, f32x2 y) {
  f32x2 d;
  if (BCAST_HI) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(d) : "v"(x), "s"(y));
  else asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(d) : "v"(x), "s"(y));
  return d;
}
template <bool BCAST_HI>
__device__ __forceinline__ f32x2 pk_mul_bcast(f32x2 x, f32x2 y) {
  f32x2 d;
  if (BCAST_HI) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(d) : "v"(x), "v"(y));
  else asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(d) : "v"(x), "v"(y));
  return d;
}
// exp2 of both halves.  gfx950 needs one wait state between a transcendental instruction and a non-transcendental
// VALU instruction that reads its result; hipcc does not insert it in front of inline asm, so it is part of the value.
__device__ __forceinline__ f32x2 pk_exp2(f32x2 x) {
  f32x2 e = {__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};
  asm volatile("s_nop 0" : "+v"(e));
  return e;
}

constexpr int SEM_MAX_C = 192;  // 3 class slots per lane
constexpr int SEM_PAR = 12;     // floats per pixel pair in LDS: w0e p,q | w1e | w2e | w3e | -m e | g / sum

template <int MODE>
__global__ __launch_bounds__(256) void sem_ce_kernel(const float* __restrict__ sout, const int64_t* __restrict__ labels,
                                                     float* __restrict__ dsout, StepAccum* __restrict__ acc, int view,
                                                     int B, int Hc, int Wc, int H, int W, int C, int cs) {
  constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
  __shared__ float red[4];
  __shared__ __attribute__((aligned(16))) float s_par[4][32 * SEM_PAR];
  __shared__ __attribute__((aligned(16))) float s_hist[4][4 * SEM_MAX_C];
  const int wave_in_blk = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  float* const par = s_par[wave_in_blk];
  float* const hist = s_hist[wave_in_blk];
  const int TX = Wc + 1, TY = Hc + 1;
  const int ntile = B * TX * TY;
  constexpr bool FWD = (MODE & 1) != 0, BWD = (MODE & 2) != 0;
  float nll_acc = 0.f;  // per lane (= per pixel slot of the tiles this wave visits)
  const float g = BWD ? acc->coef_sem / (float)acc->sem_cnt[view] : 0.f;
  const DetTarget t_ds = det_resolve(dsout);   // (deterministic mode: fixed-point shadow of d(convSout))
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  if (BWD) {
#pragma unroll
    for (int i = 0; i < 4 * SEM_MAX_C / 256; ++i) *reinterpret_cast<f32x4*>(hist + (i * 64 + lane) * 4) = zero4;
  }
  for (int tile = blockIdx.x * 4 + wave_in_blk; tile < ntile; tile += gridDim.x * 4) {
    const int tx = tile % TX - 1, ty = (tile / TX) % TY - 1, n = tile / (TX * TY);
    const int cy0 = max(ty, 0), cy1 = min(ty + 1, Hc - 1), cx0 = max(tx, 0), cx1 = min(tx + 1, Wc - 1);
    const int cidx[4] = {cy0 * Wc + cx0, cy0 * Wc + cx1, cy1 * Wc + cx0, cy1 * Wc + cx1};
    // this lane's pixel: bilinear weights and label
    const int y = 8 * ty + 4 + (lane >> 3), x = 8 * tx + 4 + (lane & 7);
    const bool inside = y >= 0 && y < H && x >= 0 && x < W;
    int label_l = C;
    float wy1_l = 0.f, wx1_l = 0.f;
    if (inside) {
      int a0, a1;
      up_src(y, Hc, H, a0, a1, wy1_l);
      up_src(x, Wc, W, a0, a1, wx1_l);
      const int64_t lv = labels[((size_t)n * H + y) * W + x];
      label_l = (uint64_t)lv < (uint64_t)C ? (int)lv : C;  // (the same test as sem_count_kernel, on all 64 bits)
    }
    // labels outside [0, C] (torch raises; e.g. 255 / -1 from a dataset) are treated like the ignore index C: they must never
    // index the corner logits or the LDS histogram below
    const bool counted = inside && (unsigned)label_l < (unsigned)C;
    const unsigned long long counted_mask = __ballot(counted);
    if (counted_mask == 0ull) continue;  // wave-uniform: ignored / outside pixels contribute nothing
    const float wy0_l = 1.f - wy1_l, wx0_l = 1.f - wx1_l;
    const float w4_l[4] = {wy0_l * wx0_l, wy0_l * wx1_l, wy1_l * wx0_l, wy1_l * wx1_l};
    const float* cp[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) cp[k] = sout + ((size_t)n * Hc * Wc + cidx[k]) * cs;
    // corner logits of this lane's classes (pass B) and their maxima over the classes (the shift of pass A)
    float cv[4][3];
    bool cok[3];
    float m_l = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) cok[j] = lane + 64 * j < C;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float mx = -INFINITY;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        cv[k][j] = cok[j] ? cp[k][lane + 64 * j] : 0.f;
        if (cok[j]) mx = fmaxf(mx, cv[k][j]);
      }
      m_l = fmaf(w4_l[k], wave_max(mx), m_l);
    }
    // base-2 domain: l' = l log2 e, m' = m log2 e
    const f32x2 wp01 = {w4_l[0] * LOG2E, w4_l[1] * LOG2E}, wp23 = {w4_l[2] * LOG2E, w4_l[3] * LOG2E};
    const float nme_l = -m_l * LOG2E;
    // ---- pass A: lanes = pixels, two classes per instruction ----
    float se;
    {
      const f32x2 nme2 = {nme_l, nme_l};
      f32x2 se2 = {0.f, 0.f};
      int c = 0;
#define SEM_CLASS_PAIR(S0, S1, S2, S3)                                                         \
      {                                                                                        \
        f32x2 l2 = pk_mul_bcast_sy<true>(wp23, S3);                                            \
        l2 = pk_fma_bcast_sy<false>(wp23, S2, l2);                                             \
        l2 = pk_fma_bcast_sy<true>(wp01, S1, l2);                                              \
        l2 = pk_fma_bcast_sy<false>(wp01, S0, l2);                                             \
        se2 = pk_add(se2, pk_exp2(pk_add(l2, nme2)));                                          \
      }
      for (; c + 7 < C; c += 8) {  // 16 scalar 8-byte loads in flight, then 4 class pairs
        f32x2 sv[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int k = 0; k < 4; ++k) sv[u][k] = *reinterpret_cast<const f32x2*>(cp[k] + c + 2 * u);
#pragma unroll
        for (int u = 0; u < 4; ++u) SEM_CLASS_PAIR(sv[u][0], sv[u][1], sv[u][2], sv[u][3])
      }
      for (; c + 1 < C; c += 2) {
        const f32x2 s0 = *reinterpret_cast<const f32x2*>(cp[0] + c), s1 = *reinterpret_cast<const f32x2*>(cp[1] + c);
        const f32x2 s2 = *reinterpret_cast<const f32x2*>(cp[2] + c), s3 = *reinterpret_cast<const f32x2*>(cp[3] + c);
        SEM_CLASS_PAIR(s0, s1, s2, s3)
      }
#undef SEM_CLASS_PAIR
      se = se2[0] + se2[1];
      if (c < C) {  // odd class count
        const float l = fmaf(wp01[0], cp[0][c], fmaf(wp01[1], cp[1][c], fmaf(wp23[0], cp[2][c], wp23[1] * cp[3][c])));
        se += __builtin_amdgcn_exp2f(l + nme_l);
      }
    }
    if (FWD && counted) {  // the label logit, gathered: sum_k w_k corner_k[label]
      const float ll = fmaf(w4_l[0], cp[0][label_l], fmaf(w4_l[1], cp[1][label_l], fmaf(w4_l[2], cp[2][label_l], w4_l[3] * cp[3][label_l])));
      nll_acc += (m_l + logf(se)) - ll;
    }
    if (BWD) {
      // ---- pass B: lanes = classes, two pixels per instruction ----
      const float gi_l = counted ? g / se : 0.f;  // uncounted pixels: e is finite (<= 1), d = 0
      {
        float* pr = par + (lane >> 1) * SEM_PAR + (lane & 1);
        pr[0] = wp01[0]; pr[2] = wp01[1]; pr[4] = wp23[0]; pr[6] = wp23[1]; pr[8] = nme_l; pr[10] = gi_l;
        if (counted) {  // -g [c == label] w_k, in the base-2 domain like the accumulators
          atomicAdd(hist + 0 * SEM_MAX_C + label_l, -g * wp01[0]);
          atomicAdd(hist + 1 * SEM_MAX_C + label_l, -g * wp01[1]);
          atomicAdd(hist + 2 * SEM_MAX_C + label_l, -g * wp23[0]);
          atomicAdd(hist + 3 * SEM_MAX_C + label_l, -g * wp23[1]);
        }
      }
      __builtin_amdgcn_wave_barrier();  // LDS executes one wave's instructions in order; this only pins the compiler
      f32x2 cva[4], cvb[4];  // classes (lane, lane + 64) and (lane + 128, -) of the four corners
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        cva[k] = f32x2{cv[k][0], cv[k][1]};
        cvb[k] = f32x2{cv[k][2], 0.f};
      }
      f32x2 dacc[4][3];
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < 3; ++j) dacc[k][j] = f32x2{0.f, 0.f};
      for (int i = 0; i < 32; ++i) {
        if (((counted_mask >> (2 * i)) & 3ull) == 0ull) continue;  // wave-uniform
        const f32x4 q0 = *reinterpret_cast<const f32x4*>(par + i * SEM_PAR);
        const f32x4 q1 = *reinterpret_cast<const f32x4*>(par + i * SEM_PAR + 4);
        const f32x4 q2 = *reinterpret_cast<const f32x4*>(par + i * SEM_PAR + 8);
        const f32x2 W0 = lo2(q0), W1 = hi2(q0), W2 = lo2(q1), W3 = hi2(q1), NM = lo2(q2), GI = hi2(q2);
#define SEM_CLASS_SLOT(J, CV, HI)                                                              \
        {                                                                                      \
          f32x2 l2 = pk_mul_bcast<HI>(CV[3], W3);                                              \
          l2 = pk_fma_bcast<HI>(CV[2], W2, l2);                                                \
          l2 = pk_fma_bcast<HI>(CV[1], W1, l2);                                                \
          l2 = pk_fma_bcast<HI>(CV[0], W0, l2);                                                \
          const f32x2 d2 = pk_mul(pk_exp2(pk_add(l2, NM)), GI);                                \
          dacc[0][J] = pk_fma(W0, d2, dacc[0][J]);                                             \
          dacc[1][J] = pk_fma(W1, d2, dacc[1][J]);                                             \
          dacc[2][J] = pk_fma(W2, d2, dacc[2][J]);                                             \
          dacc[3][J] = pk_fma(W3, d2, dacc[3][J]);                                             \
        }
        SEM_CLASS_SLOT(0, cva, false)
        if (C > 64) SEM_CLASS_SLOT(1, cva, true)
        if (C > 128) SEM_CLASS_SLOT(2, cvb, false)
#undef SEM_CLASS_SLOT
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int j = 0; j < 3; ++j)
        if (cok[j]) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float v = ((dacc[k][j][0] + dacc[k][j][1]) + hist[k * SEM_MAX_C + lane + 64 * j]) * LN2;
            if (v != 0.f) facc_add(t_ds, dsout + ((size_t)n * Hc * Wc + cidx[k]) * cs + lane + 64 * j, v);
          }
        }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int i = 0; i < 4 * SEM_MAX_C / 256; ++i) *reinterpret_cast<f32x4*>(hist + (i * 64 + lane) * 4) = zero4;
    }
  }
  if (FWD) {
    const float tot = block_sum_of_waves(wave_sum(nll_acc), red);
    if (threadIdx.x == 0) acc_add_loss(&acc->sem_sum[view], (double)tot);
  }
}

// ------------------------------------------------------------------------------------------------
// The same loss with lanes = (x, class) - the form the training step runs for H = 8 Hc, W = 8 Wc, C <= 144 (133 here).
//
// sem_ce_kernel above pays for the two lane layouts it needs (pixels for the sum over the classes, classes for the sum over
// the pixels) by interpolating and exponentiating every (pixel, class) TWICE, the second time with 133 classes in 3 x 64 lane
// slots (69 % of the lanes): 9 + 13 vector instructions per (pixel, class) pair = 27 ns of SIMD issue per element
// (tools/ubench/valu_rate.hip: v_pk_* 2.1 ns, v_exp_f32 3.45 ns, plain fp32 1.3-1.4 ns per wave instruction at >= 2 waves per SIMD).
// Here a lane owns the column x = xl + 4 xr (xl = lane & 3, xr = 0, 1) and the classes c = 16 blk + cl (cl = lane >> 2,
// blk < 9: 133 of 144 slots = 92 %) of the shifted 8x8 tile, all 8 rows y in registers:
//   * every tile has the SAME interpolation weights ((2 y + 1) / 16, (2 x + 1) / 16 - the tile is shifted by (4,4), so its 64
//     pixels lie between the same four cells; at the image border the four cells coincide pairwise and any convex weights
//     give the clamped value), so they are compile-time constants per lane and exact;
//   * the shift of the log-sum-exp goes into the OPERAND: c''_k[c] = (corner_k[c] - M_k) log2 e with M_k = max_c corner_k[c];
//     the interpolation of c'' is (l_c - m) log2 e with m = sum_k w_k M_k >= max_c l_c, the same bound as above, for free;
//   * in a lane, l'(y) = T0 + (2 y + 1) / 16 (T1 - T0) is LINEAR in y (T_a = the column interpolation of corner row a), so the
//     eight exponentials are two geometric progressions: e(0) = exp2(l'(0)), e(y + 1) = e(y) r and e(7) = exp2(l'(7)),
//     e(y - 1) = e(y) / r with r = exp2((T1 - T0) / 8) - 4 v_exp_f32 and 6 multiplies for 8 values.  Both ends are anchored
//     (the larger end is one of them, l' <= 0, and a progression runs three steps; the two progressions advance as ONE packed
//     multiply by (r, 1 / r)); the exponent of r is clamped to +-126 so that 0 * inf cannot appear, underflow: SEMX_K below;
//   * the ANCHORS AND RATIOS stay in registers (4 per column and class block, 72 per lane) for the gradient, whose sweep runs the
//     progressions again (3 packed multiplies) under 2 packed fma per row pair: sum over y in the lane, over the two columns with
//     the x weights, over the four x lanes of a quad on the DPP path (9 instructions per 16 classes); lane xl keeps corner xl;
//   * sum over the classes: 16 partial sums per lane (its 8 rows x 2 columns), transposed through LDS - lane (xl, cl) receives
//     the total of pixel slot j = cl and plays that pixel for the label logit, the NLL, g / sum and the -g [c == label]
//     histogram; g / sum goes back through LDS (16 values per lane);
//   * atomics: see the launch geometry in front of the kernel (5 cell vectors per step of four tiles instead of 16).
// ~90 vector + 8 transcendental instructions per 16 classes and lane = 7-8 ns per element instead of 27.
// Numerics: same bound m, same base-2 domain; three chained multiplies add <= 2 ulp to an exponential.
#ifndef SEMX_WGS
#define SEMX_WGS 2  // workgroups per CU = waves per SIMD the register allocation aims at
#endif
#ifndef SEMX_ABL
#define SEMX_ABL 0  // compile-time perf ablation (tools/ubench/sem_ce_bench.hip -DSEMX_ABL=n): 1 no global atomics, 2 no LDS label
                    // histogram, 4 no gradient sweep, 8 no next-tile requests ahead (the loads sit in front of their use),
                    // 16 no group reduction in front of the label histogram
#endif
constexpr int SEMX_NB = 9;                 // 16-class blocks per lane
constexpr int SEMX_TP = 16 * 16 + 16;      // floats per x-lane plane of the transposition (+16: every lane of a read on its own bank)
constexpr int SEMX_HS = SEMX_NB * 16;      // label histogram: floats per corner
// Every exponential carries the factor 2^K (added to the operand, taken out of log(sum) and, in the gradient, with the x weights).
// The bound m may sit far above the largest logit of a pixel (neighbouring cells that disagree: up to the logit range), and a
// progression that starts from an underflowed end loses the terms up to 2^(-4/7 (126 + K)) of 2^0 = the bound: with K = 116
// the loss of a pixel stays exact to fp32 while its largest term is above 2^-114 of the bound = 79 in natural units (the direct
// evaluation above, without K: 2^-102, 70).  144 classes x 2^116 < 2^124: the sums cannot overflow.
constexpr float SEMX_K = 116.f, SEMX_2K = 0x1p116f, SEMX_2MK = 0x1p-116f;
// "Logit" of the class slots past C: below every real logit, and so large that subtracting a maximum or adding K does not change
// it - the operand is the SAME in the four corners (ratio 1, no inf - inf) and its exponential is 0.
constexpr float SEMX_NOCLASS = -1e30f;

// every lane has a source with the controls used here (quad_perm, row_ror).  dpp_f: operand of an addition / maximum (the compiler
// folds it into v_add_f32_dpp when the old value is 0); dpp_mov: a value of its own (no initialisation of the destination)
template <int CTRL>
__device__ __forceinline__ float dpp_f(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, false));
}
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), CTRL, 0xF, 0xF, false));
}
// reduction over the 16 lanes that share lane & 3, result in all of them: rotations by 4 and 8 inside the 16-lane rows, then
// v_permlane16_swap / v_permlane32_swap of the value with itself (tools/ubench/permlane_probe.hip: [0] = rows a0 b0 a2 b2 /
// a0 a1 b0 b1, [1] = a1 b1 a3 b3 / a2 a3 b2 b3)
template <typename Op>
__device__ __forceinline__ float xlane_allreduce16(float v, Op op) {
  v = op(v, dpp_f<0x124>(v));  // row_ror:4
  v = op(v, dpp_f<0x128>(v));  // row_ror:8
  {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = op(__uint_as_float(r[0]), __uint_as_float(r[1]));
  }
  {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = op(__uint_as_float(r[0]), __uint_as_float(r[1]));
  }
  return v;
}

// Launch geometry: one workgroup = four waves on four CONSECUTIVE TILE ROWS of one image, walking the same range of tile columns in
// lockstep (one barrier per step).  A cell's gradient has four contributing tiles; instead of four atomics per (cell, class)
//   * the two right corners of a tile wait in LDS for the next tile of the walk and join its left corners, and
//   * the bottom-left total of a wave goes to the wave below through LDS and joins its top-left total,
// so that a step of four tiles issues five cell vectors of atomics instead of sixteen (the L2 retires roughly one fp32 atomic per
// channel and clock: 576 per tile were 105 of the kernel's 238 us).  Everything that leaves the workgroup is still an atomic add -
// seams between workgroups, image borders (where corner cells coincide) and complete cells alike need no case distinction.
struct SemXcGeom {
  int row_groups;   // ceil((Hc + 1) / 4)
  int x_splits;     // column ranges per tile row
  int x_per_split;  // ceil((Wc + 1) / x_splits)
};
__host__ __device__ inline SemXcGeom sem_xc_geom(int B, int Hc, int Wc, int target_workgroups) {
  SemXcGeom g;
  g.row_groups = (Hc + 1 + 3) / 4;
  const int per = B * g.row_groups;
  int xs = (target_workgroups + per - 1) / per;
  xs = xs < 1 ? 1 : (xs > Wc + 1 ? Wc + 1 : xs);
  g.x_per_split = (Wc + 1 + xs - 1) / xs;
  g.x_splits = (Wc + 1 + g.x_per_split - 1) / g.x_per_split;
  return g;
}

// NB: 16-class blocks per lane (3, 6 or 9: the smallest that holds C; class slots past C run along as zeros - no branch per block)
template <int MODE, int NB>
__global__ __launch_bounds__(256, SEMX_WGS) void sem_ce_xc_kernel(const float* __restrict__ sout0, const int64_t* __restrict__ labels0,
                                                           float* __restrict__ dsout0, StepAccum* __restrict__ acc, int view0,
                                                           int B, int Hc, int Wc, int C, int cs, SemXcGeom geom,
                                                           const float* __restrict__ sout1 = nullptr,
                                                           const int64_t* __restrict__ labels1 = nullptr,
                                                           float* __restrict__ dsout1 = nullptr) {
  // both views of the pair in one launch: the workgroups past B x row groups x column ranges belong to the second pointer set, view0 + 1
  const int wg_per_view = B * geom.row_groups * geom.x_splits;
  const bool second = (int)blockIdx.x >= wg_per_view;
  const float* __restrict__ sout = second ? sout1 : sout0;
  const int64_t* __restrict__ labels = second ? labels1 : labels0;
  float* __restrict__ dsout = second ? dsout1 : dsout0;
  const int view = view0 + (second ? 1 : 0);
  const int wg = (int)blockIdx.x - (second ? wg_per_view : 0);
  constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
  constexpr bool FWD = (MODE & 1) != 0, BWD = (MODE & 2) != 0;
  __shared__ float red[4];
  __shared__ __attribute__((aligned(16))) float s_tr[4][4 * SEMX_TP];
  __shared__ __attribute__((aligned(16))) float s_gi[4][64];
  __shared__ __attribute__((aligned(16))) float s_hist[4][4 * SEMX_HS];
  __shared__ float s_carry[4][2 * SEMX_HS];  // right corners of the previous tile of each wave: [top, bottom][class]
  __shared__ float s_down[4][2][SEMX_HS];  // bottom-left totals on their way to the wave below, double-buffered over the steps
  const int wave_in_blk = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  float* const tr = s_tr[wave_in_blk];
  float* const gis = s_gi[wave_in_blk];
  float* const hist = s_hist[wave_in_blk];
  const int H = 8 * Hc, W = 8 * Wc;
  const int TX = Wc + 1, TY = Hc + 1;
  const int xl = lane & 3, cl = lane >> 2;
  // this workgroup: image n, tile rows 4 rg .. 4 rg + 3 (one per wave), tile columns x_begin .. x_end
  const int xs = wg % geom.x_splits, rg = (wg / geom.x_splits) % geom.row_groups;
  const int n = wg / (geom.x_splits * geom.row_groups);
  const int tyi = rg * 4 + wave_in_blk;  // tile row index 0 .. TY - 1 (a wave past the last row only keeps the barriers company)
  const bool row_ok = tyi < TY;
  const int ty = tyi - 1;
  const int x_begin = xs * geom.x_per_split, x_end = min(TX, x_begin + geom.x_per_split);
  const int cy0 = max(ty, 0), cy1 = min(ty + 1, Hc - 1);
  const size_t cell0 = (size_t)n * Hc * Wc;
  // Rows live in PAIRS (i, 7 - i), i < 4: the two geometric progressions of a column advance together in one packed multiply.
  // Slot j = 8 xr + 2 i + h of a lane is the pixel (row h ? 7 - i : i, column xl + 4 xr).
  // interpolation weights of this lane's two columns (class role) and of its pixel (pixel role: slot pj = cl)
  const float wx1[2] = {(float)(2 * xl + 1) * 0.0625f, (float)(2 * xl + 9) * 0.0625f};
  const float wx0[2] = {1.f - wx1[0], 1.f - wx1[1]};
  const int pj = cl, pi = (pj >> 1) & 3, py = (pj & 1) ? 7 - pi : pi, px = xl + 4 * (pj >> 3);
  const float pwy1 = (float)(2 * py + 1) * 0.0625f, pwx1 = (float)(2 * px + 1) * 0.0625f;
  const float pw[4] = {(1.f - pwy1) * (1.f - pwx1), (1.f - pwy1) * pwx1, pwy1 * (1.f - pwx1), pwy1 * pwx1};
  const int y = 8 * ty + 4 + py;  // this lane's pixel row (pixel role)
  const bool y_ok = row_ok && y >= 0 && y < H;
  float nll_acc = 0.f;
  const float g = BWD ? acc->coef_sem / (float)acc->sem_cnt[view] : 0.f;
  const float gK = g * SEMX_2K;  // the sums carry 2^K
  const float wxs0[2] = {wx0[0] * SEMX_2MK, wx0[1] * SEMX_2MK}, wxs1[2] = {wx1[0] * SEMX_2MK, wx1[1] * SEMX_2MK};  // 2^-K: exact
  const DetTarget t_ds = det_resolve(dsout);   // (deterministic mode: fixed-point shadow of d(convSout))
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  if (BWD) {
#pragma unroll
    for (int i = 0; i < 4 * SEMX_HS / 256 + 1; ++i)
      if ((i * 64 + lane) * 4 < 4 * SEMX_HS) *reinterpret_cast<f32x4*>(hist + (i * 64 + lane) * 4) = zero4;
  }
  // One tile ahead: the label of this lane's pixel and the logits of its corner (k = xl, classes 16 blk + cl) are requested
  // between the two sweeps of the tile in front of them - two waves per SIMD do not hide a trip to memory by themselves.
  int64_t lab_n = C;
  float cc_n[NB];
#pragma unroll
  for (int blk = 0; blk < NB; ++blk) cc_n[blk] = SEMX_NOCLASS;
  auto request = [&](int txi) {
    const int tx = txi - 1;
    const int cx0 = max(tx, 0), cx1 = min(tx + 1, Wc - 1);
    const int x = 8 * tx + 4 + px;
    lab_n = C;
    if (y_ok && x >= 0 && x < W) lab_n = labels[((size_t)n * H + y) * W + x];
    const int ck = ((xl & 2) ? cy1 : cy0) * Wc + ((xl & 1) ? cx1 : cx0);
    const float* const cpk = sout + (cell0 + ck) * cs;
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
      cc_n[blk] = SEMX_NOCLASS;
      if (blk * 16 + cl < C) cc_n[blk] = cpk[blk * 16 + cl];
    }
  };
  if (row_ok && x_begin < x_end) request(x_begin);
  // the previous tile's right corners (lanes xl = 1, 3) are this tile's left corners: they wait in LDS (registers are short)
  float* const carry = s_carry[wave_in_blk] + (xl >> 1) * SEMX_HS + cl;
  if (BWD && (xl & 1)) {
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) carry[blk * 16] = 0.f;
  }
  // x_end - x_begin tiles, then one more step that only hands on the right corners of the last tile
  for (int txi = x_begin; txi <= x_end; ++txi) {
    const int tx = txi - 1;
    const int cx0 = max(tx, 0), cx1 = min(tx + 1, Wc - 1);
    const int cidx[4] = {cy0 * Wc + cx0, cy0 * Wc + cx1, cy1 * Wc + cx0, cy1 * Wc + cx1};
    // ---- pixel role: label of slot pj (a pixel outside the image holds C) ----
    const int label_l = (txi < x_end && (uint64_t)lab_n < (uint64_t)C) ? (int)lab_n : C;  // (the test of sem_count_kernel, on 64 bits)
    const bool counted = label_l < C;
    const bool work = __ballot(counted) != 0ull;  // wave-uniform: ignored / outside pixels contribute nothing
    float dres[NB];  // corner k = xl, classes 16 blk + cl: d(convSout) of this tile
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) dres[blk] = 0.f;
    if (work) {
      float cc[NB];
#pragma unroll
      for (int blk = 0; blk < NB; ++blk) cc[blk] = cc_n[blk];
      // the label logits of the four corners (pixel role), consumed after the forward sweep
      float lc[4] = {0.f, 0.f, 0.f, 0.f};
      if (FWD && counted) {
#pragma unroll
        for (int k = 0; k < 4; ++k) lc[k] = sout[(cell0 + cidx[k]) * cs + label_l];
      }
      // ---- class role: maximum over the classes of corner k = xl ----
      float mx = -INFINITY;
#pragma unroll
      for (int blk = 0; blk < NB; ++blk)
        mx = fmaxf(mx, cc[blk]);
      const float Mk = xlane_allreduce16(mx, [](float a, float b) { return fmaxf(a, b); });
      // ---- forward: anchors and ratios of the tile's exponentials in registers, partial sums over this lane's classes ----
      f32x2 E[NB][2][2];  // rows (0, 7) and the ratios (r, 1 / r) of each column: the gradient sweep runs the progressions again
      f32x2 SE[2][4];
#pragma unroll
      for (int xr = 0; xr < 2; ++xr)
#pragma unroll
        for (int i = 0; i < 4; ++i) SE[xr][i] = f32x2{0.f, 0.f};
#pragma unroll
      for (int blk = 0; blk < NB; ++blk) {
        {
          const float c2 = fmaf(cc[blk] - Mk, LOG2E, SEMX_K);  // (class slots past C: SEMX_NOCLASS)
          const float c00 = dpp_mov<0x00>(c2), c01 = dpp_mov<0x55>(c2), c10 = dpp_mov<0xAA>(c2), c11 = dpp_mov<0xFF>(c2);
          const float d0 = c01 - c00, d1 = c11 - c10;
#pragma unroll
          for (int xr = 0; xr < 2; ++xr) {
            const float T0 = fmaf(wx1[xr], d0, c00), T1 = fmaf(wx1[xr], d1, c10);
            const float D = T1 - T0;
            const float s = __builtin_amdgcn_fmed3f(D * 0.125f, -126.f, 126.f);
            const f32x2 R = {__builtin_amdgcn_exp2f(s), __builtin_amdgcn_exp2f(-s)};
            const f32x2 P0 = {__builtin_amdgcn_exp2f(fmaf(D, 0.0625f, T0)), __builtin_amdgcn_exp2f(fmaf(D, 0.9375f, T0))};  // rows 0, 7
            const f32x2 P1 = P0 * R, P2 = P1 * R, P3 = P2 * R;                                                              // 1, 6 ...
            E[blk][xr][0] = P0; E[blk][xr][1] = R;
            SE[xr][0] += P0; SE[xr][1] += P1; SE[xr][2] += P2; SE[xr][3] += P3;
          }
        }
        __builtin_amdgcn_sched_barrier(0);  // one class block at a time: interleaved blocks cost registers
      }
      if (!(SEMX_ABL & 8) && txi + 1 < x_end) request(txi + 1);
      // ---- sum over the classes: transpose the 16 partial sums of each lane through LDS; lane (xl, cl) gets slot pj = cl ----
#pragma unroll
      for (int xr = 0; xr < 2; ++xr)
#pragma unroll
        for (int h = 0; h < 2; ++h)
          *reinterpret_cast<f32x4*>(tr + xl * SEMX_TP + cl * 16 + xr * 8 + h * 4) =
              f32x4{SE[xr][2 * h][0], SE[xr][2 * h][1], SE[xr][2 * h + 1][0], SE[xr][2 * h + 1][1]};
      __builtin_amdgcn_wave_barrier();  // LDS executes one wave's instructions in order; this only pins the compiler
      float sp = 0.f;
#pragma unroll
      for (int src = 0; src < 16; ++src) sp += tr[xl * SEMX_TP + src * 16 + cl];
      __builtin_amdgcn_wave_barrier();
      // ---- pixel role: NLL, g / sum, label histogram ----
      // (cross-lane reads stay outside divergent code: a DPP source lane that is switched off delivers 0)
      const float Ms[4] = {dpp_mov<0x00>(Mk), dpp_mov<0x55>(Mk), dpp_mov<0xAA>(Mk), dpp_mov<0xFF>(Mk)};
      if (FWD && counted) {  // the label logit relative to the shift: sum_k w_k (corner_k[label] - M_k)
        float ll = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) ll = fmaf(pw[k], lc[k] - Ms[k], ll);
        nll_acc += (logf(sp) - SEMX_K * LN2) - ll;
      }
      if (BWD) {
        gis[xl * 16 + pj] = counted ? gK / sp : 0.f;  // g / sum; uncounted pixels: e is finite, d = 0
        // -g [c == label] w_k, summed per (corner, label) in the LDS histogram.  A segmentation map has one or two labels in most
        // 8x8 tiles, and 64 LDS atomics on one address are slow (~40 us of the launch): while the first pixel still to be done
        // shares its label with >= 8 pixels, the group is reduced on the DPP path instead (sums of w_y, w_x, w_y w_x over the
        // group give the four corner weights) and lands in the histogram as one 4-lane atomic.  The rest takes the atomics.
        unsigned long long todo = __ballot(counted);
        if (!(SEMX_ABL & 16)) {
          while (todo != 0ull) {
            const int lead = __ffsll((long long)todo) - 1;
            const int lab = __builtin_amdgcn_readlane(label_l, lead);
            const bool mine = counted && label_l == lab;
            const unsigned long long grp = __ballot(mine) & todo;
            if (__popcll(grp) < 8) break;
            todo &= ~grp;
            const float sy = wave_sum(mine ? pwy1 : 0.f), sx = wave_sum(mine ? pwx1 : 0.f), sxy = wave_sum(mine ? pwy1 * pwx1 : 0.f);
            const float cnt = (float)__popcll(grp);
            // corner k = 2 a + b: sum of wy_a wx_b, wy_0 = 1 - wy_1, wx_0 = 1 - wx_1
            const float sk = lane == 0 ? (cnt - sy) - (sx - sxy) : lane == 1 ? sx - sxy : lane == 2 ? sy - sxy : sxy;
            if (lane < 4 && !(SEMX_ABL & 2)) atomicAdd(hist + lane * SEMX_HS + lab, -g * sk);
          }
        }
        if (counted && ((todo >> lane) & 1ull)) {
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (!(SEMX_ABL & 2)) atomicAdd(hist + k * SEMX_HS + label_l, -g * pw[k]);
        }
        __builtin_amdgcn_wave_barrier();
        // Q_a[xr][i] = wy_a(rows i, 7 - i) g / sum of this lane's columns;  wy_1(i) = (2 i + 1) / 16 = wy_0(7 - i)
        f32x2 Q0[2][4], Q1[2][4];
#pragma unroll
        for (int xr = 0; xr < 2; ++xr)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(gis + xl * 16 + xr * 8 + h * 4);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              const int i = 2 * h + u;
              const float wa = (float)(2 * i + 1) * 0.0625f;   // wy_1(i) = wy_0(7 - i)
              const f32x2 gp = {v[2 * u], v[2 * u + 1]};
              Q1[xr][i] = gp * f32x2{wa, 1.f - wa};
              Q0[xr][i] = gp * f32x2{1.f - wa, wa};
            }
          }
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) {
          if (!(SEMX_ABL & 4)) {
            float V[2][2];  // [corner row a][column xr]
#pragma unroll
            for (int xr = 0; xr < 2; ++xr) {
              f32x2 P = E[blk][xr][0];
              const f32x2 R = E[blk][xr][1];
              f32x2 S0 = Q0[xr][0] * P, S1 = Q1[xr][0] * P;
#pragma unroll
              for (int i = 1; i < 4; ++i) {
                P = P * R;
                S0 = __builtin_elementwise_fma(Q0[xr][i], P, S0);
                S1 = __builtin_elementwise_fma(Q1[xr][i], P, S1);
              }
              V[0][xr] = S0[0] + S0[1];
              V[1][xr] = S1[0] + S1[1];
            }
            // the two columns with the x weights, then the four x lanes of the quad; lane xl keeps corner xl = 2 a + b
            float R[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) {
              float w0 = fmaf(wxs0[1], V[a][1], wxs0[0] * V[a][0]), w1 = fmaf(wxs1[1], V[a][1], wxs1[0] * V[a][0]);
              w0 += dpp_f<0xB1>(w0);  // quad_perm [1,0,3,2]
              w1 += dpp_f<0xB1>(w1);
              R[a] = (xl & 1) ? w1 : w0;
            }
            R[0] += dpp_f<0x4E>(R[0]);  // quad_perm [2,3,0,1]
            R[1] += dpp_f<0x4E>(R[1]);
            dres[blk] = (xl & 2) ? R[1] : R[0];
          }
          __builtin_amdgcn_sched_barrier(0);  // one class block at a time: interleaved blocks cost registers (spills at 256)
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int blk = 0; blk < NB; ++blk)
          dres[blk] += hist[xl * SEMX_HS + blk * 16 + cl];  // (class slots past C: untouched zeros)
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 4 * SEMX_HS / 256 + 1; ++i)
          if ((i * 64 + lane) * 4 < 4 * SEMX_HS) *reinterpret_cast<f32x4*>(hist + (i * 64 + lane) * 4) = zero4;
      }
    } else if (!(SEMX_ABL & 8) && row_ok && txi + 1 < x_end) {
      request(txi + 1);
    }
    if ((SEMX_ABL & 8) && row_ok && txi + 1 < x_end) request(txi + 1);
    if (BWD) {
      // ---- left corners = this tile's (xl = 0, 2) + the previous tile's right corners (xl = 1, 3): complete along x ----
      float tl[NB];
#pragma unroll
      for (int blk = 0; blk < NB; ++blk) tl[blk] = dres[blk];
      __builtin_amdgcn_wave_barrier();
      if (!(xl & 1)) {  // (reads first: LDS executes the wave's instructions in order)
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) tl[blk] += carry[blk * 16];
      }
      __builtin_amdgcn_wave_barrier();
      if (xl & 1) {
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) carry[blk * 16] = dres[blk];
      }
      __builtin_amdgcn_wave_barrier();
      // bottom-left (xl = 2): to the wave below, or out (last wave of the workgroup)
      const int par = (txi - x_begin) & 1;
      float* const dn = s_down[wave_in_blk][par];
      // cell of this lane's left corner (the step behind the last tile column: the right corners of that column)
      const int ckl = ((xl & 2) ? cy1 : cy0) * Wc + min(cx0, Wc - 1);
      float* const dk = dsout + (cell0 + ckl) * cs + cl;
      if (xl == 2) {
        if (wave_in_blk < 3 && tyi + 1 < TY) {  // (the wave below works on a tile row)
#pragma unroll
          for (int blk = 0; blk < NB; ++blk)
            dn[blk * 16 + cl] = tl[blk];
        } else if (row_ok && !(SEMX_ABL & 1)) {  // last wave of the workgroup / last tile row of the image
#pragma unroll
          for (int blk = 0; blk < NB; ++blk)
            if (blk * 16 + cl < C && tl[blk] != 0.f) facc_add(t_ds, dk + blk * 16, tl[blk]);
        }
      }
      __syncthreads();
      // top-left (xl = 0) + the bottom-left of the wave above: out
      if (xl == 0 && row_ok) {
        const float* const up = s_down[(wave_in_blk + 3) & 3][par];
#pragma unroll
        for (int blk = 0; blk < NB; ++blk)
          if (blk * 16 + cl < C) {
            const float v = tl[blk] + (wave_in_blk > 0 ? up[blk * 16 + cl] : 0.f);
            if (v != 0.f && !(SEMX_ABL & 1)) facc_add(t_ds, dk + blk * 16, v);
          }
      }
    }
  }
  if (FWD) {
    const float tot = block_sum_of_waves(wave_sum(nll_acc), red);
    if (threadIdx.x == 0) acc_add_loss(&acc->sem_sum[view], (double)tot);
  }
}

// API output only: materialise sem [N,C,H,W] = bilinear upsample of the NHWC convSout map.
__global__ void sem_upsample_nchw_kernel(const float* __restrict__ sout, float* __restrict__ out, int N, int Hc, int Wc,
                                         int H, int W, int C, int cs) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)N * C * H * W) return;
  const int x = (int)(idx % W), y = (int)((idx / W) % H), c = (int)((idx / ((long)W * H)) % C);
  const int n = (int)(idx / ((long)W * H * C));
  int y0, y1, x0, x1;
  float ly, lx;
  up_src(y, Hc, H, y0, y1, ly);
  up_src(x, Wc, W, x0, x1, lx);
  const float* b = sout + (size_t)n * Hc * Wc * cs + c;
  const float p00 = b[(size_t)(y0 * Wc + x0) * cs], p01 = b[(size_t)(y0 * Wc + x1) * cs];
  const float p10 = b[(size_t)(y1 * Wc + x0) * cs], p11 = b[(size_t)(y1 * Wc + x1) * cs];
  out[idx] = (1.f - ly) * ((1.f - lx) * p00 + lx * p01) + ly * ((1.f - lx) * p10 + lx * p11);
}

// autograd-compat path only: dsem [N,C,H,W] -> dsout NHWC (gather form, no atomics; dsout += ...)
__global__ void sem_upsample_bwd_nchw_kernel(const float* __restrict__ dsem, float* __restrict__ dsout, int N, int Hc,
                                             int Wc, int H, int W, int C, int cs) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total_cells = (long)N * C * Hc * Wc;
  // the launch is sized for N*C*H*W threads (same as the forward); only the first N*C*Hc*Wc do work
  if (idx >= total_cells) return;
  const int cx = (int)(idx % Wc), cy = (int)((idx / Wc) % Hc), c = (int)((idx / ((long)Wc * Hc)) % C);
  const int n = (int)(idx / ((long)Wc * Hc * C));
  const int sy = H / Hc, sx = W / Wc;
  float s = 0.f;
  for (int y = max(cy * sy - sy, 0); y < min(cy * sy + 2 * sy, H); ++y) {
    int y0, y1;
    float ly;
    up_src(y, Hc, H, y0, y1, ly);
    const float wy = (y0 == cy ? 1.f - ly : 0.f) + (y1 == cy ? ly : 0.f);
    if (wy == 0.f) continue;
    for (int x = max(cx * sx - sx, 0); x < min(cx * sx + 2 * sx, W); ++x) {
      int x0, x1;
      float lx;
      up_src(x, Wc, W, x0, x1, lx);
      const float wx = (x0 == cx ? 1.f - lx : 0.f) + (x1 == cx ? lx : 0.f);
      if (wx != 0.f) s += wy * wx * dsem[(((size_t)n * C + c) * H + y) * W + x];
    }
  }
  dsout[((size_t)n * Hc * Wc + cy * Wc + cx) * cs + c] += s;
}

// column sums of an NHWC matrix [rows][cs] -> out[C] (accumulated): convSout bias gradient.
// block = 256 threads = 64 float4 column quads x 4 row lanes (rows are read as contiguous float4 runs); a block owns
// COLSUM_ROWS rows; the row lanes meet in LDS, one atomic per column and block.  cs % 4 == 0, cs <= 256.
constexpr int COLSUM_ROWS = 512;
__device__ __forceinline__ void colsum_block(const float* __restrict__ m, float* __restrict__ out, int rows, int C, int cs, int bx,
                                             float4 (*red)[64]) {   // red: [4][64] float4 of LDS
  const int rl = threadIdx.x >> 6;
  const int r0 = bx * COLSUM_ROWS, r1 = min(r0 + COLSUM_ROWS, rows);
  for (int q = threadIdx.x & 63; q * 4 < cs; q += 64) {  // one pass per 256 columns (one for every shipped model)
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    int r = r0 + rl;
    for (; r + 28 < r1; r += 32) {  // latency-bound: eight row loads in flight per thread
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(m + (size_t)(r + 4 * u) * cs + q * 4);
#pragma unroll
      for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
    for (; r < r1; r += 4) {
      const float4 v = *reinterpret_cast<const float4*>(m + (size_t)r * cs + q * 4);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    red[rl][q & 63] = s;
  }
  __syncthreads();
  if (rl == 0)
    for (int q = threadIdx.x & 63; q * 4 < cs; q += 64) {
      // (columns beyond 256 would need one LDS round per pass; cs <= 256 is asserted on the host)
      const float4 a = red[0][q & 63], b = red[1][q & 63], c = red[2][q & 63], d = red[3][q & 63];
      const float t[4] = {(a.x + b.x) + (c.x + d.x), (a.y + b.y) + (c.y + d.y), (a.z + b.z) + (c.z + d.z),
                          (a.w + b.w) + (c.w + d.w)};
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (q * 4 + e < C) facc_add(out + q * 4 + e, t[e]);
    }
}
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ m0, float* __restrict__ out, int rows, int C,
                                                     int cs, const float* __restrict__ m1 = nullptr) {   // (blockIdx.y: matrix 0 / 1)
  __shared__ float4 red[4][64];
  colsum_block(blockIdx.y ? m1 : m0, out, rows, C, cs, blockIdx.x, red);
}

// ------------------------------------------------------------------------------------------------
// Device sampler for the sparse descriptor loss (distribution-level equivalent of the reference's
// numpy/torch CPU sampling; parity tests pass the reference's own indices instead).
// ------------------------------------------------------------------------------------------------
constexpr int SAMPLER_MAX_CELLS = 8192;  // 480x640 has 4800 cells; the sort runs on cap = max(2048, next power of two) keys

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

// One block of 1024 threads per image.  match_a/match_b: [B][n_match] cell indices (u + v*Wc).
__global__ __launch_bounds__(1024) void sample_matches_kernel(const float* __restrict__ Hn, const float* __restrict__ Hcell,
                                                              uint64_t seed_arg, const uint64_t* __restrict__ seed_dev,
                                                              int32_t* __restrict__ match_a, int32_t* __restrict__ match_b,
                                                              int Hc, int Wc, int n_match, int cap) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sampler_smem[];  // 12 bytes per key slot
  uint64_t* const key = reinterpret_cast<uint64_t*>(sampler_smem);  // (random key << 32) | cell_a
  int32_t* const cellb = reinterpret_cast<int32_t*>(key + cap);
  __shared__ float Hs[9];
  __shared__ int nvalid;
  const int img = blockIdx.x, tid = threadIdx.x;
  const int ncell = Hc * Wc;
  const uint64_t seed = seed_arg + (seed_dev ? *seed_dev : 0);  // captured steps keep the seed in device memory
  if (tid == 0) {
    // H_cell = inv(T) @ H @ T, T = [[2/Wc,0,-1],[0,2/Hc,-1],[0,0,1]]  (utils/homographies.py:270-276): the caller's host
    // matrix (the reference's own op sequence -> bit-identical matches) or the analytic form
    if (Hcell != nullptr) {
      for (int k = 0; k < 9; ++k) Hs[k] = Hcell[img * 9 + k];
    } else {
      float P[9];
      pixel_homography_analytic(Hn + img * 9, Hc, Wc, P);
      for (int k = 0; k < 9; ++k) Hs[k] = P[k];
    }
    nvalid = 0;
  }
  __syncthreads();
  for (int i = tid; i < cap; i += 1024) {
    uint64_t k = ~0ull;
    int32_t cb = 0;
    if (i < ncell) {
      const float u = (float)(i % Wc), v = (float)(i / Wc);
      float wu, wv;
      warp_point_exact(Hs, u, v, wu, wv);
      const float ub = rintf(wu), vb = rintf(wv);  // torch.round: half to even
      if (ub >= 0.f && ub <= (float)(Wc - 1) && vb >= 0.f && vb <= (float)(Hc - 1)) {
        const uint32_t r = (uint32_t)(splitmix64(seed ^ ((uint64_t)img << 40) ^ (uint64_t)i) >> 33);  // 31 bits
        k = ((uint64_t)r << 32) | (uint32_t)i;
        cb = (int)ub + (int)vb * Wc;
        atomicAdd(&nvalid, 1);
      }
    }
    key[i] = k;
    cellb[i] = cb;
  }
  __syncthreads();
  // bitonic sort of the keys (ascending): valid cells come first in random order
  for (int k2 = 2; k2 <= cap; k2 <<= 1) {
    for (int j = k2 >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < cap; i += 1024) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const bool up = (i & k2) == 0;
          const uint64_t a = key[i], b = key[ixj];
          if ((a > b) == up) {
            key[i] = b;
            key[ixj] = a;
          }
        }
      }
      __syncthreads();
    }
  }
  const int nv = nvalid;
  for (int j = tid; j < n_match; j += 1024) {
    int src = j;
    if (j >= nv) src = nv > 0 ? (int)(splitmix64(seed ^ 0xA5A5A5A5ull ^ ((uint64_t)img << 40) ^ (uint64_t)j) % (uint64_t)nv) : 0;
    const int ca = nv > 0 ? (int)(uint32_t)(key[src] & 0xFFFFFFFFull) : 0;
    match_a[(size_t)img * n_match + j] = ca;
    match_b[(size_t)img * n_match + j] = nv > 0 ? cellb[ca] : 0;
  }
}

__global__ void sample_nonmatches_kernel(uint64_t seed_arg, const uint64_t* __restrict__ seed_dev, int32_t* __restrict__ nm,
                                         long total, int Hc, int Wc) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const uint64_t seed = seed_arg + (seed_dev ? *seed_dev : 0);  // captured steps keep the seed in device memory
  const uint64_t r = splitmix64(seed ^ 0x5EED5EEDull ^ ((uint64_t)i * 0x9E3779B97F4A7C15ull));
  const float u1 = (float)(uint32_t)(r >> 40) * (1.f / 16777216.f);          // 24 bits -> [0,1)
  const float u2 = (float)(uint32_t)((r >> 16) & 0xFFFFFF) * (1.f / 16777216.f);
  const int u = min((int)floorf(u1 * Wc), Wc - 1), v = min((int)floorf(u2 * Hc), Hc - 1);
  nm[i] = u + v * Wc;
}

}  // namespace sspk
