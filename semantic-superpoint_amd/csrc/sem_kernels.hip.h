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
__global__ __launch_bounds__(256) void sem_count_kernel(const int64_t* __restrict__ labels0, const int64_t* __restrict__ labels1,
                                                        long n, int C, StepAccum* __restrict__ acc) {
  __shared__ float red[4];
  const int view = blockIdx.y;
  const int64_t* __restrict__ labels = view ? labels1 : labels0;
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
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ m, float* __restrict__ out, int rows, int C,
                                                     int cs) {
  __shared__ float4 red[4][64];
  const int rl = threadIdx.x >> 6;
  const int r0 = blockIdx.x * COLSUM_ROWS, r1 = min(r0 + COLSUM_ROWS, rows);
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
