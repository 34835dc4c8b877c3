// Wavefront-level loss kernels of the pair step (one wave64 per cell / per match).
//   detector : labels2Dto3D (utils/utils.py:408-440) + getMasks (Train_model_frontend_all.py:373-386)
//              + BCE(softmax) (Train_model_heatmap_all.py:173-178), forward and d/d semi in one pass
//   sparse descriptor loss: match term (pixelwise_contrastive_loss.py:160-206, bilinear grid_sample with
//              align_corners=True at normPts coordinates) and non-match term (:238-263, sparse_loss.py:154)
//   MultiTaskLoss (Train_model_heatmap_all.py:62-77)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "det.hip.h"

namespace sspk {

// Wave-wide reductions on the DPP data path (cross-lane moves inside the VALU, ~8 cycles each) instead of six
// ds_bpermute round trips through the LDS unit (~100 cycles each): xor 1, xor 2 inside the quads, mirror inside half
// rows and rows (every lane then holds its 16-lane row total), row_bcast15 / row_bcast31 accumulate the rows into lane
// 63, v_readlane broadcasts.  The result is identical in every lane.
template <typename Op>
__device__ __forceinline__ float wave_reduce(float v, float identity, Op op) {
  auto dpp = [&](float x, int ctrl_sel) {
    const int xi = __float_as_int(x), idn = __float_as_int(identity);
    int r;
    switch (ctrl_sel) {
      case 0: r = __builtin_amdgcn_update_dpp(idn, xi, 0xB1, 0xF, 0xF, false); break;   // quad_perm [1,0,3,2]
      case 1: r = __builtin_amdgcn_update_dpp(idn, xi, 0x4E, 0xF, 0xF, false); break;   // quad_perm [2,3,0,1]
      case 2: r = __builtin_amdgcn_update_dpp(idn, xi, 0x141, 0xF, 0xF, false); break;  // row_half_mirror
      case 3: r = __builtin_amdgcn_update_dpp(idn, xi, 0x140, 0xF, 0xF, false); break;  // row_mirror
      case 4: r = __builtin_amdgcn_update_dpp(idn, xi, 0x142, 0xA, 0xF, false); break;  // row_bcast15 -> rows 1, 3
      default: r = __builtin_amdgcn_update_dpp(idn, xi, 0x143, 0xC, 0xF, false); break; // row_bcast31 -> rows 2, 3
    }
    return __int_as_float(r);
  };
  v = op(v, dpp(v, 0));
  v = op(v, dpp(v, 1));
  v = op(v, dpp(v, 2));
  v = op(v, dpp(v, 3));
  v = op(v, dpp(v, 4));
  v = op(v, dpp(v, 5));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_sum(float v) {
  return wave_reduce(v, 0.f, [](float a, float b) { return a + b; });
}
__device__ __forceinline__ float wave_max(float v) {
  return wave_reduce(v, -__builtin_inff(), [](float a, float b) { return fmaxf(a, b); });
}
__device__ __forceinline__ float wave_prod(float v) {
  return wave_reduce(v, 1.f, [](float a, float b) { return a * b; });
}

// block (4 waves) sum of a per-wave value held by lane 0 of each wave; result valid in thread 0
__device__ __forceinline__ float block_sum_of_waves(float v, float* smem4) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) smem4[wave] = v;
  __syncthreads();
  return smem4[0] + smem4[1] + smem4[2] + smem4[3];
}

// Pairs per ssp_pair_step call (per-image accumulators of the sparse descriptor loss below; ssp_config.max_batch bounds a forward)
constexpr int SSP_MAX_PAIRS = 128;   // (StepAccum stays below the 64 KiB scratch contract of the ssp_op_* loss operators)
constexpr int SSP_DENSE_REPS = 1024;   // replica slots the dense descriptor loss spreads its three sums over (<= SSP_MAX_PAIRS * 16)
// Device-side accumulators / coefficients of one pair step (doubles for order-insensitive sums).
struct StepAccum {
  double det_sum[2];    // sum over cells of mask * sum_c BCE, per view
  double mask_cnt[2];   // mask.sum() per view
  double sem_sum[2];    // sum of NLL over non-ignored pixels
  double sem_cnt[2];    // number of non-ignored pixels
  double pos_sum[SSP_MAX_PAIRS * 16];   // per image x 16 replicas: sum_k max(0, 1 - <a,b>)
  double neg_sum[SSP_MAX_PAIRS * 16];   // per image x 16 replicas: sum max(0, <a,b> - 0.2)
  unsigned int nnz[SSP_MAX_PAIRS * 16]; // per image x 16 replicas: number of non-zero non-match hinges
  unsigned int nnz_img[SSP_MAX_PAIRS];  // (unused since the readers sum the 16 replicas themselves: nnz_of_image; kept for the layout)
  double dense_sum[SSP_DENSE_REPS];     // dense descriptor loss: replicas of sum (pos + neg) * valid (the first SSP_DENSE_REPS
                                        // slots of pos_sum / neg_sum hold the rest)
  float coef_det, coef_pos, coef_neg, coef_sem;  // d total / d (loss_det sum), d/d pos mean, d/d neg mean, d/d sem sum
};

static_assert(sizeof(StepAccum) <= 65536, "the ssp_op_* loss operators carve a StepAccum out of a 64 KiB scratch");

// Number of non-zero non-match hinges of an image: the sum of its 16 replica counters.  The readers (non-match backward, step end)
// add them up themselves - a one-block kernel between the forward and the backward descriptor kernels did it before and waited
// ~0.1 ms for a free slot beside the persistent segmentation-loss kernel of the other stream.
__device__ __forceinline__ unsigned nnz_of_image(const StepAccum* acc, int img) {
  unsigned nz = 0;
#pragma unroll
  for (int r = 0; r < 16; ++r) nz += acc->nnz[img * 16 + r];
  return nz;
}

// ---- cell masks: one wave per cell; lane = dy*8+dx --------------------------------------------
// blockIdx.y = view of the pair (second pointer set optional)
__global__ __launch_bounds__(256) void cell_mask_kernel(const float* __restrict__ mask2d0, float* __restrict__ cellmask0,
                                                        double* __restrict__ mask_cnt0, int B, int H, int W,
                                                        const float* __restrict__ mask2d1 = nullptr, float* __restrict__ cellmask1 = nullptr,
                                                        double* __restrict__ mask_cnt1 = nullptr) {
  const float* __restrict__ mask2d = blockIdx.y ? mask2d1 : mask2d0;
  float* __restrict__ cellmask = blockIdx.y ? cellmask1 : cellmask0;
  double* __restrict__ mask_cnt = blockIdx.y ? mask_cnt1 : mask_cnt0;
  __shared__ float red[4];
  const int Hc = H / 8, Wc = W / 8;
  const int lane = threadIdx.x & 63;
  const int ncell = B * Hc * Wc;
  float cnt = 0.f;
  for (int cell = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; cell < ncell; cell += (gridDim.x * blockDim.x) >> 6) {
    const int cx = cell % Wc, cy = (cell / Wc) % Hc, n = cell / (Wc * Hc);
    const float m = mask2d[((size_t)n * H + cy * 8 + (lane >> 3)) * W + cx * 8 + (lane & 7)];
    const float p = wave_prod(m);
    if (lane == 0) cellmask[cell] = p;
    cnt += p;
  }
  const float tot = block_sum_of_waves(cnt, red);
  if (threadIdx.x == 0) unsafeAtomicAdd(mask_cnt, (double)tot);
}

// ---- labels2Dto3D as an operator (parity tests): target [B,65,Hc,Wc] NCHW, same arithmetic as the
// fused detector kernel below ----------------------------------------------------------------------
__global__ __launch_bounds__(256) void labels2dto3d_kernel(const float* __restrict__ labels2d, float* __restrict__ target,
                                                           int B, int H, int W) {
  const int Hc = H / 8, Wc = W / 8;
  const int cell = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (cell >= B * Hc * Wc) return;
  const int cx = cell % Wc, cy = (cell / Wc) % Hc, n = cell / (Wc * Hc);
  const float lab = labels2d[((size_t)n * H + cy * 8 + (lane >> 3)) * W + cx * 8 + (lane & 7)];
  const float lsum = wave_sum(lab);
  float dust = 1.f - lsum;
  if (dust < 1.f) dust = 0.f;
  const float dn = lsum + dust;
  target[(((size_t)n * 65 + lane) * Hc + cy) * Wc + cx] = lab / dn;
  if (lane == 0) target[(((size_t)n * 65 + 64) * Hc + cy) * Wc + cx] = dust / dn;
}

// ---- detector loss fwd+bwd: one wave per cell --------------------------------------------------
// ypb: raw convPb output NHWC [cells][cs] (65 channels), BN affine applied here.
// dsemi (may be nullptr): d total / d semi, NHWC [cells][cs] (pads written 0).
// blockIdx.y != 0: the second pointer set, view + 1 (both views of the pair in one launch)
__global__ __launch_bounds__(256) void detector_loss_kernel(const float* __restrict__ ypb0, const float* __restrict__ scale0,
                                                            const float* __restrict__ shift0,
                                                            const float* __restrict__ labels2d0,
                                                            const float* __restrict__ cellmask0, float* __restrict__ dsemi0,
                                                            StepAccum* __restrict__ acc, int view0, int B, int H, int W,
                                                            int cs, const float* __restrict__ ypb1 = nullptr,
                                                            const float* __restrict__ scale1 = nullptr,
                                                            const float* __restrict__ shift1 = nullptr,
                                                            const float* __restrict__ labels2d1 = nullptr,
                                                            const float* __restrict__ cellmask1 = nullptr,
                                                            float* __restrict__ dsemi1 = nullptr) {
  const bool v1 = blockIdx.y != 0;
  const float* __restrict__ ypb = v1 ? ypb1 : ypb0;
  const float* __restrict__ scale = v1 ? scale1 : scale0;
  const float* __restrict__ shift = v1 ? shift1 : shift0;
  const float* __restrict__ labels2d = v1 ? labels2d1 : labels2d0;
  const float* __restrict__ cellmask = v1 ? cellmask1 : cellmask0;
  float* __restrict__ dsemi = v1 ? dsemi1 : dsemi0;
  const int view = view0 + (v1 ? 1 : 0);
  __shared__ float red[4];
  const int Hc = H / 8, Wc = W / 8;
  const int lane = threadIdx.x & 63;
  const int ncell = B * Hc * Wc;
  const float sc_l = scale[lane], sh_l = shift[lane], sc_d = scale[64], sh_d = shift[64];
  const float coef0 = (dsemi != nullptr) ? acc->coef_det / ((float)acc->mask_cnt[view] + 1e-5f) : 0.f;
  float lsum_acc = 0.f;
  for (int cell = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; cell < ncell; cell += (gridDim.x * blockDim.x) >> 6) {
    const int cx = cell % Wc, cy = (cell / Wc) % Hc, n = cell / (Wc * Hc);
    // labels2Dto3D: channel = dy*8+dx ; dustbin = 1 - sum, zeroed when < 1 ; renormalise
    const float lab = labels2d[((size_t)n * H + cy * 8 + (lane >> 3)) * W + cx * 8 + (lane & 7)];
    const float lsum = wave_sum(lab);
    float dust = 1.f - lsum;
    if (dust < 1.f) dust = 0.f;
    const float dn = lsum + dust;
    const float t = lab / dn, td = dust / dn;
    // logits
    const float* yp = ypb + (size_t)cell * cs;
    const float s = fmaf(yp[lane], sc_l, sh_l);
    const float sd = fmaf(yp[64], sc_d, sh_d);
    const float mx = fmaxf(wave_max(s), sd);
    const float e = expf(s - mx), ed = expf(sd - mx);
    const float inv = 1.f / (wave_sum(e) + ed);
    const float p = e * inv, pd = ed * inv;
    // BCELoss: -(t*max(log p,-100) + (1-t)*max(log(1-p),-100))
    const float l = -(t * fmaxf(logf(p), -100.f) + (1.f - t) * fmaxf(log1pf(-p), -100.f));
    const float ld = -(td * fmaxf(logf(pd), -100.f) + (1.f - td) * fmaxf(log1pf(-pd), -100.f));
    const float m = cellmask[cell];
    lsum_acc += (wave_sum(l) + ld) * m;
    if (dsemi != nullptr) {
      const float coef = coef0 * m;
      // BCE backward: (p - t) / max(p*(1-p), 1e-12) ; softmax backward: p_j * (g_j - sum_c g_c p_c)
      const float g = (p - t) / fmaxf(p * (1.f - p), 1e-12f);
      const float gd = (pd - td) / fmaxf(pd * (1.f - pd), 1e-12f);
      const float dot = wave_sum(g * p) + gd * pd;
      float* dp = dsemi + (size_t)cell * cs;
      dp[lane] = coef * p * (g - dot);
      if (lane == 0) dp[64] = coef * pd * (gd - dot);
      if (lane >= 1 && lane < cs - 64) dp[64 + lane] = 0.f;
    }
  }
  const float tot = block_sum_of_waves(lsum_acc, red);
  if (threadIdx.x == 0) acc_add_loss(&acc->det_sum[view], (double)tot);
}

// ---- sparse descriptor loss --------------------------------------------------------------------
// desc_a/desc_b: normalised descriptors NHWC [B][Hc*Wc][256].
// match term: one wave per match. Coordinates follow the reference bit-for-bit:
//   g = u / Wc * 2 - 1 (normPts, utils/utils.py:745-755); ix = ((g + 1) / 2) * (Wc - 1) (grid_sample,
//   align_corners=True) -> bilinear blend of 4 cells.
struct Bilin {
  int i00, i01, i10, i11;
  float w00, w01, w10, w11;
};
__device__ __forceinline__ Bilin bilin_setup(int cell, int Hc, int Wc) {
  const float u = (float)(cell % Wc), v = (float)(cell / Wc);
  const float gx = u / (float)Wc * 2.f - 1.f, gy = v / (float)Hc * 2.f - 1.f;
  const float ix = ((gx + 1.f) / 2.f) * (float)(Wc - 1), iy = ((gy + 1.f) / 2.f) * (float)(Hc - 1);
  const float fx = floorf(ix), fy = floorf(iy);
  const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
  const float ax = ix - fx, ay = iy - fy;
  Bilin b;
  // zeros padding: out-of-range corners contribute nothing (weight 0, index clamped)
  const bool vx0 = x0 >= 0 && x0 < Wc, vx1 = x1 >= 0 && x1 < Wc, vy0 = y0 >= 0 && y0 < Hc, vy1 = y1 >= 0 && y1 < Hc;
  const int cx0 = min(max(x0, 0), Wc - 1), cx1 = min(max(x1, 0), Wc - 1);
  const int cy0 = min(max(y0, 0), Hc - 1), cy1 = min(max(y1, 0), Hc - 1);
  b.i00 = cy0 * Wc + cx0; b.i01 = cy0 * Wc + cx1; b.i10 = cy1 * Wc + cx0; b.i11 = cy1 * Wc + cx1;
  b.w00 = (vx0 && vy0) ? (1.f - ax) * (1.f - ay) : 0.f;
  b.w01 = (vx1 && vy0) ? ax * (1.f - ay) : 0.f;
  b.w10 = (vx0 && vy1) ? (1.f - ax) * ay : 0.f;
  b.w11 = (vx1 && vy1) ? ax * ay : 0.f;
  return b;
}
// method "1d" of match_loss (pixelwise_contrastive_loss.py:185-188): index_select at the cell itself - one corner, weight 1
__device__ __forceinline__ Bilin bilin_cell(int cell) {
  Bilin b;
  b.i00 = b.i01 = b.i10 = b.i11 = cell;
  b.w00 = 1.f; b.w01 = b.w10 = b.w11 = 0.f;
  return b;
}
// variants of the sparse descriptor loss (sparse_loss.py:76-77; every shipped config: method "2d", dist "cos" = flags 0)
constexpr int DESC_METHOD_1D = 1;   // matches sampled by index_select instead of the bilinear grid_sample
constexpr int DESC_EUCLIDEAN = 2;   // squared distance / (max(0, ||a - b|| - 0.2))^2 instead of the hinges on the dot product
__device__ __forceinline__ float4 f4_fma(float w, float4 a, float4 acc) {
  return make_float4(fmaf(w, a.x, acc.x), fmaf(w, a.y, acc.y), fmaf(w, a.z, acc.z), fmaf(w, a.w, acc.w));
}
__device__ __forceinline__ void atomic_add4(float* p, float4 v) {
  facc_add(p, v.x);
  facc_add(p + 1, v.y);
  facc_add(p + 2, v.z);
  facc_add(p + 3, v.w);
}


// Work split of the sparse descriptor-loss kernels: one wave per match, 4 per block, and the blocks of ONE image stay on
// ONE XCD (hardware dispatches block b to XCD b % 8): image = 8 * round + xcd.  The kernels gather 1 KB descriptor rows of
// the image's 1.2 MB descriptor map; with the images spread over all XCDs every 4 MB L2 saw eight maps at once and the
// gathers fell through to the Infinity Cache / HBM (1.9 GB fetched per launch for 39 MB of descriptors).
// grid: desc_grid(B, n_match).  Returns the global match index (image * n_match + k) or -1.
static inline int desc_grid(int B, int n_match) { return ((B + 7) / 8) * 8 * ((n_match + 3) / 4); }
__device__ __forceinline__ int desc_wave_of_block(int B, int n_match, int& img) {
  const int bpi = (n_match + 3) >> 2;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int round = slot / bpi;
  img = round * 8 + xcd;
  const int k = (slot - round * bpi) * 4 + (threadIdx.x >> 6);
  return (img < B && k < n_match) ? img * n_match + k : -1;
}
// Channel mapping: lane l holds channels l, l+64, l+128, l+192, so every load AND every atomic instruction of a wave
// covers 256 contiguous bytes (with 4 consecutive channels per lane the scatter hit each cache line 4 times).
// EUC (compile time: the shipped form keeps its instruction stream): dist "euclidean"; flags & DESC_METHOD_1D (run time): method "1d"
template <bool BWD, bool EUC = false>
__global__ __launch_bounds__(256) void desc_match_kernel(const float* __restrict__ desc_a, const float* __restrict__ desc_b,
                                                         const int32_t* __restrict__ match_a,
                                                         const int32_t* __restrict__ match_b, float* __restrict__ dd_a,
                                                         float* __restrict__ dd_b, StepAccum* __restrict__ acc, int B,
                                                         int Hc, int Wc, int n_match, int flags) {
  const DetTarget t_a = det_resolve(dd_a), t_b = det_resolve(dd_b);   // (deterministic mode: fixed-point shadows)
  int img;
  const int w = desc_wave_of_block(B, n_match, img);  // one wave per match, images pinned to XCDs
  const int lane = threadIdx.x & 63;
  if (w < 0) return;
  const bool m1d = (flags & DESC_METHOD_1D) != 0;   // wave-uniform
  constexpr bool euc = EUC;
  const size_t base = (size_t)img * Hc * Wc * 256 + lane;
  const Bilin ba = m1d ? bilin_cell(match_a[w]) : bilin_setup(match_a[w], Hc, Wc);
  const Bilin bb = m1d ? bilin_cell(match_b[w]) : bilin_setup(match_b[w], Hc, Wc);
  float va[4] = {0.f, 0.f, 0.f, 0.f}, vb[4] = {0.f, 0.f, 0.f, 0.f};
  // torch accumulates nw, ne, sw, se in this order
  const int ia[4] = {ba.i00, ba.i01, ba.i10, ba.i11}, ib[4] = {bb.i00, bb.i01, bb.i10, bb.i11};
  const float wa[4] = {ba.w00, ba.w01, ba.w10, ba.w11}, wb[4] = {bb.w00, bb.w01, bb.w10, bb.w11};
  if (m1d) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      va[j] = desc_a[base + (size_t)ia[0] * 256 + 64 * j];
      vb[j] = desc_b[base + (size_t)ib[0] * 256 + 64 * j];
    }
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k)   // (all 32 row loads of a wave in flight)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        va[j] = fmaf(wa[k], desc_a[base + (size_t)ia[k] * 256 + 64 * j], va[j]);
        vb[j] = fmaf(wb[k], desc_b[base + (size_t)ib[k] * 256 + 64 * j], vb[j]);
      }
  }
  // cos: max(0, 1 - <a, b>) (pixelwise_contrastive_loss.py:200-204);  euclidean: |a - b|^2 (:205-206)
  float term;
  if (!euc) {
    term = fmaxf(1.f - wave_sum(va[0] * vb[0] + va[1] * vb[1] + va[2] * vb[2] + va[3] * vb[3]), 0.f);
  } else {
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) q = fmaf(va[j] - vb[j], va[j] - vb[j], q);
    term = wave_sum(q);
  }
  // (the gradient of the match term does not depend on any sum over the batch - coef_pos is set by step_begin_kernel - so a training
  // step runs the BWD instantiation alone: loss sum and scatter from one gather of the eight corner rows)
  if (lane == 0) acc_add_loss(&acc->pos_sum[img * 16 + ((w >> 2) & 15)], (double)term);  // 16 replicas / image
  if (BWD && (euc || term > 0.f)) {
    const float c = acc->coef_pos / ((float)n_match * (float)B) * (euc ? 2.f : -1.f);  // d total / d <a, b>  resp.  d total / d (a - b) / (a - b)
#pragma unroll
    for (int k = 0; k < 4; ++k) {   // (method "1d": the weights of corners 1 - 3 are 0)
      if (wa[k] != 0.f) {
#pragma unroll
        for (int j = 0; j < 4; ++j) facc_add(t_a, dd_a + base + (size_t)ia[k] * 256 + 64 * j, wa[k] * c * (euc ? va[j] - vb[j] : vb[j]));
      }
      if (wb[k] != 0.f) {
#pragma unroll
        for (int j = 0; j < 4; ++j) facc_add(t_b, dd_b + base + (size_t)ib[k] * 256 + 64 * j, wb[k] * c * (euc ? vb[j] - va[j] : va[j]));
      }
    }
  }
}

// non-match term, forward: one wave per match k; 4 non-matches per iteration (16 lanes each, 16 channels per lane).
// EUC: dist "euclidean" - `dots` holds ||a - b|| and the term is (max(0, d - 0.2))^2 (pixelwise_contrastive_loss.py:249-258).
// a-side = integer-cell gather of image a at match_a[k] (sparse_loss.py:55-58,245), b-side = nonmatch_b.
// The dot products are kept (12.8 MB at B = 32) so that the backward only touches the hard negatives.
template <bool EUC = false>
__global__ __launch_bounds__(256) void desc_nonmatch_fwd_kernel(const float* __restrict__ desc_a,
                                                                const float* __restrict__ desc_b,
                                                                const int32_t* __restrict__ match_a,
                                                                const int32_t* __restrict__ nonmatch_b,
                                                                float* __restrict__ dots, StepAccum* __restrict__ acc,
                                                                int B, int Hc, int Wc, int n_match, int n_non, int flags) {
  int img;
  const int w = desc_wave_of_block(B, n_match, img);  // one wave per match, images pinned to XCDs
  const int lane = threadIdx.x & 63;
  if (w < 0) return;
  constexpr bool euc = EUC;
  (void)flags;
  const int grp = lane >> 4, l16 = lane & 15;
  const size_t ibase = (size_t)img * Hc * Wc * 256;
  const float* ap = desc_a + ibase + (size_t)match_a[w] * 256 + l16 * 4;
  float4 a[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const float4*>(ap + i * 64);
  float hsum = 0.f;
  unsigned cnt = 0;
  const int32_t* nm = nonmatch_b + (size_t)w * n_non;
  for (int j0 = 0; j0 < n_non; j0 += 4) {
    const int j = j0 + grp;
    const bool valid = j < n_non;
    const int bi = valid ? nm[j] : 0;
    const float* bp = desc_b + ibase + (size_t)bi * 256 + l16 * 4;
    float d = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 b = *reinterpret_cast<const float4*>(bp + i * 64);
      if (!euc) {
        d += a[i].x * b.x + a[i].y * b.y + a[i].z * b.z + a[i].w * b.w;
      } else {
        const float ex = a[i].x - b.x, ey = a[i].y - b.y, ez = a[i].z - b.z, ew = a[i].w - b.w;
        d += ex * ex + ey * ey + ez * ez + ew * ew;
      }
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) d += __shfl_xor(d, o);
    if (euc) d = sqrtf(d);   // (a - b).norm(2, 1): pixelwise_contrastive_loss.py:249
    if (l16 == 0 && valid) {
      if (dots != nullptr) dots[(size_t)w * n_non + j] = d;
      float h = fmaxf(d - 0.2f, 0.f);
      if (euc) h *= h;
      hsum += h;
      cnt += (h != 0.f) ? 1u : 0u;
    }
  }
  hsum = wave_sum(hsum);
  const float c = wave_sum((float)cnt);
  if (lane == 0) {
    const int rep = img * 16 + ((w >> 2) & 15);  // 16 replicas / image: ~60 instead of 1000 atomics per address
    acc_add_loss(&acc->neg_sum[rep], (double)hsum);
    atomicAdd(&acc->nnz[rep], (unsigned)(c + 0.5f));
  }
}

// non-match term, backward: one wave per match k reads its stored dot products and walks only the hard negatives
// (dot > 0.2).  Lane l holds channels l, l+64, l+128, l+192: loads and atomics are 256 contiguous bytes per wave.
template <bool EUC = false>
__global__ __launch_bounds__(256) void desc_nonmatch_bwd_kernel(const float* __restrict__ desc_a,
                                                                const float* __restrict__ desc_b,
                                                                const int32_t* __restrict__ match_a,
                                                                const int32_t* __restrict__ nonmatch_b,
                                                                const float* __restrict__ dots, float* __restrict__ dd_a,
                                                                float* __restrict__ dd_b, const StepAccum* __restrict__ acc,
                                                                int B, int Hc, int Wc, int n_match, int n_non, int flags) {
  constexpr bool euc = EUC;  // d term / d a = 2 (d - 0.2) (a - b) / d instead of b
  (void)flags;
  const DetTarget t_a = det_resolve(dd_a), t_b = det_resolve(dd_b);   // (deterministic mode: fixed-point shadows)
  int img;
  const int w = desc_wave_of_block(B, n_match, img);  // one wave per match, images pinned to XCDs
  const int lane = threadIdx.x & 63;
  if (w < 0) return;
  const size_t ibase = (size_t)img * Hc * Wc * 256 + lane;
  const float wgt = acc->coef_neg / (((float)nnz_of_image(acc, img) + 1.f) * (float)B);
  const int32_t* nm = nonmatch_b + (size_t)w * n_non;
  const float* dk = dots + (size_t)w * n_non;
  const int ma = match_a[w];
  float a[4], ga[4] = {0.f, 0.f, 0.f, 0.f};
  bool have_a = false;
  for (int j0 = 0; j0 < n_non; j0 += 64) {
    const int j = j0 + lane;
    const bool hard = j < n_non && (dk[j] - 0.2f) > 0.f;
    unsigned long long mask = __ballot(hard);
    if (mask != 0ull && !have_a) {
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = desc_a[ibase + (size_t)ma * 256 + 64 * i];
      have_a = true;
    }
    while (mask != 0ull) {
      const int bit = __ffsll((long long)mask) - 1;
      mask &= mask - 1;
      const int bi = nm[j0 + bit];  // wave-uniform
      if (!euc) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float bv = desc_b[ibase + (size_t)bi * 256 + 64 * i];
          ga[i] = fmaf(wgt, bv, ga[i]);
          facc_add(t_b, dd_b + ibase + (size_t)bi * 256 + 64 * i, wgt * a[i]);
        }
      } else {
        const float dn = dk[j0 + bit];  // wave-uniform, > 0.2
        const float sgrad = wgt * 2.f * (dn - 0.2f) / dn;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float e = a[i] - desc_b[ibase + (size_t)bi * 256 + 64 * i];
          ga[i] = fmaf(sgrad, e, ga[i]);
          facc_add(t_b, dd_b + ibase + (size_t)bi * 256 + 64 * i, -sgrad * e);
        }
      }
    }
  }
  if (have_a) {
#pragma unroll
    for (int i = 0; i < 4; ++i) facc_add(t_a, dd_a + ibase + (size_t)ma * 256 + 64 * i, ga[i]);
  }
}

// ---- MultiTaskLoss coefficients (before the loss kernels) and scalars / eta gradient (after) ----
__global__ void step_begin_kernel(StepAccum* acc, const float* __restrict__ eta, int multi_task, float lambda_loss,
                                  float lamda_d, int semantic, int B) {
  // one wave (launched with 64 threads): the replica slots in use (16 per image, at least the dense loss's 1024) are cleared in
  // parallel (a single thread took 17 us)
  if (blockIdx.x != 0) return;
  const int nclear = max(B * 16, SSP_DENSE_REPS);
  for (int i = threadIdx.x; i < nclear; i += blockDim.x) {
    acc->pos_sum[i] = acc->neg_sum[i] = 0.0;
    if (i < SSP_DENSE_REPS) acc->dense_sum[i] = 0.0;
    acc->nnz[i] = 0u;
  }
  if (threadIdx.x != 0) return;
  for (int i = 0; i < 2; ++i) acc->det_sum[i] = acc->mask_cnt[i] = acc->sem_sum[i] = acc->sem_cnt[i] = 0.0;
  if (multi_task) {
    acc->coef_det = expf(-eta[0]);
    acc->coef_pos = 0.5f * expf(-eta[1]);
    acc->coef_neg = 0.5f * expf(-eta[1]);
    acc->coef_sem = expf(-eta[2]);
  } else {  // uniform sum (Train_model_heatmap_all.py:363-365)
    acc->coef_det = 1.f;
    acc->coef_pos = lambda_loss * lamda_d;
    acc->coef_neg = lambda_loss;
    acc->coef_sem = 1.f;
  }
}

__global__ void step_end_kernel(const StepAccum* acc, const float* __restrict__ eta, float* __restrict__ deta,
                                float* __restrict__ scal, int B, int n_match, int multi_task, float lambda_loss,
                                float lamda_d, int semantic, int train, int dense, int cells, int nviews) {
  // one wave (launched with 64 threads): lane i reduces the replicas of image i (sparse loss) or a slice of the 1024
  // replica slots (dense loss); thread 0 then combines in the order of the former single-thread loop
  if (blockIdx.x != 0) return;
  __shared__ float s_p[SSP_MAX_PAIRS], s_q[SSP_MAX_PAIRS];
  __shared__ double s_d[3][64];
  const int li = threadIdx.x;
  if (lambda_loss > 0.f && dense) {
    double ps = 0, ns = 0, ls = 0;
    for (int i = li; i < SSP_DENSE_REPS; i += 64) {
      ps += acc->pos_sum[i];
      ns += acc->neg_sum[i];
      ls += acc->dense_sum[i];
    }
    s_d[0][li] = ps; s_d[1][li] = ns; s_d[2][li] = ls;
  } else if (lambda_loss > 0.f) {
    for (int img = li; img < B; img += 64) {
      double ps = 0, ns = 0;
      for (int r = 0; r < 16; ++r) {
        ps += acc->pos_sum[img * 16 + r];
        ns += acc->neg_sum[img * 16 + r];
      }
      s_p[img] = (float)ps / (float)n_match;
      s_q[img] = (float)ns / ((float)nnz_of_image(acc, img) + 1.f);
    }
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  const float det0 = (float)acc->det_sum[0] / ((float)acc->mask_cnt[0] + 1e-5f);
  // single-view step (nviews == 1): the warped terms are the constants 0 of Train_model_heatmap_all.py:330-332
  const float det1 = nviews > 1 ? (float)acc->det_sum[1] / ((float)acc->mask_cnt[1] + 1e-5f) : 0.f;
  float pos = 0.f, neg = 0.f, ldesc = 0.f;
  if (lambda_loss > 0.f && dense) {  // utils/utils.py:884-890
    double ps = 0, ns = 0, ls = 0;
    for (int i = 0; i < 64; ++i) {
      ps += s_d[0][i];
      ns += s_d[1][i];
      ls += s_d[2][i];
    }
    const double norm = (double)B * (acc->mask_cnt[1] + 1.0) * (double)cells;
    pos = (float)(ps / norm);
    neg = (float)(ns / norm);
    ldesc = (float)(ls / norm);
  } else if (lambda_loss > 0.f) {
    for (int i = 0; i < B; ++i) {
      const float p = s_p[i], q = s_q[i];
      pos += p;
      neg += q;
      ldesc += lamda_d * p + q;
    }
    pos /= (float)B;
    neg /= (float)B;
    ldesc /= (float)B;
  }
  float sem0 = 0.f, sem1 = 0.f;
  if (semantic) {
    sem0 = (float)(acc->sem_sum[0] / acc->sem_cnt[0]);
    if (nviews > 1) sem1 = (float)(acc->sem_sum[1] / acc->sem_cnt[1]);
  }
  float loss;
  if (multi_task) {
    const float e0 = expf(-eta[0]), e1 = expf(-eta[1]), e2 = expf(-eta[2]);
    loss = (det0 + det1) * e0 + eta[0] + 0.5f * (pos + neg) * e1 + 0.5f * eta[1];
    if (semantic) loss += (sem0 + sem1) * e2 + eta[2];
    if (train) {
      deta[0] += 1.f - (det0 + det1) * e0;
      deta[1] += 0.5f - 0.5f * (pos + neg) * e1;
      if (semantic) deta[2] += 1.f - (sem0 + sem1) * e2;
    }
  } else {
    loss = det0 + det1 + sem0 + sem1;
    if (lambda_loss > 0.f) loss += lambda_loss * ldesc;
  }
  scal[0] = loss; scal[1] = det0; scal[2] = det1; scal[3] = ldesc; scal[4] = sem0; scal[5] = sem1;
  scal[6] = pos; scal[7] = neg; scal[8] = eta[0]; scal[9] = eta[1]; scal[10] = eta[2];
}

// operator-level helper: mean positive / negative distances of batch_descriptor_loss_sparse
// operator form of the detector loss (ssp_op_detector_loss): accumulator reset / final division
__global__ void detector_op_prep_kernel(StepAccum* acc) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  acc->det_sum[0] = acc->mask_cnt[0] = 0.0;
  acc->coef_det = 1.f;
}
__global__ void fill_affine_identity_kernel(float* p, int c) {  // p[0..c) = 1 (scale), p[c..2c) = 0 (shift)
  for (int i = threadIdx.x; i < 2 * c; i += blockDim.x) p[i] = i < c ? 1.f : 0.f;
}
__global__ void detector_op_finish_kernel(const StepAccum* acc, float* out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  out[0] = (float)acc->det_sum[0] / ((float)acc->mask_cnt[0] + 1e-5f);
}

// operator form of the segmentation loss (ssp_op_sem_loss)
__global__ void sem_op_prep_kernel(StepAccum* acc) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  acc->sem_sum[0] = acc->sem_cnt[0] = 0.0;
  acc->coef_sem = 1.f;
}
__global__ void sem_op_finish_kernel(const StepAccum* acc, float* out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  out[0] = (float)(acc->sem_sum[0] / acc->sem_cnt[0]);
}

// operator form of the sparse descriptor loss (ssp_op_sparse_loss): d total / d (mean positive term), d total / d (mean negative term)
__global__ void sparse_op_prep_kernel(StepAccum* acc, float coef_pos, float coef_neg) {
  acc->coef_pos = coef_pos;
  acc->coef_neg = coef_neg;
}
__global__ void sparse_loss_means_kernel(const StepAccum* acc, float* out, int B, int n_match) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float pos = 0.f, neg = 0.f;
  for (int i = 0; i < B; ++i) {
    double ps = 0, ns = 0;
    unsigned nz = 0;
    for (int r = 0; r < 16; ++r) {
      ps += acc->pos_sum[i * 16 + r];
      ns += acc->neg_sum[i * 16 + r];
      nz += acc->nnz[i * 16 + r];
    }
    pos += (float)ps / (float)n_match;
    neg += (float)ns / ((float)nz + 1.f);
  }
  out[0] = pos / (float)B;
  out[1] = neg / (float)B;
}

}  // namespace sspk
