// Dense descriptor loss: utils/utils.py:779-893 `descriptor_loss`, selected by
// model.dense_loss.enable (Train_model_heatmap_all.py:131-137, call :348-350).
//   dot[b,i,j]  = <desc[b,:,i], desc_w[b,:,j]>                 i, j: the Hc*Wc cells of the image / warped image
//   mask[b,i,j] = |centre(j) - H_b(centre(i))| <= descriptor_dist    (cell centres in pixels, :829-858)
//   loss  = sum (lamda_d mask max(0, 1 - dot) + (1 - mask) max(0, dot - 0.2)) valid_j / norm          (:880-890)
//   pos   = sum  lamda_d mask max(0, 1 - dot) / norm,  neg = sum (1 - mask) max(0, dot - 0.2) / norm  (no valid_j)
//   norm  = B (sum valid + 1) Hc Wc                                                                   (:884)
// dense_dots_kernel: the B x cells x cells Gram matrix on the fp32 matrix cores straight from the NHWC descriptor
// maps (operands are contiguous along K = 256 channels, no LDS), loss sums and d(total)/d(dot) in its epilogue;
// dense_grad_kernel: the two batched GEMMs of the backward, dDesc = C Desc_w and dDesc_w = C^T Desc.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "conv_mfma.hip.h"
#include "loss_kernels.hip.h"

namespace sspk {

struct DenseArgs {
  const float* da;     // [B][cells][256] normalised descriptors of the image
  const float* db;     // [B][cells][256] normalised descriptors of the warped image
  const float* hn;     // [B][3][3] homographies, normalised coordinates, image -> warped
  const float* valid;  // [B][cells] cell mask of the warped image (mask_3D_flattened)
  float* coef;         // [B][cells][cells] d total / d dot, or nullptr (no gradient)
  StepAccum* acc;
  int B, Hc, Wc;
  float lamda_d, dist;
  int multi_task;
};

// grid (cdiv(cells, 64), cdiv(cells, 64), B); block = 4 waves = 2 x 2 sub-tiles of 32 x 32
__global__ __launch_bounds__(256) void dense_dots_kernel(const DenseArgs a) {
  __shared__ float red[3][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int cells = a.Hc * a.Wc;
  const int b = blockIdx.z;
  const int i0 = blockIdx.y * 64 + (wave >> 1) * 32, j0 = blockIdx.x * 64 + (wave & 1) * 32;
  const float* pa = a.da + ((size_t)b * cells + min(i0 + li, cells - 1)) * 256 + lh * 4;
  const float* pb = a.db + ((size_t)b * cells + min(j0 + li, cells - 1)) * 256 + lh * 4;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll 4
  for (int g = 0; g < 32; ++g) {  // 8 channels per step: MFMA e pairs channels g*8 + e and g*8 + 4 + e
    const float4 av = *reinterpret_cast<const float4*>(pa + g * 8);
    const float4 bv = *reinterpret_cast<const float4*>(pb + g * 8);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc, 0, 0, 0);
  }
  // epilogue: lane holds column j = j0 + li and rows i = i0 + (r & 3) + 8 (r >> 2) + 4 lh
  const float H = (float)(a.Hc * 8), W = (float)(a.Wc * 8);
  const int j = j0 + li;
  const bool jok = j < cells;
  const float cyj = (float)((j / a.Wc) * 8 + 4), cxj = (float)((j % a.Wc) * 8 + 4);
  const float vj = jok ? a.valid[(size_t)b * cells + j] : 0.f;
  const float* h = a.hn + b * 9;
  const double norm = (double)a.B * (a.acc->mask_cnt[1] + 1.0) * (double)cells;  // mask_cnt[1] = valid.sum() (cell_mask_kernel)
  const float cscale = a.coef ? a.acc->coef_neg / (float)norm : 0.f;            // 0.5 exp(-eta_desc) or lambda_loss
  float s_pos = 0.f, s_neg = 0.f, s_loss = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = i0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
    if (i < cells && jok) {
      // centre of cell i -> normalised (x, y) -> homography -> pixels (normPts / warp_points / denormPts)
      const float cy = (float)((i / a.Wc) * 8 + 4), cx = (float)((i % a.Wc) * 8 + 4);
      const float ny = cy / H * 2.f - 1.f, nx = cx / W * 2.f - 1.f;
      const float X = h[0] * nx + h[1] * ny + h[2], Y = h[3] * nx + h[4] * ny + h[5], Z = h[6] * nx + h[7] * ny + h[8];
      const float py = (Y / Z + 1.f) * H / 2.f, px = (X / Z + 1.f) * W / 2.f;
      const float dy = cyj - py, dx = cxj - px;
      const bool m = sqrtf(dy * dy + dx * dx) <= a.dist;
      const float dot = acc[r];
      const float pos = m ? a.lamda_d * fmaxf(1.f - dot, 0.f) : 0.f;
      const float neg = m ? 0.f : fmaxf(dot - 0.2f, 0.f);
      s_pos += pos;
      s_neg += neg;
      s_loss += (pos + neg) * vj;
      if (a.coef) {
        const float d = m ? (dot < 1.f ? -a.lamda_d : 0.f) : (dot > 0.2f ? 1.f : 0.f);
        a.coef[((size_t)b * cells + i) * cells + j] = cscale * d * (a.multi_task ? 1.f : vj);
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s_pos += __shfl_xor(s_pos, o);
    s_neg += __shfl_xor(s_neg, o);
    s_loss += __shfl_xor(s_loss, o);
  }
  if (lane == 0) { red[0][wave] = s_pos; red[1][wave] = s_neg; red[2][wave] = s_loss; }
  __syncthreads();
  if (tid == 0) {
    // 1024 replicas of each sum (the StepAccum arrays of the sparse loss): same-address atomics stay rare
    const int rep = (blockIdx.x + 19 * blockIdx.y + 7 * blockIdx.z) & (SSP_DENSE_REPS - 1);
    acc_add_loss(&a.acc->pos_sum[rep], (double)((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])));
    acc_add_loss(&a.acc->neg_sum[rep], (double)((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])));
    acc_add_loss(&a.acc->dense_sum[rep], (double)((red[2][0] + red[2][1]) + (red[2][2] + red[2][3])));
  }
}

// Batched GEMM of the backward: out[b][m][n] = sum_k A[b](m, k) * Bm[b][k][n], n < 256.
//   A_T == false: A(m, k) = coef[b][m][k]   (dDesc   = C   Desc_w)
//   A_T == true : A(m, k) = coef[b][k][m]   (dDesc_w = C^T Desc)
// Block = 4 waves, tile 64 (m) x 64 (n), K chunks of 32 staged in LDS as [k][64 + 4]; grid (4, cdiv(cells, 64), B).
template <bool A_T>
__global__ __launch_bounds__(256) void dense_grad_kernel(const float* __restrict__ coef, const float* __restrict__ bm,
                                                         float* __restrict__ out, int cells) {
  constexpr int P = 68;
  __shared__ float sA[32 * P], sB[32 * P];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int b = blockIdx.z, m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const float* cb = coef + (size_t)b * cells * cells;
  const float* bb = bm + (size_t)b * cells * 256;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
  for (int k0 = 0; k0 < cells; k0 += 32) {
    __syncthreads();
    // A tile: 64 m x 32 k
    if (!A_T) {  // coef[m][k]: k contiguous -> thread (m = tid >> 2, 8 k's)
      const int m = tid >> 2, kq = (tid & 3) * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = k0 + kq + e;
        sA[(kq + e) * P + m] = (m0 + m < cells && k < cells) ? cb[(size_t)(m0 + m) * cells + k] : 0.f;
      }
    } else {     // coef[k][m]: m contiguous -> thread (k = tid >> 3, 8 m's)
      const int k = tid >> 3, mq = (tid & 7) * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e)
        sA[k * P + mq + e] = (k0 + k < cells && m0 + mq + e < cells) ? cb[(size_t)(k0 + k) * cells + m0 + mq + e] : 0.f;
    }
    {            // B tile: 32 k x 64 n, n contiguous
      const int k = tid >> 3, nq = (tid & 7) * 8;
      const bool ok = k0 + k < cells;
      const float4 v0 = ok ? *reinterpret_cast<const float4*>(bb + (size_t)(k0 + k) * 256 + n0 + nq) : make_float4(0, 0, 0, 0);
      const float4 v1 = ok ? *reinterpret_cast<const float4*>(bb + (size_t)(k0 + k) * 256 + n0 + nq + 4) : make_float4(0, 0, 0, 0);
      *reinterpret_cast<float4*>(sB + k * P + nq) = v0;
      *reinterpret_cast<float4*>(sB + k * P + nq + 4) = v1;
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 16; ++s)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sA[(2 * s + lh) * P + wm + li], sB[(2 * s + lh) * P + wn + li], acc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * lh;
    if (m < cells) out[((size_t)b * cells + m) * 256 + n0 + wn + li] = acc[r];
  }
}

// operator-level helpers (ssp_op_dense_loss): a private StepAccum with mask_cnt[1] = valid.sum(), coef_neg = scale
__global__ __launch_bounds__(256) void dense_op_prep_kernel(StepAccum* acc, const float* __restrict__ valid, int n, float scale) {
  __shared__ float red[256];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += valid[i];
  red[threadIdx.x] = s;
  for (int i = threadIdx.x; i < SSP_DENSE_REPS; i += 256) acc->pos_sum[i] = acc->neg_sum[i] = acc->dense_sum[i] = 0.0;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < 256; ++i) t += red[i];
    acc->mask_cnt[1] = t;
    acc->coef_neg = scale;
  }
}

__global__ void dense_op_finish_kernel(const StepAccum* acc, float* __restrict__ out3, int B, int cells) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double ps = 0, ns = 0, ls = 0;
  for (int i = 0; i < SSP_DENSE_REPS; ++i) {
    ps += acc->pos_sum[i];
    ns += acc->neg_sum[i];
    ls += acc->dense_sum[i];
  }
  const double norm = (double)B * (acc->mask_cnt[1] + 1.0) * (double)cells;
  out3[0] = (float)(ls / norm);
  out3[1] = (float)(ps / norm);
  out3[2] = (float)(ns / norm);
}

}  // namespace sspk
