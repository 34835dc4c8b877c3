// Winograd F(2x2, 3x3) convolution on the fp32 matrix cores (forward and data-gradient 3x3 convs with Cin % 16 == 0).
//
//   Y(2x2) = A^T [ sum_ci (G g G^T) (.) (B^T d B) ] A        d: 4x4 input patch, g: 3x3 filter   (Lavin & Gray 2016)
//
// The 16 element-wise products are 16 independent GEMMs [tiles x Cin] x [Cin x Cout]: 16 multiplies per 2x2 outputs
// instead of 36, i.e. 2.25x fewer MFMA FLOPs than the direct implicit GEMM of conv_mfma_kernel for the same
// (algorithmic) convolution, still in fp32 arithmetic throughout (the transforms only add and halve).
//
// Block = 512 threads = 8 waves (2 per SIMD), ONE block per CU (150 KB of LDS):
//   output tile 8x32 (WIDE) or 32x8 pixels = 64 Winograd tiles x 64 output channels;
//   wave w = (M-tile w >> 2 (32 tiles), N-tile (w >> 1) & 1 (32 channels), component half w & 1): 8 x f32x16 acc.
// Per 16-channel K-chunk:
//   1. the raw (TH+2) x (TW+2) halo (340 pixels x 16 channels) and the 64 KB of pre-transformed weights
//      (pack_weights_wino_kernel, [cob][chunk][component][g][h][64][4]) are prefetched into registers while the MFMAs
//      of the previous chunk run, then written to LDS (BatchNorm + ReLU of the producer applied once per element here);
//   2. thread (tile, channel quad, row half) reads 3 x 4 raw pixels from LDS, forms its two rows of V = B^T d B and
//      writes 8 components to sA[component][tile][16] (quad index XOR-swizzled by (tile >> 1) & 3 instead of padding);
//   3. 64 MFMAs per wave on fragments read with ds_read_b128 exactly like conv_mfma_kernel ("tap" -> "component").
// Epilogue: each lane holds 8 components (two rows of the 4x4 product matrix) of its 16 tiles; the two component
// halves write their partial outputs to two LDS staging tiles, which are summed on the way to the 16-byte stores;
// the BatchNorm partial sums are taken from the summed values.  Persistent XCD-aware grid and dual-problem launches
// as in conv_mfma_kernel.
#pragma once
#include <type_traits>
#include "conv_mfma.hip.h"

namespace sspk {

constexpr int WC = 16;                          // Winograd components
constexpr int WTILES = 64;                      // 2x2-output tiles per block
constexpr int WA_FLOATS = WC * WTILES * CK;     // transformed input chunk  (64 KB, swizzled, no padding)
constexpr int WB_FLOATS = WC * CK * NB;         // transformed weight chunk (64 KB)
constexpr int WHALO = 340;                      // (8+2) x (32+2) = (32+2) x (8+2) raw halo pixels
constexpr int WR_FLOATS = WHALO * CK;           // raw halo chunk (21.25 KB)
constexpr int WINO_LDS_BYTES = (WA_FLOATS + WB_FLOATS + WR_FLOATS) * 4;
constexpr int WINO_THREADS = 512;

// raw halo pixel (r, c) of an HC-column halo (HC even), channel quad q -> float offset in sR: the column pair c >> 1
// shares a 128-byte row and the slot inside it alternates from pair to pair, so that the stride-2 pixel reads of the
// transform spread over all banks; rows are HC * CK floats apart
__device__ __forceinline__ int wino_raw_off(int r, int c, int q, int HC) {
  return ((r * (HC >> 1) + (c >> 1)) * 2 + ((c ^ (c >> 1)) & 1)) * CK + q * 4;
}

// relu(v * scale + shift) of a loaded quad, 0 for zero-padding pixels: two packed fmas + one v_med3 per element
// (clamp to [0, +inf] or, for padding, to [0, 0]) instead of fma + max + select, on the staging path of the MFMA loop
__device__ __forceinline__ f32x4 bn_relu_quad(f32x4 v, f32x4 sc, f32x4 sh, bool padding) {
  const float hi = padding ? 0.f : __builtin_inff();
  v = pk4_fma(v, sc, sh);
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = __builtin_amdgcn_fmed3f(v[e], 0.f, hi);
  return v;
}

template <int IN_MODE, bool WIDE>
__global__ __launch_bounds__(WINO_THREADS) void conv_wino_kernel(const ConvArgs a) {
  constexpr int TTX = WIDE ? 16 : 4;            // tiles per block row
  constexpr int TH = WIDE ? 8 : 32, TW = WIDE ? 32 : 8;
  constexpr int HC = TW + 2;                    // halo columns
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sA = smem;
  float* sB = smem + WA_FLOATS;
  float* sR = smem + WA_FLOATS + WB_FLOATS;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar wave roles
  const int li = lane & 31, lh = lane >> 5;
  const int chalf = wave & 1, nt = (wave >> 1) & 1, mt = wave >> 2;

  // ---- work assignment (as conv_mfma_kernel) ----
  const int nslot = gridDim.x >> 3;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int per_cob = nslot / a.ncob;
  const int cob = slot % a.ncob, jj = slot / a.ncob;
  const int ntiles = a.N * a.tiles_y * a.tiles_x;
  const int xpp = 8 / a.nprob;
  const int prob = xcd / xpp, xl = xcd - prob * xpp;
  const int per_t = (ntiles + xpp - 1) / xpp;
  const int t_end = min(ntiles, (xl + 1) * per_t);
  int tile = xl * per_t + jj;
  if (jj >= per_cob || tile >= t_end) return;
  const float* const p_in = prob ? a.in2 : a.in;
  float* const p_out = prob ? a.out2 : a.out;
  const float* const p_scale = prob ? a.in_scale2 : a.in_scale;
  const float* const p_shift = prob ? a.in_shift2 : a.in_shift;
  double* const p_stats = prob ? a.stats2 : a.stats;

  // ---- staging roles ----
  const int q4 = tid & 3;
  // raw halo: items tid + 512 k (k < 3), item = pixel * 4 + quad
  int rrc[3], r_lds[3];  // (row | col << 8) of the item's halo pixel, its float offset in sR
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int p = (tid + WINO_THREADS * k) >> 2, r = p / HC, c = p - r * HC;
    rrc[k] = r | (c << 8);
    r_lds[k] = wino_raw_off(r, c, q4, HC);
  }
  const bool r2 = tid + 2 * WINO_THREADS < WHALO * 4;  // the third item exists
  // transform: (tile, quad, row half)
  const int t_tile = (tid >> 2) & 63, t_half = tid >> 8;
  const int t_ty = t_tile / TTX, t_tx = t_tile % TTX;
  const int pixb = a.in_cs * 4, rowb = a.W * pixb;
  f32x4 hreg[3], wreg[8];
  f32x4 psc = {1.f, 1.f, 1.f, 1.f}, psh = {0.f, 0.f, 0.f, 0.f};
  constexpr unsigned OOB = 0x80000000u;
  int ld_n, ld_ty0, ld_tx0;
  unsigned hoff[3];
  const size_t img_floats = (size_t)a.H * a.W * a.in_cs;
  __amdgpu_buffer_rsrc_t rsrc_in;
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wpk), 0, a.wpk_bytes, 0x00020000);
#define WINO_DECODE_TILE(T)                                                                              \
  {                                                                                                      \
    const int tx_ = (T) % a.tiles_x, t2_ = (T) / a.tiles_x;                                              \
    ld_tx0 = tx_ * TW;                                                                                   \
    ld_ty0 = (t2_ % a.tiles_y) * TH;                                                                     \
    ld_n = t2_ / a.tiles_y;                                                                              \
    _Pragma("unroll") for (int k = 0; k < 3; ++k) {                                                      \
      const int gy = ld_ty0 - 1 + (rrc[k] & 255), gx = ld_tx0 - 1 + (rrc[k] >> 8);                       \
      const bool ok = (k < 2 || r2) && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;     \
      hoff[k] = ok ? (unsigned)(gy * rowb + gx * pixb + (a.in_co + q4 * 4) * 4) : OOB;                   \
    }                                                                                                    \
    rsrc_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p_in) + (size_t)ld_n * img_floats, 0, \
                                                a.in_bytes, 0x00020000);                                 \
  }
#define WINO_ISSUE_LOADS(CHUNK)                                                                          \
  if (!(a.ablate & 1)) {                                                                                 \
    if (IN_MODE != 0) {                                                                                  \
      psc = *reinterpret_cast<const f32x4*>(p_scale + (CHUNK) * CK + q4 * 4);                            \
      psh = *reinterpret_cast<const f32x4*>(p_shift + (CHUNK) * CK + q4 * 4);                            \
    }                                                                                                    \
    const int soff_ = (CHUNK) * CK * 4;                                                                  \
    _Pragma("unroll") for (int k = 0; k < 3; ++k)                                                        \
      hreg[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_in, hoff[k], soff_, 0)); \
    const int wbase_ = (cob * a.nchunks + (CHUNK)) * WB_FLOATS * 4;                                      \
    _Pragma("unroll") for (int j = 0; j < 8; ++j)                                                        \
      wreg[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, tid * 16, wbase_ + j * 8192, 0)); \
  }

  WINO_DECODE_TILE(tile)
  WINO_ISSUE_LOADS(0)

  // MFMA fragment offsets: A = sA[(comp * 64 + tile) * 16 + ((quad ^ swz) * 4)], quad = 2 g + lh
  const int m_tile = mt * 32 + li;
  const int swz = (m_tile >> 1) & 3;
  const int a_off0 = (chalf * 8 * WTILES + m_tile) * CK + ((lh ^ swz) << 2);
  const int a_off1 = (chalf * 8 * WTILES + m_tile) * CK + (((2 + lh) ^ swz) << 2);
  const int b_off = (chalf * 8 * (CK / 8) * 2 * NB + lh * NB + nt * 32 + li) * 4;
  // transform offsets
  const int t_swz = (t_tile >> 1) & 3;
  float* const t_dst = sA + ((2 * t_half * 4) * WTILES + t_tile) * CK + ((q4 ^ t_swz) << 2);
  // raw pixels (row 2 t_ty + t_half + i, column 2 t_tx + j): columns 0 / 3 sit in slot (t_tx & 1) of the column pairs
  // t_tx / t_tx + 1, columns 1 / 2 in the other slot; rows are HC * CK floats apart
  const int t_src_s = wino_raw_off(2 * t_ty + t_half, 2 * t_tx, q4, HC);
  const int t_src_n = wino_raw_off(2 * t_ty + t_half, 2 * t_tx + 1, q4, HC);

  // per-thread BatchNorm partial sums of channel quad (tid & 15)
  f32x4 ssum = {0.f, 0.f, 0.f, 0.f}, ssq = {0.f, 0.f, 0.f, 0.f};
  const int co_l = cob * NB + nt * 32 + li;
  const float bias_v = (a.bias != nullptr && co_l < a.Cout) ? a.bias[co_l] : 0.f;

  for (;;) {  // ---- one output tile per iteration ----
    const int n = ld_n, ty0 = ld_ty0, tx0 = ld_tx0;
    unsigned hmask = 0;  // validity of the raw items in the registers now
#pragma unroll
    for (int k = 0; k < 3; ++k) hmask |= (hoff[k] != OOB ? 1u : 0u) << k;
    const int next_tile = tile + per_cob;
    const bool has_next = next_tile < t_end;

    f32x16 acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;

    for (int chunk = 0; chunk < a.nchunks; ++chunk) {
      __syncthreads();  // every wave has finished the MFMA reads (sA, sB) and the transform reads (sR)
      if (!(a.ablate & 2)) {
        // raw halo -> LDS with BatchNorm + ReLU of the producer (zero outside the image), weights -> LDS
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          if (k < 2 || r2) {
            f32x4 v = hreg[k];
            if (IN_MODE != 0) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = fmaxf(fmaf(v[e], psc[e], psh[e]), 0.f);
              if (!((hmask >> k) & 1u)) v = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            *reinterpret_cast<f32x4*>(sR + r_lds[k]) = v;
          }
        }
        f32x4* wdst = reinterpret_cast<f32x4*>(sB);
#pragma unroll
        for (int j = 0; j < 8; ++j) wdst[tid + WINO_THREADS * j] = wreg[j];
      }
      __syncthreads();
      if (!(a.ablate & 2)) {
        // two rows of V = B^T d B for (tile, quad): half 0 -> V rows 0, 1 from d rows 0..2; half 1 -> rows 2, 3 from 1..3
        f32x4 ta[4], tb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float* src = sR + ((j == 0 || j == 3) ? t_src_s : t_src_n) + (j >= 2 ? 2 * CK : 0);
          const f32x4 x = *reinterpret_cast<const f32x4*>(src);
          const f32x4 y = *reinterpret_cast<const f32x4*>(src + HC * CK);
          const f32x4 z = *reinterpret_cast<const f32x4*>(src + 2 * HC * CK);
          if (t_half == 0) {
            ta[j] = x - z;  // T0 = d0 - d2
            tb[j] = y + z;  // T1 = d1 + d2
          } else {
            ta[j] = y - x;  // T2 = d2 - d1
            tb[j] = x - z;  // T3 = d1 - d3
          }
        }
        *reinterpret_cast<f32x4*>(t_dst + 0 * WTILES * CK) = ta[0] - ta[2];
        *reinterpret_cast<f32x4*>(t_dst + 1 * WTILES * CK) = ta[1] + ta[2];
        *reinterpret_cast<f32x4*>(t_dst + 2 * WTILES * CK) = ta[2] - ta[1];
        *reinterpret_cast<f32x4*>(t_dst + 3 * WTILES * CK) = ta[1] - ta[3];
        *reinterpret_cast<f32x4*>(t_dst + 4 * WTILES * CK) = tb[0] - tb[2];
        *reinterpret_cast<f32x4*>(t_dst + 5 * WTILES * CK) = tb[1] + tb[2];
        *reinterpret_cast<f32x4*>(t_dst + 6 * WTILES * CK) = tb[2] - tb[1];
        *reinterpret_cast<f32x4*>(t_dst + 7 * WTILES * CK) = tb[1] - tb[3];
      }
      __syncthreads();
      {
        const bool last = chunk + 1 == a.nchunks;
        if (last && has_next) WINO_DECODE_TILE(next_tile)
        const int nxt = last ? 0 : chunk + 1;
        WINO_ISSUE_LOADS(nxt)
        __builtin_amdgcn_sched_barrier(0);
      }
      // ---- MFMA: 8 components x 2 k-groups x 4 k-pairs ----
      if (!(a.ablate & 8))
#pragma unroll
      for (int c = 0; c < 8; c += 2) {
#pragma unroll
        for (int g = 0; g < CK / 8; ++g) {
          const int ao = g ? a_off1 : a_off0;
          const float4 a0 = *reinterpret_cast<const float4*>(sA + ao + c * WTILES * CK);
          const float4 a1 = *reinterpret_cast<const float4*>(sA + ao + (c + 1) * WTILES * CK);
          const float4 b0 = *reinterpret_cast<const float4*>(sB + b_off + (c * (CK / 8) + g) * 2 * NB * 4);
          const float4 b1 = *reinterpret_cast<const float4*>(sB + b_off + ((c + 1) * (CK / 8) + g) * 2 * NB * 4);
          acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b0.x, acc[c], 0, 0, 0);
          acc[c + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b1.x, acc[c + 1], 0, 0, 0);
          acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b0.y, acc[c], 0, 0, 0);
          acc[c + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b1.y, acc[c + 1], 0, 0, 0);
          acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b0.z, acc[c], 0, 0, 0);
          acc[c + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b1.z, acc[c + 1], 0, 0, 0);
          acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b0.w, acc[c], 0, 0, 0);
          acc[c + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b1.w, acc[c + 1], 0, 0, 0);
        }
      }
    }

    // ---- tile epilogue ----
    // lane: co = nt*32 + li, tiles m = (r&3) + 8*(r>>2) + 4*lh of M-tile mt, components (2*chalf + {0,1}, 0..3).
    // S = A^T M: s0 = M0 + M1 + M2, s1 = M1 - M2 - M3 (per column); half 0 holds rows 0, 1 and half 1 rows 2, 3, so each
    // half contributes (s0, s1) = (M0 + M1, M1) resp. (M2, -M2 - M3); the partial outputs Y = S A meet in the store loop.
    if (!(a.ablate & 4)) {
      const bool full = (ty0 + TH <= a.H) && (tx0 + TW <= a.W);
      __syncthreads();  // MFMA reads of sA / sB finished: the two staging tiles (one per half) may overwrite them
      if (!(a.ablate & 32)) {
        float* const stg = smem + chalf * (TH * TW * NB);
        const float bz = chalf == 0 ? bias_v : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = (r & 3) + 8 * (r >> 2) + 4 * lh;
          const int sl = mt * 32 + m;
          const int ty = sl / TTX, tx = sl % TTX;
          float s0[4], s1[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (chalf == 0) {
              s0[j] = acc[j][r] + acc[4 + j][r];
              s1[j] = acc[4 + j][r];
            } else {
              s0[j] = acc[j][r];
              s1[j] = -acc[j][r] - acc[4 + j][r];
            }
          }
          float* o = stg + ((2 * ty) * TW + 2 * tx) * NB + nt * 32 + li;
          o[0] = s0[0] + s0[1] + s0[2] + bz;
          o[NB] = s0[1] - s0[2] - s0[3] + bz;
          o[TW * NB] = s1[0] + s1[1] + s1[2] + bz;
          o[TW * NB + NB] = s1[1] - s1[2] - s1[3] + bz;
        }
      }
      __syncthreads();
      const int q16 = tid & 15;
      const int co4 = cob * NB + q16 * 4;
      const int nvalid = min(4, a.Cout - co4);
#pragma unroll 4
      for (int k = 0; k < (TH * TW * 16) / WINO_THREADS; ++k) {
        const int lp = (tid >> 4) + (WINO_THREADS / 16) * k;
        const int orow = lp / TW, ocol = lp - orow * TW;
        const int oy = ty0 + orow, ox = tx0 + ocol;
        if (nvalid > 0 && (full || (oy < a.H && ox < a.W))) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(smem + lp * NB + q16 * 4) +
                          *reinterpret_cast<const f32x4*>(smem + TH * TW * NB + lp * NB + q16 * 4);
          ssum += v;
          ssq += v * v;
          if (a.ablate & 16) continue;
          float* p = p_out + ((size_t)(n * a.H + oy) * a.W + ox) * a.out_cs + a.out_co + co4;
          if (nvalid == 4) {
            *reinterpret_cast<f32x4*>(p) = v;
          } else {
            p[0] = v[0];
            if (nvalid > 1) p[1] = v[1];
            if (nvalid > 2) p[2] = v[2];
          }
        }
      }
    }
    if (!has_next) break;
    tile = next_tile;
  }

  if (p_stats != nullptr) {
    __syncthreads();
    float* red = smem;  // [32 pixel groups][16 quads][8]
    *reinterpret_cast<f32x4*>(red + tid * 8) = ssum;
    *reinterpret_cast<f32x4*>(red + tid * 8 + 4) = ssq;
    __syncthreads();
    if (tid < 128) {
      const int ch = tid >> 1, which = tid & 1;  // channel ch of the block: quad ch >> 2, element ch & 3
      float t = 0.f;
      for (int gq = 0; gq < WINO_THREADS / 16; ++gq) t += red[(gq * 16 + (ch >> 2)) * 8 + which * 4 + (ch & 3)];
      const int co = cob * NB + ch;
      if (co < a.Cout)
        acc_add_stats(p_stats + (size_t)(blockIdx.x % NREP) * 2 * a.Cout + which * a.Cout + co, (double)t);
    }
  }
}

#undef WINO_DECODE_TILE
#undef WINO_ISSUE_LOADS

// ------------------------------------------------------------------------------------------------
// Winograd weight gradient F(3x3, 2x2) (the transpose of F(2x2, 3x3)):
//   dW(3x3) = G^T [ sum over 2x2-output tiles (B^T d B) (.) (A dy A^T) ] G      d: 4x4 input patch, dy: 2x2 dY tile
// 16 multiplies per tile and (ci, co) instead of 36.  The 16 component products are GEMMs over K = tiles:
// block = 8 waves, owns a 64 ci x 64 co slab of all 16 components (wave = (co half, component row i): 4 components x
// 2 M-tiles = 128 accumulator registers) and walks a contiguous range of 128-pixel block tiles (32 Winograd tiles) of
// BOTH views.  LDS holds only the RAW input halo and dY tile ([pixel][64 channels], BatchNorm + ReLU of the producer
// applied on the way in); both transforms run in registers between the LDS reads and the MFMAs:
//   lane (li, lh): k = lh selects the tile of the pair, one ds_read_b64 gives channels 2 li, 2 li + 1 = the rows li of
//   M-tiles 0 / 1;  T[i][c] = d[ra][c] +- d[rb][c],  V[i][j] from T[i][0..3];  D[i][j] from the 2x2 dY values of co.
// The partial slabs [block][component][ci][co] are summed over the splits, transformed with G^T . G and accumulated
// into the OIHW gradient by wgrad_wino_reduce_kernel.
// ------------------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <bool WIDE>
struct WgradWinoGeom {
  static constexpr int TH = WIDE ? 4 : 16, TW = WIDE ? 32 : 8;  // 128 output pixels = 32 tiles per step
  static constexpr int TTX = TW / 2;
  static constexpr int HT = TH + 2, WT = TW + 2;
  static constexpr int X_FLOATS = HT * WT * 64, D_FLOATS = TH * TW * 64;
  static constexpr int LDS_BYTES = (X_FLOATS + D_FLOATS + 256) * 4;
};

// The 16 K-steps (32 Winograd tiles, lane half lh = tile of the pair) of one block tile, fully unrolled: every LDS offset
// is base + compile-time constant (the halo rows ra / rb of the wave's component row are folded into the lane bases) and
// all two-wide arithmetic is written on f32x2 so that it compiles to v_pk_fma_f32 / v_pk_add_f32: 13 VALU instructions per
// 8 MFMAs instead of ~34.  On this chip every VALU instruction beside a v_mfma_f32_32x32x2_f32 costs matrix-pipe time one
// for one (the fp32 MFMA executes on the vector ALUs: DESIGN.md section 8, tools/ubench/mfma_overlap.hip).
//   T[c] = d[ra][c] + sg d[rb][c];   (r0, r1) = c0 dy[0][0..1] + c1 dy[1][0..1]
struct WgsNoHook {
  template <typename S> __device__ __forceinline__ void operator()(S) const {}
};
// `hook(std::integral_constant<int, S>)` runs behind the LDS reads of step S, in the shadow of its first MFMA (wgrad_wino_fused_kernel
// issues the next tile's global loads there, two per step)
template <typename G, typename HOOK = WgsNoHook>
__device__ __forceinline__ void wgrad_wino_steps(f32x16 (&acc)[4][2], const float* __restrict__ xa0,
                                                 const float* __restrict__ xb0, const float* __restrict__ db0,
                                                 const f32x2 sg, const f32x2 c0, const f32x2 c1, HOOK&& hook = HOOK()) {
  f32x2 U[4], Wv[4], T[4], top, bot, r;
#define WGS_LOAD(S)                                                                                          \
  {                                                                                                          \
    constexpr int t0_ = 2 * (S);  /* even tile of the pair; the odd one (lh = 1) is folded into the lane bases */ \
    constexpr int ty_ = t0_ / G::TTX, tx_ = t0_ % G::TTX;                                                    \
    constexpr int xo_ = ((2 * ty_) * G::WT + 2 * tx_) * 64, do_ = ((2 * ty_) * G::TW + 2 * tx_) * 64;        \
    _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                          \
      U[c] = *reinterpret_cast<const f32x2*>(xa0 + xo_ + c * 64);                                            \
      Wv[c] = *reinterpret_cast<const f32x2*>(xb0 + xo_ + c * 64);                                           \
    }                                                                                                        \
    top[0] = db0[do_]; top[1] = db0[do_ + 64];                                                               \
    bot[0] = db0[do_ + G::TW * 64]; bot[1] = db0[do_ + G::TW * 64 + 64];                                     \
  }
  // first half of a step: consumes the raw operands (their registers are free for the next step's LDS reads)
#define WGS_HEAD()                                                                                           \
  {                                                                                                          \
    _Pragma("unroll") for (int c = 0; c < 4; ++c) T[c] = pk_fma(sg, Wv[c], U[c]);                   \
    r = pk_fma(c1, bot, pk_mul(c0, top));                                                                   \
  }
  // second half: the operands of the 8 MFMAs, the first MFMA, then - in ITS shadow - the LDS reads of step s + 1 and the hook's
  // global loads (a non-VALU instruction directly behind an MFMA issues for free while the matrix pipe is busy; in front of the
  // group, where they stood until round 6, each of the 8 cost ~5 cycles: PERF_LOG round 6 section 0), then the other 7 MFMAs.
  // The reads land under those 7 (448 cycles).
#define WGS_STEP(S)                                                                                          \
  WGS_HEAD()                                                                                                 \
  {                                                                                                          \
    const f32x2 V0 = pk_sub(T[0], T[2]), V1 = pk_add(T[1], T[2]), V2 = pk_sub(T[2], T[1]), V3 = pk_sub(T[1], T[3]); \
    const float D0 = r[0], D1 = r[0] + r[1], D2 = r[0] - r[1], D3 = -r[1];                                   \
    /* hipcc does not track the VALU-write -> MFMA-read wait states (2) for registers written by inline asm: */ \
    /* all operands are complete before the fence, the first MFMA reads V0, written >= 3 instructions earlier */ \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(V0[0], D0, acc[0][0], 0, 0, 0);                         \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    if ((S) + 1 < 16) WGS_LOAD(((S) + 1 < 16 ? (S) + 1 : 0))                                                 \
    hook(std::integral_constant<int, (S)>{});                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(V0[1], D0, acc[0][1], 0, 0, 0);                         \
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(V1[0], D1, acc[1][0], 0, 0, 0);                         \
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(V1[1], D1, acc[1][1], 0, 0, 0);                         \
    acc[2][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(V2[0], D2, acc[2][0], 0, 0, 0);                         \
    acc[2][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(V2[1], D2, acc[2][1], 0, 0, 0);                         \
    acc[3][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(V3[0], D3, acc[3][0], 0, 0, 0);                         \
    acc[3][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(V3[1], D3, acc[3][1], 0, 0, 0);                         \
  }                                                                                                          \
  __builtin_amdgcn_sched_barrier(0);
  WGS_LOAD(0)
  WGS_STEP(0) WGS_STEP(1) WGS_STEP(2) WGS_STEP(3) WGS_STEP(4) WGS_STEP(5) WGS_STEP(6) WGS_STEP(7)
  WGS_STEP(8) WGS_STEP(9) WGS_STEP(10) WGS_STEP(11) WGS_STEP(12) WGS_STEP(13) WGS_STEP(14) WGS_STEP(15)
#undef WGS_LOAD
#undef WGS_HEAD
#undef WGS_STEP
}

template <int IN_MODE, bool WIDE>
__global__ __launch_bounds__(512) void wgrad_wino_kernel(const WgradArgs a) {
  using G = WgradWinoGeom<WIDE>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sX = smem;
  float* sD = smem + G::X_FLOATS;
  float* sS = smem + G::X_FLOATS + G::D_FLOATS;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar wave roles
  const int li = lane & 31, lh = lane >> 5;
  const int coh = wave & 1, irow = wave >> 1;

  int bid = blockIdx.x;
  const int split = bid % a.nsplit;
  bid /= a.nsplit;
  const int cob = bid % a.ncob;
  const int cib = bid / a.ncob;
  const int tot_tiles = a.ntiles * a.nprob;
  const int per = (tot_tiles + a.nsplit - 1) / a.nsplit;
  const int t_begin = split * per, t_end = min(tot_tiles, t_begin + per);

  f32x16 acc[4][2];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][e][r] = 0.f;

  const int q16 = tid & 15;
  const int ci0 = cib * 64 + q16 * 4;
  const bool civalid = ci0 < a.Cin;
  if (IN_MODE != 0 && tid < 32) {
    const int pr = tid >> 4;
    f32x4 sc0 = {1.f, 1.f, 1.f, 1.f}, sh0 = {0.f, 0.f, 0.f, 0.f};
    if (civalid && pr < a.nprob) {
      sc0 = *reinterpret_cast<const f32x4*>((pr ? a.in_scale2 : a.in_scale) + ci0);
      sh0 = *reinterpret_cast<const f32x4*>((pr ? a.in_shift2 : a.in_shift) + ci0);
    }
    *reinterpret_cast<f32x4*>(sS + pr * 128 + q16 * 4) = sc0;
    *reinterpret_cast<f32x4*>(sS + pr * 128 + 64 + q16 * 4) = sh0;
  }
  const int co0 = cob * 64 + q16 * 4;
  const bool covalid = co0 < a.Cout;

  constexpr int NX = (G::HT * G::WT + 31) / 32;  // halo pixels per thread (pp = (tid >> 4) + 32 i)
  constexpr int ND = G::TH * G::TW / 32;
  f32x4 xreg[NX], dreg[ND];
  unsigned xmask = 0;

  const int xpix = a.in_cs * 4, xrow = a.W * xpix, dpix = a.dout_cs * 4, drow = a.W * dpix;
  const int xq = civalid ? (a.in_co + ci0) * 4 : -1, dq = covalid ? (a.dout_co + co0) * 4 : -1;
  constexpr unsigned OOB = 0x80000000u;
  // byte offsets of this thread's staging slots (slot i = pixel (tid >> 4) + 32 i of the halo / dY raster) relative to
  // the first halo / dY pixel of a tile; OOB for slots past the raster and for channel quads outside the tensor.
  // Tiles whose halo lies inside the image (3/4 of them at 240x320) add the tile origin as the SCALAR offset of the
  // buffer load: no per-slot address arithmetic in the MFMA phase (it cost ~90 VALU instructions per tile and thread).
  unsigned xoffv[NX], doffv[ND];
  unsigned xmask_in = 0;
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    const int pp = (tid >> 4) + 32 * i, r = pp / G::WT, c = pp - r * G::WT;
    const bool ok = xq >= 0 && pp < G::HT * G::WT;
    xoffv[i] = ok ? (unsigned)(r * xrow + c * xpix + xq) : OOB;
    xmask_in |= (ok ? 1u : 0u) << i;
  }
#pragma unroll
  for (int i = 0; i < ND; ++i) {
    const int pp = (tid >> 4) + 32 * i, r = pp / G::TW, c = pp - r * G::TW;
    doffv[i] = dq >= 0 ? (unsigned)(r * drow + c * dpix + dq) : OOB;
  }
  bool cur_inside = false, nxt_inside = false;  // wave-uniform: the staged / the prefetched tile is an interior tile
  // one buffer descriptor per image (32-bit offsets inside it); an offset beyond num_records returns 0
#define WGW_ISSUE(TILE)                                                                                       \
  if (!(a.ablate & 1)) {                                                                                      \
    const int pr_ = (TILE) >= a.ntiles ? 1 : 0;                                                               \
    const int tl_ = (TILE) - pr_ * a.ntiles;                                                                  \
    const int tx_ = tl_ % a.tiles_x, t2_ = tl_ / a.tiles_x;                                                   \
    const int ty0_ = (t2_ % a.tiles_y) * G::TH, tx0_ = tx_ * G::TW, n_ = t2_ / a.tiles_y;                     \
    const __amdgpu_buffer_rsrc_t rx_ = __builtin_amdgcn_make_buffer_rsrc(                                     \
        const_cast<float*>(pr_ ? a.in2 : a.in) + (size_t)n_ * a.H * a.W * a.in_cs, 0, a.H * xrow, 0x00020000); \
    const __amdgpu_buffer_rsrc_t rd_ = __builtin_amdgcn_make_buffer_rsrc(                                     \
        const_cast<float*>(pr_ ? a.dout2 : a.dout) + (size_t)n_ * a.H * a.W * a.dout_cs, 0, a.H * drow, 0x00020000); \
    nxt_inside = ty0_ >= 1 && ty0_ + G::TH + 1 <= a.H && tx0_ >= 1 && tx0_ + G::TW + 1 <= a.W;               \
    if (nxt_inside) {                                                                                         \
      const int xb_ = (ty0_ - 1) * xrow + (tx0_ - 1) * xpix, db_ = ty0_ * drow + tx0_ * dpix;                 \
      _Pragma("unroll") for (int i = 0; i < NX; ++i)                                                          \
        xreg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx_, xoffv[i], xb_, 0));    \
      _Pragma("unroll") for (int i = 0; i < ND; ++i)                                                          \
        dreg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rd_, doffv[i], db_, 0));    \
      xmask = xmask_in;                                                                                       \
    } else {                                                                                                  \
      xmask = 0;                                                                                              \
      _Pragma("unroll") for (int i = 0; i < NX; ++i) {                                                        \
        const int pp_ = (tid >> 4) + 32 * i, r_ = pp_ / G::WT, c_ = pp_ - r_ * G::WT;                         \
        const int gy = ty0_ - 1 + r_, gx = tx0_ - 1 + c_;                                                     \
        const bool ok = xq >= 0 && pp_ < G::HT * G::WT && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W; \
        xreg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(                            \
            rx_, ok ? (unsigned)(gy * xrow + gx * xpix + xq) : OOB, 0, 0));                                   \
        xmask |= (ok ? 1u : 0u) << i;                                                                         \
      }                                                                                                       \
      _Pragma("unroll") for (int i = 0; i < ND; ++i) {                                                        \
        const int pp_ = (tid >> 4) + 32 * i, r_ = pp_ / G::TW, c_ = pp_ - r_ * G::TW;                         \
        const int gy = ty0_ + r_, gx = tx0_ + c_;                                                             \
        const bool ok = dq >= 0 && gy < a.H && gx < a.W;                                                      \
        dreg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(                            \
            rd_, ok ? (unsigned)(gy * drow + gx * dpix + dq) : OOB, 0, 0));                                   \
      }                                                                                                       \
    }                                                                                                         \
  }

  // per-wave constants of component row i:  T[i][c] = d[ra][c] + sg d[rb][c];  R[q] = c0 dy[0][q] + c1 dy[1][q]
  const int ra = irow == 0 ? 0 : irow == 2 ? 2 : 1;
  const int rb = irow == 2 ? 1 : irow == 3 ? 3 : 2;
  const float sg = irow == 1 ? 1.f : -1.f;
  const float c0 = irow == 3 ? 0.f : 1.f;
  const float c1 = irow == 0 ? 0.f : irow == 1 ? 1.f : -1.f;

  if (t_begin < t_end) WGW_ISSUE(t_begin)
  for (int tile = t_begin; tile < t_end; ++tile) {
    __syncthreads();  // all waves finished reading the previous tile's LDS image
    cur_inside = nxt_inside;
    if (!(a.ablate & 2)) {
      const int cur_prob = tile >= a.ntiles ? 1 : 0;
      f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
      if (IN_MODE != 0) {
        sc = *reinterpret_cast<const f32x4*>(sS + cur_prob * 128 + q16 * 4);
        sh = *reinterpret_cast<const f32x4*>(sS + cur_prob * 128 + 64 + q16 * 4);
      }
      if (IN_MODE != 0 && cur_inside) {
        // interior tile: no zero padding anywhere (channel quads outside the tensor carry scale 1 / shift 0 and load 0)
#pragma unroll
        for (int i = 0; i < NX; ++i) {
          const int pp = (tid >> 4) + 32 * i;
          if (pp < G::HT * G::WT) {
            f32x4 v = pk4_fma(xreg[i], sc, sh);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            *reinterpret_cast<f32x4*>(sX + pp * 64 + q16 * 4) = v;
          }
        }
      } else {
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        const int pp = (tid >> 4) + 32 * i;
        if (pp < G::HT * G::WT) {
          f32x4 v = xreg[i];  // 0 from the buffer load where the pixel / channel quad is outside
          if (IN_MODE != 0) v = bn_relu_quad(v, sc, sh, !((xmask >> i) & 1u));
          *reinterpret_cast<f32x4*>(sX + pp * 64 + q16 * 4) = v;
        }
      }
      }
#pragma unroll
      for (int i = 0; i < ND; ++i) {
        const int pp = (tid >> 4) + 32 * i;
        *reinterpret_cast<f32x4*>(sD + pp * 64 + q16 * 4) = dreg[i];  // 0 where outside
      }
    }
    __syncthreads();
    {
      const int nxt = min(tile + 1, t_end - 1);  // unconditional prefetch (redundant on the last tile)
      WGW_ISSUE(nxt)
      __builtin_amdgcn_sched_barrier(0);
    }
    if (!(a.ablate & 8)) {
      // lane bases: channel pair 2 li of the raw rows ra / rb, tile column offset of the lane half; dY column of this lane
      const float* xa0 = sX + (ra * G::WT + 2 * lh) * 64 + 2 * li;
      const float* xb0 = sX + (rb * G::WT + 2 * lh) * 64 + 2 * li;
      const float* db0 = sD + (2 * lh) * 64 + coh * 32 + li;
      wgrad_wino_steps<G>(acc, xa0, xb0, db0, f32x2{sg, sg}, f32x2{c0, c0}, f32x2{c1, c1});
    }
  }
#undef WGW_ISSUE
  // partial slab: [blk][component][ci 64][co 64]; M-tile e, row m <-> input channel 2 m + e
  float* dst = a.partial + (size_t)blockIdx.x * WC * 4096;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * lh;
        dst[(irow * 4 + j) * 4096 + (2 * m + e) * 64 + coh * 32 + li] = acc[j][e][r];
      }
}

// Sums the 16-component partial slabs over the splits, applies dW = G^T M G and ACCUMULATES into the OIHW gradient.
// block = 256 threads = 64 consecutive co x 4 split groups, one input channel per block row.
__device__ __forceinline__ void wgrad_wino_reduce_block(const float* __restrict__ partial, float* __restrict__ dw, int Cin,
                                                        int Cout, int ncob, int nsplit, int bid, float (*red)[WC][64]) {
  const int o = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int cob = bid % ncob, ci = bid / ncob;
  const int co = cob * 64 + o, cib = ci >> 6;
  float m[WC];
#pragma unroll
  for (int c = 0; c < WC; ++c) m[c] = 0.f;
  const float* src = partial + (size_t)((cib * ncob + cob) * nsplit) * WC * 4096 + (ci & 63) * 64 + o;
  for (int k = grp; k < nsplit; k += 4)
#pragma unroll
    for (int c = 0; c < WC; ++c) m[c] += src[((size_t)k * WC + c) * 4096];
#pragma unroll
  for (int c = 0; c < WC; ++c) red[grp][c][o] = m[c];
  __syncthreads();
  if (grp != 0 || co >= Cout) return;
#pragma unroll
  for (int c = 0; c < WC; ++c) m[c] = (red[0][c][o] + red[1][c][o]) + (red[2][c][o] + red[3][c][o]);
  float p[3][4];  // P = G^T M
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    p[0][j] = m[0 * 4 + j] + 0.5f * (m[1 * 4 + j] + m[2 * 4 + j]);
    p[1][j] = 0.5f * (m[1 * 4 + j] - m[2 * 4 + j]);
    p[2][j] = 0.5f * (m[1 * 4 + j] + m[2 * 4 + j]) + m[3 * 4 + j];
  }
  float* out = dw + ((size_t)co * Cin + ci) * 9;
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    out[u * 3 + 0] += p[u][0] + 0.5f * (p[u][1] + p[u][2]);
    out[u * 3 + 1] += 0.5f * (p[u][1] - p[u][2]);
    out[u * 3 + 2] += 0.5f * (p[u][1] + p[u][2]) + p[u][3];
  }
}

__global__ __launch_bounds__(256) void wgrad_wino_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw,
                                                                int Cin, int Cout, int ncob, int nsplit) {
  __shared__ float red[4][WC][64];
  wgrad_wino_reduce_block(partial, dw, Cin, Cout, ncob, nsplit, blockIdx.x, red);
}

// The reductions of ALL Winograd weight-gradient launches of a backward pass in one launch (each wgrad launch keeps its
// own slice of the partial-slab buffer until then): 9 launches of ~20 us less per step.
struct WredJob {
  const float* partial;
  float* dw;
  int cin, cout, ncob, nsplit;
  int nblocks;  // blocks of the job: ncob * cin (f4 == 2: three times as many)
  int block0;
  int f4;  // 0: 16-component slabs of wgrad_wino_kernel (ncob = 64-channel blocks); 1: 9-tap slabs [9][64][32] of
           // wgrad_wino4_kernel, already through G^T . G (ncob = 32-channel blocks); 2: 12-component slabs [4][3][64][64] of
           // wgrad_wino_fused_kernel (right-hand product applied)
};

// wgrad_wino4_kernel's partial slabs [pair * nsplit + k][9 taps][64 ci][32 co] summed over the splits and ACCUMULATED into
// the OIHW gradient.  block = 256 threads = 32 consecutive co x 8 split groups, one input channel per block.
__device__ __forceinline__ void wgrad_wino4_reduce_block(const float* __restrict__ partial, float* __restrict__ dw, int Cin,
                                                         int Cout, int ncob, int nsplit, int bid, float* red) {
  const int o = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int cob = bid % ncob, ci = bid / ncob;
  const int co = cob * 32 + o, cib = ci >> 6;
  float m[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) m[t] = 0.f;
  const float* src = partial + (size_t)((cib * ncob + cob) * nsplit) * (9 * 2048) + (ci & 63) * 32 + o;
  for (int k = grp; k < nsplit; k += 8)
#pragma unroll
    for (int t = 0; t < 9; ++t) m[t] += src[((size_t)k * 9 + t) * 2048];
#pragma unroll
  for (int t = 0; t < 9; ++t) red[(grp * 9 + t) * 32 + o] = m[t];
  __syncthreads();
  if (grp != 0 || co >= Cout) return;
  float* out = dw + ((size_t)co * Cin + ci) * 9;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) s += red[(g * 9 + t) * 32 + o];
    out[t] += s;
  }
}
__global__ __launch_bounds__(256) void wgrad_wino4_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw,
                                                                 int Cin, int Cout, int ncob, int nsplit) {
  __shared__ float red[8 * 9 * 32];
  wgrad_wino4_reduce_block(partial, dw, Cin, Cout, ncob, nsplit, blockIdx.x, red);
}

// wgrad_wino_fused_kernel's slabs [(cib * ncob + cob) * nsplit + k][row i 4][x 3][64 ci][64 co] (the right-hand product of
// dW = G^T M G applied by the workgroup) summed over the splits, the left-hand product, ACCUMULATED into the OIHW gradient.
// block = one input channel, 64 consecutive co, ONE column x of the taps (the left-hand product mixes the rows i of a column only:
// three blocks per (ci, cob) pair without an exchange); 256 threads = 16 co quads (16-byte loads) x 16 split groups - a 64 x 64
// layer has 256 splits and only 64 pairs, so the splits, not the pairs, have to carry the parallelism.
__device__ __forceinline__ void wgrad_fused12_reduce_block(const float* __restrict__ partial, float* __restrict__ dw, int Cin,
                                                           int Cout, int ncob, int nsplit, int rel, float* red) {
  const int q = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int x = rel % 3, bid = rel / 3;
  const int cob = bid % ncob, ci = bid / ncob, cib = ci >> 6;
  float4 m[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) m[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  const float* src = partial + (size_t)((cib * ncob + cob) * nsplit) * (12 * 4096) + x * 4096 + (ci & 63) * 64 + q * 4;
  for (int k = grp; k < nsplit; k += 16)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 v = *reinterpret_cast<const float4*>(src + ((size_t)k * 12 + i * 3) * 4096);
      m[i].x += v.x; m[i].y += v.y; m[i].z += v.z; m[i].w += v.w;
    }
#pragma unroll
  for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(red + ((grp * 4 + i) * 16 + q) * 4) = m[i];   // [grp 16][i 4][co 64]
  __syncthreads();
  const int o = threadIdx.x & 63, u = threadIdx.x >> 6, co = cob * 64 + o;
  if (u == 3 || co >= Cout) return;
  float r[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int g = 0; g < 16; g += 2) { a0 += red[((g * 4 + i) * 64) + o]; a1 += red[(((g + 1) * 4 + i) * 64) + o]; }
    r[i] = a0 + a1;
  }
  const float v = u == 0 ? r[0] + 0.5f * (r[1] + r[2]) : u == 1 ? 0.5f * (r[1] - r[2]) : 0.5f * (r[1] + r[2]) + r[3];
  dw[((size_t)co * Cin + ci) * 9 + u * 3 + x] += v;
}
__global__ __launch_bounds__(256) void wgrad_fused12_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw,
                                                                   int Cin, int Cout, int ncob, int nsplit) {
  __shared__ __attribute__((aligned(16))) float red[4 * WC * 64];
  wgrad_fused12_reduce_block(partial, dw, Cin, Cout, ncob, nsplit, blockIdx.x, red);
}

constexpr int WRED_MAX_JOBS = 16;
struct WredJobs {
  int n;
  WredJob j[WRED_MAX_JOBS];
};
// (wgrad_wino_reduce_multi_kernel, the launch that runs these jobs, lives in backward_tail.hip.h beside the other tail reductions)

// OIHW 3x3 weights -> U = G g G^T in the LDS image of conv_wino_kernel: [cob][chunk][component][g][h][64][4].
// transpose_flip: the data-gradient convolution (input channels = Cout_w, output = Cin_w, taps mirrored).
__global__ void pack_weights_wino_kernel(const float* __restrict__ w, float* __restrict__ dst, int Cout_w, int Cin_w,
                                         int transpose_flip, int nchunks_total, int chunk_off, int cob_off, int ncob,
                                         int nchunks) {
  const int per_chunk = WB_FLOATS;
  const int total = ncob * nchunks * per_chunk;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  int t = idx;
  const int e = t & 3;
  t >>= 2;
  const int nn = t & 63;
  t >>= 6;
  const int h = t & 1;
  t >>= 1;
  const int g = t % (CK / 8);
  t /= (CK / 8);
  const int comp = t % WC;
  t /= WC;
  const int chunk = t % nchunks;
  const int cob = t / nchunks;
  const int co = cob * NB + nn;
  const int ci = chunk * CK + g * 8 + h * 4 + e;
  float k[3][3];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      float v = 0.f;
      if (!transpose_flip) {
        if (co < Cout_w && ci < Cin_w) v = w[(((size_t)co * Cin_w + ci) * 3 + ky) * 3 + kx];
      } else {
        if (co < Cin_w && ci < Cout_w) v = w[(((size_t)ci * Cin_w + co) * 3 + (2 - ky)) * 3 + (2 - kx)];
      }
      k[ky][kx] = v;
    }
  const int i = comp >> 2, j = comp & 3;
  float r[3];  // row i of G g
#pragma unroll
  for (int x = 0; x < 3; ++x)
    r[x] = i == 0 ? k[0][x] : i == 1 ? 0.5f * (k[0][x] + k[1][x] + k[2][x]) : i == 2 ? 0.5f * (k[0][x] - k[1][x] + k[2][x]) : k[2][x];
  const float u = j == 0 ? r[0] : j == 1 ? 0.5f * (r[0] + r[1] + r[2]) : j == 2 ? 0.5f * (r[0] - r[1] + r[2]) : r[2];
  dst[((size_t)(cob + cob_off) * nchunks_total + chunk + chunk_off) * per_chunk +
      (((comp * (CK / 8) + g) * 2 + h) * NB + nn) * 4 + e] = u;
}

}  // namespace sspk
