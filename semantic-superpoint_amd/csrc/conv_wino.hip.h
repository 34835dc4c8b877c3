// Winograd F(2x2, 3x3) convolution on the fp32 matrix cores (forward and data-gradient 3x3 convs with Cin % 16 == 0).
//
//   Y(2x2) = A^T [ sum_ci (G g G^T) (.) (B^T d B) ] A        d: 4x4 input patch, g: 3x3 filter   (Lavin & Gray 2016)
//
// The 16 element-wise products are 16 independent GEMMs [tiles x Cin] x [Cin x Cout]: 16 multiplies per 2x2 outputs
// instead of 36, i.e. 2.25x fewer MFMA FLOPs than the direct implicit GEMM of conv_mfma_kernel for the same
// (algorithmic) convolution, still in exact fp32 arithmetic (the transforms only add and halve).
//
// Block = 256 threads = 4 waves, ONE block per CU (144 KB of LDS, 256 accumulator registers per lane):
//   output tile 8x32 (WIDE) or 32x8 pixels = 64 Winograd tiles x 64 output channels;
//   wave w owns M-tile (w >> 1) (32 tiles) x N-tile (w & 1) (32 channels) for ALL 16 components: 16 x f32x16 acc.
// Per 16-channel K-chunk a thread (tile = tid >> 2, channel quad = tid & 3) loads its 4x4 patch (16 x b128, BatchNorm
// + ReLU of the producer applied on the way, zero outside the image), transforms it in registers and writes the 16
// components to LDS as [component][tile][CS]; the pre-transformed weights G g G^T arrive packed as
// [cob][chunk][component][g][h][64][4] (pack_weights_wino_kernel), the same fragment layout as conv_mfma_kernel with
// "tap" replaced by "component".  The output transform runs on the accumulators (each lane holds all 16 components of
// its 16 tiles), then the tile goes through LDS for 16-byte stores exactly like conv_mfma_kernel.
// Same persistent XCD-aware grid, dual-problem launches, register-staged prefetch and BatchNorm partial sums.
#pragma once
#include "conv_mfma.hip.h"

namespace sspk {

constexpr int WC = 16;                          // Winograd components
constexpr int WTILES = 64;                      // 2x2-output tiles per block
constexpr int WA_FLOATS = WC * WTILES * CS;     // transformed input chunk  (81920 B)
constexpr int WB_FLOATS = WC * CK * NB;         // transformed weight chunk (65536 B)
constexpr int WINO_LDS_BYTES = (WA_FLOATS + WB_FLOATS) * 4;

template <int IN_MODE, bool WIDE>
__global__ __launch_bounds__(256) void conv_wino_kernel(const ConvArgs a) {
  constexpr int TTX = WIDE ? 16 : 4;            // tiles per block row
  constexpr int TH = WIDE ? 8 : 32, TW = WIDE ? 32 : 8;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sA = smem;
  float* sB = smem + WA_FLOATS;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int mt = wave >> 1, nt = wave & 1;

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

  // ---- staging role: one (tile, channel quad) per thread ----
  const int q4 = tid & 3, st_slot = tid >> 2;
  const int st_ty = st_slot / TTX, st_tx = st_slot % TTX;
  const int pixb = a.in_cs * 4, rowb = a.W * pixb;
  f32x4 hreg[16], wreg[16];
  f32x4 psc = {1.f, 1.f, 1.f, 1.f}, psh = {0.f, 0.f, 0.f, 0.f};
  constexpr unsigned OOB = 0x80000000u;
  int ld_n, ld_ty0, ld_tx0, vbase;
  unsigned pmask;  // bit 4*i+j: patch pixel (i, j) lies inside the image
  const size_t img_floats = (size_t)a.H * a.W * a.in_cs;
  __amdgpu_buffer_rsrc_t rsrc_in;
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wpk), 0, a.wpk_bytes, 0x00020000);
#define WINO_DECODE_TILE(T)                                                                              \
  {                                                                                                      \
    const int tx_ = (T) % a.tiles_x, t2_ = (T) / a.tiles_x;                                              \
    ld_tx0 = tx_ * TW;                                                                                   \
    ld_ty0 = (t2_ % a.tiles_y) * TH;                                                                     \
    ld_n = t2_ / a.tiles_y;                                                                              \
    const int py0_ = ld_ty0 + 2 * st_ty - 1, px0_ = ld_tx0 + 2 * st_tx - 1;                              \
    unsigned rm_ = 0, cm_ = 0;                                                                           \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                      \
      rm_ |= ((unsigned)(py0_ + i) < (unsigned)a.H ? 1u : 0u) << i;                                      \
      cm_ |= ((unsigned)(px0_ + i) < (unsigned)a.W ? 1u : 0u) << i;                                      \
    }                                                                                                    \
    pmask = 0;                                                                                           \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) pmask |= (((rm_ >> i) & 1u) ? cm_ : 0u) << (4 * i);    \
    vbase = py0_ * rowb + px0_ * pixb + (a.in_co + q4 * 4) * 4;                                          \
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
    _Pragma("unroll") for (int k = 0; k < 16; ++k) {                                                     \
      const unsigned vo_ = ((pmask >> k) & 1u) ? (unsigned)(vbase + (k >> 2) * rowb + (k & 3) * pixb) : OOB; \
      hreg[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_in, vo_, soff_, 0)); \
    }                                                                                                    \
    const int wbase_ = (cob * a.nchunks + (CHUNK)) * WB_FLOATS * 4;                                      \
    _Pragma("unroll") for (int j = 0; j < 16; ++j)                                                       \
      wreg[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, tid * 16, wbase_ + j * 4096, 0)); \
  }

  WINO_DECODE_TILE(tile)
  WINO_ISSUE_LOADS(0)

  float ssum = 0.f, ssq = 0.f;
  const int co_l = cob * NB + nt * 32 + li;
  const bool covalid = co_l < a.Cout;
  const float bias_v = (a.bias != nullptr && covalid) ? a.bias[co_l] : 0.f;
  const int a_off = (mt * 32 + li) * CS + lh * 4;
  const int b_off = (lh * NB + nt * 32 + li) * 4;

  for (;;) {  // ---- one output tile per iteration ----
    const int n = ld_n, ty0 = ld_ty0, tx0 = ld_tx0;
    const unsigned cmask = pmask;  // validity of the patch that is in the registers now
    const int next_tile = tile + per_cob;
    const bool has_next = next_tile < t_end;

    f32x16 acc[WC];
#pragma unroll
    for (int c = 0; c < WC; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;

    for (int chunk = 0; chunk < a.nchunks; ++chunk) {
      __syncthreads();  // every wave has finished reading the LDS image of the previous step
      if (!(a.ablate & 2)) {
        // BatchNorm + ReLU of the producer, zero padding, then V = B^T d B per channel -- in place in hreg
        if (IN_MODE != 0) {
#pragma unroll
          for (int k = 0; k < 16; ++k) {
#pragma unroll
            for (int e = 0; e < 4; ++e) hreg[k][e] = fmaxf(fmaf(hreg[k][e], psc[e], psh[e]), 0.f);
            if (!((cmask >> k) & 1u)) hreg[k] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {  // rows: T = B^T d
          const f32x4 d0 = hreg[j], d1 = hreg[4 + j], d2 = hreg[8 + j], d3 = hreg[12 + j];
          hreg[j] = d0 - d2;
          hreg[4 + j] = d1 + d2;
          hreg[8 + j] = d2 - d1;
          hreg[12 + j] = d1 - d3;
        }
        float* dst = sA + st_slot * CS + q4 * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {  // columns: V = T B, written straight to LDS
          const f32x4 t0 = hreg[i * 4], t1 = hreg[i * 4 + 1], t2 = hreg[i * 4 + 2], t3 = hreg[i * 4 + 3];
          *reinterpret_cast<f32x4*>(dst + (i * 4 + 0) * WTILES * CS) = t0 - t2;
          *reinterpret_cast<f32x4*>(dst + (i * 4 + 1) * WTILES * CS) = t1 + t2;
          *reinterpret_cast<f32x4*>(dst + (i * 4 + 2) * WTILES * CS) = t2 - t1;
          *reinterpret_cast<f32x4*>(dst + (i * 4 + 3) * WTILES * CS) = t1 - t3;
        }
        f32x4* wdst = reinterpret_cast<f32x4*>(sB);
#pragma unroll
        for (int j = 0; j < 16; ++j) wdst[tid + 256 * j] = wreg[j];
      }
      __syncthreads();
      {
        const bool last = chunk + 1 == a.nchunks;
        if (last && has_next) WINO_DECODE_TILE(next_tile)
        const int nxt = last ? 0 : chunk + 1;
        WINO_ISSUE_LOADS(nxt)
        __builtin_amdgcn_sched_barrier(0);
      }
      // ---- MFMA: 16 components x 2 k-groups x 4 k-pairs ----
      if (!(a.ablate & 8))
#pragma unroll
      for (int c = 0; c < WC; c += 2) {
#pragma unroll
        for (int g = 0; g < CK / 8; ++g) {
          const float4 a0 = *reinterpret_cast<const float4*>(sA + a_off + c * WTILES * CS + g * 8);
          const float4 a1 = *reinterpret_cast<const float4*>(sA + a_off + (c + 1) * WTILES * CS + g * 8);
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

    // ---- tile epilogue: output transform on the accumulators, LDS transpose, 16-byte stores ----
    if (!(a.ablate & 4)) {
      const bool full = (ty0 + TH <= a.H) && (tx0 + TW <= a.W);
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int sl = mt * 32 + m;
        const int ty = sl / TTX, tx = sl % TTX;
        float s0[4], s1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          s0[j] = acc[0 * 4 + j][r] + acc[1 * 4 + j][r] + acc[2 * 4 + j][r];
          s1[j] = acc[1 * 4 + j][r] - acc[2 * 4 + j][r] - acc[3 * 4 + j][r];
        }
        const float y00 = s0[0] + s0[1] + s0[2] + bias_v, y01 = s0[1] - s0[2] - s0[3] + bias_v;
        const float y10 = s1[0] + s1[1] + s1[2] + bias_v, y11 = s1[1] - s1[2] - s1[3] + bias_v;
        const int orow = 2 * ty, ocol = 2 * tx;
        float* o = smem + (orow * TW + ocol) * NB + nt * 32 + li;
        o[0] = y00;
        o[NB] = y01;
        o[TW * NB] = y10;
        o[TW * NB + NB] = y11;
        if (p_stats != nullptr && covalid) {
          const bool r0 = full || ty0 + orow < a.H, r1 = full || ty0 + orow + 1 < a.H;
          const bool c0 = full || tx0 + ocol < a.W, c1 = full || tx0 + ocol + 1 < a.W;
          if (r0 && c0) { ssum += y00; ssq += y00 * y00; }
          if (r0 && c1) { ssum += y01; ssq += y01 * y01; }
          if (r1 && c0) { ssum += y10; ssq += y10 * y10; }
          if (r1 && c1) { ssum += y11; ssq += y11 * y11; }
        }
      }
      __syncthreads();
      const int q16 = tid & 15;
      const int co4 = cob * NB + q16 * 4;
      const int nvalid = min(4, a.Cout - co4);
#pragma unroll 4
      for (int k = 0; k < (TH * TW) / 16; ++k) {
        const int lp = (tid >> 4) + 16 * k;
        const int orow = lp / TW, ocol = lp - orow * TW;
        const int oy = ty0 + orow, ox = tx0 + ocol;
        if (nvalid > 0 && (full || (oy < a.H && ox < a.W))) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(smem + lp * NB + q16 * 4);
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
    float* red = smem;  // [4 waves][32][2]
    const float s = ssum + __shfl_xor(ssum, 32), q = ssq + __shfl_xor(ssq, 32);
    if (lh == 0) {
      red[(wave * 32 + li) * 2 + 0] = s;
      red[(wave * 32 + li) * 2 + 1] = q;
    }
    __syncthreads();
    if (tid < 128) {
      const int ch = tid >> 1, which = tid & 1;  // ch in 0..63: N-tile ch >> 5 is held by waves (ch >> 5) and 2 + (ch >> 5)
      const int w0 = ch >> 5;
      const float t = red[(w0 * 32 + (ch & 31)) * 2 + which] + red[((w0 + 2) * 32 + (ch & 31)) * 2 + which];
      const int co = cob * NB + ch;
      if (co < a.Cout)
        unsafeAtomicAdd(p_stats + (size_t)(blockIdx.x % NREP) * 2 * a.Cout + which * a.Cout + co, (double)t);
    }
  }
}

#undef WINO_DECODE_TILE
#undef WINO_ISSUE_LOADS

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
