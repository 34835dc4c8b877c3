// Winograd F(2x2, 3x3) convolution with bf16 MATRIX-CORE OPERANDS (ssp_set_conv_algo(3), an opt-in reduced-precision
// mode; the default and every reported headline number use the fp32 kernels): the software-pipelined kernel of
// conv_wino_pipe.hip.h with the transformed input tiles and the transformed weights rounded to bf16 (RNE) on their way
// into LDS and v_mfma_f32_32x32x8_bf16 (K = 8 = one 8-channel stage per instruction, fp32 accumulation) instead of four
// v_mfma_f32_32x32x2_f32.  Activations, transforms, BatchNorm, bias, statistics and the outputs stay fp32 ("bf16
// compute / fp32 master", BASELINE configs[3]).  LDS: 2 x (16 KB input + 16 KB weights) + raw halo + scale/shift +
// a dedicated 64 KB staging tile = 147 KB.  Weights: pack_weights_wino8_bf16_kernel, [cob][chunk8][part][component][co][8 k].
//
// NT = 2 (ssp_set_conv_algo(7), "bf16x2"): every operand element x is carried as TWO bf16 values, hi = bf16(x) and
// lo = bf16(x - hi), i.e. 16 significant bits, and a product a * b is evaluated as a_hi b_hi + a_hi b_lo + a_lo b_hi
// (three matrix-core instructions, fp32 accumulation; the dropped a_lo b_lo term is 2^-16 of the product).  The rounding
// error of a product falls from 2^-9 to ~2^-16, below the cancellation in the Winograd output transform that makes the
// one-term mode useless for gradients.  LDS: 2 x 2 x (16 KB + 16 KB) + raw halo + scale/shift =
// 147 KB; the epilogue stages in the consumed 64 KB buffer like the fp32 kernel.
#pragma once
#include "conv_wino_pipe.hip.h"

namespace sspk {

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
constexpr int QA_FLOATS = WC * WTILES * PK / 2;   // 16 KB of bf16
constexpr int QB_FLOATS = WC * PK * NB / 2;       // 16 KB of bf16
constexpr int BF16_LDS_BYTES = (2 * (QA_FLOATS + QB_FLOATS) + PR_FLOATS + PS_FLOATS + 8 * 32 * NB + NB) * 4;  // + sign of gamma (pooled output)
constexpr int BF16X2_LDS_BYTES = (2 * 2 * (QA_FLOATS + QB_FLOATS) + PR_FLOATS + PS_FLOATS + NB) * 4;

// hi / lo split of four fp32 values into bf16 (round to nearest even both times)
__device__ __forceinline__ void bf16_split(const f32x4 v, bf16x4& hi, bf16x4& lo) {
  hi = __builtin_convertvector(v, bf16x4);
  lo = __builtin_convertvector(v - __builtin_convertvector(hi, f32x4), bf16x4);
}

template <int IN_MODE, bool WIDE, int NT = 1>
__global__ __launch_bounds__(WINO_THREADS) void conv_wino_bf16_kernel(const ConvArgs a) {
  constexpr int BUF_FLOATS = NT * (QA_FLOATS + QB_FLOATS);   // [A parts][B parts]
  constexpr int B_BASE = NT * QA_FLOATS;
  constexpr int TTX = WIDE ? 16 : 4;
  constexpr int TH = WIDE ? 8 : 32, TW = WIDE ? 32 : 8;
  constexpr int HC = TW + 2;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // [sA0][sB0][sA1][sB1][sR][sS][staging]: bf16 images are half the size of the fp32 kernel's, so the epilogue has its
  // own 64 KB staging tile
  float* const sR = smem + 2 * BUF_FLOATS;
  float* const sS = sR + PR_FLOATS;  // scale[Cin] | shift[Cin] of the producer's BatchNorm (IN_MODE 1)
  float* const sStageDedicated = sS + PS_FLOATS;  // NT == 1 only
  float* const sG = NT == 1 ? sStageDedicated + 8 * 32 * NB : sS + PS_FLOATS;  // +-1 per output channel: sign of gamma

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
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
  const int tile0 = xl * per_t + jj;
  if (jj >= per_cob || tile0 >= t_end) return;
  const float* const p_in = prob ? a.in2 : a.in;
  float* const p_out = prob ? a.out2 : a.out;
  const float* const p_scale = prob ? a.in_scale2 : a.in_scale;
  const float* const p_shift = prob ? a.in_shift2 : a.in_shift;
  double* const p_stats = prob ? a.stats2 : a.stats;
  // raw 2x2-pooled copy of the output for a BatchNorm + ReLU + MaxPool consumer (ConvArgs::pool_out, see conv_wino_pipe_kernel)
  float* const p_pool = IN_MODE == 0 ? nullptr : (prob ? a.pool_out[1] : a.pool_out[0]);
  const int nst = a.Cin / PK;                                  // stages per tile
  const int my_tiles = (t_end - tile0 + per_cob - 1) / per_cob;
  const int nstages = my_tiles * nst;

  // ---- staging roles ----
  const int q2 = tid & 1;
  // raw halo items tid + 512 k (k < 2), item = pixel * 2 + quad
  int rrc[2], r_lds[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int p = (tid + WINO_THREADS * k) >> 1, r = p / HC, c = p - r * HC;
    rrc[k] = r | (c << 8);
    r_lds[k] = pipe_raw_off<WIDE>(p, q2);
  }
  const bool r1 = tid + WINO_THREADS < WHALO * 2;  // the second item exists
  // transform: (quad, tile, V row)
  const int t_tile = (tid >> 1) & 63, t_row = tid >> 7;
  const int t_ty = t_tile / TTX, t_tx = t_tile % TTX;
  const int t_ra = t_row == 0 ? 0 : t_row == 2 ? 2 : 1;   // T[i] = d[ra] + sg d[rb]
  const int t_rb = t_row == 2 ? 1 : t_row == 3 ? 3 : 2;
  const float t_sg = t_row == 1 ? 1.f : -1.f;
  const int t_dst = ((t_row * 4) * WTILES + t_tile) * 4 + q2 * 2;  // float units: 8 bf16 = 16 bytes per (component, tile)
  const int pixb = a.in_cs * 4, rowb = a.W * pixb;
  f32x4 hreg[2], wreg[2 * NT];
  f32x4 psc = {1.f, 1.f, 1.f, 1.f}, psh = {0.f, 0.f, 0.f, 0.f};
  constexpr unsigned OOB = 0x80000000u;
  unsigned hoff[2] = {OOB, OOB};
  const size_t img_floats = (size_t)a.H * a.W * a.in_cs;
  __amdgpu_buffer_rsrc_t rsrc_in;
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wpk), 0, a.wpk_bytes, 0x00020000);

  // load cursor: the (tile, chunk) whose global loads are issued next
  int ld_tile = tile0, ld_chunk = 0;
#define PIPE_ISSUE_LOADS()                                                                                  \
  {                                                                                                         \
    if (ld_chunk == 0) {                                                                                    \
      const int tt_ = min(ld_tile, t_end - 1);  /* past the end: harmless redundant loads of the last tile */ \
      const int tx_ = tt_ % a.tiles_x, t2_ = tt_ / a.tiles_x;                                               \
      const int ty0_ = (t2_ % a.tiles_y) * TH, tx0_ = tx_ * TW, n_ = t2_ / a.tiles_y;                       \
      _Pragma("unroll") for (int k = 0; k < 2; ++k) {                                                       \
        const int gy = ty0_ - 1 + (rrc[k] & 255), gx = tx0_ - 1 + (rrc[k] >> 8);                            \
        const bool ok = (k == 0 || r1) && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;     \
        hoff[k] = ok ? (unsigned)(gy * rowb + gx * pixb + (a.in_co + q2 * 4) * 4) : OOB;                    \
      }                                                                                                     \
      rsrc_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p_in) + (size_t)n_ * img_floats, 0,    \
                                                  a.in_bytes, 0x00020000);                                  \
    }                                                                                                       \
    if (IN_MODE != 0) {                                                                                     \
      psc = *reinterpret_cast<const f32x4*>(sS + ld_chunk * PK + q2 * 4);                                   \
      psh = *reinterpret_cast<const f32x4*>(sS + 1024 + ld_chunk * PK + q2 * 4);                            \
    }                                                                                                       \
    _Pragma("unroll") for (int k = 0; k < 2; ++k)                                                           \
      hreg[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_in, hoff[k], ld_chunk * PK * 4, 0)); \
    const int wbase_ = (cob * nst + ld_chunk) * NT * QB_FLOATS * 4;                                         \
    _Pragma("unroll") for (int j = 0; j < 2 * NT; ++j)                                                      \
      wreg[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, tid * 16, wbase_ + j * 8192, 0)); \
    if (++ld_chunk == nst) { ld_chunk = 0; ld_tile += per_cob; }                                            \
  }
  // registers -> LDS for the stage whose loads are in the registers (buffer index B)
#define PIPE_WRITE_STAGE(B)                                                                                 \
  {                                                                                                         \
    _Pragma("unroll") for (int k = 0; k < 2; ++k) {                                                         \
      if (k == 0 || r1) {                                                                                   \
        f32x4 v = hreg[k];                                                                                  \
        if (IN_MODE != 0) {                                                                                 \
          _Pragma("unroll") for (int e = 0; e < 4; ++e) v[e] = fmaxf(fmaf(v[e], psc[e], psh[e]), 0.f);      \
          if (hoff[k] == OOB) v = f32x4{0.f, 0.f, 0.f, 0.f};                                                \
        }                                                                                                   \
        *reinterpret_cast<f32x4*>(sR + r_lds[k]) = v;                                                       \
      }                                                                                                     \
    }                                                                                                       \
    f32x4* wdst = reinterpret_cast<f32x4*>(smem + (B) * BUF_FLOATS + B_BASE);                               \
    _Pragma("unroll") for (int j = 0; j < 2 * NT; ++j) wdst[tid + WINO_THREADS * j] = wreg[j];              \
  }
  // MFMA fragment offsets (floats, relative to the buffer base)
  const int m_tile = mt * 32 + li;
  const int a_off = (chalf * 8 * WTILES + m_tile) * 4 + lh * 2;                 // lane: 4 bf16 = k 4 lh .. 4 lh + 3
  const int b_off = B_BASE + (chalf * 8 * NB + nt * 32 + li) * 4 + lh * 2;      // sB[part][component][co][8 k] bf16

  f32x4 ssum = {0.f, 0.f, 0.f, 0.f}, ssq = {0.f, 0.f, 0.f, 0.f};
  const int co_l = cob * NB + nt * 32 + li;
  const float bias_v = (a.bias != nullptr && co_l < a.Cout) ? a.bias[co_l] : 0.f;

  // transform source offsets of the two raw rows of this thread's V row (the rotation depends on the raster index)
  int t_u[4], t_w[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    t_u[j] = pipe_raw_off<WIDE>((2 * t_ty + t_ra) * HC + 2 * t_tx + j, q2);
    t_w[j] = pipe_raw_off<WIDE>((2 * t_ty + t_rb) * HC + 2 * t_tx + j, q2);
  }
  // four transformed values -> the bf16 image(s) of the buffer: hi part, and the lo part QA_FLOATS further (NT == 2)
#define PIPE_STORE_V(DST, VAL)                                                                              \
  {                                                                                                         \
    const f32x4 v__ = (VAL);                                                                                \
    if (NT == 1) {                                                                                          \
      *reinterpret_cast<bf16x4*>(DST) = __builtin_convertvector(v__, bf16x4);                               \
    } else {                                                                                                \
      bf16x4 hi__, lo__;                                                                                    \
      bf16_split(v__, hi__, lo__);                                                                          \
      *reinterpret_cast<bf16x4*>(DST) = hi__;                                                               \
      *reinterpret_cast<bf16x4*>((DST) + QA_FLOATS) = lo__;                                                 \
    }                                                                                                       \
  }
  // one V row (4 components) of (tile, quad): sR -> sA of buffer B
#define PIPE_TRANSFORM(B)                                                                                   \
  {                                                                                                         \
    f32x4 t[4];                                                                                             \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                         \
      const f32x4 u = *reinterpret_cast<const f32x4*>(sR + t_u[j]);                                         \
      const f32x4 w = *reinterpret_cast<const f32x4*>(sR + t_w[j]);                                         \
      t[j] = u + t_sg * w;                                                                                  \
    }                                                                                                       \
    float* d_ = smem + (B) * BUF_FLOATS + t_dst;                                                            \
    PIPE_STORE_V(d_ + 0 * WTILES * 4, t[0] - t[2])                                                          \
    PIPE_STORE_V(d_ + 1 * WTILES * 4, t[1] + t[2])                                                          \
    PIPE_STORE_V(d_ + 2 * WTILES * 4, t[2] - t[1])                                                          \
    PIPE_STORE_V(d_ + 3 * WTILES * 4, t[1] - t[3])                                                          \
  }

  if (IN_MODE != 0) {
    for (int c = tid; c < a.Cin; c += WINO_THREADS) {
      sS[c] = p_scale[c];
      sS[1024 + c] = p_shift[c];
    }
    if (p_pool != nullptr && tid < NB) {
      const int co_ = cob * NB + tid;
      sG[tid] = (co_ < a.Cout && a.pool_gamma[co_] < 0.f) ? -1.f : 1.f;
    }
    __syncthreads();
  } else if (a.bnr_mode != 0) {
    // fused BatchNorm-backward sums of the layer below (ConvArgs::bnr_*, as in conv_wino_pipe_kernel)
    if (tid < NB) {
      const int co_ = cob * NB + tid;
      float q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f;
      if (co_ < a.Cout) {
        if (a.bnr_mode == 1) {
          const float is_ = a.bnr_p3[prob][co_];
          q0 = a.bnr_p0[prob][co_]; q1 = a.bnr_p1[prob][co_]; q2 = is_; q3 = -a.bnr_p2[prob][co_] * is_;
        } else {
          const float g_ = a.bnr_p1[prob][co_];
          q0 = a.bnr_p0[prob][co_]; q1 = g_ != 0.f ? 1.f / g_ : 0.f;
        }
      }
      sS[tid] = q0; sS[NB + tid] = q1; sS[2 * NB + tid] = q2; sS[3 * NB + tid] = q3;
    }
    __syncthreads();
  }
  const float* const p_bnr = prob ? a.bnr_t2 : a.bnr_t;
  // ---- prologue: stage 0 into buffer 0, loads of stage 1 in flight ----
  PIPE_ISSUE_LOADS()
  PIPE_WRITE_STAGE(0)
  __syncthreads();
  PIPE_TRANSFORM(0)
  PIPE_ISSUE_LOADS()
  __syncthreads();

  f32x16 acc[8];
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;

  // components C and C + 1 of this wave's half: fragment reads, then 2 x 4 MFMAs on two alternating accumulators.
  // The staging work of the next stage is sliced BETWEEN the MFMA groups (fenced with sched_barrier so that the
  // compiler keeps the order): a wave that has just issued an MFMA group owns the issue slots of the ~250 cycles the
  // matrix pipe needs for it and for the group of the other wave of its SIMD.
#define PIPE_FRAG(C)                                                                                        \
  const s16x4 a0_##C = *reinterpret_cast<const s16x4*>(cA + a_off + (C) * WTILES * 4);                      \
  const s16x4 a1_##C = *reinterpret_cast<const s16x4*>(cA + a_off + ((C) + 1) * WTILES * 4);               \
  const s16x4 b0_##C = *reinterpret_cast<const s16x4*>(cA + b_off + (C) * NB * 4);                          \
  const s16x4 b1_##C = *reinterpret_cast<const s16x4*>(cA + b_off + ((C) + 1) * NB * 4);                    \
  const s16x4 a0l_##C = NT == 2 ? *reinterpret_cast<const s16x4*>(cA + QA_FLOATS + a_off + (C) * WTILES * 4) : a0_##C;          \
  const s16x4 a1l_##C = NT == 2 ? *reinterpret_cast<const s16x4*>(cA + QA_FLOATS + a_off + ((C) + 1) * WTILES * 4) : a1_##C;    \
  const s16x4 b0l_##C = NT == 2 ? *reinterpret_cast<const s16x4*>(cA + QB_FLOATS + b_off + (C) * NB * 4) : b0_##C;              \
  const s16x4 b1l_##C = NT == 2 ? *reinterpret_cast<const s16x4*>(cA + QB_FLOATS + b_off + ((C) + 1) * NB * 4) : b1_##C;
#define PIPE_MFMA_LO(C)                                                                                     \
  if (NT == 2) {  /* small terms first */                                                                   \
    acc[C] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a0l_##C, b0_##C, acc[C], 0, 0, 0);                    \
    acc[(C) + 1] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a1l_##C, b1_##C, acc[(C) + 1], 0, 0, 0);        \
    acc[C] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a0_##C, b0l_##C, acc[C], 0, 0, 0);                    \
    acc[(C) + 1] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a1_##C, b1l_##C, acc[(C) + 1], 0, 0, 0);        \
  }                                                                                                         \
  acc[C] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a0_##C, b0_##C, acc[C], 0, 0, 0);                       \
  acc[(C) + 1] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a1_##C, b1_##C, acc[(C) + 1], 0, 0, 0);
#define PIPE_MFMA_HI(C)
#define PIPE_FENCE() __builtin_amdgcn_sched_barrier(0)

  int tile = tile0, chunk = 0;
  for (int g = 0; g < nstages; ++g) {
    const int buf = g & 1;
    const float* const cA = smem + buf * BUF_FLOATS;
    float* const nB = smem + (buf ^ 1) * BUF_FLOATS;
    // ---- first half: components 0..3 of this wave's half || registers (stage g+1) -> LDS, loads of stage g+2 ----
    {
      PIPE_FRAG(0)
      PIPE_FENCE();
      PIPE_MFMA_LO(0)
      PIPE_FENCE();
#pragma unroll
      for (int k = 0; k < 2; ++k) {  // raw halo of stage g+1 -> sR (BatchNorm + ReLU of the producer, zero padding)
        if (k == 0 || r1) {
          f32x4 v = hreg[k];
          if (IN_MODE != 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(fmaf(v[e], psc[e], psh[e]), 0.f);
            if (hoff[k] == OOB) v = f32x4{0.f, 0.f, 0.f, 0.f};
          }
          *reinterpret_cast<f32x4*>(sR + r_lds[k]) = v;
        }
      }
      PIPE_FENCE();
      PIPE_MFMA_HI(0)
      PIPE_FENCE();
      PIPE_FRAG(2)
      f32x4* wdst = reinterpret_cast<f32x4*>(nB + B_BASE);  // weights of stage g+1 -> sB of the other buffer
#pragma unroll
      for (int j = 0; j < 2 * NT; ++j) wdst[tid + WINO_THREADS * j] = wreg[j];
      PIPE_FENCE();
      PIPE_MFMA_LO(2)
      PIPE_FENCE();
      PIPE_ISSUE_LOADS()
      PIPE_FENCE();
      PIPE_MFMA_HI(2)
    }
    // the fragments of components 4, 5 come from the SAME buffer: read them before the barrier so that the matrix pipe
    // restarts right after it
    PIPE_FRAG(4)
    __syncthreads();
    // ---- second half: components 4..7 || transform of stage g+1: sR -> sA of the other buffer ----
    {
      const f32x4 u0 = *reinterpret_cast<const f32x4*>(sR + t_u[0]), w0 = *reinterpret_cast<const f32x4*>(sR + t_w[0]);
      const f32x4 u1 = *reinterpret_cast<const f32x4*>(sR + t_u[1]), w1 = *reinterpret_cast<const f32x4*>(sR + t_w[1]);
      const f32x4 u2 = *reinterpret_cast<const f32x4*>(sR + t_u[2]), w2 = *reinterpret_cast<const f32x4*>(sR + t_w[2]);
      const f32x4 u3 = *reinterpret_cast<const f32x4*>(sR + t_u[3]), w3 = *reinterpret_cast<const f32x4*>(sR + t_w[3]);
      PIPE_FENCE();
      PIPE_MFMA_LO(4)
      PIPE_FENCE();
      f32x4 t[4];
      t[0] = u0 + t_sg * w0;
      t[1] = u1 + t_sg * w1;
      t[2] = u2 + t_sg * w2;
      t[3] = u3 + t_sg * w3;
      float* d_ = nB + t_dst;
      PIPE_STORE_V(d_ + 0 * WTILES * 4, t[0] - t[2])
      PIPE_STORE_V(d_ + 1 * WTILES * 4, t[1] + t[2])
      PIPE_FENCE();
      PIPE_MFMA_HI(4)
      PIPE_FENCE();
      PIPE_FRAG(6)
      PIPE_STORE_V(d_ + 2 * WTILES * 4, t[2] - t[1])
      PIPE_STORE_V(d_ + 3 * WTILES * 4, t[1] - t[3])
      PIPE_FENCE();
      PIPE_MFMA_LO(6)
      PIPE_MFMA_HI(6)
    }
    __syncthreads();

    if (++chunk == nst) {
      // ---- tile epilogue: the consumed (sA, sB) pair of this stage is the 64 KB staging tile ----
      const int tx_i = tile % a.tiles_x, t2 = tile / a.tiles_x;
      const int ty0 = (t2 % a.tiles_y) * TH, tx0 = tx_i * TW, n = t2 / a.tiles_y;
      const bool full = (ty0 + TH <= a.H) && (tx0 + TW <= a.W);
      // Two rounds over the tile halves (bit 4 of the tile slot = accumulator registers 0..7 / 8..15): in each round
      // BOTH component halves write their partial outputs of 32 tiles to two 32 KB staging half-tiles, which meet in
      // the 16-byte store loop.  Compact tile index csl = (sl & 15) | (sl >> 5) << 4.
      // staging tile: dedicated (NT == 1: the bf16 images are too small) or the consumed 64 KB buffer (NT == 2)
      float* const sStage = NT == 1 ? sStageDedicated : smem + buf * BUF_FLOATS;
      float* const stg = sStage + chalf * (TH * TW * NB / 2);
      const float bz = chalf == 0 ? bias_v : 0.f;
      const int q16 = tid & 15;
      const int co4 = cob * NB + q16 * 4;
      const int nvalid = min(4, a.Cout - co4);
#pragma unroll
      for (int rd = 0; rd < 2; ++rd) {
#pragma unroll
        for (int r8 = 0; r8 < 8; ++r8) {
          const int r = rd * 8 + r8;
          const int csl = ((r8 & 3) + 8 * (r8 >> 2) + 4 * lh) | (mt << 4);
          const int cty = csl / TTX, ctx = csl % TTX;
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
          float* o = stg + ((2 * cty) * TW + 2 * ctx) * NB + nt * 32 + li;
          o[0] = s0[0] + s0[1] + s0[2] + bz;
          o[NB] = s0[1] - s0[2] - s0[3] + bz;
          o[TW * NB] = s1[0] + s1[1] + s1[2] + bz;
          o[TW * NB + NB] = s1[1] - s1[2] - s1[3] + bz;
        }
        constexpr int NSTORE = (TH * TW * 8) / WINO_THREADS;
        const bool fast = full && (cob + 1) * NB <= a.Cout;  // block-uniform
        f32x4 tpre[NSTORE];
        if (IN_MODE == 0 && a.bnr_mode != 0 && fast) {  // requested before the barrier (see conv_wino_pipe_kernel)
#pragma unroll
          for (int k = 0; k < NSTORE; ++k) {
            const int lp = (tid >> 4) + (WINO_THREADS / 16) * k;
            const int crow = lp / TW, ccol = lp - crow * TW;
            const int csl = (crow >> 1) * TTX + (ccol >> 1);
            const int sl = (csl & 15) | (rd << 4) | ((csl >> 4) << 5);
            const int oy = ty0 + 2 * (sl / TTX) + (crow & 1), ox = tx0 + ccol;
            tpre[k] = *reinterpret_cast<const f32x4*>(p_bnr + ((size_t)(n * a.H + oy) * a.W + ox) * a.bnr_cs + a.bnr_co + co4);
          }
        }
        __syncthreads();
        const float* const s0p = sStage;
        if (fast) {
          // straight-line fast path (as in conv_wino_pipe_kernel): all eight LDS reads of the round first
          f32x4 va[NSTORE], vb[NSTORE];
#pragma unroll
          for (int k = 0; k < NSTORE; ++k) {
            const int lp = (tid >> 4) + (WINO_THREADS / 16) * k;
            va[k] = *reinterpret_cast<const f32x4*>(s0p + lp * NB + q16 * 4);
            vb[k] = *reinterpret_cast<const f32x4*>(s0p + TH * TW * NB / 2 + lp * NB + q16 * 4);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int k = 0; k < NSTORE; ++k) {
            const int lp = (tid >> 4) + (WINO_THREADS / 16) * k;
            const int crow = lp / TW, ccol = lp - crow * TW;
            const int csl = (crow >> 1) * TTX + (ccol >> 1);
            const int sl = (csl & 15) | (rd << 4) | ((csl >> 4) << 5);
            const int oy = ty0 + 2 * (sl / TTX) + (crow & 1), ox = tx0 + ccol;
            const f32x4 v = pk4_add(va[k], vb[k]);
            if (IN_MODE == 0 && a.bnr_mode != 0) {
              const f32x4 t = tpre[k];
              const f32x4 q0 = *reinterpret_cast<const f32x4*>(sS + q16 * 4), q1 = *reinterpret_cast<const f32x4*>(sS + NB + q16 * 4);
              f32x4 dz, xh;
              if (a.bnr_mode == 1) {
                const f32x4 q2 = *reinterpret_cast<const f32x4*>(sS + 2 * NB + q16 * 4), q3 = *reinterpret_cast<const f32x4*>(sS + 3 * NB + q16 * 4);
                const f32x4 z = pk4_fma(t, q0, q1);
                xh = pk4_fma(t, q2, q3);
#pragma unroll
                for (int e = 0; e < 4; ++e) dz[e] = z[e] > 0.f ? v[e] : 0.f;
              } else {
                xh = (t - q0) * q1;
#pragma unroll
                for (int e = 0; e < 4; ++e) dz[e] = t[e] > 0.f ? v[e] : 0.f;
              }
              ssum = pk4_add(ssum, dz);
              ssq = pk4_fma(dz, xh, ssq);
            } else {
              ssum = pk4_add(ssum, v);
              ssq = pk4_fma(v, v, ssq);
            }
            *reinterpret_cast<f32x4*>(p_out + ((size_t)(n * a.H + oy) * a.W + ox) * a.out_cs + a.out_co + co4) = v;
          }
        } else
#pragma unroll
        for (int k = 0; k < (TH * TW * 8) / WINO_THREADS; ++k) {
          const int lp = (tid >> 4) + (WINO_THREADS / 16) * k;  // compact pixel of the half tile
          const int crow = lp / TW, ccol = lp - crow * TW;
          const int csl = (crow >> 1) * TTX + (ccol >> 1);
          const int sl = (csl & 15) | (rd << 4) | ((csl >> 4) << 5);
          const int oy = ty0 + 2 * (sl / TTX) + (crow & 1), ox = tx0 + ccol;
          if (nvalid > 0 && (full || (oy < a.H && ox < a.W))) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(s0p + lp * NB + q16 * 4) +
                            *reinterpret_cast<const f32x4*>(s0p + TH * TW * NB / 2 + lp * NB + q16 * 4);
            if (IN_MODE == 0 && a.bnr_mode != 0) {
              const f32x4 t = *reinterpret_cast<const f32x4*>(p_bnr + ((size_t)(n * a.H + oy) * a.W + ox) * a.bnr_cs + a.bnr_co + co4);
              const f32x4 q0 = *reinterpret_cast<const f32x4*>(sS + q16 * 4), q1 = *reinterpret_cast<const f32x4*>(sS + NB + q16 * 4);
              f32x4 dz, xh;
              if (a.bnr_mode == 1) {
                const f32x4 q2 = *reinterpret_cast<const f32x4*>(sS + 2 * NB + q16 * 4), q3 = *reinterpret_cast<const f32x4*>(sS + 3 * NB + q16 * 4);
                const f32x4 z = pk4_fma(t, q0, q1);
                xh = pk4_fma(t, q2, q3);
#pragma unroll
                for (int e = 0; e < 4; ++e) dz[e] = z[e] > 0.f ? v[e] : 0.f;
              } else {
                xh = (t - q0) * q1;
#pragma unroll
                for (int e = 0; e < 4; ++e) dz[e] = t[e] > 0.f ? v[e] : 0.f;
              }
              ssum += dz;
              ssq += dz * xh;
            } else {
              ssum += v;
              ssq += v * v;
            }
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
        if (IN_MODE != 0 && p_pool != nullptr) {
          const int w_ = tid >> 4, wr = w_ / TTX, wc = w_ - wr * TTX;
          const int sl = (w_ & 15) | (rd << 4) | ((w_ >> 4) << 5);
          const int py = (ty0 >> 1) + sl / TTX, px = (tx0 >> 1) + sl % TTX;
          if (nvalid > 0 && 2 * py < a.H && 2 * px < a.W) {
            const f32x4 sg = *reinterpret_cast<const f32x4*>(sG + q16 * 4);
            f32x4 m;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int lp = (2 * wr + (i >> 1)) * TW + 2 * wc + (i & 1);
              const f32x4 v = pk4_add(*reinterpret_cast<const f32x4*>(s0p + lp * NB + q16 * 4),
                                      *reinterpret_cast<const f32x4*>(s0p + TH * TW * NB / 2 + lp * NB + q16 * 4)) * sg;
              if (i == 0) m = v;
              else {
#pragma unroll
                for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], v[e]);
              }
            }
            m *= sg;
            float* pp = p_pool + ((size_t)(n * (a.H >> 1) + py) * (a.W >> 1) + px) * a.Cout + co4;
            if (nvalid == 4) *reinterpret_cast<f32x4*>(pp) = m;
            else { pp[0] = m[0]; if (nvalid > 1) pp[1] = m[1]; if (nvalid > 2) pp[2] = m[2]; }
          }
        }
        __syncthreads();  // round 1 / the next-but-one stage overwrite the staging half-tiles
      }
#pragma unroll
      for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
      chunk = 0;
      tile += per_cob;
    }
  }
#undef PIPE_ISSUE_LOADS
#undef PIPE_WRITE_STAGE
#undef PIPE_TRANSFORM
#undef PIPE_STORE_V
#undef PIPE_FRAG
#undef PIPE_MFMA_LO
#undef PIPE_MFMA_HI
#undef PIPE_FENCE

  if (p_stats != nullptr) {
    __syncthreads();
    float* red = smem;
    *reinterpret_cast<f32x4*>(red + tid * 8) = ssum;
    *reinterpret_cast<f32x4*>(red + tid * 8 + 4) = ssq;
    __syncthreads();
    if (tid < 128) {
      const int ch = tid >> 1, which = tid & 1;
      float t = 0.f;
      for (int gq = 0; gq < WINO_THREADS / 16; ++gq) t += red[(gq * 16 + (ch >> 2)) * 8 + which * 4 + (ch & 3)];
      const int co = cob * NB + ch;
      if (co < a.Cout)
        acc_add_stats_or_grad(p_stats + (size_t)(blockIdx.x % NREP) * 2 * a.Cout + which * a.Cout + co, (double)t, IN_MODE == 0 && a.bnr_mode != 0);
    }
  }
}

// Winograd F(3x3, 2x2) weight gradient with bf16 matrix-core operands: wgrad_wino_kernel (conv_wino.hip.h) with the
// transformed input values V and the transformed dY values D rounded to bf16 in registers and one
// v_mfma_f32_32x32x8_bf16 per 8 Winograd tiles (K = tiles; lane half lh supplies 4 consecutive tiles) instead of four
// fp32 MFMAs per tile pair; fp32 accumulation, fp32 partial slabs, the same reduce kernel.
// NT = 2: hi + lo split of both operands in registers, three MFMAs per product block (see conv_wino_bf16_kernel).
template <int IN_MODE, bool WIDE, int NT = 1>
__global__ __launch_bounds__(512) void wgrad_wino_bf16_kernel(const WgradArgs a) {
  using G = WgradWinoGeom<WIDE>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sX = smem;
  float* sD = smem + G::X_FLOATS;
  float* sS = smem + G::X_FLOATS + G::D_FLOATS;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
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
  unsigned xmask = 0, dmask = 0;

  // halo / dY raster positions of this thread's staging slots (slot i = pixel (tid >> 4) + 32 i), packed r | c << 8
  int xrc[NX], drc[ND];
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    const int pp = (tid >> 4) + 32 * i, r = pp / G::WT;
    xrc[i] = pp < G::HT * G::WT ? (r | ((pp - r * G::WT) << 8)) : 0xFFFF;  // 0xFFFF: r = 255 is never inside the image
  }
#pragma unroll
  for (int i = 0; i < ND; ++i) {
    const int pp = (tid >> 4) + 32 * i, r = pp / G::TW;
    drc[i] = r | ((pp - r * G::TW) << 8);
  }
  const int xpix = a.in_cs * 4, xrow = a.W * xpix, dpix = a.dout_cs * 4, drow = a.W * dpix;
  const int xq = civalid ? (a.in_co + ci0) * 4 : -1, dq = covalid ? (a.dout_co + co0) * 4 : -1;
  constexpr unsigned OOB = 0x80000000u;
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
    xmask = 0; dmask = 0;                                                                                     \
    _Pragma("unroll") for (int i = 0; i < NX; ++i) {                                                          \
      const int gy = ty0_ - 1 + (xrc[i] & 255), gx = tx0_ - 1 + (xrc[i] >> 8);                                \
      const bool ok = xq >= 0 && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;                \
      xreg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(                              \
          rx_, ok ? (unsigned)(gy * xrow + gx * xpix + xq) : OOB, 0, 0));                                     \
      xmask |= (ok ? 1u : 0u) << i;                                                                           \
    }                                                                                                         \
    _Pragma("unroll") for (int i = 0; i < ND; ++i) {                                                          \
      const int gy = ty0_ + (drc[i] & 255), gx = tx0_ + (drc[i] >> 8);                                        \
      const bool ok = dq >= 0 && gy < a.H && gx < a.W;                                                        \
      dreg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(                              \
          rd_, ok ? (unsigned)(gy * drow + gx * dpix + dq) : OOB, 0, 0));                                     \
      dmask |= (ok ? 1u : 0u) << i;                                                                           \
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
    if (!(a.ablate & 2)) {
      const int cur_prob = tile >= a.ntiles ? 1 : 0;
      f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
      if (IN_MODE != 0) {
        sc = *reinterpret_cast<const f32x4*>(sS + cur_prob * 128 + q16 * 4);
        sh = *reinterpret_cast<const f32x4*>(sS + cur_prob * 128 + 64 + q16 * 4);
      }
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        const int pp = (tid >> 4) + 32 * i;
        if (pp < G::HT * G::WT) {
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
          if ((xmask >> i) & 1u) {
            v = xreg[i];
            if (IN_MODE != 0) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = fmaxf(fmaf(v[e], sc[e], sh[e]), 0.f);
            }
          }
          *reinterpret_cast<f32x4*>(sX + pp * 64 + q16 * 4) = v;
        }
      }
#pragma unroll
      for (int i = 0; i < ND; ++i) {
        const int pp = (tid >> 4) + 32 * i;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if ((dmask >> i) & 1u) v = dreg[i];
        *reinterpret_cast<f32x4*>(sD + pp * 64 + q16 * 4) = v;
      }
    }
    __syncthreads();
    {
      const int nxt = min(tile + 1, t_end - 1);  // unconditional prefetch (redundant on the last tile)
      WGW_ISSUE(nxt)
      __builtin_amdgcn_sched_barrier(0);
    }
    if (!(a.ablate & 8))
#pragma unroll 1
    for (int s = 0; s < 4; ++s) {  // 8 Winograd tiles per MFMA: lane half lh supplies tiles 8 s + 4 lh .. + 3
      bf16x4 Vb[4][2], Db[4], Vl[4][2], Dl[4];
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        const int t = 8 * s + 4 * lh + tt;
        const int ty = t / G::TTX, tx = t - ty * G::TTX;
        const float* xa = sX + ((2 * ty + ra) * G::WT + 2 * tx) * 64 + 2 * li;
        const float* xb = sX + ((2 * ty + rb) * G::WT + 2 * tx) * 64 + 2 * li;
        f32x2 T[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const f32x2 u = *reinterpret_cast<const f32x2*>(xa + c * 64), w = *reinterpret_cast<const f32x2*>(xb + c * 64);
          T[c][0] = fmaf(sg, w[0], u[0]);
          T[c][1] = fmaf(sg, w[1], u[1]);
        }
        const f32x2 V[4] = {T[0] - T[2], T[1] + T[2], T[2] - T[1], T[1] - T[3]};
        const float* db = sD + ((2 * ty) * G::TW + 2 * tx) * 64 + coh * 32 + li;
        const float r0 = c0 * db[0] + c1 * db[G::TW * 64];
        const float r1 = c0 * db[64] + c1 * db[G::TW * 64 + 64];
        const float Dv[4] = {r0, r0 + r1, r0 - r1, -r1};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const __bf16 hi = (__bf16)V[j][e];
            Vb[j][e][tt] = hi;
            if (NT == 2) Vl[j][e][tt] = (__bf16)(V[j][e] - (float)hi);
          }
          const __bf16 dh = (__bf16)Dv[j];
          Db[j][tt] = dh;
          if (NT == 2) Dl[j][tt] = (__bf16)(Dv[j] - (float)dh);
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          if (NT == 2) {  // small terms first
            acc[j][e] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(__builtin_bit_cast(s16x4, Vl[j][e]), __builtin_bit_cast(s16x4, Db[j]), acc[j][e], 0, 0, 0);
            acc[j][e] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(__builtin_bit_cast(s16x4, Vb[j][e]), __builtin_bit_cast(s16x4, Dl[j]), acc[j][e], 0, 0, 0);
          }
          acc[j][e] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(__builtin_bit_cast(s16x4, Vb[j][e]), __builtin_bit_cast(s16x4, Db[j]), acc[j][e], 0, 0, 0);
        }
      }
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


// OIHW 3x3 weights -> bf16(G g G^T) in the LDS image of conv_wino_bf16_kernel: [cob][chunk8][part][component][co 64][k 8]
// (parts = 1: the bf16 value; parts = 2: hi = bf16(u), lo = bf16(u - hi))
__global__ void pack_weights_wino8_bf16_kernel(const float* __restrict__ w, __bf16* __restrict__ dst, int Cout_w, int Cin_w,
                                               int transpose_flip, int nchunks_total, int chunk_off, int cob_off, int ncob,
                                               int nchunks, int parts) {
  const int per_chunk = WC * NB * PK;
  const int total = ncob * nchunks * per_chunk;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  int t = idx;
  const int kk = t & 7;
  t >>= 3;
  const int nn = t & 63;
  t >>= 6;
  const int comp = t % WC;
  t /= WC;
  const int chunk = t % nchunks;
  const int cob = t / nchunks;
  const int co = cob * NB + nn;
  const int ci = chunk * PK + kk;
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
  float r[3];
#pragma unroll
  for (int x = 0; x < 3; ++x)
    r[x] = i == 0 ? k[0][x] : i == 1 ? 0.5f * (k[0][x] + k[1][x] + k[2][x]) : i == 2 ? 0.5f * (k[0][x] - k[1][x] + k[2][x]) : k[2][x];
  const float u = j == 0 ? r[0] : j == 1 ? 0.5f * (r[0] + r[1] + r[2]) : j == 2 ? 0.5f * (r[0] - r[1] + r[2]) : r[2];
  const size_t o = ((size_t)(cob + cob_off) * nchunks_total + chunk + chunk_off) * parts * per_chunk + (comp * NB + nn) * PK + kk;
  const __bf16 hi = (__bf16)u;
  dst[o] = hi;
  if (parts == 2) dst[o + per_chunk] = (__bf16)(u - (float)hi);
}

}  // namespace sspk
