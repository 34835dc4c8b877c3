// Winograd F(2x2, 3x3) convolution, software-pipelined variant of conv_wino_kernel (same math, same tile / wave
// decomposition, see conv_wino.hip.h): the K loop runs in 8-channel STAGES with double-buffered LDS images, so that
// the staging of stage g+1 is interleaved with the MFMAs of stage g instead of stalling the matrix pipe:
//
//   stage g, first half : 16 MFMAs / wave on sA[g&1]  ||  raw halo (g+1): registers -> sR; halo loads of stage g+2
//   barrier
//   stage g, second half: 16 MFMAs / wave             ||  transform sR -> sA[~g&1]
//   barrier
//
// The weight (B) fragments come straight from L2 into the MFMA operand registers, one component pair ahead (template
// parameter GB, default; GB = false stages them through sB like the input: conv algo 5).
// LDS: 2 x (32 KB transformed input + 32 KB sB = epilogue staging / LDS-staged weights) + 10.6 KB raw halo + 8 KB
// BatchNorm parameters = 146.6 KB, one block per CU.
// The flat stage index runs over (tile, 8-channel chunk) pairs of the block's persistent tile list, so the pipeline
// never drains between tiles; the tile epilogue (output transform, the two component halves meeting in ONE 64 KB
// staging tile = the just-consumed sA/sB pair, 16-byte stores, BatchNorm statistics - forward: of the output; data
// gradient: pass 1 of the BatchNorm backward of the layer below, ConvArgs::bnr_mode) sits between two stages.
// Weights: pack_weights_wino8_kernel, [cob][chunk8][component][h][64][4].
#pragma once
#include "conv_wino.hip.h"

#ifndef PIPE_ABL
#define PIPE_ABL 0  // compile-time perf ablation (tools/archive/ablate_pipe.py): 1 no epilogue, 4 no end-of-stage barrier, 8 no MFMA,
                    // 16 no global stores, 512 no halo loads, 1024 cycle-stamp trace, 2048 no weight-fragment loads (B operands
                    // constant), 4096 no LDS fragment reads (A operands constant), 8192 halo loads confined to 64 KB (cache
                    // hits).  CAUTION: variants that make operands constant / zero also lower the power draw - the same
                    // kernel is 2.6 % (sustained) to 13 % (short bursts) faster on zero data (profiles/r02_conv_power_probe.txt)
#endif

namespace sspk {

constexpr int PK = 8;                              // channels per stage
constexpr int PA_FLOATS = WC * WTILES * PK;        // 8192 floats = 32 KB
constexpr int PB_FLOATS = WC * PK * NB;            // 8192 floats = 32 KB
constexpr int PR_FLOATS = (WHALO + 7) / 8 * 8 * PK;  // 2752 floats: whole 8-pixel rotation groups
constexpr int PS_FLOATS = 2 * 1024;                // BatchNorm scale | shift of up to 1024 input channels
constexpr int PG_FLOATS = NB;                       // +-1 per output channel of the block: sign of gamma (pooled output)
constexpr int PIPE_LDS_BYTES = (2 * (PA_FLOATS + PB_FLOATS) + PR_FLOATS + PS_FLOATS + PG_FLOATS) * 4;

// raw halo pixel p (raster index), quad q (0/1) -> float offset in sR.  ds_read_b128 is serviced in four groups of 16
// NON-contiguous lanes ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, + 32) on 64 banks (256 B): the stride-2 pixel reads
// of the transform (lane = (tile, quad)) are conflict-free when the 8 pixels of a lane group fall on 8 different 32-byte
// positions of the 256-byte bank row.  8x32 tiles (34-pixel halo rows): the pixels are rotated inside aligned groups of 8 by a
// function of the group index found by exhaustive search over the lane-group model (tools/lds_conflicts.py): 8 -> 4 LDS cycles
// per read.  32x8 tiles (10-pixel halo rows, a lane group spans four of them): the same family only reached 6.5 cycles
// (`SQ_LDS_BANK_CONFLICT / SQ_INSTS_LDS` 0.77, profiles/r05_ssp_pmc_sq_summary.txt); round 6: plain raster order with the pixel
// column XOR-ed by bit 1 of the halo row (as conv_wino_p2_kernel) is conflict-free: 4 cycles.  The 16-byte writes (8 contiguous
// lanes = 4 consecutive raster pixels = 128 contiguous bytes either way, 32 banks) stay conflict-free.
#ifndef PIPE_XOR
#define PIPE_XOR 1
#endif
template <bool WIDE>
__device__ __forceinline__ int pipe_raw_off(int p, int q) {
  if (!WIDE && PIPE_XOR) {
    const int r = p / 10, c = p - r * 10;
    return (r * 10 + (c ^ ((r >> 1) & 1))) * PK + q * 4;
  }
  const int b = p >> 3;
  const int g = WIDE ? (b + 6 * (b >> 1)) : ((b >> 1) + 6 * (b >> 2));
  return ((p & ~7) + ((p + g) & 7)) * PK + q * 4;
}

// Output transform of one epilogue round for one wave: its 8 components (rows 0,1 or rows 2,3 of M) of accumulator
// registers rd*8 .. rd*8+7 -> partial outputs Y = A^T M A (2x2 pixels per tile) in the staging half-tile `o`.
// CHALF 0: rows 0,1 -> top = m0 + m1, bottom = m1;  CHALF 1: rows 2,3 -> top = m2, bottom = -m2 - m3.
template <int CHALF, int TTX, int TW>
__device__ __forceinline__ void pipe_out_rows(const f32x16 (&acc)[8], int rd, int lh, int mt, float* __restrict__ o) {
#pragma unroll
  for (int r8 = 0; r8 < 8; r8 += 2) {
    const int r = rd * 8 + r8;
    f32x2 top[4], bot[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x2 a = {acc[j][r], acc[j][r + 1]}, b = {acc[4 + j][r], acc[4 + j][r + 1]};
      if (CHALF == 0) { top[j] = pk_add(a, b); bot[j] = b; }
      else { top[j] = a; bot[j] = pk_nadd(a, b); }
    }
    // three groups of four independent packed instructions (the asm statements keep their program order: a nested
    // pk_add(pk_add(..)) would issue every dependent pair back to back and stall on the result)
    const f32x2 s00 = pk_add(top[0], top[1]), s01 = pk_sub(top[1], top[2]);
    const f32x2 s10 = pk_add(bot[0], bot[1]), s11 = pk_sub(bot[1], bot[2]);
    const f32x2 y00 = pk_add(s00, top[2]), y01 = pk_sub(s01, top[3]);
    const f32x2 y10 = pk_add(s10, bot[2]), y11 = pk_sub(s11, bot[3]);
    const int csl = ((r8 & 3) + 8 * (r8 >> 2) + 4 * lh) | (mt << 4);  // even r8: tiles csl, csl + 1 are x neighbours
    const int cty = csl / TTX, ctx = csl % TTX;
    float* p = o + ((2 * cty) * TW + 2 * ctx) * NB;
    p[0] = y00[0]; p[NB] = y01[0]; p[2 * NB] = y00[1]; p[3 * NB] = y01[1];
    p[TW * NB] = y10[0]; p[TW * NB + NB] = y11[0]; p[TW * NB + 2 * NB] = y10[1]; p[TW * NB + 3 * NB] = y11[1];
  }
}

// GB (default; false = conv algo 5): the weight (B) fragments are loaded from global memory / L2 straight into the MFMA
// operand registers, one component pair ahead of their use, instead of being staged through LDS (the packed weight
// image IS the fragment layout): 32 KB less LDS writes and 64 KB less LDS reads per stage, 8 registers less; the sB
// halves of the LDS image then only serve as the epilogue's staging tile.  +2.3 % on the pair step (1938 vs 1893).
template <int IN_MODE, bool WIDE, bool GB = true>
__global__ __launch_bounds__(WINO_THREADS) void conv_wino_pipe_kernel(const ConvArgs a) {
  constexpr int TTX = WIDE ? 16 : 4;
  constexpr int TH = WIDE ? 8 : 32, TW = WIDE ? 32 : 8;
  constexpr int HC = TW + 2;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // [sA0][sB0][sA1][sB1][sR]: each (sA, sB) pair doubles as the 64 KB output staging tile of the epilogue
  float* const sR = smem + 2 * (PA_FLOATS + PB_FLOATS);
  float* const sS = sR + PR_FLOATS;  // scale[Cin] | shift[Cin] of the producer's BatchNorm (IN_MODE 1)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: the per-wave roles below become uniform branches
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
  const f32x2 t_sg2 = {t_sg, t_sg};
  // sA: [component][tile][8 channels]; the two channel quads of a tile are swapped for tiles 16-31 / 48-63, which puts the
  // 16 lanes of every ds_read_b128 lane group of the MFMA fragment reads (lane = tile) on 16 different 16-byte slots of
  // the bank row (the former (tile >> 2) & 1 swizzle assumed contiguous lane groups and was 2-way conflicted)
  const int t_dst = ((t_row * 4) * WTILES + t_tile) * PK + ((q2 ^ ((t_tile >> 4) & 1)) << 2);
  const int pixb = a.in_cs * 4, rowb = a.W * pixb;
  f32x4 hreg[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, wreg[4];
  f32x4 psc = {1.f, 1.f, 1.f, 1.f}, psh = {0.f, 0.f, 0.f, 0.f};
  constexpr unsigned OOB = 0x80000000u;
  unsigned hoff[2] = {OOB, OOB};
  const size_t img_floats = (size_t)a.H * a.W * a.in_cs;
  __amdgpu_buffer_rsrc_t rsrc_in;
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wpk), 0, a.wpk_bytes, 0x00020000);

  // load cursor: the (tile, chunk) whose global loads are issued next
  int ld_tile = tile0, ld_chunk = 0;
#define PIPE_ISSUE_HALO()                                                                                   \
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
    if (!(PIPE_ABL & 512))                                                                                  \
    _Pragma("unroll") for (int k = 0; k < 2; ++k)                                                           \
      hreg[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(                            \
          rsrc_in, (PIPE_ABL & 8192) ? (hoff[k] & 0xFFFFu) : hoff[k], ld_chunk * PK * 4, 0));               \
  }
  // weights of the same (tile, chunk); advances the load cursor
#define PIPE_ISSUE_W()                                                                                      \
  {                                                                                                         \
    const int wbase_ = (cob * nst + ld_chunk) * PB_FLOATS * 4;                                              \
    if (!GB)                                                                                                \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                           \
      wreg[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, tid * 16, wbase_ + j * 8192, 0)); \
    if (++ld_chunk == nst) { ld_chunk = 0; ld_tile += per_cob; }                                            \
  }
#define PIPE_ISSUE_LOADS() { PIPE_ISSUE_HALO() PIPE_ISSUE_W() }
  // registers -> LDS for the stage whose loads are in the registers (buffer index B)
#define PIPE_WRITE_STAGE(B)                                                                                 \
  {                                                                                                         \
    _Pragma("unroll") for (int k = 0; k < 2; ++k) {                                                         \
      if (k == 0 || r1) {                                                                                   \
        f32x4 v = hreg[k];                                                                                  \
        if (IN_MODE != 0) v = bn_relu_quad(v, psc, psh, hoff[k] == OOB);                                    \
        *reinterpret_cast<f32x4*>(sR + r_lds[k]) = v;                                                       \
      }                                                                                                     \
    }                                                                                                       \
    f32x4* wdst = reinterpret_cast<f32x4*>(smem + (B) * (PA_FLOATS + PB_FLOATS) + PA_FLOATS);               \
    if (!GB)                                                                                                \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) wdst[tid + WINO_THREADS * j] = wreg[j];                   \
  }
  // MFMA fragment offsets (floats, relative to the buffer base)
  const int m_tile = mt * 32 + li;
  const int a_off = (chalf * 8 * WTILES + m_tile) * PK + ((lh ^ ((m_tile >> 4) & 1)) << 2);
  const int b_off = PA_FLOATS + ((chalf * 8 * 2 + lh) * NB + nt * 32 + li) * 4;
  // GB: byte offset of this lane's quad inside a component's [h][64][4] weight block, two register sets of B fragments
  const int b_voff = (lh * NB + nt * 32 + li) * 16;
  f32x4 bA0 = {0.f, 0.f, 0.f, 0.f}, bA1 = bA0, bB0 = bA0, bB1 = bA0;
  const float4 abl_a = make_float4(1.f + tid * 1e-3f, 0.5f, 0.25f, 2.f);  // PIPE_ABL & 4096: constant A fragments
#define PIPE_BLOAD(S0, S1, C, CHUNK)                                                                        \
  if (GB && !(PIPE_ABL & 2048)) {                                                                                                 \
    const int so_ = ((cob * nst + (CHUNK)) * PB_FLOATS + (chalf * 8 + (C)) * 2 * NB * 4) * 4;               \
    S0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, b_voff, so_, 0));          \
    S1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, b_voff, so_ + 2 * NB * 16, 0)); \
  }

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
  // one V row (4 components) of (tile, quad): sR -> sA of buffer B
#define PIPE_TRANSFORM(B)                                                                                   \
  {                                                                                                         \
    f32x4 t[4];                                                                                             \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                         \
      const f32x4 u = *reinterpret_cast<const f32x4*>(sR + t_u[j]);                                         \
      const f32x4 w = *reinterpret_cast<const f32x4*>(sR + t_w[j]);                                         \
      t[j] = pk4_fma_s(t_sg2, w, u);                                                                        \
    }                                                                                                       \
    float* d_ = smem + (B) * (PA_FLOATS + PB_FLOATS) + t_dst;                                               \
    *reinterpret_cast<f32x4*>(d_ + 0 * WTILES * PK) = pk4_sub(t[0], t[2]);                                  \
    *reinterpret_cast<f32x4*>(d_ + 1 * WTILES * PK) = pk4_add(t[1], t[2]);                                  \
    *reinterpret_cast<f32x4*>(d_ + 2 * WTILES * PK) = pk4_sub(t[2], t[1]);                                  \
    *reinterpret_cast<f32x4*>(d_ + 3 * WTILES * PK) = pk4_sub(t[1], t[3]);                                  \
  }

  if (IN_MODE != 0) {
    if (a.lazy.mode == 0) {
      for (int c = tid; c < a.Cin; c += WINO_THREADS) {
        sS[c] = p_scale[c];
        sS[1024 + c] = p_shift[c];
      }
    } else {   // (BnLazy: the producer's statistics -> affine here; workgroup 0 stores for the later readers)
      for (int c = tid; c < a.Cin; c += WINO_THREADS) {
        float sc_, sh_;
        bn_lazy_affine(a.lazy, prob, c, sc_, sh_);
        sS[c] = sc_;
        sS[1024 + c] = sh_;
        if (blockIdx.x == 0) bn_lazy_store(a.lazy, c);
      }
    }
    __syncthreads();
  } else if (a.bnr_mode != 0) {
    // fused BatchNorm-backward sums (ConvArgs::bnr_*): the four per-channel parameters of this block's 64 output channels.
    // mode 1: {scale, shift, invstd, -mean * invstd} (xhat = y * invstd - mean * invstd); mode 2: {beta, 1 / gamma, -, -}
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
  // (only instantiated for IN_MODE 1: the producers of pooled layers read a BatchNorm + ReLU input themselves, and the
  // data-gradient variants, already at the register limit, stay free of the code)
  float* const p_pool = IN_MODE == 0 ? nullptr : (prob ? a.pool_out[1] : a.pool_out[0]);
  float* const sG = sS + PS_FLOATS;
  if (IN_MODE != 0 && p_pool != nullptr) {
    if (tid < NB) {
      const int co_ = cob * NB + tid;
      sG[tid] = (co_ < a.Cout && a.pool_gamma[co_] < 0.f) ? -1.f : 1.f;
    }
    __syncthreads();
  }
  // ---- prologue: stage 0 into buffer 0, loads of stage 1 in flight ----
  PIPE_ISSUE_LOADS()
  PIPE_WRITE_STAGE(0)
  __syncthreads();
  PIPE_TRANSFORM(0)
  PIPE_ISSUE_LOADS()
  PIPE_BLOAD(bA0, bA1, 0, 0)
  __syncthreads();

  // component (1,1) (accumulator 5 of the first component half) starts at the conv bias: it enters all four outputs
  // of a tile with coefficient +1, which saves the bias adds of the epilogue
  const float acc5_init = chalf == 0 ? bias_v : 0.f;
  f32x16 acc[8];
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = c == 5 ? acc5_init : 0.f;

  // components C and C + 1 of this wave's half: fragment reads, then 2 x 4 MFMAs on two alternating accumulators.
  // The staging work of the next stage is sliced BETWEEN the MFMA groups (fenced with sched_barrier so that the
  // compiler keeps the order): a wave that has just issued an MFMA group owns the issue slots of the ~250 cycles the
  // matrix pipe needs for it and for the group of the other wave of its SIMD.
#define PIPE_FRAG(C, S0, S1)                                                                                \
  const float4 a0_##C = (PIPE_ABL & 4096) ? abl_a : *reinterpret_cast<const float4*>(cA + a_off + (C) * WTILES * PK);       \
  const float4 a1_##C = (PIPE_ABL & 4096) ? abl_a : *reinterpret_cast<const float4*>(cA + a_off + ((C) + 1) * WTILES * PK); \
  const f32x4 b0_##C = GB ? S0 : *reinterpret_cast<const f32x4*>(cA + b_off + (C) * 2 * NB * 4);            \
  const f32x4 b1_##C = GB ? S1 : *reinterpret_cast<const f32x4*>(cA + b_off + ((C) + 1) * 2 * NB * 4);
#define PIPE_MFMA_LO(C)                                                                                     \
  if (!(PIPE_ABL & 8)) {                                                                                    \
  acc[C] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0_##C.x, b0_##C[0], acc[C], 0, 0, 0);                       \
  acc[(C) + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1_##C.x, b1_##C[0], acc[(C) + 1], 0, 0, 0);           \
  acc[C] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0_##C.y, b0_##C[1], acc[C], 0, 0, 0);                       \
  acc[(C) + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1_##C.y, b1_##C[1], acc[(C) + 1], 0, 0, 0); }
#define PIPE_MFMA_HI(C)                                                                                     \
  if (!(PIPE_ABL & 8)) {                                                                                    \
  acc[C] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0_##C.z, b0_##C[2], acc[C], 0, 0, 0);                       \
  acc[(C) + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1_##C.z, b1_##C[2], acc[(C) + 1], 0, 0, 0);           \
  acc[C] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0_##C.w, b0_##C[3], acc[C], 0, 0, 0);                       \
  acc[(C) + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1_##C.w, b1_##C[3], acc[(C) + 1], 0, 0, 0); }
#define PIPE_FENCE() __builtin_amdgcn_sched_barrier(0)

#if PIPE_ABL & 1024  // cycle stamps of every wave of block 8 -> the stats buffer (tools/archive/ablate_pipe.py trace)
  unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0;
  unsigned long long ets[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // epilogue, per round: start, transform done, barrier passed, stores issued
#define PIPE_TS(V) V = __builtin_amdgcn_s_memtime();
#define PIPE_ETS(I) ets[I] = __builtin_amdgcn_s_memtime();
#else
#define PIPE_TS(V)
#define PIPE_ETS(I)
#endif
  int tile = tile0, chunk = 0;
  for (int g = 0; g < nstages; ++g) {
    const int buf = g & 1;
    PIPE_TS(ts0)
    const float* const cA = smem + buf * (PA_FLOATS + PB_FLOATS);
    float* const nB = smem + (buf ^ 1) * (PA_FLOATS + PB_FLOATS);
    // ---- first half: components 0..3 of this wave's half || raw halo (stage g+1) -> sR, halo loads of stage g+2 ----
    {
      PIPE_FRAG(0, bA0, bA1)
      PIPE_BLOAD(bB0, bB1, 2, chunk)
      PIPE_FENCE();
      PIPE_MFMA_LO(0)
      PIPE_FENCE();
#pragma unroll
      for (int k = 0; k < 2; ++k) {  // raw halo of stage g+1 -> sR (BatchNorm + ReLU of the producer, zero padding)
        if (k == 0 || r1) {
          f32x4 v = hreg[k];
          if (IN_MODE != 0) v = bn_relu_quad(v, psc, psh, hoff[k] == OOB);
          *reinterpret_cast<f32x4*>(sR + r_lds[k]) = v;
        }
      }
      PIPE_FENCE();
      PIPE_MFMA_HI(0)
      PIPE_FENCE();
      PIPE_FRAG(2, bB0, bB1)
      PIPE_BLOAD(bA0, bA1, 4, chunk)
      PIPE_ISSUE_HALO()  // a full stage ahead of their use
      PIPE_FENCE();
      PIPE_MFMA_LO(2)
      PIPE_MFMA_HI(2)
    }
    // the fragments of components 4, 5 come from the SAME buffer: read them before the barrier so that the matrix pipe
    // restarts right after it
    PIPE_FRAG(4, bA0, bA1)
    PIPE_BLOAD(bB0, bB1, 6, chunk)
    PIPE_TS(ts1)
    __syncthreads();
    PIPE_TS(ts2)
    // ---- second half: components 4..7 || transform of stage g+1: sR -> sA, weights (g+1) -> sB of the other buffer,
    // weight loads of stage g+2 ----
    {
      const f32x4 u0 = *reinterpret_cast<const f32x4*>(sR + t_u[0]), w0 = *reinterpret_cast<const f32x4*>(sR + t_w[0]);
      const f32x4 u1 = *reinterpret_cast<const f32x4*>(sR + t_u[1]), w1 = *reinterpret_cast<const f32x4*>(sR + t_w[1]);
      const f32x4 u2 = *reinterpret_cast<const f32x4*>(sR + t_u[2]), w2 = *reinterpret_cast<const f32x4*>(sR + t_w[2]);
      const f32x4 u3 = *reinterpret_cast<const f32x4*>(sR + t_u[3]), w3 = *reinterpret_cast<const f32x4*>(sR + t_w[3]);
      PIPE_FENCE();
      PIPE_MFMA_LO(4)
      PIPE_FENCE();
      f32x4 t[4];
      t[0] = pk4_fma_s(t_sg2, w0, u0);
      t[1] = pk4_fma_s(t_sg2, w1, u1);
      t[2] = pk4_fma_s(t_sg2, w2, u2);
      t[3] = pk4_fma_s(t_sg2, w3, u3);
      float* d_ = nB + t_dst;
      *reinterpret_cast<f32x4*>(d_ + 0 * WTILES * PK) = pk4_sub(t[0], t[2]);
      *reinterpret_cast<f32x4*>(d_ + 1 * WTILES * PK) = pk4_add(t[1], t[2]);
      PIPE_FENCE();
      PIPE_MFMA_HI(4)
      PIPE_FENCE();
      PIPE_FRAG(6, bB0, bB1)
      PIPE_BLOAD(bA0, bA1, 0, (chunk + 1 == nst ? 0 : chunk + 1))  // first pair of the next stage
      *reinterpret_cast<f32x4*>(d_ + 2 * WTILES * PK) = pk4_sub(t[2], t[1]);
      *reinterpret_cast<f32x4*>(d_ + 3 * WTILES * PK) = pk4_sub(t[1], t[3]);
      PIPE_FENCE();
      PIPE_MFMA_LO(6)
      PIPE_FENCE();
      f32x4* wdst = reinterpret_cast<f32x4*>(nB + PA_FLOATS);  // weights of stage g+1 -> sB of the other buffer
      if (!GB)
#pragma unroll
      for (int j = 0; j < 4; ++j) wdst[tid + WINO_THREADS * j] = wreg[j];
      PIPE_ISSUE_W()
      PIPE_FENCE();
      PIPE_MFMA_HI(6)
    }
    PIPE_TS(ts3)
    if (!(PIPE_ABL & 4)) __syncthreads();
#if PIPE_ABL & 1024
    if (blockIdx.x == 8 && lane == 0 && g < 100 && p_stats != nullptr) {
      unsigned long long* q = reinterpret_cast<unsigned long long*>(p_stats) + (g * 8 + wave) * 4;
      q[0] = ts0; q[1] = ts1; q[2] = ts2; q[3] = ts3;
    }
#endif

    if (++chunk == nst) {
      if (PIPE_ABL & 1) {
        if (tid == 1023) p_out[0] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + acc[4][0] + acc[5][0] + acc[6][0] + acc[7][0];
      } else {
      // ---- tile epilogue: the consumed (sA, sB) pair of this stage is the 64 KB staging tile ----
      const int tx_i = tile % a.tiles_x, t2 = tile / a.tiles_x;
      const int ty0 = (t2 % a.tiles_y) * TH, tx0 = tx_i * TW, n = t2 / a.tiles_y;
      const bool full = (ty0 + TH <= a.H) && (tx0 + TW <= a.W);
      // Two rounds over the tile halves (bit 4 of the tile slot = accumulator registers 0..7 / 8..15): in each round
      // BOTH component halves write their partial outputs of 32 tiles to two 32 KB staging half-tiles, which meet in
      // the 16-byte store loop.  Compact tile index csl = (sl & 15) | (sl >> 5) << 4.
      float* const stg = smem + buf * (PA_FLOATS + PB_FLOATS) + chalf * (TH * TW * NB / 2);
      const int q16 = tid & 15;
      const int co4 = cob * NB + q16 * 4;
      const int nvalid = min(4, a.Cout - co4);
      mfma_results_guard();  // the output transform reads the accumulators from inline asm
#pragma unroll
      for (int rd = 0; rd < 2; ++rd) {
        // accumulator registers in pairs (r, r + 1) = Winograd tiles (ctx, ctx + 1): packed adds.  The conv bias is
        // already inside component (1,1) (accumulator 5 of the first half), which enters all four outputs with +1.
        PIPE_ETS(rd * 4 + 0)
        // fused BatchNorm-backward sums: the y / pooled-activation quads of this round's four store iterations are
        // requested BEFORE the output transform and the barrier (loading each one right where it is used exposed a full
        // HBM round trip per iteration: +1.1 us x 8 per tile, measured on a 240x320 data-gradient launch)
        constexpr int NSTORE = (TH * TW * 8) / WINO_THREADS;
        f32x4 tpre[NSTORE];
        if (IN_MODE == 0 && a.bnr_mode != 0) {
#pragma unroll
          for (int k = 0; k < NSTORE; ++k) {
            const int lp = (tid >> 4) + (WINO_THREADS / 16) * k;
            const int crow = lp / TW, ccol = lp - crow * TW;
            const int csl = (crow >> 1) * TTX + (ccol >> 1);
            const int sl = (csl & 15) | (rd << 4) | ((csl >> 4) << 5);
            const int oy = ty0 + 2 * (sl / TTX) + (crow & 1), ox = tx0 + ccol;
            const bool ok = nvalid > 0 && (full || (oy < a.H && ox < a.W));
            const float* tp = p_bnr + ((size_t)(n * a.H + (ok ? oy : 0)) * a.W + (ok ? ox : 0)) * a.bnr_cs + a.bnr_co + (nvalid > 0 ? co4 : 0);
            tpre[k] = *reinterpret_cast<const f32x4*>(tp);  // clamped address: always valid, unused when !ok
          }
        }
        if (chalf == 0) pipe_out_rows<0, TTX, TW>(acc, rd, lh, mt, stg + nt * 32 + li);
        else pipe_out_rows<1, TTX, TW>(acc, rd, lh, mt, stg + nt * 32 + li);
        PIPE_ETS(rd * 4 + 1)
        __syncthreads();
        PIPE_ETS(rd * 4 + 2)
        const float* const s0p = smem + buf * (PA_FLOATS + PB_FLOATS);
        if ((WIDE ? full : true) && (cob + 1) * NB <= a.Cout) {
          // whole channel block inside the tensor (block-uniform: every shipped layer; 8x32 tiles: full tiles only, their
          // partial tiles are rare and the predication costs the full ones 1 %): all eight LDS reads of the round
          // first, then sums / statistics / four (predicated) 16-byte stores.  The general loop below compiles to four
          // branchy blocks that each wait for their own LDS round trip (2700 of the ~9500 cycles of a tile epilogue were
          // spent there, ~850 now: profiles/r02_conv_cycle_trace.txt).
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
            const bool fullk = WIDE ? true : full;
            const bool ok = fullk || (oy < a.H && ox < a.W);
            f32x4 v = pk4_add(va[k], vb[k]);
            if (!fullk) {  // block-uniform; pixels of a partial tile beyond the image: no statistics, no store
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = ok ? v[e] : 0.f;
            }
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
            if (!(PIPE_ABL & 16) && ok)
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
            const f32x4 v = pk4_add(*reinterpret_cast<const f32x4*>(s0p + lp * NB + q16 * 4),
                                    *reinterpret_cast<const f32x4*>(s0p + TH * TW * NB / 2 + lp * NB + q16 * 4));
            if (IN_MODE == 0 && a.bnr_mode != 0) {
              // BatchNorm-backward sums of the layer below (see ConvArgs::bnr_mode): v is its activation gradient
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
            if (PIPE_ABL & 16) continue;
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
          // pooled raw output (ConvArgs::pool_out): a Winograd tile IS a pooling window; thread = (window of this round,
          // channel quad) reads the four summed pixels back from the staging tile
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
        PIPE_ETS(rd * 4 + 3)
        __syncthreads();  // round 1 / the next-but-one stage overwrite the staging half-tiles
      }
      }
#if PIPE_ABL & 1024
      if (blockIdx.x == 8 && lane == 0 && g < 100 && p_stats != nullptr) {
        unsigned long long* q = reinterpret_cast<unsigned long long*>(p_stats) + 3200 + ((g / nst) * 8 + wave) * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) q[i] = ets[i];
      }
#endif
#pragma unroll
      for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = c == 5 ? acc5_init : 0.f;
      chunk = 0;
      tile += per_cob;
    }
  }
#undef PIPE_ISSUE_LOADS
#undef PIPE_ISSUE_HALO
#undef PIPE_ISSUE_W
#undef PIPE_WRITE_STAGE
#undef PIPE_TRANSFORM
#undef PIPE_FRAG
#undef PIPE_BLOAD
#undef PIPE_MFMA_LO
#undef PIPE_MFMA_HI
#undef PIPE_FENCE

  if (p_stats != nullptr && !(PIPE_ABL & 1024)) {
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

// a x + b y + c z with ONE rounding pattern everywhere (hipcc otherwise picks, per call site, which of the three products stays
// unfused - the per-element and the per-cell packing kernels would produce images that differ in the last bit)
__device__ __forceinline__ float wino_dot3(float a, float x, float b, float y, float c, float z) {
  return __builtin_fmaf(c, z, __builtin_fmaf(b, y, __fmul_rn(a, x)));
}

// OIHW 3x3 weights -> U = G g G^T in the LDS image of conv_wino_pipe_kernel: [cob][chunk8][component][h][64][4]
__device__ __forceinline__ void pack_wino8_element(const float* __restrict__ w, float* __restrict__ dst, int Cout_w,
                                                   int Cin_w, int transpose_flip, int nchunks_total, int chunk_off,
                                                   int cob_off, int ncob, int nchunks, int idx) {
  const int per_chunk = PB_FLOATS;
  const int total = ncob * nchunks * per_chunk;
  if (idx >= total) return;
  int t = idx;
  const int e = t & 3;
  t >>= 2;
  const int nn = t & 63;
  t >>= 6;
  const int h = t & 1;
  t >>= 1;
  const int comp = t % WC;
  t /= WC;
  const int chunk = t % nchunks;
  const int cob = t / nchunks;
  const int co = cob * NB + nn;
  const int ci = chunk * PK + h * 4 + e;
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
  dst[((size_t)(cob + cob_off) * nchunks_total + chunk + chunk_off) * per_chunk + ((comp * 2 + h) * NB + nn) * 4 + e] = u;
}

// OIHW 3x3 weights -> U = G g G^T of Winograd F(4x4,3x3) (G: 6x3) in the fragment image of conv_wino4_kernel:
// [cob][chunk8][component 36][h][64][4]
constexpr int W4_PACK_FLOATS = 36 * PK * NB;
__device__ __forceinline__ void pack_wino4_element(const float* __restrict__ w, float* __restrict__ dst, int Cout_w,
                                                   int Cin_w, int transpose_flip, int nchunks_total, int chunk_off,
                                                   int cob_off, int ncob, int nchunks, int idx) {
  const int per_chunk = W4_PACK_FLOATS;
  const int total = ncob * nchunks * per_chunk;
  if (idx >= total) return;
  int t = idx;
  const int e = t & 3;
  t >>= 2;
  const int nn = t & 63;
  t >>= 6;
  const int h = t & 1;
  t >>= 1;
  const int comp = t % 36;
  t /= 36;
  const int chunk = t % nchunks;
  const int cob = t / nchunks;
  const int co = cob * NB + nn;
  const int ci = chunk * PK + h * 4 + e;
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
  const int i = comp / 6, j = comp % 6;
  // rows of G: {1/4, 0, 0}, {-1/6, -1/6, -1/6}, {-1/6, 1/6, -1/6}, {1/24, 1/12, 1/6}, {1/24, -1/12, 1/6}, {0, 0, 1}
  const float g0i = i == 0 ? 0.25f : i == 1 || i == 2 ? -1.f / 6.f : i == 5 ? 0.f : 1.f / 24.f;
  const float g1i = i == 0 || i == 5 ? 0.f : i == 1 ? -1.f / 6.f : i == 2 ? 1.f / 6.f : i == 3 ? 1.f / 12.f : -1.f / 12.f;
  const float g2i = i == 0 ? 0.f : i == 1 || i == 2 ? -1.f / 6.f : i == 5 ? 1.f : 1.f / 6.f;
  const float g0j = j == 0 ? 0.25f : j == 1 || j == 2 ? -1.f / 6.f : j == 5 ? 0.f : 1.f / 24.f;
  const float g1j = j == 0 || j == 5 ? 0.f : j == 1 ? -1.f / 6.f : j == 2 ? 1.f / 6.f : j == 3 ? 1.f / 12.f : -1.f / 12.f;
  const float g2j = j == 0 ? 0.f : j == 1 || j == 2 ? -1.f / 6.f : j == 5 ? 1.f : 1.f / 6.f;
  float r[3];
#pragma unroll
  for (int x = 0; x < 3; ++x) r[x] = wino_dot3(g0i, k[0][x], g1i, k[1][x], g2i, k[2][x]);
  const float u = wino_dot3(g0j, r[0], g1j, r[1], g2j, r[2]);
  dst[((size_t)(cob + cob_off) * nchunks_total + chunk + chunk_off) * per_chunk + ((comp * 2 + h) * NB + nn) * 4 + e] = u;
}

__global__ void pack_weights_wino4_kernel(const float* __restrict__ w, float* __restrict__ dst, int Cout_w, int Cin_w,
                                          int transpose_flip, int nchunks_total, int chunk_off, int cob_off, int ncob,
                                          int nchunks) {
  pack_wino4_element(w, dst, Cout_w, Cin_w, transpose_flip, nchunks_total, chunk_off, cob_off, ncob, nchunks,
                     blockIdx.x * blockDim.x + threadIdx.x);
}

__global__ void pack_weights_wino8_kernel(const float* __restrict__ w, float* __restrict__ dst, int Cout_w, int Cin_w,
                                          int transpose_flip, int nchunks_total, int chunk_off, int cob_off, int ncob,
                                          int nchunks) {
  pack_wino8_element(w, dst, Cout_w, Cin_w, transpose_flip, nchunks_total, chunk_off, cob_off, ncob, nchunks,
                     blockIdx.x * blockDim.x + threadIdx.x);
}

// All 3x3 layers of the network (forward and data-gradient images) in ONE launch: the weights change at every optimizer
// step, so they are re-packed per forward; 20 launches of ~5 us each were 0.7 % of the pair step.
struct PackJob {
  const float* w;
  float* dst;
  int cout_w, cin_w, tf, nchunks_total, chunk_off, cob_off, ncob, nchunks;
  int block0;  // first block of this job (256 elements per block)
  int w4;      // F(4x4,3x3) image (pack_wino4_element) instead of the F(2x2,3x3) one
};
constexpr int PACK_MAX_JOBS = 32;
struct PackJobs {
  int n;
  PackJob j[PACK_MAX_JOBS];
};
// One thread per (cob, chunk, h, output channel, input-channel quad element) CELL: the 3x3 filter is read once and all 36 / 16
// components are produced from registers with the expressions of pack_wino4_element / pack_wino8_element.  The per-element form read each filter 36 / 16 times with a stride of 9 floats between lanes: ~95 us for the 2.6 M
// floats of a step's images, latency-bound, 0.29 ms beside the first-layer convolution it is meant to hide under.
template <bool W4>
__device__ __forceinline__ void pack_wino_cell(const PackJob& q, int cell) {
  constexpr int NC = W4 ? 36 : WC;
  constexpr int per_chunk = NC * PK * NB;
  const int cells = q.ncob * q.nchunks * (PK * NB);
  if (cell >= cells) return;
  int t = cell;
  const int e = t & 3;
  t >>= 2;
  const int nn = t & 63;
  t >>= 6;
  const int h = t & 1;
  t >>= 1;
  const int chunk = t % q.nchunks;
  const int cob = t / q.nchunks;
  const int co = cob * NB + nn;
  const int ci = chunk * PK + h * 4 + e;
  float k[3][3];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      float v = 0.f;
      if (!q.tf) {
        if (co < q.cout_w && ci < q.cin_w) v = q.w[(((size_t)co * q.cin_w + ci) * 3 + ky) * 3 + kx];
      } else {
        if (co < q.cin_w && ci < q.cout_w) v = q.w[(((size_t)ci * q.cin_w + co) * 3 + (2 - ky)) * 3 + (2 - kx)];
      }
      k[ky][kx] = v;
    }
  float* const d = q.dst + ((size_t)(cob + q.cob_off) * q.nchunks_total + chunk + q.chunk_off) * per_chunk + (h * NB + nn) * 4 + e;
#pragma unroll
  for (int comp = 0; comp < NC; ++comp) {
    float u;
    if (W4) {
      const int i = comp / 6, j = comp % 6;
      const float g0i = i == 0 ? 0.25f : i == 1 || i == 2 ? -1.f / 6.f : i == 5 ? 0.f : 1.f / 24.f;
      const float g1i = i == 0 || i == 5 ? 0.f : i == 1 ? -1.f / 6.f : i == 2 ? 1.f / 6.f : i == 3 ? 1.f / 12.f : -1.f / 12.f;
      const float g2i = i == 0 ? 0.f : i == 1 || i == 2 ? -1.f / 6.f : i == 5 ? 1.f : 1.f / 6.f;
      const float g0j = j == 0 ? 0.25f : j == 1 || j == 2 ? -1.f / 6.f : j == 5 ? 0.f : 1.f / 24.f;
      const float g1j = j == 0 || j == 5 ? 0.f : j == 1 ? -1.f / 6.f : j == 2 ? 1.f / 6.f : j == 3 ? 1.f / 12.f : -1.f / 12.f;
      const float g2j = j == 0 ? 0.f : j == 1 || j == 2 ? -1.f / 6.f : j == 5 ? 1.f : 1.f / 6.f;
      float r[3];
#pragma unroll
      for (int x = 0; x < 3; ++x) r[x] = wino_dot3(g0i, k[0][x], g1i, k[1][x], g2i, k[2][x]);
      u = wino_dot3(g0j, r[0], g1j, r[1], g2j, r[2]);
    } else {
      const int i = comp >> 2, j = comp & 3;
      float r[3];
#pragma unroll
      for (int x = 0; x < 3; ++x)
        r[x] = i == 0 ? k[0][x] : i == 1 ? 0.5f * (k[0][x] + k[1][x] + k[2][x]) : i == 2 ? 0.5f * (k[0][x] - k[1][x] + k[2][x]) : k[2][x];
      u = j == 0 ? r[0] : j == 1 ? 0.5f * (r[0] + r[1] + r[2]) : j == 2 ? 0.5f * (r[0] - r[1] + r[2]) : r[2];
    }
    d[(size_t)comp * 2 * NB * 4] = u;
  }
}
// block0 counts blocks of 256 CELLS (pack_all: cdiv(ncob * nchunks * PK * NB, 256) per job)
__global__ __launch_bounds__(256) void pack_weights_wino8_multi_kernel(const PackJobs J) {
  int k = 0;
  while (k + 1 < J.n && (int)blockIdx.x >= J.j[k + 1].block0) ++k;  // wave-uniform linear search, <= 32 jobs
  const PackJob& q = J.j[k];
  const int cell = ((int)blockIdx.x - q.block0) * blockDim.x + threadIdx.x;
  if (q.w4) pack_wino_cell<true>(q, cell);
  else pack_wino_cell<false>(q, cell);
}

}  // namespace sspk
