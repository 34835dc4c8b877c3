// Winograd F(4x4, 3x3) convolution for the large maps (240x320, 120x160): 36 transform components per 6x6 input patch,
// 4x4 outputs per tile = 2.25 matrix multiplies per output instead of the 4 of F(2x2,3x3) (conv_wino_pipe.hip.h), with
// about the same transform work per output.  fp32 error against fp64 on a 64 -> 64 layer's data: rel-L2 1.5e-6 (F(2x2):
// 2.4e-7; tools/wino_error_probe.py).
//
//   workgroup = 8 waves = 32 tiles (32x16 or 16x32 output pixels) x 64 output channels, one workgroup per CU;
//   wave = (I, J, nt): the 3x3 quadrant (rows 3I.., columns 3J..) of the 6x6 component matrix M, output-channel half nt:
//          9 accumulators of 32 channels x 32 tiles = 144 registers.
//
// Operand roles as in conv_wino_p2_kernel: A = weight fragment (32 output channels x 2 k), B = transformed input fragment
// (2 k x 32 tiles); an accumulator lane holds ONE tile (column = lane & 31) and 16 output channels (row = (reg & 3) + 8 (reg
// >> 2) + 4 (lane >> 5)), so four consecutive channels sit in four consecutive registers (16-byte LDS / global accesses).
//
// Stage pipeline (8 input channels per stage = 36 MFMAs per wave; transformed input double-buffered, ONE raw-halo buffer):
//
//   first half : components 0..4  ||  transform raw(g+1): sR -> sA[~g&1]   (waves 0..5 = rows 0..5 of V = B^T d B)
//   barrier A
//   second half: components 5..8  ||  halo (g+2): registers -> sR (BatchNorm + ReLU of the producer), halo loads (g+3)
//   barrier B
//
// Every MFMA operand is fetched two components (8 MFMAs) ahead into one of three rotating register sets (9 = 3 x 3
// components per stage, so the rotation is the same in every stage): weights from L2 (the packed image
// [cob][chunk8][component][h][64][4] IS the fragment layout), inputs from LDS.
//
// Epilogue Y = A^T M A (4x6 . 6x6 . 6x4) over four waves per channel half, two exchange steps per register quad:
//   step 1: Q = M[I,J] A[J,:]  (3 x 4);  the waves (I,0) and (I,1) swap the two output columns they do not keep -> R (3 x 2)
//   step 2: P = A^T[:,I] R      (4 x 2);  the waves (0,J) and (1,J) swap the two output rows they do not keep   -> Y (2 x 2)
// so every wave ends up with the 2x2 pixel block (2I, 2J) of its lane's tile - one pooling window - for 16 channels in
// registers: 16-byte global stores straight from registers, BatchNorm sums by shuffles, pooled raw output in registers.
// 10 instead of 16 exchanged values per (tile, channel) and wave; the step-1 buffer is the just-consumed sA image.
//
// Raw halo in LDS: [row][pixel][8 channels], row pitch 40 / 24 pixels (a multiple of the 256-byte bank row); the pixels
// of halo row R are rotated by (R >> 2) & 3 inside aligned groups of 8: the transform's ds_read_b128 (lane = (quad, tile),
// a 16-lane group = 8 tiles with distinct (tx & 1, ty & 3)) then hits 8 distinct 32-byte slots.
#pragma once
#include "conv_wino_p2.hip.h"

#ifndef W4_ABL
#define W4_ABL 0  // compile-time perf ablation: 1 no tile epilogue
#endif

namespace sspk {

constexpr int W4C = 36;                                // Winograd components
constexpr int W4_THREADS = 512;
constexpr int W4_TILES = 32;                           // 4x4-pixel tiles per workgroup
constexpr int W4_A_FLOATS = W4C * W4_TILES * PK;       // 9216 floats = 36 KB transformed input per buffer
constexpr int W4_B_FLOATS = W4C * PK * NB;             // 18432 floats: packed weights per (cob, 8-channel chunk)
constexpr int W4_R_FLOATS = 34 * 24 * PK;              // raw halo: 34 rows x 24 pixels (16x32 tiles) >= 18 x 40 (32x16)
constexpr int W4_XA_FLOATS = 2 * 6 * 256;              // step-1 exchange of waves 6, 7 (waves 0..5: the consumed sA image)
constexpr int W4_X2_FLOATS = 8 * 4 * 256;              // step-2 exchange: [wave][value][lane][4]
constexpr int W4_S_FLOATS = 2048 + 2 * NB;             // BatchNorm scale | shift (or 4 x 64 bnr parameters), bias, pool sign
constexpr int W4_LDS_BYTES = (2 * W4_A_FLOATS + W4_R_FLOATS + W4_XA_FLOATS + W4_X2_FLOATS + W4_S_FLOATS) * 4;  // 153600
static_assert(6 * 6 * 256 == W4_A_FLOATS, "six waves' step-1 slots fill one transformed-input buffer");
static_assert(W4_LDS_BYTES <= 160 * 1024, "one workgroup per CU");
static_assert(W4_R_FLOATS * 4 < 65536, "16-bit raw-halo byte addresses");

// k (wave-uniform, in a scalar register pair) * y + z
__device__ __forceinline__ f32x2 pk_fma_k(f32x2 k, f32x2 y, f32x2 z) {
  f32x2 d;
  asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "s"(k), "v"(y), "v"(z));
  return d;
}
__device__ __forceinline__ f32x2 pk_mul_k(f32x2 k, f32x2 y) {
  f32x2 d;
  asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "s"(k), "v"(y));
  return d;
}
// inline-constant multiplier (4.0, -4.0, 2.0, -2.0), broadcast to both halves: no register for the constant
#define W4_PK_IMM(NAME, IMM)                                                                                 \
  __device__ __forceinline__ f32x2 NAME(f32x2 y, f32x2 z) {                                                  \
    f32x2 d;                                                                                                 \
    asm volatile("v_pk_fma_f32 %0, " IMM ", %1, %2 op_sel_hi:[0,1,1]" : "=v"(d) : "v"(y), "v"(z));          \
    return d;                                                                                                \
  }                                                                                                          \
  __device__ __forceinline__ f32x4 NAME##4(f32x4 y, f32x4 z) { return cat2(NAME(lo2(y), lo2(z)), NAME(hi2(y), hi2(z))); }
W4_PK_IMM(pk_fma_p4, "4.0")
W4_PK_IMM(pk_fma_m4, "-4.0")
W4_PK_IMM(pk_fma_p2, "2.0")
W4_PK_IMM(pk_fma_m2, "-2.0")
#undef W4_PK_IMM
__device__ __forceinline__ f32x4 pk4_fma_k(f32x2 k, f32x4 y, f32x4 z) { return cat2(pk_fma_k(k, lo2(y), lo2(z)), pk_fma_k(k, hi2(y), hi2(z))); }
__device__ __forceinline__ f32x4 pk4_mul_k(f32x2 k, f32x4 y) { return cat2(pk_mul_k(k, lo2(y)), pk_mul_k(k, hi2(y))); }

// tile index (= MFMA column = lane & 31) -> tile coordinates inside the workgroup's 8x4 / 4x8 tile block; tiles with the
// same index mod 8 differ in (tx & 1, ty & 3)
template <bool WIDE>
__device__ __forceinline__ void w4_tile_xy(int t, int& ty, int& tx) {
  if (WIDE) { tx = 2 * (t >> 3) + ((t >> 2) & 1); ty = t & 3; }
  else { tx = 2 * ((t >> 3) & 1) + ((t >> 2) & 1); ty = 4 * (t >> 4) + (t & 3); }
}

template <int IN_MODE, bool WIDE>
__global__ __launch_bounds__(W4_THREADS) void conv_wino4_kernel(const ConvArgs a) {
  constexpr int TTX = WIDE ? 8 : 4, TTY = WIDE ? 4 : 8;
  constexpr int TH = 4 * TTY, TW = 4 * TTX;
  constexpr int HR = TH + 2, HC = TW + 2;
  constexpr int PITCH = (HC + 7) / 8 * 8;              // pixels per LDS halo row: whole rotation groups
  constexpr int NHALO = HR * HC;                       // 612
  constexpr int ROWF = PITCH * PK;                     // floats per LDS halo row
  static_assert(HR * PITCH * PK <= W4_R_FLOATS, "raw halo buffer");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // [sR][sA0][sA1][sXa][sX2][sS]: the raw halo sits at LDS offset 0, so that two of its byte addresses fit one register
  float* const sR = smem;
  float* const sA = smem + W4_R_FLOATS;
  float* const sXa = sA + 2 * W4_A_FLOATS;
  float* const sX2 = sXa + W4_XA_FLOATS;
  float* const sS = sX2 + W4_X2_FLOATS;                // IN_MODE 1: scale[Cin] | shift[Cin]; bnr: 4 x 64 parameters
  float* const sBias = sS + 2048;
  float* const sG = sBias + NB;                        // +-1: sign of gamma (pooled raw output)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int nt = wave & 1, qJ = (wave >> 1) & 1, qI = wave >> 2;

  // ---- work assignment (as conv_wino_pipe_kernel): XCD-aware persistent tile list ----
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
  const float* const p_bnr = prob ? a.bnr_t2 : a.bnr_t;
  float* const p_pool = IN_MODE == 0 ? nullptr : (prob ? a.pool_out[1] : a.pool_out[0]);
  const int nst = a.Cin / PK;
  const int my_tiles = (t_end - tile0 + per_cob - 1) / per_cob;
  const int nstages = my_tiles * nst;

  // ---- staging roles: raw halo items tid + 512 k (k < 3), item = pixel * 2 + quad ----
  // (LDS addresses of the staging and transform roles are recomputed from t_key after every tile epilogue, W4_ADDR_SETUP:
  // the epilogue needs their registers)
  const int q2 = tid & 1;
  int t_key = tid;
  int r_lds[3];
  const bool r2 = tid + 2 * W4_THREADS < NHALO * 2;  // the third item exists
  const int pixb = a.in_cs * 4, rowb = a.W * pixb;
  f32x4 hreg[3];
  int h_chunk = 0;  // 8-channel chunk of the halo loads in hreg (BatchNorm parameters of the producer at W4_HALO_BN)
  constexpr unsigned OOB = 0x80000000u;
  unsigned hoff[3] = {OOB, OOB, OOB};
  const size_t img_floats = (size_t)a.H * a.W * a.in_cs;
  __amdgpu_buffer_rsrc_t rsrc_in;
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wpk), 0, a.wpk_bytes, 0x00020000);

  int ld_tile = tile0, ld_chunk = 0;
#define W4_ISSUE_HALO()                                                                                     \
  {                                                                                                         \
    if (ld_chunk == 0) {                                                                                    \
      const int tt_ = min(ld_tile, t_end - 1);  /* past the end: harmless redundant loads of the last tile */ \
      const int tx_ = tt_ % a.tiles_x, t2_ = tt_ / a.tiles_x;                                               \
      const int ty0_ = (t2_ % a.tiles_y) * TH, tx0_ = tx_ * TW, n_ = t2_ / a.tiles_y;                       \
      _Pragma("unroll") for (int k = 0; k < 3; ++k) {                                                       \
        const int p_ = (tid + W4_THREADS * k) >> 1, r_ = p_ / HC, c_ = p_ - r_ * HC;                        \
        const int gy = ty0_ - 1 + r_, gx = tx0_ - 1 + c_;                                                   \
        const bool ok = (k < 2 || r2) && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;      \
        hoff[k] = ok ? (unsigned)(gy * rowb + gx * pixb + (a.in_co + q2 * 4) * 4) : OOB;                    \
      }                                                                                                     \
      rsrc_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p_in) + (size_t)n_ * img_floats, 0,    \
                                                  a.in_bytes, 0x00020000);                                  \
    }                                                                                                       \
    h_chunk = ld_chunk;                                                                                     \
    _Pragma("unroll") for (int k = 0; k < 3; ++k)                                                           \
      hreg[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_in, hoff[k], ld_chunk * PK * 4, 0)); \
    if (++ld_chunk == nst) { ld_chunk = 0; ld_tile += per_cob; }                                            \
  }
  f32x4 psc, psh;
#define W4_HALO_PAR()                                                                                       \
  if (IN_MODE != 0) {                                                                                       \
    psc = *reinterpret_cast<const f32x4*>(sS + h_chunk * PK + q2 * 4);                                      \
    psh = *reinterpret_cast<const f32x4*>(sS + 1024 + h_chunk * PK + q2 * 4);                               \
  }
#define W4_HALO_BN(K) if (IN_MODE != 0 && !(W4_ABL & 4)) hreg[K] = bn_relu_quad(hreg[K], psc, psh, hoff[K] == OOB);
#define W4_HALO_WR(K) if (!(W4_ABL & 4) && ((K) < 2 || r2)) *reinterpret_cast<f32x4*>(sR + r_lds[K]) = hreg[K];

  // ---- transform roles: wave i < 6 computes row i of V = B^T d B for (tile, channel quad) = (tid >> 1) & 31, tid & 1 ----
  //   T[c] = k0 d[r0][c] + k1 d[r0+1][c] + k2 d[r0+2][c] + d[rl][c]   (rows of B^T; rows 0 and 5 have k1 = 0)
  const bool tw = (W4_ABL & 2) ? false : wave < 6;
  const int t_r0 = wave == 0 ? 0 : 1, t_rl = wave == 5 ? 5 : 4;
  const float k0f = (wave == 1) ? -4.f : (wave == 3) ? -2.f : (wave == 4) ? 2.f : 4.f;
  const float k1f = (wave == 0 || wave == 5) ? 0.f : (wave == 1 || wave == 2) ? -4.f : -1.f;
  const float k2f = (wave == 0 || wave == 5) ? -5.f : (wave == 1) ? 1.f : (wave == 2) ? -1.f : (wave == 3) ? 2.f : -2.f;
  // (readfirstlane: the coefficients must reach the packed fmas in scalar register pairs, not in per-lane selects)
#define W4_UNI(X) __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, (X))))
  const float k0u = W4_UNI(k0f), k1u = W4_UNI(k1f), k2u = W4_UNI(k2f);
#undef W4_UNI
  const f32x2 tk0 = {k0u, k0u}, tk1 = {k1u, k1u}, tk2 = {k2u, k2u};
  int t_ab[6], t_dst;  // byte addresses of the first row read | of the last row read << 16
#define W4_ADDR_SETUP()                                                                                     \
  {                                                                                                         \
    asm volatile("" : "+v"(t_key));  /* opaque: the addresses below are recomputed, not kept live */        \
    _Pragma("unroll") for (int k = 0; k < 3; ++k) {                                                         \
      const int p_ = (t_key + W4_THREADS * k) >> 1, r_ = p_ / HC, c_ = p_ - r_ * HC;                        \
      r_lds[k] = (r_ * PITCH + (c_ & ~7) + ((c_ + ((r_ >> 2) & 3)) & 7)) * PK + (t_key & 1) * 4;            \
    }                                                                                                       \
    const int tt_ = (t_key >> 1) & 31;                                                                      \
    int ty_, tx_;                                                                                           \
    w4_tile_xy<WIDE>(tt_, ty_, tx_);                                                                        \
    _Pragma("unroll") for (int c = 0; c < 6; ++c) {                                                         \
      const int C = 4 * tx_ + c, Ra = 4 * ty_ + t_r0, Rb = 4 * ty_ + t_rl;                                  \
      const int ta_ = (Ra * PITCH + (C & ~7) + ((C + ((Ra >> 2) & 3)) & 7)) * PK + (t_key & 1) * 4;         \
      const int tb_ = (Rb * PITCH + (C & ~7) + ((C + ((Rb >> 2) & 3)) & 7)) * PK + (t_key & 1) * 4;         \
      t_ab[c] = (ta_ * 4) | ((tb_ * 4) << 16);                                                              \
    }                                                                                                       \
    t_dst = (wave * 6 * W4_TILES + tt_) * PK + (((t_key & 1) ^ ((tt_ >> 3) & 1)) << 2);                     \
  }
  W4_ADDR_SETUP()
#define W4_LD(P) (*reinterpret_cast<const f32x4*>(P))
#define W4_TR_RD(C)                                                                                         \
  if (tw) {                                                                                                 \
    const char* pa_ = reinterpret_cast<const char*>(smem) + (t_ab[C] & 0xffff);                            \
    raw0 = W4_LD(pa_); raw1 = W4_LD(pa_ + ROWF * 4); raw2 = W4_LD(pa_ + 2 * ROWF * 4);                      \
    raw3 = W4_LD(reinterpret_cast<const char*>(smem) + ((unsigned)t_ab[C] >> 16));                          \
  }
#define W4_TR_T(DST) if (tw) DST = pk4_fma_k(tk2, raw2, pk4_fma_k(tk1, raw1, pk4_fma_k(tk0, raw0, raw3)));
#define W4_TR_WR(DSTBUF, J, V) *reinterpret_cast<f32x4*>((DSTBUF) + t_dst + (J) * W4_TILES * PK) = (V);
  // whole transform of one stage, unsliced (prologue)
#define W4_TRANSFORM(DSTBUF)                                                                                \
  if (tw) {                                                                                                 \
    f32x4 T_[6];                                                                                            \
    _Pragma("unroll") for (int c = 0; c < 6; ++c) {                                                         \
      const char* pa_ = reinterpret_cast<const char*>(smem) + (t_ab[c] & 0xffff);                          \
      const f32x4 d0 = W4_LD(pa_), d1 = W4_LD(pa_ + ROWF * 4), d2 = W4_LD(pa_ + 2 * ROWF * 4);              \
      const f32x4 d3 = W4_LD(reinterpret_cast<const char*>(smem) + ((unsigned)t_ab[c] >> 16));              \
      T_[c] = pk4_fma_k(tk2, d2, pk4_fma_k(tk1, d1, pk4_fma_k(tk0, d0, d3)));                               \
    }                                                                                                       \
    const f32x4 ta_ = pk_fma_m44(T_[2], T_[4]), tb_ = pk_fma_m44(T_[1], T_[3]);                             \
    const f32x4 tc_ = pk4_sub(T_[4], T_[2]), te_ = pk4_sub(T_[3], T_[1]);                                   \
    W4_TR_WR(DSTBUF, 0, pk_fma_p44(pk4_sub(T_[0], T_[2]), tc_))       /* 4 T0 - 5 T2 + T4 */                \
    W4_TR_WR(DSTBUF, 1, pk4_add(ta_, tb_))                                                                  \
    W4_TR_WR(DSTBUF, 2, pk4_sub(ta_, tb_))                                                                  \
    W4_TR_WR(DSTBUF, 3, pk_fma_p24(te_, tc_))                                                               \
    W4_TR_WR(DSTBUF, 4, pk_fma_m24(te_, tc_))                                                               \
    W4_TR_WR(DSTBUF, 5, pk_fma_m44(te_, pk4_sub(T_[5], T_[3])))       /* 4 T1 - 5 T3 + T5 */                \
  }

  // ---- per-block parameters in LDS ----
  if (tid < NB) {
    const int co_ = cob * NB + tid;
    sBias[tid] = (a.bias != nullptr && co_ < a.Cout) ? a.bias[co_] : 0.f;
    if (IN_MODE != 0) sG[tid] = (p_pool != nullptr && co_ < a.Cout && a.pool_gamma[co_] < 0.f) ? -1.f : 1.f;
  }
  if (IN_MODE != 0) {
    for (int c = tid; c < a.Cin; c += W4_THREADS) {
      sS[c] = p_scale[c];
      sS[1024 + c] = p_shift[c];
    }
  } else if (a.bnr_mode != 0) {
    // fused BatchNorm-backward sums (ConvArgs::bnr_*): the four per-channel parameters of this block's 64 output channels.
    // mode 1: {scale, shift, invstd, -mean * invstd} (xhat = y * invstd - mean * invstd); mode 2: {beta, 1 / gamma, -, -}
    if (tid < NB) {
      const int co_ = cob * NB + tid;
      float q0 = 0.f, q1 = 0.f, q2_ = 0.f, q3 = 0.f;
      if (co_ < a.Cout) {
        if (a.bnr_mode == 1) {
          const float is_ = a.bnr_p3[prob][co_];
          q0 = a.bnr_p0[prob][co_]; q1 = a.bnr_p1[prob][co_]; q2_ = is_; q3 = -a.bnr_p2[prob][co_] * is_;
        } else {
          const float g_ = a.bnr_p1[prob][co_];
          q0 = a.bnr_p0[prob][co_]; q1 = g_ != 0.f ? 1.f / g_ : 0.f;
        }
      }
      sS[tid] = q0; sS[NB + tid] = q1; sS[2 * NB + tid] = q2_; sS[3 * NB + tid] = q3;
    }
  }
  __syncthreads();

  // ---- prologue: sA[0] = transformed stage 0, sR = raw halo of stage 1, halo loads of stage 2 in flight ----
  W4_ISSUE_HALO()
  W4_HALO_PAR() W4_HALO_BN(0) W4_HALO_BN(1) W4_HALO_BN(2)
  W4_HALO_WR(0) W4_HALO_WR(1) W4_HALO_WR(2)
  __syncthreads();
  W4_TRANSFORM(sA)
  W4_ISSUE_HALO()
  __syncthreads();
  W4_HALO_PAR() W4_HALO_BN(0) W4_HALO_BN(1) W4_HALO_BN(2)
  W4_HALO_WR(0) W4_HALO_WR(1) W4_HALO_WR(2)
  W4_ISSUE_HALO()
  __syncthreads();

  f32x16 acc[9];
#pragma unroll
  for (int c = 0; c < 9; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  float stat_acc[4] = {0.f, 0.f, 0.f, 0.f};  // per register quad: lane li holds value (li >> 2) & 7 (sums 0..3, weighted sums 4..7)

  // ---- operand fetch: component CI (0..8) of this wave's quadrant = global component cbase + (CI / 3) * 6 + CI % 3 ----
  const int cbase = 18 * qI + 3 * qJ;
  const int in_off = (cbase * W4_TILES + li) * PK + ((lh ^ ((li >> 3) & 1)) << 2);
  const int w_voff = (lh * NB + nt * 32 + li) * 16;
  f32x4 F0, F1, F2, Wt0, Wt1, Wt2;
#define W4_CO(CI) (((CI) / 3) * 6 + (CI) % 3)
#define W4_FETCH(S, BUF, CI, CHUNK)                                                                         \
  {                                                                                                         \
    F##S = W4_LD((BUF) + in_off + W4_CO(CI) * W4_TILES * PK);                                               \
    const int so_ = (((cob * nst + (CHUNK)) * W4_B_FLOATS) + (cbase + W4_CO(CI)) * 2 * NB * 4) * 4;         \
    Wt##S = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, w_voff, so_, 0));       \
  }
#define W4_MM(CI, E, S)                                                                                     \
  acc[CI] = __builtin_amdgcn_mfma_f32_32x32x2f32(Wt##S[E], F##S[E], acc[CI], 0, 0, 0);                      \
  __builtin_amdgcn_sched_barrier(0);
#define W4_FENCE() __builtin_amdgcn_sched_barrier(0)

  W4_FETCH(0, sA, 0, 0)
  W4_FETCH(1, sA, 1, 0)

  int tile = tile0, chunk = 0;
  for (int g = 0; g < nstages; ++g) {
    const int buf = g & 1;
    const float* const cA = sA + buf * W4_A_FLOATS;
    float* const nA = sA + (buf ^ 1) * W4_A_FLOATS;
    const int nchunk = chunk + 1 == nst ? 0 : chunk + 1;
    f32x4 raw0, raw1, raw2, raw3, T0, T1, T2, T3, T4, T5, ta, tb, tc, te;
    // ---- first half: components 0..4 || transform of stage g+1 (column order 0, 2, 4, 1, 3, 5) ----
    W4_FETCH(2, cA, 2, chunk)
    W4_FENCE();
    W4_MM(0, 0, 0) W4_TR_RD(0) W4_FENCE();
    W4_MM(0, 1, 0) W4_TR_T(T0) W4_TR_RD(2) W4_FENCE();
    W4_MM(0, 2, 0) W4_TR_T(T2) W4_TR_RD(4) W4_FENCE();
    W4_MM(0, 3, 0) W4_TR_T(T4) W4_TR_RD(1) W4_FENCE();
    W4_FETCH(0, cA, 3, chunk)
    W4_FENCE();
    W4_MM(1, 0, 1)
    if (tw) {
      tc = pk4_sub(T4, T2);
      ta = pk_fma_m44(T2, T4);
      W4_TR_WR(nA, 0, pk_fma_p44(pk4_sub(T0, T2), tc))
    }
    W4_FENCE();
    W4_MM(1, 1, 1) W4_TR_T(T1) W4_TR_RD(3) W4_FENCE();
    W4_MM(1, 2, 1) W4_TR_T(T3) W4_TR_RD(5) W4_FENCE();
    W4_MM(1, 3, 1) W4_TR_T(T5) W4_FENCE();
    W4_FETCH(1, cA, 4, chunk)
    W4_FENCE();
    W4_MM(2, 0, 2)
    if (tw) {
      te = pk4_sub(T3, T1);
      tb = pk_fma_m44(T1, T3);
      W4_TR_WR(nA, 5, pk_fma_m44(te, pk4_sub(T5, T3)))
    }
    W4_FENCE();
    W4_MM(2, 1, 2)
    if (tw) { W4_TR_WR(nA, 1, pk4_add(ta, tb)) W4_TR_WR(nA, 2, pk4_sub(ta, tb)) }
    W4_FENCE();
    W4_MM(2, 2, 2)
    if (tw) { W4_TR_WR(nA, 3, pk_fma_p24(te, tc)) W4_TR_WR(nA, 4, pk_fma_m24(te, tc)) }
    W4_FENCE();
    W4_MM(2, 3, 2)
    W4_FETCH(2, cA, 5, chunk)
    W4_FENCE();
    W4_MM(3, 0, 0) W4_MM(3, 1, 0) W4_MM(3, 2, 0) W4_MM(3, 3, 0)
    W4_FETCH(0, cA, 6, chunk)
    W4_FENCE();
    W4_MM(4, 0, 1) W4_MM(4, 1, 1) W4_MM(4, 2, 1) W4_MM(4, 3, 1)
    __syncthreads();  // barrier A: sA[~g&1] complete, sR free
    // ---- second half: components 5..8 || halo (g+2): registers -> sR, halo loads (g+3) ----
    W4_FETCH(1, cA, 7, chunk)
    W4_FENCE();
    W4_MM(5, 0, 2) W4_HALO_PAR() W4_HALO_BN(0) W4_FENCE();
    W4_MM(5, 1, 2) W4_HALO_WR(0) W4_FENCE();
    W4_MM(5, 2, 2) W4_HALO_BN(1) W4_FENCE();
    W4_MM(5, 3, 2) W4_HALO_WR(1) W4_FENCE();
    W4_FETCH(2, cA, 8, chunk)
    W4_FENCE();
    W4_MM(6, 0, 0) W4_HALO_BN(2) W4_FENCE();
    W4_MM(6, 1, 0) W4_HALO_WR(2) W4_FENCE();
    W4_MM(6, 2, 0)
    W4_ISSUE_HALO()  // a full stage ahead of their use
    W4_FENCE();
    W4_MM(6, 3, 0)
    // first components of the next stage (after a tile epilogue they are fetched behind it: the epilogue needs the registers)
    if (nchunk != 0) W4_FETCH(0, nA, 0, nchunk)
    W4_FENCE();
    W4_MM(7, 0, 1) W4_MM(7, 1, 1) W4_MM(7, 2, 1) W4_MM(7, 3, 1)
    if (nchunk != 0) W4_FETCH(1, nA, 1, nchunk)
    W4_FENCE();
    W4_MM(8, 0, 2) W4_MM(8, 1, 2) W4_MM(8, 2, 2) W4_MM(8, 3, 2)
    __syncthreads();  // barrier B: sR = raw(g+2) complete, sA[g&1] consumed

    if (++chunk == nst) {
#if W4_ABL & 1
      if (tid == 1023) p_out[0] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + acc[4][0] + acc[5][0] + acc[6][0] + acc[7][0] + acc[8][0];
#else
      // ---- tile epilogue ----
      const int tx_i = tile % a.tiles_x, t2 = tile / a.tiles_x;
      const int ty0 = (t2 % a.tiles_y) * TH, tx0 = tx_i * TW, n = t2 / a.tiles_y;
      int e_ty, e_tx;
      w4_tile_xy<WIDE>(li, e_ty, e_tx);
      const int oy = ty0 + 4 * e_ty + 2 * qI, ox = tx0 + 4 * e_tx + 2 * qJ;  // this lane's 2x2 pixel block
      const int co_l = nt * 32 + 4 * lh;                                     // + 8 gq + e: local channel of register 4 gq + e
      const __amdgpu_buffer_rsrc_t rsrc_out = __builtin_amdgcn_make_buffer_rsrc(
          p_out + (size_t)n * a.H * a.W * a.out_cs, 0, (unsigned)(a.H * a.W * a.out_cs) * 4u, 0x00020000);
      const unsigned obase = (unsigned)(((oy * a.W + ox) * a.out_cs + a.out_co + cob * NB + co_l) * 4);
      unsigned ooff[4];
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const bool in_ = (oy + (p >> 1) < a.H) && (ox + (p & 1) < a.W);
        ooff[p] = in_ ? obase + (unsigned)(((p >> 1) * a.W + (p & 1)) * a.out_cs) * 4u : OOB;
      }
      unsigned toff[4] = {OOB, OOB, OOB, OOB};
      __amdgpu_buffer_rsrc_t rsrc_t = rsrc_out;
      if (IN_MODE == 0 && a.bnr_mode != 0) {
        rsrc_t = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p_bnr) + (size_t)n * a.H * a.W * a.bnr_cs, 0,
                                                   (unsigned)(a.H * a.W * a.bnr_cs) * 4u, 0x00020000);
        const unsigned tbase = (unsigned)(((oy * a.W + ox) * a.bnr_cs + a.bnr_co + cob * NB + co_l) * 4);
#pragma unroll
        for (int p = 0; p < 4; ++p)
          toff[p] = ooff[p] != OOB ? tbase + (unsigned)(((p >> 1) * a.W + (p & 1)) * a.bnr_cs) * 4u : OOB;
      }
      // exchange slots: step 1 [wave][6 values][lane][4] in the consumed sA image (waves 0..5) / sXa (waves 6, 7),
      // partner = the other column half (wave ^ 2); step 2 [wave][4 values][lane][4] in sX2, partner = the other row half
      float* const sAc = sA + buf * W4_A_FLOATS;
      float* const x1w = (wave < 6 ? sAc + wave * 1536 : sXa + (wave - 6) * 1536) + lane * 4;
      const int pw = wave ^ 2;
      const float* const x1r = (pw < 6 ? sAc + pw * 1536 : sXa + (pw - 6) * 1536) + lane * 4;
      float* const x2w = sX2 + wave * 1024 + lane * 4;
      const float* const x2r = sX2 + (wave ^ 4) * 1024 + lane * 4;
      const bool tail = (cob + 1) * NB > a.Cout;  // block-uniform: channel quads that straddle Cout (operator tests only)
      mfma_results_guard();  // the output transform reads the accumulators from inline asm
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
#define W4_Q(C) (f32x4{acc[C][4 * gq], acc[C][4 * gq + 1], acc[C][4 * gq + 2], acc[C][4 * gq + 3]})
        // step 1: Q = M[I,J] A[J,:]; J = 0: Q = {m0 + m1 + m2, m1 - m2 | m1 + m2, m1 - m2}, J = 1: {m0 + m1, 2 (m0 - m1) |
        // 4 (m0 + m1), 8 (m0 - m1) + m2} (own quadrant columns m0..m2); kept half | sent half swap roles with J
        f32x4 kq[3][2];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const f32x4 m0 = W4_Q(3 * r), m1 = W4_Q(3 * r + 1), m2 = W4_Q(3 * r + 2);
          if (qJ == 0) {
            const f32x4 s = pk4_add(m1, m2), d = pk4_sub(m1, m2);
            kq[r][0] = pk4_add(m0, s); kq[r][1] = d;
            *reinterpret_cast<f32x4*>(x1w + (2 * r) * 256) = s;
            *reinterpret_cast<f32x4*>(x1w + (2 * r + 1) * 256) = d;
          } else {
            const f32x4 s = pk4_add(m0, m1), d = pk4_sub(m0, m1);
            *reinterpret_cast<f32x4*>(x1w + (2 * r) * 256) = s;
            const f32x4 d2 = pk4_add(d, d);
            *reinterpret_cast<f32x4*>(x1w + (2 * r + 1) * 256) = d2;
            const f32x4 s2 = pk4_add(s, s);
            kq[r][0] = pk4_add(s2, s2); kq[r][1] = pk_fma_p44(d2, m2);
          }
        }
#undef W4_Q
#pragma unroll
        for (int c = 0; c < 9; ++c)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[c][4 * gq + e] = 0.f;
        // fused BatchNorm-backward sums: the layer-below tensor at this lane's four pixels (latency under the barriers)
        f32x4 tq[4];
        if (IN_MODE == 0 && a.bnr_mode != 0) {
#pragma unroll
          for (int p = 0; p < 4; ++p)
            tq[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_t, toff[p], gq * 32, 0));
        }
        __syncthreads();
        f32x4 rr[3][2];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int x = 0; x < 2; ++x) rr[r][x] = pk4_add(kq[r][x], W4_LD(x1r + (2 * r + x) * 256));
        // step 2: P = A^T[:,I] R, same coefficient pattern along the rows
        f32x4 kp[2][2];
#pragma unroll
        for (int x = 0; x < 2; ++x) {
          if (qI == 0) {
            const f32x4 s = pk4_add(rr[1][x], rr[2][x]), d = pk4_sub(rr[1][x], rr[2][x]);
            kp[0][x] = pk4_add(rr[0][x], s); kp[1][x] = d;
            *reinterpret_cast<f32x4*>(x2w + x * 256) = s;
            *reinterpret_cast<f32x4*>(x2w + (2 + x) * 256) = d;
          } else {
            const f32x4 s = pk4_add(rr[0][x], rr[1][x]), d = pk4_sub(rr[0][x], rr[1][x]);
            *reinterpret_cast<f32x4*>(x2w + x * 256) = s;
            const f32x4 d2 = pk4_add(d, d), s2 = pk4_add(s, s);
            *reinterpret_cast<f32x4*>(x2w + (2 + x) * 256) = d2;
            kp[0][x] = pk4_add(s2, s2); kp[1][x] = pk_fma_p44(d2, rr[2][x]);
          }
        }
        __syncthreads();
        const f32x4 bq = W4_LD(sBias + co_l + 8 * gq);
        f32x4 s1a = {0.f, 0.f, 0.f, 0.f}, s2a = s1a, pm;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const f32x4 v = pk4_add(pk4_add(kp[p >> 1][p & 1], W4_LD(x2r + p * 256)), bq);
          f32x4 s1v, xw;
          if (IN_MODE == 0 && a.bnr_mode != 0) {
            const f32x4 q0 = W4_LD(sS + co_l + 8 * gq), q1 = W4_LD(sS + NB + co_l + 8 * gq);
            const f32x4 t = tq[p];
            f32x4 dz, xh;
            if (a.bnr_mode == 1) {
              const f32x4 q2v = W4_LD(sS + 2 * NB + co_l + 8 * gq), q3 = W4_LD(sS + 3 * NB + co_l + 8 * gq);
              const f32x4 z = pk4_fma(t, q0, q1);
              xh = pk4_fma(t, q2v, q3);
#pragma unroll
              for (int e = 0; e < 4; ++e) dz[e] = z[e] > 0.f ? v[e] : 0.f;
            } else {
              xh = (t - q0) * q1;
#pragma unroll
              for (int e = 0; e < 4; ++e) dz[e] = t[e] > 0.f ? v[e] : 0.f;
            }
            s1v = dz * (ooff[p] != OOB ? 1.f : 0.f); xw = xh;
          } else {
            s1v = v * (ooff[p] != OOB ? 1.f : 0.f); xw = v;
          }
          s1a = pk4_add(s1a, s1v);
          s2a = pk4_fma(s1v, xw, s2a);
          if (IN_MODE != 0 && p_pool != nullptr) {
            const f32x4 sv = v * W4_LD(sG + co_l + 8 * gq);
            if (p == 0) pm = sv;
            else {
#pragma unroll
              for (int e = 0; e < 4; ++e) pm[e] = fmaxf(pm[e], sv[e]);
            }
          }
          if (!tail) {
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), rsrc_out,
                                                   ooff[p], gq * 32, 0);
          } else {
            const int nvalid = a.Cout - (cob * NB + co_l + 8 * gq);
            if (ooff[p] != OOB && nvalid > 0) {
              float* q = p_out + (size_t)n * a.H * a.W * a.out_cs + (ooff[p] >> 2) + 8 * gq;
              q[0] = v[0];
              if (nvalid > 1) q[1] = v[1];
              if (nvalid > 2) q[2] = v[2];
              if (nvalid > 3) q[3] = v[3];
            }
          }
        }
        if (IN_MODE != 0 && p_pool != nullptr) {
          // the 2x2 block IS a pooling window (ConvArgs::pool_out: max for gamma >= 0, min for gamma < 0)
          if (oy < a.H && ox < a.W) {
            const f32x4 m = pm * W4_LD(sG + co_l + 8 * gq);
            *reinterpret_cast<f32x4*>(p_pool + ((size_t)(n * (a.H >> 1) + (oy >> 1)) * (a.W >> 1) + (ox >> 1)) * a.Cout + cob * NB +
                                      co_l + 8 * gq) = m;
          }
        }
        if (p_stats != nullptr) {
          // reduce-scatter of the 8 values over the 32 tile lanes (never across lh): 4 + 2 + 1 exchanges, then two plain
          // butterfly steps; lane li ends with value ((li >> 4) & 1) * 4 + ((li >> 3) & 1) * 2 + ((li >> 2) & 1)
          float st[8] = {s1a[0], s1a[1], s1a[2], s1a[3], s2a[0], s2a[1], s2a[2], s2a[3]};
#pragma unroll
          for (int w = 16, nv = 4; w >= 4; w >>= 1, nv >>= 1) {
            const bool up = (li & w) != 0;
#pragma unroll
            for (int i = 0; i < nv; ++i) {
              const float snd = up ? st[i] : st[i + nv];
              const float kp_ = up ? st[i + nv] : st[i];
              st[i] = kp_ + __shfl_xor(snd, w);
            }
          }
          st[0] += __shfl_xor(st[0], 2);
          st[0] += __shfl_xor(st[0], 1);
          stat_acc[gq] += st[0];
        }
      }
#endif
      chunk = 0;
      tile += per_cob;
      W4_ADDR_SETUP()
      W4_FETCH(0, nA, 0, 0)
      W4_FETCH(1, nA, 1, 0)
    }
  }
#undef W4_ISSUE_HALO
#undef W4_ADDR_SETUP
#undef W4_HALO_PAR
#undef W4_HALO_BN
#undef W4_HALO_WR
#undef W4_TR_RD
#undef W4_TR_T
#undef W4_TR_WR
#undef W4_TRANSFORM
#undef W4_FETCH
#undef W4_MM
#undef W4_FENCE
#undef W4_CO

  if (p_stats != nullptr && (li & 3) == 0) {
    const int which = (li >> 4) & 1, e = 2 * ((li >> 3) & 1) + ((li >> 2) & 1);
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const int co = cob * NB + nt * 32 + 4 * lh + 8 * gq + e;
      if (co < a.Cout)
        unsafeAtomicAdd(p_stats + (size_t)(blockIdx.x % NREP) * 2 * a.Cout + which * a.Cout + co, (double)stat_acc[gq]);
    }
  }
}
#undef W4_LD

}  // namespace sspk
