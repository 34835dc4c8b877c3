// Winograd F(4x4, 3x3) convolution for the large maps (240x320, 120x160): 36 transform components per 6x6 input patch,
// 4x4 outputs per tile = 2.25 matrix multiplies per output instead of the 4 of F(2x2,3x3) (conv_wino_pipe.hip.h), with
// about the same transform work per output.  fp32 error against fp64 on a 64 -> 64 layer's data: rel-L2 1.5e-6 (F(2x2):
// 2.4e-7; tools/archive/wino_error_probe.py).
//
//   workgroup = 4 waves (ONE per SIMD) = 32 tiles (32x16 or 16x32 output pixels) x 64 output channels, one per CU;
//   wave = (I, nt): rows 3I .. 3I+2 of the 6x6 component matrix M (18 components), output-channel half nt:
//          18 accumulators of 32 channels x 32 tiles = 288 registers of the wave's 512 (256 of them accumulation registers).
// The accumulators alone are 295 KB of the CU's 512 KB register file: with eight waves (the first version of this
// kernel, 144 + 112 registers per wave) every per-thread cost - operand sets, halo registers, LDS addresses - is paid
// twice and the kernel spilled 60-80 registers; four waves of 512 registers hold everything, and the output transform
// along the columns of M stays inside a wave.  One wave per SIMD hides no latency by itself: every operand is fetched
// one component pair (8 MFMAs) ahead and the transform reads its raw pixels one MFMA slot ahead.
//
// Operand roles as in conv_wino_p2_kernel: A = weight fragment (32 output channels x 2 k), B = transformed input fragment
// (2 k x 32 tiles); an accumulator lane holds ONE tile (column = lane & 31) and 16 output channels (row = (reg & 3) + 8 (reg
// >> 2) + 4 (lane >> 5)), so four consecutive channels sit in four consecutive registers (16-byte LDS / global accesses).
//
// Stage pipeline (8 input channels per stage = 72 MFMAs per wave in 9 component pairs, the two accumulators of a pair
// alternating; transformed input and raw halo double-buffered, ONE barrier per stage):
//
//   first half : pairs 0..4  ||  transform raw(g+1): sR[~g&1] -> sA[~g&1]  (the wave's transform task, below)
//   second half: pairs 5..8  ||  halo (g+2): registers -> sR[g&1] (BatchNorm + ReLU of the producer), halo loads (g+4)
//   barrier    (ONE per stage since round 6: raw halo and transformed input are both double-buffered)
//
// Three rotating operand sets (9 = 3 x 3 pairs per stage: the same rotation in every stage; two of them live at a time):
// weights from L2 (the packed
// image [cob][chunk8][component][h][64][4] IS the fragment layout), inputs from LDS.
//
// Epilogue Y = A^T M A (4x6 . 6x6 . 6x4) per register quad: R = M[I,:] A (3 x 4) inside the wave, P = A^T[:,I] R (4 x 4
// partial sums); the waves (0, nt) and (1, nt) swap the two output rows they do not keep (8 values per tile and channel
// through LDS, one barrier, double-buffered in the just-consumed sA image and sX) and each finishes a 2x4 pixel block =
// two pooling windows of its lane's tile for 16 channels in registers: 16-byte global stores straight from registers,
// BatchNorm sums by shuffles, pooled raw output in registers.
//
// Raw halo in LDS: [row][pixel][8 channels] at LDS offset 0, row pitch 40 / 24 pixels (a multiple of the 256-byte bank
// row); the pixels of halo row R are rotated by (R >> 2) & 3 inside aligned groups of 8: the transform's ds_read_b128
// (lane = (quad, tile), a 16-lane group = 8 tiles with distinct (tx & 1, ty & 3)) then hits 8 distinct 32-byte slots.
#pragma once
#include "conv_wino_p2.hip.h"
#include <type_traits>

#ifndef W4_NT
#define W4_NT 2  // cache-policy bits of the output stores and of the fused BatchNorm-backward tensor loads: 2 = nt (streamed once:
                 // they should not push the halo lines out of the XCD's L2; 0.65 instead of 0.68 ms per 64 -> 64 launch at 240x320)
#endif
#ifndef W4_TRACE
#define W4_TRACE 0  // compile-time perf trace (never in the shipped library): s_memtime stamps around the phases of a stage and the tile
                  // epilogue, summed per wave of workgroup 0 into ConvArgs::trace[wave * 8 + k]: k = 0 first half (pairs 0..4 beside
                  // the transform), 1 wait at barrier A, 2 second half (pairs 5..8 beside the halo staging), 3 wait at barrier B,
                  // 4 tile epilogue, 5 whole loop, 6 stages; tools/dbg/w4_trace.sh
#endif
#ifndef W4_ABL
#define W4_ABL 0  // compile-time perf ablation: 1 no tile epilogue, 2 no transform, 4 no halo staging, 16 no output stores,
                  // 32 no barrier in the epilogue rounds, 64 accumulators not cleared, 128 no barriers in the stage loop, 256 no
                  // weight loads, 512 no fragment reads from LDS in the stage loop, 1024 halo loads confined to 64 KB (cache
                  // hits), 2048 constants instead of the fused BatchNorm-backward tensor loads of the epilogue; results:
                  // profiles/r02_w4_ablation.txt, r03_kernel_experiments.txt
#endif

namespace sspk {

constexpr int W4C = 36;                                // Winograd components
constexpr int W4_THREADS = 256;
constexpr int W4_TILES = 32;                           // 4x4-pixel tiles per workgroup
constexpr int W4_A_FLOATS = W4C * W4_TILES * PK;       // 9216 floats = 36 KB transformed input per buffer
constexpr int W4_B_FLOATS = W4C * PK * NB;             // 18432 floats: packed weights per (cob, 8-channel chunk)
constexpr int W4_R_FLOATS = 34 * 24 * PK;              // raw halo: 34 rows x 24 pixels (16x32 tiles) >= 18 x 40 (32x16)
constexpr int W4_X_FLOATS = 4 * 8 * 256;               // row exchange: [wave][8 values][lane][4] = 32 KB
constexpr int W4_MAX_CIN = 256;                        // (scale | shift of the producer's BatchNorm in LDS; w4_eligible)
constexpr int W4_S_FLOATS = 2 * W4_MAX_CIN + 2 * NB;   // BatchNorm scale | shift (or 4 x 64 bnr parameters), bias, pool sign
// TWO raw-halo buffers (round 6): halo (g+2) goes into the buffer whose transform finished a stage ago while transform (g+1) reads
// the other one - ONE barrier per stage instead of two
constexpr int W4_LDS_BYTES = (2 * W4_A_FLOATS + 2 * W4_R_FLOATS + W4_X_FLOATS + W4_S_FLOATS) * 4;  // 161280
static_assert(W4_X_FLOATS <= W4_A_FLOATS, "the consumed transformed-input buffer is the second exchange buffer");
static_assert(W4_LDS_BYTES <= 160 * 1024, "one workgroup per CU");
static_assert(2 * W4_R_FLOATS * 4 < 65536, "16-bit raw-halo byte addresses (+ the buffer offset in the instruction)");
static_assert(W4_B_FLOATS == W4_PACK_FLOATS, "pack_wino4_element");

// k (wave-uniform, in a scalar register pair) * y + z
__device__ __forceinline__ f32x2 pk_fma_k(f32x2 k, f32x2 y, f32x2 z) {
  f32x2 d;
  asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "s"(k), "v"(y), "v"(z));
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

// Accumulators 0..15 live in FIXED accumulation registers a[16 A : 16 A + 15], touched by inline asm only (MFMA, clear,
// read).  hipcc keeps every accumulator of a builtin MFMA in accumulation registers once a kernel may use more than 256
// registers, and with 18 of them (288 registers) it spilled whole accumulators at the head of the stage loop; as
// compiler-managed values in vector registers they would not fit either.  The compiler sees none of these registers
// (w4_claim_agprs() makes it reserve all 256 in the kernel descriptor), so the hazards are handled here: consecutive
// MFMAs of one accumulator chain need no wait states, MFMA operands come from loads (s_waitcnt is inserted for asm
// operands), mfma_results_guard() precedes the reads, and a cleared accumulator is next used dozens of instructions later.
template <int A>
__device__ __forceinline__ void w4_mfma_a(float w, float f) {
  asm volatile("v_mfma_f32_32x32x2_f32 a[%2:%3], %0, %1, a[%2:%3]" : : "v"(w), "v"(f), "n"(A * 16), "n"(A * 16 + 15));
}
template <int R>
__device__ __forceinline__ float w4_acc_rd() {
  float x;
  asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(x) : "n"(R));
  return x;
}
template <int R>
__device__ __forceinline__ f32x4 w4_acc_quad() { return f32x4{w4_acc_rd<R>(), w4_acc_rd<R + 1>(), w4_acc_rd<R + 2>(), w4_acc_rd<R + 3>()}; }
// all 16 accumulators = 0, except accumulator 7 = component (1,1) of the I = 0 waves, which starts at `c11` (the conv bias:
// (1,1) enters all sixteen outputs of a tile with coefficient +1, which saves the bias adds of the epilogue)
template <int R = 0>
__device__ __forceinline__ void w4_acc_clear(float c11) {
  if constexpr (R / 16 == 7) asm volatile("v_accvgpr_write_b32 a[%0], %1" : : "n"(R), "v"(c11));
  else asm volatile("v_accvgpr_write_b32 a[%0], 0" : : "n"(R));
  if constexpr (R + 1 < 256) w4_acc_clear<R + 1>(c11);
}
// every one of a0..a255 is named: the compiler must treat all of them as clobbered here, and the kernel descriptor reserves
// the full accumulation-register file (hipbuild.verify_binary checks the BINARY: exactly the asm's own a-register
// instructions, no scratch, no spills)
#define W4_A8(B) "a" #B "0", "a" #B "1", "a" #B "2", "a" #B "3", "a" #B "4", "a" #B "5", "a" #B "6", "a" #B "7", "a" #B "8", "a" #B "9"
__device__ __forceinline__ void w4_claim_agprs() {
  asm volatile("" : : : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", W4_A8(1), W4_A8(2), W4_A8(3), W4_A8(4), W4_A8(5),
               W4_A8(6), W4_A8(7), W4_A8(8), W4_A8(9), W4_A8(10), W4_A8(11), W4_A8(12), W4_A8(13), W4_A8(14), W4_A8(15), W4_A8(16),
               W4_A8(17), W4_A8(18), W4_A8(19), W4_A8(20), W4_A8(21), W4_A8(22), W4_A8(23), W4_A8(24), "a250", "a251", "a252",
               "a253", "a254", "a255");
}
#undef W4_A8

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
  constexpr int NH = 5;                                // halo items (pixel, channel quad) per thread: 1224 / 256
  static_assert(HR * PITCH * PK <= W4_R_FLOATS, "raw halo buffer");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // [sR0][sR1][sA0][sA1][sX][sS]: the raw halo sits at LDS offset 0, so that two of its byte addresses fit one register
  float* const sR = smem;
  float* const sA = smem + 2 * W4_R_FLOATS;
  float* const sX = sA + 2 * W4_A_FLOATS;
  float* const sS = sX + W4_X_FLOATS;                  // IN_MODE 1: scale[Cin] | shift[Cin]; bnr: 4 x 64 parameters
  float* const sBias = sS + 2 * W4_MAX_CIN;
  float* const sG = sBias + NB;                        // +-1: sign of gamma (pooled raw output)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int nt = wave & 1, qI = wave >> 1;

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

  // ---- staging roles: raw halo items tid + 256 k (k < 5), item = pixel * 2 + quad ----
  const int q2 = tid & 1;
  int r_lds[NH];
#pragma unroll
  for (int k = 0; k < NH; ++k) {
    const int p = (tid + W4_THREADS * k) >> 1, r = p / HC, c = p - r * HC;
    r_lds[k] = (r * PITCH + (c & ~7) + ((c + ((r >> 2) & 3)) & 7)) * PK + q2 * 4;
  }
  const bool r4 = tid + 4 * W4_THREADS < NHALO * 2;  // the fifth item exists
  const int pixb = a.in_cs * 4, rowb = a.W * pixb;
  // TWO halo register sets, loaded two stages ahead of their use (set = stage parity): a stage (~2 us) is not enough distance
  // for an HBM round trip under load, and one wave per SIMD hides no latency by itself (0.065 of a 0.68 ms launch was this
  // wait).  Each set carries the 8-channel chunk of its loads (BatchNorm parameters at W4_HALO_BN) and the padding mask of
  // its slots - hoff[] belongs to the tile being LOADED and is rewritten when the loads move on to the next tile.
  f32x4 hregA[NH], hregB[NH];
  int h_chunkA = 0, h_chunkB = 0;
  unsigned h_padA = 0, h_padB = 0;  // bit k: slot k of the set is zero padding (outside the map)
  constexpr unsigned OOB = 0x80000000u;
  unsigned hoff[NH];
#pragma unroll
  for (int k = 0; k < NH; ++k) hoff[k] = OOB;
  const size_t img_floats = (size_t)a.H * a.W * a.in_cs;
  __amdgpu_buffer_rsrc_t rsrc_in;
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wpk), 0, a.wpk_bytes, 0x00020000);

  int ld_tile = tile0, ld_chunk = 0;
#define W4_ISSUE_HALO(SET)                                                                                  \
  {                                                                                                         \
    if (ld_chunk == 0) {                                                                                    \
      const int tt_ = min(ld_tile, t_end - 1);  /* past the end: harmless redundant loads of the last tile */ \
      const int tx_ = tt_ % a.tiles_x, t2_ = tt_ / a.tiles_x;                                               \
      const int ty0_ = (t2_ % a.tiles_y) * TH, tx0_ = tx_ * TW, n_ = t2_ / a.tiles_y;                       \
      _Pragma("unroll") for (int k = 0; k < NH; ++k) {                                                      \
        const int p_ = (tid + W4_THREADS * k) >> 1, r_ = p_ / HC, c_ = p_ - r_ * HC;                        \
        const int gy = ty0_ - 1 + r_, gx = tx0_ - 1 + c_;                                                   \
        const bool ok = (k < NH - 1 || r4) && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W; \
        hoff[k] = ok ? (unsigned)(gy * rowb + gx * pixb + (a.in_co + q2 * 4) * 4) : OOB;                    \
      }                                                                                                     \
      rsrc_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p_in) + (size_t)n_ * img_floats, 0,    \
                                                  a.in_bytes, 0x00020000);                                  \
    }                                                                                                       \
    h_chunk##SET = ld_chunk;                                                                                \
    h_pad##SET = 0;                                                                                         \
    _Pragma("unroll") for (int k = 0; k < NH; ++k) {                                                        \
      h_pad##SET |= (hoff[k] == OOB ? 1u : 0u) << k;                                                        \
      hreg##SET[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_in, (W4_ABL & 1024) ? (hoff[k] & 0xFFFFu) : hoff[k], ld_chunk * PK * 4, 0)); \
    }                                                                                                       \
    if (++ld_chunk == nst) { ld_chunk = 0; ld_tile += per_cob; }                                            \
  }
  f32x4 psc, psh;
#define W4_HALO_PAR(SET)                                                                                    \
  if (IN_MODE != 0) {                                                                                       \
    psc = *reinterpret_cast<const f32x4*>(sS + h_chunk##SET * PK + q2 * 4);                                 \
    psh = *reinterpret_cast<const f32x4*>(sS + W4_MAX_CIN + h_chunk##SET * PK + q2 * 4);                    \
  }
#define W4_HALO_BN(K, SET) if (IN_MODE != 0 && !(W4_ABL & 4)) hreg##SET[K] = bn_relu_quad(hreg##SET[K], psc, psh, (h_pad##SET >> (K)) & 1u);
#define W4_HALO_WR(K, SET) if (!(W4_ABL & 4) && ((K) < NH - 1 || r4)) *reinterpret_cast<f32x4*>(sR + rw_off + r_lds[K]) = hreg##SET[K];

  // ---- transform roles (round 6): V = B^T d B per (tile, channel quad) = (tid >> 1) & 31, tid & 1; the six rows of V are shared by
  // the four waves as TASKS with common subexpressions instead of one row + a duplicated second row per wave (8 row transforms for
  // 6 rows, 3 fma per raw column and row):
  //   wave 0: rows 1 & 2     p = d4 - 4 d2, q = d3 - 4 d1:   T1 = p + q,   T2 = p - q      (4 packed quad operations per column
  //   wave 1: rows 3 & 4     p = d4 -   d2, q = d3 -   d1:   T3 = p + 2 q, T4 = p - 2 q     for two rows: raw rows 1..4)
  //   wave 2: row 0          T0 = 4 d0 - 5 d2 + d4                                          (2 per column: raw rows 0, 2, 4)
  //   wave 3: row 5          T5 = 4 d1 - 5 d3 + d5                                          (raw rows 1, 3, 5)
  // followed by the row pass V[r][.] = T[r][.] B per row (12 quad operations).  Critical wave: 96 packed instructions + 24 LDS
  // reads per stage instead of 130 + 48; waves 2, 3 (48 + 18) wait at barrier A.
  constexpr bool twB = !(W4_ABL & 2);
  const int t_tile = (tid >> 1) & 31;
  int t_ty, t_tx;
  w4_tile_xy<WIDE>(t_tile, t_ty, t_tx);
  const bool t_pair = wave < 2;                       // wave-uniform task kind
  const int t_r0 = wave == 2 ? 0 : 1;                 // first raw row read
  const int t_row = wave == 0 ? 1 : wave == 1 ? 3 : wave == 2 ? 0 : 5;   // (first) V row written
  // (readfirstlane: the coefficients must reach the packed fmas in scalar register pairs, not in per-lane selects)
#define W4_UNI(X) __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, (X))))
  const float kp_ = W4_UNI(wave == 0 ? -4.f : wave == 1 ? -1.f : -5.f);   // pair: p = d4 + kp d2 (= kq); single: -5
  const float ks_ = W4_UNI(wave == 1 ? 2.f : 1.f);                        // pair: T = p +- ks q
  const float kn_ = W4_UNI(wave == 1 ? -2.f : -1.f);
#undef W4_UNI
  const f32x2 tkp = {kp_, kp_}, tks = {ks_, ks_}, tkn = {kn_, kn_};
  const int dLb = wave == 3 ? ROWF * 4 : 0;           // row 5: its last raw row is row 5, not row 4
  int t_ab[6];  // byte addresses of the first raw row read | of raw row 4 << 16
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    const int C = 4 * t_tx + c, Ra = 4 * t_ty + t_r0, Rb = 4 * t_ty + 4;
    const int ta_ = (Ra * PITCH + (C & ~7) + ((C + ((Ra >> 2) & 3)) & 7)) * PK + q2 * 4;
    const int tb_ = (Rb * PITCH + (C & ~7) + ((C + ((Rb >> 2) & 3)) & 7)) * PK + q2 * 4;
    t_ab[c] = (ta_ * 4) | ((tb_ * 4) << 16);
  }
  const int t_dst = (t_row * 6 * W4_TILES + t_tile) * PK + ((q2 ^ ((t_tile >> 3) & 1)) << 2);
  constexpr int DST_B = 6 * W4_TILES * PK;            // the pair's second row
#define W4_LD(P) (*reinterpret_cast<const f32x4*>(P))
#define W4_FENCE() __builtin_amdgcn_sched_barrier(0)
  // raw pixels of column C, one MFMA slot ahead of their use: pair task rx0..3 = raw rows 1..4; single task rx0, rx1 = the two
  // rows at distance 2 from the first one, rx3 = the last row (4 or 5)
#define W4_TR_RD(C)                                                                                         \
  {                                                                                                         \
    const char* pa_ = reinterpret_cast<const char*>(smem) + (t_ab[C] & 0xffff);                             \
    rx0 = W4_LD(pa_); rx1 = W4_LD(pa_ + ROWF * 4); rx2 = W4_LD(pa_ + 2 * ROWF * 4);                         \
    rx3 = W4_LD(reinterpret_cast<const char*>(smem) + ((unsigned)t_ab[C] >> 16));                           \
  }
// pair task, round 6b: two raw-pixel register sets (X = rxa / rxb) so that the reads of the NEXT column are the first instructions
// behind an MFMA (they issue in its shadow; behind the column arithmetic they each took an issue slot of their own), with the two
// byte addresses (pa_n, pb_n) unpacked at the end of the previous slice
#define W4_TR_ADDR(C) { pa_n = (t_ab[C] & 0xffff) + tr_off; pb_n = (int)((unsigned)t_ab[C] >> 16) + tr_off; }
#define W4_TR_RDX(X)                                                                                        \
  {                                                                                                         \
    const char* pa_ = reinterpret_cast<const char*>(smem) + pa_n;                                           \
    X##0 = W4_LD(pa_); X##1 = W4_LD(pa_ + ROWF * 4); X##2 = W4_LD(pa_ + 2 * ROWF * 4);                      \
    X##3 = W4_LD(reinterpret_cast<const char*>(smem) + pb_n);                                               \
  }
#define W4_PCX(C, X) { const f32x4 p_ = pk4_fma_k(tkp, X##1, X##3), q_ = pk4_fma_k(tkp, X##0, X##2); TA##C = pk4_fma_k(tks, q_, p_); TB##C = pk4_fma_k(tkn, q_, p_); }
#define W4_TR_RD3(C)                                                                                        \
  {                                                                                                         \
    const char* pa_ = reinterpret_cast<const char*>(smem) + (t_ab[C] & 0xffff);                             \
    rx0 = W4_LD(pa_); rx1 = W4_LD(pa_ + 2 * ROWF * 4);                                                      \
    rx3 = W4_LD(reinterpret_cast<const char*>(smem) + (((unsigned)t_ab[C] >> 16) + dLb));                   \
  }
  // column pass: pair -> TA##C, TB##C; single -> TA##C
#define W4_PC(C) { const f32x4 p_ = pk4_fma_k(tkp, rx1, rx3), q_ = pk4_fma_k(tkp, rx0, rx2); TA##C = pk4_fma_k(tks, q_, p_); TB##C = pk4_fma_k(tkn, q_, p_); }
#define W4_SC(C) TA##C = pk_fma_p44(rx0, pk4_fma_k(tkp, rx1, rx3));
#define W4_TR_WR(DSTBUF, J, V) *reinterpret_cast<f32x4*>((DSTBUF) + t_dst + (J) * W4_TILES * PK) = (V);
  // row pass of one V row from its six column-pass values T##0..T##5 (in four slices; a, c / b, e: even / odd columns)
#define W4_RP_A(T, DSTBUF, TA_, TC_) { TC_ = pk4_sub(T##4, T##2); TA_ = pk_fma_m44(T##2, T##4); W4_TR_WR(DSTBUF, 0, pk_fma_p44(pk4_sub(T##0, T##2), TC_)) }
#define W4_RP_B(T, DSTBUF) { te = pk4_sub(T##3, T##1); tb = pk_fma_m44(T##1, T##3); W4_TR_WR(DSTBUF, 5, pk_fma_m44(te, pk4_sub(T##5, T##3))) }
#define W4_RP_C(TA_, TB_, DSTBUF) { W4_TR_WR(DSTBUF, 1, pk4_add(TA_, TB_)) W4_TR_WR(DSTBUF, 2, pk4_sub(TA_, TB_)) }
#define W4_RP_D(TC_, TE_, DSTBUF) { W4_TR_WR(DSTBUF, 3, pk_fma_p24(TE_, TC_)) W4_TR_WR(DSTBUF, 4, pk_fma_m24(TE_, TC_)) }
  // the slices of the two task kinds (S = slice number; the stage body puts one slice behind one MFMA, the prologue runs them
  // back to back); row A's even-column values (ta, tc) stay live across the odd columns, row B's go through ta2, tc2
// (the row-pass outputs of a slice are written at the START of the next slice - pw0 / pw1 -: a ds_write_b128 right behind an MFMA
// issues in its shadow, behind the arithmetic that produced its data it took ~10 cycles of its own)
#define W4_RPX_A(T, TA_, TC_) { TC_ = pk4_sub(T##4, T##2); TA_ = pk_fma_m44(T##2, T##4); pw0 = pk_fma_p44(pk4_sub(T##0, T##2), TC_); }
#define W4_RPX_B(T) { te = pk4_sub(T##3, T##1); tb = pk_fma_m44(T##1, T##3); pw0 = pk_fma_m44(te, pk4_sub(T##5, T##3)); }
#define W4_RPX_C(TA_) { pw0 = pk4_add(TA_, tb); pw1 = pk4_sub(TA_, tb); }
#define W4_RPX_D(TC_) { pw0 = pk_fma_p24(te, TC_); pw1 = pk_fma_m24(te, TC_); }
#define W4_PSL(S, DSTBUF)                                                                                   \
  if (twB) {                                                                                                \
    if constexpr ((S) == 0) { W4_TR_ADDR(0) W4_TR_RDX(rxa) W4_TR_ADDR(2) }                                  \
    else if constexpr ((S) == 1) { W4_TR_RDX(rxb) W4_FENCE(); W4_PCX(0, rxa) W4_TR_ADDR(4) }                \
    else if constexpr ((S) == 2) { W4_TR_RDX(rxa) W4_FENCE(); W4_PCX(2, rxb) W4_TR_ADDR(1) }                \
    else if constexpr ((S) == 3) { W4_TR_RDX(rxb) W4_FENCE(); W4_PCX(4, rxa) W4_TR_ADDR(3) }                \
    else if constexpr ((S) == 4) { W4_TR_RDX(rxa) W4_FENCE(); W4_RPX_A(TA, ta, tc) }                        \
    else if constexpr ((S) == 5) { W4_TR_WR(DSTBUF, 0, pw0) W4_FENCE(); W4_RPX_A(TB, ta2, tc2) }            \
    else if constexpr ((S) == 6) { W4_TR_WR((DSTBUF) + DST_B, 0, pw0) W4_FENCE(); W4_PCX(1, rxb) W4_TR_ADDR(5) } \
    else if constexpr ((S) == 7) { W4_TR_RDX(rxb) W4_FENCE(); W4_PCX(3, rxa) }                              \
    else if constexpr ((S) == 8) W4_PCX(5, rxb)                                                             \
    else if constexpr ((S) == 9) W4_RPX_B(TA)                                                               \
    else if constexpr ((S) == 10) { W4_TR_WR(DSTBUF, 5, pw0) W4_FENCE(); W4_RPX_C(ta) }                     \
    else if constexpr ((S) == 11) { W4_TR_WR(DSTBUF, 1, pw0) W4_TR_WR(DSTBUF, 2, pw1) W4_FENCE(); W4_RPX_D(tc) } \
    else if constexpr ((S) == 12) { W4_TR_WR(DSTBUF, 3, pw0) W4_TR_WR(DSTBUF, 4, pw1) W4_FENCE(); W4_RPX_B(TB) } \
    else if constexpr ((S) == 13) { W4_TR_WR((DSTBUF) + DST_B, 5, pw0) W4_FENCE(); W4_RPX_C(ta2) }          \
    else if constexpr ((S) == 14) { W4_TR_WR((DSTBUF) + DST_B, 1, pw0) W4_TR_WR((DSTBUF) + DST_B, 2, pw1) W4_FENCE(); W4_RPX_D(tc2) } \
    else if constexpr ((S) == 15) { W4_TR_WR((DSTBUF) + DST_B, 3, pw0) W4_TR_WR((DSTBUF) + DST_B, 4, pw1) } \
  }
#define W4_TR_RD3X(X)                                                                                       \
  {                                                                                                         \
    const char* pa_ = reinterpret_cast<const char*>(smem) + pa_n;                                           \
    X##0 = W4_LD(pa_); X##1 = W4_LD(pa_ + 2 * ROWF * 4);                                                    \
    X##3 = W4_LD(reinterpret_cast<const char*>(smem) + (pb_n + dLb));                                       \
  }
#define W4_SCX(C, X) TA##C = pk_fma_p44(X##0, pk4_fma_k(tkp, X##1, X##3));
#define W4_SSL(S, DSTBUF)                                                                                   \
  if (twB) {                                                                                                \
    if constexpr ((S) == 0) { W4_TR_ADDR(0) W4_TR_RD3X(rxa) W4_TR_ADDR(2) }                                 \
    else if constexpr ((S) == 1) { W4_TR_RD3X(rxb) W4_FENCE(); W4_SCX(0, rxa) W4_TR_ADDR(4) }               \
    else if constexpr ((S) == 2) { W4_TR_RD3X(rxa) W4_FENCE(); W4_SCX(2, rxb) W4_TR_ADDR(1) }               \
    else if constexpr ((S) == 3) { W4_TR_RD3X(rxb) W4_FENCE(); W4_SCX(4, rxa) W4_TR_ADDR(3) }               \
    else if constexpr ((S) == 4) { W4_TR_RD3X(rxa) W4_FENCE(); W4_RPX_A(TA, ta, tc) }                       \
    else if constexpr ((S) == 5) { W4_TR_WR(DSTBUF, 0, pw0) W4_FENCE(); W4_SCX(1, rxb) W4_TR_ADDR(5) }      \
    else if constexpr ((S) == 6) { W4_TR_RD3X(rxb) W4_FENCE(); W4_SCX(3, rxa) }                             \
    else if constexpr ((S) == 7) W4_SCX(5, rxb)                                                             \
    else if constexpr ((S) == 8) W4_RPX_B(TA)                                                               \
    else if constexpr ((S) == 9) { W4_TR_WR(DSTBUF, 5, pw0) W4_FENCE(); W4_RPX_C(ta) }                      \
    else if constexpr ((S) == 10) { W4_TR_WR(DSTBUF, 1, pw0) W4_TR_WR(DSTBUF, 2, pw1) W4_FENCE(); W4_RPX_D(tc) } \
    else if constexpr ((S) == 11) { W4_TR_WR(DSTBUF, 3, pw0) W4_TR_WR(DSTBUF, 4, pw1) }                     \
  }
#define W4_XF_REGS f32x4 rx0, rx1, rx2, rx3, rxa0, rxa1, rxa2, rxa3, rxb0, rxb1, rxb2, rxb3; int pa_n, pb_n; f32x4 pw0, pw1, TA0, TA1, TA2, TA3, TA4, TA5, TB0, TB1, TB2, TB3, TB4, TB5, ta, tb, tc, te, ta2, tc2;
  // the whole task, unsliced (prologue)
#define W4_TRANSFORM_ALL(DSTBUF)                                                                            \
  {                                                                                                         \
    W4_XF_REGS                                                                                              \
    if (t_pair) {                                                                                           \
      W4_PSL(0, DSTBUF) W4_PSL(1, DSTBUF) W4_PSL(2, DSTBUF) W4_PSL(3, DSTBUF) W4_PSL(4, DSTBUF) W4_PSL(5, DSTBUF) W4_PSL(6, DSTBUF) W4_PSL(7, DSTBUF) \
      W4_PSL(8, DSTBUF) W4_PSL(9, DSTBUF) W4_PSL(10, DSTBUF) W4_PSL(11, DSTBUF) W4_PSL(12, DSTBUF) W4_PSL(13, DSTBUF) W4_PSL(14, DSTBUF) \
      W4_PSL(15, DSTBUF)                                                                                    \
    } else {                                                                                                \
      W4_SSL(0, DSTBUF) W4_SSL(1, DSTBUF) W4_SSL(2, DSTBUF) W4_SSL(3, DSTBUF) W4_SSL(4, DSTBUF) W4_SSL(5, DSTBUF) W4_SSL(6, DSTBUF) W4_SSL(7, DSTBUF) \
      W4_SSL(8, DSTBUF) W4_SSL(9, DSTBUF) W4_SSL(10, DSTBUF) W4_SSL(11, DSTBUF)                             \
    }                                                                                                       \
  }

  // ---- per-block parameters in LDS ----
  if (tid < NB) {
    const int co_ = cob * NB + tid;
    sBias[tid] = (a.bias != nullptr && co_ < a.Cout) ? a.bias[co_] : 0.f;
    if (IN_MODE != 0) sG[tid] = (p_pool != nullptr && co_ < a.Cout && a.pool_gamma[co_] < 0.f) ? -1.f : 1.f;
  }
  if (IN_MODE != 0) {
    if (a.lazy.mode == 0) {
      for (int c = tid; c < a.Cin; c += W4_THREADS) {
        sS[c] = p_scale[c];
        sS[W4_MAX_CIN + c] = p_shift[c];
      }
    } else {   // (BnLazy: the producer's statistics -> affine here; workgroup 0 stores for the later readers)
      for (int c = tid; c < a.Cin; c += W4_THREADS) {
        float sc_, sh_;
        bn_lazy_affine(a.lazy, prob, c, sc_, sh_);
        sS[c] = sc_;
        sS[W4_MAX_CIN + c] = sh_;
        if (blockIdx.x == 0) bn_lazy_store(a.lazy, c);
      }
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
#define W4_HALO_ALL(SET)                                                                                    \
  W4_HALO_PAR(SET) W4_HALO_BN(0, SET) W4_HALO_BN(1, SET) W4_HALO_BN(2, SET) W4_HALO_BN(3, SET) W4_HALO_BN(4, SET) \
  W4_HALO_WR(0, SET) W4_HALO_WR(1, SET) W4_HALO_WR(2, SET) W4_HALO_WR(3, SET) W4_HALO_WR(4, SET)
  W4_ISSUE_HALO(A)     // stage 0
  W4_ISSUE_HALO(B)     // stage 1
  { const int rw_off = 0; W4_HALO_ALL(A) }                 // raw(0) -> sR[0]
  __syncthreads();
  { const int tr_off = 0; W4_TRANSFORM_ALL(sA) }
  W4_ISSUE_HALO(A)     // stage 2: consumed by stage 0 of the loop (even stages use set A)
  { const int rw_off = W4_R_FLOATS; W4_HALO_ALL(B) }       // raw(1) -> sR[1]
  W4_ISSUE_HALO(B)     // stage 3: consumed by stage 1 of the loop
  __syncthreads();
#undef W4_HALO_ALL

  // accumulators 0..15: a[0:255] (w4_mfma_a / w4_acc_quad / w4_acc_clear); 16, 17: ordinary vector registers
  w4_claim_agprs();
  const float c11 = qI == 0 ? sBias[nt * 32 + li] : 0.f;  // (after the parameter barrier above)
  w4_acc_clear(c11);
  // (cleared from an opaque zero: under register pressure hipcc materialised the constant in an accumulation register - a0 -
  // and copied it around, i.e. into accumulators it does not know about)
  f32x16 acc16, acc17;
#define W4_CLEAR_V()                                                                                        \
  {                                                                                                         \
    float z_;                                                                                               \
    asm volatile("v_mov_b32 %0, 0" : "=v"(z_));                                                             \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) { acc16[r] = z_; acc17[r] = z_; }                        \
  }
  W4_CLEAR_V()
  f32x2 stat_s1 = {0.f, 0.f}, stat_s2 = {0.f, 0.f};  // BatchNorm sums of this lane's output channel (two partial sums each)

  // ---- operand fetch: component pair P (components 2P, 2P + 1 of this wave's 18 = global components 18 I + ..) ----
  const int cbase = 18 * qI;
  const int in_off = (cbase * W4_TILES + li) * PK + ((lh ^ ((li >> 3) & 1)) << 2);
  const int w_voff = (lh * NB + nt * 32 + li) * 16;
  // Weight sets 3, 4 hold pairs 0, 1 of the NEXT stage: together with the weights of pairs 7, 8 they are fetched right
  // before the halo loads of a stage are issued.  The memory counter (vmcnt) retires loads in order: the first weight fetch
  // issued AFTER the halo loads cannot be consumed before they have come back from HBM, so that fetch is pushed 4.5 pairs
  // (~1 us) behind the halo issue instead of one pair (the coupling cost 0.05 ms of a 0.75 ms launch: halo staging +0.12 ms
  // with the weight loads in the loop, +0.07 without, profiles/r02_w4_ablation.txt).
  f32x4 Fa0, Fb0, Wa0, Wb0, Fa1, Fb1, Wa1, Wb1, Fa2, Fb2, Wa2, Wb2, Wa3, Wb3, Wa4, Wb4;
#define W4_FETCH_F(S, BUF, P)                                                                               \
  if (!(W4_ABL & 512) || g < 0) {                                                                           \
    Fa##S = W4_LD((BUF) + in_off + (2 * (P)) * W4_TILES * PK);                                              \
    Fb##S = W4_LD((BUF) + in_off + (2 * (P) + 1) * W4_TILES * PK);                                          \
  }
#define W4_FETCH_W(S, P, CHUNK)                                                                             \
  if (!(W4_ABL & 256) || g < 0) {                                                                           \
    const int so_ = (((cob * nst + (CHUNK)) * W4_B_FLOATS) + (cbase + 2 * (P)) * 2 * NB * 4) * 4;           \
    Wa##S = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, w_voff, so_, 0));       \
    Wb##S = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, w_voff, so_ + 2 * NB * 16, 0)); \
  }
#define W4_FETCH(S, BUF, P, CHUNK) { W4_FETCH_F(S, BUF, P) W4_FETCH_W(S, P, CHUNK) }
  // MFMA number I (0..7) of pair P: component 2P + (I & 1), k pair I >> 1
#define W4_MMX(P, I, S, SW)                                                                                 \
  if ((P) == 8) {                                                                                           \
    if (((I) & 1) == 0) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc16) : "v"(Fa##S[(I) >> 1]), "v"(Wa##SW[(I) >> 1])); \
    else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc17) : "v"(Fb##S[(I) >> 1]), "v"(Wb##SW[(I) >> 1]));           \
  } else if (((I) & 1) == 0) w4_mfma_a<((P) < 8 ? 2 * (P) : 0)>(Fa##S[(I) >> 1], Wa##SW[(I) >> 1]);          \
  else w4_mfma_a<((P) < 8 ? 2 * (P) + 1 : 0)>(Fb##S[(I) >> 1], Wb##SW[(I) >> 1]);                            \
  __builtin_amdgcn_sched_barrier(0);
#define W4_MM(P, I, S) W4_MMX(P, I, S, S)
#define W4_MM8(P, S) W4_MM(P, 0, S) W4_MM(P, 1, S) W4_MM(P, 2, S) W4_MM(P, 3, S) W4_MM(P, 4, S) W4_MM(P, 5, S) W4_MM(P, 6, S) W4_MM(P, 7, S)

  int g = -1;  // (the ablation macros test g < 0 = prologue)
  W4_FETCH_F(0, sA, 0)
  W4_FETCH_W(3, 0, 0)
  W4_FETCH_W(4, 1, 0)
  if (W4_ABL & (256 | 512)) { W4_FETCH(0, sA, 0, 0) W4_FETCH(1, sA, 1, 0) W4_FETCH(2, sA, 2, 0) }

  // first half of a stage: pairs 0..4 (40 MFMAs); transform slice k of the wave's task behind MFMA slot k (slots = the MFMAs of
  // pairs 0, 2 and the first four of pairs 1, 3; the pair task has 16 slices, the single-row task 12)
#define W4_PSLN(S) W4_PSL(S, nA)
#define W4_SSLN(S) W4_SSL(S, nA)
#define W4_HALF1(SL)                                                                                                  \
    W4_MMX(0, 0, 0, 3) SL(0) W4_FENCE();                                                                              \
    W4_MMX(0, 1, 0, 3) SL(1) W4_FENCE();                                                                              \
    W4_MMX(0, 2, 0, 3) SL(2) W4_FENCE();                                                                              \
    W4_MMX(0, 3, 0, 3) SL(3) W4_FENCE();                                                                              \
    W4_MMX(0, 4, 0, 3) SL(4) W4_FENCE();                                                                              \
    W4_MMX(0, 5, 0, 3) SL(5) W4_FENCE();                                                                              \
    W4_MMX(0, 6, 0, 3) SL(6) W4_FENCE();                                                                              \
    W4_MMX(0, 7, 0, 3) W4_FETCH(2, cA, 2, chunk) W4_FENCE(); SL(7) W4_FENCE();                                        \
    W4_MMX(1, 0, 1, 4) SL(8) W4_FENCE();                                                                              \
    W4_MMX(1, 1, 1, 4) SL(9) W4_FENCE();                                                                              \
    W4_MMX(1, 2, 1, 4) SL(10) W4_FENCE();                                                                             \
    W4_MMX(1, 3, 1, 4) SL(11) W4_FENCE();                                                                             \
    W4_MMX(1, 4, 1, 4) W4_MMX(1, 5, 1, 4) W4_MMX(1, 6, 1, 4) W4_MMX(1, 7, 1, 4)                                       \
    W4_FETCH(0, cA, 3, chunk)                                                                                         \
    W4_FENCE();                                                                                                       \
    W4_MM(2, 0, 2) SL(12) W4_FENCE();                                                                                 \
    W4_MM(2, 1, 2) SL(13) W4_FENCE();                                                                                 \
    W4_MM(2, 2, 2) SL(14) W4_FENCE();                                                                                 \
    W4_MM(2, 3, 2) SL(15) W4_FENCE();                                                                                 \
    W4_MM(2, 4, 2) W4_MM(2, 5, 2) W4_MM(2, 6, 2)                                                                      \
    W4_MM(2, 7, 2) W4_FETCH(1, cA, 4, chunk) W4_FENCE();                                                              \
    W4_MM8(3, 0)                                                                                                      \
    W4_FETCH(2, cA, 5, chunk)                                                                                         \
    W4_FENCE();                                                                                                       \
    W4_MM8(4, 1)

  // One stage (8 input channels, 72 MFMAs per wave) of the pipeline; SET = the halo register set of the stage's parity, BUFV
  // = the transformed-input buffer it consumes (both = the stage parity: the stage loop is unrolled by two, nst is even).
  //   first half : pairs 0..4  ||  transform of stage g+1 (barrier A: sA[~g&1] complete, sR free)
  //   second half: pairs 5..8  ||  halo (g+2): set -> sR, then the halo loads of stage g+4 into the same set; every weight
  //                needed before those loads come back is fetched BEFORE them (in-order load counter), incl. pairs 0, 1 of
  //                the next stage; barrier B: sR = raw(g+2) complete, sA[g&1] consumed
#define W4_STAGE(SET, BUFV)                                                                                           \
    const int buf = (BUFV);                                                                                           \
    const float* const cA = sA + buf * W4_A_FLOATS;                                                                   \
    float* const nA = sA + (buf ^ 1) * W4_A_FLOATS;                                                                   \
    const int nchunk = chunk + 1 == nst ? 0 : chunk + 1;                                                              \
    const int rw_off = buf * W4_R_FLOATS;            /* halo (g+2) -> sR[g & 1] (floats) */                             \
    const int tr_off = (buf ^ 1) * W4_R_FLOATS * 4;  /* transform (g+1) <- sR[(g+1) & 1] (bytes) */                      \
    W4_XF_REGS                                                                                                        \
    W4_FETCH_F(1, cA, 1)                                                                                              \
    W4_FENCE();                                                                                                       \
    if (t_pair) { W4_HALF1(W4_PSLN) } else { W4_HALF1(W4_SSLN) }                                                      \
    W4_HALO_PAR(SET)   /* (in front of the barrier: the LDS round trip of the BN parameters was exposed in front of the first BN slice) */ \
    W4_FENCE();                                                                                                       \
    W4_T(0)                                                                                                           \
    W4_T(1)   /* (no barrier here since round 6: the raw halo is double-buffered) */                                    \
    W4_FETCH(0, cA, 6, chunk)                                                                                         \
    W4_FENCE();                                                                                                       \
    W4_MM(5, 0, 2) W4_HALO_BN(0, SET) W4_FENCE();                                                                     \
    W4_MM(5, 1, 2) W4_HALO_WR(0, SET) W4_FENCE();                                                                     \
    W4_MM(5, 2, 2) W4_HALO_BN(1, SET) W4_FENCE();                                                                     \
    W4_MM(5, 3, 2) W4_HALO_WR(1, SET) W4_FENCE();                                                                     \
    W4_MM(5, 4, 2) W4_HALO_BN(2, SET) W4_FENCE();                                                                     \
    W4_MM(5, 5, 2) W4_HALO_WR(2, SET) W4_FENCE();                                                                     \
    W4_MM(5, 6, 2) W4_HALO_BN(3, SET) W4_FENCE();                                                                     \
    W4_MM(5, 7, 2) W4_HALO_WR(3, SET) W4_FENCE();                                                                     \
    W4_FETCH(1, cA, 7, chunk)                                                                                         \
    W4_FETCH_W(2, 8, chunk)                                                                                           \
    W4_FENCE();                                                                                                       \
    /* (the weight window of the next stage behind the NEXT MFMA: ~26 scalar / memory instructions do not fit one shadow) */ \
    W4_MM(6, 0, 0) W4_FETCH_W(3, 0, nchunk) W4_FETCH_W(4, 1, nchunk) W4_FENCE(); W4_HALO_BN(4, SET) W4_FENCE();       \
    W4_MM(6, 1, 0) W4_HALO_WR(4, SET) W4_FENCE();                                                                     \
    W4_MM(6, 2, 0)                                                                                                    \
    W4_ISSUE_HALO(SET)                                                                                                \
    W4_FENCE();                                                                                                       \
    W4_MM(6, 3, 0) W4_MM(6, 4, 0) W4_MM(6, 5, 0) W4_MM(6, 6, 0) W4_MM(6, 7, 0)                                        \
    W4_FETCH_F(2, cA, 8)                                                                                              \
    W4_FENCE();                                                                                                       \
    W4_MM8(7, 1)                                                                                                      \
    W4_T(2)                                                                                                           \
    if (!(W4_ABL & 128)) __syncthreads();                                                                             \
    W4_T(3)                                                                                                           \
    /* the first fragments of the next stage (other waves wrote them during THIS stage: behind the stage's one barrier), */ \
    /* then pair 8 from operands fetched in front of the barrier: its 8 MFMAs cover the LDS round trip                   */ \
    if (nchunk != 0) W4_FETCH_F(0, nA, 0)                                                                             \
    W4_FENCE();                                                                                                       \
    W4_MM8(8, 2)

#if W4_TRACE
  unsigned long long tc_[5] = {0, 0, 0, 0, 0}, tp_ = __builtin_readcyclecounter();
  const unsigned long long tk0_ = tp_;
#define W4_T(K) { const unsigned long long tn_ = __builtin_readcyclecounter(); tc_[K] += tn_ - tp_; tp_ = tn_; }
#else
#define W4_T(K)
#endif
  static_assert(PK == 8, "stage = 8 input channels");
  int tile = tile0, chunk = 0;
  for (g = 0; g < nstages; g += 2) {  // nst = Cin / 8 is even (Cin % 16 == 0): a tile ends behind an odd stage
    { W4_STAGE(A, 0) }
    ++chunk;
    { W4_STAGE(B, 1) }
    const int buf = 1;
    float* const nA = sA;

    if (++chunk == nst) {
#if W4_ABL & 1
      if (tid == 1023) {
        p_out[0] = w4_acc_rd<0>() + w4_acc_rd<17>() + w4_acc_rd<250>() + acc16[0] + acc17[1];
      }
#else
      // ---- tile epilogue ----
      // accumulator element (lane (li, lh), register 4 gq + e) = output channel nt * 32 + li of tile e + 4 lh + 8 gq, i.e.
      // (w4_tile_xy) tile row e (+ 4 (gq >> 1) for the 16x32 blocks), tile column 2 gq + lh (2 (gq & 1) + lh): everything
      // but the lh part of a pixel address is wave-uniform (scalar offsets), and the 32 lanes of a half wave store 128
      // contiguous bytes of one pixel
      const int tx_i = tile % a.tiles_x, t2 = tile / a.tiles_x;
      const int ty0 = (t2 % a.tiles_y) * TH, tx0 = tx_i * TW, n = t2 / a.tiles_y;
      const bool full = (ty0 + TH <= a.H) && (tx0 + TW <= a.W);
      const int co = cob * NB + nt * 32 + li;
      const bool co_ok = co < a.Cout;
      const __amdgpu_buffer_rsrc_t rsrc_out = __builtin_amdgcn_make_buffer_rsrc(
          p_out + (size_t)n * a.H * a.W * a.out_cs, 0, (unsigned)(a.H * a.W * a.out_cs) * 4u, 0x00020000);
      // lane part of an output address (bytes): channel + the lh tile column (4 pixels); OOB for channels beyond Cout
      unsigned lane_o[4], lane_t[4] = {OOB, OOB, OOB, OOB};  // + pixel column x
#pragma unroll
      for (int x = 0; x < 4; ++x) lane_o[x] = co_ok ? (unsigned)(((4 * lh + x) * a.out_cs + a.out_co + co) * 4) : OOB;
      const int row_u = ty0 + 2 * qI, col_u = tx0;  // + compile-time offsets of (gq, e, pixel)
      __amdgpu_buffer_rsrc_t rsrc_t = rsrc_out;
      float bq0 = 0.f, bq1 = 0.f, bq2 = 0.f, bq3 = 0.f;
      if (IN_MODE == 0 && a.bnr_mode != 0) {
        rsrc_t = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p_bnr) + (size_t)n * a.H * a.W * a.bnr_cs, 0,
                                                   (unsigned)(a.H * a.W * a.bnr_cs) * 4u, 0x00020000);
#pragma unroll
        for (int x = 0; x < 4; ++x) lane_t[x] = co_ok ? (unsigned)(((4 * lh + x) * a.bnr_cs + a.bnr_co + co) * 4) : OOB;
        bq0 = sS[nt * 32 + li]; bq1 = sS[NB + nt * 32 + li]; bq2 = sS[2 * NB + nt * 32 + li]; bq3 = sS[3 * NB + nt * 32 + li];
      }
      const float sg = (IN_MODE != 0 && p_pool != nullptr) ? sG[nt * 32 + li] : 1.f;
      // raw pooled copy [N, H/2, W/2, Cout]: buffer stores like the output's (lane part: channel + the lh tile column's two
      // windows; pooled row / column of the quad in the scalar offset) - as plain pointer stores every one of the 32 stores of a
      // tile cost ~17 instructions of 64-bit address arithmetic and an exec-mask branch (round 6)
      __amdgpu_buffer_rsrc_t rsrc_pool = rsrc_out;
      unsigned lane_pl[2] = {OOB, OOB};
      if (IN_MODE != 0 && p_pool != nullptr) {
        const unsigned pimg = (unsigned)((a.H >> 1) * (a.W >> 1) * a.Cout) * 4u;
        rsrc_pool = __builtin_amdgcn_make_buffer_rsrc(p_pool + (size_t)n * (a.H >> 1) * (a.W >> 1) * a.Cout, 0, pimg, 0x00020000);
#pragma unroll
        for (int wnd = 0; wnd < 2; ++wnd) lane_pl[wnd] = co_ok ? (unsigned)(((2 * lh + wnd) * a.Cout + co) * 4) : OOB;
      }
      // row exchange [wave][8 values][lane][4], partner = the other row half (wave ^ 2); even register quads through the
      // consumed sA image, odd ones through sX: one barrier per quad
      float* const sAc = sA + buf * W4_A_FLOATS;
      mfma_results_guard();  // the output transform reads the accumulators from inline asm
      auto round = [&](auto GQ) {
        constexpr int gq = decltype(GQ)::value;
        constexpr int g_row = WIDE ? 0 : 16 * (gq >> 1), g_col = WIDE ? 8 * gq : 8 * (gq & 1);  // pixel offsets of the quad's tiles
        float* const xb = (gq & 1) ? sX : sAc;
        float* const xw = xb + wave * 2048 + lane * 4;
        const float* const xr = xb + (wave ^ 2) * 2048 + lane * 4;
#define W4_Q(C) ((C) < 16 ? w4_acc_quad<((C) < 16 ? (C) : 0) * 16 + 4 * gq>()                                    \
                  : (C) == 16 ? f32x4{acc16[4 * gq], acc16[4 * gq + 1], acc16[4 * gq + 2], acc16[4 * gq + 3]}  \
                              : f32x4{acc17[4 * gq], acc17[4 * gq + 1], acc17[4 * gq + 2], acc17[4 * gq + 3]})
        // R = M[I,:] A, one row of M at a time: R[.][0] = m0 + m1 + m2 + m3 + m4, [1] = m1 - m2 + 2 (m3 - m4), [2] = m1 + m2 +
        // 4 (m3 + m4), [3] = m1 - m2 + 8 (m3 - m4) + m5
#define W4_RR(R, O)                                                                                         \
        f32x4 O[4];                                                                                         \
        {                                                                                                   \
          const f32x4 m1 = W4_Q(6 * (R) + 1), m2 = W4_Q(6 * (R) + 2), m3 = W4_Q(6 * (R) + 3), m4 = W4_Q(6 * (R) + 4);  \
          const f32x4 s1 = pk4_add(m1, m2), d1 = pk4_sub(m1, m2), s2 = pk4_add(m3, m4), d2 = pk4_sub(m3, m4); \
          O[0] = pk4_add(pk4_add(W4_Q(6 * (R)), s1), s2);                                                   \
          O[1] = pk_fma_p24(d2, d1);                                                                        \
          O[2] = pk_fma_p44(s2, s1);                                                                        \
          O[3] = pk_fma_p44(pk4_add(d2, d2), pk4_add(d1, W4_Q(6 * (R) + 5)));                               \
        }
        // P = A^T[:,I] R, the same coefficient pattern along the rows, folded row by row (fewer live registers): I = 0 keeps
        // {r0 + r1 + r2, r1 - r2}, sends {r1 + r2, r1 - r2}; I = 1 sends {r0 + r1, 2 (r0 - r1)}, keeps {4 (r0 + r1), 8 (r0 -
        // r1) + r2}
        f32x4 kp[2][4];
        if (qI == 0) {
          W4_RR(1, o1)
          W4_RR(2, o2)
#pragma unroll
          for (int x = 0; x < 4; ++x) {
            const f32x4 s = pk4_add(o1[x], o2[x]), d = pk4_sub(o1[x], o2[x]);
            kp[0][x] = s; kp[1][x] = d;
            *reinterpret_cast<f32x4*>(xw + x * 256) = s;
            *reinterpret_cast<f32x4*>(xw + (4 + x) * 256) = d;
          }
          W4_RR(0, o0)
#pragma unroll
          for (int x = 0; x < 4; ++x) kp[0][x] = pk4_add(kp[0][x], o0[x]);
        } else {
          W4_RR(0, o0)
          W4_RR(1, o1)
#pragma unroll
          for (int x = 0; x < 4; ++x) {
            const f32x4 s = pk4_add(o0[x], o1[x]), d = pk4_sub(o0[x], o1[x]);
            const f32x4 d2 = pk4_add(d, d), s2 = pk4_add(s, s);
            *reinterpret_cast<f32x4*>(xw + x * 256) = s;
            *reinterpret_cast<f32x4*>(xw + (4 + x) * 256) = d2;
            kp[0][x] = pk4_add(s2, s2); kp[1][x] = d2;
          }
          W4_RR(2, o2)
#pragma unroll
          for (int x = 0; x < 4; ++x) kp[1][x] = pk_fma_p44(kp[1][x], o2[x]);
        }
#undef W4_RR
#undef W4_Q
        // scalar (wave-uniform) byte offset of pixel P = yy * 4 + x of tile row E of this quad, for a tensor of CS channels
        // (the pixel column rides in the lane offset, the pixel row in the scalar offset: two scalar registers per tensor
        // instead of 32 hoisted offsets per round)
        const unsigned so_o = (unsigned)(((row_u + g_row) * a.W + col_u + g_col) * a.out_cs) * 4u, sr_o = (unsigned)(a.W * a.out_cs) * 4u;
        const unsigned so_t = (unsigned)(((row_u + g_row) * a.W + col_u + g_col) * a.bnr_cs) * 4u, sr_t = (unsigned)(a.W * a.bnr_cs) * 4u;
#define W4_SOFF(E, P, T) (so_##T + (unsigned)(4 * (E) + ((P) >> 2)) * sr_##T)
        // lane offset of pixel column x, OOB where the pixel is outside the map (partial tile blocks only)
#define W4_VOFF(T, E, P) ((FULL || (4 * (E) + ((P) >> 2) < lim_y && ((P) & 3) < lim_x)) ? lane_##T[(P) & 3] : OOB)
        const int lim_y = a.H - row_u - g_row, lim_x = a.W - col_u - g_col - 4 * lh;
        // the second half of the round, straight-line per (bnr mode, full tile block): both are block-uniform
        auto fin = [&](auto MODE_, auto FULL_) {
          constexpr int MODE = decltype(MODE_)::value;
          constexpr bool FULL = decltype(FULL_)::value;
          // fused BatchNorm-backward sums: the layer-below tensor at this lane's 8 x 4 (pixel, tile) positions (first pixel
          // row before the barrier, second row while the first is finished)
          f32x4 tq[8];
#define W4_TQ(P0)                                                                                           \
          if (MODE != 0) {                                                                                  \
            _Pragma("unroll") for (int p = (P0); p < (P0) + 4; ++p)                                         \
            _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                   \
              tq[p][e] = (W4_ABL & 2048) ? 0.5f : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_t, W4_VOFF(t, e, p), W4_SOFF(e, p, t), W4_NT)); \
          }
          W4_TQ(0)
          if (!(W4_ABL & 32)) __syncthreads();
          W4_TQ(4)
#undef W4_TQ
          f32x4 pm[2];
#pragma unroll
          for (int p = 0; p < 8; ++p) {
            const f32x4 v = pk4_add(kp[p >> 2][p & 3], W4_LD(xr + p * 256));  // (the bias rides in component (1,1))
            f32x4 s1v = v, xv = v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const unsigned vo = W4_VOFF(o, e, p);
              if (MODE == 1) {
                const float t = tq[p][e];
                s1v[e] = fmaf(t, bq0, bq1) > 0.f ? v[e] : 0.f; xv[e] = fmaf(t, bq2, bq3);
              } else if (MODE == 2) {
                const float t = tq[p][e];
                s1v[e] = t > 0.f ? v[e] : 0.f; xv[e] = (t - bq0) * bq1;
              }
              if (!FULL) s1v[e] = vo != OOB ? s1v[e] : 0.f;
              const float ve = v[e];  // (__builtin_bit_cast of the element expression v[e] itself reads element 0)
              if (!(W4_ABL & 16)) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, ve), rsrc_out, vo, W4_SOFF(e, p, o), W4_NT);
            }
            stat_s1 = pk_add(pk_add(stat_s1, lo2(s1v)), hi2(s1v));
            stat_s2 = pk_fma(hi2(s1v), hi2(xv), pk_fma(lo2(s1v), lo2(xv), stat_s2));
            if (IN_MODE != 0 && p_pool != nullptr) {
              // pixels (row, 0..1) and (row, 2..3) of the 2x4 block are two pooling windows (ConvArgs::pool_out)
              const f32x4 sv = v * sg;
              const int wnd = (p >> 1) & 1;
              if (p < 4 && (p & 1) == 0) pm[wnd] = sv;
              else {
#pragma unroll
                for (int e = 0; e < 4; ++e) pm[wnd][e] = fmaxf(pm[wnd][e], sv[e]);
              }
            }
          }
          if (IN_MODE != 0 && p_pool != nullptr) {
            // pooled pixel (py, px) = ((row_u + g_row) / 2 + 2 e, (col_u + g_col) / 2 + 2 lh + wnd): row and quad column are scalar
            const unsigned so_p = (unsigned)((((row_u + g_row) >> 1) * (a.W >> 1) + ((col_u + g_col) >> 1)) * a.Cout) * 4u;
            const unsigned sr_p = (unsigned)((a.W >> 1) * a.Cout) * 8u;
#pragma unroll
            for (int wnd = 0; wnd < 2; ++wnd)
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const unsigned vo = (FULL || (4 * e < lim_y && 2 * wnd < lim_x)) ? lane_pl[wnd] : OOB;
                const float pv = pm[wnd][e] * sg;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, pv), rsrc_pool, vo, so_p + (unsigned)e * sr_p, 0);
              }
          }
        };
        using T_ = std::true_type;
        using F_ = std::false_type;
        const int mode = IN_MODE == 0 ? a.bnr_mode : 0;
        if (full) {
          if (mode == 0) fin(std::integral_constant<int, 0>{}, T_{});
          else if (mode == 1) fin(std::integral_constant<int, 1>{}, T_{});
          else fin(std::integral_constant<int, 2>{}, T_{});
        } else {
          if (mode == 0) fin(std::integral_constant<int, 0>{}, F_{});
          else if (mode == 1) fin(std::integral_constant<int, 1>{}, F_{});
          else fin(std::integral_constant<int, 2>{}, F_{});
        }
#undef W4_SOFF
#undef W4_VOFF
      };
      round(std::integral_constant<int, 0>{});
      round(std::integral_constant<int, 1>{});
      round(std::integral_constant<int, 2>{});
      round(std::integral_constant<int, 3>{});
      if (!(W4_ABL & 64)) w4_acc_clear(c11);
      W4_CLEAR_V()
#endif
      W4_T(4)
      chunk = 0;
      tile += per_cob;
      W4_FETCH_F(0, nA, 0)
    }
  }
#if W4_TRACE
  if (a.trace != nullptr && blockIdx.x == 0 && lane == 0) {
#pragma unroll
    for (int k = 0; k < 5; ++k) a.trace[wave * 8 + k] = tc_[k];
    a.trace[wave * 8 + 5] = __builtin_readcyclecounter() - tk0_;
    a.trace[wave * 8 + 6] = (unsigned long long)nstages;
  }
#endif
#undef W4_T
#undef W4_ISSUE_HALO
#undef W4_STAGE
#undef W4_CLEAR_V
#undef W4_HALO_PAR
#undef W4_HALO_BN
#undef W4_HALO_WR
#undef W4_TR_RD
#undef W4_TR_WR
#undef W4_TRANSFORM_ALL
#undef W4_HALF1
#undef W4_PSLN
#undef W4_SSLN
#undef W4_PSL
#undef W4_SSL
#undef W4_PC
#undef W4_SC
#undef W4_RP_A
#undef W4_RP_B
#undef W4_RP_C
#undef W4_RP_D
#undef W4_TR_RD3
#undef W4_TR_RDX
#undef W4_TR_ADDR
#undef W4_PCX
#undef W4_TR_RD3X
#undef W4_SCX
#undef W4_RPX_A
#undef W4_RPX_B
#undef W4_RPX_C
#undef W4_RPX_D
#undef W4_XF_REGS
#undef W4_FETCH
#undef W4_FETCH_F
#undef W4_FETCH_W
#undef W4_MMX
#undef W4_MM
#undef W4_MM8
#undef W4_FENCE

  if (p_stats != nullptr) {
    // the two lane halves hold the same channel (different tiles)
    float t1 = stat_s1[0] + stat_s1[1], t2 = stat_s2[0] + stat_s2[1];
    t1 += __shfl_xor(t1, 32);
    t2 += __shfl_xor(t2, 32);
    const int co = cob * NB + nt * 32 + li;
    if (lh == 0 && co < a.Cout) {
      double* q = p_stats + (size_t)(blockIdx.x % NREP) * 2 * a.Cout + co;
      // forward: statistics of the output; data gradient with a fused BatchNorm-backward reduction (bnr_mode != 0): S1, S2
      if (IN_MODE == 0 && a.bnr_mode != 0) { acc_add_grad(q, (double)t1); acc_add_grad(q + a.Cout, (double)t2); }
      else { acc_add_stats(q, (double)t1); acc_add_stats(q + a.Cout, (double)t2); }
    }
  }
}
#undef W4_LD

}  // namespace sspk
