// Winograd F(2x2, 3x3) convolution, second-generation pipelined kernel: TWO independent 4-wave workgroups per CU.
//
// conv_wino_pipe_kernel (one 8-wave workgroup per CU) keeps the matrix pipe busy 62-67 % of the cycles: its eight waves
// run in lock step, so at each of the two barriers per 8-channel stage and during the tile epilogue EVERY wave of the CU
// has stopped issuing MFMAs (profiles/r01_conv_cycle_trace.txt).  Here the same per-wave work (32 tiles x 32 output
// channels x 8 Winograd components = 128 accumulator registers, 32 MFMAs per 8-channel stage) is organised as
//
//   workgroup = 4 waves (one per SIMD) = 32 Winograd tiles (8x16 or 16x8 pixels) x 64 output channels,
//   wave = (component half, output-channel half); 78 KB of LDS -> two workgroups per CU,
//
// so each SIMD hosts two waves of DIFFERENT workgroups that share no barrier: while one sits at a barrier, stages its
// halo or runs its epilogue, the other one issues MFMAs.
//
// Operand roles are swapped against the first-generation kernels: A = weight fragment (32 output channels x 2 k), B =
// transformed input fragment (2 k x 32 tiles), so the accumulator of a lane holds ONE tile (column = lane & 31) and 16
// output channels (row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)): four consecutive channels sit in four consecutive
// registers, and the epilogue stores 16-byte quads straight from registers.  Epilogue per tile: output transform in
// registers; the two component halves of a channel half exchange ONE pixel row of partial sums through LDS (8 ds_write_b128
// + 8 ds_read_b128 per lane, one barrier) and each finishes one pixel row of the 2x2 outputs: 8 global 16-byte stores per
// lane; the BatchNorm sums (forward statistics, or pass 1 of the BatchNorm backward of the layer below: ConvArgs::bnr_mode)
// are reduce-scattered over the 32 tile lanes with 31 wave shuffles into ONE persistent register per lane.
//
// Stage pipeline (8 input channels per stage, transformed input double-buffered in LDS, ONE raw-halo buffer):
//
//   stage g, first half : MFMAs of component pairs 0, 1 on sA[g&1]  ||  transform raw(g+1): sR -> sA[~g&1]
//   barrier A                                        (sA[~g&1] complete; sR free)
//   stage g, second half: MFMAs of component pairs 2, 3             ||  halo (g+2): registers -> sR (BatchNorm + ReLU of
//                                                                       the producer), halo loads (g+3) issued
//   barrier B                                        (sR = raw(g+2) complete)
//
// Every MFMA operand is fetched ONE component pair (8 MFMAs = 512 cycles) ahead of its use into one of two register
// sets: weight fragments from L2 (the packed weight image [cob][chunk8][component][h][64][4] IS the fragment layout),
// input fragments from LDS - including the first pair of stage g+1, read from sA[~g&1] during the second half of stage g.
// A wave therefore never waits for an LDS or L2 round trip between two MFMA groups; the first-generation kernel exposed
// three LDS round trips per stage (tools/archive/ablate_p2.py: the non-MFMA work of a stage was 2700 cycles long and overlapped
// with the 4096 MFMA cycles of the two waves of a SIMD for 700 cycles only).
// The raw halo lives in LDS as [row][pixel][8 channels] with a row stride of 592 bytes (8x16 tiles; 320 for 16x8).  The
// transform's ds_read_b128 (pixel column 2 tx + j of rows 2 ty + r over the 32 tiles of a wave) is serviced in 16-lane groups
// of NON-contiguous lanes ({0-3, 12-15, 20-27}, ...: MI355X_MICROARCH.md): with plain base + j * 32 byte addressing the 8x16
// tiles are conflict-free, the 16x8 tiles (the 30x40 maps: every launch of the step) are 2-way conflicted whatever the row
// stride (a lane group spans four halo rows; `SQ_LDS_BANK_CONFLICT / SQ_INSTS_LDS` 1.37, profiles/r05_ssp_pmc_sq_summary.txt).
// P2_XOR (round 6): for the 16x8 tiles the pixel column of halo row r is XOR-ed with (r >> 1) & 1 (p2_raw_col; model and search:
// tools/lds_conflicts.py p2_transform): 8 -> 4 LDS cycles per read.  A thread's four columns 2 tx + j then sit at base +
// (j ^ f) * 32 bytes, i.e. even and odd j have their own base register.
#pragma once
#include "conv_wino_pipe.hip.h"

#ifndef P2_ABL
#define P2_ABL 0  // compile-time perf ablation (tools/archive/ablate_p2.py): 1 no epilogue, 2 no halo staging / transform, 4 no barriers
                  // in the stage loop, 8 no MFMA, 16 no weight loads in the stage loop, 32 no fragment reads from LDS
#endif

namespace sspk {

constexpr int P2_THREADS = 256;
constexpr int P2_TILES = 32;                         // Winograd tiles per workgroup
constexpr int P2_HALO = 180;                         // (8+2) x (16+2) = (16+2) x (8+2) raw halo pixels
constexpr int P2_A_FLOATS = WC * P2_TILES * PK;      // 4096 floats = 16 KB transformed input per buffer
constexpr int P2_R_FLOATS = 1480;                    // raw halo: 10 rows x 592 B (8x16 tiles) / 18 rows x 320 B (16x8)
constexpr int P2_X_FLOATS = 4 * 8 * 64 * 4;          // exchange buffer: [wave][reg group * 2 + pixel][lane][4] = 32 KB
constexpr int P2_S_FLOATS = 2048 + NB;               // BatchNorm scale | shift of <= 1024 input channels (or bnr params), bias
constexpr int P2_LDS_BYTES = (2 * P2_A_FLOATS + P2_R_FLOATS + P2_X_FLOATS + P2_S_FLOATS) * 4;  // 79744
static_assert(2 * P2_LDS_BYTES <= 160 * 1024, "two workgroups per CU");

// float offset of (component, tile, channel quad) in a transformed-input buffer: a tile's 8 channels are 32 contiguous
// bytes, the two quads swapped for tiles 8-15 and 24-31 so that the 16-lane groups of ds_read_b128 hit 16 distinct slots
__device__ __forceinline__ int p2_a_off(int comp, int tile, int quad) {
  return (comp * P2_TILES + tile) * PK + ((quad ^ ((tile >> 3) & 1)) << 2);
}

#ifndef P2_XOR
#define P2_XOR 1
#endif
// pixel column c of raw halo row r -> column slot in the LDS row (16x8 tiles only: the halo row has 10 pixels, c ^ 1 stays inside)
template <bool WIDE>
__device__ __forceinline__ int p2_raw_col(int r, int c) { return (WIDE || !P2_XOR) ? c : (c ^ ((r >> 1) & 1)); }

template <int IN_MODE, bool WIDE>
__global__ __launch_bounds__(P2_THREADS, 2) void conv_wino_p2_kernel(const ConvArgs a) {
  constexpr int TTX = WIDE ? 8 : 4;                  // tiles per workgroup row
  constexpr int TH = WIDE ? 8 : 16, TW = WIDE ? 16 : 8;
  constexpr int HC = TW + 2;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const sR = smem + 2 * P2_A_FLOATS;
  float* const sX = sR + P2_R_FLOATS;
  float* const sS = sX + P2_X_FLOATS;                // IN_MODE 1: scale[Cin] | shift[Cin]; bnr: 4 x 64 parameters
  float* const sBias = sS + 2048;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: the per-wave roles become uniform branches
  const int li = lane & 31, lh = lane >> 5;
  const int chalf = wave & 1, nt = wave >> 1;

  // ---- work assignment (as conv_mfma_kernel): XCD-aware persistent tile list ----
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
  const int nst = a.Cin / PK;                                  // stages per tile
  const int my_tiles = (t_end - tile0 + per_cob - 1) / per_cob;
  const int nstages = my_tiles * nst;

  // ---- staging roles ----
  const int q2 = tid & 1;
  // raw halo items tid + 256 k (k < 2), item = pixel * 2 + quad; LDS image [row][pixel][8] floats, row stride SROW
  constexpr int SROW = WIDE ? 148 : 80;
  int rrc[2], r_lds[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int p = (tid + P2_THREADS * k) >> 1, r = p / HC, c = p - r * HC;
    rrc[k] = r | (c << 8);
    r_lds[k] = r * SROW + p2_raw_col<WIDE>(r, c) * PK + q2 * 4;
  }
  const bool r1 = tid + P2_THREADS < P2_HALO * 2;  // the second item exists
  // transform: (quad, tile, V row); the V row is wave-uniform
  const int t_tile = (tid >> 1) & 31, t_row = wave;
  const int t_ty = t_tile / TTX, t_tx = t_tile % TTX;
  const int t_ra = t_row == 0 ? 0 : t_row == 2 ? 2 : 1;   // T[i] = d[ra] + sg d[rb]
  const int t_rb = t_row == 2 ? 1 : t_row == 3 ? 3 : 2;
  const float t_sg = t_row == 1 ? 1.f : -1.f;
  const int t_dst = p2_a_off(t_row * 4, t_tile, q2);
  // pixel column 2 tx + j of raw rows ra / rb: columns j = 0, 2 at t_?e + j * PK, columns j = 1, 3 at t_?o + (j - 1) * PK
  const int t_u = (2 * t_ty + t_ra) * SROW + 2 * t_tx * PK + q2 * 4;
  const int t_w = (2 * t_ty + t_rb) * SROW + 2 * t_tx * PK + q2 * 4;
  const int t_fu = p2_raw_col<WIDE>(2 * t_ty + t_ra, 0), t_fw = p2_raw_col<WIDE>(2 * t_ty + t_rb, 0);
  const int t_ue = t_u + t_fu * PK, t_uo = t_u + (t_fu ^ 1) * PK;
  const int t_we = t_w + t_fw * PK, t_wo = t_w + (t_fw ^ 1) * PK;
  const int pixb = a.in_cs * 4, rowb = a.W * pixb;
  f32x4 hreg[2];
  f32x4 psc = {1.f, 1.f, 1.f, 1.f}, psh = {0.f, 0.f, 0.f, 0.f};
  constexpr unsigned OOB = 0x80000000u;
  unsigned hoff[2] = {OOB, OOB};
  const size_t img_floats = (size_t)a.H * a.W * a.in_cs;
  __amdgpu_buffer_rsrc_t rsrc_in;
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wpk), 0, a.wpk_bytes, 0x00020000);

  // load cursor: the (tile, chunk) whose halo loads are issued next
  int ld_tile = tile0, ld_chunk = 0;
#define P2_ISSUE_HALO()                                                                                     \
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
    if (++ld_chunk == nst) { ld_chunk = 0; ld_tile += per_cob; }                                            \
  }
#define P2_WRITE_RAW()                                                                                      \
  {                                                                                                         \
    _Pragma("unroll") for (int k = 0; k < 2; ++k) {                                                         \
      if (k == 0 || r1) {                                                                                   \
        f32x4 v = hreg[k];                                                                                  \
        if (IN_MODE != 0) v = bn_relu_quad(v, psc, psh, hoff[k] == OOB);                                    \
        *reinterpret_cast<f32x4*>(sR + r_lds[k]) = v;                                                       \
      }                                                                                                     \
    }                                                                                                       \
  }
  // MFMA fragment offsets: input fragment (B operand) of this lane's tile / channel quad; weight fragment (A operand):
  // byte offset of this lane's quad inside a component's [h][64][4] block
  const int in_off = p2_a_off(chalf * 8, li, lh);
  const int w_voff = (lh * NB + nt * 32 + li) * 16;
  f32x4 wA0 = {0.f, 0.f, 0.f, 0.f}, wA1 = wA0, wB0 = wA0, wB1 = wA0;
#define P2_WLOAD(S0, S1, C, CHUNK)                                                                          \
  if (!(P2_ABL & 16) || g < 0) {                                                                            \
    const int so_ = ((cob * nst + (CHUNK)) * PB_FLOATS + (chalf * 8 + (C)) * 2 * NB * 4) * 4;               \
    S0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, w_voff, so_, 0));         \
    S1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, w_voff, so_ + 2 * NB * 16, 0)); \
  }

  // one V row (4 components) of (tile, quad): sR -> the transformed-input buffer DST
#define P2_TRANSFORM_READ()                                                                                 \
  const f32x4 u0 = *reinterpret_cast<const f32x4*>(sR + t_ue), w0 = *reinterpret_cast<const f32x4*>(sR + t_we);                   \
  const f32x4 u1 = *reinterpret_cast<const f32x4*>(sR + t_uo), w1 = *reinterpret_cast<const f32x4*>(sR + t_wo);                   \
  const f32x4 u2 = *reinterpret_cast<const f32x4*>(sR + t_ue + 2 * PK), w2 = *reinterpret_cast<const f32x4*>(sR + t_we + 2 * PK); \
  const f32x4 u3 = *reinterpret_cast<const f32x4*>(sR + t_uo + 2 * PK), w3 = *reinterpret_cast<const f32x4*>(sR + t_wo + 2 * PK);
#define P2_TRANSFORM_WRITE(DST)                                                                             \
  {                                                                                                         \
    const f32x4 t0 = u0 + t_sg * w0, t1 = u1 + t_sg * w1, t2 = u2 + t_sg * w2, t3 = u3 + t_sg * w3;         \
    float* d_ = (DST) + t_dst;                                                                              \
    *reinterpret_cast<f32x4*>(d_ + 0 * P2_TILES * PK) = t0 - t2;                                            \
    *reinterpret_cast<f32x4*>(d_ + 1 * P2_TILES * PK) = t1 + t2;                                            \
    *reinterpret_cast<f32x4*>(d_ + 2 * P2_TILES * PK) = t2 - t1;                                            \
    *reinterpret_cast<f32x4*>(d_ + 3 * P2_TILES * PK) = t1 - t3;                                            \
  }

  // ---- per-block parameters in LDS ----
  if (tid < NB) {
    const int co_ = cob * NB + tid;
    sBias[tid] = (a.bias != nullptr && co_ < a.Cout) ? a.bias[co_] : 0.f;
  }
  if (IN_MODE != 0) {
    if (a.lazy.mode == 0) {
      for (int c = tid; c < a.Cin; c += P2_THREADS) {
        sS[c] = p_scale[c];
        sS[1024 + c] = p_shift[c];
      }
    } else {   // (BnLazy: the producer's statistics -> affine here; workgroup 0 stores for the later readers)
      for (int c = tid; c < a.Cin; c += P2_THREADS) {
        float sc_, sh_;
        bn_lazy_affine(a.lazy, prob, c, sc_, sh_);
        sS[c] = sc_;
        sS[1024 + c] = sh_;
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
  int g = -1;  // (the ablation macros test g < 0 = prologue)
  P2_ISSUE_HALO()
  P2_WRITE_RAW()
  __syncthreads();
  {
    P2_TRANSFORM_READ()
    P2_TRANSFORM_WRITE(smem)
  }
  P2_ISSUE_HALO()
  __syncthreads();
  P2_WRITE_RAW()
  P2_ISSUE_HALO()
  __syncthreads();

  f32x16 acc[8];
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  float stat_acc = 0.f;  // lane li: which = li >> 4 (sum / weighted sum), accumulator register li & 15, summed over tiles

  // operand fetch: input fragments of component pair (C, C + 1) from the transformed-input buffer BUF, weight fragments
  // of the same pair of 8-channel chunk CHUNK from L2
  f32x4 fA0, fA1, fB0, fB1;
#define P2_FREAD(F0, F1, BUF, C)                                                                            \
  if (!(P2_ABL & 32) || g < 0) {                                                                            \
    F0 = *reinterpret_cast<const f32x4*>((BUF) + in_off + (C) * P2_TILES * PK);                             \
    F1 = *reinterpret_cast<const f32x4*>((BUF) + in_off + ((C) + 1) * P2_TILES * PK);                       \
  }
  // ONE MFMA: number I (0..7) of component pair (C, C + 1): component C + (I & 1), k pair I >> 1.  The stage body below
  // puts a small slice of the staging work behind every single MFMA (fenced with sched_barrier so that the compiler keeps
  // the order): a wave issues in order, so only instructions placed BETWEEN two of its MFMAs run in the shadow of the
  // first one (64 cycles of matrix pipe = room for ~10 VALU / LDS / VMEM issues); work placed between two GROUPS of
  // back-to-back MFMAs adds its issue time to the stage instead (tools/archive/ablate_p2.py: MFMA-only 0.59 ms + non-MFMA-only
  // 0.37 ms gave 0.86 ms with group-wise placement, also with two independent workgroups per CU).
#define P2_MM(I, C, W0, W1, F0, F1)                                                                         \
  if (!(P2_ABL & 8)) {                                                                                      \
    if (((I) & 1) == 0) acc[C] = __builtin_amdgcn_mfma_f32_32x32x2f32(W0[(I) >> 1], F0[(I) >> 1], acc[C], 0, 0, 0);           \
    else acc[(C) + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(W1[(I) >> 1], F1[(I) >> 1], acc[(C) + 1], 0, 0, 0);              \
  }                                                                                                         \
  __builtin_amdgcn_sched_barrier(0);
#define P2_FENCE() __builtin_amdgcn_sched_barrier(0)

  P2_FREAD(fA0, fA1, smem, 0)
  P2_WLOAD(wA0, wA1, 0, 0)
  if (P2_ABL & 32) { fB0 = fA0; fB1 = fA1; }
  if (P2_ABL & 16) { P2_WLOAD(wB0, wB1, 2, 0) }

  int tile = tile0, chunk = 0;
  for (g = 0; g < nstages; ++g) {
    const int buf = g & 1;
    const float* const cA = smem + buf * P2_A_FLOATS;
    float* const nA = smem + (buf ^ 1) * P2_A_FLOATS;
    // ---- first half: component pairs 0, 1 || transform of stage g+1: sR -> the other buffer ----
    {
      P2_FREAD(fB0, fB1, cA, 2)
      P2_WLOAD(wB0, wB1, 2, chunk)
      P2_FENCE();
#if !(P2_ABL & 2)
      P2_MM(0, 0, wA0, wA1, fA0, fA1)
      const f32x4 u0 = *reinterpret_cast<const f32x4*>(sR + t_ue), w0 = *reinterpret_cast<const f32x4*>(sR + t_we);
      P2_FENCE();
      P2_MM(1, 0, wA0, wA1, fA0, fA1)
      const f32x4 u1 = *reinterpret_cast<const f32x4*>(sR + t_uo), w1 = *reinterpret_cast<const f32x4*>(sR + t_wo);
      P2_FENCE();
      P2_MM(2, 0, wA0, wA1, fA0, fA1)
      const f32x4 u2 = *reinterpret_cast<const f32x4*>(sR + t_ue + 2 * PK), w2 = *reinterpret_cast<const f32x4*>(sR + t_we + 2 * PK);
      P2_FENCE();
      P2_MM(3, 0, wA0, wA1, fA0, fA1)
      const f32x4 u3 = *reinterpret_cast<const f32x4*>(sR + t_uo + 2 * PK), w3 = *reinterpret_cast<const f32x4*>(sR + t_wo + 2 * PK);
      P2_FENCE();
      P2_MM(4, 0, wA0, wA1, fA0, fA1)
      const f32x4 t0 = u0 + t_sg * w0;
      P2_FENCE();
      P2_MM(5, 0, wA0, wA1, fA0, fA1)
      const f32x4 t1 = u1 + t_sg * w1;
      P2_FENCE();
      P2_MM(6, 0, wA0, wA1, fA0, fA1)
      const f32x4 t2 = u2 + t_sg * w2;
      P2_FENCE();
      P2_MM(7, 0, wA0, wA1, fA0, fA1)
      const f32x4 t3 = u3 + t_sg * w3;
      P2_FENCE();
      P2_FREAD(fA0, fA1, cA, 4)
      P2_WLOAD(wA0, wA1, 4, chunk)
      P2_FENCE();
      float* const d_ = nA + t_dst;
      P2_MM(0, 2, wB0, wB1, fB0, fB1)
      *reinterpret_cast<f32x4*>(d_ + 0 * P2_TILES * PK) = t0 - t2;
      P2_FENCE();
      P2_MM(1, 2, wB0, wB1, fB0, fB1)
      *reinterpret_cast<f32x4*>(d_ + 1 * P2_TILES * PK) = t1 + t2;
      P2_FENCE();
      P2_MM(2, 2, wB0, wB1, fB0, fB1)
      *reinterpret_cast<f32x4*>(d_ + 2 * P2_TILES * PK) = t2 - t1;
      P2_FENCE();
      P2_MM(3, 2, wB0, wB1, fB0, fB1)
      *reinterpret_cast<f32x4*>(d_ + 3 * P2_TILES * PK) = t1 - t3;
      P2_FENCE();
#else
      P2_MM(0, 0, wA0, wA1, fA0, fA1) P2_MM(1, 0, wA0, wA1, fA0, fA1) P2_MM(2, 0, wA0, wA1, fA0, fA1) P2_MM(3, 0, wA0, wA1, fA0, fA1)
      P2_MM(4, 0, wA0, wA1, fA0, fA1) P2_MM(5, 0, wA0, wA1, fA0, fA1) P2_MM(6, 0, wA0, wA1, fA0, fA1) P2_MM(7, 0, wA0, wA1, fA0, fA1)
      P2_FREAD(fA0, fA1, cA, 4)
      P2_WLOAD(wA0, wA1, 4, chunk)
      P2_FENCE();
      P2_MM(0, 2, wB0, wB1, fB0, fB1) P2_MM(1, 2, wB0, wB1, fB0, fB1) P2_MM(2, 2, wB0, wB1, fB0, fB1) P2_MM(3, 2, wB0, wB1, fB0, fB1)
#endif
      P2_MM(4, 2, wB0, wB1, fB0, fB1)
      P2_MM(5, 2, wB0, wB1, fB0, fB1)
      P2_MM(6, 2, wB0, wB1, fB0, fB1)
      P2_MM(7, 2, wB0, wB1, fB0, fB1)
    }
    if (!(P2_ABL & 4)) __syncthreads();  // barrier A
    // ---- second half: component pairs 2, 3 || halo (g+2): registers -> sR, halo loads (g+3) ----
    {
      P2_FREAD(fB0, fB1, cA, 6)
      P2_WLOAD(wB0, wB1, 6, chunk)
      P2_FENCE();
      P2_MM(0, 4, wA0, wA1, fA0, fA1)
#if !(P2_ABL & 2)
      if (IN_MODE != 0) hreg[0] = bn_relu_quad(hreg[0], psc, psh, hoff[0] == OOB);
      P2_FENCE();
      P2_MM(1, 4, wA0, wA1, fA0, fA1)
      *reinterpret_cast<f32x4*>(sR + r_lds[0]) = hreg[0];
      P2_FENCE();
      P2_MM(2, 4, wA0, wA1, fA0, fA1)
      if (IN_MODE != 0) hreg[1] = bn_relu_quad(hreg[1], psc, psh, hoff[1] == OOB);
      P2_FENCE();
      P2_MM(3, 4, wA0, wA1, fA0, fA1)
      if (r1) *reinterpret_cast<f32x4*>(sR + r_lds[1]) = hreg[1];
      P2_FENCE();
      P2_MM(4, 4, wA0, wA1, fA0, fA1)
      P2_ISSUE_HALO()  // a full stage ahead of their use
      P2_FENCE();
#else
      P2_MM(1, 4, wA0, wA1, fA0, fA1) P2_MM(2, 4, wA0, wA1, fA0, fA1) P2_MM(3, 4, wA0, wA1, fA0, fA1) P2_MM(4, 4, wA0, wA1, fA0, fA1)
#endif
      P2_MM(5, 4, wA0, wA1, fA0, fA1)
      P2_MM(6, 4, wA0, wA1, fA0, fA1)
      P2_MM(7, 4, wA0, wA1, fA0, fA1)
      P2_FREAD(fA0, fA1, nA, 0)                                        // first pair of the next stage
      P2_WLOAD(wA0, wA1, 0, (chunk + 1 == nst ? 0 : chunk + 1))
      P2_FENCE();
      P2_MM(0, 6, wB0, wB1, fB0, fB1)
      P2_MM(1, 6, wB0, wB1, fB0, fB1)
      P2_MM(2, 6, wB0, wB1, fB0, fB1)
      P2_MM(3, 6, wB0, wB1, fB0, fB1)
      P2_MM(4, 6, wB0, wB1, fB0, fB1)
      P2_MM(5, 6, wB0, wB1, fB0, fB1)
      P2_MM(6, 6, wB0, wB1, fB0, fB1)
      P2_MM(7, 6, wB0, wB1, fB0, fB1)
    }
    if (!(P2_ABL & 4)) __syncthreads();  // barrier B

    if ((P2_ABL & 1) && ++chunk == nst) {  // ablation: keep the accumulators alive, skip the epilogue
      if (tid == 1023) p_out[0] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + acc[4][0] + acc[5][0] + acc[6][0] + acc[7][0];
      chunk = 0;
      tile += per_cob;
    }
    if (!(P2_ABL & 1) && ++chunk == nst) {
      // ---- tile epilogue (branch-free on the full-channel path: buffer loads / stores clip the out-of-image pixels) ----
      const int tx_i = tile % a.tiles_x, t2 = tile / a.tiles_x;
      const int ty0 = (t2 % a.tiles_y) * TH, tx0 = tx_i * TW, n = t2 / a.tiles_y;
      // this lane's tile; this wave finishes pixel row `chalf` of its 2x2 outputs (columns px = 0, 1)
      const int oy = ty0 + 2 * (li / TTX) + chalf, ox = tx0 + 2 * (li % TTX);
      const bool in0 = oy < a.H && ox < a.W, in1 = oy < a.H && ox + 1 < a.W;
      const int co_l = nt * 32 + 4 * lh;                 // + 8 g + e: local output channel of register 4 g + e
      const __amdgpu_buffer_rsrc_t rsrc_out = __builtin_amdgcn_make_buffer_rsrc(
          p_out + (size_t)n * a.H * a.W * a.out_cs, 0, (unsigned)(a.H * a.W * a.out_cs) * 4u, 0x00020000);
      const unsigned obase = (unsigned)(((oy * a.W + ox) * a.out_cs + a.out_co + cob * NB + co_l) * 4);
      const unsigned ooff[2] = {in0 ? obase : OOB, in1 ? obase + (unsigned)a.out_cs * 4u : OOB};
      const float pmask[2] = {in0 ? 1.f : 0.f, in1 ? 1.f : 0.f};
      // output transform Y = A^T M A on this wave's two rows of M: keep[px] = partial of the own pixel row, the other
      // row's partial goes to the partner wave (same channel half, other component half)
      f32x4 keep[2][4];
      float* const xw = sX + (wave * 8) * 256 + lane * 4;
      const float* const xr = sX + ((wave ^ 1) * 8) * 256 + lane * 4;
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        f32x4 s0, s1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * gq + e;
          float top[4], bot[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float ra = acc[j][r], rb = acc[4 + j][r];
            if (chalf == 0) { top[j] = ra + rb; bot[j] = rb; }          // rows 0, 1 of M
            else { top[j] = ra; bot[j] = -(ra + rb); }                  // rows 2, 3 of M
          }
          const float y00 = top[0] + top[1] + top[2], y01 = top[1] - top[2] - top[3];
          const float y10 = bot[0] + bot[1] + bot[2], y11 = bot[1] - bot[2] - bot[3];
          if (chalf == 0) { keep[0][gq][e] = y00; keep[1][gq][e] = y01; s0[e] = y10; s1[e] = y11; }
          else { keep[0][gq][e] = y10; keep[1][gq][e] = y11; s0[e] = y00; s1[e] = y01; }
        }
        *reinterpret_cast<f32x4*>(xw + (gq * 2 + 0) * 256) = s0;
        *reinterpret_cast<f32x4*>(xw + (gq * 2 + 1) * 256) = s1;
      }
#pragma unroll
      for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
      // fused BatchNorm-backward sums: the layer-below tensor at this lane's two pixels, issued once the accumulators
      // are dead (the latency overlaps the exchange barrier)
      f32x4 tq[2][2];  // two rounds in flight
      __amdgpu_buffer_rsrc_t rsrc_t = rsrc_out;
      unsigned toff[2] = {OOB, OOB};
#define P2_TQ_LOAD(GQ)                                                                                      \
      if (IN_MODE == 0 && a.bnr_mode != 0) {                                                                \
        tq[0][(GQ) & 1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_t, toff[0], (GQ) * 32, 0)); \
        tq[1][(GQ) & 1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_t, toff[1], (GQ) * 32, 0)); \
      }
      if (IN_MODE == 0 && a.bnr_mode != 0) {
        rsrc_t = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p_bnr) + (size_t)n * a.H * a.W * a.bnr_cs, 0, (unsigned)(a.H * a.W * a.bnr_cs) * 4u, 0x00020000);
        const unsigned tbase = (unsigned)(((oy * a.W + ox) * a.bnr_cs + a.bnr_co + cob * NB + co_l) * 4);
        toff[0] = in0 ? tbase : OOB; toff[1] = in1 ? tbase + (unsigned)a.bnr_cs * 4u : OOB;
      }
      P2_TQ_LOAD(0)
      __syncthreads();
      const bool tail = (cob + 1) * NB > a.Cout;  // block-uniform: channel quads that straddle Cout (operator tests only)
      float st[32];
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        if (IN_MODE == 0) __builtin_amdgcn_sched_barrier(0);  // (one round of the layer-below tensor ahead, not all four)
        if (gq < 3) P2_TQ_LOAD(gq + 1)
        if (IN_MODE == 0) __builtin_amdgcn_sched_barrier(0);
        const f32x4 bq = *reinterpret_cast<const f32x4*>(sBias + co_l + 8 * gq);
#pragma unroll
        for (int px = 0; px < 2; ++px) {
          const f32x4 v = keep[px][gq] + *reinterpret_cast<const f32x4*>(xr + (gq * 2 + px) * 256) + bq;
          f32x4 s1v, s2v;
          if (IN_MODE == 0 && a.bnr_mode != 0) {
            const f32x4 q0 = *reinterpret_cast<const f32x4*>(sS + co_l + 8 * gq), q1 = *reinterpret_cast<const f32x4*>(sS + NB + co_l + 8 * gq);
            const f32x4 t = tq[px][gq & 1];
            f32x4 dz, xh;
            if (a.bnr_mode == 1) {
              const f32x4 q2v = *reinterpret_cast<const f32x4*>(sS + 2 * NB + co_l + 8 * gq), q3 = *reinterpret_cast<const f32x4*>(sS + 3 * NB + co_l + 8 * gq);
              const f32x4 z = __builtin_elementwise_fma(t, q0, q1);
              xh = __builtin_elementwise_fma(t, q2v, q3);
#pragma unroll
              for (int e = 0; e < 4; ++e) dz[e] = z[e] > 0.f ? v[e] : 0.f;
            } else {
              xh = (t - q0) * q1;
#pragma unroll
              for (int e = 0; e < 4; ++e) dz[e] = t[e] > 0.f ? v[e] : 0.f;
            }
            s1v = dz * pmask[px]; s2v = s1v * xh;
          } else {
            s1v = v * pmask[px]; s2v = s1v * v;
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (px == 0) { st[4 * gq + e] = s1v[e]; st[16 + 4 * gq + e] = s2v[e]; }
            else { st[4 * gq + e] += s1v[e]; st[16 + 4 * gq + e] += s2v[e]; }
          }
          if (!tail) {
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), rsrc_out,
                                                   ooff[px], gq * 32, 0);
          } else {
            const int nvalid = a.Cout - (cob * NB + co_l + 8 * gq);
            if (ooff[px] != OOB && nvalid > 0) {
              float* p = p_out + (size_t)n * a.H * a.W * a.out_cs + (ooff[px] >> 2) + 8 * gq;
              p[0] = v[0];
              if (nvalid > 1) p[1] = v[1];
              if (nvalid > 2) p[2] = v[2];
              if (nvalid > 3) p[3] = v[3];
            }
          }
        }
      }
      if (p_stats != nullptr) {
        // reduce-scatter over the 32 tile lanes (never across lh): 16 + 8 + 4 + 2 + 1 shuffles; afterwards lane li holds the
        // sum over the tiles of value li (values 0..15: sums of registers 0..15, 16..31: the weighted sums)
#pragma unroll
        for (int w = 16; w >= 1; w >>= 1) {
          const bool up = (li & w) != 0;
#pragma unroll
          for (int i = 0; i < w; ++i) {
            const float snd = up ? st[i] : st[i + w];
            const float kp = up ? st[i + w] : st[i];
            st[i] = kp + __shfl_xor(snd, w);
          }
        }
        stat_acc += st[0];
      }
      chunk = 0;
      tile += per_cob;
    }
  }
#undef P2_ISSUE_HALO
#undef P2_WRITE_RAW
#undef P2_WLOAD
#undef P2_TRANSFORM_READ
#undef P2_TRANSFORM_WRITE
#undef P2_FREAD
#undef P2_MM
#undef P2_FENCE

  if (p_stats != nullptr) {
    const int which = li >> 4, r = li & 15;
    const int co = cob * NB + nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
    if (co < a.Cout)
      acc_add_stats_or_grad(p_stats + (size_t)(blockIdx.x % NREP) * 2 * a.Cout + which * a.Cout + co, (double)stat_acc, IN_MODE == 0 && a.bnr_mode != 0);
  }
}

}  // namespace sspk
