// Winograd weight gradient F(3x3, 4x4) - the transpose of conv_wino4_kernel's F(4x4, 3x3):
//   dW(3x3) = G^T [ sum over 4x4-output tiles (B^T d B) (.) (A dy A^T) ] G      d: 6x6 input patch, dy: 4x4 dY tile
// 36 multiplies per 16 pixels and (ci, co) = 1/4 of the direct 144 (F(3x3,2x2), wgrad_wino_kernel: 16/36).  The 36
// component products are GEMMs over K = tiles, [64 ci x 2 tiles] x [2 tiles x 32 co] per v_mfma_f32_32x32x2_f32.
//
//   workgroup = 4 waves (ONE per SIMD, 512 registers each) = a 64 ci x 32 co slab of all 36 components;
//   wave (qa, qb) = the 3x3 block of components (3 qa .. 3 qa + 2, 3 qb .. 3 qb + 2) x 2 M-tiles (ci = 2 m + e) = 18
//   accumulators: 16 in FIXED accumulation registers a[0:255] behind inline asm (conv_wino4.hip.h explains why: hipcc
//   spills whole accumulators beyond 16), 2 in vector registers.
//
// LDS holds only RAW data of an 8-tile block (4x32 or 16x8 dY pixels): the input halo [pixel][64 ci] (BatchNorm + ReLU of
// the producer applied on the way in) and the dY tile [pixel][32 co]; both transforms run in registers between the LDS
// reads and the MFMAs, per K step of two tiles (lane half lh = tile of the pair):
//   A operand: V[i][j] of (tile lh, channels 2 li, 2 li + 1): a wave needs only rows 3 qa .. and columns 3 qb .. of V =
//   B^T d B, i.e. a 5x5 sub-patch of d (25 ds_read_b64), 30 + 18 packed VALU instructions;
//   B operand: D[i][j] = (A dy A^T)[i][j] of (tile lh, channel li): the dY image is [row][column pair][32 co][2], so 8
//   ds_read_b64 bring the tile as column pairs and the row stage runs packed: 8 + 9 VALU instructions.
// The LDS reads of K step s + 1 are issued before the MFMAs of step s (one wave per SIMD hides no latency by itself).
// The fp32 MFMA executes on the vector ALUs (every VALU instruction is matrix-pipe time): per K step 18
// MFMAs (1152 cycles) stand against ~76 VALU + 41 LDS instructions - F(3x3,2x2) pays 13 + 14 per 8 MFMAs, but executes
// 1.78x as many MFMAs per pixel.
//
// Epilogue: each wave applies its part of G^T . G to its 9 components (dW partial of 9 taps), the four waves are summed
// through LDS, and the block writes ONE [9][64][32] slab (1/4 of the F(3x3,2x2) kernel's 16-component slab per (ci, co));
// wgrad_wino_reduce_multi_kernel sums the splits into the OIHW gradient.
#pragma once
#include "conv_wino4.hip.h"

#ifndef WG4_ABL
#define WG4_ABL 0  // compile-time perf ablation (tools/archive/ablate_wgrad4.sh): 1 no MFMAs, 2 no transforms (raw values as operands),
                   // 4 no LDS reads in the K steps, 8 no staging (BatchNorm + LDS writes), 16 no global loads
#endif

namespace sspk {

template <bool WIDE>
struct Wgrad4Geom {
  static constexpr int TH = WIDE ? 4 : 16, TW = WIDE ? 32 : 8;   // 128 dY pixels = 8 Winograd tiles of 4x4 per block
  static constexpr int HT = TH + 2, WT = TW + 2;                  // input halo 6x34 | 18x10
  static constexpr int NPX = HT * WT;
  static constexpr int X_FLOATS = NPX * 64, D_FLOATS = TH * TW * 32;
  static constexpr int S_FLOATS = 9 * 64 * 32;                    // the block's output slab (epilogue)
  static constexpr int NX = (NPX + 15) / 16;                      // halo items per thread: pixel (tid >> 4) + 16 i, quad tid & 15
  static constexpr int O_WORDS = 2 * (NX + TH * TW / 64) * 256;   // per-thread staging offsets + slot coordinates, parked in LDS
  static constexpr int LDS_FLOATS = (X_FLOATS + D_FLOATS + 256 + O_WORDS) > S_FLOATS ? (X_FLOATS + D_FLOATS + 256 + O_WORDS) : S_FLOATS;
  static constexpr int LDS_BYTES = LDS_FLOATS * 4;
  static constexpr int ND = TH * TW / 64;                         // dY items per thread: pixel PAIR (tid >> 3) + 32 i, quad tid & 7
};
constexpr int WG4_THREADS = 256;
constexpr int WG4_SLAB = 9 * 64 * 32;  // floats per partial slab

// accumulator A (0..15) += V (.) D in fixed accumulation registers; 16 / 17 in vector registers
template <int A>
__device__ __forceinline__ void wg4_mfma_a(float v, float d) {
  asm volatile("v_mfma_f32_32x32x2_f32 a[%2:%3], %0, %1, a[%2:%3]" : : "v"(v), "v"(d), "n"(A * 16), "n"(A * 16 + 15));
}
__device__ __forceinline__ void wg4_mfma_v(f32x16& acc, float v, float d) {
  asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(v), "v"(d));
}
template <int R = 0>
__device__ __forceinline__ void wg4_acc_clear() {
  asm volatile("v_accvgpr_write_b32 a[%0], 0" : : "n"(R));
  if constexpr (R + 1 < 256) wg4_acc_clear<R + 1>();
}

// one 5-vector (rows or columns QX .. QX + 4 of a 6-vector d) -> the three entries 3 QX .. 3 QX + 2 of B^T d:
//   QX = 0: (4 d0 - 5 d2 + d4,  -4 d1 - 4 d2 + d3 + d4,  4 d1 - 4 d2 - d3 + d4)        z = d0 .. d4
//   QX = 1: (-2 d1 - d2 + 2 d3 + d4,  2 d1 - d2 - 2 d3 + d4,  4 d1 - 5 d3 + d5)         z = d1 .. d5
template <int QX>
__device__ __forceinline__ void wg4_bt3(const f32x2 (&z)[5], f32x2 km5, f32x2& o0, f32x2& o1, f32x2& o2) {
  if (QX == 0) {
    o0 = pk_fma_p4(z[0], pk_fma_k(km5, z[2], z[4]));
    const f32x2 p = pk_fma_m4(z[2], z[4]), q = pk_fma_m4(z[1], z[3]);
    o1 = pk_add(p, q);
    o2 = pk_sub(p, q);
  } else {
    const f32x2 p = pk_sub(z[3], z[1]), q = pk_sub(z[2], z[0]);
    o0 = pk_fma_p2(q, p);
    o1 = pk_fma_m2(q, p);
    o2 = pk_fma_p4(z[0], pk_fma_k(km5, z[2], z[4]));
  }
}
// rows of A = (A^T)^T (6x4): (1,0,0,0) (1,1,1,1) (1,-1,1,-1) (1,2,4,8) (1,-2,4,-8) (0,0,0,1).
// Row stage, packed over a column pair: y0..y3 = the four rows of the pair -> entries 3 QX .. 3 QX + 2 of A y
template <int QX>
__device__ __forceinline__ void wg4_a3_rows(f32x2 y0, f32x2 y1, f32x2 y2, f32x2 y3, f32x2& o0, f32x2& o1, f32x2& o2) {
  if (QX == 0) {
    const f32x2 s02 = pk_add(y0, y2), s13 = pk_add(y1, y3);
    o0 = y0;
    o1 = pk_add(s02, s13);
    o2 = pk_sub(s02, s13);
  } else {
    const f32x2 u = pk_fma_p4(y2, y0), v = pk_fma_p4(y3, y1);
    o0 = pk_fma_p2(v, u);
    o1 = pk_fma_m2(v, u);
    o2 = y3;
  }
}
// Column stage on one row held as two column pairs a = (t0, t1), b = (t2, t3) -> entries 3 QX .. 3 QX + 2 of (t A^T)
template <int QX>
__device__ __forceinline__ void wg4_a3_cols(f32x2 a, f32x2 b, float& o0, float& o1, float& o2) {
  if (QX == 0) {
    const f32x2 s = pk_add(a, b);  // (t0 + t2, t1 + t3)
    o0 = a[0];
    o1 = s[0] + s[1];
    o2 = s[0] - s[1];
  } else {
    const f32x2 u = pk_fma_p4(b, a);  // (t0 + 4 t2, t1 + 4 t3)
    o0 = fmaf(2.f, u[1], u[0]);
    o1 = fmaf(-2.f, u[1], u[0]);
    o2 = b[1];
  }
}

// The four K steps (8 tiles) of one staged block for the wave (QA, QB).  xl: lane base into the raw halo (rows QA.., columns
// QB.., lane half -> tile, channel pair 2 li); dl: lane base into the dY image (lane half -> tile, channel li).
// Software pipeline: the raw operands of K step s + 1 are read from LDS BEFORE the 18 MFMAs of step s are issued and land
// under them (a wave's own MFMAs are all that can cover an LDS round trip here); the transforms of step s + 1 follow the
// MFMAs - their VALU instructions would cost matrix-pipe time wherever they stood.
template <int QA, int QB, bool WIDE>
__device__ __forceinline__ void wgrad4_block(f32x16& acc16, f32x16& acc17, const float* __restrict__ xl,
                                             const float* __restrict__ dl) {
  using G = Wgrad4Geom<WIDE>;
  const f32x2 km5 = {-5.f, -5.f};
  f32x2 zx[5][5];  // raw input sub-patch [column][row]
  f32x2 yd[2][4];  // raw dY tile [column pair][row]
#if WG4_ABL & 4
  {
    f32x2 o_;
    asm volatile("v_mov_b32 %0, 1.0\n\tv_mov_b32 %1, 1.0" : "=v"(o_[0]), "=v"(o_[1]));
#pragma unroll
    for (int c = 0; c < 5; ++c)
#pragma unroll
      for (int r = 0; r < 5; ++r) zx[c][r] = o_;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) yd[c][r] = o_;
  }
#endif
  // raw reads of K step S in two parts: A = the dY tile and input columns 0..2 (issued BEFORE the MFMAs of step S - 1, in
  // flight under them: 46 registers), B = input columns 3, 4 (issued after those MFMAs; they land while the transforms of
  // step S work through part A).  Reading everything ahead needs 66 registers across the MFMA block and spilled.
#define WG4_READ_X(S, C0, C1)                                                                               \
  {                                                                                                         \
    _Pragma("unroll") for (int c = (C0); c < (C1); ++c)                                                     \
    _Pragma("unroll") for (int r = 0; r < 5; ++r) {                                                         \
      const int pp = WIDE ? r * G::WT + 8 * (S) + c : (4 * (S) + r) * G::WT + c;                            \
      if (!(WG4_ABL & 4)) zx[c][r] = *reinterpret_cast<const f32x2*>(xl + pp * 64);                         \
    }                                                                                                       \
  }
#define WG4_READ_A(S)                                                                                       \
  {                                                                                                         \
    _Pragma("unroll") for (int cp = 0; cp < 2; ++cp)                                                        \
    _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                         \
      const int pq = WIDE ? r * (G::TW / 2) + 4 * (S) + cp : (4 * (S) + r) * (G::TW / 2) + cp;              \
      if (!(WG4_ABL & 4)) yd[cp][r] = *reinterpret_cast<const f32x2*>(dl + pq * 64);                        \
    }                                                                                                       \
    WG4_READ_X(S, 0, 3)                                                                                     \
  }
  WG4_READ_A(0)
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    WG4_READ_X(s, 3, 5)
    __builtin_amdgcn_sched_barrier(0);
    // ---- B operand: D[i'][j'] = (A dy A^T)[3 QA + i'][3 QB + j'] of this lane's (tile, output channel) ----
    f32x2 ty[3][2];
    float D[3][3];
#if WG4_ABL & 2
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) D[i][j] = yd[j & 1][i][j >> 1];
#else
#pragma unroll
    for (int cp = 0; cp < 2; ++cp) wg4_a3_rows<QA>(yd[cp][0], yd[cp][1], yd[cp][2], yd[cp][3], ty[0][cp], ty[1][cp], ty[2][cp]);
#pragma unroll
    for (int i = 0; i < 3; ++i) wg4_a3_cols<QB>(ty[i][0], ty[i][1], D[i][0], D[i][1], D[i][2]);
#endif
    // ---- A operand: V[i'][j'] = (B^T d B)[3 QA + i'][3 QB + j'] of (tile, channels 2 li, 2 li + 1) ----
    f32x2 T[3][5];
    f32x2 V[3][3];
#if WG4_ABL & 2
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) V[i][j] = zx[i + j][(i * 2 + j) % 5];
#else
#pragma unroll
    for (int c = 0; c < 5; ++c) wg4_bt3<QA>(zx[c], km5, T[0][c], T[1][c], T[2][c]);
#pragma unroll
    for (int i = 0; i < 3; ++i) wg4_bt3<QB>(T[i], km5, V[i][0], V[i][1], V[i][2]);
#endif
    // ---- part A of the next step, then 18 MFMAs: component k = 3 i' + j', M-tile e.  hipcc inserts no wait states for
    // operands of inline asm: every operand is complete before the fence and the first MFMA issues >= 2 instructions
    // after the last VALU write ----
    __builtin_amdgcn_sched_barrier(0);
    if (s == 0) WG4_READ_A(1)
    if (s == 1) WG4_READ_A(2)
    if (s == 2) WG4_READ_A(3)
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 1");
#if WG4_ABL & 1
#define WG4_MM(K) asm volatile("" : : "v"(V[(K) / 3][(K) % 3]), "v"(D[(K) / 3][(K) % 3]));
    WG4_MM(0) WG4_MM(1) WG4_MM(2) WG4_MM(3) WG4_MM(4) WG4_MM(5) WG4_MM(6) WG4_MM(7) WG4_MM(8)
#undef WG4_MM
#else
#define WG4_MM(K)                                                                       \
    wg4_mfma_a<2 * (K)>(V[(K) / 3][(K) % 3][0], D[(K) / 3][(K) % 3]);                   \
    wg4_mfma_a<2 * (K) + 1>(V[(K) / 3][(K) % 3][1], D[(K) / 3][(K) % 3]);
    WG4_MM(0) WG4_MM(1) WG4_MM(2) WG4_MM(3) WG4_MM(4) WG4_MM(5) WG4_MM(6) WG4_MM(7)
#undef WG4_MM
    wg4_mfma_v(acc16, V[2][2][0], D[2][2]);
    wg4_mfma_v(acc17, V[2][2][1], D[2][2]);
#endif
    __builtin_amdgcn_sched_barrier(0);
  }
#undef WG4_READ_A
#undef WG4_READ_X
}


template <int IN_MODE, bool WIDE>
__global__ __launch_bounds__(WG4_THREADS) void wgrad_wino4_kernel(const WgradArgs a) {
  using G = Wgrad4Geom<WIDE>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sX = smem;
  float* sD = smem + G::X_FLOATS;
  float* sS = smem + G::X_FLOATS + G::D_FLOATS;  // producer scale | shift of the 64 input channels, per problem
  unsigned* const sO = reinterpret_cast<unsigned*>(sS + 256) + threadIdx.x;  // this thread's staging offsets [item][256]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int qa = wave >> 1, qb = wave & 1;

  int bid = blockIdx.x;
  const int split = bid % a.nsplit;
  bid /= a.nsplit;
  const int cob = bid % a.ncob;   // 32-channel output block
  const int cib = bid / a.ncob;   // 64-channel input block
  const int tot_tiles = a.ntiles * a.nprob;
  const int per = (tot_tiles + a.nsplit - 1) / a.nsplit;
  const int t_begin = split * per, t_end = min(tot_tiles, t_begin + per);

  w4_claim_agprs();
  wg4_acc_clear();
  f32x16 acc16, acc17;
  {
    float z_;
    asm volatile("v_mov_b32 %0, 0" : "=v"(z_));  // (an opaque zero: see conv_wino4_kernel)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc16[r] = z_; acc17[r] = z_; }
  }

  const int q16 = tid & 15, q8 = tid & 7;
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
  const int co0 = cob * 32 + q8 * 4;
  const bool covalid = co0 < a.Cout;

  f32x4 xreg[G::NX], dreg[G::ND][2];  // dY: the two pixels (even, odd column) of a pair
  unsigned xmask = 0;
  const int xpix = a.in_cs * 4, xrow = a.W * xpix, dpix = a.dout_cs * 4, drow = a.W * dpix;
  const int xq = civalid ? (a.in_co + ci0) * 4 : -1, dq = covalid ? (a.dout_co + co0) * 4 : -1;
  constexpr unsigned OOB = 0x80000000u;
  // byte offsets of this thread's staging slots relative to the first halo / dY pixel of a block (interior blocks add the
  // block origin as the scalar offset of the buffer load), OOB for slots past the raster / channel quads outside the
  // tensor.  They live in LDS, not in registers: 15 registers across the whole kernel were what hipcc spilled to scratch
  // once the raw operands of the next K step stayed in registers across the MFMAs.
  unsigned xmask_in = 0;
#pragma unroll
  for (int i = 0; i < G::NX; ++i) {
    const int pp = (tid >> 4) + 16 * i, r = pp / G::WT, c = pp - r * G::WT;
    const bool ok = xq >= 0 && pp < G::NPX;
    sO[i * 256] = ok ? (unsigned)(r * xrow + c * xpix + xq) : OOB;
    sO[(G::NX + G::ND + i) * 256] = ok ? (unsigned)((r << 16) | c) : 0xFFFFFFFFu;  // border blocks: slot coordinates (row << 16 | column)
    xmask_in |= (ok ? 1u : 0u) << i;
  }
#pragma unroll
  for (int i = 0; i < G::ND; ++i) {
    const int pq = (tid >> 3) + 32 * i, r = pq / (G::TW / 2), c = 2 * (pq - r * (G::TW / 2));  // even column of the pair
    sO[(G::NX + i) * 256] = dq >= 0 ? (unsigned)(r * drow + c * dpix + dq) : OOB;
    sO[(2 * G::NX + G::ND + i) * 256] = dq >= 0 ? (unsigned)((r << 16) | c) : 0xFFFFFFFFu;
  }
  bool nxt_inside = false;  // wave-uniform: the prefetched block is an interior block (no per-slot address arithmetic)
#define WG4_ISSUE(TILE)                                                                                       \
  {                                                                                                           \
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
      _Pragma("unroll") for (int i = 0; i < G::NX; ++i)                                                       \
        xreg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx_, sO[i * 256], xb_, 0)); \
      _Pragma("unroll") for (int i = 0; i < G::ND; ++i) {                                                     \
        const unsigned do_ = sO[(G::NX + i) * 256];                                                           \
        dreg[i][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rd_, do_, db_, 0));      \
        dreg[i][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rd_, do_, db_ + dpix, 0)); \
      }                                                                                                       \
      xmask = xmask_in;                                                                                       \
    } else {                                                                                                  \
      /* border block: per-slot bounds from the slot coordinates parked in LDS (as loop-invariant registers they cost 30+ */ \
      /* registers across the whole kernel: hipcc hoisted every pp / WT, pp % WT out of the block loop)                  */ \
      xmask = 0;                                                                                              \
      _Pragma("unroll") for (int i = 0; i < G::NX; ++i) {                                                     \
        const unsigned rc_ = sO[(G::NX + G::ND + i) * 256];                                                   \
        const int gy = ty0_ - 1 + (int)(rc_ >> 16), gx = tx0_ - 1 + (int)(rc_ & 0xFFFFu);                     \
        const bool ok = rc_ != 0xFFFFFFFFu && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;   \
        xreg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(                            \
            rx_, ok ? (unsigned)(gy * xrow + gx * xpix + xq) : OOB, 0, 0));                                   \
        xmask |= (ok ? 1u : 0u) << i;                                                                         \
      }                                                                                                       \
      _Pragma("unroll") for (int i = 0; i < G::ND; ++i) {                                                     \
        const unsigned rc_ = sO[(2 * G::NX + G::ND + i) * 256];                                               \
        const int gy = ty0_ + (int)(rc_ >> 16), gx = tx0_ + (int)(rc_ & 0xFFFFu);                             \
        _Pragma("unroll") for (int e = 0; e < 2; ++e) {                                                       \
          const bool ok = rc_ != 0xFFFFFFFFu && gy < a.H && gx + e < a.W;                                     \
          dreg[i][e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(                       \
              rd_, ok ? (unsigned)(gy * drow + (gx + e) * dpix + dq) : OOB, 0, 0));                           \
        }                                                                                                     \
      }                                                                                                       \
    }                                                                                                         \
  }

  // lane bases of the fragment reads (floats): the wave's 5x5 sub-patch starts at halo row qa, column qb; tile of the pair
  // = lane half (4 pixel columns = 2 column pairs further); channel pair 2 li resp. output channel li.
  // dY image: [row][column pair][32 co][2 columns] (a lane's ds_read_b64 = its channel at both columns of a pair)
  const float* const xl = sX + ((qa * G::WT + qb + 4 * lh) * 64 + 2 * li);
  const float* const dl = sD + (128 * lh + 2 * li);

  if (t_begin < t_end) WG4_ISSUE(t_begin)
  for (int tile = t_begin; tile < t_end; ++tile) {
    __syncthreads();  // all waves finished reading the previous block's LDS image
    if (!(WG4_ABL & 8)) {
      const int cur_prob = tile >= a.ntiles ? 1 : 0;
      f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
      if (IN_MODE != 0) {
        sc = *reinterpret_cast<const f32x4*>(sS + cur_prob * 128 + q16 * 4);
        sh = *reinterpret_cast<const f32x4*>(sS + cur_prob * 128 + 64 + q16 * 4);
      }
#pragma unroll
      for (int i = 0; i < G::NX; ++i) {
        const int pp = (tid >> 4) + 16 * i;
        if (pp < G::NPX) {
          f32x4 v = xreg[i];  // 0 from the buffer load where the pixel / channel quad is outside
          if (IN_MODE != 0) v = bn_relu_quad(v, sc, sh, !((xmask >> i) & 1u));
          *reinterpret_cast<f32x4*>(sX + pp * 64 + q16 * 4) = v;
        }
      }
#pragma unroll
      for (int i = 0; i < G::ND; ++i) {
        const int pq = (tid >> 3) + 32 * i;
        float* q = sD + (pq * 32 + q8 * 4) * 2;  // 4 channels x (even, odd column): 0 where outside
#pragma unroll
        for (int e = 0; e < 4; ++e) *reinterpret_cast<f32x2*>(q + 2 * e) = f32x2{dreg[i][0][e], dreg[i][1][e]};
      }
    }
    __syncthreads();
    {
      const int nxt = min(tile + 1, t_end - 1);  // unconditional prefetch (redundant on the last block)
      if (!(WG4_ABL & 16)) WG4_ISSUE(nxt)
      __builtin_amdgcn_sched_barrier(0);
    }
    // wave roles are scalar: four straight-line instances of the K-step code
    if (wave == 0) wgrad4_block<0, 0, WIDE>(acc16, acc17, xl, dl);
    else if (wave == 1) wgrad4_block<0, 1, WIDE>(acc16, acc17, xl, dl);
    else if (wave == 2) wgrad4_block<1, 0, WIDE>(acc16, acc17, xl, dl);
    else wgrad4_block<1, 1, WIDE>(acc16, acc17, xl, dl);
  }
#undef WG4_ISSUE

  // ---- epilogue: P[u][v] = sum_{i', j'} G[3 qa + i'][u] G[3 qb + j'][v] M[i'][j'] per (ci, co), summed over the four waves ----
  // G of F(4x4,3x3): rows (1/4,0,0) (-1/6,-1/6,-1/6) (-1/6,1/6,-1/6) (1/24,1/12,1/6) (1/24,-1/12,1/6) (0,0,1)
  float ga[3][3], gb[3][3];
  {
    const float g0[3][3] = {{0.25f, 0.f, 0.f}, {-1.f / 6.f, -1.f / 6.f, -1.f / 6.f}, {-1.f / 6.f, 1.f / 6.f, -1.f / 6.f}};
    const float g1[3][3] = {{1.f / 24.f, 1.f / 12.f, 1.f / 6.f}, {1.f / 24.f, -1.f / 12.f, 1.f / 6.f}, {0.f, 0.f, 1.f}};
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        ga[i][u] = qa ? g1[i][u] : g0[i][u];
        gb[i][u] = qb ? g1[i][u] : g0[i][u];
      }
  }
  mfma_results_guard();  // the accumulators are read from inline asm
  float* const slab = smem;  // [9][64 ci][32 co]
  auto emit = [&](auto E_, auto R_, bool first) {
    constexpr int e = decltype(E_)::value, r = decltype(R_)::value;
    float m[3][3];
#define WG4_RD(K) m[(K) / 3][(K) % 3] = w4_acc_rd<(2 * (K) + e) * 16 + r>();
    WG4_RD(0) WG4_RD(1) WG4_RD(2) WG4_RD(3) WG4_RD(4) WG4_RD(5) WG4_RD(6) WG4_RD(7)
#undef WG4_RD
    m[2][2] = e ? acc17[r] : acc16[r];
    float rr[3][3];  // R[u][j'] = sum_i' ga[i'][u] M[i'][j']
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
      for (int j = 0; j < 3; ++j) rr[u][j] = fmaf(ga[2][u], m[2][j], fmaf(ga[1][u], m[1][j], ga[0][u] * m[0][j]));
    const int ci = 2 * ((r & 3) + 8 * (r >> 2) + 4 * lh) + e;
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
      for (int v = 0; v < 3; ++v) {
        const float p = fmaf(rr[u][2], gb[2][v], fmaf(rr[u][1], gb[1][v], rr[u][0] * gb[0][v]));
        float* q = slab + ((u * 3 + v) * 64 + ci) * 32 + li;
        *q = first ? p : *q + p;
      }
  };
  auto emit_all = [&](bool first) {
#define WG4_E(R) emit(std::integral_constant<int, 0>{}, std::integral_constant<int, R>{}, first); \
                 emit(std::integral_constant<int, 1>{}, std::integral_constant<int, R>{}, first);
    WG4_E(0) WG4_E(1) WG4_E(2) WG4_E(3) WG4_E(4) WG4_E(5) WG4_E(6) WG4_E(7)
    WG4_E(8) WG4_E(9) WG4_E(10) WG4_E(11) WG4_E(12) WG4_E(13) WG4_E(14) WG4_E(15)
#undef WG4_E
  };
  __syncthreads();  // the raw images are dead
  // one wave after the other adds its part (same lane -> address mapping in every wave: no atomics)
#pragma nounroll
  for (int w = 0; w < 4; ++w) {
    if (wave == w) emit_all(w == 0);
    __syncthreads();
  }
  float* dst = a.partial + (size_t)blockIdx.x * WG4_SLAB;
#pragma unroll
  for (int k = 0; k < WG4_SLAB / 4 / WG4_THREADS; ++k)
    *reinterpret_cast<f32x4*>(dst + (tid + WG4_THREADS * k) * 4) = *reinterpret_cast<const f32x4*>(slab + (tid + WG4_THREADS * k) * 4);
}

}  // namespace sspk
