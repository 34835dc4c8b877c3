// wgrad_wino_kernel (Winograd F(3x3,2x2) weight gradient, conv_wino.hip.h) with the APPLY pass of the layer's own
// BatchNorm + ReLU + MaxPool2d(2) backward fused into its dY staging:
//
//   dY(px) = gamma invstd ( dZ(px) - S1/n - xhat(px) S2/n ),   dZ(px) = dOut(window of px) if px is the FIRST maximum of
//   relu(z) in its 2x2 window and z(px) > 0, else 0            (bn_bwd_kernel<true, true, true>, torch's max_pool2d routing)
//
// Each thread owns one pooling window x 4 output channels of the 128-pixel dY tile: it loads the window's four raw conv
// outputs y (the same four 16-byte loads the plain kernel spends on dY itself) + ONE pooled gradient quad, evaluates the
// formula, writes the four dY quads into the LDS tile the MFMA phase reads AND to HBM for the data-gradient convolution of
// the layer (blocks of the first input-channel block only).  The separate APPLY pass - y read, pooled dOut read, dY write
// at HBM speed: 0.55 ms per step for the 240x320 layer alone - disappears; the weight gradient is MFMA-bound (2 TB/s of
// HBM traffic), so the extra quarter-size read and the dY write ride for free; what it pays is ~110 VALU instructions per
// tile and thread beside 128 MFMAs per wave.
//
// POOL = false: the same for a layer followed by BatchNorm + ReLU only (dZ = dOut [z > 0]): a thread's four pixel slots of the
// plain kernel load y and dOut (8 quads instead of 4) and write dY.
//
// Registers: the plain kernel sits at 256 with 2 spills.  The per-thread staging offsets and halo slot coordinates are
// PARKED IN LDS here (hipcc otherwise hoists every pp / WT, pp % WT out of the tile loop: wgrad_wino4.hip.h).
#pragma once
#include "conv_wino.hip.h"
#include <type_traits>
#ifndef SSP_LEGACY_ALGOS
#define SSP_LEGACY_ALGOS 0
#endif
#if SSP_LEGACY_ALGOS
#include "conv_wino_bf16.hip.h"   // (the bf16-operand MFMA phase of the retired mixed mode 8: BF16 = true below)
#endif

#ifndef WGF_TRACE
#define WGF_TRACE 0  // compile-time perf trace (never in the shipped library): s_memtime stamps around the phases of a tile, summed per
                   // wave of workgroup 0 into WgradArgs::trace[wave * 8 + k]: k = 0 wait at the top barrier, 1 staging (halo LDS
                   // writes + the fused APPLY + dY stores), 2 wait at the second barrier, 3 issue of the next tile's loads, 4 MFMA
                   // phase, 5 whole loop, 6 tiles, 7 the loop in 100 MHz ticks (-> the shader clock the launch ran at); tools/dbg/wgf_trace.sh
#endif

namespace sspk {

template <bool WIDE>
struct WgradFusedGeom {
  using G = WgradWinoGeom<WIDE>;
  static constexpr int NX = (G::HT * G::WT + 31) / 32;  // halo pixels per thread (pp = (tid >> 4) + 32 i)
  static constexpr int F_FLOATS = 2 * 10 * 64;          // per problem: scale, shift, -mean invstd, invstd, gamma invstd, A, B | f_lazy: S1, S2, bias term
  static constexpr int O_WORDS = (2 * NX + 9) * 512;    // parked per thread: halo offsets, halo coordinates, dY-side offsets / coordinates, mask
  static constexpr int LDS_BYTES = (G::X_FLOATS + G::D_FLOATS + 256 + F_FLOATS + O_WORDS) * 4;
};

// BF16: the MFMA phase of wgrad_wino_bf16_kernel<.., NT = 1> (transformed operands rounded to bf16 in registers, one
// v_mfma_f32_32x32x8_bf16 per 8 tiles; the mixed bf16 mode 8) on the same fp32 LDS tile images.
template <int IN_MODE, bool WIDE, bool POOL, bool BF16 = false>
__global__ __launch_bounds__(512) void wgrad_wino_fused_kernel(const WgradArgs a) {
  static_assert(!BF16 || SSP_LEGACY_ALGOS, "the bf16-operand form belongs to the retired conv algorithms (-DSSP_LEGACY_ALGOS=1)");
  using G = WgradWinoGeom<WIDE>;
  using GF = WgradFusedGeom<WIDE>;
  constexpr int NX = GF::NX;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sX = smem;
  float* sD = smem + G::X_FLOATS;
  float* sS = smem + G::X_FLOATS + G::D_FLOATS;
  float* sF = sS + 256;
  unsigned* const sO = reinterpret_cast<unsigned*>(sF + GF::F_FLOATS) + threadIdx.x;  // [item][512]

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
  // BatchNorm-backward parameters of this block's 64 output channels, per problem: dY = gs dZ + (A xhat + B),
  // xhat = y invstd + cc;  gs = gamma invstd, A = -gs S2/n, B = -gs S1/n, cc = -mean invstd
  if (tid >= 64 && tid < 64 + 128) {
    const int pr = (tid - 64) >> 6, ch = (tid - 64) & 63, co = cob * 64 + ch;
    float v[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float lz[3] = {0.f, 0.f, 0.f};   // f_lazy: (float)S1, (float)S2, the bias-gradient term of this view
    if (co < a.Cout && pr < a.nprob) {
      const float is = a.f_invstd[pr][co], gs = a.f_gamma[co] * is;
      v[0] = a.f_scale[pr][co]; v[1] = a.f_shift[pr][co]; v[2] = -a.f_mean[pr][co] * is; v[3] = is; v[4] = gs;
      float k1, k2;
      if (!POOL && a.f_lazy) {   // the replica reduction of bn_bwd_sums_kernel, here
        const double* __restrict__ q = a.f_bsums[pr];
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {   // two rounds of 16 replicas: 32 loads in flight
          double u1[NREP / 2], u2[NREP / 2];
#pragma unroll
          for (int k = 0; k < NREP / 2; ++k) {
            u1[k] = q[(size_t)(hh * (NREP / 2) + k) * 2 * a.Cout + co];
            u2[k] = q[(size_t)(hh * (NREP / 2) + k) * 2 * a.Cout + a.Cout + co];
          }
#pragma unroll
          for (int k = 0; k < NREP / 2; ++k) { s1 += u1[k]; s2 += u2[k]; }
        }
        k1 = (float)(s1 / a.f_count); k2 = (float)(s2 / a.f_count);
        lz[0] = (float)s1; lz[1] = (float)s2;
        lz[2] = a.f_gamma[co] * is * (float)(s1 - a.f_count * (double)k1);   // (bn_bwd_sums_kernel: the conv bias gradient, rounding noise)
      } else {
        k1 = a.f_k12[pr][co]; k2 = a.f_k12[pr][a.Cout + co];
      }
      v[5] = -gs * k2; v[6] = -gs * k1;
    }
#pragma unroll
    for (int k = 0; k < 7; ++k) sF[(pr * 7 + k) * 64 + ch] = v[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) sF[2 * 7 * 64 + (pr * 3 + k) * 64 + ch] = lz[k];
  }
  const int co0 = cob * 64 + q16 * 4;
  const bool covalid = co0 < a.Cout;

  constexpr int NP = POOL ? 1 : 4;  // gradient quads per thread: the window's pooled one / one per pixel slot
  f32x4 xreg[NX], yreg[4], preg[NP];
  unsigned xmask = 0;
  unsigned wvalid = 0;  // POOL: the prefetched window lies inside the map (bit 0); else: validity bits of the four pixel slots

  const int xpix = a.in_cs * 4, xrow = a.W * xpix;
  const int ypix = a.f_ycs * 4, yrow = a.W * ypix;       // y and dY share the geometry [N,H,W,f_ycs], channel offset 0
  const int ppix = a.dout_cs * 4, prow = (POOL ? (a.W >> 1) : a.W) * ppix;  // gradient wrt the (pooled) activation
  const int xq = civalid ? (a.in_co + ci0) * 4 : -1;
  constexpr unsigned OOB = 0x80000000u;
  // this thread's pooling window of the dY tile: window w = tid >> 4 (32 windows of 2x2 pixels), channel quad q16
  const int w_ = tid >> 4;
  const int wy = WIDE ? (w_ >> 4) : (w_ >> 2), wx = WIDE ? (w_ & 15) : (w_ & 3);
  {
    unsigned xmask_in = 0;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int pp = (tid >> 4) + 32 * i, r = pp / G::WT, c = pp - r * G::WT;
      const bool ok = xq >= 0 && pp < G::HT * G::WT;
      sO[i * 512] = ok ? (unsigned)(r * xrow + c * xpix + xq) : OOB;
      sO[(NX + i) * 512] = ok ? (unsigned)((r << 16) | c) : 0xFFFFFFFFu;
      xmask_in |= (ok ? 1u : 0u) << i;
    }
    if (POOL) {
      sO[(2 * NX) * 512] = covalid ? (unsigned)(2 * wy * yrow + 2 * wx * ypix + co0 * 4) : OOB;               // y / dY window
      sO[(2 * NX + 1) * 512] = covalid ? (unsigned)(wy * prow + wx * ppix + (a.dout_co + co0) * 4) : OOB;     // pooled gradient
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {  // pixel slot i = pixel (tid >> 4) + 32 i of the dY raster: y / dY offset, coordinates
        const int pp = (tid >> 4) + 32 * i, r = pp / G::TW, c = pp - r * G::TW;
        sO[(2 * NX + i) * 512] = covalid ? (unsigned)(r * yrow + c * ypix + co0 * 4) : OOB;
        sO[(2 * NX + 4 + i) * 512] = covalid ? (unsigned)((r << 16) | c) : 0xFFFFFFFFu;
      }
    }
    sO[(2 * NX + 8) * 512] = xmask_in;
  }
  bool nxt_inside = false;  // wave-uniform: the prefetched tile is an interior tile
  // Tile walker (wave-uniform scalars; every per-tile instruction of this loop is issue time: two waves per SIMD share one issue
  // port and the CU's eight waves ONE scalar unit, and no MFMA runs while they stage / issue): the coordinates of the tile whose
  // loads are in flight advance incrementally (no divisions per tile), the three buffer descriptors are rebuilt only when the
  // image changes, and the staging below reuses them for the dY stores of the tile it consumes.
  int w_tile = t_begin, w_prob = 0, w_n = 0, w_ty0 = 0, w_tx0 = 0;
  {
    w_prob = t_begin >= a.ntiles ? 1 : 0;
    const int tl_ = t_begin - w_prob * a.ntiles;
    const int tx_ = tl_ % a.tiles_x, t2_ = tl_ / a.tiles_x;
    w_ty0 = (t2_ % a.tiles_y) * G::TH; w_tx0 = tx_ * G::TW; w_n = t2_ / a.tiles_y;
  }
  const int w_xend = a.tiles_x * G::TW, w_yend = a.tiles_y * G::TH;
  auto walk = [&]() __attribute__((always_inline)) {   // -> the next tile (the caller stops at t_end - 1)
    ++w_tile;
    w_tx0 += G::TW;
    if (w_tx0 >= w_xend) {
      w_tx0 = 0; w_ty0 += G::TH;
      if (w_ty0 >= w_yend) {
        w_ty0 = 0; ++w_n;
        if (w_tile == a.ntiles) { w_n = 0; w_prob = 1; }
      }
    }
  };
  __amdgpu_buffer_rsrc_t rx_, ry_, rp_;
  int r_key = -1;   // (problem, image) the descriptors were built for
  // Offsets of the next tile's loads (WGF_PREP) - the loads themselves are ISSUED IN SLICES BETWEEN THE MFMA STEPS of the current
  // tile (wgf_load_slice, round 6).  As one block behind the staging barrier they cost 3100 (older waves of a SIMD) to 6900 cycles
  // (younger ones) per tile, with ~60 instructions on the interior path: eight waves push 96 KB of 16-byte loads into the CU's
  // memory pipeline at once and every wave stalls on the full queue before it can start its MFMA phase
  // (profiles/r05_wgf_phase_trace.txt: "issue"); spread over the first steps of a 12-16 k cycle MFMA phase they never queue.
  unsigned xvo[NX], yo[4] = {0u, 0u, 0u, 0u}, po1 = 0u;
  int xso = 0, yb = 0, pb = 0;
#define WGF_PREP()                                                                                            \
  {                                                                                                           \
    const int pr_ = w_prob, n_ = w_n;                                                                         \
    const int ty0_ = __builtin_amdgcn_readfirstlane(w_ty0), tx0_ = __builtin_amdgcn_readfirstlane(w_tx0);     \
    if (r_key != pr_ * 65536 + n_) {                                                                          \
      r_key = pr_ * 65536 + n_;                                                                               \
      rx_ = __builtin_amdgcn_make_buffer_rsrc(                                                                \
          const_cast<float*>(pr_ ? a.in2 : a.in) + (size_t)n_ * a.H * a.W * a.in_cs, 0, a.H * xrow, 0x00020000); \
      ry_ = __builtin_amdgcn_make_buffer_rsrc(                                                                \
          const_cast<float*>(a.f_y[pr_]) + (size_t)n_ * a.H * a.W * a.f_ycs, 0, a.H * yrow, 0x00020000);      \
      rp_ = __builtin_amdgcn_make_buffer_rsrc(                                                                \
          const_cast<float*>(pr_ ? a.dout2 : a.dout) + (size_t)n_ * (POOL ? (a.H >> 1) * (a.W >> 1) : a.H * a.W) * a.dout_cs, 0, \
          (POOL ? (a.H >> 1) : a.H) * prow, 0x00020000);                                                      \
    }                                                                                                         \
    nxt_inside = ty0_ >= 1 && ty0_ + G::TH + 1 <= a.H && tx0_ >= 1 && tx0_ + G::TW + 1 <= a.W;               \
    /* (scalar offsets through readfirstlane: mixed into the per-lane validity tests below they became vector values and every   */ \
    /* y load a waterfall loop)                                                                                                */ \
    yb = __builtin_amdgcn_readfirstlane(ty0_ * yrow + tx0_ * ypix);                                           \
    pb = __builtin_amdgcn_readfirstlane(POOL ? (ty0_ >> 1) * prow + (tx0_ >> 1) * ppix : ty0_ * prow + tx0_ * ppix + a.dout_co * 4); \
    /* ONE set of load instructions for interior and border tiles (the branches only pick the offsets)                         */ \
    xso = 0;                                                                                                  \
    if (nxt_inside) {                                                                                         \
      xso = __builtin_amdgcn_readfirstlane((ty0_ - 1) * xrow + (tx0_ - 1) * xpix);                            \
      _Pragma("unroll") for (int i = 0; i < NX; ++i) xvo[i] = sO[i * 512];                                    \
      xmask = sO[(2 * NX + 8) * 512];                                                                         \
    } else {                                                                                                  \
      xmask = 0;                                                                                              \
      _Pragma("unroll") for (int i = 0; i < NX; ++i) {                                                        \
        const unsigned rc_ = sO[(NX + i) * 512];                                                              \
        const int gy = ty0_ - 1 + (int)(rc_ >> 16), gx = tx0_ - 1 + (int)(rc_ & 0xFFFFu);                     \
        const bool ok = rc_ != 0xFFFFFFFFu && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;   \
        xvo[i] = ok ? (unsigned)(gy * xrow + gx * xpix + xq) : OOB;                                           \
        xmask |= (ok ? 1u : 0u) << i;                                                                         \
      }                                                                                                       \
    }                                                                                                         \
    if (POOL) {                                                                                               \
      unsigned yo_ = sO[(2 * NX) * 512], po_ = sO[(2 * NX + 1) * 512];                                        \
      /* H and W are even: a window is inside the map or outside as a whole */                               \
      wvalid = (yo_ != OOB && (nxt_inside || (ty0_ + 2 * wy < a.H && tx0_ + 2 * wx < a.W))) ? 1u : 0u;       \
      if (!wvalid) { yo_ = OOB; po_ = OOB; }                                                                  \
      yo[0] = yo_; po1 = po_;                                                                                 \
    } else {                                                                                                  \
      wvalid = 0;                                                                                             \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                         \
        unsigned yo_ = sO[(2 * NX + i) * 512];                                                                \
        if (!nxt_inside) {                                                                                    \
          const unsigned rc_ = sO[(2 * NX + 4 + i) * 512];                                                    \
          if (!(rc_ != 0xFFFFFFFFu && ty0_ + (int)(rc_ >> 16) < a.H && tx0_ + (int)(rc_ & 0xFFFFu) < a.W)) yo_ = OOB; \
        }                                                                                                     \
        wvalid |= (yo_ != OOB ? 1u : 0u) << i;                                                                \
        yo[i] = yo_;                                                                                          \
      }                                                                                                       \
    }                                                                                                         \
  }
  // load number IDX of the prepared tile: the X halo quads, then the four y quads, then the gradient quad(s)
  auto wgf_load = [&](auto IDX_) __attribute__((always_inline)) {
    constexpr int IDX = decltype(IDX_)::value;
    if constexpr (IDX < NX) {
      xreg[IDX] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx_, xvo[IDX], xso, 0));
    } else if constexpr (IDX < NX + 4) {
      constexpr int k = IDX - NX;
      if (POOL) yreg[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ry_, yo[0], yb + (k >> 1) * yrow + (k & 1) * ypix, 0));
      else yreg[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ry_, yo[k], yb, 0));
    } else if constexpr (IDX < NX + 4 + NP) {
      constexpr int k = IDX - NX - 4;
      // y and the gradient share the pixel geometry; their channel strides may differ only by the scalar part
      if (POOL) preg[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rp_, po1, pb, 0));
      else preg[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rp_, yo[k], pb, 0));
    }
  };
  constexpr int NLOAD = NX + 4 + NP;
  // the two loads behind MFMA step S (all issued within the first (NLOAD + 1) / 2 <= 8 of the 16 steps)
  bool w_more = false;  // the tile in the MFMA phase is not this block's last one
  auto wgf_load_slice = [&](auto S_) __attribute__((always_inline)) {
    constexpr int S = decltype(S_)::value;
    // step 0 also moves the walker and prepares the offsets: as a phase of its own between the staging barrier and the MFMA
    // phase that cost 1.0 (interior tiles) to 2.3 k cycles (border tiles) of LDS / scalar latency per tile with no MFMA in flight
    if constexpr (S == 0) {
      if (!BF16) {
        if (w_more) walk();  // unconditional prefetch (the last tile loads itself again)
        WGF_PREP()
      }
    }
    if constexpr (2 * S < NLOAD) wgf_load(std::integral_constant<int, (2 * S < NLOAD ? 2 * S : 0)>{});
    if constexpr (2 * S + 1 < NLOAD) wgf_load(std::integral_constant<int, (2 * S + 1 < NLOAD ? 2 * S + 1 : 0)>{});
  };
#define WGF_LOAD_ALL()                                                                                        \
  {                                                                                                           \
    wgf_load_slice(std::integral_constant<int, 0>{}); wgf_load_slice(std::integral_constant<int, 1>{});       \
    wgf_load_slice(std::integral_constant<int, 2>{}); wgf_load_slice(std::integral_constant<int, 3>{});       \
    wgf_load_slice(std::integral_constant<int, 4>{}); wgf_load_slice(std::integral_constant<int, 5>{});       \
    wgf_load_slice(std::integral_constant<int, 6>{}); wgf_load_slice(std::integral_constant<int, 7>{});       \
  }
  static_assert(NLOAD <= 16, "eight load slices");

  // per-wave constants of component row i:  T[i][c] = d[ra][c] + sg d[rb][c];  R[q] = c0 dy[0][q] + c1 dy[1][q]
  const int ra = irow == 0 ? 0 : irow == 2 ? 2 : 1;
  const int rb = irow == 2 ? 1 : irow == 3 ? 3 : 2;
  const float sg = irow == 1 ? 1.f : -1.f;
  const float c0 = irow == 3 ? 0.f : 1.f;
  const float c1 = irow == 0 ? 0.f : irow == 1 ? 1.f : -1.f;

  __syncthreads();  // sF / sS
  if (!POOL && a.f_lazy && cib == 0 && split == 0 && tid < 64) {   // one workgroup per output-channel block: dgamma, dbeta, bias gradient
    const int co = cob * 64 + tid;
    if (co < a.Cout)
      for (int pr = 0; pr < a.nprob; ++pr) {   // view 0 then view 1, like bn_bwd_sums_kernel
        a.f_dbeta[co] += sF[2 * 7 * 64 + (pr * 3 + 0) * 64 + tid];
        a.f_dgamma[co] += sF[2 * 7 * 64 + (pr * 3 + 1) * 64 + tid];
        if (a.f_dbias != nullptr) a.f_dbias[co] += sF[2 * 7 * 64 + (pr * 3 + 2) * 64 + tid];
      }
  }
  if (t_begin < t_end) {  // the first tile: slice 0 prepares it (w_more = false: the walker stays)
    if (BF16) WGF_PREP()
    WGF_LOAD_ALL()
  }
#if WGF_TRACE
  unsigned long long tc_[5] = {0, 0, 0, 0, 0}, tp_ = __builtin_readcyclecounter();
  const unsigned long long tk0_ = tp_, tr0_ = __builtin_amdgcn_s_memrealtime();   // (s_memrealtime: the constant 100 MHz counter)
#define WGF_T(K) { const unsigned long long tn_ = __builtin_readcyclecounter(); tc_[K] += tn_ - tp_; tp_ = tn_; }
#else
#define WGF_T(K)
#endif
  for (int tile = t_begin; tile < t_end; ++tile) {
    __syncthreads();  // all waves finished reading the previous tile's LDS image
    WGF_T(0)
    // the walker stands on this tile (its loads were issued one iteration ago); the dY stores below use its coordinates and the
    // y descriptor's image (ry_ belongs to the same image: dY shares y's geometry)
    const int cur_prob = w_prob, cur_n = w_n, cur_ty0 = w_ty0, cur_tx0 = w_tx0;
    {
      f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
      if (IN_MODE != 0) {
        sc = *reinterpret_cast<const f32x4*>(sS + cur_prob * 128 + q16 * 4);
        sh = *reinterpret_cast<const f32x4*>(sS + cur_prob * 128 + 64 + q16 * 4);
      }
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        const int pp = (tid >> 4) + 32 * i;
        if (pp < G::HT * G::WT) {
          f32x4 v = xreg[i];  // 0 from the buffer load where the pixel / channel quad is outside
          if (IN_MODE != 0) v = bn_relu_quad(v, sc, sh, !((xmask >> i) & 1u));
          *reinterpret_cast<f32x4*>(sX + pp * 64 + q16 * 4) = v;
        }
      }
      // ---- the layer's BatchNorm + ReLU + MaxPool backward APPLY on this thread's window ----
      const float* pf = sF + cur_prob * 7 * 64 + q16 * 4;
      const f32x4 fsc = *reinterpret_cast<const f32x4*>(pf), fsh = *reinterpret_cast<const f32x4*>(pf + 64);
      const f32x4 fcc = *reinterpret_cast<const f32x4*>(pf + 128), fis = *reinterpret_cast<const f32x4*>(pf + 192);
      const f32x4 fgs = *reinterpret_cast<const f32x4*>(pf + 256), fA = *reinterpret_cast<const f32x4*>(pf + 320);
      const f32x4 fB = *reinterpret_cast<const f32x4*>(pf + 384);
      f32x4 o[4];
      if (POOL) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          // first maximum of relu(z) in window scan order (torch: strictly greater replaces)
          float z[4], best = -1.f;
          int bk = 0;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            z[k] = fmaf(yreg[k][e], fsc[e], fsh[e]);
            const float av = fmaxf(z[k], 0.f);
            if (av > best) { best = av; bk = k; }
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float dz = (k == bk && z[k] > 0.f) ? preg[0][e] : 0.f;
            const float xh = fmaf(yreg[k][e], fis[e], fcc[e]);
            const float v = fmaf(fgs[e], dz, fmaf(fA[e], xh, fB[e]));
            o[k][e] = wvalid ? v : 0.f;  // 0 outside the map / the tensor, like a zero-filled dY load
          }
        }
        // tile image for the MFMA phase
#pragma unroll
        for (int k = 0; k < 4; ++k)
          *reinterpret_cast<f32x4*>(sD + ((2 * wy + (k >> 1)) * G::TW + 2 * wx + (k & 1)) * 64 + q16 * 4) = o[k];
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float z = fmaf(yreg[k][e], fsc[e], fsh[e]);
            const float dz = z > 0.f ? preg[POOL ? 0 : k][e] : 0.f;
            const float xh = fmaf(yreg[k][e], fis[e], fcc[e]);
            const float v = fmaf(fgs[e], dz, fmaf(fA[e], xh, fB[e]));
            o[k][e] = ((wvalid >> k) & 1u) ? v : 0.f;
          }
          *reinterpret_cast<f32x4*>(sD + ((tid >> 4) + 32 * k) * 64 + q16 * 4) = o[k];
        }
      }
      // dY for the data-gradient convolution of this layer (every tile exactly once: the blocks of input-channel block 0)
      if (cib == 0) {
        const int ty0 = cur_ty0, tx0 = cur_tx0, n = cur_n;
        const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(
            a.f_dy[cur_prob] + (size_t)n * a.H * a.W * a.f_ycs, 0, a.H * yrow, 0x00020000);
        const int yb = ty0 * yrow + tx0 * ypix;
        if (POOL) {
          const unsigned yo = wvalid ? sO[(2 * NX) * 512] : OOB;  // stores beyond the descriptor are dropped
#pragma unroll
          for (int k = 0; k < 4; ++k)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, o[k]),
                                                   rdy, yo, yb + (k >> 1) * yrow + (k & 1) * ypix, 0);
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, o[k]),
                                                   rdy, ((wvalid >> k) & 1u) ? sO[(2 * NX + k) * 512] : OOB, yb, 0);
        }
      }
    }
    WGF_T(1)
    __syncthreads();
    WGF_T(2)
    {
      // (sliced between the MFMA steps instead - tile state behind step 0, the loads behind steps 1 - 7 - the MFMA phase grew by more
      // than this phase is long: 27.1 k instead of 25.0 k cycles per tile, profiles/r05_wgf_phase_trace.txt)
      w_more = tile + 1 < t_end;
      if (BF16) {
        if (w_more) walk();
        WGF_PREP()
        WGF_LOAD_ALL()
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    WGF_T(3)
    {
      // lane bases: channel pair 2 li of the raw rows ra / rb, tile column offset of the lane half; dY column of this lane
      const float* xa0 = sX + (ra * G::WT + 2 * lh) * 64 + 2 * li;
      const float* xb0 = sX + (rb * G::WT + 2 * lh) * 64 + 2 * li;
      const float* db0 = sD + (2 * lh) * 64 + coh * 32 + li;
      if (!BF16) wgrad_wino_steps<G>(acc, xa0, xb0, db0, f32x2{sg, sg}, f32x2{c0, c0}, f32x2{c1, c1}, wgf_load_slice);
    }
#if SSP_LEGACY_ALGOS
    if (BF16) {
      // (pinned: hipcc converts each value on its own as soon as it exists and merges the halves with v_perm_b32)
      auto cvt2 = [](float lo, float hi) -> unsigned {
        unsigned d;
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(d) : "v"(lo), "v"(hi));
        return d;
      };
      typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
#pragma unroll 1
      for (int s = 0; s < 4; ++s) {  // 8 Winograd tiles per MFMA: lane half lh supplies tiles 8 s + 4 lh .. + 3
        // Two tiles at a time: the values of tiles (tt, tt + 1) convert PAIRWISE (one v_cvt_pk_bf16_f32 per two values, already
        // in operand order) - converted one by one, the four values of an operand register were assembled with a v_perm_b32
        // each; the loop is bound by its vector instructions (8 bf16 MFMAs = 128 matrix-pipe cycles beside ~150 of them)
        unsigned Vp[4][2][2], Dp[4][2];  // [component][m-tile][tile pair]: two bf16 values each
#pragma unroll
        for (int tp = 0; tp < 2; ++tp) {
          f32x2 Vt[2][4];
          float Dt[2][4];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int t = 8 * s + 4 * lh + 2 * tp + u;
            const int ty = t / G::TTX, tx = t - ty * G::TTX;
            const float* xa = sX + ((2 * ty + ra) * G::WT + 2 * tx) * 64 + 2 * li;
            const float* xb = sX + ((2 * ty + rb) * G::WT + 2 * tx) * 64 + 2 * li;
            f32x2 T[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const f32x2 uu = *reinterpret_cast<const f32x2*>(xa + c * 64), w = *reinterpret_cast<const f32x2*>(xb + c * 64);
              T[c][0] = fmaf(sg, w[0], uu[0]);
              T[c][1] = fmaf(sg, w[1], uu[1]);
            }
            Vt[u][0] = T[0] - T[2]; Vt[u][1] = T[1] + T[2]; Vt[u][2] = T[2] - T[1]; Vt[u][3] = T[1] - T[3];
            const float* db = sD + ((2 * ty) * G::TW + 2 * tx) * 64 + coh * 32 + li;
            const float r0 = c0 * db[0] + c1 * db[G::TW * 64];
            const float r1 = c0 * db[64] + c1 * db[G::TW * 64 + 64];
            Dt[u][0] = r0; Dt[u][1] = r0 + r1; Dt[u][2] = r0 - r1; Dt[u][3] = -r1;
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            Vp[j][0][tp] = cvt2(Vt[0][j][0], Vt[1][j][0]);
            Vp[j][1][tp] = cvt2(Vt[0][j][1], Vt[1][j][1]);
            Dp[j][tp] = cvt2(Dt[0][j], Dt[1][j]);
          }
        }
        // hipcc inserts no wait states between an inline-asm result and an MFMA that reads it: all operands complete, then the MFMAs
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 1");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const u32x2_t db4 = {Dp[j][0], Dp[j][1]};
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const u32x2_t vb4 = {Vp[j][e][0], Vp[j][e][1]};
            acc[j][e] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(__builtin_bit_cast(s16x4, vb4), __builtin_bit_cast(s16x4, db4), acc[j][e], 0, 0, 0);
          }
        }
      }
    }
#endif
    WGF_T(4)
  }
#if WGF_TRACE
  if (a.trace != nullptr && blockIdx.x == 0 && lane == 0) {
#pragma unroll
    for (int k = 0; k < 5; ++k) a.trace[wave * 8 + k] = tc_[k];
    a.trace[wave * 8 + 5] = __builtin_readcyclecounter() - tk0_;
    a.trace[wave * 8 + 6] = (unsigned long long)(t_end - t_begin);
    a.trace[wave * 8 + 7] = __builtin_amdgcn_s_memrealtime() - tr0_;   // loop cycles / this = shader clock / 100 MHz
  }
#endif
#undef WGF_T
#undef WGF_PREP
#undef WGF_LOAD_ALL
  // The right-hand product of dW = G^T M G runs over the four components of this wave's row: applied here, in registers, so that
  // 12 instead of 16 components per (ci, co) travel to HBM and back (the left-hand product runs over the rows = waves and stays
  // with wgrad_fused12_reduce_block; done here through LDS it cost the launch more than the smaller slab saved: PERF_LOG 3c).
  // partial slab: [blk][row i][x 3][ci 64][co 64]; M-tile e, row m <-> input channel 2 m + e
  float* dst = a.partial + (size_t)blockIdx.x * (12 * 4096) + (irow * 3) * 4096 + coh * 32 + li;
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = (r & 3) + 8 * (r >> 2) + 4 * lh;
      const float t = 0.5f * (acc[1][e][r] + acc[2][e][r]), d = 0.5f * (acc[1][e][r] - acc[2][e][r]);
      dst[0 * 4096 + (2 * m + e) * 64] = acc[0][e][r] + t;
      dst[1 * 4096 + (2 * m + e) * 64] = d;
      dst[2 * 4096 + (2 * m + e) * 64] = t + acc[3][e][r];
    }
}

}  // namespace sspk
