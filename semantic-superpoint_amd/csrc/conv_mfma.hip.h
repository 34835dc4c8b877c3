// Implicit-GEMM convolution kernels for gfx950 (CDNA4), fp32 in / fp32 accumulate on the matrix cores
// (v_mfma_f32_32x32x2_f32: exact f32, 157 TF peak).  NHWC activations, stride 1, "same" padding.
//
//   conv_mfma_kernel   forward conv and data-gradient conv (dgrad = conv with flipped/transposed weights)
//   wgrad_mfma_kernel  weight-gradient: dW[tap][ci][co] = sum_pixels X[pixel+tap][ci] * dY[pixel][co]
//
// The BatchNorm+ReLU(+2x2 max-pool) of the PRODUCING layer is applied while the input tile is staged
// into LDS ("normalise on load"), so activations make one HBM round trip per layer
// (reference: models/unet_parts.py:10-48 runs conv, BN, ReLU, pool as four separate passes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pk_math.hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace sspk {

constexpr int CK = 16;       // input channels staged per K-chunk
constexpr int CS = CK + 4;   // LDS pixel stride in floats (pad 4 -> conflict-free ds_read_b128)
constexpr int NB = 64;       // output channels per block
constexpr int NREP = 32;     // replicas of every fp64 statistics accumulator (spreads same-address atomics)

// ---- BatchNorm statistics -> affine on the CONSUMER side -------------------------------------------------------------------------
// A 7-microsecond bn_finalize_kernel between two convolutions costs ~13 us of the step (dispatch of a dependent launch + its own
// latency chain; PERF_LOG round 6 section 4).  In training, the convolution that READS a layer under BatchNorm + ReLU therefore derives
// scale / shift in its prologue from the producer's raw statistics (32 replicas of sum and sum of squares per channel: the launch
// boundary has made them visible) - every workgroup for the view it works on, and workgroup 0 additionally for both views, storing
// scale / shift / mean / invstd for the later readers (weight gradients, BatchNorm backward) and updating the running statistics in the
// reference's order (view 0, then view 1).  All of them evaluate ONE function (bn_lazy_stats + bn_affine_of), so the values used
// and the values stored are the same bits.  The replicas are added in index order (bn_finalize_kernel: pairwise) - the fp64 sums may
// differ in their last bit from that kernel's; under ssp_set_deterministic every addend is a multiple of one quantum and any order is exact.
struct BnLazy {
  const double* stats[2] = {nullptr, nullptr};   // [NREP][2C] of the producer layer, per view
  const float* gamma = nullptr;
  const float* beta = nullptr;
  float* scale[2] = {nullptr, nullptr};          // where workgroup 0 stores (mode 2)
  float* shift[2] = {nullptr, nullptr};
  float* mean[2] = {nullptr, nullptr};
  float* invstd[2] = {nullptr, nullptr};
  float* running_mean = nullptr;
  float* running_var = nullptr;
  int64_t* nbt = nullptr;
  double count = 0.0;
  int C = 0;
  int nviews = 0;
  int mode = 0;   // 0: the arrays in_scale / in_shift are valid; 2: derive them here, workgroup 0 stores
};
__device__ __forceinline__ void bn_lazy_stats(const double* __restrict__ stats, int C, int c, double count, double& mean, double& var) {
  double s1 = 0.0, s2 = 0.0;
#pragma unroll
  for (int h = 0; h < 2; ++h) {   // two rounds of 16 replicas: 32 loads in flight
    double v1[NREP / 2], v2[NREP / 2];
#pragma unroll
    for (int k = 0; k < NREP / 2; ++k) {
      v1[k] = stats[(size_t)(h * (NREP / 2) + k) * 2 * C + c];
      v2[k] = stats[(size_t)(h * (NREP / 2) + k) * 2 * C + C + c];
    }
#pragma unroll
    for (int k = 0; k < NREP / 2; ++k) { s1 += v1[k]; s2 += v2[k]; }
  }
  mean = s1 / count;
  var = s2 / count - mean * mean;
  if (var < 0) var = 0;
}
__device__ __forceinline__ void bn_affine_of(double mean, double var, float gamma, float beta, float& invstd, float& sc, float& sh) {
  invstd = (float)(1.0 / sqrt(var + 1e-5));
  sc = gamma * invstd;
  sh = beta - (float)mean * sc;
}
// scale / shift of channel c of view `view` (every workgroup)
__device__ __forceinline__ void bn_lazy_affine(const BnLazy& z, int view, int c, float& sc, float& sh) {
  double mean, var;
  float invstd;
  bn_lazy_stats(z.stats[view], z.C, c, z.count, mean, var);
  bn_affine_of(mean, var, z.gamma[c], z.beta[c], invstd, sc, sh);
}
// workgroup 0: what bn_finalize_kernel stored, for every view
__device__ __forceinline__ void bn_lazy_store(const BnLazy& z, int c) {
  for (int v = 0; v < z.nviews; ++v) {   // view 0 then view 1: the running statistics are updated in the reference's order
    double mean, var;
    float invstd, sc, sh;
    bn_lazy_stats(z.stats[v], z.C, c, z.count, mean, var);
    bn_affine_of(mean, var, z.gamma[c], z.beta[c], invstd, sc, sh);
    const double unbiased = z.count > 1 ? var * z.count / (z.count - 1) : var;
    z.running_mean[c] = (float)(0.9 * (double)z.running_mean[c] + 0.1 * mean);
    z.running_var[c] = (float)(0.9 * (double)z.running_var[c] + 0.1 * unbiased);
    if (c == 0 && z.nbt != nullptr) *z.nbt += 1;
    z.mean[v][c] = (float)mean;
    z.invstd[v][c] = invstd;
    z.scale[v][c] = sc;
    z.shift[v][c] = sh;
  }
}

struct ConvArgs {
  const float* in;        // NHWC [N, H*(pool?2:1), W*(pool?2:1), in_cs]
  const float* wpk;       // packed weights [cob][chunk][tap][g][h][64][4]
  const float* bias;      // [Cout] or nullptr
  float* out;             // NHWC [N,H,W,out_cs]
  const float* in_scale;  // [Cin] (IN_MODE != 0)
  const float* in_shift;
  double* stats;          // [NREP][2*Cout] sum, sumsq of the (biased) conv output, or nullptr
  // second, independent problem of identical shape and weights (the other view of the pair): nprob == 2 splits
  // the persistent grid by XCD (0-3 -> problem 0, 4-7 -> problem 1) so that the 30x40 maps fill the chip
  const float* in2;
  float* out2;
  const float* in_scale2;
  const float* in_shift2;
  double* stats2;
  int nprob;
  int N, H, W;
  int Cin, in_cs, in_co;
  int Cout, out_cs, out_co;
  int tiles_x, tiles_y;
  int nchunks, ncob;
  unsigned in_bytes, out_bytes, wpk_bytes;  // buffer descriptor ranges: in_bytes = bytes of ONE input image
  int ablate;             // perf-debug only (tools/archive/ablate_conv.py): 1 no global loads, 2 no LDS writes, 4 no stores, 8 no MFMA
  // Data-gradient launches of conv_wino_pipe_kernel: pass 1 of the BatchNorm backward of the layer BELOW (whose activation
  // gradient this launch produces) accumulated in the epilogue into `stats` (= that layer's backward sums, same
  // [NREP][2 C] layout as the forward statistics): S1 = sum dZ, S2 = sum dZ * xhat.
  //   bnr_mode 1: ReLU layer,        dZ = out * [y * scale + shift > 0], xhat = (y - mean) * invstd,  bnr_t = y
  //   bnr_mode 2: ReLU + 2x2 max-pool, dZ = out * [apool > 0],           xhat = (apool - beta) / gamma, bnr_t = apool
  // bnr_t has the geometry of `out` ([N,H,W,bnr_cs], channel offset bnr_co); bnr_p*[k] = per-channel parameter arrays of
  // problem k: mode 1 {scale, shift, mean, invstd}, mode 2 {beta, gamma, -, -}.
  int bnr_mode = 0;
  const float* bnr_t = nullptr;
  const float* bnr_t2 = nullptr;
  int bnr_cs = 0, bnr_co = 0;
  const float* bnr_p0[2] = {nullptr, nullptr};
  const float* bnr_p1[2] = {nullptr, nullptr};
  const float* bnr_p2[2] = {nullptr, nullptr};
  const float* bnr_p3[2] = {nullptr, nullptr};
  // Forward launches of conv_wino_pipe_kernel whose output feeds BatchNorm + ReLU + MaxPool2d(2): the epilogue also
  // writes pool_out[k] = [N, H/2, W/2, Cout] = the per-channel max (gamma >= 0) or min (gamma < 0) of every 2x2 window
  // of the RAW conv output.  maxpool(relu(scale y + shift)) = relu(scale pool(y) + shift) with that choice (scale =
  // gamma * invstd, invstd > 0; fma and relu are monotonic, so the values are bit-identical): the consumers apply
  // BatchNorm + ReLU on load to a quarter-size tensor and the separate pooled-activation pass disappears.
  float* pool_out[2] = {nullptr, nullptr};
  const float* pool_gamma = nullptr;
  // perf-debug only (a build with SSP_HIPCC_EXTRA=-DW4_TRACE=1 and SSP_W4_TRACE=N in the environment): per-phase cycle sums of the four
  // waves of workgroup 0 of conv_wino4_kernel, else nullptr
  unsigned long long* trace = nullptr;
  BnLazy lazy;   // training forward: BatchNorm affine of the INPUT layer derived here instead of by a bn_finalize_kernel launch
};

// lane/row index m (0..31) of an MFMA M-tile -> pixel (r,c) inside the SH x SW sub-rectangle.
// SW==32: one image row.  SW==8: 4x8 patch; the Thue-Morse column swizzle makes the four 16-lane
// groups of ds_read_b128 hit 16 distinct bank quads with an LDS row pitch of 12 pixels.
template <int SW>
__device__ __forceinline__ void mpix(int m, int& r, int& c) {
  if (SW == 32) {
    r = 0;
    c = m;
  } else {
    const int b = m >> 2;
    r = b >> 1;
    c = (m & 3) + 4 * (__popc(b) & 1);
  }
}

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

template <int IN_MODE>
__device__ __forceinline__ float4 xform(float4 v, float4 sc, float4 sh) {
  if (IN_MODE == 0) return v;
  float4 o;
  o.x = fmaxf(fmaf(v.x, sc.x, sh.x), 0.f);
  o.y = fmaxf(fmaf(v.y, sc.y, sh.y), 0.f);
  o.z = fmaxf(fmaf(v.z, sc.z, sh.z), 0.f);
  o.w = fmaxf(fmaf(v.w, sc.w, sh.w), 0.f);
  return o;
}

__device__ __forceinline__ float4 max4(float4 a, float4 b) {
  return make_float4(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w));
}

// Loads the (transformed) 4-channel value of conv-input pixel (n,gy,gx); zero outside the image.
template <int IN_MODE>
__device__ __forceinline__ float4 load_in(const float* __restrict__ in, int n, int gy, int gx, int H, int W, int cs,
                                          int coff, bool cvalid, float4 sc, float4 sh) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (cvalid && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) {
    if (IN_MODE != 2) {
      const size_t off = ((size_t)(n * H + gy) * W + gx) * cs + coff;
      v = xform<IN_MODE>(*reinterpret_cast<const float4*>(in + off), sc, sh);
    } else {
      const int W2 = 2 * W;
      const size_t off = ((size_t)(n * 2 * H + 2 * gy) * W2 + 2 * gx) * cs + coff;
      const float4 a = xform<1>(*reinterpret_cast<const float4*>(in + off), sc, sh);
      const float4 b = xform<1>(*reinterpret_cast<const float4*>(in + off + cs), sc, sh);
      const float4 c = xform<1>(*reinterpret_cast<const float4*>(in + off + (size_t)W2 * cs), sc, sh);
      const float4 d = xform<1>(*reinterpret_cast<const float4*>(in + off + (size_t)W2 * cs + cs), sc, sh);
      v = max4(max4(a, b), max4(c, d));
    }
  }
  return v;
}

template <int KS, int SH, int SW>
struct ConvGeom {
  static constexpr int TAPS = KS * KS;
  static constexpr int PAD = KS / 2;
  static constexpr int TH = 8 * SH;
  static constexpr int TW = SW;
  static constexpr int HT = TH + 2 * PAD;
  static constexpr int WT = TW + 2 * PAD;
  static constexpr int RP = (SW == 32) ? WT : 12;  // LDS row pitch in pixels
  static constexpr int A_FLOATS = HT * RP * CS;
  static constexpr int B_FLOATS = TAPS * CK * NB;
  static constexpr int O_FLOATS = TH * TW * NB;  // output tile staged for the transposed store
  static constexpr int LDS_BYTES = ((A_FLOATS + B_FLOATS) > O_FLOATS ? (A_FLOATS + B_FLOATS) : O_FLOATS) * 4;
};

// Block: 256 threads (4 waves).  Output tile: (8*SH) x SW pixels x 64 output channels; wave w owns M-tiles
// 2w, 2w+1 (32 pixels each) x both 32-channel N-tiles -> 4 accumulators of 32x32.
//
// PERSISTENT: the grid is 2 blocks per CU (a multiple of 8).  Block b runs on XCD b%8 (observed placement; only
// speed depends on it), owns ONE 64-channel output block `cob` and walks the tiles of its XCD's contiguous tile
// range, so neighbouring tiles (shared halos) and the cob's packed weights stay in that XCD's L2.  The loads of
// the next (tile, chunk) step are issued before the MFMAs of the current step, the output stores of a tile are
// fire-and-forget and drain under the next tile's MFMAs, and the BatchNorm sums are kept in registers until the
// block ends (one atomic flush per block instead of one per tile).
template <int KS, int IN_MODE, int SH, int SW>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(const ConvArgs a) {
  using G = ConvGeom<KS, SH, SW>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sA = smem;
  float* sB = smem + G::A_FLOATS;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;

  // ---- work assignment ----
  const int nslot = gridDim.x >> 3;             // blocks per XCD
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int per_cob = nslot / a.ncob;           // blocks per (XCD, cob)
  const int cob = slot % a.ncob, jj = slot / a.ncob;
  const int ntiles = a.N * a.tiles_y * a.tiles_x;
  const int xpp = 8 / a.nprob;                  // XCDs per problem
  const int prob = xcd / xpp, xl = xcd - prob * xpp;
  const int per_t = (ntiles + xpp - 1) / xpp;
  const int t_end = min(ntiles, (xl + 1) * per_t);
  int tile = xl * per_t + jj;
  if (jj >= per_cob || tile >= t_end) return;   // whole block exits before any barrier
  const float* const p_in = prob ? a.in2 : a.in;
  float* const p_out = prob ? a.out2 : a.out;
  const float* const p_scale = prob ? a.in_scale2 : a.in_scale;
  const float* const p_shift = prob ? a.in_shift2 : a.in_shift;
  double* const p_stats = prob ? a.stats2 : a.stats;

  int pr, pc;
  mpix<SW>(li, pr, pc);
  int aoff[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) aoff[mt] = (((wave * 2 + mt) * SH + pr) * G::RP + pc) * CS + lh * 4;
  const int boff = (lh * NB + li) * 4;

  const int q4 = tid & 3;
  constexpr int NH = (G::HT * G::WT + 63) / 64;
  constexpr int NW = G::B_FLOATS / 4 / 256;
  f32x4 hreg[NH], wreg[NW];
  f32x4 psc = {1.f, 1.f, 1.f, 1.f}, psh = {0.f, 0.f, 0.f, 0.f};
  // byte offset (without channel chunk) of halo slot i for the tile being LOADED; OOB marker = outside the image:
  // a raw buffer load beyond num_records returns 0 without touching memory.
  constexpr unsigned OOB = 0x80000000u;
  unsigned hoff[NH];
  int lds_off[NH];  // float offset in sA (tile independent), -1 = unused slot
#pragma unroll
  for (int i = 0; i < NH; ++i) {
    const int pp = (tid >> 2) + 64 * i;
    const int r = pp / G::WT, c = pp - r * G::WT;
    lds_off[i] = (pp < G::HT * G::WT) ? (r * G::RP + c) * CS + q4 * 4 : -1;
  }
  // tile decode + halo slot offsets of the tile whose loads are issued next
  int ld_n, ld_ty0, ld_tx0;
#define SSP_DECODE_TILE(T)                                                                               \
  {                                                                                                      \
    const int tx_ = (T) % a.tiles_x, t2_ = (T) / a.tiles_x;                                              \
    ld_tx0 = tx_ * G::TW;                                                                                \
    ld_ty0 = (t2_ % a.tiles_y) * G::TH;                                                                  \
    ld_n = t2_ / a.tiles_y;                                                                              \
    _Pragma("unroll") for (int i = 0; i < NH; ++i) {                                                     \
      const int pp = (tid >> 2) + 64 * i;                                                                \
      const int r = pp / G::WT, c = pp - r * G::WT;                                                      \
      const int gy = ld_ty0 + r - G::PAD, gx = ld_tx0 + c - G::PAD;                                      \
      const bool ok = pp < G::HT * G::WT && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W; \
      hoff[i] = ok ? (unsigned)((gy * a.W + gx) * a.in_cs + a.in_co + q4 * 4) * 4u : OOB;                \
    }                                                                                                    \
    /* one descriptor per image: 32-bit offsets stay inside it whatever the batch size */               \
    rsrc_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p_in) + (size_t)ld_n * img_floats, 0, \
                                                a.in_bytes, 0x00020000);                                 \
  }
#define SSP_ISSUE_LOADS(CHUNK)                                                                          \
  if (!(a.ablate & 1)) {                                                                                \
    const int c0_ = (CHUNK) * CK + q4 * 4;                                                              \
    const bool cvalid_ = c0_ < a.Cin;                                                                   \
    if (IN_MODE != 0) {                                                                                 \
      const int cc_ = cvalid_ ? c0_ : 0;                                                                \
      psc = *reinterpret_cast<const f32x4*>(p_scale + cc_);                                             \
      psh = *reinterpret_cast<const f32x4*>(p_shift + cc_);                                             \
    }                                                                                                   \
    if (IN_MODE != 2) {                                                                                 \
      const int soff_ = (CHUNK) * CK * 4;                                                               \
      _Pragma("unroll") for (int i = 0; i < NH; ++i) {                                                  \
        const unsigned vo_ = partial_k ? (cvalid_ ? hoff[i] : OOB) : hoff[i];                           \
        hreg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_in, vo_, soff_, 0)); \
      }                                                                                                 \
    }                                                                                                   \
    const int wbase_ = (cob * a.nchunks + (CHUNK)) * G::B_FLOATS * 4;                                   \
    _Pragma("unroll") for (int j = 0; j < NW; ++j)                                                      \
      wreg[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, tid * 16, wbase_ + j * 4096, 0)); \
  }

  const size_t img_floats = (size_t)a.H * a.W * a.in_cs;
  __amdgpu_buffer_rsrc_t rsrc_in;
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wpk), 0, a.wpk_bytes, 0x00020000);
  const bool partial_k = (a.Cin % CK) != 0;  // last K-chunk has channel quads beyond Cin (65/133-channel dY)

  SSP_DECODE_TILE(tile)
  SSP_ISSUE_LOADS(0)

  float ssum[2] = {0.f, 0.f}, ssq[2] = {0.f, 0.f};
  float bias_v[2];
  bool covalid[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int co = cob * NB + nt * 32 + li;
    covalid[nt] = co < a.Cout;
    bias_v[nt] = (a.bias != nullptr && covalid[nt]) ? a.bias[co] : 0.f;
  }

  for (;;) {  // ---- one output tile per iteration ----
    const int n = ld_n, ty0 = ld_ty0, tx0 = ld_tx0;  // the tile being computed (= the one just prefetched)
    unsigned hmask = 0;                               // validity of its halo slots
#pragma unroll
    for (int i = 0; i < NH; ++i) hmask |= (hoff[i] != OOB ? 1u : 0u) << i;
    const int next_tile = tile + per_cob;
    const bool has_next = next_tile < t_end;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    for (int chunk = 0; chunk < a.nchunks; ++chunk) {
      __syncthreads();  // every wave has finished reading the LDS image of the previous step
      if (!(a.ablate & 2)) {
        const int c0 = chunk * CK + q4 * 4;
        const bool cvalid = c0 < a.Cin;
        if (IN_MODE != 2) {
#pragma unroll
          for (int i = 0; i < NH; ++i) {
            if (lds_off[i] >= 0) {
              f32x4 v = {0.f, 0.f, 0.f, 0.f};
              if (cvalid && ((hmask >> i) & 1u)) {
                v = hreg[i];
                if (IN_MODE != 0) {
#pragma unroll
                  for (int e = 0; e < 4; ++e) v[e] = fmaxf(fmaf(v[e], psc[e], psh[e]), 0.f);
                }
              }
              *reinterpret_cast<f32x4*>(sA + lds_off[i]) = v;
            }
          }
        } else {
          const int coff = a.in_co + c0;
          const float4 sc4 = make_float4(psc[0], psc[1], psc[2], psc[3]), sh4 = make_float4(psh[0], psh[1], psh[2], psh[3]);
#pragma unroll 2
          for (int pp = tid >> 2; pp < G::HT * G::WT; pp += 64) {
            const int r = pp / G::WT, c = pp - r * G::WT;
            const float4 v = load_in<IN_MODE>(p_in, n, ty0 + r - G::PAD, tx0 + c - G::PAD, a.H, a.W, a.in_cs, coff,
                                              cvalid, sc4, sh4);
            *reinterpret_cast<float4*>(sA + (r * G::RP + c) * CS + q4 * 4) = v;
          }
        }
        f32x4* dst = reinterpret_cast<f32x4*>(sB);
#pragma unroll
        for (int j = 0; j < NW; ++j) dst[tid + 256 * j] = wreg[j];
      }
      __syncthreads();
      // Unconditional prefetch of the next step: next chunk of this tile, or chunk 0 of the block's next tile
      // (of this tile again when it was the last: redundant but free of control flow, so the loads stay in
      // flight during the MFMAs below).
      {
        const bool last = chunk + 1 == a.nchunks;
        if (last && has_next) SSP_DECODE_TILE(next_tile)
        const int nxt = last ? 0 : chunk + 1;
        SSP_ISSUE_LOADS(nxt)
        __builtin_amdgcn_sched_barrier(0);  // hipcc otherwise sinks the loads below the MFMAs (checked in the ISA)
      }
      // ---- MFMA ----
      if (!(a.ablate & 8))
#pragma unroll
      for (int tap = 0; tap < G::TAPS; ++tap) {
        const int dy = tap / KS, dx = tap % KS;
#pragma unroll
        for (int g = 0; g < CK / 8; ++g) {
          const float4 a0 = *reinterpret_cast<const float4*>(sA + aoff[0] + (dy * G::RP + dx) * CS + g * 8);
          const float4 a1 = *reinterpret_cast<const float4*>(sA + aoff[1] + (dy * G::RP + dx) * CS + g * 8);
          const float4 b0 = *reinterpret_cast<const float4*>(sB + boff + (tap * (CK / 8) + g) * 2 * NB * 4);
          const float4 b1 = *reinterpret_cast<const float4*>(sB + boff + (tap * (CK / 8) + g) * 2 * NB * 4 + 32 * 4);
          const float av0[4] = {a0.x, a0.y, a0.z, a0.w}, av1[4] = {a1.x, a1.y, a1.z, a1.w};
          const float bv0[4] = {b0.x, b0.y, b0.z, b0.w}, bv1[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[e], bv0[e], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[e], bv1[e], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[e], bv0[e], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[e], bv1[e], acc[1][1], 0, 0, 0);
          }
        }
      }
    }

    // ---- tile epilogue ----
    // Global stores are issue-bound (one dword store per accumulator register cost ~12 % of the kernel), so the
    // tile is transposed through LDS: 64 ds_write_b32 per lane into a [pixel][64 co] image, then 16 x
    // (ds_read_b128 + global_store_dwordx4) per lane, a wave writing 4 pixels x 256 contiguous bytes.
    // Bias is added and the BatchNorm partial sums are taken from the registers on the way.
    if (!(a.ablate & 4)) {
      const bool full = (ty0 + G::TH <= a.H) && (tx0 + G::TW <= a.W);
      __syncthreads();  // all waves have finished their MFMA reads of sA/sB: the LDS image can be reused
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = (r & 3) + 8 * (r >> 2) + 4 * lh;
            int rr, cc;
            mpix<SW>(m, rr, cc);
            const int orow = (wave * 2 + mt) * SH + rr;
            const float v = acc[mt][nt][r] + bias_v[nt];
            smem[(orow * G::TW + cc) * NB + nt * 32 + li] = v;
            if (p_stats != nullptr && covalid[nt] && (full || (ty0 + orow < a.H && tx0 + cc < a.W))) {
              ssum[nt] += v;
              ssq[nt] += v * v;
            }
          }
        }
      }
      __syncthreads();
      const int q16 = tid & 15;               // channel quad of the 64-channel block
      const int co4 = cob * NB + q16 * 4;
      const int nvalid = min(4, a.Cout - co4);  // channels of this quad that exist (<= 0: none)
#pragma unroll 4
      for (int k = 0; k < (G::TH * G::TW) / 16; ++k) {
        const int lp = (tid >> 4) + 16 * k;
        const int orow = lp / G::TW, ocol = lp - orow * G::TW;
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
    float* red = smem;  // [4 waves][2 nt][32][2]
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      float s = ssum[nt] + __shfl_xor(ssum[nt], 32);
      float q = ssq[nt] + __shfl_xor(ssq[nt], 32);
      if (lh == 0) {
        red[((wave * 2 + nt) * 32 + li) * 2 + 0] = s;
        red[((wave * 2 + nt) * 32 + li) * 2 + 1] = q;
      }
    }
    __syncthreads();
    if (tid < 128) {
      const int ch = tid >> 1, which = tid & 1;  // ch in 0..63 -> nt = ch>>5, li = ch&31
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) t += red[((w * 2 + (ch >> 5)) * 32 + (ch & 31)) * 2 + which];
      const int co = cob * NB + ch;
      if (co < a.Cout)
        acc_add_stats(p_stats + (size_t)(blockIdx.x % NREP) * 2 * a.Cout + which * a.Cout + co, (double)t);
    }
  }
}

#undef SSP_DECODE_TILE
#undef SSP_ISSUE_LOADS

// ------------------------------------------------------------------------------------------------
// Weight gradient.  Block = 4 waves; block owns a 64(ci) x 64(co) x TAPS slab and loops over spatial tiles
// of 128 pixels, keeping the slab in accumulators (9 x 32x32 per wave: wave = (ci half, co half)).
// Partial slabs of the `nsplit` blocks are written to scratch and summed by wgrad_reduce_kernel.
// ------------------------------------------------------------------------------------------------
struct WgradArgs {
  const float* in;        // conv input (pre-transform), NHWC
  const float* dout;      // dY NHWC [N,H,W,dout_cs]
  float* partial;         // [ncib*ncob*nsplit][TAPS][64][64]
  const float* in_scale;
  const float* in_shift;
  // second problem (other view of the pair) accumulated into the SAME weight gradient: tiles [ntiles, 2*ntiles)
  const float* in2;
  const float* dout2;
  const float* in_scale2;
  const float* in_shift2;
  int nprob;
  int N, H, W;
  int Cin, in_cs, in_co;
  int Cout, dout_cs, dout_co;
  int tiles_x, tiles_y, ntiles;  // per-image tiles and total tiles of ONE problem
  int ncib, ncob, nsplit;
  int ablate;  // perf-debug only (wgrad_wino_kernel): 1 no global loads, 2 no LDS writes, 8 no MFMA loop, 64 no LDS reads
  // wgrad_wino_fused_kernel: the APPLY pass of this layer's BatchNorm + ReLU + MaxPool backward rides the dY staging.
  // dout / dout2 then hold the gradient wrt the POOLED activation [N,H/2,W/2,dout_cs]; f_y = the layer's raw conv output
  // [N,H,W,f_ycs] (channel offset 0), f_dy = dY written for the data-gradient convolution (same geometry as f_y); per
  // problem k: scale / shift / mean / invstd of the layer and f_k12 = {S1/n, S2/n} (bn_bwd_sums_kernel); f_gamma shared.
  const float* f_y[2] = {nullptr, nullptr};
  float* f_dy[2] = {nullptr, nullptr};
  const float* f_scale[2] = {nullptr, nullptr};
  const float* f_shift[2] = {nullptr, nullptr};
  const float* f_mean[2] = {nullptr, nullptr};
  const float* f_invstd[2] = {nullptr, nullptr};
  const float* f_k12[2] = {nullptr, nullptr};
  const float* f_gamma = nullptr;
  int f_ycs = 0;
  // f_lazy (layers without pooling): there was no bn_bwd_sums_kernel launch - every workgroup reduces the 32 replicas of the backward
  // sums S1, S2 itself (f_bsums[k]: [NREP][2 Cout], complete since the data-gradient launch above), and the workgroups of the first
  // input-channel block and split add dgamma / dbeta / the bias gradient, view 0 then view 1 like that kernel
  int f_lazy = 0;
  const double* f_bsums[2] = {nullptr, nullptr};
  double f_count = 0.0;
  float* f_dgamma = nullptr;
  float* f_dbeta = nullptr;
  float* f_dbias = nullptr;
  unsigned long long* trace = nullptr;  // WGF_TRACE builds only (wgrad_wino_fused.hip.h): phase cycle sums of workgroup 0
};

template <int KS, int SH, int SW>
struct WgradGeom {
  static constexpr int TAPS = KS * KS;
  static constexpr int PAD = KS / 2;
  static constexpr int TH = 2 * SH;   // 64-pixel tiles: 2x32 (wide maps) or 8x8
  static constexpr int TW = SW;
  static constexpr int HT = TH + 2 * PAD;
  static constexpr int WT = TW + 2 * PAD;
  static constexpr int P = TH * TW;  // 64
  static constexpr int X_FLOATS = HT * WT * 64;
  static constexpr int D_FLOATS = P * 64;
  static constexpr int LDS_BYTES = (X_FLOATS + D_FLOATS + 256) * 4;  // + scale/shift of the 64 input channels x 2 problems
};

// Block = 4 waves = (ci half, co half); 2 blocks per CU (51 KB LDS, <= 256 registers incl. 144 accumulators).
// Each block walks a CONTIGUOUS range of 64-pixel tiles; the global loads of tile t+1 (X halo rows + dY tile,
// 13 float4 per lane) are issued before the 288 MFMAs of tile t and written to LDS (with the producer's
// BN/ReLU applied) afterwards.  IN_MODE 2 (pooled input, 4 loads per pixel) stages its X tile synchronously.
template <int KS, int IN_MODE, int SH, int SW>
__global__ __launch_bounds__(256, 2) void wgrad_mfma_kernel(const WgradArgs a) {
  using G = WgradGeom<KS, SH, SW>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sX = smem;
  float* sD = smem + G::X_FLOATS;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int cih = wave >> 1, coh = wave & 1;

  int bid = blockIdx.x;
  const int split = bid % a.nsplit;
  bid /= a.nsplit;
  const int cob = bid % a.ncob;
  const int cib = bid / a.ncob;
  const int tot_tiles = a.ntiles * a.nprob;
  const int per = (tot_tiles + a.nsplit - 1) / a.nsplit;
  const int t_begin = split * per, t_end = min(tot_tiles, t_begin + per);

  f32x16 acc[G::TAPS];
#pragma unroll
  for (int t = 0; t < G::TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const int q16 = tid & 15;  // channel quad 0..15 of the 64-channel slab
  const int ci0 = cib * 64 + q16 * 4;
  const bool civalid = ci0 < a.Cin;
  // BN scale/shift of this block's 64 input channels live in LDS (frees 8 VGPRs of a 256-register kernel)
  float* sS = smem + G::X_FLOATS + G::D_FLOATS;
  if (IN_MODE != 0 && tid < 32) {
    const int pr = tid >> 4;  // problem 0 / 1
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

  constexpr int NX = (G::HT * G::WT + 15) / 16;  // halo pixels per thread (pp = (tid>>4) + 16*i)
  constexpr int ND = G::P / 16;                  // dY pixels per thread
  f32x4 xreg[NX], dreg[ND];
  unsigned xmask = 0, dmask = 0;  // validity of the slots of the tile whose loads are in flight

#define SSP_WG_ISSUE(TILE)                                                                                    \
  {                                                                                                           \
    const int pr_ = (TILE) >= a.ntiles ? 1 : 0;                                                               \
    const int tl_ = (TILE) - pr_ * a.ntiles;                                                                  \
    const float* const pin_ = pr_ ? a.in2 : a.in;                                                             \
    const float* const pdo_ = pr_ ? a.dout2 : a.dout;                                                         \
    const int tx_ = tl_ % a.tiles_x, t2_ = tl_ / a.tiles_x;                                                   \
    const int ty0_ = (t2_ % a.tiles_y) * G::TH, tx0_ = tx_ * G::TW, n_ = t2_ / a.tiles_y;                     \
    xmask = 0; dmask = 0;                                                                                     \
    if (IN_MODE != 2) {                                                                                       \
      _Pragma("unroll") for (int i = 0; i < NX; ++i) {                                                        \
        const int pp = (tid >> 4) + 16 * i;                                                                   \
        const int r = pp / G::WT, c = pp - r * G::WT;                                                         \
        const int gy = ty0_ + r - G::PAD, gx = tx0_ + c - G::PAD;                                             \
        const bool ok = pp < G::HT * G::WT && civalid && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W; \
        const size_t off = ok ? ((size_t)(n_ * a.H + gy) * a.W + gx) * a.in_cs + a.in_co + ci0 : (size_t)0;   \
        xreg[i] = *reinterpret_cast<const f32x4*>(pin_ + off);                                                \
        xmask |= (ok ? 1u : 0u) << i;                                                                         \
      }                                                                                                       \
    }                                                                                                         \
    _Pragma("unroll") for (int i = 0; i < ND; ++i) {                                                          \
      const int pp = (tid >> 4) + 16 * i;                                                                     \
      const int r = pp / G::TW, c = pp - r * G::TW;                                                           \
      const int gy = ty0_ + r, gx = tx0_ + c;                                                                 \
      const bool ok = covalid && gy < a.H && gx < a.W;                                                        \
      const size_t off = ok ? ((size_t)(n_ * a.H + gy) * a.W + gx) * a.dout_cs + a.dout_co + co0 : (size_t)0; \
      dreg[i] = *reinterpret_cast<const f32x4*>(pdo_ + off);                                                  \
      dmask |= (ok ? 1u : 0u) << i;                                                                           \
    }                                                                                                         \
  }

  if (t_begin < t_end) SSP_WG_ISSUE(t_begin)
  for (int tile = t_begin; tile < t_end; ++tile) {
    __syncthreads();  // all waves finished reading the previous tile's LDS image
    {
      const int cur_prob = tile >= a.ntiles ? 1 : 0;
      f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
      if (IN_MODE != 0) {  // written before the first barrier above by threads 0..31
        sc = *reinterpret_cast<const f32x4*>(sS + cur_prob * 128 + q16 * 4);
        sh = *reinterpret_cast<const f32x4*>(sS + cur_prob * 128 + 64 + q16 * 4);
      }
      if (IN_MODE != 2) {
#pragma unroll
        for (int i = 0; i < NX; ++i) {
          const int pp = (tid >> 4) + 16 * i;
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
      } else {
        int t = tile - cur_prob * a.ntiles;
        const float* const pin = cur_prob ? a.in2 : a.in;
        const int tx = t % a.tiles_x;
        t /= a.tiles_x;
        const int ty0 = (t % a.tiles_y) * G::TH, tx0 = tx * G::TW, n = t / a.tiles_y;
        const float4 sc4 = make_float4(sc[0], sc[1], sc[2], sc[3]), sh4 = make_float4(sh[0], sh[1], sh[2], sh[3]);
        for (int pp = tid >> 4; pp < G::HT * G::WT; pp += 16) {
          const int r = pp / G::WT, c = pp - r * G::WT;
          const float4 v = load_in<IN_MODE>(pin, n, ty0 + r - G::PAD, tx0 + c - G::PAD, a.H, a.W, a.in_cs,
                                            a.in_co + ci0, civalid, sc4, sh4);
          *reinterpret_cast<float4*>(sX + pp * 64 + q16 * 4) = v;
        }
      }
#pragma unroll
      for (int i = 0; i < ND; ++i) {
        const int pp = (tid >> 4) + 16 * i;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if ((dmask >> i) & 1u) v = dreg[i];
        *reinterpret_cast<f32x4*>(sD + pp * 64 + q16 * 4) = v;
      }
    }
    __syncthreads();
    {
      const int nxt = min(tile + 1, t_end - 1);  // unconditional prefetch (redundant on the last tile)
      SSP_WG_ISSUE(nxt)
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll 4
    for (int s = 0; s < G::P / 2; ++s) {
      const int p = 2 * s + lh;
      const int r = p / G::TW, c = p - r * G::TW;
      const float b = sD[p * 64 + coh * 32 + li];
      const float* xb = sX + (r * G::WT + c) * 64 + cih * 32 + li;
#pragma unroll
      for (int tap = 0; tap < G::TAPS; ++tap) {
        const int dy = tap / KS, dx = tap % KS;
        const float av = xb[(dy * G::WT + dx) * 64];
        acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b, acc[tap], 0, 0, 0);
      }
    }
  }
#undef SSP_WG_ISSUE
  // partial slab: [blk][tap][ci 64][co 64]
  float* dst = a.partial + (size_t)blockIdx.x * G::TAPS * 4096;
#pragma unroll
  for (int tap = 0; tap < G::TAPS; ++tap)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = (r & 3) + 8 * (r >> 2) + 4 * lh;  // ci within the 32-half
      dst[tap * 4096 + (cih * 32 + m) * 64 + coh * 32 + li] = acc[tap][r];
    }
}

// Sums the partial slabs over splits and ACCUMULATES into the OIHW gradient tensor.
// block = 256 threads = 64 outputs x 4 split groups
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw,
                                                           int Cin, int Cout, int KS, int ncob, int nsplit) {
  __shared__ float red[256];
  const int taps = KS * KS;
  const int total = Cout * Cin * taps;
  const int o = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int idx = blockIdx.x * 64 + o;
  float s = 0.f;
  int co = 0, ci = 0, tap = 0;
  if (idx < total) {
    // idx enumerates (tap, ci, co) with co fastest so that reads of `partial` coalesce
    co = idx % Cout;
    ci = (idx / Cout) % Cin;
    tap = idx / (Cout * Cin);
    const int cob = co >> 6, cib = ci >> 6;
    const float* src = partial + ((size_t)((cib * ncob + cob) * nsplit) * taps + tap) * 4096 + (ci & 63) * 64 + (co & 63);
    for (int k = grp; k < nsplit; k += 4) s += src[(size_t)k * taps * 4096];
  }
  red[threadIdx.x] = s;
  __syncthreads();
  if (grp == 0 && idx < total)
    dw[((size_t)co * Cin + ci) * taps + tap] += red[o] + red[64 + o] + red[128 + o] + red[192 + o];
}

// The same reduction for SEVERAL weight gradients in one launch (the bf16 path defers the reductions of a backward pass: 13 launches
// of ~20 us each were 0.28 ms of a 7.1 ms step; one launch also fills the chip).  blocks -> job by the table's block prefix.
constexpr int WREDB_MAX_JOBS = 24;
struct WredBJob {
  const float* partial;
  float* dw;
  int cin, cout, ks, ncob, nsplit;
  int block0;   // first block of the job; a job has ceil(cout * cin * ks * ks / 64) blocks
};
struct WredBJobs {
  int n;
  WredBJob j[WREDB_MAX_JOBS];
};
__global__ __launch_bounds__(256) void wgrad_reduce_multi_kernel(const WredBJobs J) {
  __shared__ float4 red4[256];
  int k = 0;
#pragma unroll 1
  for (int i = 1; i < J.n; ++i) k = (int)blockIdx.x >= J.j[i].block0 ? i : k;
  const WredBJob& q = J.j[k];
  const int Cin = q.cin, Cout = q.cout, taps = q.ks * q.ks, nsplit = q.nsplit;
  const int total = Cout * Cin * taps;
  const int base = ((int)blockIdx.x - q.block0) * 64;   // 64 consecutive outputs (tap, ci, co), co fastest
  if ((Cout & 63) == 0) {
    // the 64 outputs are ONE 256-byte row of every slab: 16 lanes x float4, 16 split groups with 16-byte loads in flight
    const int o4 = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int idx = base + 4 * o4;
    const int co = idx % Cout, ci = (idx / Cout) % Cin, tap = idx / (Cout * Cin);
    const int cob = co >> 6, cib = ci >> 6;
    const float* src = q.partial + ((size_t)((cib * q.ncob + cob) * nsplit) * taps + tap) * 4096 + (ci & 63) * 64 + (co & 63);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r = grp; r < nsplit; r += 16) {
      const float4 v = *reinterpret_cast<const float4*>(src + (size_t)r * taps * 4096);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    red4[threadIdx.x] = s;
    __syncthreads();
    if (grp == 0) {
      float4 t = red4[o4];
#pragma unroll
      for (int g = 1; g < 16; ++g) { const float4 v = red4[g * 16 + o4]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
      const float tv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) q.dw[((size_t)(co + e) * Cin + ci) * taps + tap] += tv[e];
    }
    return;
  }
  float* const red = reinterpret_cast<float*>(red4);
  const int o = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int idx = base + o;
  float s = 0.f;
  int co = 0, ci = 0, tap = 0;
  if (idx < total) {
    co = idx % Cout;
    ci = (idx / Cout) % Cin;
    tap = idx / (Cout * Cin);
    const int cob = co >> 6, cib = ci >> 6;
    const float* src = q.partial + ((size_t)((cib * q.ncob + cob) * nsplit) * taps + tap) * 4096 + (ci & 63) * 64 + (co & 63);
    for (int r = grp; r < nsplit; r += 4) s += src[(size_t)r * taps * 4096];
  }
  red[threadIdx.x] = s;
  __syncthreads();
  if (grp == 0 && idx < total)
    q.dw[((size_t)co * Cin + ci) * taps + tap] += red[o] + red[64 + o] + red[128 + o] + red[192 + o];
}

// Packs OIHW weights into the LDS image of conv_mfma_kernel: [cob][chunk][tap][g][h][64][4].
// transpose_flip: build the data-gradient convolution (input channels = Cout_w, output = Cin_w, taps mirrored).
__global__ void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ dst, int Cout_w, int Cin_w, int KS,
                                    int transpose_flip, int nchunks_total, int chunk_off, int cob_off, int ncob,
                                    int nchunks) {
  const int taps = KS * KS;
  const int per_chunk = taps * CK * NB;
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
  const int tap = t % taps;
  t /= taps;
  const int chunk = t % nchunks;
  const int cob = t / nchunks;
  const int co = cob * NB + nn;                 // conv output channel
  const int ci = chunk * CK + g * 8 + h * 4 + e;  // conv input channel
  const int ky = tap / KS, kx = tap % KS;
  float v = 0.f;
  if (!transpose_flip) {
    if (co < Cout_w && ci < Cin_w) v = w[(((size_t)co * Cin_w + ci) * KS + ky) * KS + kx];
  } else {
    if (co < Cin_w && ci < Cout_w) v = w[(((size_t)ci * Cin_w + co) * KS + (KS - 1 - ky)) * KS + (KS - 1 - kx)];
  }
  dst[((size_t)(cob + cob_off) * nchunks_total + chunk + chunk_off) * per_chunk +
      (((tap * (CK / 8) + g) * 2 + h) * NB + nn) * 4 + e] = v;
}

}  // namespace sspk
