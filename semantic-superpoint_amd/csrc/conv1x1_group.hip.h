// Grouped pointwise (1x1) convolutions of the heads on v_mfma_f32_32x32x2_f32: convPb 256 -> 65, convDb 256 -> 256, convSout
// 256 -> 133 (models/SuperPointNet_gauss2.py:59-66, SuperPointNet_gauss2_ssmall.py:57-70 of the reference), forward and data
// gradient, BOTH views, in ONE launch.
//
// Why not conv_mfma_kernel<1, ...> (the implicit-GEMM kernel these layers used before): it stages the pixels through LDS in
// 16-channel chunks (32 MFMAs per wave between two barriers), once per 64-channel output block (4x for 256 outputs), rounds
// 65 / 133 outputs up to 128 / 192, and three launches of 0.6-2.3 rounds over the chip each pay their own tail.
//
// This kernel is a plain GEMM, D[pixel][cout] = sum_k A[pixel][k] W[k][cout], with the WEIGHTS STATIONARY in LDS:
//   * a problem = (layer, <= 4 n-tiles = 128 output channels, view); its weight image (<= 256 x 128 floats = 128 KB, packed in
//     operand order [chunk][n-tile][k-quad][lane][4] by pack_g1_kernel) is copied into LDS ONCE by the workgroups assigned to
//     it - one 8-wave workgroup per CU, the CUs shared out between the problems in proportion to their MFMA counts - and read
//     back with one ds_read_b128 per four MFMAs (fetched one k-quad ahead).  After that copy there is no barrier: each wave
//     walks its own 32-pixel tiles (tile = wave index + k x waves of the problem).
//   * a pointwise conv has no halo, so a lane's A operand IS its own pixel's channel vector: loaded straight from global
//     memory into the MFMA operand registers (four 16-byte loads per lane and 32-channel chunk, 64 contiguous bytes per lane -
//     every 128-byte line is consumed by its two half-waves inside the same four instructions), one chunk ahead, across tile
//     boundaries too; BatchNorm + ReLU of the producing layer applied in registers.
// History (same box, forward + data gradient of the three heads): work-queue version with the weights re-staged per 128-pixel
// item through a double-buffered LDS image and one barrier per chunk 0.24 + 0.22 ms - its ablation (G1_ABL) showed 0.17 ms of
// non-MFMA time that did not overlap with the 0.14 ms of MFMAs; weights straight from L2 without LDS 0.34 + 0.39 ms.
// Partial K (65 / 133 input channels of the data gradient): the last chunk runs only the k-quads that hold channels; channels
// past K are masked in registers (the rows of the packed image are zero as well).
// Forward launches accumulate the BatchNorm statistics, data-gradient launches pass 1 of the BatchNorm backward of the layer
// below (per-lane channel sums over the wave's tiles, one cross-wave reduction and one atomic per channel and workgroup).
#pragma once
#include "conv_mfma.hip.h"
#include <type_traits>

#ifndef G1_ABL
#define G1_ABL 0  // compile-time perf ablation (SSP_HIPCC_EXTRA=-DG1_ABL=n): 1 no MFMA, 2 no pixel-operand loads, 4 no stores / no
                  // BatchNorm-backward tensor loads
#endif

namespace sspk {

constexpr int G1_KC = 32;      // input channels per K-chunk
constexpr int G1_NT = 4;       // 32-channel n-tiles per problem
constexpr int G1_PX = 32;      // pixels per wave tile
constexpr int G1_KMAX = 512;   // input channels (scale / shift image in LDS)
constexpr int G1_MAXP = 12;    // problems per launch
constexpr int G1_WAVES = 8;    // waves per workgroup (one workgroup per CU)
constexpr int G1_TILE_FLOATS = 4 * 64 * 4;   // one (chunk, n-tile) of the packed image: [k-quad][lane][4]
constexpr int G1_W_FLOATS = 32 * G1_TILE_FLOATS;  // resident weight image: chunks x n-tiles <= 32 (128 KB)
constexpr int G1_LDS_BYTES = (G1_W_FLOATS + 2 * G1_KMAX) * 4;

struct G1Prob {
  const float* in;        // [npx][in_cs], channels in_co .. in_co + K
  float* out;             // [npx][out_cs], channels out_co .. out_co + N
  const float* wpk;       // packed image of this problem's FIRST n-tile: [chunk][nt_total][4][64][4]
  const float* bias;      // [N] or nullptr
  const float* in_scale;  // [K] (IN_MODE 1)
  const float* in_shift;
  double* stats;          // [NREP][2 stats_c] at this problem's first channel, or nullptr
  int stats_c;            // channels of the whole layer (row pitch of stats)
  // BNR launches (data gradients): `stats` receives pass 1 of the BatchNorm backward of the layer BELOW, whose activation
  // gradient this problem writes - S1 = sum dZ, S2 = sum dZ xhat with dZ = out [y scale + shift > 0], xhat = (y - mean) invstd.
  // bnr_y: that layer's raw conv output, same geometry as `out` ([npx][out_cs], channel out_co); parameters at the problem's
  // first channel.
  const float* bnr_y;
  const float* bnr_scale; const float* bnr_shift; const float* bnr_mean; const float* bnr_invstd;
  int in_cs, in_co, out_cs, out_co;
  int K, N;               // input channels, output channels of THIS problem (<= 128)
  int nchunks, nt, nt_total;
  int npx;
  int wg0, nwg;           // workgroups [wg0, wg0 + nwg) of the launch work on this problem
  unsigned in_bytes, out_bytes;
};
struct G1Args {
  G1Prob p[G1_MAXP];
  int nprob;
};

// All tiles of one wave: tile = gw, gw + tw, ... (32 pixels each) of problem p with NT n-tiles.
template <int IN_MODE, int NT, bool BNR>
__device__ __forceinline__ void g1_run(const G1Prob& p, const float* sW, const float* sSc, float (&ssum)[G1_NT],
                                       float (&ssq)[G1_NT], int gw, int tw, int lane, int li, int lh) {
  constexpr unsigned OOB = 0x80000000u;
  const int nchunks = p.nchunks, K = p.K;
  const int ntiles = (p.npx + G1_PX - 1) / G1_PX;
  if (gw >= ntiles) return;
  const __amdgpu_buffer_rsrc_t rsrc_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_out = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rsrc_y = rsrc_out;
  if (BNR) rsrc_y = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bnr_y), 0, p.out_bytes, 0x00020000);
  const float* const wl = sW + lane * 4;  // + ((chunk * NT + t) * 4 + q) * 256
  const unsigned pitch = (unsigned)p.out_cs * 4u;
  const bool want_stats = p.stats != nullptr;

  f32x4 araw[4];
#define G1_ISSUE_A(VOFF, CHUNK)                                                                                       \
  {                                                                                                                   \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                                   \
      const unsigned vo_ = ((CHUNK) * G1_KC + 16 * lh + 4 * q < K) ? (VOFF) : OOB;                                    \
      if (!(G1_ABL & 2)) araw[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_in, vo_ + q * 16, (CHUNK) * G1_KC * 4, 0)); \
      else araw[q] = f32x4{1.f, 2.f, 3.f, (float)(CHUNK)};                                                           \
    }                                                                                                                 \
  }
#define G1_VOFF(TILE) (((TILE) * G1_PX + li < p.npx) ? (unsigned)((((TILE) * G1_PX + li) * p.in_cs + p.in_co + 16 * lh) * 4) : OOB)
  unsigned voffA = G1_VOFF(gw);
  G1_ISSUE_A(voffA, 0)

  for (int tile = gw; tile < ntiles; tile += tw) {
    const int px0 = tile * G1_PX;
    const bool full = px0 + G1_PX <= p.npx;
    const int rows_left = p.npx - px0 - 4 * lh;  // pixel m of this lane exists while m < rows_left
    float yv[BNR ? 2 : 1][BNR ? 16 : 1];
    // BatchNorm-backward tensor values of n-tile T into yv[SET]: the addresses of that n-tile's stores
#define G1_LOAD_Y(T, SET)                                                                                             \
    {                                                                                                                 \
      unsigned vo_ = (32 * (T) + li < p.N) ? (unsigned)(((px0 + 4 * lh) * p.out_cs + p.out_co + 32 * (T) + li) * 4) : OOB; \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                                \
        const int m_ = (r & 3) + 8 * (r >> 2);                                                                        \
        const unsigned vr_ = (full || m_ < rows_left) ? vo_ : OOB;                                                    \
        if (!(G1_ABL & 4)) yv[SET][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_y, vr_, 0, 0)); \
        else yv[SET][r] = (float)vr_;                                                                                 \
        vo_ += ((r & 3) == 3 ? 5u : 1u) * pitch;                                                                      \
      }                                                                                                               \
    }
    // the pixel operand of the wave's NEXT tile (this one again at the end: redundant loads, but no branch around them)
    const unsigned voffN = tile + tw < ntiles ? G1_VOFF(tile + tw) : voffA;
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    f32x4 bq[2][NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bq[0][t] = *reinterpret_cast<const f32x4*>(wl + (t * 4) * 256);

    for (int chunk = 0; chunk < nchunks; ++chunk) {
      // the lane's 16 channels of this chunk: [32 chunk + 16 lh, + 16)
      float av[16];
      {
        const int kb = chunk * G1_KC + 16 * lh;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 v = araw[q];
          if (IN_MODE != 0) {
            const f32x4 sc = *reinterpret_cast<const f32x4*>(sSc + kb + 4 * q);
            const f32x4 sh = *reinterpret_cast<const f32x4*>(sSc + G1_KMAX + kb + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(fmaf(v[e], sc[e], sh[e]), 0.f);
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) av[4 * q + e] = v[e];
        }
        if (kb + 16 > K) {  // channels past K (the last chunk of a 65 / 133-channel data gradient)
#pragma unroll
          for (int e = 0; e < 16; ++e) av[e] = (kb + e < K) ? av[e] : 0.f;
        }
      }
      if (BNR && chunk + 1 == nchunks) G1_LOAD_Y(0, 0)  // the epilogue's first BatchNorm-backward tensor values, under the last MFMAs
      {  // next chunk of this tile, or chunk 0 of the next tile
        const bool last = chunk + 1 == nchunks;
        const unsigned vn = last ? voffN : voffA;
        const int cn = last ? 0 : chunk + 1;
        G1_ISSUE_A(vn, cn)
      }
      __builtin_amdgcn_sched_barrier(0);  // the loads stay above the MFMAs
      const int nq = min(4, (K - chunk * G1_KC + 3) >> 2);  // k-quads of this chunk that hold channels (of the lh = 0 half)
      const float* const wc = wl + chunk * (NT * G1_TILE_FLOATS);
      const float* const wn = wl + min(chunk + 1, nchunks - 1) * (NT * G1_TILE_FLOATS);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        // fragments of the next k-quad (of the next chunk's first one at q = 3) while this one is multiplied
#pragma unroll
        for (int t = 0; t < NT; ++t)
          bq[(q + 1) & 1][t] = *reinterpret_cast<const f32x4*>((q < 3 ? wc : wn) + (t * 4 + ((q + 1) & 3)) * 256);
        if (q < nq) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
              if (!(G1_ABL & 1)) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[4 * q + e], bq[q & 1][t][e], acc[t], 0, 0, 0);
              else acc[t][e] += av[4 * q + e] * bq[q & 1][t][e];
            }
        }
      }
    }
    voffA = voffN;

    // ---- epilogue: lane = output channel 32 t + li, register r = pixel (r & 3) + 8 (r >> 2) + 4 lh of the wave's 32 ----
    // Addresses advance in a vector register (an out-of-range marker stays out of range under these additions): sixteen scalar
    // row offsets per n-tile would be hoisted and spilled through v_writelane / v_readlane.  Channels past N hold exact zeros
    // (zero weights, no bias), so a full 32-pixel tile needs no per-element validity selects.
    auto tiles = [&](auto FULL_) {  // straight-line per (full 32-pixel tile or not): wave-uniform
      constexpr bool FULL = decltype(FULL_)::value;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int ch = 32 * t + li;
        const bool chok = ch < p.N;
        const float bv = (p.bias != nullptr && chok) ? p.bias[ch] : 0.f;
        const unsigned vbase = chok ? (unsigned)(((px0 + 4 * lh) * p.out_cs + p.out_co + ch) * 4) : OOB;
        float s1 = 0.f, s2 = 0.f;
        float bsc = 0.f, bsh = 0.f, bis = 0.f, bnm = 0.f;
        if (BNR) {
          if (chok) { bsc = p.bnr_scale[ch]; bsh = p.bnr_shift[ch]; bis = p.bnr_invstd[ch]; bnm = -p.bnr_mean[ch] * bis; }
          if (t + 1 < NT) G1_LOAD_Y(t + 1, (t + 1) & 1)  // the next n-tile's values while this one is finished
        }
        unsigned vo = vbase;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = (r & 3) + 8 * (r >> 2);
          const float v = BNR ? acc[t][r] : acc[t][r] + bv;
          const bool ok = FULL || m < rows_left;
          if (!(G1_ABL & 4) || v == 123.456f) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc_out, ok ? vo : OOB, 0, 0);
          vo += ((r & 3) == 3 ? 5u : 1u) * pitch;
          if (BNR) {
            // (a load outside the tensor returned y = 0: the gate then depends on the shift alone, but v of such a position is
            // an exact zero for channels past N and is masked by `ok` for pixels past the end)
            const float dz = (ok && fmaf(yv[t & 1][r], bsc, bsh) > 0.f) ? v : 0.f;
            s1 += dz;
            s2 = fmaf(dz, fmaf(yv[t & 1][r], bis, bnm), s2);
          } else {
            const float vm = (FULL || (ok && chok)) ? v : 0.f;
            s1 += vm;
            s2 = fmaf(vm, vm, s2);
          }
        }
        if (want_stats) { ssum[t] += s1; ssq[t] += s2; }
      }
    };
    if (full) tiles(std::true_type{});
    else tiles(std::false_type{});
#undef G1_LOAD_Y
  }
#undef G1_ISSUE_A
#undef G1_VOFF
}

template <int IN_MODE, bool BNR>
__global__ __launch_bounds__(64 * G1_WAVES) void conv1x1_group_kernel(const G1Args a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const sW = smem;                      // [chunk][nt][4][64][4] of this workgroup's problem
  float* const sSc = smem + G1_W_FLOATS;       // scale[G1_KMAX], shift[G1_KMAX]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  int pi = 0;
  while (pi + 1 < a.nprob && (int)blockIdx.x >= a.p[pi + 1].wg0) ++pi;
  const G1Prob& p = a.p[pi];
  const int w = (int)blockIdx.x - p.wg0;
  if (w >= p.nwg) return;  // (whole workgroup)
  // ---- the problem's weights and input affine become resident ----
  {
    const int per_chunk = p.nt * (G1_TILE_FLOATS / 4);  // float4s
    const int total = p.nchunks * per_chunk;
    for (int i = tid; i < total; i += 64 * G1_WAVES) {
      const int c = i / per_chunk, r = i - c * per_chunk;
      *reinterpret_cast<f32x4*>(sW + (size_t)i * 4) =
          *reinterpret_cast<const f32x4*>(p.wpk + ((size_t)c * p.nt_total * (G1_TILE_FLOATS / 4) + r) * 4);
    }
    if (IN_MODE != 0) {
      for (int k = tid; k < G1_KMAX; k += 64 * G1_WAVES) {
        const bool ok = k < p.K;
        sSc[k] = ok ? p.in_scale[k] : 0.f;
        sSc[G1_KMAX + k] = ok ? p.in_shift[k] : 0.f;
      }
    }
  }
  __syncthreads();

  float ssum[G1_NT], ssq[G1_NT];
#pragma unroll
  for (int t = 0; t < G1_NT; ++t) ssum[t] = ssq[t] = 0.f;
  const int gw = w * G1_WAVES + wave, tw = p.nwg * G1_WAVES;
  switch (p.nt) {
    case 1: g1_run<IN_MODE, 1, BNR>(p, sW, sSc, ssum, ssq, gw, tw, lane, li, lh); break;
    case 2: g1_run<IN_MODE, 2, BNR>(p, sW, sSc, ssum, ssq, gw, tw, lane, li, lh); break;
    case 3: g1_run<IN_MODE, 3, BNR>(p, sW, sSc, ssum, ssq, gw, tw, lane, li, lh); break;
    default: g1_run<IN_MODE, 4, BNR>(p, sW, sSc, ssum, ssq, gw, tw, lane, li, lh); break;
  }
  if (p.stats == nullptr) return;  // (whole workgroup)
  // ---- channel sums: the eight waves through LDS (the weight image is dead), one atomic per channel and workgroup ----
  __syncthreads();
  float* const sRed = smem;  // [wave][G1_NT][32][2]
#pragma unroll
  for (int t = 0; t < G1_NT; ++t) {
    const float s = ssum[t] + __shfl_xor(ssum[t], 32), v = ssq[t] + __shfl_xor(ssq[t], 32);
    if (lh == 0) {
      sRed[((wave * G1_NT + t) * 32 + li) * 2 + 0] = s;
      sRed[((wave * G1_NT + t) * 32 + li) * 2 + 1] = v;
    }
  }
  __syncthreads();
  if (tid < 256) {
    const int ch = tid >> 1, which = tid & 1;  // 128 channels x {sum, second sum}
    if (ch < p.N) {
      float tsum = 0.f;
#pragma unroll
      for (int ww = 0; ww < G1_WAVES; ++ww) tsum += sRed[((ww * G1_NT + (ch >> 5)) * 32 + (ch & 31)) * 2 + which];
      acc_add_stats_or_grad(p.stats + (size_t)(blockIdx.x % NREP) * 2 * p.stats_c + which * p.stats_c + ch, (double)tsum, BNR);
    }
  }
}

// Packed operand image of one pointwise layer: dst[chunk][tile][q][lh][li][j] = W[k = 32 chunk + 16 lh + 4 q + j][n = 32 tile + li],
// W[k][n] = w[n][k] (forward: n = output channel of the OIHW tensor) or w[k][n] (transpose: the data gradient), zero outside.
struct G1PackJob {
  const float* w;
  float* dst;
  int cout_w, cin_w, transpose;
  int nchunks, nt_total;
  int block0;
};
constexpr int G1_PACK_MAX_JOBS = 8;
struct G1PackJobs {
  int n;
  G1PackJob j[G1_PACK_MAX_JOBS];
};
__global__ __launch_bounds__(256) void pack_g1_kernel(const G1PackJobs J) {
  int k = 0;
  while (k + 1 < J.n && (int)blockIdx.x >= J.j[k + 1].block0) ++k;
  const G1PackJob& q = J.j[k];
  const int idx = ((int)blockIdx.x - q.block0) * 256 + threadIdx.x;
  const int total = q.nchunks * q.nt_total * G1_TILE_FLOATS;
  if (idx >= total) return;
  const int j = idx & 3, li = (idx >> 2) & 31, lh = (idx >> 7) & 1, kq = (idx >> 8) & 3;
  const int tile = (idx >> 10) % q.nt_total, chunk = (idx >> 10) / q.nt_total;
  const int kk = G1_KC * chunk + 16 * lh + 4 * kq + j, n = 32 * tile + li;
  const int K = q.transpose ? q.cout_w : q.cin_w, N = q.transpose ? q.cin_w : q.cout_w;
  float v = 0.f;
  if (kk < K && n < N) v = q.transpose ? q.w[(size_t)kk * q.cin_w + n] : q.w[(size_t)n * q.cin_w + kk];
  q.dst[idx] = v;
}

// ------------------------------------------------------------------------------------------------------------------------
// Grouped weight gradient of the pointwise layers: dW[n][k] = sum over the pixels of both views of Xact[px][k] dY[px][n],
// Xact = relu(scale x + shift) of the layer's input (BatchNorm + ReLU on load).  M = input channels, N = output channels, the
// MFMA K dimension runs over PIXELS, so neither operand needs LDS: a lane's A value of K-step s is its channel of pixel
// p + 2 s + lh, a B value its output channel of the same pixel - both straight from global memory, eight K-steps ahead.
//   * The 32 rows of m-tile j of channel group g are the channels 128 g + 4 row + j: a lane's 8-byte load (4 row + 2 (w & 1),
//     + 1) feeds the TWO m-tiles of its wave, and the four waves of a workgroup cover 256 channels with fully used lines.
//   * A part = (layer, <= 4 n-tiles = 128 output channels); every part gets a share of the grid proportional to its MFMA
//     count, each workgroup a contiguous pixel range of one view (static split: one [256][128] partial slab per workgroup,
//     summed into the OIHW gradient by wgrad1x1_reduce_kernel).  Wave = 2 m-tiles x NT n-tiles = <= 128 accumulator registers.
// ------------------------------------------------------------------------------------------------------------------------
constexpr int G1W_MAXP = 8;          // parts per launch
constexpr int G1W_DEPTH = 8;         // K-steps (pixel pairs) in flight per wave
constexpr int G1W_SLAB = 256 * 128;  // floats of a partial slab: [256 input channels][128 output channels]

struct G1WPart {
  const float* x[2];       // [npx][x_cs] raw input of the layer (the conv output of the 3x3 head below), per view
  const float* scale[2];   // BatchNorm + ReLU on load, [256]
  const float* shift[2];
  const float* dy[2];      // [npx][dy_cs], channels dy_co .. dy_co + N of THIS part
  float* dw;               // OIHW gradient [cout][256] of the layer, at this part's first output channel (row n0)
  int x_cs, x_co, dy_cs, dy_co;
  int N, nt;               // output channels / n-tiles of this part
  int npx;
  int wg0, nwg;            // workgroups [wg0, wg0 + nwg) of the launch: the first nwg / 2 on view 0 (nwg for a single view)
  int nviews;
};
struct G1WArgs {
  G1WPart p[G1W_MAXP];
  int nparts;
  float* partial;          // [gridDim.x][256][128]
};

template <int NT>
__device__ __forceinline__ void g1w_run(const G1WPart& p, int view, int px_lo, int px_hi, float* slab, int wave, int li, int lh) {
  constexpr unsigned OOB = 0x80000000u;
  const int g = wave >> 1, jh = wave & 1;
  const int ch0 = 128 * g + 4 * li + 2 * jh;  // this lane's two input channels (rows li of m-tiles 4 g + 2 jh, + 1)
  const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x[view]), 0, (unsigned)((size_t)p.npx * p.x_cs * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_d = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy[view]), 0, (unsigned)((size_t)p.npx * p.dy_cs * 4), 0x00020000);
  const f32x2 sc = *reinterpret_cast<const f32x2*>(p.scale[view] + ch0), sh = *reinterpret_cast<const f32x2*>(p.shift[view] + ch0);
  // lane parts of the addresses of pixel px_lo + lh; a K-step advances two pixels (scalar offsets)
  const unsigned vx = (unsigned)(((px_lo + lh) * p.x_cs + p.x_co + ch0) * 4);
  unsigned vd[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) vd[t] = (32 * t + li < p.N) ? (unsigned)(((px_lo + lh) * p.dy_cs + p.dy_co + 32 * t + li) * 4) : OOB;
  const int sx = 2 * p.x_cs * 4, sd = 2 * p.dy_cs * 4;
  const int nsteps = (px_hi - px_lo + 1) >> 1;
  const int npix = px_hi - px_lo;  // pixel 2 s + lh of the range exists while 2 s + lh < npix

  f32x2 xa[G1W_DEPTH];
  float db[G1W_DEPTH][NT];
#define G1W_ISSUE(U, S)                                                                                      \
  {                                                                                                          \
    const bool ok_ = 2 * (S) + lh < npix;                                                                    \
    xa[U] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrc_x, ok_ ? vx : OOB, (S) * sx, 0)); \
    _Pragma("unroll") for (int t = 0; t < NT; ++t)                                                           \
      db[U][t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_d, ok_ ? vd[t] : OOB, (S) * sd, 0)); \
  }
  // (fenced: in the order of consumption, or the first use of slot 0 - and with it every trip of the loop - would have to wait
  // for ALL loads: hipcc otherwise issues the input-side loads of the eight slots last, slot 0 at the very end)
#pragma unroll
  for (int u = 0; u < G1W_DEPTH; ++u) {
    G1W_ISSUE(u, u)
    asm volatile("" ::: "memory");  // (the scheduler fence alone does not bind the IR passes)
    __builtin_amdgcn_sched_barrier(0);
  }

  f32x16 acc[2][NT];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.f;

  for (int s0 = 0; s0 < nsteps; s0 += G1W_DEPTH) {
#pragma unroll
    for (int u = 0; u < G1W_DEPTH; ++u) {
      // (steps past the range hold zeros in dY: whatever relu(shift) the input side produces there is multiplied by 0)
      const float a0 = fmaxf(fmaf(xa[u][0], sc[0], sh[0]), 0.f), a1 = fmaxf(fmaf(xa[u][1], sc[1], sh[1]), 0.f);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, db[u][t], acc[0][t], 0, 0, 0);
        acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, db[u][t], acc[1][t], 0, 0, 0);
      }
      // The slot is reloaded BEHIND the MFMAs that consumed it (its registers are dead: the loads land in place), seven steps
      // before its next use.  Issued ahead of them, hipcc gives the loads fresh registers and copies them into the ring at the
      // end of the loop body behind an s_waitcnt vmcnt(0).  The fences keep hipcc from moving the loads across MFMA groups.
      __builtin_amdgcn_sched_barrier(0);
      G1W_ISSUE(u, s0 + u + G1W_DEPTH)
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#undef G1W_ISSUE
  // slab[channel][n]: accumulator (m, t), register r, lane (li, lh) = channel 128 g + 4 ((r & 3) + 8 (r >> 2) + 4 lh) + 2 jh + m,
  // output channel 32 t + li; the part's unused n-tiles are never read by the reduction
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
        slab[(128 * g + 4 * row + 2 * jh + m) * 128 + 32 * t + li] = acc[m][t][r];
      }
}

__global__ __launch_bounds__(256, 2) void wgrad1x1_group_kernel(const G1WArgs a) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  int pi = 0;
  while (pi + 1 < a.nparts && (int)blockIdx.x >= a.p[pi + 1].wg0) ++pi;
  const G1WPart& p = a.p[pi];
  const int w = (int)blockIdx.x - p.wg0;
  if (w >= p.nwg) return;
  const int per_view = p.nwg / p.nviews;
  const int view = w / per_view, wv = w - view * per_view;
  // contiguous, even-length pixel ranges (a K-step is a pixel pair)
  const int pairs = (p.npx + 1) >> 1;
  const int lo = (int)(((long)pairs * wv) / per_view) * 2, hi = min(p.npx, (int)(((long)pairs * (wv + 1)) / per_view) * 2);
  float* const slab = a.partial + (size_t)blockIdx.x * G1W_SLAB;
  switch (p.nt) {
    case 1: g1w_run<1>(p, view, lo, hi, slab, wave, li, lh); break;
    case 2: g1w_run<2>(p, view, lo, hi, slab, wave, li, lh); break;
    case 3: g1w_run<3>(p, view, lo, hi, slab, wave, li, lh); break;
    default: g1w_run<4>(p, view, lo, hi, slab, wave, li, lh); break;
  }
}

// dw[n][k] += sum over the part's workgroups of slab[k][n]; block = 256 threads = 8 input channels x 32 output channels
__device__ __forceinline__ void wgrad1x1_reduce_block(const G1WArgs& a, int b) {
  int pi = 0;
  // blocks per part: 32 channel groups x nt
  while (pi + 1 < a.nparts && b >= 32 * a.p[pi].nt) { b -= 32 * a.p[pi].nt; ++pi; }
  const G1WPart& p = a.p[pi];
  if (b >= 32 * p.nt) return;
  const int t = b / 32, kg = b - t * 32;
  const int n = 32 * t + (threadIdx.x & 31), k = 8 * kg + (threadIdx.x >> 5);
  if (n >= p.N) return;
  const float* src = a.partial + (size_t)p.wg0 * G1W_SLAB + k * 128 + n;
  float s0 = 0.f, s1 = 0.f;
  int w = 0;
  for (; w + 1 < p.nwg; w += 2) {
    s0 += src[(size_t)w * G1W_SLAB];
    s1 += src[(size_t)(w + 1) * G1W_SLAB];
  }
  if (w < p.nwg) s0 += src[(size_t)w * G1W_SLAB];
  p.dw[(size_t)n * 256 + k] += s0 + s1;
}
__global__ __launch_bounds__(256) void wgrad1x1_reduce_kernel(const G1WArgs a) { wgrad1x1_reduce_block(a, blockIdx.x); }

}  // namespace sspk
