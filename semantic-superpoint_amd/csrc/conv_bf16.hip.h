// Direct (implicit-GEMM) 3x3 / 1x1 convolution on the bf16 matrix cores of gfx950 with bf16 NHWC activations in HBM
// (BASELINE configs[3]: "bf16 compute / fp32 master"; conv algorithm 12).  v_mfma_f32_32x32x16_bf16, fp32 accumulation.
//
//   conv_bf16_kernel   forward convolution and data-gradient convolution (= convolution with the flipped / transposed
//                      weight image) of models/unet_parts.py:14-21 and the heads of models/SuperPointNet_gauss2*.py
//
// Operand roles: A = weights (M = 64 output channels of the block, two 32-row tiles), B = pixels (N = 32 pixels per tile:
// two rows of 16 of the 16x16 output tile), K = 16 input channels of one tap per instruction.  A lane therefore holds ONE
// pixel and 16 output channels per accumulator tile: the epilogue rounds channel pairs to bf16 in registers and writes
// 8-byte items into a [pixel][64 channel] LDS tile, which leaves as 128 contiguous bytes per pixel.
// The BatchNorm + ReLU of the PRODUCING layer is applied while the input halo is staged into LDS (raw bf16 y -> fp32 ->
// fma, max -> bf16 operand), padding pixels are zero in the ACTIVATED domain.
// Semantics (= oracle/cpu_ref.py, operand_dtype=torch.bfloat16): operands = bf16(activated input), bf16(weight); products
// accumulated in fp32; the stored output is bf16(acc + bias); BatchNorm statistics are those of the STORED tensor.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "conv_mfma.hip.h"

namespace sspk {

constexpr int CB_T = 16;     // output tile: 16 x 16 pixels = 8 pixel tiles of 2 rows x 16 columns (2 per wave)
constexpr int CB_KC = 32;    // input channels per LDS stage (two MFMA k-steps)
constexpr int CB_PS = 80;    // bytes per halo pixel slot: 64 of data + 16 (20 dwords: 16 pixels with distinct slot index
                             // modulo 16 cover the 64 banks once per ds_read_b128 lane group)
constexpr int CB_NB = 64;    // output channels per work unit
constexpr int CB_OS_BF16 = 144;  // bytes per pixel of the bf16 output tile in LDS (128 + 16)
constexpr int CB_OS_F32 = 272;   // fp32 output half tile (256 + 16)

template <int KS>
struct ConvBGeom {
  static constexpr int TAPS = KS * KS;
  static constexpr int HT = CB_T + KS - 1;             // halo rows / columns
  static constexpr int H_BYTES = HT * HT * CB_PS;
  static constexpr int W_BYTES = TAPS * 2 * 2 * 1024;  // [tap][k-step][m-tile][lane] x 16 bytes
  static constexpr int O_BYTES = 256 * CB_OS_BF16;     // >= 128 * CB_OS_F32
  static constexpr int STAGE_BYTES = H_BYTES + W_BYTES;
  static constexpr int LDS_BYTES = (STAGE_BYTES > O_BYTES ? STAGE_BYTES : O_BYTES);
};

struct ConvBArgs {
  const void* in[2];         // NHWC [N,H,W,in_cs], bf16 (or fp32: IN_F32) per view
  const uint16_t* wpk;       // pack_weights_bf16_kernel: [cob][chunk32][tap][kstep][mtile][lane][8] bf16
  const float* bias;         // [Cout] or nullptr
  void* out[2];              // NHWC [N,H,W,out_cs], bf16 (or fp32: OUT_F32)
  const float* in_scale[2];  // [Cin] BatchNorm affine of the producing layer (IN_MODE 1)
  const float* in_shift[2];
  double* stats[2];          // [NREP][2 Cout] sum, sum of squares of the stored output, or nullptr
  uint16_t* pool_out[2];     // [N,H/2,W/2,Cout] raw pooled copy (per-channel max for gamma >= 0, min for gamma < 0) or nullptr
  const float* pool_gamma;
  // Fused BatchNorm-backward reduction (conv_bf16_ws_kernel as a DATA GRADIENT only): the output of this launch is dOut of the layer
  // below, whose raw (pooled) bf16 output t sits at the same pixels.  The copy-out accumulates that layer's pass-1 sums of the
  // STORED values, S1 = sum dz, S2 = sum dz xhat with dz = [t scale + shift > 0] dOut, xhat = (t - mean) invstd, into bnr_sums
  // ([NREP][2 Cout] doubles) - the work of bn_bwd_kernel<RELU, false, false, bf16> without its launch and its read of dOut.
  const uint16_t* bnr_t[2];  // [N,H,W,Cout] or nullptr (no fusion)
  const float* bnr_scale[2]; const float* bnr_shift[2]; const float* bnr_mean[2]; const float* bnr_invstd[2];
  double* bnr_sums[2];
  int nviews, N, H, W;
  int Cin, in_cs, in_co;
  int Cout, out_cs, out_co;
  int tiles_x, tiles_y, nchunks, ncob;
  unsigned in_img_bytes;     // bytes of ONE input image (buffer descriptor range)
  int ablate;                // perf-debug only (SSP_CONVB_ABLATE): 1 no global loads, 2 no LDS staging writes, 4 no epilogue, 8 no MFMA loop
  unsigned long long* trace; // perf-debug only (SSP_CONVB_TRACE=1): per-phase cycle sums of workgroup 0 (conv_bf16_ws_kernel), else nullptr
};

// bias of the 64 output channels of block `cob` in the accumulator layout (bias4[mt][q][e] = channel cob * 64 + mt * 32 + 8 q +
// 4 lg + e): 32 branch-free dword buffer loads (channels past Cout and a null bias read 0 through the descriptor), ONE wait.  (The
// first form - a 16-byte load when aligned, else four guarded scalar loads - compiled to eight load / s_waitcnt vmcnt(0) pairs
// inside exec branches: 8 serial L2 latencies per tile, 0.10 of the 0.52 ms of the 64 -> 64 layer @240x320.)
__device__ __forceinline__ void cb_load_bias(const float* bias, int Cout, int cob, int lg, f32x4 (&bias4)[2][4]) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bias), 0, bias != nullptr ? Cout * 4 : 0, 0x00020000);
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        bias4[mt][q][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (cob * CB_NB + mt * 32 + 8 * q + 4 * lg + e) * 4, 0, 0));
}

// lane (0..31) of a pixel tile -> (row 0..1, column 0..15).  With RS = halo row pitch (18 for 3x3, 16 for 1x1) the slot index
// r * RS + c of the 16 lanes of each ds_read_b128 lane group ({0-3,12-15,20-27}, {4-11,16-19,28-31}) is distinct modulo 16.
template <int RS>
__device__ __forceinline__ void cb_lane_pixel(int j, int& r, int& c) {
  int k;
  if (j < 4) { r = 0; k = j; }
  else if (j < 12) { r = 0; k = j + 4; }
  else if (j < 16) { r = 0; k = j - 8; }
  else if (j < 20) { r = 1; k = j - 16; }
  else if (j < 28) { r = 1; k = j - 12; }
  else { r = 1; k = j - 24; }
  c = (k - (RS & 15) * r) & 15;
}

// IN_MODE 0: plain input; 1: BatchNorm + ReLU of the producing layer on load.  IN_F32 / OUT_F32: fp32 tensors at that end.
template <int KS, int IN_MODE, bool IN_F32, bool OUT_F32>
__global__ __launch_bounds__(256, 2) void conv_bf16_kernel(const ConvBArgs a) {
  using G = ConvBGeom<KS>;
  constexpr int HT = G::HT, PAD = KS / 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
  unsigned char* const sH = smem_b;
  unsigned char* const sW = smem_b + G::H_BYTES;
  unsigned char* const sO = smem_b;
  float* const s_red = reinterpret_cast<float*>(smem_b);  // [256][17] of the statistics flush (between two units)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lj = lane & 31, lg = lane >> 5;

  // ---- work assignment: contiguous unit range per XCD, blocks of an XCD interleaved ----
  const int T = a.N * a.tiles_y * a.tiles_x;
  const int U = a.nviews * a.ncob * T;
  const int nslot = gridDim.x >> 3, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int per = (U + 7) >> 3;
  const int u_end = min(U, (xcd + 1) * per);
  int u = xcd * per + slot;
  if (u >= u_end) return;

  int pr, pc;
  cb_lane_pixel<HT>(lj, pr, pc);
  // byte offset of this lane's two pixels (tap (0,0)) in the halo image, + k-half
  int boff[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) boff[nt] = ((4 * wave + 2 * nt + pr) * HT + pc) * CB_PS + lg * 16;
  const int aoff = lane * 16;

  // staging slots of this thread: halo slot = (tid >> 2) + 64 i, 16-byte part = tid & 3 (8 channels)
  constexpr int NHS = (HT * HT * 4 + 255) / 256;
  const int part = tid & 3;
  const int hs_lds0 = (tid >> 2) * CB_PS + part * 16;   // LDS offset of slot i: + i * 64 * CB_PS (an immediate)
  constexpr int NWS = G::W_BYTES / 16 / 256;  // weight items per thread (9 for 3x3, 1 for 1x1)

  constexpr int CPT = OUT_F32 ? 4 : 8;   // output channels per copy-out item (16 bytes)
  constexpr int TPP = CB_NB / CPT;       // threads per pixel
  float st_s[CPT], st_q[CPT];
#pragma unroll
  for (int e = 0; e < CPT; ++e) st_s[e] = st_q[e] = 0.f;
  int st_key = -1;

  auto flush_stats = [&](int key) {
    // block reduction of the per-thread sums over the threads with the same channel item, then fp64 atomics
    const int view = key / a.ncob, cob = key - view * a.ncob;
    const bool grad = a.bnr_t[0] != nullptr;   // fused BatchNorm-backward sums (below): S1, S2 of the layer below instead of statistics
    double* const p_stats = grad ? a.bnr_sums[view] : a.stats[view];
    __syncthreads();
#pragma unroll
    for (int e = 0; e < CPT; ++e) {
      s_red[tid * 17 + e] = st_s[e];
      s_red[tid * 17 + 8 + e] = st_q[e];
      st_s[e] = st_q[e] = 0.f;
    }
    __syncthreads();
    if (tid < 128) {
      const int ch = tid & 63, which = tid >> 6;   // channel of the block, 0 = sum / 1 = sum of squares
      const int item = ch / CPT, e = ch - item * CPT;
      double t = 0.0;
      for (int k = item; k < 256; k += TPP) t += (double)s_red[k * 17 + which * 8 + e];
      const int co = cob * CB_NB + ch;
      if (co < a.Cout && p_stats != nullptr)
        acc_add_stats_or_grad(p_stats + (size_t)(blockIdx.x % NREP) * 2 * a.Cout + which * a.Cout + co, t, grad);
    }
  };

  // ---- software pipeline over the (unit, chunk) stages of this block: the global loads of stage s + 1 (halo, weight image,
  // BatchNorm affine) are issued right after stage s has been written to LDS and stay in flight under its MFMAs and epilogue ----
  constexpr int IN_ES = IN_F32 ? 4 : 2;
  constexpr unsigned OOB = 0x80000000u;
  // descriptors of the stage whose loads are in flight / being staged ("ld_"), decoded from a unit index.  The slot
  // coordinates of this thread inside a halo are tile independent: per slot one base offset (relative to the tile origin) and
  // its (row, column); a tile adds its origin, and only tiles that touch the image border test the coordinates.
  int ld_view = 0, ld_cob = 0, ld_n = 0, ld_ty0 = 0, ld_tx0 = 0, ld_vc = 0;
  unsigned hs_g[NHS];     // byte offset of slot i inside the image (OOB marker: outside the image / unused slot)
  int hs_rel[NHS];        // ((hy - PAD) * W + (hx - PAD)) * in_cs * es + channel part: offset relative to the tile origin
  int hs_yx[NHS];         // (row << 16) | (column & 0xffff) relative to the tile origin; row -30000: unused slot of the last round
#pragma unroll
  for (int i = 0; i < NHS; ++i) {
    const int s = (tid >> 2) + 64 * i;
    const int hy = s / HT, hx = s - hy * HT;
    hs_yx[i] = ((s < HT * HT ? hy - PAD : -30000) << 16) | ((hx - PAD) & 0xffff);
    hs_rel[i] = (((hy - PAD) * a.W + (hx - PAD)) * a.in_cs + a.in_co + part * 8) * IN_ES;
  }
  __amdgpu_buffer_rsrc_t rsrc_in;
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t*>(a.wpk), 0, (unsigned)(a.ncob * a.nchunks * G::W_BYTES), 0x00020000);
  auto decode = [&](int uu) {
    const int vc = uu / T, t = uu - vc * T;
    ld_vc = vc;
    ld_view = vc / a.ncob; ld_cob = vc - ld_view * a.ncob;
    const int txi = t % a.tiles_x, t2 = t / a.tiles_x;
    const int tyi = t2 % a.tiles_y;
    ld_n = t2 / a.tiles_y;
    ld_ty0 = tyi * CB_T; ld_tx0 = txi * CB_T;
    rsrc_in = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(a.in[ld_view])) + (size_t)ld_n * a.in_img_bytes, 0, a.in_img_bytes, 0x00020000);
    const int org = (ld_ty0 * a.W + ld_tx0) * a.in_cs * IN_ES;
    const bool interior = ld_ty0 >= PAD && ld_tx0 >= PAD && ld_ty0 + CB_T + PAD <= a.H && ld_tx0 + CB_T + PAD <= a.W;
    if (interior) {
#pragma unroll
      for (int i = 0; i < NHS; ++i) hs_g[i] = (hs_yx[i] >> 16) > -30000 ? (unsigned)(org + hs_rel[i]) : OOB;
    } else {
#pragma unroll
      for (int i = 0; i < NHS; ++i) {
        const int gy = ld_ty0 + (hs_yx[i] >> 16), gx = ld_tx0 + (short)(hs_yx[i] & 0xffff);
        const bool ok = (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;   // (unused slots: gy < 0)
        hs_g[i] = ok ? (unsigned)(org + hs_rel[i]) : OOB;
      }
    }
  };
  u32x4 hv[NHS], hv2[IN_F32 ? NHS : 1], wv[NWS];
  float sc[8], sh[8];
  auto issue = [&](int chunk) {
    const int c0 = chunk * CB_KC + part * 8;
#pragma unroll
    for (int i = 0; i < NHS; ++i) {
      const unsigned vo = (c0 < a.Cin && !(a.ablate & 1)) ? hs_g[i] : OOB;
      hv[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_in, vo, chunk * CB_KC * IN_ES, 0));
      if constexpr (IN_F32) hv2[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_in, vo, chunk * CB_KC * IN_ES + 16, 0));
    }
    const int wbase = (ld_cob * a.nchunks + chunk) * G::W_BYTES;   // (wave-uniform: scalar offset of the buffer load)
#pragma unroll
    for (int i = 0; i < NWS; ++i) wv[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, tid * 16, wbase + 4096 * i, 0));
    if (IN_MODE == 1) {
      const float* const p_scale = a.in_scale[ld_view];
      const float* const p_shift = a.in_shift[ld_view];
      if (c0 + 8 <= a.Cin && ((reinterpret_cast<uintptr_t>(p_scale + c0) | reinterpret_cast<uintptr_t>(p_shift + c0)) & 15) == 0) {
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(p_scale + c0), s1 = *reinterpret_cast<const f32x4*>(p_scale + c0 + 4);
        const f32x4 h0 = *reinterpret_cast<const f32x4*>(p_shift + c0), h1 = *reinterpret_cast<const f32x4*>(p_shift + c0 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { sc[e] = s0[e]; sc[4 + e] = s1[e]; sh[e] = h0[e]; sh[4 + e] = h1[e]; }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int c = min(c0 + e, a.Cin - 1);
          sc[e] = p_scale[c];
          sh[e] = p_shift[c];
        }
      }
    }
  };
  decode(u);
  issue(0);
  if ((a.ablate & 16) && blockIdx.x >= (gridDim.x >> 1)) {  // (perf-debug: start the second workgroup of a CU half a tile late)
    for (int i = 0; i < (a.ablate >> 8); ++i) __builtin_amdgcn_s_sleep(127);
  }
  int chunk = 0;
  f32x16 acc[2][2];
  for (;;) {
    // the stage being staged and computed now: unit u, `chunk`
    const int view = ld_view, cob = ld_cob, n = ld_n, ty0 = ld_ty0, tx0 = ld_tx0;
    if (chunk == 0) {
      if ((a.stats[0] != nullptr || a.bnr_t[0] != nullptr) && ld_vc != st_key) {
        if (st_key >= 0) flush_stats(st_key);
        st_key = ld_vc;
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    }
    {
      // ---- stage the halo of 32 channels and the 64 x 32 x taps weight image ----
      const int c0 = chunk * CB_KC + part * 8;
      const bool cfull = c0 + 8 <= a.Cin;
      __syncthreads();  // every wave has finished reading the previous stage (or the output tile of the previous unit)
      if (!(a.ablate & 2)) {
#pragma unroll
      for (int i = 0; i < NHS; ++i) {
        if ((tid >> 2) + 64 * i >= HT * HT) continue;   // (only the last round has unused slots)
        // branch-free: transform whatever the (possibly out-of-range, then zero) load returned, select zero for padding
        u32x4 o;
        float f[8];
        if constexpr (IN_F32) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            f[e] = u32_as_f32(hv[i][e]);
            f[4 + e] = u32_as_f32(hv2[i][e]);
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            f[2 * e] = bf16_lo(hv[i][e]);
            f[2 * e + 1] = bf16_hi(hv[i][e]);
          }
        }
        if (IN_MODE == 1) {
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] = fmaxf(fmaf(f[e], sc[e], sh[e]), 0.f);
        }
        if (IN_F32 || IN_MODE == 1) {
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = pack_bf16(f[2 * e], f[2 * e + 1]);
        } else {
          o = hv[i];
        }
        if (!cfull) {  // ragged input channels (the pointwise data gradients): zero beyond Cin
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (c0 + 2 * e >= a.Cin) o[e] = 0u;
            else if (c0 + 2 * e + 1 >= a.Cin) o[e] &= 0xffffu;
          }
        }
        if (IN_MODE == 1) {  // padding is zero in the ACTIVATED domain (an out-of-range load returned 0, relu(shift) need not be)
          const bool pad = hs_g[i] == OOB || c0 >= a.Cin;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = pad ? 0u : o[e];
        }
        *reinterpret_cast<u32x4*>(sH + hs_lds0 + i * 64 * CB_PS) = o;
      }
#pragma unroll
      for (int i = 0; i < NWS; ++i) *reinterpret_cast<u32x4*>(sW + (tid + 256 * i) * 16) = wv[i];
      }
      __syncthreads();
      // ---- the next stage's loads: next chunk of this unit, or chunk 0 of the block's next unit ----
      const bool last_chunk = chunk + 1 == a.nchunks;
      const bool has_next = !last_chunk || u + nslot < u_end;
      if (has_next) {
        if (last_chunk) decode(u + nslot);
        issue(last_chunk ? 0 : chunk + 1);
        __builtin_amdgcn_sched_barrier(0);  // (the loads are requested BEFORE the MFMA loop)
      }
      // ---- MFMA: taps x 2 k-steps x (2 channel tiles x 2 pixel tiles).  The four operand fragments of step s + 1 are requested
      // before the four MFMAs of step s are issued (two register sets, pinned with sched_barrier). ----
      constexpr int NSTEP = G::TAPS * 2;
      s16x8 fa[2][2], fb[2][2];
      auto fetch = [&](int step, s16x8 (&qa)[2], s16x8 (&qb)[2]) {
        const int tap = step >> 1, ks = step & 1;
        const int dy = tap / KS, dx = tap % KS;
        qa[0] = *reinterpret_cast<const s16x8*>(sW + aoff + ((tap * 2 + ks) * 2 + 0) * 1024);
        qa[1] = *reinterpret_cast<const s16x8*>(sW + aoff + ((tap * 2 + ks) * 2 + 1) * 1024);
        qb[0] = *reinterpret_cast<const s16x8*>(sH + boff[0] + (dy * HT + dx) * CB_PS + ks * 32);
        qb[1] = *reinterpret_cast<const s16x8*>(sH + boff[1] + (dy * HT + dx) * CB_PS + ks * 32);
      };
      if (!(a.ablate & 8)) {
      fetch(0, fa[0], fb[0]);
#pragma unroll
      for (int step = 0; step < NSTEP; ++step) {
        const int cur = step & 1;
        if (step + 1 < NSTEP) fetch(step + 1, fa[cur ^ 1], fb[cur ^ 1]);
        __builtin_amdgcn_sched_barrier(0);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][0], fb[cur][0], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][0], fb[cur][1], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][1], fb[cur][0], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][1], fb[cur][1], acc[1][1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      }
      if (!last_chunk) { ++chunk; continue; }
      chunk = 0;
      if (!has_next) u = u_end; else u += nslot;   // (the descriptors "ld_" already belong to the next unit)
    }
    const bool done = u >= u_end;
    if (a.ablate & 4) { if (done) break; continue; }

    // ---- epilogue: bias, rounding, [pixel][channel] tile through LDS, 16-byte stores, statistics of the stored values ----
    f32x4 bias4[2][4];
    cb_load_bias(a.bias, a.Cout, cob, lg, bias4);
    unsigned char* const p_out = reinterpret_cast<unsigned char*>(a.out[view]);
    const bool do_stats = a.stats[0] != nullptr;
    constexpr int NROUND = OUT_F32 ? 2 : 1;
#pragma unroll
    for (int round = 0; round < NROUND; ++round) {
      __syncthreads();  // MFMA reads of the stage (round 0) / copy-out of the previous half (round 1) are done
      if (!OUT_F32) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const int lp = (4 * wave + 2 * nt + pr) * 16 + pc;
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              u32x2 o;
              o[0] = pack_bf16(acc[mt][nt][4 * q] + bias4[mt][q][0], acc[mt][nt][4 * q + 1] + bias4[mt][q][1]);
              o[1] = pack_bf16(acc[mt][nt][4 * q + 2] + bias4[mt][q][2], acc[mt][nt][4 * q + 3] + bias4[mt][q][3]);
              *reinterpret_cast<u32x2*>(sO + lp * CB_OS_BF16 + (mt * 32 + 8 * q + 4 * lg) * 2) = o;
            }
        }
      } else {
        const int lp = wave * 32 + lj;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (round == 0 ? acc[mt][0][4 * q + e] : acc[mt][1][4 * q + e]) + bias4[mt][q][e];
            *reinterpret_cast<f32x4*>(sO + lp * CB_OS_F32 + (mt * 32 + 8 * q + 4 * lg) * 4) = o;
          }
      }
      __syncthreads();
      const int item = tid % TPP;                 // 16-byte item of the pixel's 64 channels
      const int co0 = cob * CB_NB + item * CPT;
      const int nvalid = min(CPT, a.Cout - co0);  // <= 0: none
      constexpr int NPX = OUT_F32 ? 128 : 256;
      constexpr int PSTEP = 256 / TPP;
      if constexpr (!OUT_F32) {
        // item k of this thread: pixel row (tid >> 7) + 2 k, column (tid >> 3) & 15 of the tile: one voffset, the row step is a
        // scalar offset of the buffer store; rows below the image fall off the end of the per-image descriptor
        const int col = (tid >> 3) & 15, row0 = tid >> 7;
        const unsigned out_img_bytes = (unsigned)a.H * a.W * a.out_cs * 2u;
        const __amdgpu_buffer_rsrc_t rsrc_out = __builtin_amdgcn_make_buffer_rsrc(p_out + (size_t)n * out_img_bytes, 0, out_img_bytes, 0x00020000);
        const bool col_ok = nvalid > 0 && tx0 + col < a.W;
        const unsigned vo = col_ok ? (unsigned)((((ty0 + row0) * a.W + tx0 + col) * a.out_cs + a.out_co + co0) * 2) : OOB;
        const int rstep = 2 * a.W * a.out_cs * 2;
        const bool full = ty0 + CB_T <= a.H;
        f32x2 ps[4], pq[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { ps[e] = f32x2{st_s[2 * e], st_s[2 * e + 1]}; pq[e] = f32x2{st_q[2 * e], st_q[2 * e + 1]}; }
        // Fused BatchNorm-backward reduction (ConvBArgs::bnr_*, as in conv_bf16_ws_kernel: this launch is a data gradient, its output
        // is dOut of the layer below, whose raw bf16 output t has the geometry of the output tensor): S1 += dz, S2 += dz xhat of the
        // STORED values, dz = [t scale + shift > 0] dOut, xhat = (t - mean) invstd.  t in two batches of four rows.
        const bool do_bnr = a.bnr_t[0] != nullptr;
        f32x2 b_sc[4], b_sh[4], b_mu[4], b_is[4];
        u32x4 tv[4];
        __amdgpu_buffer_rsrc_t rsrc_t = rsrc_out;
        if (do_bnr) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int c = min(co0 + e, a.Cout - 1);
            b_sc[e >> 1][e & 1] = a.bnr_scale[view][c]; b_sh[e >> 1][e & 1] = a.bnr_shift[view][c];
            b_mu[e >> 1][e & 1] = a.bnr_mean[view][c]; b_is[e >> 1][e & 1] = a.bnr_invstd[view][c];
          }
          rsrc_t = __builtin_amdgcn_make_buffer_rsrc(
              const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(a.bnr_t[view])) + (size_t)n * out_img_bytes, 0, out_img_bytes, 0x00020000);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int lp = (tid >> 3) + 32 * k;
          if (do_bnr && (k & 3) == 0) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) tv[kk] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_t, vo, rstep * (k + kk), 0));
          }
          u32x4 v = *reinterpret_cast<const u32x4*>(sO + lp * CB_OS_BF16 + item * 16);
          ssp_store_b128(v, rsrc_out, vo, rstep * k);
          if (do_bnr) {
            const bool ok = col_ok && (full || ty0 + row0 + 2 * k < a.H);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const uint32_t w = ok ? v[e] : 0u, yb = tv[k & 3][e];
              const f32x2 d2 = {bf16_lo(w), bf16_hi(w)}, y2 = {bf16_lo(yb), bf16_hi(yb)};
              const f32x2 z2 = pk_fma(y2, b_sc[e], b_sh[e]);
              const f32x2 dz = {z2[0] > 0.f ? d2[0] : 0.f, z2[1] > 0.f ? d2[1] : 0.f};
              const f32x2 xh = pk_mul(pk_sub(y2, b_mu[e]), b_is[e]);
              ps[e] = pk_add(ps[e], dz);
              pq[e] = pk_fma(dz, xh, pq[e]);
            }
          } else if (do_stats) {
            const bool ok = col_ok && (full || ty0 + row0 + 2 * k < a.H);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const uint32_t w = ok ? v[e] : 0u;
              const f32x2 f = {bf16_lo(w), bf16_hi(w)};
              ps[e] = pk_add(ps[e], f);
              pq[e] = pk_fma(f, f, pq[e]);
            }
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { st_s[2 * e] = ps[e][0]; st_s[2 * e + 1] = ps[e][1]; st_q[2 * e] = pq[e][0]; st_q[2 * e + 1] = pq[e][1]; }
      } else {
#pragma unroll 4
      for (int k = 0; k < NPX / PSTEP; ++k) {
        const int lp = tid / TPP + PSTEP * k;
        int rr, cc;
        cb_lane_pixel<HT>(lp & 31, rr, cc);
        const int row = 4 * (lp >> 5) + 2 * round + rr, col = cc;
        const int oy = ty0 + row, ox = tx0 + col;
        if (nvalid <= 0 || oy >= a.H || ox >= a.W) continue;
        const size_t eo = ((size_t)(n * a.H + oy) * a.W + ox) * a.out_cs + a.out_co + co0;
        const f32x4 v = *reinterpret_cast<const f32x4*>(sO + lp * CB_OS_F32 + item * 16);
        float* p = reinterpret_cast<float*>(p_out) + eo;
        if (nvalid == 4) *reinterpret_cast<f32x4*>(p) = v;
        else {
          p[0] = v[0];
          if (nvalid > 1) p[1] = v[1];
          if (nvalid > 2) p[2] = v[2];
        }
        if (do_stats) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (e < nvalid) { st_s[e] += v[e]; st_q[e] = fmaf(v[e], v[e], st_q[e]); }
        }
      }
      }
      if (!OUT_F32 && a.pool_out[0] != nullptr && nvalid > 0) {
        // raw 2x2-pooled copy: per-channel max (gamma >= 0) or min (gamma < 0) of the window, so that
        // maxpool(relu(bn(y))) == relu(bn(pooled)) bit for bit (scale = gamma * invstd has the sign of gamma)
        uint16_t* const p_pool = a.pool_out[view];
        float gsign[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) gsign[e] = a.pool_gamma[co0 + e];
        const int Hp = a.H >> 1, Wp = a.W >> 1;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int pp = (tid >> 3) + 32 * k;   // pooled pixel of the 8 x 8 pooled tile
          const int py = pp >> 3, px = pp & 7;
          const int oy = (ty0 >> 1) + py, ox = (tx0 >> 1) + px;
          if (oy >= Hp || ox >= Wp) continue;
          const int lp = (2 * py) * 16 + 2 * px;
          const u32x4 v0 = *reinterpret_cast<const u32x4*>(sO + lp * CB_OS_BF16 + item * 16);
          const u32x4 v1 = *reinterpret_cast<const u32x4*>(sO + (lp + 1) * CB_OS_BF16 + item * 16);
          const u32x4 v2 = *reinterpret_cast<const u32x4*>(sO + (lp + 16) * CB_OS_BF16 + item * 16);
          const u32x4 v3 = *reinterpret_cast<const u32x4*>(sO + (lp + 17) * CB_OS_BF16 + item * 16);
          u32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float l0 = bf16_lo(v0[e]), l1 = bf16_lo(v1[e]), l2 = bf16_lo(v2[e]), l3 = bf16_lo(v3[e]);
            const float h0 = bf16_hi(v0[e]), h1 = bf16_hi(v1[e]), h2 = bf16_hi(v2[e]), h3 = bf16_hi(v3[e]);
            const float lo = gsign[2 * e] >= 0.f ? fmaxf(fmaxf(l0, l1), fmaxf(l2, l3)) : fminf(fminf(l0, l1), fminf(l2, l3));
            const float hi = gsign[2 * e + 1] >= 0.f ? fmaxf(fmaxf(h0, h1), fmaxf(h2, h3)) : fminf(fminf(h0, h1), fminf(h2, h3));
            o[e] = (__builtin_bit_cast(uint32_t, lo) >> 16) | (__builtin_bit_cast(uint32_t, hi) & 0xffff0000u);
          }
          *reinterpret_cast<u32x4*>(p_pool + ((size_t)(n * Hp + oy) * Wp + ox) * a.Cout + co0) = o;
        }
      }
    }
    if (done) break;
  }
  if ((a.stats[0] != nullptr || a.bnr_t[0] != nullptr) && st_key >= 0) flush_stats(st_key);
}

// OIHW fp32 weights -> bf16 operand image of conv_bf16_kernel: [cob][chunk32][tap][kstep 2][mtile 2][lane 64][8]:
// lane l holds output channel cob 64 + mtile 32 + (l & 31), input channels chunk 32 + kstep 16 + 8 (l >> 5) + e.
// transpose_flip: the data-gradient convolution (conv input channels = Cout_w, output = Cin_w, taps mirrored).
__global__ void pack_weights_bf16_kernel(const float* __restrict__ w, uint16_t* __restrict__ dst, int Cout_w, int Cin_w, int KS,
                                         int transpose_flip, int nchunks_total, int chunk_off, int ncob, int nchunks) {
  const int taps = KS * KS;
  const int per_chunk = taps * 2 * 2 * 64 * 8;
  const long total = (long)ncob * nchunks * per_chunk;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  long t = idx;
  const int e = (int)(t & 7); t >>= 3;
  const int l = (int)(t & 63); t >>= 6;
  const int mt = (int)(t & 1); t >>= 1;
  const int ks = (int)(t & 1); t >>= 1;
  const int tap = (int)(t % taps); t /= taps;
  const int chunk = (int)(t % nchunks);
  const int cob = (int)(t / nchunks);
  const int co = cob * 64 + mt * 32 + (l & 31);
  const int ci = chunk * 32 + ks * 16 + (l >> 5) * 8 + e;
  const int ky = tap / KS, kx = tap % KS;
  float v = 0.f;
  if (!transpose_flip) {
    if (co < Cout_w && ci < Cin_w) v = w[(((size_t)co * Cin_w + ci) * KS + ky) * KS + kx];
  } else {
    if (co < Cin_w && ci < Cout_w) v = w[(((size_t)ci * Cin_w + co) * KS + (KS - 1 - ky)) * KS + (KS - 1 - kx)];
  }
  const uint32_t pk = pack_bf16(v, 0.f);
  dst[((size_t)cob * nchunks_total + chunk + chunk_off) * per_chunk + (idx % per_chunk)] = (uint16_t)(pk & 0xffffu);
}

// every bf16 operand image of a step in ONE launch (26 images per step for SuperPointNet_gauss2_ssmall: one launch of ~8 us each
// was 0.2 ms of a 7.1 ms step): a table of jobs by value, blocks -> job by the table's block prefix
constexpr int PACKB_MAX_JOBS = 40;
struct PackBJob {
  const float* w;
  uint16_t* dst;
  int cout_w, cin_w, ks, tf, nchunks_total, chunk_off, ncob, nchunks;
  int block0;   // first block of the job
};
struct PackBJobs {
  int n;
  PackBJob j[PACKB_MAX_JOBS];
};

__global__ void pack_weights_bf16_multi_kernel(const PackBJobs J) {
  int k = 0;
#pragma unroll 1
  for (int i = 1; i < J.n; ++i) k = (int)blockIdx.x >= J.j[i].block0 ? i : k;
  const PackBJob& q = J.j[k];
  const int KS = q.ks, taps = KS * KS;
  const int per_chunk = taps * 2 * 2 * 64 * 8;
  const long total = (long)q.ncob * q.nchunks * per_chunk;
  const long idx = (long)(blockIdx.x - q.block0) * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  long t = idx;
  const int e = (int)(t & 7); t >>= 3;
  const int l = (int)(t & 63); t >>= 6;
  const int mt = (int)(t & 1); t >>= 1;
  const int ks = (int)(t & 1); t >>= 1;
  const int tap = (int)(t % taps); t /= taps;
  const int chunk = (int)(t % q.nchunks);
  const int cob = (int)(t / q.nchunks);
  const int co = cob * 64 + mt * 32 + (l & 31);
  const int ci = chunk * 32 + ks * 16 + (l >> 5) * 8 + e;
  const int ky = tap / KS, kx = tap % KS;
  float v = 0.f;
  if (!q.tf) {
    if (co < q.cout_w && ci < q.cin_w) v = q.w[(((size_t)co * q.cin_w + ci) * KS + ky) * KS + kx];
  } else {
    if (co < q.cin_w && ci < q.cout_w) v = q.w[(((size_t)ci * q.cin_w + co) * KS + (KS - 1 - ky)) * KS + (KS - 1 - kx)];
  }
  const uint32_t pk = pack_bf16(v, 0.f);
  q.dst[((size_t)cob * q.nchunks_total + chunk + q.chunk_off) * per_chunk + (idx % per_chunk)] = (uint16_t)(pk & 0xffffu);
}

// a = bf16(relu(bn(y))) of a dense bf16 NHWC tensor, materialised ONCE for an input that many convolution units read: the three 3x3
// heads take the 30x40 output of encoder layer 7 through 12 (view, 64-channel block) units per tile, each of which would load,
// activate and write the same halo.  Same arithmetic as the on-load transform (one fp32 fma, round to nearest even, ReLU on the
// rounded value), so the heads see identical operands; their launches then run the staging-free form (IN_MODE 0).
__global__ __launch_bounds__(256) void bn_relu_bf16_kernel(const uint16_t* __restrict__ y0, const uint16_t* __restrict__ y1,
                                                           const float* __restrict__ sc0, const float* __restrict__ sh0,
                                                           const float* __restrict__ sc1, const float* __restrict__ sh1,
                                                           uint16_t* __restrict__ o0, uint16_t* __restrict__ o1, long nitems, int C) {
  typedef short s16x2 __attribute__((ext_vector_type(2)));
  const uint16_t* const y = blockIdx.y ? y1 : y0;
  const float* const sc = blockIdx.y ? sc1 : sc0;
  const float* const sh = blockIdx.y ? sh1 : sh0;
  uint16_t* const o = blockIdx.y ? o1 : o0;
  const int ipp = C >> 3;   // 16-byte items per pixel
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nitems; i += (long)gridDim.x * blockDim.x) {
    const int c0 = (int)(i % ipp) * 8;
    const u32x4 v = *reinterpret_cast<const u32x4*>(y + i * 8);
    const f32x4 s0 = *reinterpret_cast<const f32x4*>(sc + c0), s1 = *reinterpret_cast<const f32x4*>(sc + c0 + 4);
    const f32x4 h0 = *reinterpret_cast<const f32x4*>(sh + c0), h1 = *reinterpret_cast<const f32x4*>(sh + c0 + 4);
    const float scv[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
    const float shv[8] = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
    u32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float zl = __builtin_fmaf(bf16_lo(v[e]), scv[2 * e], shv[2 * e]);
      const float zh = __builtin_fmaf(bf16_hi(v[e]), scv[2 * e + 1], shv[2 * e + 1]);
      const s16x2 m = __builtin_elementwise_max(__builtin_bit_cast(s16x2, pack_bf16(zl, zh)), s16x2{0, 0});
      r[e] = __builtin_bit_cast(uint32_t, m);
    }
    *reinterpret_cast<u32x4*>(o + i * 8) = r;
  }
}

}  // namespace sspk
