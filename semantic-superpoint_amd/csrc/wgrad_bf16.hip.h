// Weight gradient of a 3x3 / 1x1 convolution on the bf16 matrix cores (conv algorithm 12, BASELINE configs[3]):
//   dW[co][ci][tap] = sum over pixels of dY[pixel][co] * act(X)[pixel + tap][ci]      (autograd of models/unet_parts.py:14-21)
// v_mfma_f32_32x32x16_bf16 with K = PIXELS: A = dY^T (M = 32 output channels), B = X (N = 32 input channels), one instruction
// per 16 pixels of an image row and tap.  Both tensors are NHWC (channels contiguous), the matrix cores want 8 consecutive K
// per lane: the fragments come out of the [pixel][channel] LDS tiles through ds_read_b64_tr_b16 (a 4 pixel x 16 channel
// block per 16 lanes, delivered transposed), so nothing is transposed in HBM or while staging.
// Workgroup = 4 waves = one 64 co x 64 ci x taps slab (wave = co half x ci half, 9 accumulator tiles), walks 16x16 pixel
// tiles of both views, writes ONE fp32 partial slab; wgrad_reduce_kernel sums the slabs into the OIHW gradient.
// LDS: X halo 18 x 18 pixels x 128 B + dY tile 256 x 128 B; the two 64-byte halves of a pixel slot are swapped when bit 1
// of the slot index is set, which makes the four pixel rows of a transposing read hit four distinct bank quarters.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "conv_bf16.hip.h"

namespace sspk {

typedef short s16x4 __attribute__((ext_vector_type(4)));

struct WgradBArgs {
  const void* x[2];          // conv input NHWC [N,H,W,x_cs] bf16 per view (raw y of the producing layer when IN_MODE 1)
  const void* dy[2];         // dY NHWC [N,H,W,dy_cs] bf16 (fp32: DY_F32)
  const float* x_scale[2];
  const float* x_shift[2];
  float* partial;            // [ncib * ncob * nsplit][taps][64 ci][64 co]
  int nviews, N, H, W;
  int Cin, x_cs, x_co;
  int Cout, dy_cs, dy_co;
  int tiles_x, tiles_y;
  int ncib, ncob, nsplit;
  unsigned x_img_bytes, dy_img_bytes;
};

template <int KS>
struct WgradBGeom {
  static constexpr int TAPS = KS * KS;
  static constexpr int HT = CB_T + KS - 1;
  static constexpr int X_BYTES = ((HT * HT + 31) / 32) * 32 * 128;   // whole 32-slot staging rounds
  static constexpr int D_BYTES = 256 * 128;
  static constexpr int LDS_BYTES = X_BYTES + D_BYTES;
};

__device__ __forceinline__ s16x8 wb_tr_pair(const unsigned char* p0, const unsigned char* p1) {
  typedef __attribute__((address_space(3))) s16x4* lp;
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(p0));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(p1));
  return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <int KS, int IN_MODE, bool DY_F32>
__global__ __launch_bounds__(256, 2) void wgrad_bf16_kernel(const WgradBArgs a) {
  using G = WgradBGeom<KS>;
  constexpr int HT = G::HT, PAD = KS / 2, TAPS = G::TAPS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
  unsigned char* const sX = smem_b;
  unsigned char* const sD = smem_b + G::X_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int mh = wave >> 1, nh = wave & 1;                // output-channel half, input-channel half of this wave
  const int li = lane & 15, sub = (lane >> 4) & 1, lg = lane >> 5, lt = li >> 2;
  const int pair = blockIdx.x / a.nsplit, split = blockIdx.x - pair * a.nsplit;
  const int cib = pair / a.ncob, cob = pair - cib * a.ncob;
  const int ntiles = a.nviews * a.N * a.tiles_y * a.tiles_x;

  // lane-constant parts of the transposing-read addresses (see the header comment)
  const int baseA = (8 * lg + lt) * 128 + ((mh ^ (lt >> 1)) << 6) + (16 * sub + 4 * (li & 3)) * 2;
  int baseB[3];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) {
    const int beta = ((lt + dx) >> 1) & 1;
    baseB[dx] = (8 * lg + lt) * 128 + ((nh ^ beta) << 6) + (16 * sub + 4 * (li & 3)) * 2;
  }

  // staging: 16-byte item = 8 channels; chunk = tid & 7 is fixed per thread, slot = (tid >> 3) + 32 k
  const int c8 = tid & 7;
  float sc[8], sh[8];
  if (IN_MODE == 1) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = min(cib * 64 + c8 * 8 + e, a.Cin - 1);
      sc[e] = a.x_scale[0][c];  // (re-read per view below when the views differ)
      sh[e] = a.x_shift[0][c];
    }
  }
  int sc_view = 0;

  f32x16 acc[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  constexpr int NXS = (HT * HT * 8 + 255) / 256;  // X items per thread
  constexpr int NDS = 8;                          // dY items per thread
  constexpr unsigned OOB = 0x80000000u;

  for (int t = split; t < ntiles; t += a.nsplit) {
    const int txi = t % a.tiles_x, t2 = t / a.tiles_x;
    const int tyi = t2 % a.tiles_y, t3 = t2 / a.tiles_y;
    const int n = t3 % a.N, view = t3 / a.N;
    const int ty0 = tyi * CB_T, tx0 = txi * CB_T;
    if (IN_MODE == 1 && view != sc_view) {
      sc_view = view;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = min(cib * 64 + c8 * 8 + e, a.Cin - 1);
        sc[e] = a.x_scale[view][c];
        sh[e] = a.x_shift[view][c];
      }
    }
    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(a.x[view])) + (size_t)n * a.x_img_bytes, 0, a.x_img_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_d = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(a.dy[view])) + (size_t)n * a.dy_img_bytes, 0, a.dy_img_bytes, 0x00020000);
    // ---- stage both tiles.  bf16 tensors go straight into LDS by LDS-DMA (no registers: the 144 accumulator registers
    // leave no room for ~80 staging registers): the LDS image is lane-linear (wave w, round k fills slots 32 k + 8 w .. + 7),
    // the half swap of a slot is applied to the SOURCE address of the lane.  X is then activated IN PLACE. ----
    const int cx = cib * 64 + c8 * 8, cd = cob * 64 + c8 * 8;
    typedef __attribute__((address_space(3))) void* ldsp;
    u32x4 dv[DY_F32 ? NDS : 1], dv2[DY_F32 ? NDS : 1];
    if constexpr (DY_F32) {
#pragma unroll
      for (int k = 0; k < NDS; ++k) {
        const int s = (tid >> 3) + 32 * k;
        const int gy = ty0 + (s >> 4), gx = tx0 + (s & 15);
        const bool ok = gy < a.H && gx < a.W && cd < a.Cout;
        const unsigned vo = ok ? (unsigned)(((gy * a.W + gx) * a.dy_cs + a.dy_co + cd) * 4) : OOB;
        dv[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_d, vo, 0, 0));
        dv2[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_d, vo, 16, 0));
      }
    }
    __syncthreads();  // every wave has finished the transposing reads of the previous tile
#pragma unroll
    for (int k = 0; k < NXS; ++k) {
      int s = (tid >> 3) + 32 * k;   // (slots >= HT * HT of the last round land in the padding of the X region)
      asm volatile("" : "+v"(s));    // (opaque: hipcc otherwise hoists the slot coordinates of all rounds out of the tile loop
                                     // and keeps ~30 registers alive beside the 144 accumulators)
      const int cs8 = ((((c8 >> 2) ^ (s >> 1)) & 1) << 2) | (c8 & 3);   // channel item that lives at position c8 of slot s
      const int cxx = cib * 64 + cs8 * 8;
      const int hy = s / HT, hx = s - hy * HT;
      const int gy = ty0 + hy - PAD, gx = tx0 + hx - PAD;
      const bool ok = s < HT * HT && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W && cxx < a.Cin;
      const unsigned vo = ok ? (unsigned)(((gy * a.W + gx) * a.x_cs + a.x_co + cxx) * 2) : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (ldsp)(sX + (32 * k + 8 * wave) * 128), 16, vo, 0, 0, 0);
    }
    if constexpr (!DY_F32) {
#pragma unroll
      for (int k = 0; k < NDS; ++k) {
        const int s = (tid >> 3) + 32 * k;
        const int cs8 = ((((c8 >> 2) ^ (s >> 1)) & 1) << 2) | (c8 & 3);
        const int cdd = cob * 64 + cs8 * 8;
        const int gy = ty0 + (s >> 4), gx = tx0 + (s & 15);
        const bool ok = gy < a.H && gx < a.W && cdd < a.Cout;
        const unsigned vo = ok ? (unsigned)(((gy * a.W + gx) * a.dy_cs + a.dy_co + cdd) * 2) : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_d, (ldsp)(sD + (32 * k + 8 * wave) * 128), 16, vo, 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int k = 0; k < NDS; ++k) {
        const int s = (tid >> 3) + 32 * k;
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          o[e] = pack_bf16(u32_as_f32(dv[k][2 * e]), u32_as_f32(dv[k][2 * e + 1]));
          o[2 + e] = pack_bf16(u32_as_f32(dv2[k][2 * e]), u32_as_f32(dv2[k][2 * e + 1]));
        }
        if (cd + 8 > a.Cout) {  // ragged output channels (pointwise heads): zero beyond Cout
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (cd + 2 * e >= a.Cout) o[e] = 0u;
            else if (cd + 2 * e + 1 >= a.Cout) o[e] &= 0xffffu;
          }
        }
        *reinterpret_cast<u32x4*>(sD + s * 128 + ((((c8 >> 2) ^ (s >> 1)) & 1) << 6) + (c8 & 3) * 16) = o;
      }
    }
    __syncthreads();  // (the DMA has landed: hipcc drains vmcnt before the barrier)
    if (IN_MODE == 1 || (a.Cin & 63) != 0) {
      // activate the raw bf16 halo in place: this thread owns channel item c8 of slots (tid >> 3) + 32 k
#pragma unroll
      for (int k = 0; k < NXS; ++k) {
        int s = (tid >> 3) + 32 * k;
        asm volatile("" : "+v"(s));
        if (s >= HT * HT) continue;
        const int hy = s / HT, hx = s - hy * HT;
        const int gy = ty0 + hy - PAD, gx = tx0 + hx - PAD;
        const bool ok = (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W && cx < a.Cin;
        u32x4* const q = reinterpret_cast<u32x4*>(sX + s * 128 + ((((c8 >> 2) ^ (s >> 1)) & 1) << 6) + (c8 & 3) * 16);
        u32x4 o = {0u, 0u, 0u, 0u};
        if (ok) {
          const u32x4 v = *q;
          float f[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) { f[2 * e] = bf16_lo(v[e]); f[2 * e + 1] = bf16_hi(v[e]); }
          if (IN_MODE == 1) {
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = fmaxf(fmaf(f[e], sc[e], sh[e]), 0.f);
          }
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (cx + e >= a.Cin) f[e] = 0.f;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = pack_bf16(f[2 * e], f[2 * e + 1]);
        }
        *q = o;
      }
      __syncthreads();
    }
    // ---- 16 image rows x taps: K = the 16 pixels of a row.  The fragments of step s + 1 (the next tap's X, the next row's dY) are
    // requested before the MFMA of step s is issued (two register sets, pinned with sched_barrier). ----
    {
      s16x8 fa[2], fb[2];
      auto fetch_a = [&](int row) {
        const unsigned char* pa = sD + baseA + row * 16 * 128;
        return wb_tr_pair(pa, pa + 4 * 128);
      };
      auto fetch_b = [&](int row, int tap) {
        const int dy = tap / KS, dx = tap % KS;
        const int rho = (HT & 2) ? ((row + dy) & 1) : 0;   // bit 1 of the slot index flips with the halo row (pitch 18)
        const unsigned char* pb = sX + (baseB[dx] ^ (rho << 6)) + ((row + dy) * HT + dx) * 128;
        return wb_tr_pair(pb, pb + 4 * 128);
      };
      fa[0] = fetch_a(0);
      fb[0] = fetch_b(0, 0);
#pragma unroll 1
      for (int ry = 0; ry < CB_T; ry += 2) {
#pragma unroll
        for (int st = 0; st < 2 * TAPS; ++st) {   // (row parity, tap); ry is even, so the parities are compile-time
          const int rr = st / TAPS, tap = st - rr * TAPS;
          const int nst = st + 1, nrr = (nst / TAPS) & 1, ntap = nst % TAPS;
          const bool last = st + 1 == 2 * TAPS;     // the next step is row ry + 2 (if any)
          if (!last || ry + 2 < CB_T) {
            const int nrow = last ? ry + 2 : ry + nrr;
            fb[(st + 1) & 1] = fetch_b(nrow, ntap);
            if (ntap == 0) fa[nrr] = fetch_a(nrow);
          }
          __builtin_amdgcn_sched_barrier(0);
          acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[rr], fb[st & 1], acc[tap], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }

  // ---- partial slab [tap][ci][co] ----
  float* const slab = a.partial + (size_t)blockIdx.x * TAPS * 4096;
  const int ci = nh * 32 + (lane & 31);
#pragma unroll
  for (int tap = 0; tap < TAPS; ++tap)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 v = {acc[tap][4 * q], acc[tap][4 * q + 1], acc[tap][4 * q + 2], acc[tap][4 * q + 3]};
      *reinterpret_cast<f32x4*>(slab + tap * 4096 + ci * 64 + mh * 32 + 8 * q + 4 * lg) = v;
    }
}

}  // namespace sspk
