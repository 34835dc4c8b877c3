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
// Two workgroups per CU (78 KB each): one's staging / activation runs under the other's MFMA phase.  The kernel is bound by
// instruction issue (round 5: 2155 -> ~870 instructions per tile and wave for the same 144 MFMAs, PERF_LOG 5c): the 3x3 MFMA loop
// walks the HALO rows - one X fragment per (halo row, column shift) feeds the three taps above each other - , the activation
// runs at constant addresses in the forward kernel's 20 vector instructions per 8 values, and tiles whose halo lies inside the
// map are staged without per-slot coordinates or bounds.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

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
  // FUSE 1 / 2: pass 2 (APPLY) of THIS layer's BatchNorm + ReLU (+ MaxPool2d(2)) backward rides the dY staging (the idea of
  // wgrad_wino_fused_kernel on the bf16 path; bn_bwd_kernel<true, POOL, true, uint16_t> is the separate pass): the kernel stages the
  // layer's raw output y where dY would go, turns it into dY IN PLACE - dY = gs dZ + (P y + Q), dZ = dOut [z > 0] routed to the first
  // maximum of its 2x2 window (FUSE 2) - and the blocks of input-channel block 0 write dY to HBM for the data gradient.
  const void* f_y[2] = {nullptr, nullptr};     // [N,H,W,dy_cs] bf16 (the geometry of dY: dy_cs, dy_co)
  const void* f_dout[2] = {nullptr, nullptr};  // gradient wrt the layer's activation, bf16: [N,H,W,f_dcs] (1) or [N,H/2,W/2,f_dcs] (2)
  void* f_dy[2] = {nullptr, nullptr};          // dY out (the geometry of dY)
  const float* f_scale[2] = {nullptr, nullptr};
  const float* f_shift[2] = {nullptr, nullptr};
  const float* f_mean[2] = {nullptr, nullptr};
  const float* f_invstd[2] = {nullptr, nullptr};
  const float* f_k12[2] = {nullptr, nullptr};  // {S1/n, S2/n} [2 Cout] (bn_bwd_sums_kernel)
  const float* f_gamma = nullptr;
  int f_dcs = 0, f_dco = 0;
};

template <int KS>
struct WgradBGeom {
  static constexpr int TAPS = KS * KS;
  static constexpr int HT = CB_T + KS - 1;
  static constexpr int X_BYTES = ((HT * HT + 31) / 32) * 32 * 128;   // whole 32-slot staging rounds
  static constexpr int D_BYTES = 256 * 128;
  static constexpr int LDS_BYTES = X_BYTES + D_BYTES;
  static constexpr int P_BYTES = 2 * 5 * 64 * 4 + 2 * 2 * 64 * 4;   // FUSE: per view {scale, shift, P, Q, gs} of the block's 64 output
                                                                    // channels + {scale, shift} of its 64 INPUT channels (3.5 KB)
};

__device__ __forceinline__ s16x8 wb_tr_pair(const unsigned char* p0, const unsigned char* p1) {
  typedef __attribute__((address_space(3))) s16x4* lp;
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(p0));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(p1));
  return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <int KS, int IN_MODE, bool DY_F32, int FUSE = 0>
__global__ __launch_bounds__(256, 2) void wgrad_bf16_kernel(const WgradBArgs a) {
  using G = WgradBGeom<KS>;
  constexpr int HT = G::HT, PAD = KS / 2, TAPS = G::TAPS;
  static_assert(FUSE == 0 || (KS == 3 && !DY_F32), "the fused APPLY rides the bf16 dY staging of the 3x3 layers");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
  unsigned char* const sX = smem_b;
  unsigned char* const sD = smem_b + G::X_BYTES;
  float* const sP = reinterpret_cast<float*>(smem_b + G::LDS_BYTES);   // (FUSE only)
  float* const sAff = sP + 2 * 5 * 64;                                  // (FUSE only) the X affine: the registers go to the dY side

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

  int baseB2[2][3];   // by the parity of the halo row (3x3: pitch 18, bit 1 of the slot index flips with the row)
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) { baseB2[0][dx] = baseB[dx]; baseB2[1][dx] = baseB[dx] ^ 64; }

  // staging: 16-byte item = 8 channels; chunk = tid & 7 is fixed per thread, slot = (tid >> 3) + 32 k.  Bit 1 of the slot index is
  // (tid >> 4) & 1 in every round, so the half swap of a slot is a per-thread constant: position c8 of this thread's slots holds
  // channel item cs8, and this thread's own item c8 lives at byte qswz of the slot.
  const int c8 = tid & 7;
  const int cs8 = ((((c8 >> 2) ^ (tid >> 4)) & 1) << 2) | (c8 & 3);
  const int qswz = ((((c8 >> 2) ^ (tid >> 4)) & 1) << 6) + (c8 & 3) * 16;
  const int cxx2 = (cib * 64 + cs8 * 8) * 2, cdd = cob * 64 + cs8 * 8;
  const unsigned lane_d = (unsigned)((((tid >> 7) * a.W + ((tid >> 3) & 15)) * a.dy_cs + cdd) * 2);
  // BatchNorm affine of this thread's 8 input channels as packed pairs; ZERO beyond Cin: max(0 x + 0, 0) = 0 is the padding value,
  // so the activation needs no per-channel mask
  f32x2 sc2[4], sh2[4];
  auto load_affine = [&](int view) __attribute__((always_inline)) {
    if constexpr (FUSE != 0) return;   // (fused forms: from the LDS table sAff, per tile)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = cib * 64 + c8 * 8 + e;
      const bool in = c < a.Cin;
      sc2[e >> 1][e & 1] = in ? a.x_scale[view][min(c, a.Cin - 1)] : 0.f;
      sh2[e >> 1][e & 1] = in ? a.x_shift[view][min(c, a.Cin - 1)] : 0.f;
    }
  };
  if (IN_MODE == 1) load_affine(0);  // (re-read per view below when the views differ)
  int sc_view = 0;
  if constexpr (FUSE != 0 && IN_MODE == 1) {
    if (tid >= 128) {
      const int view = (tid - 128) >> 6, ch = tid & 63, c = cib * 64 + ch;
      const bool in = view < a.nviews && c < a.Cin;
      sAff[(view * 2 + 0) * 64 + ch] = in ? a.x_scale[view][c] : 0.f;
      sAff[(view * 2 + 1) * 64 + ch] = in ? a.x_shift[view][c] : 0.f;
    }
  }

  if constexpr (FUSE != 0) {
    // dY = gs (dZ - S1/n - xhat S2/n), xhat = (y - mean) invstd   ==   gs dZ + (P y + Q):  P = -gs invstd S2/n,  Q = gs (mean invstd S2/n - S1/n)
    if (tid < 128) {
      const int view = tid >> 6, ch = tid & 63, co = cob * 64 + ch;
      float v[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
      if (view < a.nviews && co < a.Cout) {
        const float is = a.f_invstd[view][co], gs = a.f_gamma[co] * is, mu = a.f_mean[view][co];
        const float k1 = a.f_k12[view][co], k2 = a.f_k12[view][a.Cout + co];
        v[0] = a.f_scale[view][co]; v[1] = a.f_shift[view][co]; v[2] = -gs * is * k2; v[3] = gs * (mu * is * k2 - k1); v[4] = gs;
      }
#pragma unroll
      for (int j = 0; j < 5; ++j) sP[(view * 5 + j) * 64 + ch] = v[j];
    }
  }
  // FUSE: this thread's window items (2x2 pixels x channel item c8): windows (tid >> 3) + 32 j of the tile's 8 x 8, j = 0, 1; bit 0 of
  // the window column = bit 1 of its pixels' slot indices is (tid >> 3) & 1 for both, so the item sits at ONE byte offset of its slots
  // (recomputed from an opaque copy of the thread index where they are used: as kernel-lifetime values they cost the one register
  // the fused forms do not have)
#define WGB_WINDOW_CONSTS()                                                                               \
  int t_ = tid;                                                                                           \
  asm volatile("" : "+v"(t_));                                                                            \
  const int f_wx = (t_ >> 3) & 7, f_wy0 = t_ >> 6;   /* window column, row (of j = 0; j = 1: + 4) */      \
  const int f_c8 = t_ & 7;                                                                                \
  const int qswz_d = ((((f_c8 >> 2) ^ (t_ >> 3)) & 1) << 6) + (f_c8 & 3) * 16;                            \
  const bool f_chok = cob * 64 + f_c8 * 8 < a.Cout;

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
      load_affine(view);
    }
    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(a.x[view])) + (size_t)n * a.x_img_bytes, 0, a.x_img_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_d = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(FUSE != 0 ? a.f_y[view] : a.dy[view])) + (size_t)n * a.dy_img_bytes, 0,
        a.dy_img_bytes, 0x00020000);
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
    // tiles whose halo lies inside the map, for a layer of whole 64-channel blocks: no per-slot coordinates or bounds on either side
    const bool inside = ty0 >= PAD && tx0 >= PAD && ty0 + CB_T + PAD <= a.H && tx0 + CB_T + PAD <= a.W;
    const bool fastx = inside && (a.Cin & 63) == 0, fastd = inside && (a.Cout & 63) == 0;
    unsigned okmask = 0;  // bit k: X slot (tid >> 3) + 32 k is a pixel of the map (general path; the activation zeroes the others)
    // FUSE: the gradient wrt the activation at this thread's window items (one pooled pixel per window, or its four pixels)
    // (FUSE 2: both windows' pooled quads now; FUSE 1: the four pixels of window 0 now, those of window 1 - into the same registers -
    // when window 0 is done, under the activation of X)
    constexpr int NDQ = FUSE == 2 ? 2 : FUSE == 1 ? 4 : 1;
    u32x4 dq[NDQ];
    unsigned f_ok = 0;   // bit 4 j + k: pixel k of window j of this thread lies inside the map and its channel item exists
    const int f_Wo = FUSE == 2 ? a.W >> 1 : a.W;
    const unsigned f_obytes = (unsigned)((FUSE == 2 ? a.H >> 1 : a.H) * f_Wo * a.f_dcs) * 2u;
    const __amdgpu_buffer_rsrc_t rsrc_o = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(FUSE != 0 ? a.f_dout[view] : a.dy[view])) + (size_t)n * f_obytes, 0,
        f_obytes, 0x00020000);
    auto load_dq = [&](auto J_) __attribute__((always_inline)) {
      constexpr int j = decltype(J_)::value;
      WGB_WINDOW_CONSTS()
      (void)qswz_d;
      const int wy = f_wy0 + 4 * j;
      if constexpr (FUSE == 2) {   // (H and W even: a window is inside the map or outside as a whole)
        const bool ok = f_chok && ty0 + 2 * wy < a.H && tx0 + 2 * f_wx < a.W;
        f_ok |= (ok ? 15u : 0u) << (4 * j);
        const unsigned vo = ok ? (unsigned)(((((ty0 >> 1) + wy) * f_Wo + (tx0 >> 1) + f_wx) * a.f_dcs + a.f_dco + cob * 64 + f_c8 * 8) * 2) : OOB;
        dq[j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_o, vo, 0, 0));
      } else {                     // (no pooling behind the layer: the map may be odd, the pixels of a window are tested one by one)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int gy = ty0 + 2 * wy + (k >> 1), gx = tx0 + 2 * f_wx + (k & 1);
          const bool ok = f_chok && gy < a.H && gx < a.W;
          f_ok |= (ok ? 1u : 0u) << (4 * j + k);
          const unsigned vo = ok ? (unsigned)(((gy * f_Wo + gx) * a.f_dcs + a.f_dco + cob * 64 + f_c8 * 8) * 2) : OOB;
          dq[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_o, vo, 0, 0));
        }
      }
    };
    if constexpr (FUSE != 0) {
      load_dq(std::integral_constant<int, 0>{});
      if constexpr (FUSE == 2) load_dq(std::integral_constant<int, 1>{});
    }
    __syncthreads();  // every wave has finished the transposing reads of the previous tile
    if (fastx) {
      const int sbx = (((ty0 - PAD) * a.W + tx0 - PAD) * a.x_cs + a.x_co) * 2;   // (wave-uniform: the scalar offset of the load)
#pragma unroll
      for (int k = 0; k < NXS; ++k) {
        unsigned s = (tid >> 3) + 32 * k;   // (slots >= HT * HT of the last round land in the padding of the X region: whatever
        asm volatile("" : "+v"(s));         //  they load - rows below the tile, or 0 beyond the image - is never read)
        const unsigned hy = HT == 16 ? s >> 4 : __umul24(s, 3641u) >> 16;   // s / 18 for s < 352
        const unsigned hx = s - __umul24(hy, (unsigned)HT);
        const unsigned vo = __umul24(__umul24(hy, (unsigned)a.W) + hx, (unsigned)(a.x_cs * 2)) + cxx2;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (ldsp)(sX + (32 * k + 8 * wave) * 128), 16, vo, sbx, 0, 0);
      }
    } else {
#pragma unroll
      for (int k = 0; k < NXS; ++k) {
        int s = (tid >> 3) + 32 * k;
        asm volatile("" : "+v"(s));    // (opaque: hipcc otherwise hoists the slot coordinates of all rounds out of the tile loop
                                       // and keeps ~30 registers alive beside the 144 accumulators)
        const int hy = HT == 16 ? s >> 4 : (int)(__umul24((unsigned)s, 3641u) >> 16), hx = s - hy * HT;
        const int gy = ty0 + hy - PAD, gx = tx0 + hx - PAD;
        const bool in_map = (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
        const bool ok = s < HT * HT && in_map && cxx2 < a.Cin * 2;
        okmask |= (in_map ? 1u : 0u) << k;
        const unsigned vo = ok ? (unsigned)(((gy * a.W + gx) * a.x_cs + a.x_co) * 2 + cxx2) : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (ldsp)(sX + (32 * k + 8 * wave) * 128), 16, vo, 0, 0, 0);
      }
    }
    if constexpr (!DY_F32) {
      if (fastd) {
        // slot (tid >> 3) + 32 k = pixel row (tid >> 7) + 2 k, column (tid >> 3) & 15: the round advances two image rows
        const int sbd = ((ty0 * a.W + tx0) * a.dy_cs + a.dy_co) * 2, rows2 = a.W * a.dy_cs * 4;
#pragma unroll
        for (int k = 0; k < NDS; ++k)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_d, (ldsp)(sD + (32 * k + 8 * wave) * 128), 16, lane_d, sbd + k * rows2, 0, 0);
      } else {
#pragma unroll
        for (int k = 0; k < NDS; ++k) {
          const int s = (tid >> 3) + 32 * k;
          const int gy = ty0 + (s >> 4), gx = tx0 + (s & 15);
          const bool ok = gy < a.H && gx < a.W && cdd < a.Cout;
          const unsigned vo = ok ? (unsigned)(((gy * a.W + gx) * a.dy_cs + a.dy_co + cdd) * 2) : OOB;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_d, (ldsp)(sD + (32 * k + 8 * wave) * 128), 16, vo, 0, 0, 0);
        }
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
        *reinterpret_cast<u32x4*>(sD + s * 128 + qswz) = o;
      }
    }
    __syncthreads();  // (the DMA has landed: hipcc drains vmcnt before the barrier)
    // ---- FUSE: y -> dY in place, one window item at a time: dZ = dOut [z > 0] at the FIRST maximum of z over the window (FUSE 2: torch's
    // max_pool2d routing - strictly greater replaces - ; all z <= 0: nothing), dY = gs dZ + (P y + Q); parameters per channel pair from
    // the table (registers: 144 accumulators leave no room for 40 resident parameters).  Window 0 before, window 1 behind the activation
    // of X; the barrier that closes the staging orders both before the MFMA phase. ----
    const __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<unsigned char*>(FUSE != 0 ? a.f_dy[view] : nullptr) + (size_t)n * a.dy_img_bytes, 0, a.dy_img_bytes, 0x00020000);
    auto apply_window = [&](auto J_) __attribute__((always_inline)) {
      constexpr int j = decltype(J_)::value;
      typedef float pf2 __attribute__((ext_vector_type(2)));
      WGB_WINDOW_CONSTS()
      (void)f_chok;
      const float* const pp = sP + view * 5 * 64 + f_c8 * 8;
      const int wy = f_wy0 + 4 * j;
      unsigned char* const q0 = sD + ((2 * wy) * 16 + 2 * f_wx) * 128 + qswz_d;   // pixel (0, 0) of the window; (r, c): + (16 r + c) 128
      u32x4 v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const u32x4*>(q0 + (16 * (k >> 1) + (k & 1)) * 128);
      const unsigned okb = (f_ok >> (4 * j)) & 15u;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const pf2 sc_ = *reinterpret_cast<const pf2*>(pp + 2 * e), sh_ = *reinterpret_cast<const pf2*>(pp + 64 + 2 * e);
        const pf2 P_ = *reinterpret_cast<const pf2*>(pp + 128 + 2 * e), Q_ = *reinterpret_cast<const pf2*>(pp + 192 + 2 * e);
        const pf2 gs_ = *reinterpret_cast<const pf2*>(pp + 256 + 2 * e);
        const f32x2 sc2_ = {sc_[0], sc_[1]}, sh2_ = {sh_[0], sh_[1]}, P2_ = {P_[0], P_[1]}, Q2_ = {Q_[0], Q_[1]}, gs2_ = {gs_[0], gs_[1]};
        if constexpr (FUSE == 2) {
          f32x2 z[4], lin[4], dz[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const f32x2 y2 = {bf16_lo(v[k][e]), bf16_hi(v[k][e])};
            z[k] = pk_fma(y2, sc2_, sh2_);
            lin[k] = pk_fma(y2, P2_, Q2_);
          }
          const f32x2 d2 = {bf16_lo(dq[j][e]), bf16_hi(dq[j][e])};
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const float m = fmaxf(fmaxf(z[0][u], z[1][u]), fmaxf(z[2][u], z[3][u]));
            const float pm = m > 0.f ? d2[u] : 0.f;
            const bool e0 = z[0][u] == m, e1 = !e0 && z[1][u] == m, e2 = !(e0 || e1) && z[2][u] == m;
            const bool e3 = !(e0 || e1 || e2) && z[3][u] == m;
            dz[0][u] = e0 ? pm : 0.f; dz[1][u] = e1 ? pm : 0.f; dz[2][u] = e2 ? pm : 0.f; dz[3][u] = e3 ? pm : 0.f;
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const f32x2 o2 = pk_fma(gs2_, dz[k], lin[k]);
            v[k][e] = ((okb >> k) & 1u) ? pack_bf16(o2[0], o2[1]) : 0u;   // 0 outside the map / the tensor, like a zero-filled dY load
          }
        } else {
          // (pixel by pixel: the four pixels are independent here, and values kept across them are registers this form does not have)
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const f32x2 y2 = {bf16_lo(v[k][e]), bf16_hi(v[k][e])};
            const f32x2 z = pk_fma(y2, sc2_, sh2_);
            const f32x2 d2 = {bf16_lo(dq[k][e]), bf16_hi(dq[k][e])};
            const f32x2 dz = {z[0] > 0.f ? d2[0] : 0.f, z[1] > 0.f ? d2[1] : 0.f};
            const f32x2 o2 = pk_fma(gs2_, dz, pk_fma(y2, P2_, Q2_));
            v[k][e] = ((okb >> k) & 1u) ? pack_bf16(o2[0], o2[1]) : 0u;
          }
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) *reinterpret_cast<u32x4*>(q0 + (16 * (k >> 1) + (k & 1)) * 128) = v[k];
      if (cib == 0) {   // dY for the data gradient of this layer: every tile exactly once
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const unsigned vo = ((okb >> k) & 1u) ? (unsigned)((((ty0 + 2 * wy + (k >> 1)) * a.W + tx0 + 2 * f_wx + (k & 1)) * a.dy_cs + a.dy_co + cob * 64 + f_c8 * 8) * 2) : OOB;
          ssp_store_b128(v[k], rsrc_y, vo, 0);
        }
      }
    };
    if constexpr (FUSE != 0) {
      apply_window(std::integral_constant<int, 0>{});
      if constexpr (FUSE == 1) load_dq(std::integral_constant<int, 1>{});
    }
    if constexpr (IN_MODE == 1) {
      // BatchNorm + ReLU + rounding of the raw bf16 halo IN PLACE: this thread owns channel item c8 of slots (tid >> 3) + 32 k, i.e.
      // one 16-byte address + 4096 k.  The forward kernel's arithmetic (conv_bf16_ws_kernel, stage_halo): packed fma, round, ReLU
      // on the packed result as a 16-bit integer max with 0 - 20 vector instructions per 8 values.
      typedef short s16x2 __attribute__((ext_vector_type(2)));
      if constexpr (FUSE != 0) {
        const float* const pa_ = sAff + view * 128 + c8 * 8;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          sc2[e] = f32x2{pa_[2 * e], pa_[2 * e + 1]};
          sh2[e] = f32x2{pa_[64 + 2 * e], pa_[64 + 2 * e + 1]};
        }
      }
#pragma unroll
      for (int k = 0; k < NXS; ++k) {
        if (32 * k + 32 > HT * HT && (tid >> 3) + 32 * k >= HT * HT) continue;   // (the last round is partly padding)
        u32x4* const q = reinterpret_cast<u32x4*>(sX + (tid >> 3) * 128 + qswz + k * 4096);
        const u32x4 v = *q;
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const f32x2 z = pk_fma(f32x2{bf16_lo(v[e]), bf16_hi(v[e])}, sc2[e], sh2[e]);
          const s16x2 r = __builtin_elementwise_max(__builtin_bit_cast(s16x2, pack_bf16(z[0], z[1])), s16x2{0, 0});
          o[e] = __builtin_bit_cast(uint32_t, r);
        }
        if (!fastx) {   // padding is zero in the ACTIVATED domain
          const bool pad = !((okmask >> k) & 1u);
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = pad ? 0u : o[e];
        }
        *q = o;
      }
      if constexpr (FUSE != 0) apply_window(std::integral_constant<int, 1>{});
      __syncthreads();
    } else if ((a.Cin & 63) != 0) {
      // (no activation, a ragged last channel block: the channels beyond Cin of a loaded item are zeroed)
#pragma unroll
      for (int k = 0; k < NXS; ++k) {
        if (32 * k + 32 > HT * HT && (tid >> 3) + 32 * k >= HT * HT) continue;
        u32x4* const q = reinterpret_cast<u32x4*>(sX + (tid >> 3) * 128 + qswz + k * 4096);
        u32x4 o = *q;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (cx + 2 * e >= a.Cin) o[e] = 0u;
          else if (cx + 2 * e + 1 >= a.Cin) o[e] &= 0xffffu;
        }
        *q = o;
      }
      if constexpr (FUSE != 0) apply_window(std::integral_constant<int, 1>{});
      __syncthreads();
    } else if constexpr (FUSE != 0) {
      apply_window(std::integral_constant<int, 1>{});
      __syncthreads();   // (the in-place dY, before the MFMA phase reads it)
    }
    if constexpr (KS == 3) {
      // ---- halo-row major: the X fragment of (halo row r, column shift dx) feeds the three taps (dy, dx) with the dY rows r - dy
      // (K = the 16 pixels of an image row): 54 X + 16 dY fragment fetches per tile for 144 MFMAs (tap major: 144 + 16).  X
      // fragments are requested two steps (2 - 6 MFMAs) ahead into three rotating registers, dY rows one halo row ahead into four;
      // every address is a lane constant + an immediate. ----
      s16x8 fa[4], fb[3];
      auto fetch_a = [&](int row) __attribute__((always_inline)) {
        const unsigned char* pa = sD + baseA + row * 16 * 128;
        return wb_tr_pair(pa, pa + 4 * 128);
      };
      auto fetch_b = [&](int st) __attribute__((always_inline)) {   // step st = 3 r + dx
        const int r = st / 3, dx = st - 3 * r;
        const unsigned char* pb = sX + baseB2[r & 1][dx] + (r * HT + dx) * 128;   // (bit 1 of the slot index flips with the halo row)
        return wb_tr_pair(pb, pb + 4 * 128);
      };
      fb[0] = fetch_b(0);
      fa[0] = fetch_a(0);
      fb[1] = fetch_b(1);
#pragma unroll
      for (int st = 0; st < 3 * HT; ++st) {
        const int r = st / 3, dx = st - 3 * r;
        if (st + 2 < 3 * HT) fb[(st + 2) % 3] = fetch_b(st + 2);
        if (dx == 0 && r + 1 < CB_T) fa[(r + 1) & 3] = fetch_a(r + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          const int row = r - dy;
          if (row >= 0 && row < CB_T)
            acc[dy * 3 + dx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[row & 3], fb[st % 3], acc[dy * 3 + dx], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      // ---- pointwise: 16 image rows, K = the 16 pixels of a row; the fragments of row s + 1 are requested before the MFMA of row s ----
      s16x8 fa[2], fb[2];
      auto fetch_a = [&](int row) __attribute__((always_inline)) {
        const unsigned char* pa = sD + baseA + row * 16 * 128;
        return wb_tr_pair(pa, pa + 4 * 128);
      };
      auto fetch_b = [&](int row) __attribute__((always_inline)) {
        const unsigned char* pb = sX + baseB[0] + row * HT * 128;
        return wb_tr_pair(pb, pb + 4 * 128);
      };
      fa[0] = fetch_a(0);
      fb[0] = fetch_b(0);
#pragma unroll
      for (int ry = 0; ry < CB_T; ++ry) {
        if (ry + 1 < CB_T) {
          fa[(ry + 1) & 1] = fetch_a(ry + 1);
          fb[(ry + 1) & 1] = fetch_b(ry + 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ry & 1], fb[ry & 1], acc[0], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
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

#undef WGB_WINDOW_CONSTS

}  // namespace sspk
