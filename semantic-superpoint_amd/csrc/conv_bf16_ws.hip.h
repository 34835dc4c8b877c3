// Wave-specialised form of conv_bf16_kernel for the 3x3 layers with bf16 tensors at both ends (forward: IN_MODE 1, data
// gradient: IN_MODE 0) - the launches that carry the bf16 path (csrc/conv_bf16.hip.h has the semantics and the generic kernel,
// which keeps the pointwise / fp32-ended variants).
//
// Why.  The generic kernel runs load -> stage -> MFMA -> epilogue in every wave; measured on 64 -> 64 @240x320 the four phases
// simply add up (0.57 ms for 0.165 ms of MFMAs): ~3000 bookkeeping instructions per tile sit in the same instruction stream as
// the 144 MFMAs, and a second workgroup per CU adds 15 %.  Here ONE 8-wave workgroup per CU splits the roles:
//   waves 0-3 (consumers): nothing but ds_read_b128 + v_mfma_f32_32x32x16_bf16 (the loop that runs at 2.0 PFLOP/s in
//                          tools/ubench/mfma_lds_loop.hip), plus the accumulator -> LDS tile hand-off of a finished unit;
//   waves 4-7 (producers): global loads a stage ahead, BatchNorm + ReLU + rounding, LDS writes of the next stage's halo, the
//                          weight image when it is not already resident, and the copy-out / statistics / pooled copy of the
//                          previous unit's LDS tile.
// One s_barrier per stage (32 input channels x 9 taps of one 16x16 tile).  LDS: 2 weight slots x 36 KB (slot = chunk parity: a
// 64-channel layer keeps BOTH its chunks resident for the whole launch, so its weights are staged once per workgroup instead of
// once per tile), 2 halo buffers x 25.3 KB, one 32 KB output tile = 154.6 KB.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "conv_bf16.hip.h"

namespace sspk {

struct ConvWsGeom {
  static constexpr int HT = CB_T + 2;
  static constexpr int H_BYTES = HT * HT * CB_PS;         // 25,920
  static constexpr int W_BYTES = 9 * 2 * 2 * 1024;        // 36,864
  static constexpr int O_BYTES = 256 * 128;               // 32,768 (16-byte items XOR-swizzled by the pixel index)
  static constexpr int LDS_BYTES = 2 * W_BYTES + 2 * H_BYTES + O_BYTES;
};

template <int IN_MODE>
__global__ __launch_bounds__(512, 1) void conv_bf16_ws_kernel(const ConvBArgs a) {
  using G = ConvWsGeom;
  constexpr int HT = G::HT, PAD = 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
  unsigned char* const sW = smem_b;
  unsigned char* const sH = smem_b + 2 * G::W_BYTES;
  unsigned char* const sO = sH + 2 * G::H_BYTES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (scalar: the role branch is a real branch, not an exec mask)
  const bool producer = wave >= 4;

  // ---- work assignment: contiguous unit range per XCD, blocks of an XCD interleaved (as the generic kernel) ----
  const int T = a.N * a.tiles_y * a.tiles_x;
  const int U = a.nviews * a.ncob * T;
  const int nslot = gridDim.x >> 3, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int per = (U + 7) >> 3;
  const int u_end = min(U, (xcd + 1) * per);
  const int u0 = xcd * per + slot;
  if (u0 >= u_end) return;
  const int nunits = (u_end - u0 + nslot - 1) / nslot;
  const int nstages = nunits * a.nchunks;
  constexpr unsigned OOB = 0x80000000u;

  if (!producer) {
    // =========================== consumers ===========================
    const int lj = lane & 31, lg = lane >> 5;
    int pr, pc;
    cb_lane_pixel<HT>(lj, pr, pc);
    int boff[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) boff[nt] = ((4 * wave + 2 * nt + pr) * HT + pc) * CB_PS + lg * 16;
    const int aoff = lane * 16;
    f32x16 acc[2][2];
    f32x4 bias4[2][4];
    int chunk = 0, u = u0;
    __syncthreads();   // the producers' prologue has staged stage 0
    for (int s = 0; s < nstages; ++s) {
      if (chunk == 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        const int vc = u / T, cob = vc % a.ncob;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int co = cob * CB_NB + mt * 32 + 8 * q + 4 * lg;
            f32x4 b = {0.f, 0.f, 0.f, 0.f};
            if (a.bias != nullptr) {
              if (co + 4 <= a.Cout && (reinterpret_cast<uintptr_t>(a.bias + co) & 15) == 0) b = *reinterpret_cast<const f32x4*>(a.bias + co);
              else {
#pragma unroll
                for (int e = 0; e < 4; ++e) b[e] = co + e < a.Cout ? a.bias[co + e] : 0.f;
              }
            }
            bias4[mt][q] = b;
          }
      }
      const unsigned char* const pW = sW + (chunk & 1) * G::W_BYTES + aoff;
      const unsigned char* const pH = sH + (s & 1) * G::H_BYTES;
      s16x8 fa[2][2], fb[2][2];
      auto fetch = [&](int step, s16x8 (&qa)[2], s16x8 (&qb)[2]) {
        const int tap = step >> 1, ks = step & 1;
        const int dy = tap / 3, dx = tap % 3;
        qa[0] = *reinterpret_cast<const s16x8*>(pW + ((tap * 2 + ks) * 2 + 0) * 1024);
        qa[1] = *reinterpret_cast<const s16x8*>(pW + ((tap * 2 + ks) * 2 + 1) * 1024);
        qb[0] = *reinterpret_cast<const s16x8*>(pH + boff[0] + (dy * HT + dx) * CB_PS + ks * 32);
        qb[1] = *reinterpret_cast<const s16x8*>(pH + boff[1] + (dy * HT + dx) * CB_PS + ks * 32);
      };
      if (!(a.ablate & 8)) {
      fetch(0, fa[0], fb[0]);
#pragma unroll
      for (int step = 0; step < 18; ++step) {
        const int cur = step & 1;
        if (step + 1 < 18) fetch(step + 1, fa[cur ^ 1], fb[cur ^ 1]);
        __builtin_amdgcn_sched_barrier(0);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][0], fb[cur][0], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][0], fb[cur][1], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][1], fb[cur][0], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][1], fb[cur][1], acc[1][1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      }
      if (chunk + 1 == a.nchunks) {
        // finished unit: bias, rounding, hand the tile to the producers through LDS (they copied the previous one out a stage ago)
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
              *reinterpret_cast<u32x2*>(sO + lp * 128 + (((mt * 4 + q) ^ (lp & 7)) << 4) + lg * 8) = o;
            }
        }
        chunk = 0; u += nslot;
      } else {
        ++chunk;
      }
      __syncthreads();
    }
    return;
  }

  // =========================== producers ===========================
  const int ptid = tid - 256, pwave = wave - 4;
  const int part = ptid & 3;
  constexpr int NHS = (HT * HT * 4 + 255) / 256;   // 6 (the last round is partly empty)
  const int hs_lds0 = (ptid >> 2) * CB_PS + part * 16;
  unsigned hs_g[NHS];
  int hs_rel[NHS], hs_yx[NHS];
#pragma unroll
  for (int i = 0; i < NHS; ++i) {
    const int sl = (ptid >> 2) + 64 * i;
    const int hy = sl / HT, hx = sl - hy * HT;
    hs_yx[i] = ((sl < HT * HT ? hy - PAD : -30000) << 16) | ((hx - PAD) & 0xffff);
    hs_rel[i] = (((hy - PAD) * a.W + (hx - PAD)) * a.in_cs + a.in_co + part * 8) * 2;
  }
  __amdgpu_buffer_rsrc_t rsrc_in;
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t*>(a.wpk), 0, (unsigned)(a.ncob * a.nchunks * G::W_BYTES), 0x00020000);
  // unit descriptors: "ld_" = the unit whose stages are being loaded / staged, "st_" = the finished unit awaiting copy-out
  int ld_view = 0, ld_cob = 0, ld_n = 0, ld_ty0 = 0, ld_tx0 = 0, ld_vc = 0;
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
    const int org = (ld_ty0 * a.W + ld_tx0) * a.in_cs * 2;
    const bool interior = ld_ty0 >= PAD && ld_tx0 >= PAD && ld_ty0 + CB_T + PAD <= a.H && ld_tx0 + CB_T + PAD <= a.W;
    if (interior) {
#pragma unroll
      for (int i = 0; i < NHS; ++i) hs_g[i] = (hs_yx[i] >> 16) > -30000 ? (unsigned)(org + hs_rel[i]) : OOB;
    } else {
#pragma unroll
      for (int i = 0; i < NHS; ++i) {
        const int gy = ld_ty0 + (hs_yx[i] >> 16), gx = ld_tx0 + (short)(hs_yx[i] & 0xffff);
        const bool ok = (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
        hs_g[i] = ok ? (unsigned)(org + hs_rel[i]) : OOB;
      }
    }
  };
  u32x4 hv[NHS];
  unsigned hv_pad = 0;     // bit i: slot i of the loaded stage is padding (outside the image)
  float sc[8], sh[8];
  auto issue = [&](int chunk) {   // halo + affine of (ld_ unit, chunk) -> registers
    const int c0 = chunk * CB_KC + part * 8;
    hv_pad = 0;
#pragma unroll
    for (int i = 0; i < NHS; ++i) {
      hv[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_in, (a.ablate & 1) ? OOB : hs_g[i], chunk * CB_KC * 2, 0));
      hv_pad |= (hs_g[i] == OOB ? 1u : 0u) << i;
    }
    if (IN_MODE == 1) {
      const float* const p_scale = a.in_scale[ld_view] + c0;
      const float* const p_shift = a.in_shift[ld_view] + c0;
      const f32x4 s0 = *reinterpret_cast<const f32x4*>(p_scale), s1 = *reinterpret_cast<const f32x4*>(p_scale + 4);
      const f32x4 h0 = *reinterpret_cast<const f32x4*>(p_shift), h1 = *reinterpret_cast<const f32x4*>(p_shift + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { sc[e] = s0[e]; sc[4 + e] = s1[e]; sh[e] = h0[e]; sh[4 + e] = h1[e]; }
    }
  };
  auto stage_halo = [&](int buf) {   // registers -> activated bf16 operands in halo buffer `buf`
    if (a.ablate & 2) return;
    unsigned char* const dst = sH + buf * G::H_BYTES + hs_lds0;
    // BatchNorm + ReLU + rounding of 8 values in 20 vector instructions: 8 shifts / masks (bf16 -> fp32), 4 packed fmas, 4
    // v_cvt_pk_bf16_f32 and the ReLU on the PACKED result as a 16-bit integer max with 0 (a negative bf16 is a negative int16;
    // rounding is monotonic and odd, so relu(round(z)) == round(relu(z)))
    typedef short s16x2 __attribute__((ext_vector_type(2)));
    f32x2 sc2[4], sh2[4];
    if (IN_MODE == 1) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { sc2[e] = f32x2{sc[2 * e], sc[2 * e + 1]}; sh2[e] = f32x2{sh[2 * e], sh[2 * e + 1]}; }
    }
#pragma unroll
    for (int i = 0; i < NHS; ++i) {
      if ((ptid >> 2) + 64 * i >= HT * HT) continue;
      u32x4 o;
      if (IN_MODE == 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const f32x2 z = pk_fma(f32x2{bf16_lo(hv[i][e]), bf16_hi(hv[i][e])}, sc2[e], sh2[e]);
          const s16x2 r = __builtin_elementwise_max(__builtin_bit_cast(s16x2, pack_bf16(z[0], z[1])), s16x2{0, 0});
          o[e] = __builtin_bit_cast(uint32_t, r);
        }
        if (hv_pad != 0) {   // (border tiles only) padding is zero in the ACTIVATED domain
          const bool pad = (hv_pad >> i) & 1u;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = pad ? 0u : o[e];
        }
      } else {
        o = hv[i];
      }
      *reinterpret_cast<u32x4*>(dst + i * 64 * CB_PS) = o;
    }
  };
  int w_tag[2] = {-1, -1};
  auto stage_weights = [&](int cob, int chunk) {   // the 36 KB image of (cob, chunk) into slot chunk & 1 unless it is there already
    const int tag = cob * a.nchunks + chunk, sl = chunk & 1;
    if (w_tag[sl] == tag) return;
    w_tag[sl] = tag;
    u32x4 wv[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) wv[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, ptid * 16, tag * G::W_BYTES + 4096 * i, 0));
#pragma unroll
    for (int i = 0; i < 9; ++i) *reinterpret_cast<u32x4*>(sW + sl * G::W_BYTES + (ptid + 256 * i) * 16) = wv[i];
  };

  // statistics of the stored values: per-thread sums of this thread's 8 channels (item = ptid & 7), flushed per (view, cob)
  const int item = ptid & 7;
  f32x2 ps[4], pq[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) ps[e] = pq[e] = f32x2{0.f, 0.f};
  int st_key = -1;
  auto flush_stats = [&](int key) {
    const int view = key / a.ncob, cob = key - view * a.ncob;
    double* const p_stats = a.stats[view];
    float v[16];
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[2 * e] = ps[e][0]; v[2 * e + 1] = ps[e][1]; v[8 + 2 * e] = pq[e][0]; v[9 + 2 * e] = pq[e][1]; ps[e] = pq[e] = f32x2{0.f, 0.f}; }
#pragma unroll
    for (int i = 0; i < 16; ++i) {   // lanes with the same item: lane bits 3..5
      v[i] += __shfl_xor(v[i], 8);
      v[i] += __shfl_xor(v[i], 16);
      v[i] += __shfl_xor(v[i], 32);
    }
    if (lane < 8 && p_stats != nullptr) {
      double* const st = p_stats + (size_t)((blockIdx.x * 4 + pwave) % NREP) * 2 * a.Cout;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int co = cob * CB_NB + item * 8 + e;
        if (co < a.Cout) {
          unsafeAtomicAdd(st + co, (double)v[e]);
          unsafeAtomicAdd(st + a.Cout + co, (double)v[8 + e]);
        }
      }
    }
  };

  // copy-out of a finished unit from the LDS tile: 16-byte stores, statistics of the stored values, raw pooled copy
  auto copy_out = [&](int uu) {
    if (a.ablate & 4) return;
    const int vc = uu / T, t = uu - vc * T;
    const int view = vc / a.ncob, cob = vc - view * a.ncob;
    const int txi = t % a.tiles_x, t2 = t / a.tiles_x;
    const int n = t2 / a.tiles_y, ty0 = (t2 % a.tiles_y) * CB_T, tx0 = txi * CB_T;
    const bool do_stats = a.stats[0] != nullptr;
    if (do_stats && vc != st_key) {
      if (st_key >= 0) flush_stats(st_key);
      st_key = vc;
    }
    const int co0 = cob * CB_NB + item * 8;
    const bool ch_ok = co0 < a.Cout;   // (Cout % 8 == 0: host-checked)
    const int col = (ptid >> 3) & 15, row0 = ptid >> 7;
    unsigned char* const p_out = reinterpret_cast<unsigned char*>(a.out[view]);
    const unsigned out_img_bytes = (unsigned)a.H * a.W * a.out_cs * 2u;
    const __amdgpu_buffer_rsrc_t rsrc_out = __builtin_amdgcn_make_buffer_rsrc(p_out + (size_t)n * out_img_bytes, 0, out_img_bytes, 0x00020000);
    const bool col_ok = ch_ok && tx0 + col < a.W;
    const unsigned vo = col_ok ? (unsigned)((((ty0 + row0) * a.W + tx0 + col) * a.out_cs + a.out_co + co0) * 2) : OOB;
    const int rstep = 2 * a.W * a.out_cs * 2;
    const bool full = ty0 + CB_T <= a.H;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int lp = (ptid >> 3) + 32 * k;
      const u32x4 v = *reinterpret_cast<const u32x4*>(sO + lp * 128 + ((item ^ (lp & 7)) << 4));
      __builtin_amdgcn_raw_buffer_store_b128(v, rsrc_out, vo, rstep * k, 0);   // rows below the image fall off the descriptor
      if (do_stats) {
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
    if (a.pool_out[0] != nullptr && ch_ok) {
      // raw 2x2-pooled copy: per-channel max (gamma >= 0) or min (gamma < 0) of the window (see conv_bf16_kernel)
      uint16_t* const p_pool = a.pool_out[view];
      float gsign[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) gsign[e] = a.pool_gamma[co0 + e];
      const int Hp = a.H >> 1, Wp = a.W >> 1;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int pp = (ptid >> 3) + 32 * k;
        const int py = pp >> 3, px = pp & 7;
        const int oy = (ty0 >> 1) + py, ox = (tx0 >> 1) + px;
        if (oy >= Hp || ox >= Wp) continue;
        const int lp = (2 * py) * 16 + 2 * px;
        auto rd = [&](int q) { return *reinterpret_cast<const u32x4*>(sO + q * 128 + ((item ^ (q & 7)) << 4)); };
        const u32x4 v0 = rd(lp), v1 = rd(lp + 1), v2 = rd(lp + 16), v3 = rd(lp + 17);
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
  };

  // ---- prologue: stage 0 (and its weights) synchronously, stage 1's loads in flight ----
  int ld_u = u0, ld_chunk = 0;          // the stage whose loads are in flight (stage s + 1 inside the loop)
  auto advance = [&]() {                // -> the following stage (decodes a new unit); false past the end
    if (ld_chunk + 1 < a.nchunks) { ++ld_chunk; return true; }
    ld_chunk = 0; ld_u += nslot;
    if (ld_u >= u_end) return false;
    decode(ld_u);
    return true;
  };
  decode(ld_u);
  issue(0);
  stage_weights(ld_cob, 0);
  stage_halo(0);
  bool ld_ok = advance();
  if (ld_ok) issue(ld_chunk);
  int cs_u = u0, cs_chunk = 0;          // the stage the consumers compute
  __syncthreads();
  for (int s = 0; s < nstages; ++s) {
    if (ld_ok) {
      stage_weights(ld_cob, ld_chunk);   // slot = parity of the chunk: never the slot the consumers read (nchunks is even)
      stage_halo((s + 1) & 1);
      ld_ok = advance();
      if (ld_ok) issue(ld_chunk);        // stage s + 2: in flight across the barrier and the next stage
    }
    if (cs_chunk == 0 && s > 0) copy_out(cs_u - nslot);   // the unit that finished with stage s - 1
    if (++cs_chunk == a.nchunks) { cs_chunk = 0; cs_u += nslot; }
    __syncthreads();
  }
  copy_out(cs_u - nslot);
  if (a.stats[0] != nullptr && st_key >= 0) flush_stats(st_key);
}

}  // namespace sspk
