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
#include <type_traits>

#include "conv_bf16.hip.h"

#ifndef CONVB_WS_ABL
#define CONVB_WS_ABL 0   // compile-time perf ablations of the INNER loops (SSP_HIPCC_EXTRA=-DCONVB_WS_ABL=n): 1 halo loads out of bounds
                         // (zeros, no HBM reads), 128 no staging arithmetic, 256 no LDS writes of the staging.  The whole-phase knobs
                         // 2 (no staging), 4 (no copy-out), 8 (no MFMA loop) stay runtime bits of SSP_CONVB_ABLATE: a uniform test per
                         // slot in the staging loop cost a branch per slot (round 5)
#endif

namespace sspk {

// packed fp32 arithmetic WITHOUT `volatile` (pk_math.hip.h's forms are volatile: the scheduler treats a side-effecting asm as a
// barrier for memory operations, which pinned an s_waitcnt lgkmcnt(0) behind every ds_read_b128 of the copy-out) and with in-place
// accumulators (a fresh output register per operation cost 88 v_mov_b64 per copy-out)
__device__ __forceinline__ f32x2 ws_pk_fma(f32x2 x, f32x2 y, f32x2 z) {
  f32x2 d;
  asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(x), "v"(y), "v"(z));
  return d;
}
__device__ __forceinline__ f32x2 ws_pk_mul(f32x2 x, f32x2 y) {
  f32x2 d;
  asm("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y));
  return d;
}
__device__ __forceinline__ f32x2 ws_pk_sub(f32x2 x, f32x2 y) {
  f32x2 d;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(x), "v"(y));
  return d;
}
__device__ __forceinline__ void ws_pk_acc(f32x2& sum, f32x2& sumsq, f32x2 x) {
  asm("v_pk_add_f32 %0, %0, %1" : "+v"(sum) : "v"(x));
  asm("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(sumsq) : "v"(x));
}

struct ConvWsGeom {
  static constexpr int HT = CB_T + 2;
  static constexpr int H_BYTES = HT * HT * CB_PS;         // 25,920
  static constexpr int W_BYTES = 9 * 2 * 2 * 1024;        // 36,864
  static constexpr int O_BYTES = 256 * 128;               // 32,768 (16-byte items XOR-swizzled by the pixel index)
  static constexpr int LDS_BYTES = 2 * W_BYTES + 2 * H_BYTES + O_BYTES;
};

// Threads per workgroup: BOTH modes run THREE roles, 4 + 4 + 4 waves (consumers / stagers or LDS-DMA issuers / copy-out).  With two
// roles the producers were a single latency-bound instruction stream per SIMD (measured WITHOUT the MFMA loop: 4300-4760 cycles per
// stage for ~320 instructions - loads, LDS round trips and stores of the halo staging and of the copy-out wait for each other;
// profiles/r05_bf16_ws_role_split.txt), so the copy-out / statistics / pooled copy (forward) resp. the copy-out with the fused
// BatchNorm-backward sums (data gradient) of a finished unit runs on a third wave per SIMD beside the staging wave.  (The two-role
// form of round 4 - copy-out on the producer waves, 512 threads - is gone: 168 registers per wave is the budget of every role.)
template <int IN_MODE>
constexpr int conv_ws_threads() { return 768; }

template <int IN_MODE, bool NC2>
__global__ __launch_bounds__(conv_ws_threads<IN_MODE>(), 1) void conv_bf16_ws_kernel(const ConvBArgs a) {
  using G = ConvWsGeom;
  constexpr int HT = G::HT, PAD = 1;
  constexpr bool DMA = IN_MODE == 0;   // the data gradient stages nothing: its halo goes global -> LDS directly (buffer_load ... lds)
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
    if (a.ablate & 64) __builtin_amdgcn_s_setprio(3);
    const int lj = lane & 31, lg = lane >> 5;
    int pr, pc;
    cb_lane_pixel<HT>(lj, pr, pc);
    // B-operand addresses.  Staged form (IN_MODE 1): halo slots of CB_PS = 80 bytes, one base per pixel tile, taps as immediates.
    // DMA form (IN_MODE 0): dense 64-byte slots whose four 16-byte items are XOR-swizzled by bits 2..3 of the slot index (the image
    // buffer_load ... lds can write); the swizzle depends on the slot, so every (tile, tap, k-step) has its own address register -
    // computed once, they do not depend on the unit.
    int boff[2];
    int baddr[DMA ? 2 : 1][DMA ? 9 : 1];   // k-step 0; k-step 1 = the same slot's item ^ 2, i.e. address ^ 32 (one v_xor per read instead
                                          // of 18 more address registers: the three-role form lives on 168 registers)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      boff[nt] = ((4 * wave + 2 * nt + pr) * HT + pc) * CB_PS + lg * 16;
      if constexpr (DMA) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
          {
            const int sl = (4 * wave + 2 * nt + pr + tap / 3) * HT + pc + tap % 3;
            baddr[nt][tap] = sl * 64 + ((lg ^ ((sl >> 2) & 3)) << 4);
          }
      }
    }
    const int aoff = lane * 16;
    f32x16 acc[2][2];
    f32x4 bias4[2][4];
    int chunk = 0, u = u0, bias_vc = -1;
    const bool tr = a.trace != nullptr && blockIdx.x == 0;   // (perf-debug: cycle sums of [MFMA loop, rest of the stage, barrier])
    unsigned long long tc[3] = {0, 0, 0}, t0 = 0, t1 = 0;
    const unsigned long long tk0 = tr ? __builtin_readcyclecounter() : 0, tr0 = a.trace != nullptr ? __builtin_amdgcn_s_memrealtime() : 0;
    __syncthreads();   // the producers' prologue has staged stage 0
    auto stage = [&](auto PAR) __attribute__((always_inline)) {   // PAR = parity of the stage = of its chunk (nchunks is even)
      constexpr int Q = decltype(PAR)::value;
      if (tr) t0 = __builtin_readcyclecounter();
      if (chunk == 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        if constexpr (!DMA) {   // (the data-gradient form carries no bias: launch_conv_bf16 sends a biased in_mode-0 call to the generic kernel)
          const int vc = u / T;
          if (vc != bias_vc) { bias_vc = vc; cb_load_bias(a.bias, a.Cout, vc % a.ncob, lg, bias4); }   // (a few times per launch)
        }
      }
      const unsigned char* const pW = sW + Q * G::W_BYTES + aoff;
      const unsigned char* const pH = sH + Q * G::H_BYTES;
      // operands run PF k-steps ahead of the MFMAs that use them: with ONE consumer wave per SIMD nothing else hides the
      // ds_read latency
      constexpr int PF = DMA ? 1 : 2;   // (the staging-free form carries 18 operand address registers: one set less keeps it at 168; round 4
                                        //  measured no difference between 1, 2 and 3 k-steps of prefetch)
      s16x8 fa[PF + 1][2], fb[PF + 1][2];
      auto fetch = [&](int step, s16x8 (&qa)[2], s16x8 (&qb)[2]) __attribute__((always_inline)) {
        const int tap = step >> 1, ks = step & 1;
        const int dy = tap / 3, dx = tap % 3;
        qa[0] = *reinterpret_cast<const s16x8*>(pW + ((tap * 2 + ks) * 2 + 0) * 1024);
        qa[1] = *reinterpret_cast<const s16x8*>(pW + ((tap * 2 + ks) * 2 + 1) * 1024);
        if constexpr (DMA) {
          qb[0] = *reinterpret_cast<const s16x8*>(pH + (baddr[0][tap] ^ (ks << 5)));
          qb[1] = *reinterpret_cast<const s16x8*>(pH + (baddr[1][tap] ^ (ks << 5)));
        } else {
          qb[0] = *reinterpret_cast<const s16x8*>(pH + boff[0] + (dy * HT + dx) * CB_PS + ks * 32);
          qb[1] = *reinterpret_cast<const s16x8*>(pH + boff[1] + (dy * HT + dx) * CB_PS + ks * 32);
        }
      };
      if (!(a.ablate & 8)) {
#pragma unroll
      for (int p = 0; p < PF; ++p) fetch(p, fa[p], fb[p]);
#pragma unroll
      for (int step = 0; step < 18; ++step) {
        const int cur = step % (PF + 1);
        if (step + PF < 18) fetch(step + PF, fa[(step + PF) % (PF + 1)], fb[(step + PF) % (PF + 1)]);
        __builtin_amdgcn_sched_barrier(0);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][0], fb[cur][0], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][0], fb[cur][1], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][1], fb[cur][0], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][1], fb[cur][1], acc[1][1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      }
      if (tr) { t1 = __builtin_readcyclecounter(); tc[0] += t1 - t0; }
      if (chunk + 1 == a.nchunks) {
        // finished unit: bias, rounding, hand the tile to the producers through LDS (they copied the previous one out a stage ago)
        auto hand_over = [&](auto BIAS) __attribute__((always_inline)) {   // (the data gradient has no bias: 64 additions per tile less)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            const int lp = (4 * wave + 2 * nt + pr) * 16 + pc;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                f32x4 v = {acc[mt][nt][4 * q], acc[mt][nt][4 * q + 1], acc[mt][nt][4 * q + 2], acc[mt][nt][4 * q + 3]};
                if (decltype(BIAS)::value) v += bias4[mt][q];
                u32x2 o;
                o[0] = pack_bf16(v[0], v[1]);
                o[1] = pack_bf16(v[2], v[3]);
                *reinterpret_cast<u32x2*>(sO + lp * 128 + (((mt * 4 + q) ^ (lp & 7)) << 4) + lg * 8) = o;
              }
          }
        };
        if (!DMA && a.bias != nullptr) hand_over(std::true_type{}); else hand_over(std::false_type{});
        chunk = 0; u += nslot;
      } else {
        ++chunk;
      }
      if (tr) { t0 = __builtin_readcyclecounter(); tc[1] += t0 - t1; }
      __syncthreads();
      if (tr) tc[2] += __builtin_readcyclecounter() - t0;
    };
    for (int s = 0; s < nstages; s += 2) {   // (nstages is even: nchunks is)
      stage(std::integral_constant<int, 0>{});
      stage(std::integral_constant<int, 1>{});
    }
    if (a.trace != nullptr && wave == 0 && lane == 0 && blockIdx.x < 512) {   // every workgroup: [start, end] in 100 MHz ticks
      a.trace[64 + 2 * blockIdx.x] = tr0;
      a.trace[65 + 2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
    }
    if (tr && lane == 0) {
#pragma unroll
      for (int k = 0; k < 3; ++k) a.trace[wave * 8 + k] = tc[k];
      a.trace[wave * 8 + 7] = nstages;
      a.trace[wave * 8 + 5] = __builtin_readcyclecounter() - tk0;        // shader-clock cycles and 100 MHz ticks of the whole loop
      a.trace[wave * 8 + 6] = __builtin_amdgcn_s_memrealtime() - tr0;
    }
    return;
  }

  // =========================== producers ===========================
  const bool copier = wave >= 8;                          // (scalar) the copy-out role: waves 4-7 stage / issue the LDS-DMA, waves 8-11 copy out
  const int ptid = (tid - 256) & 255, pwave = (wave - 4) & 3;
  const int part = ptid & 3;
  constexpr int NHS = (HT * HT * 4 + 255) / 256;   // 6 (the last round is partly empty)
  const int hs_lds0 = (ptid >> 2) * CB_PS + part * 16;
  unsigned hs_g[NHS];
  int hs_rel[NHS];
  int hs_yx0 = 0;   // halo (row, column) of this thread's first slot; slot i is 64 = 3 x 18 + 10 further (kept as ONE register: six
                    // spilled the staging loop into scratch)
#pragma unroll
  for (int i = 0; i < NHS; ++i) {
    // staged form: thread -> (slot ptid / 4 + 64 i, item ptid % 4).  DMA form: 16-byte unit q = ptid + 256 i of the dense image ->
    // slot q / 4, whose item at position q % 4 is channel item (q % 4) ^ ((slot / 4) % 4)
    const int sl = (ptid >> 2) + 64 * i;
    const int cpart = DMA ? (part ^ ((sl >> 2) & 3)) : part;
    const int hy = sl / HT, hx = sl - hy * HT;
    if (i == 0) hs_yx0 = (hy << 16) | hx;
    hs_rel[i] = (((hy - PAD) * a.W + (hx - PAD)) * a.in_cs + a.in_co + cpart * 8) * 2;
  }
  __amdgpu_buffer_rsrc_t rsrc_in;
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t*>(a.wpk), 0, (unsigned)(a.ncob * a.nchunks * G::W_BYTES), 0x00020000);

  // ---- unit walker: (vc = view * ncob + cob, image, tile row, tile column) of the unit whose stages are being LOADED.  Units of a
  // workgroup are nslot apart: the step is decomposed once and added with carries (the five integer divisions of a from-scratch
  // decode were ~1100 cycles per tile on the scalar unit, twice: loads and copy-out).  st1 / st2 = the two units before it: the
  // copy-out runs one (nchunks > 2) or two (nchunks == 2) units behind the loads. ----
  int ld_vc, ld_n, ld_ty, ld_tx;
  {
    const int vc = u0 / T, t = u0 - vc * T;
    const int t2 = t / a.tiles_x;
    ld_vc = vc; ld_tx = t - t2 * a.tiles_x; ld_n = t2 / a.tiles_y; ld_ty = t2 - ld_n * a.tiles_y;
  }
  int d_vc, d_n, d_ty, d_tx;
  {
    const int r1 = nslot / a.tiles_x, r2 = r1 / a.tiles_y;
    d_tx = nslot - r1 * a.tiles_x; d_ty = r1 - r2 * a.tiles_y; d_vc = r2 / a.N; d_n = r2 - d_vc * a.N;
  }
  int ld_view = 0, ld_cob = 0;
  bool ld_border = false;
  auto setup_unit = [&]() __attribute__((always_inline)) {   // descriptor + per-slot offsets of the unit (ld_vc, ld_n, ld_ty, ld_tx)
    ld_view = ld_vc >= a.ncob ? 1 : 0; ld_cob = ld_vc - ld_view * a.ncob;
    const int ty0 = ld_ty * CB_T, tx0 = ld_tx * CB_T;
    rsrc_in = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(a.in[ld_view])) + (size_t)ld_n * a.in_img_bytes, 0, a.in_img_bytes, 0x00020000);
    const int org = (ty0 * a.W + tx0) * a.in_cs * 2;
    ld_border = !(ty0 >= PAD && tx0 >= PAD && ty0 + CB_T + PAD <= a.H && tx0 + CB_T + PAD <= a.W);
    if (!ld_border) {
#pragma unroll
      for (int i = 0; i < NHS; ++i) hs_g[i] = (ptid >> 2) + 64 * i < HT * HT ? (unsigned)(org + hs_rel[i]) : OOB;
    } else {
#pragma unroll
      for (int i = 0; i < NHS; ++i) {
        int yx = hs_yx0;
        asm volatile("" : "+v"(yx));   // (opaque: hipcc would hoist the six coordinate pairs out of the tile loop and keep them alive)
        int hy = (yx >> 16) + 3 * i, hx = (yx & 0xffff) + 10 * i;   // + 64 slots per round
        hy += hx / HT; hx %= HT;                                              // (constants: hx < 18 + 50)
        const int gy = ty0 + hy - PAD, gx = tx0 + hx - PAD;
        const bool ok = (ptid >> 2) + 64 * i < HT * HT && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
        hs_g[i] = ok ? (unsigned)(org + hs_rel[i]) : OOB;
      }
    }
  };
  int ld_u = u0, ld_chunk = 0;          // the stage whose loads are issued next
  auto advance = [&]() __attribute__((always_inline)) {                // -> the following stage; false past the end
    if (ld_chunk + 1 < a.nchunks) { ++ld_chunk; return true; }
    ld_chunk = 0; ld_u += nslot;
    if (ld_u >= u_end) return false;
    ld_tx += d_tx; int c = ld_tx >= a.tiles_x ? 1 : 0; ld_tx -= c ? a.tiles_x : 0;
    ld_ty += d_ty + c; c = ld_ty >= a.tiles_y ? 1 : 0; ld_ty -= c ? a.tiles_y : 0;
    ld_n += d_n + c; c = ld_n >= a.N ? 1 : 0; ld_n -= c ? a.N : 0;
    ld_vc += d_vc + c;
    setup_unit();
    return true;
  };

  // ---- halo loads: TWO register sets (stage parity), each in flight for two whole stages before it is staged ----
  u32x4 hv[2][NHS];
  unsigned hv_pad[2] = {0, 0};   // bit i: slot i of the set is padding (outside the image)
  int hv_view[2] = {0, 0};       // view of the set's unit (affine of the 64-channel form), border flag in bit 8
  // affine (IN_MODE 1).  NC2 (a layer of 64 input channels): both chunks' scale / shift stay in registers per view; otherwise they
  // are loaded with the halo, per set.
  f32x2 sc2[2][4], sh2[2][4];
  int aff_view = -1;
  auto load_affine = [&](int set_or_chunk, int view, int chunk) __attribute__((always_inline)) {
    const int c0 = chunk * CB_KC + part * 8;
    const float* const p_scale = a.in_scale[view] + c0;
    const float* const p_shift = a.in_shift[view] + c0;
    const f32x4 s0 = *reinterpret_cast<const f32x4*>(p_scale), s1 = *reinterpret_cast<const f32x4*>(p_scale + 4);
    const f32x4 h0 = *reinterpret_cast<const f32x4*>(p_shift), h1 = *reinterpret_cast<const f32x4*>(p_shift + 4);
    sc2[set_or_chunk][0] = f32x2{s0[0], s0[1]}; sc2[set_or_chunk][1] = f32x2{s0[2], s0[3]};
    sc2[set_or_chunk][2] = f32x2{s1[0], s1[1]}; sc2[set_or_chunk][3] = f32x2{s1[2], s1[3]};
    sh2[set_or_chunk][0] = f32x2{h0[0], h0[1]}; sh2[set_or_chunk][1] = f32x2{h0[2], h0[3]};
    sh2[set_or_chunk][2] = f32x2{h1[0], h1[1]}; sh2[set_or_chunk][3] = f32x2{h1[2], h1[3]};
  };
  auto issue = [&](auto SET) __attribute__((always_inline)) {   // halo (+ affine) of (ld_ unit, ld_chunk) -> register set SET
    constexpr int S = decltype(SET)::value;
    unsigned pad = 0;
#pragma unroll
    for (int i = 0; i < NHS; ++i) {
      hv[S][i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_in, (CONVB_WS_ABL & 1) ? OOB : hs_g[i], ld_chunk * CB_KC * 2, 0));
      pad |= (hs_g[i] == OOB ? 1u : 0u) << i;
    }
    hv_pad[S] = pad;
    hv_view[S] = ld_view | (ld_border ? 256 : 0);
    if (IN_MODE == 1 && !NC2) load_affine(S, ld_view, ld_chunk);
  };
  auto stage_halo = [&](auto SET) __attribute__((always_inline)) {   // register set SET -> activated bf16 operands in halo buffer SET
    constexpr int S = decltype(SET)::value;
    if (a.ablate & 2) return;
    unsigned char* const dst = sH + S * G::H_BYTES + hs_lds0;
    // BatchNorm + ReLU + rounding of 8 values in 20 vector instructions: 8 shifts / masks (bf16 -> fp32), 4 packed fmas, 4
    // v_cvt_pk_bf16_f32 and the ReLU on the PACKED result as a 16-bit integer max with 0 (a negative bf16 is a negative int16;
    // rounding is monotonic and odd, so relu(round(z)) == round(relu(z)))
    typedef short s16x2 __attribute__((ext_vector_type(2)));
    if (IN_MODE == 1 && NC2 && (hv_view[S] & 255) != aff_view) {   // (once or twice per launch)
      aff_view = hv_view[S] & 255;
      load_affine(0, aff_view, 0);
      load_affine(1, aff_view, 1);
    }
    auto run = [&](auto BORDER) __attribute__((always_inline)) {   // (one straight-line body per variant: a uniform test inside the
      // slot loop became a branch per slot - the inner-loop ablations are compile-time for the same reason).  Two slots' values
      // are computed before their LDS writes: 8 independent 5-instruction chains for the scheduler instead of one chain per slot
      // (three or six at once spilled registers in the 64-channel form)
      constexpr int GS = 2;
      static_assert(NHS % GS == 0, "whole groups of slots");
#pragma unroll
      for (int g3 = 0; g3 < NHS; g3 += GS) {
      u32x4 o[NHS];
#pragma unroll
      for (int i = g3; i < g3 + GS; ++i) {
        if (IN_MODE == 1 && !(CONVB_WS_ABL & 128)) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const f32x2 z = ws_pk_fma(f32x2{bf16_lo(hv[S][i][e]), bf16_hi(hv[S][i][e])}, sc2[S][e], sh2[S][e]);
            const s16x2 r = __builtin_elementwise_max(__builtin_bit_cast(s16x2, pack_bf16(z[0], z[1])), s16x2{0, 0});
            o[i][e] = __builtin_bit_cast(uint32_t, r);
          }
          if (decltype(BORDER)::value) {   // padding is zero in the ACTIVATED domain
            const bool pad = (hv_pad[S] >> i) & 1u;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[i][e] = pad ? 0u : o[i][e];
          }
        } else {
          o[i] = hv[S][i];
        }
      }
#pragma unroll
      for (int i = g3; i < g3 + GS; ++i) {
        if (i == NHS - 1 && (ptid >> 2) + 64 * i >= HT * HT) continue;   // (only the last round is partly empty: 6 x 64 slots >= 324)
        if (!(CONVB_WS_ABL & 256)) *reinterpret_cast<u32x4*>(dst + i * 64 * CB_PS) = o[i];
        else asm volatile("" :: "v"(o[i]));
      }
      }
    };
    if ((hv_view[S] & 256) != 0) run(std::true_type{}); else run(std::false_type{});   // (uniform) border tile of the image?
  };
  // weights: the 36 KB image of (cob, chunk) lives in slot chunk & 1 (never the slot the consumers read: nchunks is even).  A layer
  // of 64 input channels loads its two images once; wider layers swap a slot every stage, so the image of stage s + 2 is fetched
  // into registers an iteration ahead (in flight for a whole stage) and written to LDS one iteration later.
  int w_tag0 = -1, w_tag1 = -1;     // (two scalars: a dynamically indexed array would live in scratch memory)
  u32x4 wv[9];
  int wv_tag = -1;
  auto fetch_weights = [&](int tag, int par) __attribute__((always_inline)) {
    if (tag < 0 || (par ? w_tag1 : w_tag0) == tag) return;
    wv_tag = tag;
#pragma unroll
    for (int i = 0; i < 9; ++i) wv[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, ptid * 16, tag * G::W_BYTES + 4096 * i, 0));
  };
  auto commit_weights = [&](int par) __attribute__((always_inline)) {
    if (wv_tag < 0) return;
    if (par) w_tag1 = wv_tag; else w_tag0 = wv_tag;
    wv_tag = -1;
#pragma unroll
    for (int i = 0; i < 9; ++i) *reinterpret_cast<u32x4*>(sW + par * G::W_BYTES + (ptid + 256 * i) * 16) = wv[i];
  };

  // wide layers (more than 64 input channels: a new image every stage) of the staged form: the image of stage s + 1 goes global ->
  // LDS (9 buffer_load ... lds per wave, no registers, no ds_write) at the start of iteration s, lands under the staging arithmetic
  // and is waited for BEFORE the next halo loads are issued (so the wait never touches the prefetch)
  typedef __attribute__((address_space(3))) void* w_ldsp;
  auto dma_weights = [&](int tag, int par) __attribute__((always_inline)) {
    if (tag < 0 || (par ? w_tag1 : w_tag0) == tag) return false;
    if (par) w_tag1 = tag; else w_tag0 = tag;
#pragma unroll
    for (int i = 0; i < 9; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (w_ldsp)(sW + par * G::W_BYTES + (256 * i + 64 * pwave) * 16), 16, ptid * 16,
                                               tag * G::W_BYTES + 4096 * i, 0, 0);
    return true;
  };

  // statistics of the stored values: per-thread sums of this thread's 8 channels (item = ptid & 7), flushed per (view, cob)
  const int item = ptid & 7;
  f32x2 ps[4], pq[4];
  f32x2 b_sc[4], b_sh[4], b_mu[4], b_is[4];   // (fused BatchNorm-backward sums: parameters of this thread's 8 channels)
#pragma unroll
  for (int e = 0; e < 4; ++e) ps[e] = pq[e] = b_sc[e] = b_sh[e] = b_mu[e] = b_is[e] = f32x2{0.f, 0.f};
  int st_key = -1, pg_key = -1;
  uint32_t pg_neg[4] = {0u, 0u, 0u, 0u};
  const bool do_bnr = DMA && a.bnr_t[0] != nullptr;   // (uniform) fused BatchNorm-backward sums of the layer below (data gradient)
  auto flush_stats = [&](int key) __attribute__((always_inline)) {
    const int view = key >= a.ncob ? 1 : 0, cob = key - view * a.ncob;
    double* const p_stats = do_bnr ? a.bnr_sums[view] : a.stats[view];
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
          acc_add_stats_or_grad(st + co, (double)v[e], do_bnr);
          acc_add_stats_or_grad(st + a.Cout + co, (double)v[8 + e], do_bnr);
        }
      }
    }
  };

  // copy-out of a finished unit from the LDS tile: 16-byte stores, statistics of the stored values, raw pooled copy
  auto copy_out = [&](int vc, int n, int yx) __attribute__((always_inline)) {
    if (a.ablate & 4) return;
    const int view = vc >= a.ncob ? 1 : 0, cob = vc - view * a.ncob;
    const int ty0 = (yx >> 16) * CB_T, tx0 = (yx & 0xffff) * CB_T;
    const bool do_stats = a.stats[0] != nullptr;
    const int co0 = cob * CB_NB + item * 8;
    if ((do_stats || do_bnr) && vc != st_key) {
      if (st_key >= 0) flush_stats(st_key);
      st_key = vc;
      if (do_bnr) {   // the layer-below parameters of this thread's 8 channels (a few times per launch)
        const bool okc = co0 < a.Cout;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int c = okc ? co0 + 2 * e : 0;
          b_sc[e] = f32x2{a.bnr_scale[view][c], a.bnr_scale[view][c + 1]};
          b_sh[e] = f32x2{a.bnr_shift[view][c], a.bnr_shift[view][c + 1]};
          b_mu[e] = f32x2{a.bnr_mean[view][c], a.bnr_mean[view][c + 1]};
          b_is[e] = f32x2{a.bnr_invstd[view][c], a.bnr_invstd[view][c + 1]};
        }
      }
    }
    const bool ch_ok = co0 < a.Cout;   // (Cout % 8 == 0: host-checked)
    const int col = (ptid >> 3) & 15, row0 = ptid >> 7;
    unsigned char* const p_out = reinterpret_cast<unsigned char*>(a.out[view]);
    const unsigned out_img_bytes = (unsigned)a.H * a.W * a.out_cs * 2u;
    const __amdgpu_buffer_rsrc_t rsrc_out = __builtin_amdgcn_make_buffer_rsrc(p_out + (size_t)n * out_img_bytes, 0, out_img_bytes, 0x00020000);
    const bool col_ok = ch_ok && tx0 + col < a.W;
    const unsigned vo = col_ok ? (unsigned)((((ty0 + row0) * a.W + tx0 + col) * a.out_cs + a.out_co + co0) * 2) : OOB;
    const int rstep = 2 * a.W * a.out_cs * 2;
    const bool full = ty0 + CB_T <= a.H;
    const bool whole = full && tx0 + CB_T <= a.W && (cob + 1) * CB_NB <= a.Cout;   // (uniform) every element of the tile is stored
    // fused BatchNorm-backward sums: the layer-below tensor t at this thread's pixels (same channel block, dense [N,H,W,Cout])
    __amdgpu_buffer_rsrc_t rsrc_t = rsrc_out;
    if (do_bnr) rsrc_t = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.bnr_t[view]) + (size_t)n * a.H * a.W * a.Cout, 0,
                                                            (unsigned)a.H * a.W * a.Cout * 2u, 0x00020000);
    const unsigned vo_t = col_ok ? (unsigned)((((ty0 + row0) * a.W + tx0 + col) * a.Cout + co0) * 2) : OOB;
    const int rstep_t = 2 * a.W * a.Cout * 2;
    auto rows = [&](auto WHOLE, auto STATS, auto BNR) __attribute__((always_inline)) {   // (all uniform: straight-line code per variant)
#pragma unroll
      for (int kb = 0; kb < 8; kb += 4) {   // four LDS reads in flight before the first use
        u32x4 v[4], tv[4];
        if (decltype(BNR)::value) {   // (issued first: an L2 / HBM round trip)
#pragma unroll
          for (int k = 0; k < 4; ++k) tv[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_t, vo_t, rstep_t * (kb + k), 0));
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int lp = (ptid >> 3) + 32 * (kb + k);
          v[k] = *reinterpret_cast<const u32x4*>(sO + lp * 128 + ((item ^ (lp & 7)) << 4));
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          ssp_store_b128(v[k], rsrc_out, vo, rstep * (kb + k));   // rows below the image fall off the descriptor
          if (decltype(BNR)::value) {
            const bool ok = decltype(WHOLE)::value || (col_ok && (full || ty0 + row0 + 2 * (kb + k) < a.H));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const uint32_t w = ok ? v[k][e] : 0u, yb = tv[k][e];
              const f32x2 d2 = {bf16_lo(w), bf16_hi(w)}, y2 = {bf16_lo(yb), bf16_hi(yb)};
              const f32x2 z2 = ws_pk_fma(y2, b_sc[e], b_sh[e]);
              const f32x2 dz = {z2[0] > 0.f ? d2[0] : 0.f, z2[1] > 0.f ? d2[1] : 0.f};
              const f32x2 xh = ws_pk_mul(ws_pk_sub(y2, b_mu[e]), b_is[e]);
              asm("v_pk_add_f32 %0, %0, %1" : "+v"(ps[e]) : "v"(dz));
              asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(pq[e]) : "v"(dz), "v"(xh));
            }
          }
          if (decltype(STATS)::value) {
            const bool ok = decltype(WHOLE)::value || (col_ok && (full || ty0 + row0 + 2 * (kb + k) < a.H));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const uint32_t w = ok ? v[k][e] : 0u;
              ws_pk_acc(ps[e], pq[e], f32x2{bf16_lo(w), bf16_hi(w)});
            }
          }
        }
      }
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    if (do_stats) { if (whole) rows(T_{}, T_{}, F_{}); else rows(F_{}, T_{}, F_{}); }
    else if (do_bnr) { if (whole) rows(T_{}, F_{}, T_{}); else rows(F_{}, F_{}, T_{}); }
    else rows(T_{}, F_{}, F_{});
    if (a.pool_out[0] != nullptr && ch_ok) {
      // raw 2x2-pooled copy: per-channel max (gamma >= 0) or min (gamma < 0) of the window (see conv_bf16_kernel).  bf16 pairs are
      // compared as PACKED HALVES (v_pk_max_f16 / v_pk_min_f16): both formats are sign-magnitude with the exponent above the
      // mantissa, so the order of two finite bf16 values is the order of the same bits read as fp16 (|x| >= 2^121 would read as an
      // fp16 NaN, < 2^-119 as a denormal: not activations) - 3 instructions per pair of channels instead of unpack + fp32 max + pack.
      // The signs of this thread's 8 gammas become a 16-bit-lane mask when the channel block changes.
      uint16_t* const p_pool = a.pool_out[view];
      if (vc != pg_key) {
        pg_key = vc;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          pg_neg[e] = (a.pool_gamma[co0 + 2 * e] < 0.f ? 0x0000ffffu : 0u) | (a.pool_gamma[co0 + 2 * e + 1] < 0.f ? 0xffff0000u : 0u);
      }
      const bool any_neg = __builtin_amdgcn_ballot_w64((pg_neg[0] | pg_neg[1] | pg_neg[2] | pg_neg[3]) != 0u) != 0ull;   // (uniform)
      const int Hp = a.H >> 1, Wp = a.W >> 1;
      auto pkmax = [](uint32_t x, uint32_t y) __attribute__((always_inline)) { uint32_t r; asm("v_pk_max_f16 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; };
      auto pkmin = [](uint32_t x, uint32_t y) __attribute__((always_inline)) { uint32_t r; asm("v_pk_min_f16 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; };
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int pp = (ptid >> 3) + 32 * k;
        const int py = pp >> 3, px = pp & 7;
        const int oy = (ty0 >> 1) + py, ox = (tx0 >> 1) + px;
        const int lp = (2 * py) * 16 + 2 * px;
        auto rd = [&](int q) __attribute__((always_inline)) { return *reinterpret_cast<const u32x4*>(sO + q * 128 + ((item ^ (q & 7)) << 4)); };
        const u32x4 v0 = rd(lp), v1 = rd(lp + 1), v2 = rd(lp + 16), v3 = rd(lp + 17);
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = pkmax(pkmax(v0[e], v1[e]), pkmax(v2[e], v3[e]));
        if (any_neg) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const uint32_t mn = pkmin(pkmin(v0[e], v1[e]), pkmin(v2[e], v3[e]));
            o[e] = (mn & pg_neg[e]) | (o[e] & ~pg_neg[e]);
          }
        }
        if (oy < Hp && ox < Wp) *reinterpret_cast<u32x4*>(p_pool + ((size_t)(n * Hp + oy) * Wp + ox) * a.Cout + co0) = o;
      }
    }
  };

  {
    if (copier) {
      // ---- copy-out role (waves 8-11 of the forward): its own unit walker, one barrier per stage like everybody else; the unit
      // that finished with stage s - 1 leaves the LDS tile during stage s ----
      int co_vc = ld_vc, co_n = ld_n, co_ty = ld_ty, co_tx = ld_tx;
      __syncthreads();   // (the stagers' prologue)
      int cs = 0;
      for (int s = 0; s < nstages; ++s) {
        if (cs == 0 && s > 0) {
          copy_out(co_vc, co_n, (co_ty << 16) | co_tx);
          co_tx += d_tx; int c = co_tx >= a.tiles_x ? 1 : 0; co_tx -= c ? a.tiles_x : 0;
          co_ty += d_ty + c; c = co_ty >= a.tiles_y ? 1 : 0; co_ty -= c ? a.tiles_y : 0;
          co_n += d_n + c; c = co_n >= a.N ? 1 : 0; co_n -= c ? a.N : 0;
          co_vc += d_vc + c;
        }
        if (++cs == a.nchunks) cs = 0;
        __syncthreads();
      }
      copy_out(co_vc, co_n, (co_ty << 16) | co_tx);
      if ((a.stats[0] != nullptr || do_bnr) && st_key >= 0) flush_stats(st_key);
      return;
    }
  }
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  const bool tr = a.trace != nullptr && blockIdx.x == 0;   // (perf-debug: cycle sums of [copy-out, staging, issue, barrier, load wait, advance])
  unsigned long long tc[6] = {0, 0, 0, 0, 0, 0}, t0 = 0, t1 = 0;
  int cs_chunk = 0;                     // chunk of the stage the consumers compute
  if constexpr (DMA) {
    // ---- data gradient: the halo of stage s + 1 goes global -> LDS while the consumers compute stage s (6 buffer_load ... lds
    // per producer wave, out-of-image units read as zeros), the weights of a new (cob, chunk) through registers in the same
    // iteration; the producers' vector work is the copy-out alone ----
    typedef __attribute__((address_space(3))) void* ldsp;
    auto issue_dma = [&](auto PAR) __attribute__((always_inline)) {
      constexpr int Q = decltype(PAR)::value;
#pragma unroll
      for (int i = 0; i < NHS; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_in, (ldsp)(sH + Q * G::H_BYTES + (256 * i + 64 * pwave) * 16), 16,
                                                 (CONVB_WS_ABL & 1) ? OOB : hs_g[i], ld_chunk * CB_KC * 2, 0, 0);
    };
    setup_unit();
    issue_dma(P0{});
    fetch_weights(ld_cob * a.nchunks, 0);
    commit_weights(0);
    bool ld_ok = advance();             // -> stage 1
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    auto iteration = [&](int s, auto PAR) __attribute__((always_inline)) {   // PAR = parity of stage s + 1
      constexpr int Q = decltype(PAR)::value;
      if (tr) t0 = __builtin_readcyclecounter();
      if (ld_ok) {                       // ld_ = stage s + 1: its buffers were last read in stage s - 1
        issue_dma(PAR);
        fetch_weights(ld_cob * a.nchunks + ld_chunk, Q);
      }
      if (tr) { t1 = __builtin_readcyclecounter(); tc[2] += t1 - t0; }
      if (++cs_chunk == a.nchunks) cs_chunk = 0;
      if (tr) { t0 = __builtin_readcyclecounter(); tc[0] += t0 - t1; }
      if (ld_ok) {
        commit_weights(Q);
        ld_ok = advance();
      }
      if (tr) { t1 = __builtin_readcyclecounter(); tc[5] += t1 - t0; }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the DMA has landed (and the copy-out's stores have left)
      if (tr) { t0 = __builtin_readcyclecounter(); tc[4] += t0 - t1; }
      __syncthreads();
      if (tr) tc[3] += __builtin_readcyclecounter() - t0;
    };
    for (int s = 0; s < nstages; s += 2) {   // (nstages is even: nchunks is)
      iteration(s, P1{});
      iteration(s + 1, P0{});
    }
  } else {
  // ---- prologue: stage 0 (and its weights) synchronously; the loads of stages 1 and 2 in flight ----
  setup_unit();
  int w_next = -1;                      // weight tag of the stage whose halo was issued last
  int w_pend = -1;                      // ... and of the stage before it (= stage s + 1 at the start of iteration s)
  issue(P0{});
  fetch_weights(ld_cob * a.nchunks, 0);
  commit_weights(0);
  stage_halo(P0{});
  bool ld_ok = advance();               // -> stage 1
  if (ld_ok) {
    issue(P1{});
    if (NC2) fetch_weights(ld_cob * a.nchunks + ld_chunk, 1);   // (committed by iteration 0)
    else w_pend = ld_cob * a.nchunks + ld_chunk;
    ld_ok = advance();
  }
  if (ld_ok) { issue(P0{}); w_next = ld_cob * a.nchunks + ld_chunk; ld_ok = advance(); } else w_next = -1;   // stage 2; ld_ -> stage 3
  __syncthreads();
  // iteration s (the consumers compute stage s): copy-out of the unit that finished with stage s - 1, weights + halo of stage
  // s + 1 into the buffers of its parity, then the loads of stage s + 3 into the register set that staging freed
  auto iteration = [&](int s, auto PAR) __attribute__((always_inline)) {   // PAR = parity of stage s + 1
    constexpr int Q = decltype(PAR)::value;
    if (tr) t0 = __builtin_readcyclecounter();
    if (++cs_chunk == a.nchunks) cs_chunk = 0;
    if (tr) { t1 = __builtin_readcyclecounter(); tc[0] += t1 - t0; }
    if (s + 1 < nstages) {
      if (tr) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); t0 = __builtin_readcyclecounter(); tc[4] += t0 - t1; t1 = t0; }
      bool dma = false;
      if (NC2) {   // a 64-channel layer changes its images with the output block only: through registers, in place
        fetch_weights(w_pend, Q);
        commit_weights(Q);
      } else {
        dma = dma_weights(w_pend, Q);
      }
      stage_halo(PAR);
      if (dma) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the image has landed (nothing else is in flight here)
      if (tr) { t0 = __builtin_readcyclecounter(); tc[1] += t0 - t1; }
      w_pend = w_next;
      w_next = -1;
      if (ld_ok) {                       // ld_ = stage s + 3
        issue(PAR);
        w_next = ld_cob * a.nchunks + ld_chunk;
        if (tr) { t1 = __builtin_readcyclecounter(); tc[2] += t1 - t0; t0 = t1; }
        ld_ok = advance();
        if (tr) { t1 = __builtin_readcyclecounter(); tc[5] += t1 - t0; }
      }
    }
    if (tr) t0 = __builtin_readcyclecounter();
    __syncthreads();
    if (tr) tc[3] += __builtin_readcyclecounter() - t0;
  };
  for (int s = 0; s < nstages; s += 2) {   // (nstages is even: nchunks is)
    iteration(s, P1{});
    iteration(s + 1, P0{});
  }
  }
  if (tr && lane == 0) {
#pragma unroll
    for (int k = 0; k < 6; ++k) a.trace[wave * 8 + k] = tc[k];
    a.trace[wave * 8 + 7] = nstages;
  }
}

}  // namespace sspk
