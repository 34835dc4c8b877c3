// Packed (two-wide) fp32 vector arithmetic pinned to the v_pk_* instructions through inline asm, shared by the Winograd
// kernels and the first-layer kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "det.hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int ssp_u32x4 __attribute__((ext_vector_type(4)));

// 16-byte buffer store whose data registers stay untouched for one more wait state.  gfx950 reads the data of a store of more than
// 8 bytes from the register file for a few cycles after issue; a vector instruction that overwrites one of those registers in the
// NEXT slot corrupts the last lanes of each 16-lane row of the stored value (found in round 5: v_lshlrev_b32 v6, 16, v7 straight
// behind buffer_store_dwordx4 v[6:9] stored v7 << 16 in lanes 12-15 / 28-31 / 44-47 / 60-63).  LLVM pads this hazard for the
// vector instructions the COMPILER emits; it cannot see a vector write inside INLINE ASM - the exposure of this code base, whose
// packed arithmetic below is asm throughout (the shift of the round-5 bug came from such a helper).  The asm here keeps the value
// alive across an s_nop, and hipbuild.verify_binary scans every kernel of the library for the pattern (store_data_hazards): a new
// asm helper that may overwrite a register a wide store has just read needs the same treatment.
__device__ __forceinline__ void ssp_store_b128(ssp_u32x4 v, __amdgpu_buffer_rsrc_t rsrc, unsigned voffset, int soffset) {
  __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, voffset, soffset, 0);
  asm volatile("s_nop 0" : : "v"(v) : "memory");
}

namespace sspk {

typedef float f32x2 __attribute__((ext_vector_type(2)));

// Two-wide fp32 vector arithmetic pinned to the packed instructions (hipcc scalarises most f32x2 expressions, in
// particular every subtraction): one VALU issue slot for two values.  The fp32 MFMA executes on the vector ALUs of
// this chip, so every VALU instruction saved in a Winograd kernel is matrix-pipe time.
// HAZARDS: hipcc does not insert wait states for registers read or written by inline asm.  (a) A register written
// here must not be read by an MFMA within the next 2 instructions (complete all operands, fence, then issue the
// MFMAs); (b) accumulators of in-flight MFMAs must not be read here (mfma_results_guard() before the first use);
// (c) the result of a transcendental instruction (v_exp_f32 ...) must not be read here by the very next instruction
// (sem_kernels.hip.h: pk_exp2 carries the wait state).
// `volatile` keeps the program order of these statements among themselves.
__device__ __forceinline__ f32x2 pk_add(f32x2 x, f32x2 y) {
  f32x2 d;
  asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y));
  return d;
}
__device__ __forceinline__ f32x2 pk_sub(f32x2 x, f32x2 y) {
  f32x2 d;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(x), "v"(y));
  return d;
}
__device__ __forceinline__ f32x2 pk_nadd(f32x2 x, f32x2 y) {  // -(x + y)
  f32x2 d;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[1,1] neg_hi:[1,1]" : "=v"(d) : "v"(x), "v"(y));
  return d;
}
__device__ __forceinline__ f32x2 pk_fma(f32x2 x, f32x2 y, f32x2 z) {
  f32x2 d;
  asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(x), "v"(y), "v"(z));
  return d;
}
__device__ __forceinline__ f32x2 pk_mul(f32x2 x, f32x2 y) {
  f32x2 d;
  asm("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y));
  return d;
}
__device__ __forceinline__ f32x2 lo2(f32x4 v) { return __builtin_shufflevector(v, v, 0, 1); }
__device__ __forceinline__ f32x2 hi2(f32x4 v) { return __builtin_shufflevector(v, v, 2, 3); }
__device__ __forceinline__ f32x4 cat2(f32x2 l, f32x2 h) { return __builtin_shufflevector(l, h, 0, 1, 2, 3); }
__device__ __forceinline__ f32x4 pk4_add(f32x4 x, f32x4 y) { return cat2(pk_add(lo2(x), lo2(y)), pk_add(hi2(x), hi2(y))); }
__device__ __forceinline__ f32x4 pk4_sub(f32x4 x, f32x4 y) { return cat2(pk_sub(lo2(x), lo2(y)), pk_sub(hi2(x), hi2(y))); }
__device__ __forceinline__ f32x4 pk4_fma(f32x4 x, f32x4 y, f32x4 z) {
  return cat2(pk_fma(lo2(x), lo2(y), lo2(z)), pk_fma(hi2(x), hi2(y), hi2(z)));
}
__device__ __forceinline__ f32x4 pk4_fma_s(f32x2 s, f32x4 y, f32x4 z) {  // s (both halves) * y + z
  return cat2(pk_fma(s, lo2(y), lo2(z)), pk_fma(s, hi2(y), hi2(z)));
}
// All MFMAs issued so far have written their accumulators (16-pass MFMA: 18 wait states) - before inline asm reads them.
__device__ __forceinline__ void mfma_results_guard() {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_nop 15\n\ts_nop 3");
  __builtin_amdgcn_sched_barrier(0);
}

// ---- bf16 <-> fp32 (the bf16 path, conv algorithm 12): two bf16 values per dword, element 0 in the low half ----
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
// (NOT __builtin_bit_cast(float, v[e]) on a vector ELEMENT: hipcc / ROCm 7.2 reads element 0 for every e - pass the element
// by value through these helpers)
__device__ __forceinline__ float u32_as_f32(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ float bf16_lo(uint32_t u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {  // round to nearest even (v_cvt_pk_bf16_f32)
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float round_bf16(float v) { return bf16_lo(pack_bf16(v, 0.f)); }  // fp32 value of bf16(v)

// x.lo (BCAST_HI = false) or x.hi (true) broadcast to both halves, times y, plus z: a scalar operand that lives in one
// half of a register pair feeds two channels (the 1 -> 64 first-layer kernels: taps x channel pairs).
template <bool BCAST_HI>
__device__ __forceinline__ f32x2 pk_fma_bcast(f32x2 x, f32x2 y, f32x2 z) {
  f32x2 d;
  if (BCAST_HI) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(x), "v"(y), "v"(z));
  else asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(d) : "v"(x), "v"(y), "v"(z));
  return d;
}

}  // namespace sspk
