// Small device helpers shared by the fp32-MFMA kernel files (mfma_ops.hip, rowjobs.hip): vector types, the 32x32
// accumulator layout, the lane <-> lane^32 exchange, raw-buffer weight fragment loads.
#pragma once
#include "common.h"

namespace ptt {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// C/D layout of v_mfma_f32_32x32x2_f32: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
__device__ __forceinline__ int tile_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

__device__ __forceinline__ float xor32(float v) { return __shfl_xor(v, 32, 64); }

// lane <-> lane^32 combine on v_permlane32_swap (one VALU op, no LDS round trip): after the swap `lo`
// holds the lower half-wave's values in both halves and `hi` the upper half-wave's.
// Inline asm on purpose: with both operands holding the SAME value hipcc (ROCm 7.2) folds
// __builtin_amdgcn_permlane32_swap's two results into one and the exchange silently disappears
// (scripts/probes/permlane_probe.hip). s_nop 1 = the VALU-write -> v_permlane read hazard, inside the string.
__device__ __forceinline__ void swap_halves(float v, float& lo, float& hi) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    lo = a; hi = b;
}
__device__ __forceinline__ float max_halves(float v) { float lo, hi; swap_halves(v, lo, hi); return fmaxf(lo, hi); }
__device__ __forceinline__ float add_halves(float v) { float lo, hi; swap_halves(v, lo, hi); return lo + hi; }

// Weight fragments are fetched with raw buffer loads: address = descriptor base (SGPRs) + per-lane byte offset (one
// VGPR, constant for the whole GEMM) + wave-uniform byte offset (SGPR, advanced by the scalar unit). The flat
// global_load form made hipcc recompute a 64-bit VGPR address per fragment (v_add_co / v_addc pairs): ~10 vector-ALU
// instructions per K-block that steal issue time from the fp32 MFMAs sharing the SIMD (DESIGN.md lesson 8).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t weight_rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);   // raw, 32-bit data format
}
__device__ __forceinline__ f32x4 weight_load(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

}  // namespace ptt
