// Shared host/device helpers for libptt_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/ptt_hip.h"

namespace ptt {

// thread-local last-error text, returned by ptt_last_error_string()
char* last_error_buf();
int fail(int code, const char* fmt, ...);
int check_launch(const char* what);

static inline hipStream_t as_stream(ptt_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Developer A/B switches (docs/experiments.md "Developer switches"). A release build of the library has none: the
// structure holds the measured-best defaults as constants, no entry point calls getenv, and the per-phase cycle
// stamps of the chained kernels are compiled out. A build with -DPTT_DEV (PTT_HIP_FLAGS="-DPTT_DEV" python -m
// ptt_amd.build --force) reads the PTT_* environment variables (per call, so sweep scripts can flip them between
// launches) and keeps the stamp hooks; scripts/kernel_bench.py sweeps, scripts/sa_phases.py and
// scripts/pair_phases.py need such a build.
struct DevSwitches {
    int linear_rt = 1, linear_ct = 1;    // PTT_LINEAR_TILE="11|12|21|22"
    int linear_small = 1;                // PTT_LINEAR_SMALL=0: short launches (<= 4096 rows) on linear_kernel too
    int sa_gather1 = 0;                  // PTT_SA_GATHER1: one row per gather instruction
    int sa_stagger = 0;                  // PTT_SA_STAGGER (quanta of ~8k cycles the second co-resident workgroup starts late)
    int sa_wave = 1;                     // PTT_SA_WAVE=0: column-split kernel for small-weight levels
    int sa_rt = 2;                       // PTT_SA_RT=1: 32-row workgroups
    int sa_lds = 1;                      // PTT_SA_LDS=0: SA0 on sa_wave_kernel instead of sa_lds_kernel
    int sa_stream = 1;                   // PTT_SA_STREAM=0: SA1 / SA2 on sa_fused_kernel instead of sa_stream_kernel
    int sa_chunk = 2;                    // PTT_SA_CHUNK=n: at most n tiles per sa_stream workgroup (0: 512 workgroups).
                                         // Measured (profiles/r02d): 2, 3, 4, 6 and 12 tiles per workgroup run the
                                         // kernel equally fast; short chunks leave CU slots for the concurrent FPS /
                                         // template-branch kernels of the graphed step (3.26 vs 3.29 ms per step)
    int pair_stagger = 0;                // PTT_PAIR_STAGGER (round 1: 8 was +3 %; with the round-2 kernels 0 is 0.9 % faster, alone and in the step)
    int pair_lds_pad = 0;                // PTT_PAIR_LDS_PAD: extra LDS bytes (forces one workgroup per CU)
    int sa_lds_chunk = 2;                // PTT_SA_LDS_CHUNK=n: at most n centres per wave of sa_lds_kernel (0: one pass)
    int fps_plain = 0;                   // PTT_FPS_PLAIN=1: coordinates carried through the selects for every cloud size
    int ball_cpw = 0;                    // PTT_BALL_CPW=1|4: centres per wave of the ball-query kernels (0: by launch size)
    int fps_t = 0;                       // PTT_FPS_T: FPS threads per cloud at N <= 2048
    int group_grad_global = 0;           // PTT_GROUP_GRAD_GLOBAL: global atomics in ptt_group_grad_f32
    long long* stamps = nullptr;         // PTT_DEBUG_STAMPS=<hex device pointer> (PTT_DEV builds only)
};
const DevSwitches& dev_switches();

// hipFuncAttributeMaxDynamicSharedMemorySize, raised at most once per (kernel, device, size): the launch paths
// call this on every launch and it is a table lookup after the first.
int set_lds_limit(const void* fn, int bytes);

// wgrad_stream.hip: the weight gradient of narrow layers (Cin = 64, Cout = 64 / 128) over many rows, partial[chunk][Cout][Cin]
bool wgrad_stream_ok(int R, int Cout, int Cin, int ldz, int ldx);
int wgrad_stream_rows(int R);
int launch_wgrad_stream(const float* dZ, int ldz, const float* X, int ldx, int R, int Cout, int Cin, int rows, float* partial,
                        const float* x_scale, const float* x_shift, hipStream_t s);

// ---- wave64 cross-lane reductions on DPP (no LDS traffic) -------------------------
// After 4 row-local butterfly steps every lane of a 16-lane row holds the row result;
// row_bcast:15 / row_bcast:31 then fold the four rows into lane 63.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i32(int v) {
    return __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xF, false);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, dpp_i32<CTRL, ROW_MASK>(__builtin_bit_cast(int, v)));
}

__device__ __forceinline__ float wave_max_f32(float v) {
    v = fmaxf(v, dpp_f32<0xB1, 0xF>(v));   // quad_perm [1,0,3,2]
    v = fmaxf(v, dpp_f32<0x4E, 0xF>(v));   // quad_perm [2,3,0,1]
    v = fmaxf(v, dpp_f32<0x141, 0xF>(v));  // row_half_mirror
    v = fmaxf(v, dpp_f32<0x140, 0xF>(v));  // row_mirror
    v = fmaxf(v, dpp_f32<0x142, 0xA>(v));  // row_bcast:15 into rows 1,3
    v = fmaxf(v, dpp_f32<0x143, 0xC>(v));  // row_bcast:31 into rows 2,3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// Same reduction with the DPP modifier fused into v_max_f32 (one VALU op per step instead of
// copy + v_mov_dpp + canonicalise + max). Inputs must not be NaN. "s_nop 1" covers the
// VALU-write -> DPP-read hazard of the in-place chain.
__device__ __forceinline__ float wave_max_f32_fused(float v) {
    asm volatile(
        "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

__device__ __forceinline__ float wave_min_f32(float v) {
    v = fminf(v, dpp_f32<0xB1, 0xF>(v));
    v = fminf(v, dpp_f32<0x4E, 0xF>(v));
    v = fminf(v, dpp_f32<0x141, 0xF>(v));
    v = fminf(v, dpp_f32<0x140, 0xF>(v));
    v = fminf(v, dpp_f32<0x142, 0xA>(v));
    v = fminf(v, dpp_f32<0x143, 0xC>(v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ int wave_min_i32(int v) {
    v = min(v, dpp_i32<0xB1, 0xF>(v));
    v = min(v, dpp_i32<0x4E, 0xF>(v));
    v = min(v, dpp_i32<0x141, 0xF>(v));
    v = min(v, dpp_i32<0x140, 0xF>(v));
    v = min(v, dpp_i32<0x142, 0xA>(v));
    v = min(v, dpp_i32<0x143, 0xC>(v));
    return __builtin_amdgcn_readlane(v, 63);
}

}  // namespace ptt
