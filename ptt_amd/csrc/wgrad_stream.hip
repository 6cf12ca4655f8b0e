// N3: the weight gradient of the NARROW layers of the training step (see the kernel's comment). Its own translation unit because
// it is compiled with -mllvm -amdgpu-mfma-vgpr-form: with the accumulators in AGPRs the register allocator copies all of them to
// vector registers for the final cross-wave sum and the kernel runs at one wave per SIMD (150 + 128 registers); kept in
// vector registers throughout it is 190 registers — two waves per SIMD, which a streaming kernel needs to cover HBM latency.
#include "common.h"

namespace ptt {

typedef float f32x16t __attribute__((ext_vector_type(16)));

// Weight gradient of a NARROW layer over many rows (SA0's 64 -> 64 and 64 -> 128 convolutions over 786k / 393k rows): 4 - 8
// flops per byte read, a streaming pass over dZ and X. No LDS staging: a lane's MFMA operand IS one float of a row (channel =
// lane & 31 of the row pair member lane >> 5), so every load instruction of a wave reads two whole 128-byte lines and feeds
// the matrix unit directly; a wave keeps two blocks of J row pairs of both operands in registers (one being multiplied, one in
// flight), takes the blocks w, w + 4, ... of its workgroup's chunk and holds the whole (32 CO) x (32 CI) gradient in accumulators; the four waves are added through LDS
// in a fixed order at the end. partial[chunk][Cout][Cin] as linear_wgrad_kernel writes it. (linear_wgrad_kernel on these shapes:
// one barrier per 32 rows with at most a quarter of its staged tile used — 2.3 - 2.8x the read time.)
template <int CO, int CI, int J>
__global__ __launch_bounds__(256) void wgrad_stream_kernel(const float* __restrict__ dZ, int ldz, const float* __restrict__ X, int ldx,
                                                           int R, int rows_per_wg, float* __restrict__ partial,
                                                           const float* __restrict__ xa, const float* __restrict__ xb) {
    extern __shared__ __attribute__((aligned(16))) float ws_red[];          // 2 x (CO * CI * 16 * 64) floats
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, half = lane >> 5, col = lane & 31;
    const int r_begin = blockIdx.x * rows_per_wg, r_end = min(R, r_begin + rows_per_wg);
    f32x16t acc[CO][CI];
#pragma unroll
    for (int a = 0; a < CO; ++a)
#pragma unroll
        for (int b = 0; b < CI; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float as[CI], bs[CI];
#pragma unroll
    for (int b = 0; b < CI; ++b) { as[b] = xa ? xa[b * 32 + col] : 1.f; bs[b] = xa ? xb[b * 32 + col] : 0.f; }
    const bool act = xa != nullptr;
    // J row pairs of both operands per block, two blocks in registers: the loads of the next block are in flight while the matrix
    // unit works through the current one. Raw buffer loads: per-lane offset (row-pair member, channel) fixed, the block's row
    // offset in a scalar register, the 32-channel group in the instruction — no per-load address registers; rows >= R read 0
    // (buffer range check), rows of the next chunk are never touched (chunks are multiples of the block step).
    const __amdgpu_buffer_rsrc_t zres = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dZ), 0, (int)((size_t)R * ldz * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X), 0, (int)((size_t)R * ldx * 4), 0x00020000);
    const int zoff = (half * ldz + col) * 4, xoff = (half * ldx + col) * 4;
    float z0[J][CO], x0[J][CI], z1[J][CO], x1[J][CI];
    auto fetch = [&](float (&zv)[J][CO], float (&xv)[J][CI], int r0) {
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int zs = (r0 + 2 * j) * ldz * 4, xs = (r0 + 2 * j) * ldx * 4;           // wave-uniform
#pragma unroll
            for (int a = 0; a < CO; ++a) zv[j][a] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(zres, zoff + a * 128, zs, 0));
#pragma unroll
            for (int b = 0; b < CI; ++b) xv[j][b] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xres, xoff + b * 128, xs, 0));
        }
    };
    auto multiply = [&](float (&zv)[J][CO], float (&xv)[J][CI]) {
#pragma unroll
        for (int j = 0; j < J; ++j) {
            if (act) {
#pragma unroll
                for (int b = 0; b < CI; ++b) xv[j][b] = fmaxf(__builtin_fmaf(xv[j][b], as[b], bs[b]), 0.f);   // rows >= R: dZ is 0
            }
#pragma unroll
            for (int a = 0; a < CO; ++a)
#pragma unroll
                for (int b = 0; b < CI; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(zv[j][a], xv[j][b], acc[a][b], 0, 0, 0);
        }
    };
    constexpr int STEP = 8 * J;                         // 4 waves x 2J rows
    int r0 = __builtin_amdgcn_readfirstlane(r_begin + w * 2 * J);
    fetch(z1, x1, r0);
    for (; r0 < r_end; r0 += STEP) {                    // one multiply site: the accumulators never move
#pragma unroll
        for (int j = 0; j < J; ++j) {
#pragma unroll
            for (int a = 0; a < CO; ++a) z0[j][a] = z1[j][a];
#pragma unroll
            for (int b = 0; b < CI; ++b) x0[j][b] = x1[j][b];
        }
        if (r0 + STEP < r_end) fetch(z1, x1, r0 + STEP);
        multiply(z0, x0);
    }
    // (w0 + w1) + (w2 + w3) through LDS. The accumulators are only ever READ after the loop (stored, or added to a stored value and
    // stored again): accumulators that are also written here get copied to vector registers wholesale by the register allocator
    // and the kernel's occupancy would be that of its last 1 %.
    constexpr int TILE = CO * CI * 16 * 64;
    float* mine = ws_red + (w >> 1) * TILE;
#pragma unroll 1
    for (int p = 0; p < 2; ++p) {                       // p = 0: the odd waves store; p = 1: the even waves add theirs. ONE use site
        if ((w & 1) != p) {
#pragma unroll
            for (int a = 0; a < CO; ++a)
#pragma unroll
                for (int b = 0; b < CI; ++b) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float* slot = mine + ((a * CI + b) * 16 + r) * 64 + lane;
                        float v = acc[a][b][r];
                        if (p) v += *slot;
                        *slot = v;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
        __syncthreads();
    }
    constexpr int Cin = CI * 32;
    float* P = partial + (size_t)blockIdx.x * (CO * 32) * Cin;
    for (int e = threadIdx.x; e < TILE; e += 256) {
        const int r = (e >> 6) & 15, tile = e >> 10, a = tile / CI, b = tile - a * CI;
        P[(size_t)(a * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * Cin + b * 32 + col] = ws_red[e] + ws_red[TILE + e];
    }
}

bool wgrad_stream_ok(int R, int Cout, int Cin, int ldz, int ldx) {
    return Cin == 64 && (Cout == 64 || Cout == 128) && R >= 65536 && (long long)R * ldz * 4 < 0x7fffffffLL && (long long)R * ldx * 4 < 0x7fffffffLL;
}
// row chunking: 1024 workgroups (four per CU), chunks a multiple of 64 rows
int wgrad_stream_rows(int R) { return ((R + 1023) / 1024 + 63) / 64 * 64; }

int launch_wgrad_stream(const float* dZ, int ldz, const float* X, int ldx, int R, int Cout, int Cin, int rows, float* partial,
                        const float* x_scale, const float* x_shift, hipStream_t s) {
    const int nch = (R + rows - 1) / rows;
    const int lds = 2 * Cout * Cin * (int)sizeof(float);
    if (Cout == 64) {
        if (int rc = set_lds_limit(reinterpret_cast<const void*>(wgrad_stream_kernel<2, 2, 8>), lds)) return rc;
        hipLaunchKernelGGL((wgrad_stream_kernel<2, 2, 8>), dim3(nch), dim3(256), lds, s, dZ, ldz, X, ldx, R, rows, partial, x_scale, x_shift);
    } else {
        if (int rc = set_lds_limit(reinterpret_cast<const void*>(wgrad_stream_kernel<4, 2, 4>), lds)) return rc;
        hipLaunchKernelGGL((wgrad_stream_kernel<4, 2, 4>), dim3(nch), dim3(256), lds, s, dZ, ldz, X, ldx, R, rows, partial, x_scale, x_shift);
    }
    return PTT_OK;
}

}  // namespace ptt
