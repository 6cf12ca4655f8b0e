// Round-4 probe (VERDICT r03 "next" #5): is exact-fp32-class GEMM by three-way bf16 splits worth a kernel family?
//
//   C[M,N] = A[M,K] . W[N,K]^T   (fp32 in, fp32 out), every fp32 operand written as a0 + a1 + a2 with bf16 pieces
//   (8 + 8 + 8 mantissa bits), the product as the six terms a0b0 + a0b1 + a1b0 + a0b2 + a1b1 + a2b0 on
//   v_mfma_f32_32x32x16_bf16 (32 cycles per 32x32x16: six per 16 channels = 192 cycles against 512 for eight
//   v_mfma_f32_32x32x2_f32), fp32 accumulation. NP = 2: two pieces, three products (~16 mantissa bits), for scale.
//
// Kernel shape (the one DESIGN.md §7 said such a path would need): 256 x 256 output tile per workgroup of 8 waves
// (wave = 128 x 64 = 4 x 2 accumulator tiles), BOTH operands staged through LDS per 16-channel step — split into pieces on
// the way in (v_cvt_pk_bf16_f32 + exact fp32 residuals) — double-buffered, one barrier per step.
//
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/bf16x3_gemm_probe.hip -o /tmp/bf16x3 && /tmp/bf16x3
// prints TFLOP/s (algorithmic 2.M.N.K) and the error of sampled outputs against float64, beside the error of an fp32
// fma-chain evaluation of the same outputs (what v_mfma_f32_32x32x2_f32 computes). The exact-fp32 kernels' rates on the
// same shapes come from scripts/rows_gemm_bench.py in the same session.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int NP>
__device__ __forceinline__ void split8(const f32x4& lo, const f32x4& hi, bf16x8 (&p)[NP]) {
    float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float r = v[i];
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const __bf16 h = (__bf16)r;          // round to nearest even
            p[q][i] = h;
            r -= (float)h;                       // exact: the residual has at most 16 significant bits
        }
    }
}

// LDS rows are 32 bytes (16 bf16): byte offset of the 16-byte half h of row r, with the halves of rows 4-7 (mod 8) swapped — a
// plain [row][16] layout puts rows r and r + 4 on the same banks and the ds_read_b128 fragment reads run at half rate
__device__ __forceinline__ int swz(int row, int h) { return row * 32 + ((h ^ ((row >> 2) & 1)) << 4); }

// Weights split once per launch set (they are constants of a step): Wp[piece][n][k] bf16
template <int NP>
__global__ void presplit_kernel(const float* __restrict__ W, __bf16* __restrict__ Wp, size_t n) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        float r = W[e];
#pragma unroll
        for (int q = 0; q < NP; ++q) { const __bf16 h = (__bf16)r; Wp[(size_t)q * n + e] = h; r -= (float)h; }
    }
}

// LDS per buffer: A pieces [NP][256 rows][16 k] bf16, then W pieces [NP][TN rows][16 k] bf16.
// WN = waves along N: 4 -> a 256 x 256 tile, 8 waves, one workgroup per CU (all waves of a CU meet at the same barrier: their
// LDS / split / MFMA phases coincide); 2 -> a 256 x 128 tile, 4 waves, TWO workgroups per CU whose phases drift apart, so one's
// MFMAs run beside the other's staging. PRE: W arrives pre-split (presplit_kernel), its staging is a copy.
// ABL (timing-only ablations, results wrong): 1 = no global loads / splits / LDS writes inside the K loop; 2 = no LDS fragment
// reads either (the K loop is MFMAs and the barrier).
template <int NP, int WN, bool PRE, int ABL = 0>
__global__ __launch_bounds__(128 * WN, WN == 4 ? 1 : 2) void gemm_bf16_split(const float* __restrict__ A, const float* __restrict__ W,
                                                                               const __bf16* __restrict__ Wp, float* __restrict__ C,
                                                                               int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int T = 128 * WN, TN = 64 * WN;
    constexpr int APIECE = 256 * 16 * 2, WPIECE = TN * 16 * 2;
    constexpr int BUF = NP * (APIECE + WPIECE);
    constexpr int ASL = 512 / T, WSL = (2 * TN + T - 1) / T;          // (row, channel half) slots per thread
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wm = w / WN, wn = w % WN;                  // wave grid 2 (rows) x WN (columns): 128 x 64 per wave
    const int nbn = N / TN;
    const int bm = blockIdx.x / nbn, bn = blockIdx.x % nbn;
    const int row0 = bm * 256, col0 = bn * TN;
    f32x4 ra[ASL][2], rw[WSL][2];
    bf16x8 rwp[WSL][NP];
    auto fetch = [&](int k0) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < ASL; ++i) {
            const int sl = t + i * T, sr = sl >> 1, sk = (sl & 1) * 8;
            const float* ag = A + (size_t)(row0 + sr) * K + sk + k0;
            const bool ok = row0 + sr < M;
            ra[i][0] = ok ? *reinterpret_cast<const f32x4*>(ag) : z;
            ra[i][1] = ok ? *reinterpret_cast<const f32x4*>(ag + 4) : z;
        }
#pragma unroll
        for (int i = 0; i < WSL; ++i) {
            const int sl = t + i * T, sr = sl >> 1, sk = (sl & 1) * 8;
            if (sl < 2 * TN) {
                if constexpr (PRE) {
#pragma unroll
                    for (int q = 0; q < NP; ++q)
                        rwp[i][q] = *reinterpret_cast<const bf16x8*>(Wp + (size_t)q * N * K + (size_t)(col0 + sr) * K + sk + k0);
                } else {
                    const float* wg = W + (size_t)(col0 + sr) * K + sk + k0;
                    rw[i][0] = *reinterpret_cast<const f32x4*>(wg);
                    rw[i][1] = *reinterpret_cast<const f32x4*>(wg + 4);
                }
            }
        }
    };
    auto stage = [&](int buf) {
        unsigned char* base = smem + buf * BUF;
#pragma unroll
        for (int i = 0; i < ASL; ++i) {
            const int sl = t + i * T, sr = sl >> 1, sk = (sl & 1) * 8;
            bf16x8 pa[NP];
            split8<NP>(ra[i][0], ra[i][1], pa);
#pragma unroll
            for (int q = 0; q < NP; ++q) *reinterpret_cast<bf16x8*>(base + q * APIECE + swz(sr, sl & 1)) = pa[q];
        }
#pragma unroll
        for (int i = 0; i < WSL; ++i) {
            const int sl = t + i * T, sr = sl >> 1, sk = (sl & 1) * 8;
            if (sl < 2 * TN) {
                bf16x8 pw[NP];
                if constexpr (PRE) {
#pragma unroll
                    for (int q = 0; q < NP; ++q) pw[q] = rwp[i][q];
                } else split8<NP>(rw[i][0], rw[i][1], pw);
#pragma unroll
                for (int q = 0; q < NP; ++q) *reinterpret_cast<bf16x8*>(base + NP * APIECE + q * WPIECE + swz(sr, sl & 1)) = pw[q];
            }
        }
    };
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    fetch(0);
    stage(0);
    __syncthreads();
    const int steps = K / 16;
    // fragment addresses: lane reads 8 bf16 = 16 B of row (tile row0 + lane % 32), channel group lane / 32
    const int foff = swz(lane & 31, lane >> 5);          // tile rows start at multiples of 32: the swizzle bit is the lane's
    bf16x8 fa[4][NP], fw[2][NP];
    if (ABL == 2) {
#pragma unroll
        for (int q = 0; q < NP; ++q) {
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i][q] = *reinterpret_cast<const bf16x8*>(smem + q * APIECE + (wm * 128 + i * 32) * 32 + foff);
#pragma unroll
            for (int j = 0; j < 2; ++j) fw[j][q] = *reinterpret_cast<const bf16x8*>(smem + NP * APIECE + q * WPIECE + (wn * 64 + j * 32) * 32 + foff);
        }
    }
    for (int s = 0; s < steps; ++s) {
        if (ABL == 0 && s + 1 < steps) fetch((s + 1) * 16);
        const unsigned char* base = smem + (ABL ? 0 : (s & 1)) * BUF;
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            if (ABL == 2) break;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                fa[i][q] = *reinterpret_cast<const bf16x8*>(base + q * APIECE + (wm * 128 + i * 32) * 32 + foff);
#pragma unroll
            for (int j = 0; j < 2; ++j)
                fw[j][q] = *reinterpret_cast<const bf16x8*>(base + NP * APIECE + q * WPIECE + (wn * 64 + j * 32) * 32 + foff);
        }
        // the small terms first
#pragma unroll
        for (int o = NP - 1; o >= 0; --o)               // the kept terms are those with qa + qb <= NP - 1
#pragma unroll
            for (int qa = 0; qa < NP; ++qa) {
                const int qb = o - qa;
                if (qb < 0 || qb >= NP) continue;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][qa], fw[j][qb], acc[i][j], 0, 0, 0);
            }
        if (ABL == 0 && s + 1 < steps) stage((s + 1) & 1);
        __syncthreads();
    }
    const int half = lane >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = col0 + wn * 64 + j * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (row < M) C[(size_t)row * N + col] = acc[i][j][r];
            }
        }
}

template <int NP, int WN, bool PRE, int ABL = 0>
static double run(const float* dA, const float* dW, const __bf16* dWp, float* dC, int M, int N, int K, int reps) {
    constexpr int TN = 64 * WN;
    const int lds = 2 * NP * (256 + TN) * 16 * 2;
    auto kern = gemm_bf16_split<NP, WN, PRE, ABL>;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const dim3 grid(((M + 255) / 256) * (N / TN));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, grid, dim3(128 * WN), lds, 0, dA, dW, dWp, dC, M, N, K);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, grid, dim3(128 * WN), lds, 0, dA, dW, dWp, dC, M, N, K);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipGetLastError());
    return ms / reps;
}

static void one_shape(int M, int N, int K) {
    std::vector<float> hA((size_t)M * K), hW((size_t)N * K);
    unsigned s = 12345u + M + K;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffffff) / 8388608.0f - 1.0f; };   // 24 random mantissa bits
    for (auto& v : hA) v = rnd() * 1.7f;
    for (auto& v : hW) v = rnd() / sqrtf((float)K) * 1.3f;
    float *dA, *dW, *dC;
    CHECK(hipMalloc(&dA, hA.size() * 4)); CHECK(hipMalloc(&dW, hW.size() * 4)); CHECK(hipMalloc(&dC, (size_t)M * N * 4));
    CHECK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dW, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
    const double flop = 2.0 * M * N * K;
    __bf16* dWp;
    CHECK(hipMalloc(&dWp, (size_t)3 * N * K * 2));
    std::vector<float> hC((size_t)64 * N);
    struct Var { const char* name; int np; int wn; bool pre; };
    const Var vars[] = {{"x3  256x256 tile, 1 wg/CU, W split in the kernel", 3, 4, false}, {"x3  256x256 tile, 1 wg/CU, W pre-split        ", 3, 4, true},
                        {"x3  256x128 tile, 2 wg/CU, W split in the kernel", 3, 2, false}, {"x3  256x128 tile, 2 wg/CU, W pre-split        ", 3, 2, true},
                        {"x2  256x128 tile, 2 wg/CU, W pre-split        ", 2, 2, true}};
    for (const Var& v : vars) {
        if (v.np == 3) hipLaunchKernelGGL(presplit_kernel<3>, dim3(1024), dim3(256), 0, 0, dW, dWp, (size_t)N * K);
        else hipLaunchKernelGGL(presplit_kernel<2>, dim3(1024), dim3(256), 0, 0, dW, dWp, (size_t)N * K);
        double ms;
        if (v.np == 3 && v.wn == 4 && !v.pre) ms = run<3, 4, false>(dA, dW, dWp, dC, M, N, K, 20);
        else if (v.np == 3 && v.wn == 4) ms = run<3, 4, true>(dA, dW, dWp, dC, M, N, K, 20);
        else if (v.np == 3 && !v.pre) ms = run<3, 2, false>(dA, dW, dWp, dC, M, N, K, 20);
        else if (v.np == 3) ms = run<3, 2, true>(dA, dW, dWp, dC, M, N, K, 20);
        else ms = run<2, 2, true>(dA, dW, dWp, dC, M, N, K, 20);
        // error of the first 64 rows against float64, and of an fp32 fma chain in k order (= the exact-fp32 MFMA path)
        CHECK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
        double emax = 0, e32max = 0, scale = 0;
        for (int r = 0; r < 64; ++r)
            for (int c = 0; c < N; c += 7) {
                double ref = 0; float f = 0.f;
                for (int k = 0; k < K; ++k) { ref += (double)hA[(size_t)r * K + k] * hW[(size_t)c * K + k]; f = fmaf(hA[(size_t)r * K + k], hW[(size_t)c * K + k], f); }
                emax = fmax(emax, fabs(hC[(size_t)r * N + c] - ref));
                e32max = fmax(e32max, fabs((double)f - ref));
                scale = fmax(scale, fabs(ref));
            }
        printf("%7d x %4d x %4d  bf16 %s: %8.3f ms  %7.1f TFLOP/s   max |err| vs float64 %.3e (max |value| %.2f) ; an fp32 fma chain: %.3e\n",
               M, K, N, v.name, ms, flop / ms / 1e9, emax, scale, e32max);
    }
    printf("%7d x %4d x %4d  timing-only ablations of x3 256x256: no staging in the K loop %.3f ms (%.0f TFLOP/s), MFMAs + barrier only %.3f ms (%.0f TFLOP/s)\n",
           M, K, N, run<3, 4, true, 1>(dA, dW, dWp, dC, M, N, K, 20), flop / run<3, 4, true, 1>(dA, dW, dWp, dC, M, N, K, 20) / 1e9,
           run<3, 4, true, 2>(dA, dW, dWp, dC, M, N, K, 20), flop / run<3, 4, true, 2>(dA, dW, dWp, dC, M, N, K, 20) / 1e9);
    CHECK(hipFree(dWp));
    CHECK(hipFree(dA)); CHECK(hipFree(dW)); CHECK(hipFree(dC));
}

int main() {
    one_shape(98304, 512, 512);      // the transformers' 512 x 512 layers over the (point, neighbour) rows of a 48-frame step
    one_shape(393216, 256, 256);     // CosineSimAug / SA layers
    one_shape(49152, 512, 512);
    return 0;
}
