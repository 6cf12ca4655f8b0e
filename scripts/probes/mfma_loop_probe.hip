// Dev probe: what limits the fp32-MFMA K-loop? Variants: MFMA only / + global B loads / + LDS A reads.
// hipcc --offload-arch=gfx950 -O3 scripts/probes/mfma_loop_probe.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int DIST, int ADDR = 0>
__global__ __launch_bounds__(256, 2) void probe(const f32x4* __restrict__ Wp, float* out, int nkb, int reps) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 32 * 516; i += 256) smem[i] = (float)(i & 7);
    __syncthreads();
    f32x16 acc[4];
    for (int u = 0; u < 4; ++u) for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
    const float* arow = smem + (lane & 31) * 516 + 4 * (lane >> 5);
    const f32x4* bp = Wp + (size_t)w * 64 + lane + (ADDR == 2 ? (size_t)(blockIdx.x % 48) * 64 * 16 * 64 : 0);
    const size_t bstep = (ADDR == 1) ? 0 : 16 * 64;
    for (int rep = 0; rep < reps; ++rep) {
        f32x4 a[3], b[3][4];
#pragma unroll
        for (int s = 0; s < 3; ++s) { a[s] = f32x4{1.f, 2.f, 3.f, 4.f}; for (int u = 0; u < 4; ++u) b[s][u] = f32x4{1.f, 1.f, 1.f, 1.f}; }
        // prologue
#pragma unroll
        for (int d = 0; d < DIST; ++d) {
            if (MODE >= 1) { for (int u = 0; u < 4; ++u) b[d][u] = bp[(size_t)d * bstep + (size_t)u * 256]; }
            if (MODE >= 2) a[d] = *reinterpret_cast<const f32x4*>(arow + d * 8);
        }
        for (int kb = 0; kb < nkb; kb += 3) {
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int nx = kb + s + DIST;          // block to prefetch
                const int slot = (s + DIST) % 3;
                if (nx < nkb) {
                    if (MODE >= 1) { for (int u = 0; u < 4; ++u) b[slot][u] = bp[(size_t)nx * bstep + (size_t)u * 256]; }
                    if (MODE >= 2) a[slot] = *reinterpret_cast<const f32x4*>(arow + nx * 8);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (kb + s < nkb) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s][j], b[s][u][j], acc[u], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float s = 0.f;
    for (int u = 0; u < 4; ++u) for (int r = 0; r < 16; ++r) s += acc[u][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE, int DIST, int ADDR = 0>
void run(const char* name, int grid, const f32x4* W, float* out) {
    const int nkb = 63, reps = 40, lds = 32 * 516 * 4;
    hipFuncSetAttribute((const void*)probe<MODE, DIST, ADDR>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<MODE, DIST, ADDR><<<grid, 256, lds>>>(W, out, nkb, 2);
    hipEventRecord(e0);
    probe<MODE, DIST, ADDR><<<grid, 256, lds>>>(W, out, nkb, reps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double mf = (double)grid * 4 * reps * nkb * 16;          // MFMA instructions
    double tf = mf * 2.0 * 32 * 32 * 2 / (ms * 1e-3) / 1e12;
    printf("%-28s grid=%4d  %8.3f ms  %7.1f TFLOP/s  (%.1f cycles/MFMA/SIMD @2.4GHz, %d waves/SIMD)\n", name, grid, ms, tf,
           ms * 1e-3 * 2.4e9 / (mf / (grid < 512 ? grid * 4 : 1024)), grid / 256);
}

int main() {
    f32x4* W; float* out;
    hipMalloc(&W, 48ull * 64 * 16 * 64 * sizeof(f32x4)); hipMemset(W, 0, 48ull * 64 * 16 * 64 * sizeof(f32x4));
    hipMalloc(&out, 4096 * 256 * 4);
    for (int grid : {256, 512}) {
        run<0, 1>("mfma only", grid, W, out);
        run<1, 1>("+B global, dist 1", grid, W, out);
        run<1, 2>("+B global, dist 2", grid, W, out);
        run<2, 1>("+B global +A lds, dist 1", grid, W, out);
        run<2, 2>("+B global +A lds, dist 2", grid, W, out);
        run<1, 1, 1>("+B same addr (L1 hit) d1", grid, W, out);
        run<1, 1, 2>("+B private W copies d1", grid, W, out);
        run<1, 2, 2>("+B private W copies d2", grid, W, out);
    }
    return 0;
}
