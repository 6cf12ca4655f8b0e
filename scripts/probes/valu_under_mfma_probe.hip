// Dev probe: how fast does a wave issue VALU / LDS / readlane work while ANOTHER wave on the same SIMD streams
// fp32 MFMAs? One workgroup of 512 threads per CU: waves 0-3 (one per SIMD) run the test loop, waves 4-7 either
// idle or run a dense v_mfma_f32_32x32x2_f32 loop. Cycles per instruction of the test loop are reported.
// hipcc --offload-arch=gfx950 -O3 scripts/probes/valu_under_mfma_probe.hip -o /tmp/vprobe && /tmp/vprobe
#include <hip/hip_runtime.h>
#include <stdio.h>
#pragma clang diagnostic ignored "-Wunused-value"
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int TEST, int MFMA_ON, int PRIO>
__global__ __launch_bounds__(512, 1) void probe(float* out, long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __shared__ volatile int done;
    if (threadIdx.x == 0) done = 0;
    __syncthreads();
    if (w >= 4) {                                   // partner waves
        if (!MFMA_ON) return;
        f32x16 acc[4];
        for (int u = 0; u < 4; ++u) for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
        float a = (float)lane, b = 1.f;
        while (done < 4) {
#pragma unroll
            for (int j = 0; j < 16; ++j)
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u], 0, 0, 0);
        }
        float s = 0.f;
        for (int u = 0; u < 4; ++u) for (int r = 0; r < 16; ++r) s += acc[u][r];
        out[blockIdx.x * 512 + threadIdx.x] = s;
        return;
    }
    if (PRIO) __builtin_amdgcn_s_setprio(3);
    for (volatile int spin = 0; spin < 2000; ++spin) {}        // let the partner get going
    float x0 = lane, x1 = lane + 1, x2 = lane + 2, x3 = lane + 3, x4 = 0.5f, x5 = 0.25f, x6 = 1.5f, x7 = 2.5f;
    int n = 0;
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (TEST == 0) {            // 8 independent FMAs
            x0 = fmaf(x0, 1.0001f, 0.5f); x1 = fmaf(x1, 1.0001f, 0.5f); x2 = fmaf(x2, 1.0001f, 0.5f); x3 = fmaf(x3, 1.0001f, 0.5f);
            x4 = fmaf(x4, 1.0001f, 0.5f); x5 = fmaf(x5, 1.0001f, 0.5f); x6 = fmaf(x6, 1.0001f, 0.5f); x7 = fmaf(x7, 1.0001f, 0.5f);
            n += 8;
        } else if (TEST == 1) {     // 4 ds_write_b128 + 4 FMAs
            f32x4 v = {x0, x1, x2, x3};
            *reinterpret_cast<f32x4*>(&lds[(w * 64 + lane) * 4 + 0]) = v;
            *reinterpret_cast<f32x4*>(&lds[(w * 64 + lane) * 4 + 1024]) = v;
            *reinterpret_cast<f32x4*>(&lds[(w * 64 + lane) * 4 + 2048]) = v;
            *reinterpret_cast<f32x4*>(&lds[(w * 64 + lane) * 4 + 3072]) = v;
            x0 = fmaf(x0, 1.0001f, 0.5f); x1 = fmaf(x1, 1.0001f, 0.5f); x2 = fmaf(x2, 1.0001f, 0.5f); x3 = fmaf(x3, 1.0001f, 0.5f);
            asm volatile("" ::: "memory");
            n += 8;
        } else {                    // 4 readlane + 4 dependent FMAs (the gather's pattern)
            const int li = i & 15;
            float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x4), li));
            float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x5), li));
            float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x6), li));
            float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x7), li));
            x0 = fmaf(x0, r0, 0.5f); x1 = fmaf(x1, r1, 0.5f); x2 = fmaf(x2, r2, 0.5f); x3 = fmaf(x3, r3, 0.5f);
            n += 8;
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    if (lane == 0) cyc[blockIdx.x * 4 + w] = t1 - t0;
    out[blockIdx.x * 512 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + n;
    if (lane == 0) atomicAdd((int*)&done, 1);      // the partner waves stop when all four test waves are through
}

template <int TEST, int MFMA_ON, int PRIO>
void run(const char* name, float* out, long long* cyc) {
    const int grid = 256, iters = 4000;
    hipMemset(cyc, 0, grid * 4 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    probe<TEST, MFMA_ON, PRIO><<<grid, 512>>>(out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    static long long h[1024]; hipMemcpy(h, cyc, grid * 4 * 8, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < grid * 4; ++i) s += h[i];
    printf("%-44s %7.2f counter ticks / instruction   (kernel %.3f ms)\n", name, s / (grid * 4) / (iters * 8.0), ms);
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 1024 * 8);
    run<0, 0, 0>("FMA x8, partner idle", out, cyc);
    run<0, 1, 0>("FMA x8, partner MFMA", out, cyc);
    run<0, 1, 1>("FMA x8, partner MFMA, setprio 3", out, cyc);
    run<1, 0, 0>("ds_write_b128 x4 + FMA x4, partner idle", out, cyc);
    run<1, 1, 0>("ds_write_b128 x4 + FMA x4, partner MFMA", out, cyc);
    run<1, 1, 1>("ds_write_b128 x4 + FMA x4, MFMA, setprio 3", out, cyc);
    run<2, 0, 0>("readlane x4 + FMA x4, partner idle", out, cyc);
    run<2, 1, 0>("readlane x4 + FMA x4, partner MFMA", out, cyc);
    run<2, 1, 1>("readlane x4 + FMA x4, MFMA, setprio 3", out, cyc);
    return 0;
}
