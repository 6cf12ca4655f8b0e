#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
__device__ __forceinline__ void swap_halves(float v, float& lo, float& hi) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    lo = a; hi = b;
}
__global__ void k(const float* in, float* o) {
    float v = in[threadIdx.x] * 2.0f;
    float lo, hi; swap_halves(v, lo, hi);
    float m = fmaxf(lo, hi);
    float e = __expf(v - m);
    float lo2, hi2; swap_halves(e, lo2, hi2);
    o[threadIdx.x] = e / (lo2 + hi2);
}
int main() {
    float h[64], *d, *o; for (int i = 0; i < 64; ++i) h[i] = sinf(i * 1.7f);
    hipMalloc(&d, 256); hipMalloc(&o, 256); hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, o); float r[64]; hipMemcpy(r, o, 256, hipMemcpyDeviceToHost);
    double worst = 0; for (int i = 0; i < 64; ++i) { int j = i ^ 32; double a = 2 * h[i], b = 2 * h[j]; double m = fmax(a, b); double ref = exp(a - m) / (exp(a - m) + exp(b - m)); worst = fmax(worst, fabs(ref - r[i])); }
    printf("max err %g\n", worst); return 0;
}
