// Does the range check of a raw buffer load (stride 0) on gfx950 see the SGPR offset? Prints, for a 1024-byte buffer of
// ones followed by twos, what loads at (voffset, soffset) return: 0 = clipped by the descriptor, 2 = read past it.
//   hipcc --offload-arch=gfx950 -O2 scripts/probes/buffer_range_probe.hip -o /tmp/brp && /tmp/brp
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void probe(const float* base, float* out) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 1024, 0x00020000);
    const int voffs[4] = {0, 1020, 1024, 512}, soffs[4] = {0, 0, 0, 768};
    for (int k = 0; k < 4; ++k)
        out[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voffs[k], soffs[k], 0));
    out[4] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, 1000, 256, 0));   // v in range, v + s beyond
    out[5] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, 2000, 0, 0));     // v beyond
}
int main() {
    float *d, *o, h[512], res[8];
    for (int i = 0; i < 512; ++i) h[i] = i < 256 ? 1.f : 2.f;
    hipMalloc(&d, 2048); hipMalloc(&o, 64);
    hipMemcpy(d, h, 2048, hipMemcpyHostToDevice);
    probe<<<1, 1>>>(d, o);
    hipMemcpy(res, o, 32, hipMemcpyDeviceToHost);
    printf("(v0,s0)=%g (v1020,s0)=%g (v1024,s0)=%g (v512,s768)=%g (v1000,s256)=%g (v2000,s0)=%g\n", res[0], res[1], res[2], res[3], res[4], res[5]);
    return 0;
}
