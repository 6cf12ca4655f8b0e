// Point-cloud index ops for gfx950: furthest point sampling, ball query, gather, group,
// kNN. These replace the CUDA-only third-party `pointnet2_ops._ext` the reference calls
// (ptt/models/backbones_3d/pointnet2/pointnet2_utils.py:24,78,112,118,237,257,287) and the
// square_distance+argsort kNN of the transformer (transformer_block/variants.py:150-151).
//
// Arithmetic contract (restated in oracle/ptt_oracle.c): squared distances
// are (dx*dx + dy*dy) + dz*dz in fp32 with NO fused multiply-add, so that indices are
// bit-reproducible against the CPU oracle. This whole file is compiled with
// -ffp-contract=off and carries the pragma below.
#pragma clang fp contract(off)

#include <limits.h>
#include <stdlib.h>
#include "common.h"

namespace ptt {

__device__ __forceinline__ float sqdist3(float ax, float ay, float az, float bx, float by, float bz) {
    const float dx = ax - bx, dy = ay - by, dz = az - bz;
    return (dx * dx + dy * dy) + dz * dz;
}

// kNN candidate distance: a NON-FINITE one (NaN / infinite coordinates: outside the contract, INTEGRATION.md "Non-finite input")
// becomes FLT_MAX — still a selectable candidate, after every finite one, lowest index first — so that a round always finds a
// winner and no index leaves the cloud (a retired candidate is marked NaN: never selectable again). Finite input: unchanged.
__device__ __forceinline__ float knn_dist(float ax, float ay, float az, float bx, float by, float bz) {
    const float d = sqdist3(ax, ay, az, bx, by, bz);
    return d < __builtin_inff() ? d : 3.402823466e+38f;
}

// ------------------------------------------------------------------------------------------
// FPS. One workgroup per cloud, T threads, P points per thread held in registers together
// with their running min-distance; nothing but the chosen index leaves the CU per iteration.
// Thread t owns the CONTIGUOUS points [t*P, t*P+P): a lower lane (and a lower wave) always
// holds lower indices, so the arg-max tie-break "lowest index" is simply the lowest lane whose
// local best equals the wave maximum — one ballot + find-first-set instead of a second
// cross-lane reduction.
// Iteration = register update (branch-free selects) -> fused-DPP wave max -> ballot/ffs ->
// (T > 64) one LDS slot per wave + ONE barrier (slots are double-buffered by iteration parity)
// -> every thread folds the <= 16 slots redundantly. Skipped points (|p|^2 <= 1e-3) and
// padding carry min-dist -1 so they can never win and never change.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float lane_bcast(float v, int src_lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src_lane));
}

// LX: a copy of the cloud lives in LDS and the per-point update tracks only (distance, index): the winner's
// coordinates are read back from LDS once per iteration instead of riding through three more selects per point, and
// the squared distances are formed two points at a time with packed fp32 instructions (v_pk_add / v_pk_mul: the same
// IEEE operations, no FMA) — 8 instead of 15 vector-ALU instructions per point. It is every instruction of this
// kernel, not its 0.4 ms, that the pipelined step pays for: the FPS of the next batch runs beside the MFMA kernels
// and costs them 0.14 ms per step (scripts/probes/fps_interference_probe.py, fps_vs_pair_probe.py; 0.07 with LX). LX needs
// 12 N bytes of LDS beside the index buffer: clouds up to ~4096 points; larger ones keep the coordinates in the selects.
typedef float fps_f2 __attribute__((ext_vector_type(2)));

// Dev builds only (-DPTT_DEV): ABL removes links of the iteration's dependency chain (the RESULTS are then wrong; the time per
// iteration is what is read: scripts/fps_iteration_breakdown.py, DESIGN.md section 4 "One FPS iteration"). Bits: 1 the wave maximum
// (DPP chain), 2 ballot + find-first, 4 the exchange between the waves (slot write, barrier, fold), 8 the winner's coordinates read
// back from LDS, 16 the barrier alone (slots still written and read), 32 the scan over the thread's points. ABL = 0 is the kernel.

template <int T, int P, bool LX, int ABL = 0>
__global__ __launch_bounds__(T) void fps_kernel(const float* __restrict__ xyz, int N, int npoint,
                                               int32_t* __restrict__ idx_out) {
    constexpr int W = T / 64;
    const int b = blockIdx.x;
    const float* __restrict__ pts = xyz + (size_t)b * N * 3;
    int32_t* __restrict__ out = idx_out + (size_t)b * npoint;
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wv = t >> 6;

    // The chosen indices are collected in LDS and written out once at the end: a global store inside
    // the loop puts its write-acknowledge latency (vmcnt counts stores; the barrier waits vmcnt(0))
    // on the critical path of every iteration. LX: the cloud's coordinates follow the index buffer.
    extern __shared__ int sel_lds[];
    float* xyz_lds = reinterpret_cast<float*>(sel_lds + ((npoint + 3) & ~3));

    float px[P], py[P], pz[P], md[P];
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const int k = t * P + i;
        float x = 0.f, y = 0.f, z = 0.f, m = -1.0f;
        if (k < N) {
            x = pts[3 * k + 0];
            y = pts[3 * k + 1];
            z = pts[3 * k + 2];
            const float mag = (x * x + y * y) + z * z;
            m = (mag > 1e-3f) ? 1e10f : -1.0f;
            if (LX) { xyz_lds[3 * k + 0] = x; xyz_lds[3 * k + 1] = y; xyz_lds[3 * k + 2] = z; }
        }
        px[i] = x; py[i] = y; pz[i] = z; md[i] = m;
    }

    // slots[parity][wave] = {dist, idx(bits), x, y | z}   (LX: only the first two are used)
    __shared__ __attribute__((aligned(16))) float slots[2][W > 1 ? W : 1][8];

    float lx = pts[0], ly = pts[1], lz = pts[2];
    if (t == 0 && npoint > 0) sel_lds[0] = 0;
    if (LX) __syncthreads();

    for (int j = 1; j < npoint; ++j) {
        float best = -1.0f, bx = 0.f, by = 0.f, bz = 0.f;
        int besti = 0;
        if constexpr ((P % 2) == 0) {
            const fps_f2 lxv = {lx, lx}, lyv = {ly, ly}, lzv = {lz, lz};
#pragma unroll
            for (int i = 0; i < ((ABL & 32) ? 2 : P); i += 2) {
                const fps_f2 dx = fps_f2{px[i], px[i + 1]} - lxv, dy = fps_f2{py[i], py[i + 1]} - lyv,
                             dz = fps_f2{pz[i], pz[i + 1]} - lzv;
                const fps_f2 d = (dx * dx + dy * dy) + dz * dz;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float m = fminf(md[i + h], d[h]);
                    md[i + h] = m;
                    const bool take = m > best;      // strict: the lowest index wins ties inside a thread
                    best = take ? m : best;
                    besti = take ? t * P + i + h : besti;
                    if (!LX) {
                        bx = take ? px[i + h] : bx;
                        by = take ? py[i + h] : by;
                        bz = take ? pz[i + h] : bz;
                    }
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < P; ++i) {
                const float d = sqdist3(px[i], py[i], pz[i], lx, ly, lz);
                const float m = fminf(md[i], d);
                md[i] = m;
                const bool take = m > best;          // strict: the lowest index wins ties inside a thread
                best = take ? m : best;
                besti = take ? t * P + i : besti;
                if (!LX) {
                    bx = take ? px[i] : bx;
                    by = take ? py[i] : by;
                    bz = take ? pz[i] : bz;
                }
            }
        }
        float wmax;
        if constexpr (ABL & 1) wmax = __builtin_amdgcn_readfirstlane(best); else wmax = wave_max_f32_fused(best);
        int src;
        if constexpr (ABL & 2) {
            src = (int)(__builtin_bit_cast(unsigned, wmax) & 63u);   // still a function of the maximum
        } else {
            const unsigned long long winners = __ballot(best == wmax);
            src = __ffsll((long long)winners) - 1;                // lowest lane among the maxima
            if constexpr (ABL & 1) src &= 63;
        }
        if constexpr (W == 1) {
            int widx = __builtin_amdgcn_readlane(besti, src);
            if (wmax < 0.f) widx = 0;            // no selectable point left: index 0 (upstream's besti init)
            if constexpr ((ABL & 8) != 0) {
                lx = (float)widx * 1e-3f; ly = lx; lz = lx;
            } else if constexpr (LX) {
                lx = xyz_lds[3 * widx + 0]; ly = xyz_lds[3 * widx + 1]; lz = xyz_lds[3 * widx + 2];
            } else if (wmax < 0.f) {
                lx = pts[0]; ly = pts[1]; lz = pts[2];
            } else {
                lx = lane_bcast(bx, src); ly = lane_bcast(by, src); lz = lane_bcast(bz, src);
            }
            if (lane == 0) sel_lds[j] = widx;
        } else if constexpr ((ABL & 4) != 0) {   // no exchange between the waves: every wave follows its own maximum
            int gi = __builtin_amdgcn_readlane(besti, src);
            if (wmax < 0.f) gi = 0;
            if constexpr ((ABL & 8) != 0) { lx = (float)gi * 1e-3f; ly = lx; lz = lx; }
            else { lx = xyz_lds[3 * gi + 0]; ly = xyz_lds[3 * gi + 1]; lz = xyz_lds[3 * gi + 2]; }
            if (lane == 0 && wv == 0) sel_lds[j] = gi;
        } else {
            const int par = j & 1;
            if (lane == src) {                   // the winning lane publishes its own candidate
                if constexpr (LX) {
                    *reinterpret_cast<float2*>(slots[par][wv]) = make_float2(best, __builtin_bit_cast(float, besti));
                } else {
                    float4* s4 = reinterpret_cast<float4*>(slots[par][wv]);
                    s4[0] = make_float4(best, __builtin_bit_cast(float, besti), bx, by);
                    slots[par][wv][4] = bz;
                }
            }
            if constexpr ((ABL & 16) == 0) __syncthreads();
            float gd, gx = 0.f, gy = 0.f, gz = 0.f;
            int gi;
            if constexpr (W <= 4) {
                gd = -1.0f; gi = 0;
#pragma unroll
                for (int w = 0; w < W; ++w) {      // ascending waves hold ascending indices: strict > keeps the lowest
                    if constexpr (LX) {
                        const float2 s2 = *reinterpret_cast<const float2*>(slots[par][w]);
                        const bool take = s2.x > gd;
                        gd = take ? s2.x : gd;
                        gi = take ? __builtin_bit_cast(int, s2.y) : gi;
                    } else {
                        const float4 s4 = *reinterpret_cast<const float4*>(slots[par][w]);
                        const float sz = slots[par][w][4];
                        const bool take = s4.x > gd;
                        gd = take ? s4.x : gd;
                        gi = take ? __builtin_bit_cast(int, s4.y) : gi;
                        gx = take ? s4.z : gx;
                        gy = take ? s4.w : gy;
                        gz = take ? sz : gz;
                    }
                }
            } else {                                // many waves: fold the slots with one more wave reduction
                if constexpr (LX) {
                    float2 s2 = make_float2(-1.0f, 0.f);
                    if (lane < W) s2 = *reinterpret_cast<const float2*>(slots[par][lane]);
                    gd = wave_max_f32_fused(s2.x);
                    const int sl = __ffsll((long long)__ballot(s2.x == gd)) - 1;
                    gi = __builtin_amdgcn_readlane(__builtin_bit_cast(int, s2.y), sl);
                } else {
                    float4 s4 = make_float4(-1.0f, 0.f, 0.f, 0.f);
                    float sz = 0.f;
                    if (lane < W) { s4 = *reinterpret_cast<const float4*>(slots[par][lane]); sz = slots[par][lane][4]; }
                    gd = wave_max_f32_fused(s4.x);
                    const int sl = __ffsll((long long)__ballot(s4.x == gd)) - 1;
                    gi = __builtin_amdgcn_readlane(__builtin_bit_cast(int, s4.y), sl);
                    gx = lane_bcast(s4.z, sl); gy = lane_bcast(s4.w, sl); gz = lane_bcast(sz, sl);
                }
            }
            if (gd < 0.f) gi = 0;
            if constexpr ((ABL & 16) != 0) gi &= 1023;           // (racy slots: keep the index inside the cloud)
            if constexpr ((ABL & 8) != 0) {
                lx = (float)gi * 1e-3f; ly = lx; lz = lx;
            } else if constexpr (LX) {
                lx = xyz_lds[3 * gi + 0]; ly = xyz_lds[3 * gi + 1]; lz = xyz_lds[3 * gi + 2];
            } else {
                if (gd < 0.f) { gx = pts[0]; gy = pts[1]; gz = pts[2]; }
                lx = gx; ly = gy; lz = gz;
            }
            if (t == 0) sel_lds[j] = gi;
        }
    }
    __syncthreads();
    for (int j = t; j < npoint; j += T) out[j] = sel_lds[j];
}

template <int T, int P>
static int launch_fps(const float* xyz, int B, int N, int npoint, int32_t* idx, hipStream_t s) {
    const size_t sel_bytes = (size_t)((npoint + 3) & ~3) * sizeof(int);
    if (sel_bytes + (size_t)N * 12 + 1024 <= 65536 && !dev_switches().fps_plain) {   // fits the default 64 KB with the slots
#ifdef PTT_DEV
        if (const char* e = getenv("PTT_FPS_ABL")) {              // timing ablations (wrong results): see the kernel's header
            const int abl = atoi(e);
#define PTT_FPS_ABL_CASE(A) if (abl == A) { hipLaunchKernelGGL((fps_kernel<T, P, true, A>), dim3(B), dim3(T), sel_bytes + (size_t)N * 12, s, xyz, N, npoint, idx); return check_launch("fps_kernel(abl)"); }
            PTT_FPS_ABL_CASE(1) PTT_FPS_ABL_CASE(2) PTT_FPS_ABL_CASE(3) PTT_FPS_ABL_CASE(4) PTT_FPS_ABL_CASE(8) PTT_FPS_ABL_CASE(12) PTT_FPS_ABL_CASE(15)
            PTT_FPS_ABL_CASE(16) PTT_FPS_ABL_CASE(32) PTT_FPS_ABL_CASE(47) PTT_FPS_ABL_CASE(7) PTT_FPS_ABL_CASE(24) PTT_FPS_ABL_CASE(40)
#undef PTT_FPS_ABL_CASE
        }
#endif
        hipLaunchKernelGGL((fps_kernel<T, P, true>), dim3(B), dim3(T), sel_bytes + (size_t)N * 12, s, xyz, N, npoint, idx);
        return check_launch("fps_kernel");
    }
    hipLaunchKernelGGL((fps_kernel<T, P, false>), dim3(B), dim3(T), sel_bytes, s, xyz, N, npoint, idx);
    return check_launch("fps_kernel");
}

// ------------------------------------------------------------------------------------------
// Ball query. A wave owns CPW consecutive centres of one cloud and sweeps the cloud 64 points at a
// time (coalesced); per centre a ballot + prefix popcount keeps the hits in index order, and a
// centre stops taking part as soon as nsample hits are stored (the wave stops when all have).
// CPW = 4 for launches that fill the device anyway: a point is loaded once and tested against four
// centres — with one centre per wave every wave re-reads the whole cloud from L1/L2, and the
// 16384-point stress frames ran at half the vector-L1 bandwidth of the chip (6 G tests x 12 B in
// 4 ms). CPW = 1 for small launches (one tracklet frame: 512 centres), where waves are the scarce
// thing. The arithmetic per (centre, point) pair and the order of the hits are the same in both.
// ------------------------------------------------------------------------------------------
template <int CPW>
__device__ __forceinline__ void ball_sweep(const float* __restrict__ pts, int N, float r2, int ns, const float (&cx)[CPW],
                                           const float (&cy)[CPW], const float (&cz)[CPW], int ncentres,
                                           int32_t* __restrict__ out0, int lane) {
    int cnt[CPW], first[CPW];
#pragma unroll
    for (int c = 0; c < CPW; ++c) { cnt[c] = (c < ncentres) ? 0 : ns; first[c] = 0; }      // absent centres are "done"
    for (int base = 0; base < N; base += 64) {
        const int k = base + lane;
        const bool in = k < N;
        float px = 0.f, py = 0.f, pz = 0.f;
        if (in) { px = pts[3 * k + 0]; py = pts[3 * k + 1]; pz = pts[3 * k + 2]; }
        bool active = false;
#pragma unroll
        for (int c = 0; c < CPW; ++c) {
            if (cnt[c] >= ns) continue;                       // wave-uniform
            const float d = sqdist3(cx[c], cy[c], cz[c], px, py, pz);
            const bool hit = in && d < r2;
            const unsigned long long mask = __ballot(hit);
            if (mask != 0ull) {
                if (cnt[c] == 0) first[c] = base + (__ffsll((long long)mask) - 1);
                const int pos = cnt[c] + __popcll(mask & ((1ull << lane) - 1ull));
                if (hit && pos < ns) out0[(size_t)c * ns + pos] = k;
                cnt[c] += __popcll(mask);
            }
            active = active || cnt[c] < ns;
        }
        if (!active) break;
    }
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
        if (c >= ncentres) break;
        const int fill = (cnt[c] > 0) ? first[c] : 0;
        for (int s = cnt[c] + lane; s < ns; s += 64) out0[(size_t)c * ns + s] = fill;
    }
}

template <int CPW>
__global__ __launch_bounds__(256) void ball_query_kernel(const float* __restrict__ new_xyz,
                                                         const float* __restrict__ xyz, int BM, int M, int N,
                                                         float r2, int ns, int32_t* __restrict__ idx_out) {
    const int centre = (blockIdx.x * 4 + (threadIdx.x >> 6)) * CPW;      // CPW divides M: one cloud per wave
    if (centre >= BM) return;
    const int lane = threadIdx.x & 63;
    const int b = centre / M;
    float cx[CPW], cy[CPW], cz[CPW];
    const int nc = min(CPW, BM - centre);
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
        const int cc = centre + (c < nc ? c : 0);
        cx[c] = new_xyz[(size_t)cc * 3 + 0]; cy[c] = new_xyz[(size_t)cc * 3 + 1]; cz[c] = new_xyz[(size_t)cc * 3 + 2];
    }
    ball_sweep<CPW>(xyz + (size_t)b * N * 3, N, r2, ns, cx, cy, cz, nc, idx_out + (size_t)centre * ns, lane);
}

// Centre selection and ball query of one SA level in ONE launch (pointnet2_modules.py:79-83,90): the wave that owns
// centre c reads its coordinates through the sample index (sel == NULL: the first M points, 'sequence' sampling),
// writes them to new_xyz and the index as int64, then sweeps the cloud as ball_query_kernel does. Same results as
// select_centres_kernel + ball_query_kernel; one launch and one dependent round trip less per level, which is what a
// B = 1 tracklet frame is made of.
template <int CPW>
__global__ __launch_bounds__(256) void centres_ball_query_kernel(const float* __restrict__ xyz, const int32_t* __restrict__ sel,
                                                                 int BM, int M, int N, float r2, int ns,
                                                                 float* __restrict__ new_xyz, long long* __restrict__ idx64,
                                                                 int32_t* __restrict__ idx_out) {
    const int centre = (blockIdx.x * 4 + (threadIdx.x >> 6)) * CPW;
    if (centre >= BM) return;
    const int lane = threadIdx.x & 63;
    const int b = centre / M;
    const float* __restrict__ pts = xyz + (size_t)b * N * 3;
    float cx[CPW], cy[CPW], cz[CPW];
    const int nc = min(CPW, BM - centre);
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
        const int cc = centre + (c < nc ? c : 0);
        const int n = sel ? sel[cc] : cc - b * M;
        cx[c] = pts[3 * n + 0]; cy[c] = pts[3 * n + 1]; cz[c] = pts[3 * n + 2];
        if (lane == 0 && c < nc) {
            new_xyz[(size_t)cc * 3 + 0] = cx[c]; new_xyz[(size_t)cc * 3 + 1] = cy[c]; new_xyz[(size_t)cc * 3 + 2] = cz[c];
            if (idx64) idx64[cc] = n;
        }
    }
    ball_sweep<CPW>(pts, N, r2, ns, cx, cy, cz, nc, idx_out + (size_t)centre * ns, lane);
}

// ------------------------------------------------------------------------------------------
// gather / group and their scatter-add backward passes (channel-major features, as the
// reference hands them over).
// ------------------------------------------------------------------------------------------
__global__ void gather_kernel(const float* __restrict__ feat, const int32_t* __restrict__ idx, int C, int N, int M,
                              float* __restrict__ out, size_t total) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(e % M);
        const size_t bc = e / M;
        const int b = (int)(bc / C);
        out[e] = feat[bc * N + idx[(size_t)b * M + j]];
    }
}
__global__ void gather_grad_kernel(const float* __restrict__ go, const int32_t* __restrict__ idx, int C, int N, int M,
                                   float* __restrict__ gf, size_t total) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(e % M);
        const size_t bc = e / M;
        const int b = (int)(bc / C);
        atomicAdd(&gf[bc * N + idx[(size_t)b * M + j]], go[e]);
    }
}
__global__ void group_kernel(const float* __restrict__ feat, const int32_t* __restrict__ idx, int C, int N, int M,
                             int ns, float* __restrict__ out, size_t total) {
    const size_t mk = (size_t)M * ns;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t jk = e % mk;
        const size_t bc = e / mk;
        const int b = (int)(bc / C);
        out[e] = feat[bc * N + idx[(size_t)b * mk + jk]];
    }
}
__global__ void group_grad_kernel(const float* __restrict__ go, const int32_t* __restrict__ idx, int C, int N, int M,
                                  int ns, float* __restrict__ gf, size_t total) {
    const size_t mk = (size_t)M * ns;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t jk = e % mk;
        const size_t bc = e / mk;
        const int b = (int)(bc / C);
        atomicAdd(&gf[bc * N + idx[(size_t)b * mk + jk]], go[e]);
    }
}

// Same scatter-add with the accumulation in LDS: a workgroup owns cloud b and CC channels, keeps out[b, c0..c0+CC, 0..N)
// in shared memory, streams its CC rows of grad_out coalesced (idx re-read from cache per channel) and adds with
// ds_add_f32; every output element is then written exactly once with a plain store (no global atomics, no memset
// needed). 6x faster than the global-atomic form at the training shapes (1.8 -> 0.3 ms for 48 x 128 x 256 x 32).
// The order of the additions inside a workgroup is still not fixed, i.e. sums are not bit-reproducible run to run —
// upstream's atomicAdd kernel has the same property.
__global__ __launch_bounds__(512) void group_grad_lds_kernel(const float* __restrict__ go, const int32_t* __restrict__ idx,
                                                             int C, int N, int M, int ns, int CC, int nchunks,
                                                             float* __restrict__ gf) {
    extern __shared__ float acc[];                       // [CC][N]
    const int b = blockIdx.x / nchunks, c0 = (blockIdx.x % nchunks) * CC;
    const int cc = min(CC, C - c0), mk = M * ns;
    for (int i = threadIdx.x; i < cc * N; i += blockDim.x) acc[i] = 0.f;
    __syncthreads();
    const int32_t* ib = idx + (size_t)b * mk;
    for (int ci = 0; ci < cc; ++ci) {
        const float* g = go + ((size_t)b * C + c0 + ci) * mk;
        float* a = acc + ci * N;
        for (int jk = threadIdx.x; jk < mk; jk += blockDim.x) atomicAdd(&a[ib[jk]], g[jk]);
    }
    __syncthreads();
    float* o = gf + ((size_t)b * C + c0) * N;
    for (int i = threadIdx.x; i < cc * N; i += blockDim.x) o[i] = acc[i];
}

// ------------------------------------------------------------------------------------------
// Deterministic scatter-add (N3): out[b,c,n] = sum over the entries e of cloud b with idx[b,e] == n of src[b,c,e],
// added in ASCENDING e — the order of a sequential loop, so the result is bit-identical run to run and to the CPU
// oracle. Two kernels:
//   scatter_csr_kernel   one workgroup per cloud sorts the keys idx*Epad + e in LDS (bitonic) and writes the entry
//                        order plus the start of every bin (lower_bound per bin) — shared by all C channels;
//   scatter_add_det_kernel  a workgroup owns cloud b and a chunk of channels: per channel it stages the E source
//                        values in LDS with coalesced loads, then thread n walks bin n's entries in order.
// E <= 16384 entries per cloud (64 KB of keys); larger problems stay on the atomic kernels.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void scatter_csr_kernel(const int32_t* __restrict__ idx, int N, int E, int Epad,
                                                           int32_t* __restrict__ order, int32_t* __restrict__ start) {
    extern __shared__ unsigned keys[];                   // [Epad]
    const int b = blockIdx.x;
    const int32_t* ib = idx + (size_t)b * E;
    for (int e = threadIdx.x; e < Epad; e += blockDim.x)
        keys[e] = (e < E) ? (unsigned)ib[e] * (unsigned)Epad + (unsigned)e : 0xffffffffu;
    __syncthreads();
    for (int k = 2; k <= Epad; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < Epad; i += blockDim.x) {
                const int x = i ^ j;
                if (x > i) {
                    const unsigned a = keys[i], c = keys[x];
                    const bool up = (i & k) == 0;
                    if (up ? (a > c) : (a < c)) { keys[i] = c; keys[x] = a; }
                }
            }
            __syncthreads();
        }
    for (int e = threadIdx.x; e < E; e += blockDim.x) order[(size_t)b * E + e] = (int32_t)(keys[e] & (unsigned)(Epad - 1));
    for (int n = threadIdx.x; n <= N; n += blockDim.x) {  // first sorted slot whose bin is >= n
        const unsigned want = (unsigned)n * (unsigned)Epad;
        int lo = 0, hi = E;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (keys[mid] < want) lo = mid + 1; else hi = mid;
        }
        start[(size_t)b * (N + 1) + n] = (n == N) ? E : lo;
    }
}

// The same CSR by a stable COUNTING sort (N <= 2048 bins), ~6x faster than the bitonic network at E = 16384: wave w owns the
// contiguous entries [w * per, (w + 1) * per); (1) every wave histograms its range into its own row of counters, (2) per bin the
// rows become exclusive offsets (wave order) and the bin totals are scanned into `start`, (3) every wave walks its range in
// rounds of 64 entries, in order: the lanes of a round that share a bin rank themselves with a ballot (ascending lane = ascending
// e), place their entries at start[bin] + offset[w][bin] + rank and advance the offset. Ascending e inside every bin, as the
// sort gives it; no atomics decide an order.
__global__ __launch_bounds__(1024) void scatter_csr_count_kernel(const int32_t* __restrict__ idx, int N, int E, int32_t* __restrict__ order,
                                                                 int32_t* __restrict__ start) {
    extern __shared__ unsigned csr_smem[];              // off[16][N] | binstart[N + 1] | wave_tot[16]
    unsigned* off = csr_smem;
    unsigned* binstart = csr_smem + 16 * N;
    unsigned* wave_tot = binstart + N + 1;
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int32_t* ib = idx + (size_t)b * E;
    for (int i = t; i < 16 * N; i += 1024) off[i] = 0u;
    __syncthreads();
    const int per = ((E + 15) / 16 + 63) / 64 * 64;
    const int e0 = w * per, e1 = min(E, e0 + per);
    for (int e = e0 + lane; e < e1; e += 64) {
        const unsigned n = (unsigned)ib[e];
        if (n < (unsigned)N) atomicAdd(&off[w * N + n], 1u);              // a count: the order of the additions is immaterial
    }
    __syncthreads();
    // per bin: exclusive offsets over the waves; bins 2t, 2t + 1 (N <= 2048) of thread t, then a scan of the bin totals
    unsigned tot[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int n = 2 * t + u;
        unsigned run = 0;
        if (n < N)
            for (int k = 0; k < 16; ++k) { const unsigned c = off[k * N + n]; off[k * N + n] = run; run += c; }
        tot[u] = run;
    }
    unsigned incl = tot[0] + tot[1];
    for (int d = 1; d < 64; d <<= 1) { const unsigned v = __shfl_up(incl, d, 64); if (lane >= d) incl += v; }
    if (lane == 63) wave_tot[w] = incl;
    __syncthreads();
    unsigned base = 0;
    for (int k = 0; k < w; ++k) base += wave_tot[k];
    const unsigned excl = base + incl - (tot[0] + tot[1]);
    if (2 * t < N) binstart[2 * t] = excl;
    if (2 * t + 1 < N) binstart[2 * t + 1] = excl + tot[0];
    if (t == 1023) binstart[N] = base + incl;
    __syncthreads();
    for (int n = t; n <= N; n += 1024) start[(size_t)b * (N + 1) + n] = (int32_t)binstart[n];
    for (int r0 = e0; r0 < e1; r0 += 64) {
        const int e = r0 + lane;
        const unsigned n = e < e1 ? (unsigned)ib[e] : 0xffffffffu;
        const bool valid = n < (unsigned)N;
        unsigned long long todo = __ballot(valid);
        unsigned rank = 0, cnt = 0;
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const unsigned lb = (unsigned)__builtin_amdgcn_readlane((int)n, leader);
            const unsigned long long same = __ballot(valid && n == lb);
            if (valid && n == lb) {
                rank = (unsigned)__popcll(same & ((1ull << lane) - 1ull));
                cnt = (unsigned)__popcll(same);
            }
            todo &= ~same;
        }
        if (valid) {
            const unsigned o = off[w * N + n];                            // read by all lanes of the bin before its last lane writes
            order[(size_t)b * E + binstart[n] + o + rank] = e;
            __builtin_amdgcn_wave_barrier();
            if (rank + 1 == cnt) off[w * N + n] = o + cnt;
        }
    }
}
static int launch_scatter_csr(const int32_t* idx, int B, int N, int E, int32_t* order, int32_t* start, hipStream_t s) {
    if (N <= 2048) {
        const int lds = (16 * N + N + 1 + 16) * (int)sizeof(unsigned);
        if (int rc = set_lds_limit(reinterpret_cast<const void*>(scatter_csr_count_kernel), lds)) return rc;
        hipLaunchKernelGGL(scatter_csr_count_kernel, dim3(B), dim3(1024), lds, s, idx, N, E, order, start);
        return check_launch("scatter_csr_count_kernel");
    }
    int Epad = 2;
    while (Epad < E) Epad <<= 1;
    hipLaunchKernelGGL(scatter_csr_kernel, dim3(B), dim3(1024), (size_t)Epad * sizeof(unsigned), s, idx, N, E, Epad, order, start);
    return check_launch("scatter_csr_kernel");
}

__global__ __launch_bounds__(512) void scatter_add_det_kernel(const float* __restrict__ src, const int32_t* __restrict__ order,
                                                              const int32_t* __restrict__ start, int C, int N, int E,
                                                              int CC, int nchunks, float* __restrict__ out) {
    extern __shared__ float row[];                       // [E] source values of one channel, then [E] ints: the order
    int* ord = reinterpret_cast<int*>(row + E);
    const int b = blockIdx.x / nchunks, c0 = (blockIdx.x % nchunks) * CC;
    const int cc = min(CC, C - c0);
    for (int e = threadIdx.x; e < E; e += blockDim.x) ord[e] = order[(size_t)b * E + e];
    const int32_t* st = start + (size_t)b * (N + 1);
    for (int ci = 0; ci < cc; ++ci) {
        const float* g = src + ((size_t)b * C + c0 + ci) * E;
        __syncthreads();                                  // previous channel's readers are done (and ord is in place)
        for (int e = threadIdx.x; e < E; e += blockDim.x) row[e] = g[e];
        __syncthreads();
        float* o = out + ((size_t)b * C + c0 + ci) * N;
        for (int n = threadIdx.x; n < N; n += blockDim.x) {
            float acc = 0.f;
            for (int p = st[n], pe = st[n + 1]; p < pe; ++p) acc += row[ord[p]];
            o[n] = acc;
        }
    }
}

// centres of one SA level in a single launch: new_xyz[b,m,:] = xyz[b, idx[b,m], :] (idx NULL = the first M
// points, the 'sequence' sampling of pointnet2_modules.py:70-71) and the int64 copy of the indices the
// module returns (pointnet2_modules.py:90).
__global__ void select_centres_kernel(const float* __restrict__ xyz, const int32_t* __restrict__ idx, int N, int M,
                                      float* __restrict__ new_xyz, long long* __restrict__ idx64, size_t total) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int m = (int)(e % M);
        const size_t b = e / M;
        const int n = idx ? idx[e] : m;
        const float* src = xyz + (b * N + n) * 3;
        float* dst = new_xyz + e * 3;
        dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2];
        if (idx64) idx64[e] = n;
    }
}

static inline int grid_for(size_t total, int block) {
    size_t g = (total + block - 1) / block;
    if (g > 2048u * 8u) g = 2048u * 8u;
    if (g == 0) g = 1;
    return (int)g;
}

// ------------------------------------------------------------------------------------------
// kNN inside one cloud: one wave per query point, P candidates per lane in registers,
// k rounds of (lane-local min, DPP wave min on distance, DPP wave min on index among ties).
// ------------------------------------------------------------------------------------------
template <int P>
__global__ __launch_bounds__(256) void knn_kernel(const float* __restrict__ xyz, int BN, int N, int k,
                                                  int32_t* __restrict__ idx_out, float* __restrict__ rel_out) {
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= BN) return;
    const int lane = threadIdx.x & 63;
    const int b = q / N;
    const float* __restrict__ pts = xyz + (size_t)b * N * 3;
    const float qx = xyz[(size_t)q * 3 + 0], qy = xyz[(size_t)q * 3 + 1], qz = xyz[(size_t)q * 3 + 2];
    int32_t* __restrict__ out = idx_out + (size_t)q * k;

    float d[P];
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const int c = lane + i * 64;
        d[i] = (c < N) ? knn_dist(qx, qy, qz, pts[3 * c + 0], pts[3 * c + 1], pts[3 * c + 2]) : __builtin_inff();
    }
    for (int r = 0; r < k; ++r) {
        float best = __builtin_inff();
        int besti = INT_MAX;
#pragma unroll
        for (int i = 0; i < P; ++i) {
            const int c = lane + i * 64;
            if (c < N && d[i] < best) { best = d[i]; besti = c; }
        }
        const float wmin = wave_min_f32(best);
        const int sel = wave_min_i32((best == wmin) ? besti : INT_MAX);
        if (lane == 0) out[r] = sel;
        if (rel_out && lane == (sel & 63)) {           // the owning lane also emits xyz_query - xyz_neighbour
            float* ro = rel_out + ((size_t)q * k + r) * 3;
            ro[0] = qx - pts[3 * sel + 0]; ro[1] = qy - pts[3 * sel + 1]; ro[2] = qz - pts[3 * sel + 2];
        }
        // retire the winner: it can never be the minimum again
#pragma unroll
        for (int i = 0; i < P; ++i)
            if (lane + i * 64 == sel) d[i] = __builtin_nanf("");
    }
}


// ------------------------------------------------------------------------------------------
// Point jobs (round 4): the ball queries of SEVERAL set-abstraction levels and a kNN in ONE launch, for the launch chain of one
// tracklet frame. With 'sequence' sampling (tools/cfgs/kitti_models/ptt.yaml:42) the points of level l + 1 are the first
// centres of level l, i.e. raw[sel[k]] for the level-0 sample indices `sel`: every level's centres and points can be read
// from the RAW cloud through `sel`, so no level waits for the centre coordinates another workgroup of the same launch
// writes. A wave owns one centre (ball query: centre selection + sweep as centres_ball_query_kernel<1>) or one query point
// (kNN as knn_kernel); same arithmetic, same index order, same results as the per-level launches.
// ------------------------------------------------------------------------------------------
struct PointJobDev {
    const float* xyz;           // raw clouds (B, Nraw, 3)
    const int32_t* csel;        // (B, sel_ld) centre / query selection, NULL = identity
    const int32_t* psel;        // (B, sel_ld) point selection, NULL = identity
    float* new_xyz; long long* idx64; int32_t* idx_out; float* rel_out;
    int kind, sel_ld, B, Nraw, Npts, M, ns, wave0;
    float r2;
};
struct PointJobs { PointJobDev j[PTT_POINT_JOBS_MAX]; int n; };

__global__ __launch_bounds__(256) void point_jobs_kernel(PointJobs P) {
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    int ji = 0;
#pragma unroll
    for (int i = 1; i < PTT_POINT_JOBS_MAX; ++i)
        if (i < P.n && gw >= P.j[i].wave0) ji = i;
    const PointJobDev& J = P.j[ji];
    const int c = gw - J.wave0;                          // centre / query of this wave inside the job
    if (c >= J.B * J.M) return;
    const int lane = threadIdx.x & 63;
    const int b = c / J.M, m = c - b * J.M;
    const float* __restrict__ raw = J.xyz + (size_t)b * J.Nraw * 3;
    const int32_t* __restrict__ cs = J.csel ? J.csel + (size_t)b * J.sel_ld : nullptr;
    const int32_t* __restrict__ ps = J.psel ? J.psel + (size_t)b * J.sel_ld : nullptr;
    const int nc = cs ? cs[m] : m;
    const float cx = raw[3 * nc + 0], cy = raw[3 * nc + 1], cz = raw[3 * nc + 2];
    if (J.kind == 0) {
        if (lane == 0) {
            J.new_xyz[(size_t)c * 3 + 0] = cx; J.new_xyz[(size_t)c * 3 + 1] = cy; J.new_xyz[(size_t)c * 3 + 2] = cz;
            if (J.idx64) J.idx64[c] = nc;
        }
        int32_t* __restrict__ out = J.idx_out + (size_t)c * J.ns;
        int cnt = 0, first = 0;
        for (int base = 0; base < J.Npts && cnt < J.ns; base += 64) {
            const int k = base + lane;
            const bool in = k < J.Npts;
            const int pi = in ? (ps ? ps[k] : k) : 0;
            const float d = sqdist3(cx, cy, cz, raw[3 * pi + 0], raw[3 * pi + 1], raw[3 * pi + 2]);
            const bool hit = in && d < J.r2;
            const unsigned long long mask = __ballot(hit);
            if (mask != 0ull) {
                if (cnt == 0) first = base + (__ffsll((long long)mask) - 1);
                const int pos = cnt + __popcll(mask & ((1ull << lane) - 1ull));
                if (hit && pos < J.ns) out[pos] = k;
                cnt += __popcll(mask);
            }
        }
        const int fill = cnt > 0 ? first : 0;
        for (int s2 = cnt + lane; s2 < J.ns; s2 += 64) out[s2] = fill;
        return;
    }
    // kNN among the job's Npts (<= 128) points; the query is point m of them (M == Npts)
    float d[2];
    float px[2], py[2], pz[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int k = lane + i * 64;
        const int pi = k < J.Npts ? (ps ? ps[k] : k) : 0;
        px[i] = raw[3 * pi + 0]; py[i] = raw[3 * pi + 1]; pz[i] = raw[3 * pi + 2];
        d[i] = k < J.Npts ? knn_dist(cx, cy, cz, px[i], py[i], pz[i]) : __builtin_inff();
    }
    int32_t* __restrict__ out = J.idx_out + (size_t)c * J.ns;
    for (int r = 0; r < J.ns; ++r) {
        float best = __builtin_inff();
        int besti = INT_MAX;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int k = lane + i * 64;
            if (k < J.Npts && d[i] < best) { best = d[i]; besti = k; }
        }
        const float wmin = wave_min_f32(best);
        const int sel = wave_min_i32((best == wmin) ? besti : INT_MAX);
        if (lane == 0) out[r] = sel;
        if (J.rel_out && lane == (sel & 63)) {
            const int i = sel >> 6;
            float* ro = J.rel_out + ((size_t)c * J.ns + r) * 3;
            ro[0] = cx - (i ? px[1] : px[0]); ro[1] = cy - (i ? py[1] : py[0]); ro[2] = cz - (i ? pz[1] : pz[0]);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
            if (lane + i * 64 == sel) d[i] = __builtin_nanf("");
    }
}

// ------------------------------------------------------------------------------------------
// One launch for the index ops of a small set-abstraction level that samples with FPS and whose centres then meet in a
// Point-Transformer block (vote_aggregation + the box head's transformer at one tracklet frame: 128 votes -> 64 proposals,
// box_voting_head.py:75-86): furthest point sampling, centre selection, ball query and the kNN of the centres among
// themselves. The cloud (<= 256 points) is copied to LDS; wave 0 of EVERY workgroup runs the cloud's FPS chain (63 dependent
// iterations at 64 proposals — repeating it per workgroup is cheaper than a second and third launch behind it), then each of
// the workgroup's waves takes one centre: ball query over the cloud, kNN over the centres. Same arithmetic and tie-breaks as
// fps_kernel<64,P,true>, centres_ball_query_kernel<1> and knn_kernel: identical indices.
// ------------------------------------------------------------------------------------------
struct FbkParams {
    const float* xyz; int B, N, M, ns, k; float r2;
    int32_t* inds; long long* inds64; float* new_xyz; int32_t* idx; int32_t* knn; float* rel;
};

__global__ __launch_bounds__(256) void fps_ball_knn_kernel(FbkParams p) {
    constexpr int P = 4;                                  // points per lane of the sampling wave: N <= 256
    __shared__ int sel[128];
    __shared__ float cl[256 * 3];
    const int G = (p.M + 3) / 4;
    const int b = blockIdx.x / G, g = blockIdx.x - b * G;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const float* __restrict__ pts = p.xyz + (size_t)b * p.N * 3;
    for (int e = t; e < p.N * 3; e += 256) cl[e] = pts[e];
    __syncthreads();
    if (wv == 0) {
        float px[P], py[P], pz[P], md[P];
#pragma unroll
        for (int i = 0; i < P; ++i) {
            const int kk = lane * P + i;
            float x = 0.f, y = 0.f, z = 0.f, m = -1.0f;
            if (kk < p.N) {
                x = cl[3 * kk + 0]; y = cl[3 * kk + 1]; z = cl[3 * kk + 2];
                const float mag = (x * x + y * y) + z * z;
                m = (mag > 1e-3f) ? 1e10f : -1.0f;
            }
            px[i] = x; py[i] = y; pz[i] = z; md[i] = m;
        }
        float lx = cl[0], ly = cl[1], lz = cl[2];
        if (lane == 0) sel[0] = 0;
        for (int j = 1; j < p.M; ++j) {
            float best = -1.0f;
            int besti = 0;
#pragma unroll
            for (int i = 0; i < P; ++i) {
                const float d = sqdist3(px[i], py[i], pz[i], lx, ly, lz);
                const float m = fminf(md[i], d);
                md[i] = m;
                const bool take = m > best;              // strict: the lowest index wins ties inside a lane
                best = take ? m : best;
                besti = take ? lane * P + i : besti;
            }
            const float wmax = wave_max_f32_fused(best);
            const unsigned long long winners = __ballot(best == wmax);
            const int src = __ffsll((long long)winners) - 1;
            int widx = __builtin_amdgcn_readlane(besti, src);
            if (wmax < 0.f) widx = 0;
            lx = cl[3 * widx + 0]; ly = cl[3 * widx + 1]; lz = cl[3 * widx + 2];
            if (lane == 0) sel[j] = widx;
        }
    }
    __syncthreads();
    if (g == 0)
        for (int j = t; j < p.M; j += 256) {
            p.inds[(size_t)b * p.M + j] = sel[j];
            if (p.inds64) p.inds64[(size_t)b * p.M + j] = sel[j];
        }
    const int m = g * 4 + wv;
    if (m >= p.M) return;
    const size_t c = (size_t)b * p.M + m;
    const int nc = sel[m];
    const float cx = cl[3 * nc + 0], cy = cl[3 * nc + 1], cz = cl[3 * nc + 2];
    if (lane == 0) { p.new_xyz[c * 3 + 0] = cx; p.new_xyz[c * 3 + 1] = cy; p.new_xyz[c * 3 + 2] = cz; }
    {   // ball query of centre m over the cloud
        int32_t* __restrict__ out = p.idx + c * p.ns;
        int cnt = 0, first = 0;
        for (int base = 0; base < p.N && cnt < p.ns; base += 64) {
            const int kk = base + lane;
            const bool in = kk < p.N;
            const int pi = in ? kk : 0;
            const float d = sqdist3(cx, cy, cz, cl[3 * pi + 0], cl[3 * pi + 1], cl[3 * pi + 2]);
            const bool hit = in && d < p.r2;
            const unsigned long long mask = __ballot(hit);
            if (mask != 0ull) {
                if (cnt == 0) first = base + (__ffsll((long long)mask) - 1);
                const int pos = cnt + __popcll(mask & ((1ull << lane) - 1ull));
                if (hit && pos < p.ns) out[pos] = kk;
                cnt += __popcll(mask);
            }
        }
        const int fill = cnt > 0 ? first : 0;
        for (int s2 = cnt + lane; s2 < p.ns; s2 += 64) out[s2] = fill;
    }
    if (!p.knn) return;
    float d[2], qx[2], qy[2], qz[2];                      // kNN of centre m among the M <= 128 centres
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int kk = lane + i * 64;
        const int pi = kk < p.M ? sel[kk] : 0;
        qx[i] = cl[3 * pi + 0]; qy[i] = cl[3 * pi + 1]; qz[i] = cl[3 * pi + 2];
        d[i] = kk < p.M ? knn_dist(cx, cy, cz, qx[i], qy[i], qz[i]) : __builtin_inff();
    }
    int32_t* __restrict__ ko = p.knn + c * p.k;
    for (int r = 0; r < p.k; ++r) {
        float best = __builtin_inff();
        int besti = INT_MAX;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int kk = lane + i * 64;
            if (kk < p.M && d[i] < best) { best = d[i]; besti = kk; }
        }
        const float wmin = wave_min_f32(best);
        const int s2 = wave_min_i32((best == wmin) ? besti : INT_MAX);
        if (lane == 0) ko[r] = s2;
        if (p.rel && lane == (s2 & 63)) {
            const int i = s2 >> 6;
            float* ro = p.rel + (c * p.k + r) * 3;
            ro[0] = cx - (i ? qx[1] : qx[0]); ro[1] = cy - (i ? qy[1] : qy[0]); ro[2] = cz - (i ? qz[1] : qz[0]);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
            if (lane + i * 64 == s2) d[i] = __builtin_nanf("");
    }
}

// FPS of clouds beyond the register-resident kernel's reach (N > 16384 or npoint > 15360): the running min-distance lives
// in a caller-provided workspace (4 N bytes per cloud, each thread touching only its own strided entries: no fence), the
// coordinates are re-read from L2 every iteration and the picks go straight to global memory. One 1024-thread workgroup per
// cloud; the arg-max travels as a (distance, index) pair — larger distance, then lower index — through the wave
// (shuffles) and through one parity-double-buffered LDS slot per wave (one barrier per iteration). Same arithmetic
// contract and the same picks as fps_kernel; ~N / 1024 L2 round trips per iteration instead of none — the slow, unlimited form.
__device__ __forceinline__ void fps_better(float& v, int& i, float ov, int oi) {
    if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
}
__global__ __launch_bounds__(1024) void fps_ws_kernel(const float* __restrict__ xyz, int N, int npoint, float* __restrict__ ws,
                                                      int32_t* __restrict__ idx_out) {
    __shared__ float slot_v[2][16];
    __shared__ int slot_i[2][16];
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const float* __restrict__ pts = xyz + (size_t)b * N * 3;
    float* __restrict__ md = ws + (size_t)b * N;
    int32_t* __restrict__ out = idx_out + (size_t)b * npoint;
    for (int k = t; k < N; k += 1024) {
        const float x = pts[3 * k], y = pts[3 * k + 1], z = pts[3 * k + 2];
        md[k] = ((x * x + y * y) + z * z > 1e-3f) ? 1e10f : -1.f;          // -1: skipped, never updated, never chosen
    }
    if (t == 0) out[0] = 0;
    int last = 0;
    for (int j = 1; j < npoint; ++j) {
        const float lx = pts[3 * last], ly = pts[3 * last + 1], lz = pts[3 * last + 2];
        float best = -1.f;
        int besti = 0;
        for (int k = t; k < N; k += 1024) {
            const float m0 = md[k];
            if (m0 < 0.f) continue;
            const float d = sqdist3(pts[3 * k], pts[3 * k + 1], pts[3 * k + 2], lx, ly, lz);
            const float m = d < m0 ? d : m0;
            md[k] = m;
            if (m > best) { best = m; besti = k; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
            fps_better(best, besti, __shfl_xor(best, off, 64), __shfl_xor(besti, off, 64));
        const int par = j & 1;
        if (lane == 0) { slot_v[par][wv] = best; slot_i[par][wv] = besti; }
        __syncthreads();
        best = slot_v[par][0]; besti = slot_i[par][0];
#pragma unroll
        for (int w = 1; w < 16; ++w) fps_better(best, besti, slot_v[par][w], slot_i[par][w]);
        last = besti;
        if (t == 0) out[j] = besti;
    }
}

// A spatial processing order for the points of every cloud: Morton keys (10 bits per axis inside the cloud's bounding box) sorted
// in LDS (bitonic, one workgroup per cloud). The Point-Transformer pair kernel gives a workgroup two points and gathers the k | v
// rows of their 32 neighbours: in sampling order (furthest point sampling = as far apart as possible) consecutive workgroups
// share no neighbour and every row is fetched from HBM ~10 times at 2048 points per cloud (profiles/r04z_pmc_stress); along the
// Morton curve the workgroups that run together on an XCD gather from the same few hundred rows. Pure scheduling: results do not
// depend on the order.
__device__ __forceinline__ unsigned morton_spread10(unsigned v) {        // 10 bits -> every third bit
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}
__global__ __launch_bounds__(1024) void spatial_order_kernel(const float* __restrict__ xyz, int N, int P, int32_t* __restrict__ order) {
    extern __shared__ unsigned long long mkeys[];                         // [P]
    __shared__ float red[6][16];
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const float* __restrict__ pts = xyz + (size_t)b * N * 3;
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (int k = t; k < N; k += 1024)
#pragma unroll
        for (int c = 0; c < 3; ++c) { const float v = pts[3 * k + c]; lo[c] = fminf(lo[c], v); hi[c] = fmaxf(hi[c], v); }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float l = wave_min_f32(lo[c]), h = wave_max_f32(hi[c]);
        if (lane == 0) { red[c][wv] = l; red[3 + c][wv] = h; }
    }
    __syncthreads();
    float scale[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float l = red[c][0], h = red[3 + c][0];
        for (int w = 1; w < 16; ++w) { l = fminf(l, red[c][w]); h = fmaxf(h, red[3 + c][w]); }
        lo[c] = l;
        scale[c] = h > l ? 1023.0f / (h - l) : 0.f;
    }
    for (int k = t; k < P; k += 1024) {
        unsigned long long key = ~0ull;                                  // padding sorts to the end
        if (k < N) {
            unsigned m = 0;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float q = (pts[3 * k + c] - lo[c]) * scale[c];
                q = q < 0.f ? 0.f : (q > 1023.f ? 1023.f : q);           // (NaN coordinates: cell 0)
                m |= morton_spread10((unsigned)q) << c;
            }
            key = ((unsigned long long)m << 32) | (unsigned)k;           // the index breaks ties: the result is a permutation
        }
        mkeys[k] = key;
    }
    __syncthreads();
    for (int size = 2; size <= P; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = t; i < (P >> 1); i += 1024) {
                const int lo_i = 2 * i - (i & (stride - 1)), hi_i = lo_i + stride;
                const bool up = (lo_i & size) == 0;
                const unsigned long long a = mkeys[lo_i], c2 = mkeys[hi_i];
                if ((a > c2) == up) { mkeys[lo_i] = c2; mkeys[hi_i] = a; }
            }
            __syncthreads();
        }
    for (int k = t; k < N; k += 1024) order[(size_t)b * N + k] = b * N + (int)(unsigned)mkeys[k];
}

// ------------------------------------------------------------------------------------------
// Ball query through a uniform grid (round 5), for large clouds. The sweep above tests every (centre, point) pair until a ball
// is full: at 16384 points and radius 0.3 most balls never fill (9 hits expected) and a centre walks the whole cloud — 4.3e9 pair
// tests for level 0 of the stress frames. Here a cloud's points are binned into cells of edge >= 1.001 r (grid_build_kernel, one
// workgroup per cloud: bounding box, cell histogram / prefix / scatter with LDS atomics — the order INSIDE a cell is arbitrary),
// and a wave tests only the points of its centre's 27 neighbouring cells. The contract "the first nsample hits in index order"
// is restored by a bitmap: every hit sets bit k of an N-bit LDS bitmap, which is then read in ascending order — the result
// does not depend on the order the candidates were met in, so it is bit-identical to the sweep's (same sqdist3, same strict <).
// ------------------------------------------------------------------------------------------
constexpr int GRID_MAX_CELLS = 8192;
struct GridHead { float ox, oy, oz, inv; int nx, ny, nz, pad; };          // per cloud, at the start of its workspace slice
__host__ __device__ inline size_t grid_slice_ints(int N) { return 8 + (size_t)GRID_MAX_CELLS + 1 + (size_t)N; }

__device__ __forceinline__ int grid_cell_coord(float v, float o, float inv, int n) {
    const float q = (v - o) * inv;
    int c = (int)q;                                      // NaN -> 0 by the comparisons below
    c = (q >= 0.f) ? c : 0;
    return c < n ? c : n - 1;
}

__global__ __launch_bounds__(1024) void grid_build_kernel(const float* __restrict__ xyz, int N, float radius, int32_t* __restrict__ ws) {
    __shared__ int hist[GRID_MAX_CELLS];
    __shared__ float red[6][16];
    __shared__ int wsum[16];
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const float* __restrict__ pts = xyz + (size_t)b * N * 3;
    int32_t* __restrict__ slice = ws + (size_t)b * grid_slice_ints(N);
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (int k = t; k < N; k += 1024)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = pts[3 * k + c];
            if (v == v && fabsf(v) < 1.0e30f) { lo[c] = fminf(lo[c], v); hi[c] = fmaxf(hi[c], v); }     // NaN / inf take no part
        }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float l = wave_min_f32(lo[c]), h = wave_max_f32(hi[c]);
        if (lane == 0) { red[c][wv] = l; red[3 + c][wv] = h; }
    }
    for (int k = t; k < GRID_MAX_CELLS; k += 1024) hist[k] = 0;
    __syncthreads();
    float ext[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float l = red[c][0], h = red[3 + c][0];
        for (int w = 1; w < 16; ++w) { l = fminf(l, red[c][w]); h = fmaxf(h, red[3 + c][w]); }
        if (!(h >= l)) { l = 0.f; h = 0.f; }             // no finite point
        lo[c] = l; ext[c] = h - l;
    }
    float cell = radius * 1.001f;
    if (!(cell > 0.f)) cell = 1.f;
    int nx, ny, nz;
    for (;;) {                                           // the same loop in every thread: grow the cells until the grid fits
        const float fx = ext[0] / cell + 1.f, fy = ext[1] / cell + 1.f, fz = ext[2] / cell + 1.f;
        if (fx * fy * fz <= (float)GRID_MAX_CELLS && fx < 4096.f && fy < 4096.f && fz < 4096.f) {
            nx = (int)fx; ny = (int)fy; nz = (int)fz;
            if ((long long)nx * ny * nz <= GRID_MAX_CELLS) break;
        }
        cell *= 1.25f;
    }
    const float inv = 1.f / cell;
    const int cells = nx * ny * nz;
    for (int k = t; k < N; k += 1024) {
        const int ci = (grid_cell_coord(pts[3 * k + 2], lo[2], inv, nz) * ny + grid_cell_coord(pts[3 * k + 1], lo[1], inv, ny)) * nx +
                       grid_cell_coord(pts[3 * k], lo[0], inv, nx);
        atomicAdd(&hist[ci], 1);
    }
    __syncthreads();
    // exclusive prefix over the cells: 8 consecutive cells per thread, wave scan, cross-wave offsets
    constexpr int PER = GRID_MAX_CELLS / 1024;
    int loc[PER], sum = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) { loc[i] = hist[t * PER + i]; sum += loc[i]; }
    int incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(incl, off, 64); if (lane >= off) incl += v; }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wv; ++w) base += wsum[w];
    int run = base + incl - sum;
    int32_t* __restrict__ cstart = slice + 8;
#pragma unroll
    for (int i = 0; i < PER; ++i) { hist[t * PER + i] = run; cstart[t * PER + i] = run; run += loc[i]; }     // hist = the scatter cursors
    if (t == 1023) cstart[GRID_MAX_CELLS] = run;
    if (t == 0) {
        GridHead h{lo[0], lo[1], lo[2], inv, nx, ny, nz, cells};
        *reinterpret_cast<GridHead*>(slice) = h;
    }
    __syncthreads();
    int32_t* __restrict__ cpts = slice + 8 + GRID_MAX_CELLS + 1;
    for (int k = t; k < N; k += 1024) {
        const int ci = (grid_cell_coord(pts[3 * k + 2], lo[2], inv, nz) * ny + grid_cell_coord(pts[3 * k + 1], lo[1], inv, ny)) * nx +
                       grid_cell_coord(pts[3 * k], lo[0], inv, nx);
        cpts[atomicAdd(&hist[ci], 1)] = k;
    }
}

// One wave per centre. centres: new_xyz (B,M,3) given (sel_mode 0), or taken from the cloud through sel / the first M points
// (sel_mode 1: also written to new_xyz_out / idx64 as centres_ball_query_kernel does).
__global__ __launch_bounds__(256) void ball_query_grid_kernel(const float* __restrict__ xyz, const float* __restrict__ centres,
                                                              const int32_t* __restrict__ sel, int sel_mode, int BM, int M, int N,
                                                              float r2, int ns, const int32_t* __restrict__ ws,
                                                              float* __restrict__ new_xyz_out, long long* __restrict__ idx64,
                                                              int32_t* __restrict__ idx_out) {
    extern __shared__ unsigned bq_bits[];                                 // [4 waves][words]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int centre = blockIdx.x * 4 + wv;
    if (centre >= BM) return;
    const int words = (N + 31) >> 5;
    unsigned* __restrict__ bits = bq_bits + (size_t)wv * words;
    const int b = centre / M;
    const float* __restrict__ pts = xyz + (size_t)b * N * 3;
    const int32_t* __restrict__ slice = ws + (size_t)b * grid_slice_ints(N);
    const GridHead g = *reinterpret_cast<const GridHead*>(slice);
    const int32_t* __restrict__ cstart = slice + 8;
    const int32_t* __restrict__ cpts = slice + 8 + GRID_MAX_CELLS + 1;
    float cx, cy, cz;
    if (sel_mode) {
        const int n = sel ? sel[centre] : centre - b * M;
        cx = pts[3 * n]; cy = pts[3 * n + 1]; cz = pts[3 * n + 2];
        if (lane == 0) {
            new_xyz_out[(size_t)centre * 3] = cx; new_xyz_out[(size_t)centre * 3 + 1] = cy; new_xyz_out[(size_t)centre * 3 + 2] = cz;
            if (idx64) idx64[centre] = n;
        }
    } else {
        cx = centres[(size_t)centre * 3]; cy = centres[(size_t)centre * 3 + 1]; cz = centres[(size_t)centre * 3 + 2];
    }
    for (int w = lane; w < words; w += 64) bits[w] = 0u;
    __builtin_amdgcn_wave_barrier();
    // the centre's cell by the same arithmetic that binned the points, then its 27 neighbours (clipped). A centre outside the
    // bounding box (a ball query with foreign centres) is clamped to the border cell: its ball can only reach border cells' points
    // if it is within one cell of the box, which the clamped cell's neighbourhood covers... unless it is farther away than a cell —
    // then no point can be within r of it anyway.
    const int ix = grid_cell_coord(cx, g.ox, g.inv, g.nx), iy = grid_cell_coord(cy, g.oy, g.inv, g.ny), iz = grid_cell_coord(cz, g.oz, g.inv, g.nz);
    for (int dz = -1; dz <= 1; ++dz) {
        const int z = iz + dz;
        if (z < 0 || z >= g.nz) continue;
        for (int dy = -1; dy <= 1; ++dy) {
            const int y = iy + dy;
            if (y < 0 || y >= g.ny) continue;
            // the (up to) three cells along x are consecutive in the cell order: one contiguous candidate range
            const int x0 = ix > 0 ? ix - 1 : 0, x1 = ix + 1 < g.nx ? ix + 1 : g.nx - 1;
            const int c0 = (z * g.ny + y) * g.nx + x0;
            const int s0 = cstart[c0], s1 = cstart[c0 + (x1 - x0) + 1];
            for (int e = s0 + lane; e < s1; e += 64) {
                const int k = cpts[e];
                const float d = sqdist3(cx, cy, cz, pts[3 * k], pts[3 * k + 1], pts[3 * k + 2]);
                if (d < r2) atomicOr(&bits[k >> 5], 1u << (k & 31));
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the LDS atomics of this wave have landed
    // read the bitmap in index order: 64 words (2048 indices) per round, prefix of the popcounts across the wave
    int32_t* __restrict__ out = idx_out + (size_t)centre * ns;
    int cnt = 0, first = 0;
    for (int w0 = 0; w0 < words && cnt < ns; w0 += 64) {
        const int w = w0 + lane;
        unsigned v = w < words ? bits[w] : 0u;
        const int mine = __popc(v);
        int incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int u = __shfl_up(incl, off, 64); if (lane >= off) incl += u; }
        const int total = __shfl(incl, 63, 64);
        if (total == 0) continue;
        if (cnt == 0) {
            const unsigned long long any = __ballot(mine > 0);
            const int fl = __ffsll((long long)any) - 1;
            const unsigned fv = __shfl(v, fl, 64);
            first = ((w0 + fl) << 5) + (__ffs(fv) - 1);
        }
        int pos = cnt + incl - mine;
        while (v != 0u && pos < ns) {
            const int bit = __ffs(v) - 1;
            out[pos++] = (w << 5) + bit;
            v &= v - 1u;
        }
        cnt += total;
    }
    const int have = cnt < ns ? cnt : ns;
    const int fill = cnt > 0 ? first : 0;
    for (int s2 = have + lane; s2 < ns; s2 += 64) out[s2] = fill;
}

}  // namespace ptt

using namespace ptt;

extern "C" int ptt_fps_f32(const float* xyz, int B, int N, int npoint, int32_t* idx_out, ptt_stream_t stream) {
    if (B < 0 || N <= 0 || npoint < 0) return fail(PTT_EINVAL, "ptt_fps_f32: B=%d N=%d npoint=%d", B, N, npoint);
    if (B == 0 || npoint == 0) return PTT_OK;
    if (!xyz || !idx_out) return fail(PTT_EINVAL, "ptt_fps_f32: null pointer");
    if (npoint > 15360) return fail(PTT_EUNSUPPORTED, "ptt_fps_f32: npoint=%d exceeds the LDS index buffer (15360)", npoint);
    hipStream_t s = as_stream(stream);
    if (N <= 64) return launch_fps<64, 1>(xyz, B, N, npoint, idx_out, s);
    if (N <= 128) return launch_fps<64, 2>(xyz, B, N, npoint, idx_out, s);
    if (N <= 256) return launch_fps<64, 4>(xyz, B, N, npoint, idx_out, s);
    if (N <= 512) {
        if (const int T = dev_switches().fps_t) {       // dev: workgroup-size sweep
            if (T == 256) return launch_fps<256, 2>(xyz, B, N, npoint, idx_out, s);
            if (T == 128) return launch_fps<128, 4>(xyz, B, N, npoint, idx_out, s);
            if (T == 512) return launch_fps<512, 1>(xyz, B, N, npoint, idx_out, s);
        }
        return launch_fps<64, 8>(xyz, B, N, npoint, idx_out, s);     // one wave, no barrier: 0.36 us / iteration (256 threads: 0.42)
    }
    if (N <= 1024) {
        if (const int T = dev_switches().fps_t) {
            if (T == 64) return launch_fps<64, 16>(xyz, B, N, npoint, idx_out, s);
            if (T == 128) return launch_fps<128, 8>(xyz, B, N, npoint, idx_out, s);
            if (T == 512) return launch_fps<512, 2>(xyz, B, N, npoint, idx_out, s);
        }
        return launch_fps<256, 4>(xyz, B, N, npoint, idx_out, s);    // 0.37 us / iteration (512 threads: 0.39; scripts/fps_sweep.py)
    }
    if (N <= 2048) {
        if (const int T = dev_switches().fps_t) {       // dev: workgroup-size sweep
            if (T == 512) return launch_fps<512, 4>(xyz, B, N, npoint, idx_out, s);
            if (T == 1024) return launch_fps<1024, 2>(xyz, B, N, npoint, idx_out, s);
            if (T == 128) return launch_fps<128, 16>(xyz, B, N, npoint, idx_out, s);
        }
        return launch_fps<256, 8>(xyz, B, N, npoint, idx_out, s);   // 0.45 us / iteration; 512 threads 0.46, 128 / 1024: 0.52
    }
    if (N <= 4096) return launch_fps<512, 8>(xyz, B, N, npoint, idx_out, s);
    if (N <= 8192) return launch_fps<1024, 8>(xyz, B, N, npoint, idx_out, s);
    if (N <= 16384) return launch_fps<1024, 16>(xyz, B, N, npoint, idx_out, s);
    // 32 points per thread (clouds up to 32768 points) is the whole 128-register budget of a 1024-thread workgroup and spilled
    // 164-180 bytes per lane: no BASELINE config reaches it (the largest is 16384); such clouds take ptt_fps_ws_f32
    return fail(PTT_EUNSUPPORTED, "ptt_fps_f32: N=%d exceeds the register-resident limit 16384 (use ptt_fps_ws_f32)", N);
}

extern "C" int ptt_spatial_order_f32(const float* xyz, int B, int N, int32_t* order, ptt_stream_t stream) {
    if (B < 0 || N <= 0) return fail(PTT_EINVAL, "ptt_spatial_order_f32: B=%d N=%d", B, N);
    if (B == 0) return PTT_OK;
    if (!xyz || !order) return fail(PTT_EINVAL, "ptt_spatial_order_f32: null pointer");
    if (N > 8192 || (long long)B * N > INT_MAX) return fail(PTT_EUNSUPPORTED, "ptt_spatial_order_f32: N=%d (at most 8192 points per cloud are sorted in LDS)", N);
    int P = 2;
    while (P < N) P <<= 1;
    const int lds = P * (int)sizeof(unsigned long long);
    if (int rc = set_lds_limit(reinterpret_cast<const void*>(spatial_order_kernel), lds)) return rc;
    hipLaunchKernelGGL(spatial_order_kernel, dim3(B), dim3(1024), lds, as_stream(stream), xyz, N, P, order);
    return check_launch("spatial_order_kernel");
}

extern "C" int ptt_fps_ws_f32(const float* xyz, int B, int N, int npoint, int32_t* idx_out, float* workspace, size_t workspace_elems,
                              ptt_stream_t stream) {
    if (B < 0 || N <= 0 || npoint < 0) return fail(PTT_EINVAL, "ptt_fps_ws_f32: B=%d N=%d npoint=%d", B, N, npoint);
    if (B == 0 || npoint == 0) return PTT_OK;
    if (!xyz || !idx_out) return fail(PTT_EINVAL, "ptt_fps_ws_f32: null pointer");
    if ((long long)N * 3 > INT_MAX) return fail(PTT_EUNSUPPORTED, "ptt_fps_ws_f32: N=%d", N);
    if (!workspace || workspace_elems < (size_t)B * N)
        return fail(PTT_EWORKSPACE, "ptt_fps_ws_f32: %zu floats of workspace needed", (size_t)B * N);
    hipLaunchKernelGGL(fps_ws_kernel, dim3(B), dim3(1024), 0, as_stream(stream), xyz, N, npoint, workspace, idx_out);
    return check_launch("fps_ws_kernel");
}

// centres per wave of the ball-query kernels: 4 when the launch still has at least two waves per SIMD of the device and
// a wave's centres stay inside one cloud, else 1 (8: dev switch only)
static int ball_cpw(int BM, int M) {
    const int force = dev_switches().ball_cpw;
    if (force == 1 || (M & 3) != 0) return 1;
    if (force == 8 && (M & 7) == 0) return 8;
    if (force == 4) return 4;
    return BM >= 4 * 256 * 8 ? 4 : 1;        // 8 per wave measured slower than 4 on the 16384-point frames (2.72 vs 2.49 ms)
}

extern "C" int ptt_ball_query_f32(const float* new_xyz, const float* xyz, int B, int M, int N, float radius,
                                  int nsample, int32_t* idx_out, ptt_stream_t stream) {
    if (B < 0 || M < 0 || N <= 0 || nsample <= 0)
        return fail(PTT_EINVAL, "ptt_ball_query_f32: B=%d M=%d N=%d nsample=%d", B, M, N, nsample);
    if (B == 0 || M == 0) return PTT_OK;
    if (!new_xyz || !xyz || !idx_out) return fail(PTT_EINVAL, "ptt_ball_query_f32: null pointer");
    const float r2 = radius * radius;
    const int BM = B * M;
    const int cpw = ball_cpw(BM, M);
    if (cpw == 8)
        hipLaunchKernelGGL((ball_query_kernel<8>), dim3((BM / 8 + 3) / 4), dim3(256), 0, as_stream(stream), new_xyz, xyz, BM, M,
                           N, r2, nsample, idx_out);
    else if (cpw == 4)
        hipLaunchKernelGGL((ball_query_kernel<4>), dim3((BM / 4 + 3) / 4), dim3(256), 0, as_stream(stream), new_xyz, xyz, BM, M,
                           N, r2, nsample, idx_out);
    else
        hipLaunchKernelGGL((ball_query_kernel<1>), dim3((BM + 3) / 4), dim3(256), 0, as_stream(stream), new_xyz, xyz, BM, M,
                           N, r2, nsample, idx_out);
    return check_launch("ball_query_kernel");
}

extern "C" int ptt_gather_f32(const float* feat, const int32_t* idx, int B, int C, int N, int M, float* out,
                              ptt_stream_t stream) {
    if (B < 0 || C < 0 || N <= 0 || M < 0) return fail(PTT_EINVAL, "ptt_gather_f32: bad sizes");
    const size_t total = (size_t)B * C * M;
    if (total == 0) return PTT_OK;
    if (!feat || !idx || !out) return fail(PTT_EINVAL, "ptt_gather_f32: null pointer");
    hipLaunchKernelGGL(gather_kernel, dim3(grid_for(total, 256)), dim3(256), 0, as_stream(stream), feat, idx, C, N, M,
                       out, total);
    return check_launch("gather_kernel");
}

extern "C" int ptt_select_centres_f32(const float* xyz, const int32_t* idx, int B, int N, int M, float* new_xyz,
                                      int64_t* idx64_out, ptt_stream_t stream) {
    if (B < 0 || N <= 0 || M < 0 || (!idx && M > N)) return fail(PTT_EINVAL, "ptt_select_centres_f32: B=%d N=%d M=%d", B, N, M);
    const size_t total = (size_t)B * M;
    if (total == 0) return PTT_OK;
    if (!xyz || !new_xyz) return fail(PTT_EINVAL, "ptt_select_centres_f32: null pointer");
    hipLaunchKernelGGL(select_centres_kernel, dim3(grid_for(total, 256)), dim3(256), 0, as_stream(stream), xyz, idx, N, M,
                       new_xyz, reinterpret_cast<long long*>(idx64_out), total);
    return check_launch("select_centres_kernel");
}

extern "C" int ptt_gather_grad_f32(const float* grad_out, const int32_t* idx, int B, int C, int N, int M,
                                   float* grad_feat, ptt_stream_t stream) {
    if (B < 0 || C < 0 || N <= 0 || M < 0) return fail(PTT_EINVAL, "ptt_gather_grad_f32: bad sizes");
    const size_t nout = (size_t)B * C * N;
    if (nout == 0) return PTT_OK;
    if (!grad_feat) return fail(PTT_EINVAL, "ptt_gather_grad_f32: null pointer");
    if (hipMemsetAsync(grad_feat, 0, nout * sizeof(float), as_stream(stream)) != hipSuccess)
        return check_launch("gather_grad memset");
    const size_t total = (size_t)B * C * M;
    if (total == 0) return PTT_OK;
    if (!grad_out || !idx) return fail(PTT_EINVAL, "ptt_gather_grad_f32: null pointer");
    hipLaunchKernelGGL(gather_grad_kernel, dim3(grid_for(total, 256)), dim3(256), 0, as_stream(stream), grad_out, idx,
                       C, N, M, grad_feat, total);
    return check_launch("gather_grad_kernel");
}

extern "C" int ptt_group_f32(const float* feat, const int32_t* idx, int B, int C, int N, int M, int ns, float* out,
                             ptt_stream_t stream) {
    if (B < 0 || C < 0 || N <= 0 || M < 0 || ns < 0) return fail(PTT_EINVAL, "ptt_group_f32: bad sizes");
    const size_t total = (size_t)B * C * M * ns;
    if (total == 0) return PTT_OK;
    if (!feat || !idx || !out) return fail(PTT_EINVAL, "ptt_group_f32: null pointer");
    hipLaunchKernelGGL(group_kernel, dim3(grid_for(total, 256)), dim3(256), 0, as_stream(stream), feat, idx, C, N, M,
                       ns, out, total);
    return check_launch("group_kernel");
}

extern "C" int ptt_centres_ball_query_f32(const float* xyz, const int32_t* sel, int B, int N, int M, float radius, int nsample,
                                          float* new_xyz, int64_t* idx64_out, int32_t* idx_out, ptt_stream_t stream) {
    if (B < 0 || M < 0 || N <= 0 || nsample <= 0 || (!sel && M > N))
        return fail(PTT_EINVAL, "ptt_centres_ball_query_f32: B=%d M=%d N=%d nsample=%d", B, M, N, nsample);
    if (B == 0 || M == 0) return PTT_OK;
    if (!xyz || !new_xyz || !idx_out) return fail(PTT_EINVAL, "ptt_centres_ball_query_f32: null pointer");
    const int BM = B * M;
    const int cpw = ball_cpw(BM, M);
    if (cpw == 8)
        hipLaunchKernelGGL((centres_ball_query_kernel<8>), dim3((BM / 8 + 3) / 4), dim3(256), 0, as_stream(stream), xyz, sel, BM,
                           M, N, radius * radius, nsample, new_xyz, reinterpret_cast<long long*>(idx64_out), idx_out);
    else if (cpw == 4)
        hipLaunchKernelGGL((centres_ball_query_kernel<4>), dim3((BM / 4 + 3) / 4), dim3(256), 0, as_stream(stream), xyz, sel, BM,
                           M, N, radius * radius, nsample, new_xyz, reinterpret_cast<long long*>(idx64_out), idx_out);
    else
        hipLaunchKernelGGL((centres_ball_query_kernel<1>), dim3((BM + 3) / 4), dim3(256), 0, as_stream(stream), xyz, sel, BM, M,
                           N, radius * radius, nsample, new_xyz, reinterpret_cast<long long*>(idx64_out), idx_out);
    return check_launch("centres_ball_query_kernel");
}


extern "C" size_t ptt_ball_query_grid_workspace(int B, int N) {
    if (B <= 0 || N <= 0) return 0;
    return (size_t)B * grid_slice_ints(N) * sizeof(int32_t);
}

static int ball_query_grid_launch(const char* who, const float* xyz, const float* centres, const int32_t* sel, int sel_mode, int B, int N, int M,
                                  float radius, int nsample, float* new_xyz, int64_t* idx64, int32_t* idx_out, void* ws, size_t ws_bytes,
                                  ptt_stream_t stream) {
    if (B < 0 || M < 0 || N <= 0 || nsample <= 0) return fail(PTT_EINVAL, "%s: B=%d M=%d N=%d nsample=%d", who, B, M, N, nsample);
    if (B == 0 || M == 0) return PTT_OK;
    if (N > 131072) return fail(PTT_EUNSUPPORTED, "%s: N=%d (the hit bitmap of a wave lives in LDS: at most 131072 points)", who, N);
    if (!ws || ws_bytes < ptt_ball_query_grid_workspace(B, N) || (reinterpret_cast<uintptr_t>(ws) & 15))
        return fail(PTT_EWORKSPACE, "%s: %zu bytes of 16-byte aligned workspace needed", who, ptt_ball_query_grid_workspace(B, N));
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(grid_build_kernel, dim3(B), dim3(1024), 0, s, xyz, N, radius, static_cast<int32_t*>(ws));
    const int BM = B * M, lds = 4 * ((N + 31) / 32) * (int)sizeof(unsigned);
    if (int rc = set_lds_limit(reinterpret_cast<const void*>(ball_query_grid_kernel), lds)) return rc;
    hipLaunchKernelGGL(ball_query_grid_kernel, dim3((BM + 3) / 4), dim3(256), lds, s, xyz, centres, sel, sel_mode, BM, M, N, radius * radius,
                       nsample, static_cast<const int32_t*>(ws), new_xyz, reinterpret_cast<long long*>(idx64), idx_out);
    return check_launch(who);
}

extern "C" int ptt_ball_query_grid_f32(const float* new_xyz, const float* xyz, int B, int M, int N, float radius, int nsample,
                                       int32_t* idx_out, void* workspace, size_t workspace_bytes, ptt_stream_t stream) {
    if ((B > 0 && M > 0) && (!new_xyz || !xyz || !idx_out)) return fail(PTT_EINVAL, "ptt_ball_query_grid_f32: null pointer");
    return ball_query_grid_launch("ptt_ball_query_grid_f32", xyz, new_xyz, nullptr, 0, B, N, M, radius, nsample, nullptr, nullptr, idx_out,
                                  workspace, workspace_bytes, stream);
}

extern "C" int ptt_centres_ball_query_grid_f32(const float* xyz, const int32_t* sel, int B, int N, int M, float radius, int nsample,
                                               float* new_xyz, int64_t* idx64_out, int32_t* idx_out, void* workspace,
                                               size_t workspace_bytes, ptt_stream_t stream) {
    if (!sel && M > N) return fail(PTT_EINVAL, "ptt_centres_ball_query_grid_f32: M=%d > N=%d without a selection", M, N);
    if ((B > 0 && M > 0) && (!xyz || !new_xyz || !idx_out)) return fail(PTT_EINVAL, "ptt_centres_ball_query_grid_f32: null pointer");
    return ball_query_grid_launch("ptt_centres_ball_query_grid_f32", xyz, nullptr, sel, 1, B, N, M, radius, nsample, new_xyz, idx64_out,
                                  idx_out, workspace, workspace_bytes, stream);
}

extern "C" int ptt_fps_ball_knn_f32(const float* xyz, int B, int N, int M, float radius, int nsample, int k, int32_t* inds,
                                    int64_t* inds64, float* new_xyz, int32_t* idx, int32_t* knn, float* rel, ptt_stream_t stream) {
    if (B < 0 || N <= 0 || M <= 0 || nsample <= 0 || k < 0) return fail(PTT_EINVAL, "ptt_fps_ball_knn_f32: B=%d N=%d M=%d nsample=%d k=%d", B, N, M, nsample, k);
    if (N > 256 || M > 128 || M > N || k > M)
        return fail(PTT_EUNSUPPORTED, "ptt_fps_ball_knn_f32: at most 256 points, 128 centres, k <= centres (N=%d M=%d k=%d)", N, M, k);
    if (B == 0) return PTT_OK;
    if (!xyz || !inds || !new_xyz || !idx || (k > 0 && !knn)) return fail(PTT_EINVAL, "ptt_fps_ball_knn_f32: null pointer");
    FbkParams p;
    p.xyz = xyz; p.B = B; p.N = N; p.M = M; p.ns = nsample; p.k = k; p.r2 = radius * radius;
    p.inds = inds; p.inds64 = reinterpret_cast<long long*>(inds64); p.new_xyz = new_xyz; p.idx = idx; p.knn = k > 0 ? knn : nullptr; p.rel = rel;
    hipLaunchKernelGGL(fps_ball_knn_kernel, dim3(B * ((M + 3) / 4)), dim3(256), 0, as_stream(stream), p);
    return check_launch("fps_ball_knn_kernel");
}

extern "C" int ptt_point_jobs_f32(const ptt_point_job* jobs, int n_jobs, ptt_stream_t stream) {
    if (!jobs || n_jobs < 1 || n_jobs > PTT_POINT_JOBS_MAX) return fail(PTT_EINVAL, "ptt_point_jobs_f32: n_jobs=%d (1..%d)", n_jobs, PTT_POINT_JOBS_MAX);
    PointJobs P;
    P.n = 0;
    int waves = 0;
    for (int i = 0; i < n_jobs; ++i) {
        const ptt_point_job& j = jobs[i];
        if (j.B < 0 || j.Nraw <= 0 || j.Npts <= 0 || j.M < 0 || j.nsample <= 0 || (j.kind != 0 && j.kind != 1))
            return fail(PTT_EINVAL, "ptt_point_jobs_f32: job %d: kind=%d B=%d Nraw=%d Npts=%d M=%d nsample=%d", i, j.kind, j.B, j.Nraw, j.Npts, j.M, j.nsample);
        if (j.B == 0 || j.M == 0) continue;
        if (!j.xyz || !j.idx_out || (j.kind == 0 && !j.new_xyz)) return fail(PTT_EINVAL, "ptt_point_jobs_f32: job %d: null pointer", i);
        if ((j.centre_sel || j.point_sel) && (j.sel_ld < j.M || (j.point_sel && j.sel_ld < j.Npts)))
            return fail(PTT_EINVAL, "ptt_point_jobs_f32: job %d: sel_ld=%d", i, j.sel_ld);
        if ((!j.centre_sel && j.M > j.Nraw) || (!j.point_sel && j.Npts > j.Nraw))
            return fail(PTT_EINVAL, "ptt_point_jobs_f32: job %d: identity selection past the cloud", i);
        if (j.kind == 1 && (j.Npts > 128 || j.M != j.Npts || j.nsample > j.Npts))
            return fail(PTT_EUNSUPPORTED, "ptt_point_jobs_f32: job %d: the kNN job takes at most 128 points, every point a query (Npts=%d M=%d k=%d)", i, j.Npts, j.M, j.nsample);
        PointJobDev& D = P.j[P.n++];
        D.xyz = j.xyz; D.csel = j.centre_sel; D.psel = j.point_sel; D.new_xyz = j.new_xyz;
        D.idx64 = reinterpret_cast<long long*>(j.idx64_out); D.idx_out = j.idx_out; D.rel_out = j.rel_out;
        D.kind = j.kind; D.sel_ld = j.sel_ld; D.B = j.B; D.Nraw = j.Nraw; D.Npts = j.Npts; D.M = j.M; D.ns = j.nsample;
        D.r2 = j.radius * j.radius;
        D.wave0 = waves;
        waves += (j.B * j.M + 3) / 4 * 4;
    }
    if (P.n == 0) return PTT_OK;
    hipLaunchKernelGGL(point_jobs_kernel, dim3(waves / 4), dim3(256), 0, as_stream(stream), P);
    return check_launch("point_jobs_kernel");
}

extern "C" int ptt_group_grad_f32(const float* grad_out, const int32_t* idx, int B, int C, int N, int M, int ns,
                                  float* grad_feat, ptt_stream_t stream) {
    if (B < 0 || C < 0 || N <= 0 || M < 0 || ns < 0) return fail(PTT_EINVAL, "ptt_group_grad_f32: bad sizes");
    const size_t nout = (size_t)B * C * N;
    if (nout == 0) return PTT_OK;
    if (!grad_feat) return fail(PTT_EINVAL, "ptt_group_grad_f32: null pointer");
    const size_t total = (size_t)B * C * M * ns;
    if (total > 0 && (!grad_out || !idx)) return fail(PTT_EINVAL, "ptt_group_grad_f32: null pointer");
    if (total > 0 && N <= 16384 && !dev_switches().group_grad_global) {       // accumulate in LDS (<= 64 KB per workgroup)
        int CC = 65536 / (4 * N);
        if (CC > 16) CC = 16;
        if (CC > C) CC = C;
        const int nchunks = (C + CC - 1) / CC;
        hipLaunchKernelGGL(group_grad_lds_kernel, dim3(B * nchunks), dim3(512), (size_t)CC * N * sizeof(float),
                           as_stream(stream), grad_out, idx, C, N, M, ns, CC, nchunks, grad_feat);
        return check_launch("group_grad_lds_kernel");
    }
    if (hipMemsetAsync(grad_feat, 0, nout * sizeof(float), as_stream(stream)) != hipSuccess)
        return check_launch("group_grad memset");
    if (total == 0) return PTT_OK;
    hipLaunchKernelGGL(group_grad_kernel, dim3(grid_for(total, 256)), dim3(256), 0, as_stream(stream), grad_out, idx,
                       C, N, M, ns, grad_feat, total);
    return check_launch("group_grad_kernel");
}

extern "C" size_t ptt_scatter_add_det_workspace(int B, int N, int E) {
    if (B <= 0 || N <= 0 || E <= 0) return 0;
    return ((size_t)B * E + (size_t)B * (N + 1)) * sizeof(int32_t);
}

extern "C" int ptt_scatter_add_det_f32(const float* src, const int32_t* idx, int B, int C, int N, int E, float* out,
                                       void* workspace, size_t workspace_bytes, ptt_stream_t stream) {
    if (B < 0 || C < 0 || N <= 0 || E < 0) return fail(PTT_EINVAL, "ptt_scatter_add_det_f32: bad sizes");
    const size_t nout = (size_t)B * C * N;
    if (nout == 0) return PTT_OK;
    if (!out) return fail(PTT_EINVAL, "ptt_scatter_add_det_f32: null pointer");
    hipStream_t s = as_stream(stream);
    if (E == 0) {
        if (hipMemsetAsync(out, 0, nout * sizeof(float), s) != hipSuccess) return check_launch("scatter_add_det memset");
        return PTT_OK;
    }
    if (!src || !idx) return fail(PTT_EINVAL, "ptt_scatter_add_det_f32: null pointer");
    if (E > 16384 || (unsigned long long)N * 16384ull > 0xffffffffull)
        return fail(PTT_EUNSUPPORTED, "ptt_scatter_add_det_f32: E=%d N=%d (at most 16384 entries per cloud)", E, N);
    const size_t need = ptt_scatter_add_det_workspace(B, N, E);
    if (!workspace || workspace_bytes < need)
        return fail(PTT_EWORKSPACE, "ptt_scatter_add_det_f32: workspace of %zu bytes, %zu needed", workspace_bytes, need);
    int32_t* order = static_cast<int32_t*>(workspace);
    int32_t* start = order + (size_t)B * E;
    int rc = launch_scatter_csr(idx, B, N, E, order, start, s);
    if (rc) return rc;
    const int CC = 8, nchunks = (C + CC - 1) / CC;
    const size_t lds = (size_t)E * (sizeof(float) + sizeof(int));
    if ((rc = set_lds_limit(reinterpret_cast<const void*>(scatter_add_det_kernel), (int)lds))) return rc;
    hipLaunchKernelGGL(scatter_add_det_kernel, dim3(B * nchunks), dim3(512), lds, s, src, order, start, C, N, E, CC, nchunks,
                       out);
    return check_launch("scatter_add_det_kernel");
}

extern "C" int ptt_scatter_csr_i32(const int32_t* idx, int B, int N, int E, int32_t* order, int32_t* start, ptt_stream_t stream) {
    if (B < 0 || N <= 0 || E <= 0) return fail(PTT_EINVAL, "ptt_scatter_csr_i32: B=%d N=%d E=%d", B, N, E);
    if (B == 0) return PTT_OK;
    if (!idx || !order || !start) return fail(PTT_EINVAL, "ptt_scatter_csr_i32: null pointer");
    // the counting sort (N <= 2048 bins) has no limit on the entries; the bitonic network sorts 64 KB of keys in LDS
    if (N > 2048 && (E > 16384 || (unsigned long long)N * 16384ull > 0xffffffffull))
        return fail(PTT_EUNSUPPORTED, "ptt_scatter_csr_i32: E=%d N=%d (more than 2048 bins: at most 16384 entries per cloud)", E, N);
    return launch_scatter_csr(idx, B, N, E, order, start, as_stream(stream));
}

extern "C" int ptt_knn_rel_f32(const float* xyz, int B, int N, int k, int32_t* idx_out, float* rel_out,
                               ptt_stream_t stream);

extern "C" int ptt_knn_f32(const float* xyz, int B, int N, int k, int32_t* idx_out, ptt_stream_t stream) {
    return ptt_knn_rel_f32(xyz, B, N, k, idx_out, nullptr, stream);
}

extern "C" int ptt_knn_rel_f32(const float* xyz, int B, int N, int k, int32_t* idx_out, float* rel_out,
                               ptt_stream_t stream) {
    if (B < 0 || N <= 0 || k <= 0 || k > N) return fail(PTT_EINVAL, "ptt_knn_f32: B=%d N=%d k=%d", B, N, k);
    if (B == 0) return PTT_OK;
    if (!xyz || !idx_out) return fail(PTT_EINVAL, "ptt_knn_f32: null pointer");
    const int BN = B * N;
    const dim3 grid((BN + 3) / 4), block(256);
    hipStream_t s = as_stream(stream);
#define PTT_KNN_CASE(P)                                                                          \
    if (N <= 64 * P) {                                                                           \
        hipLaunchKernelGGL((knn_kernel<P>), grid, block, 0, s, xyz, BN, N, k, idx_out, rel_out); \
        return check_launch("knn_kernel");                                                       \
    }
    PTT_KNN_CASE(1) PTT_KNN_CASE(2) PTT_KNN_CASE(4) PTT_KNN_CASE(8) PTT_KNN_CASE(16) PTT_KNN_CASE(32) PTT_KNN_CASE(64)
#undef PTT_KNN_CASE
    return fail(PTT_EUNSUPPORTED, "ptt_knn_f32: N=%d exceeds 4096", N);
}
