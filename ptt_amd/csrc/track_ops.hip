// N4: the per-frame pre/post-processing of the reference's sequential tracking loop
// (tools/eval_utils/eval_tracking_utils.py:140-229,266-274) on the device, so that a tracklet's clouds stay resident
// in HBM and only a 4-dof box crosses PCIe per frame:
//   crop_compact_kernel  crop_center_pc (ptt/datasets/kitti/kitti_tracking_utils.py:300-339) = crop_pc in the cloud's
//                        frame -> translate -> rotate -> crop_pc in the box frame, with a STABLE compaction (the
//                        resampling below indexes the surviving points in their original order);
//   regularize_kernel    regularize_pc (:342-367): n > 2 surviving points are resampled WITH replacement to a fixed
//                        size by np.random.randint after set_manual_seed(1) — i.e. MT19937(seed 1) 32-bit outputs,
//                        masked to the smallest 2^k - 1 >= n - 1 and rejected while > n - 1; n <= 2 gives an all-zero
//                        cloud; n == size is copied through. Template clouds are the concatenation of several crops
//                        (get_model, :219-236).
//   select_box_kernel    post_process (:266-274): first arg-max of the proposal scores, its 4-dof offset and score.
// Arithmetic follows numpy's: points are float32, box quantities float64; `translate` rounds (double)p + t to float32,
// `rotate` rounds the float64 dot product to float32; comparisons are float32 point against float64 bound, strict.
#include "common.h"

namespace ptt {

// ---- MT19937 (Matsumoto & Nishimura 1998; init_genrand / genrand_int32) on the host: the draw table ----
static void mt19937_fill(uint32_t seed, uint32_t* out, int n) {
    uint32_t mt[624];
    mt[0] = seed;
    for (int i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
    int idx = 624;
    for (int o = 0; o < n; ++o) {
        if (idx >= 624) {
            for (int k = 0; k < 624; ++k) {
                const uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
                mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
            idx = 0;
        }
        uint32_t y = mt[idx++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        out[o] = y;
    }
}

// exclusive prefix of `flag` over the workgroup (T threads, T/64 waves) + the workgroup total; `wsum` is LDS [T/64]
template <int T>
__device__ __forceinline__ int block_rank(bool flag, int* wsum, int& total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const unsigned long long m = __ballot(flag);
    const int below = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[w] = __popcll(m);
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < T / 64; ++i) {
        const int s = wsum[i];
        if (i < w) base += s;
        tot += s;
    }
    __syncthreads();
    total = tot;
    return base + below;
}

// One workgroup per crop job: a stable stream compaction over the cloud in chunks of T points.
template <int T>
__global__ __launch_bounds__(T) void crop_compact_kernel(const ptt_crop_job* __restrict__ jobs) {
    __shared__ int wsum[T / 64];
    const ptt_crop_job j = jobs[blockIdx.x];
    const float* px = j.points;
    const float* py = j.points + j.ld;
    const float* pz = j.points + 2 * j.ld;
    int written = 0;
    for (int base = 0; base < j.n_points; base += T) {
        const int i = base + (int)threadIdx.x;
        bool keep = false;
        float ox = 0.f, oy = 0.f, oz = 0.f;
        if (i < j.n_points) {
            const float x = px[i], y = py[i], z = pz[i];
            // crop_pc in the cloud's frame: float32 point against float64 bounds, strict on both sides
            keep = (double)x > j.lo1[0] && (double)x < j.hi1[0] && (double)y > j.lo1[1] && (double)y < j.hi1[1] &&
                   (double)z > j.lo1[2] && (double)z < j.hi1[2];
            if (keep) {
                // PointCloud.translate: points[i,:] = points[i,:] + x[i] (float64 sum stored to the float32 array)
                const double tx = (double)(float)((double)x + j.trans[0]);
                const double ty = (double)(float)((double)y + j.trans[1]);
                const double tz = (double)(float)((double)z + j.trans[2]);
                // PointCloud.rotate: np.dot(rot (f64), points) stored to float32
                ox = (float)fma(j.rot[2], tz, fma(j.rot[1], ty, j.rot[0] * tx));
                oy = (float)fma(j.rot[5], tz, fma(j.rot[4], ty, j.rot[3] * tx));
                oz = (float)fma(j.rot[8], tz, fma(j.rot[7], ty, j.rot[6] * tx));
                keep = (double)ox > j.lo2[0] && (double)ox < j.hi2[0] && (double)oy > j.lo2[1] && (double)oy < j.hi2[1] &&
                       (double)oz > j.lo2[2] && (double)oz < j.hi2[2];
            }
        }
        int total;
        const int r = written + block_rank<T>(keep, wsum, total);
        if (keep && r < j.capacity) {
            j.out[(size_t)r * 3 + 0] = ox;
            j.out[(size_t)r * 3 + 1] = oy;
            j.out[(size_t)r * 3 + 2] = oz;
        }
        written += total;
    }
    if (threadIdx.x == 0) *j.count = written;       // may exceed capacity: the caller sized `out` for n_points
}

__device__ __forceinline__ void seg_point(const ptt_regularize_job& j, const int (&cnt)[PTT_MAX_SEGMENTS], int idx, float* dst) {
    int s = 0;
    while (s + 1 < j.n_seg && idx >= cnt[s]) { idx -= cnt[s]; ++s; }
    const float* p = j.seg[s] + (size_t)idx * 3;
    dst[0] = p[0]; dst[1] = p[1]; dst[2] = p[2];
}

// One workgroup per output cloud.
template <int T>
__global__ __launch_bounds__(T) void regularize_kernel(const ptt_regularize_job* __restrict__ jobs,
                                                       const uint32_t* __restrict__ draws, int n_draws) {
    __shared__ int wsum[T / 64];
    const ptt_regularize_job j = jobs[blockIdx.x];
    int cnt[PTT_MAX_SEGMENTS];
    int n = 0;
#pragma unroll
    for (int s = 0; s < PTT_MAX_SEGMENTS; ++s) {
        cnt[s] = (s < j.n_seg) ? min(*j.seg_count[s], j.seg_capacity[s]) : 0;
        n += cnt[s];
    }
    const int size = j.input_size;
    int used = 0;
    if (n <= 2) {                                   // regularize_pc:359-362: an (almost) empty crop is an all-zero cloud
        for (int i = threadIdx.x; i < size * 3; i += T) j.out[i] = 0.f;
    } else if (n == size) {                         // :348 the cloud already has the right size: no resampling
        for (int i = threadIdx.x; i < size; i += T) seg_point(j, cnt, i, j.out + (size_t)i * 3);
    } else {
        // np.random.randint(0, n, size) on a freshly seeded MT19937: masked rejection sampling of 32-bit outputs
        const uint32_t rng = (uint32_t)(n - 1);
        uint32_t mask = rng;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
        int filled = 0;
        for (int base = 0; base < n_draws && filled < size; base += T) {
            const int d = base + (int)threadIdx.x;
            uint32_t v = 0;
            bool ok = false;
            if (d < n_draws) { v = draws[d] & mask; ok = v <= rng; }
            int total;
            const int r = filled + block_rank<T>(ok, wsum, total);
            if (ok && r < size) seg_point(j, cnt, (int)v, j.out + (size_t)r * 3);
            if (filled + total >= size) {
                // the draw that filled the last slot: how far numpy's generator has advanced (for the host's mirror
                // of the global RNG state, get_box_by_offset:205-208 draws from it)
                if (ok && r == size - 1) used = d + 1;
            }
            filled += total;
        }
        if (filled < size) {                        // draw table too short (never with n_draws >= 4 * size + 1024)
            for (int i = filled + threadIdx.x; i < size; i += T) {
                j.out[(size_t)i * 3] = j.out[(size_t)i * 3 + 1] = j.out[(size_t)i * 3 + 2] = __builtin_nanf("");
            }
        }
    }
    if (j.info) {
        if (threadIdx.x == 0) j.info[0] = n;
        if (n > 2 && n != size) { if (used) j.info[1] = used; }
        else if (threadIdx.x == 0) j.info[1] = 0;
    }
}

// One wave per frame: first arg-max of column 4 over the P proposals (np.argmax: lowest index among equal maxima).
__global__ __launch_bounds__(256) void select_box_kernel(const float* __restrict__ boxes, int B, int P, float* __restrict__ out,
                                                         int32_t* __restrict__ idx_out) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const float* row = boxes + (size_t)b * P * 5;
    float best = -__builtin_inff();
    int bi = 0x7fffffff;
    for (int p = lane; p < P; p += 64) {
        const float s = row[(size_t)p * 5 + 4];
        if (s > best) { best = s; bi = p; }             // strict: the lowest index of a lane's equal maxima stays
    }
    const float m = wave_max_f32(best);
    const int cand = (best == m) ? bi : 0x7fffffff;
    int win = wave_min_i32(cand);
    if (win == 0x7fffffff) win = 0;                      // every score -inf / NaN: np.argmax of all-equal gives 0
    if (lane < 5) out[(size_t)b * 5 + lane] = row[(size_t)win * 5 + lane];
    if (lane == 0 && idx_out) idx_out[b] = win;
}

}  // namespace ptt

using namespace ptt;

extern "C" int ptt_mt19937_fill(uint32_t seed, uint32_t* out_host, int n) {
    if (!out_host || n < 0) return fail(PTT_EINVAL, "ptt_mt19937_fill: null buffer or n=%d", n);
    mt19937_fill(seed, out_host, n);
    return PTT_OK;
}

extern "C" int ptt_crop_compact_f32(const ptt_crop_job* jobs_device, int n_jobs, ptt_stream_t stream) {
    if (n_jobs < 0) return fail(PTT_EINVAL, "ptt_crop_compact_f32: n_jobs=%d", n_jobs);
    if (n_jobs == 0) return PTT_OK;
    if (!jobs_device) return fail(PTT_EINVAL, "ptt_crop_compact_f32: null job array");
    hipLaunchKernelGGL((crop_compact_kernel<1024>), dim3(n_jobs), dim3(1024), 0, as_stream(stream), jobs_device);
    return check_launch("crop_compact_kernel");
}

extern "C" int ptt_regularize_f32(const ptt_regularize_job* jobs_device, int n_jobs, const uint32_t* draws, int n_draws,
                                  ptt_stream_t stream) {
    if (n_jobs < 0 || n_draws < 0) return fail(PTT_EINVAL, "ptt_regularize_f32: n_jobs=%d n_draws=%d", n_jobs, n_draws);
    if (n_jobs == 0) return PTT_OK;
    if (!jobs_device || !draws) return fail(PTT_EINVAL, "ptt_regularize_f32: null pointer");
    hipLaunchKernelGGL((regularize_kernel<256>), dim3(n_jobs), dim3(256), 0, as_stream(stream), jobs_device, draws, n_draws);
    return check_launch("regularize_kernel");
}

extern "C" int ptt_select_box_f32(const float* pred_box_data, int B, int P, float* out, int32_t* idx_out,
                                  ptt_stream_t stream) {
    if (B < 0 || P <= 0) return fail(PTT_EINVAL, "ptt_select_box_f32: B=%d P=%d", B, P);
    if (B == 0) return PTT_OK;
    if (!pred_box_data || !out) return fail(PTT_EINVAL, "ptt_select_box_f32: null pointer");
    hipLaunchKernelGGL(select_box_kernel, dim3((B + 3) / 4), dim3(256), 0, as_stream(stream), pred_box_data, B, P, out, idx_out);
    return check_launch("select_box_kernel");
}
