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
#include <math.h>
#include <stdlib.h>
#include <mutex>
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
__device__ __forceinline__ void crop_compact_body(const ptt_crop_job& j) {
    __shared__ int wsum[T / 64];
    const float* px = j.points;
    const float* py = j.points + j.ld;
    const float* pz = j.points + 2 * j.ld;
    int written = 0;
    for (int base = 0; base < j.n_points; base += T) {
        const int i = base + (int)threadIdx.x;
        bool keep = false, label = false;
        float ox = 0.f, oy = 0.f, oz = 0.f;
        if (i < j.n_points) {
            const float x = px[i], y = py[i], z = pz[i];
            // crop_pc in the cloud's frame: float32 point against float64 bounds, strict on both sides
            keep = (double)x > j.lo1[0] && (double)x < j.hi1[0] && (double)y > j.lo1[1] && (double)y < j.hi1[1] &&
                   (double)z > j.lo1[2] && (double)z < j.hi1[2];
            if (keep && j.label_out) {
                // get_label_by_box on the first crop: the same translate / rotate / strict test in the ground-truth box's frame
                const double tx = (double)(float)((double)x + j.ltrans[0]);
                const double ty = (double)(float)((double)y + j.ltrans[1]);
                const double tz = (double)(float)((double)z + j.ltrans[2]);
                const float lx = (float)fma(j.lrot[2], tz, fma(j.lrot[1], ty, j.lrot[0] * tx));
                const float ly = (float)fma(j.lrot[5], tz, fma(j.lrot[4], ty, j.lrot[3] * tx));
                const float lz = (float)fma(j.lrot[8], tz, fma(j.lrot[7], ty, j.lrot[6] * tx));
                label = (double)lx > j.llo[0] && (double)lx < j.lhi[0] && (double)ly > j.llo[1] && (double)ly < j.lhi[1] &&
                        (double)lz > j.llo[2] && (double)lz < j.lhi[2];
            }
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
            if (j.label_out) j.label_out[r] = label ? 1 : 0;
        }
        written += total;
    }
    if (threadIdx.x == 0) *j.count = written;       // may exceed capacity: the caller sized `out` for n_points
}

template <int T>
__global__ __launch_bounds__(T) void crop_compact_kernel(const ptt_crop_job* __restrict__ jobs) {
    const ptt_crop_job j = jobs[blockIdx.x];
    crop_compact_body<T>(j);
}

__device__ __forceinline__ void seg_point(const ptt_regularize_job& j, const int* cnt, int idx, float* dst) {
    int s = 0;
    while (s + 1 < j.n_seg && idx >= cnt[s]) { idx -= cnt[s]; ++s; }
    const float* p = j.seg[s] + (size_t)idx * 3;
    dst[0] = p[0]; dst[1] = p[1]; dst[2] = p[2];
}

// One workgroup per output cloud.
template <int T>
__device__ __forceinline__ void regularize_body(const ptt_regularize_job* __restrict__ jobs, const uint32_t* __restrict__ draws, int n_draws) {
    __shared__ int wsum[T / 64];
    // the job and the segment sizes live in LDS: seg_point() indexes them with a run-time segment number, which as
    // private arrays meant 112 B of scratch per lane
    __shared__ ptt_regularize_job j;
    __shared__ int cnt[PTT_MAX_SEGMENTS];
    static_assert(sizeof(ptt_regularize_job) % 4 == 0 && sizeof(ptt_regularize_job) / 4 <= T, "job copied one word per thread");
    if (threadIdx.x < sizeof(ptt_regularize_job) / 4)
        reinterpret_cast<uint32_t*>(&j)[threadIdx.x] = reinterpret_cast<const uint32_t*>(jobs + blockIdx.x)[threadIdx.x];
    __syncthreads();
    if (threadIdx.x < PTT_MAX_SEGMENTS) {
        const int s = threadIdx.x;
        cnt[s] = (s < j.n_seg) ? min(*j.seg_count[s], j.seg_capacity[s]) : 0;
    }
    __syncthreads();
    int n = 0;
#pragma unroll
    for (int s = 0; s < PTT_MAX_SEGMENTS; ++s) n += cnt[s];
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
            if (threadIdx.x == 0 && j.info) j.info[1] = -1;        // says so: the host raises (never a stale draw count)
        }
    }
    if (j.info) {
        if (threadIdx.x == 0) j.info[0] = n;
        if (n > 2 && n != size) { if (used) j.info[1] = used; }
        else if (threadIdx.x == 0) j.info[1] = 0;
    }
}

template <int T>
__global__ __launch_bounds__(T) void regularize_kernel(const ptt_regularize_job* __restrict__ jobs,
                                                       const uint32_t* __restrict__ draws, int n_draws) {
    regularize_body<T>(jobs, draws, n_draws);
}

// Crop and resampling of one frame in ONE launch: workgroup w runs crop job w, then resampling job w, whose segments are that
// crop's output (and, for a template, crops of earlier launches: the first frame's). What the tracking loop launches per
// frame for a handful of tracklets (prepare_search / prepare_template, eval_tracking_utils.py:155-229): job 2b = tracklet b's
// search cloud, job 2b + 1 = its previous-frame template crop followed by get_model + regularize_pc.
__global__ __launch_bounds__(1024) void crop_regularize_kernel(const ptt_crop_job* __restrict__ cjobs, const ptt_regularize_job* __restrict__ rjobs,
                                                               const uint32_t* __restrict__ draws, int n_draws) {
    const ptt_crop_job j = cjobs[blockIdx.x];
    crop_compact_body<1024>(j);
    __threadfence();                                 // the crop's points and count are read back by this workgroup below
    __syncthreads();
    regularize_body<1024>(rjobs, draws, n_draws);
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

// The same crops with the job table passed BY VALUE in the kernel arguments (a handful of jobs: one tracklet's frame is two):
// no per-frame host-to-device copy of the table in front of the launch.
struct CropJobsByValue { ptt_crop_job j[PTT_CROP_JOBS_BY_VALUE_MAX]; };
__global__ __launch_bounds__(1024) void crop_compact_byvalue_kernel(CropJobsByValue P) { ptt::crop_compact_body<1024>(P.j[blockIdx.x]); }

extern "C" int ptt_crop_compact_host_f32(const ptt_crop_job* jobs_host, int n_jobs, ptt_stream_t stream) {
    if (n_jobs < 0 || n_jobs > PTT_CROP_JOBS_BY_VALUE_MAX)
        return fail(PTT_EINVAL, "ptt_crop_compact_host_f32: n_jobs=%d (0..%d)", n_jobs, PTT_CROP_JOBS_BY_VALUE_MAX);
    if (n_jobs == 0) return PTT_OK;
    if (!jobs_host) return fail(PTT_EINVAL, "ptt_crop_compact_host_f32: null job array");
    CropJobsByValue P;
    for (int i = 0; i < n_jobs; ++i) P.j[i] = jobs_host[i];
    hipLaunchKernelGGL(crop_compact_byvalue_kernel, dim3(n_jobs), dim3(1024), 0, as_stream(stream), P);
    return check_launch("crop_compact_byvalue_kernel");
}

extern "C" int ptt_crop_compact_f32(const ptt_crop_job* jobs_device, int n_jobs, ptt_stream_t stream) {
    if (n_jobs < 0) return fail(PTT_EINVAL, "ptt_crop_compact_f32: n_jobs=%d", n_jobs);
    if (n_jobs == 0) return PTT_OK;
    if (!jobs_device) return fail(PTT_EINVAL, "ptt_crop_compact_f32: null job array");
    hipLaunchKernelGGL((crop_compact_kernel<1024>), dim3(n_jobs), dim3(1024), 0, as_stream(stream), jobs_device);
    return check_launch("crop_compact_kernel");
}

extern "C" int ptt_crop_regularize_f32(const ptt_crop_job* crop_jobs, const ptt_regularize_job* reg_jobs_device, int n_jobs,
                                       const uint32_t* draws, int n_draws, ptt_stream_t stream) {
    if (n_jobs < 0 || n_draws < 0) return fail(PTT_EINVAL, "ptt_crop_regularize_f32: n_jobs=%d n_draws=%d", n_jobs, n_draws);
    if (n_jobs == 0) return PTT_OK;
    if (!crop_jobs || !reg_jobs_device || !draws) return fail(PTT_EINVAL, "ptt_crop_regularize_f32: null pointer");
    hipLaunchKernelGGL(crop_regularize_kernel, dim3(n_jobs), dim3(1024), 0, as_stream(stream), crop_jobs, reg_jobs_device, draws, n_draws);
    return check_launch("crop_regularize_kernel");
}

extern "C" int ptt_regularize_f32(const ptt_regularize_job* jobs_device, int n_jobs, const uint32_t* draws, int n_draws,
                                  ptt_stream_t stream) {
    if (n_jobs < 0 || n_draws < 0) return fail(PTT_EINVAL, "ptt_regularize_f32: n_jobs=%d n_draws=%d", n_jobs, n_draws);
    if (n_jobs == 0) return PTT_OK;
    if (!jobs_device || !draws) return fail(PTT_EINVAL, "ptt_regularize_f32: null pointer");
    hipLaunchKernelGGL((regularize_kernel<256>), dim3(n_jobs), dim3(256), 0, as_stream(stream), jobs_device, draws, n_draws);
    return check_launch("regularize_kernel");
}

// ---- host-side float64 box arithmetic (restates pyquaternion's documented formulas, as box_math.py does) ----
namespace {
struct Q { double w, x, y, z; };
struct M3 { double m[3][3]; };

inline Q q_mul(const Q& a, const Q& b) {
    return Q{a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.x * b.w + a.w * b.x - a.z * b.y + a.y * b.z,
             a.y * b.w + a.z * b.x + a.w * b.y - a.x * b.z, a.z * b.w - a.y * b.x + a.x * b.y + a.w * b.z};
}
inline Q q_inverse(const Q& q) {
    const double ss = q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z;
    return Q{q.w / ss, -q.x / ss, -q.y / ss, -q.z / ss};
}
inline M3 q_rotation_matrix(Q q) {
    const double n = sqrt(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
    if (!(fabs(1.0 - n * n) < 1e-14) && n > 0) { q.w /= n; q.x /= n; q.y /= n; q.z /= n; }
    const double w = q.w, x = q.x, y = q.y, z = q.z;
    M3 r;
    r.m[0][0] = x * x + w * w - z * z - y * y; r.m[0][1] = x * y - w * z - z * w + y * x; r.m[0][2] = x * z + w * y + z * x + y * w;
    r.m[1][0] = y * x + z * w + w * z + x * y; r.m[1][1] = y * y - z * z + w * w - x * x; r.m[1][2] = y * z + z * y - w * x - x * w;
    r.m[2][0] = z * x - y * w + x * z - w * y; r.m[2][1] = z * y + y * z + x * w + w * x; r.m[2][2] = z * z - y * y - x * x + w * w;
    return r;
}
inline Q q_from_matrix(const M3& R) {           // branch on the diagonal of R^T, as pyquaternion's trace method does
    double m[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) m[i][j] = R.m[j][i];
    double t; Q q;
    if (m[2][2] < 0) {
        if (m[0][0] > m[1][1]) { t = 1 + m[0][0] - m[1][1] - m[2][2]; q = Q{m[1][2] - m[2][1], t, m[0][1] + m[1][0], m[2][0] + m[0][2]}; }
        else { t = 1 - m[0][0] + m[1][1] - m[2][2]; q = Q{m[2][0] - m[0][2], m[0][1] + m[1][0], t, m[1][2] + m[2][1]}; }
    } else {
        if (m[0][0] < -m[1][1]) { t = 1 - m[0][0] - m[1][1] + m[2][2]; q = Q{m[0][1] - m[1][0], m[2][0] + m[0][2], m[1][2] + m[2][1], t}; }
        else { t = 1 + m[0][0] + m[1][1] + m[2][2]; q = Q{t, m[1][2] - m[2][1], m[2][0] - m[0][2], m[0][1] - m[1][0]}; }
    }
    const double s = 0.5 / sqrt(t);
    return Q{q.w * s, q.x * s, q.y * s, q.z * s};
}
inline void mat_vec(const M3& R, const double* v, double* out) {
    for (int i = 0; i < 3; ++i) out[i] = R.m[i][0] * v[0] + R.m[i][1] * v[1] + R.m[i][2] * v[2];
}
// min / max over the 8 corners of Box.corners() (:140-158) for the box (center, wlh, R)
inline void corner_extent(const double* center, const double* wlh, const M3& R, double* lo, double* hi) {
    static const double sx[8] = {1, 1, 1, 1, -1, -1, -1, -1}, sy[8] = {1, -1, -1, 1, 1, -1, -1, 1}, sz[8] = {1, 1, -1, -1, 1, 1, -1, -1};
    const double w = wlh[0], l = wlh[1], h = wlh[2];
    for (int a = 0; a < 3; ++a) { lo[a] = 1e300; hi[a] = -1e300; }
    for (int k = 0; k < 8; ++k) {
        const double loc[3] = {l / 2 * sx[k], w / 2 * sy[k], h / 2 * sz[k]};
        double c[3];
        mat_vec(R, loc, c);
        for (int a = 0; a < 3; ++a) {
            const double v = c[a] + center[a];
            if (v < lo[a]) lo[a] = v;
            if (v > hi[a]) hi[a] = v;
        }
    }
}
inline void box_rotate(double* center, Q& quat, const Q& q) {      // Box.rotate (:127-130)
    const M3 R = q_rotation_matrix(q);
    double c[3];
    mat_vec(R, center, c);
    center[0] = c[0]; center[1] = c[1]; center[2] = c[2];
    quat = q_mul(q, quat);
}
}  // namespace

extern "C" int ptt_track_crop_bounds(const ptt_track_box* boxes, int n, double offset, double scale, const double* extra2,
                                     ptt_crop_job* jobs_host, int job_stride) {
    if (n < 0 || job_stride < 1) return fail(PTT_EINVAL, "ptt_track_crop_bounds: n=%d job_stride=%d", n, job_stride);
    if (n == 0) return PTT_OK;
    if (!boxes || !jobs_host) return fail(PTT_EINVAL, "ptt_track_crop_bounds: null pointer");
    for (int i = 0; i < n; ++i) {
        const ptt_track_box& b = boxes[i];
        ptt_crop_job& j = jobs_host[(size_t)i * job_stride];
        const Q q{b.quat[0], b.quat[1], b.quat[2], b.quat[3]};
        const M3 R = q_rotation_matrix(q);
        const double wlh1[3] = {b.wlh[0] * (4 * scale), b.wlh[1] * (4 * scale), b.wlh[2] * (4 * scale)};
        corner_extent(b.center, wlh1, R, j.lo1, j.hi1);
        for (int a = 0; a < 3; ++a) { j.lo1[a] -= 2 * offset; j.hi1[a] += 2 * offset; j.trans[a] = -b.center[a]; }
        M3 Rt;
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { Rt.m[r][c] = R.m[c][r]; j.rot[r * 3 + c] = R.m[c][r]; }
        double nc[3] = {b.center[0] + j.trans[0], b.center[1] + j.trans[1], b.center[2] + j.trans[2]};
        Q nq = q;
        box_rotate(nc, nq, q_from_matrix(Rt));
        const double wlh2[3] = {b.wlh[0] * scale, b.wlh[1] * scale, b.wlh[2] * scale};
        corner_extent(nc, wlh2, q_rotation_matrix(nq), j.lo2, j.hi2);
        const double off2 = offset + (extra2 ? extra2[i] : 0.0);
        for (int a = 0; a < 3; ++a) { j.lo2[a] -= off2; j.hi2[a] += off2; }
    }
    return PTT_OK;
}

extern "C" int ptt_track_box_by_offset(ptt_track_box* boxes, int n, float* offsets, int offset_stride, int use_z,
                                       const int32_t* active, int64_t* rng_pos) {
    if (n < 0 || offset_stride < 4) return fail(PTT_EINVAL, "ptt_track_box_by_offset: n=%d offset_stride=%d", n, offset_stride);
    if (n == 0) return PTT_OK;
    if (!boxes || !offsets) return fail(PTT_EINVAL, "ptt_track_box_by_offset: null pointer");
    const double kPi = 3.141592653589793;
    for (int i = 0; i < n; ++i) {
        if (active && !active[i]) continue;
        ptt_track_box& b = boxes[i];
        float* off = offsets + (size_t)i * offset_stride;
        Q quat{b.quat[0], b.quat[1], b.quat[2], b.quat[3]};
        const Q rot_quat = q_from_matrix(q_rotation_matrix(quat));
        const double trans[3] = {b.center[0], b.center[1], b.center[2]};
        double c[3] = {b.center[0] - trans[0], b.center[1] - trans[1], b.center[2] - trans[2]};
        box_rotate(c, quat, q_inverse(rot_quat));
        // offset[-1] * np.pi / 180 on a float32 element: float32 product, then float32 quotient (numpy >= 2 promotion)
        const float ang32 = (off[3] * (float)kPi) / 180.0f;     // python scalars are weak: pi and 180 enter as float32
        const double theta = (double)ang32 / 2.0;
        box_rotate(c, quat, Q{cos(theta), 0.0, 0.0, sin(theta)});
        for (int a = 0; a < 2; ++a) {
            const double lim = (a == 0) ? b.wlh[0] : (b.wlh[1] < 2.0 ? b.wlh[1] : 2.0);
            if ((double)off[a] > lim) {
                uint32_t d[2] = {0, 0};
                if (rng_pos) {
                    // outputs rng_pos[i], rng_pos[i] + 1 of MT19937(1): from a table built once per process (the
                    // resampling consumes a few thousand outputs at most); beyond it, regenerate the prefix
                    static uint32_t table[32768];
                    static std::once_flag once;
                    std::call_once(once, [] { mt19937_fill(1u, table, 32768); });
                    const int64_t pos = rng_pos[i];
                    if (pos < 0) return fail(PTT_EINVAL, "ptt_track_box_by_offset: negative generator position");
                    if (pos + 2 <= 32768) {
                        d[0] = table[pos]; d[1] = table[pos + 1];
                    } else {
                        const int cnt = (int)pos + 2;
                        uint32_t* buf = (uint32_t*)malloc((size_t)cnt * sizeof(uint32_t));
                        if (!buf) return fail(PTT_EINVAL, "ptt_track_box_by_offset: out of memory");
                        mt19937_fill(1u, buf, cnt);
                        d[0] = buf[pos]; d[1] = buf[pos + 1];
                        free(buf);
                    }
                    rng_pos[i] = pos + 2;
                }
                // RandomState.uniform(-1, 1) = -1 + 2 * random_sample(), random_sample from two 32-bit outputs (53 bits)
                const double u = ((double)(d[0] >> 5) * 67108864.0 + (double)(d[1] >> 6)) / 9007199254740992.0;
                off[a] = (float)(-1.0 + 2.0 * u);
            }
        }
        c[0] += (double)off[0]; c[1] += (double)off[1]; c[2] += use_z ? (double)off[2] : 0.0;
        box_rotate(c, quat, rot_quat);
        b.center[0] = c[0] + trans[0]; b.center[1] = c[1] + trans[1]; b.center[2] = c[2] + trans[2];
        b.quat[0] = quat.w; b.quat[1] = quat.x; b.quat[2] = quat.y; b.quat[3] = quat.z;
    }
    return PTT_OK;
}

extern "C" int ptt_select_box_f32(const float* pred_box_data, int B, int P, float* out, int32_t* idx_out,
                                  ptt_stream_t stream) {
    if (B < 0 || P <= 0) return fail(PTT_EINVAL, "ptt_select_box_f32: B=%d P=%d", B, P);
    if (B == 0) return PTT_OK;
    if (!pred_box_data || !out) return fail(PTT_EINVAL, "ptt_select_box_f32: null pointer");
    hipLaunchKernelGGL(select_box_kernel, dim3((B + 3) / 4), dim3(256), 0, as_stream(stream), pred_box_data, B, P, out, idx_out);
    return check_launch("select_box_kernel");
}

// The host side of post_process for one step of B tracklets in ONE call (eval_tracking_utils.py:266-274 + the generator bookkeeping
// of the loop around it): for every tracklet the FIRST arg-max of the proposal scores (np.argmax, :267-269) out of the (B,P,5)
// read-back (P == 1: the rows are already selected), the position of numpy's global generator after this frame's resampling
// (the template's draw count if it resampled, else the search's: regularize_pc reseeds on every call, :349-350), then
// ptt_track_box_by_offset. info (B,2,2) int32 = (n, draws used) of the search / template resampling; a negative draw count (the
// draw table ran out) is reported as PTT_EINVAL. est_out (B,5) receives the selected rows with the offsets actually used.
extern "C" int ptt_track_select_update(const float* proposals, int P, const int32_t* info, ptt_track_box* boxes, int n, int use_z,
                                       const int32_t* active, int64_t* rng_pos, float* est_out) {
    if (n < 0 || P < 1) return fail(PTT_EINVAL, "ptt_track_select_update: n=%d P=%d", n, P);
    if (n == 0) return PTT_OK;
    if (!proposals || !info || !boxes || !rng_pos || !est_out) return fail(PTT_EINVAL, "ptt_track_select_update: null pointer");
    for (int b = 0; b < n; ++b) {
        const float* rows = proposals + (size_t)b * P * 5;
        int best = 0;                                                        // np.argmax: the first maximum; the first NaN if there is one
        bool nan_seen = rows[4] != rows[4];
        for (int k = 1; k < P && !nan_seen; ++k) {
            const float v = rows[k * 5 + 4];
            if (v != v) { best = k; nan_seen = true; }
            else if (v > rows[best * 5 + 4]) best = k;
        }
        for (int c = 0; c < 5; ++c) est_out[b * 5 + c] = rows[best * 5 + c];
        const int32_t* f = info + b * 4;
        if (f[1] < 0 || f[3] < 0)
            return fail(PTT_EINVAL, "ptt_track_select_update: ptt_regularize_f32 ran out of pre-drawn MT19937 outputs (tracklet %d)", b);
        const int32_t used = f[3] > 0 ? f[3] : f[1];
        if (used > 0) rng_pos[b] = used;
    }
    return ptt_track_box_by_offset(boxes, n, est_out, 5, use_z, active, rng_pos);
}

