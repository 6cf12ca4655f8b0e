// N3: hand-written kernels of the TRAINING step for the shared-MLP stages (set-abstraction levels,
// CosineSimAug): 1x1 convolution -> BatchNorm with BATCH statistics -> ReLU, three times, then a max over the
// neighbour axis — the reference's SharedMLP + F.max_pool2d in train mode (pytorch_utils.py:12-36,94-114,
// pointnet2_modules.py:84-88) and their backward (tools/train_utils/train_utils.py:47-48, loss.backward()).
//
// Layout: activations are ROWS x CHANNELS ("point-major", one row per (centre, neighbour) position), so that
//   * the convolution is ptt_linear_f32 on the fp32-MFMA linear kernel (forward and the input gradient),
//   * the weight gradient is one more MFMA GEMM over the row axis (linear_wgrad_kernel below),
//   * BatchNorm statistics are column reductions, coalesced over channels, in a FIXED order: per-workgroup partial sums
//     in float64, combined by a single finalising workgroup in chunk order — bit-reproducible run to run, which
//     torch's / MIOpen's atomics-free-but-layout-dependent kernels do not promise across versions,
//   * no NCHW <-> NHWC transposes exist (the stock step spends 6 % of its time in batched_transpose kernels).
#include <math.h>
#include "common.h"

namespace ptt {

typedef float f32x4t __attribute__((ext_vector_type(4)));
typedef float f32x16t __attribute__((ext_vector_type(16)));

constexpr int ST_ROWS = 2048;       // rows per workgroup of the column-statistics kernels

// ------------------------------------------------------------------------------------------
// Column statistics. MODE 0: s0 = sum x, s1 = sum x^2.
//                    MODE 1: BatchNorm+ReLU backward sums — dy = g where the ReLU passed (act > 0), else 0;
//                            s0 = sum dy, s1 = sum dy * xhat, xhat = (z - mean) * invstd.
// One workgroup = 256 threads = 4 row groups x 64 column lanes; thread (rg, lane) visits rows rg, rg+4, ... of its
// chunk and columns lane, lane+64, ...; float64 accumulation; partial[chunk][2][C].
// ------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void col_stats_kernel(const float* __restrict__ X, const float* __restrict__ Act,
                                                        const float* __restrict__ Z, const float* __restrict__ mean,
                                                        const float* __restrict__ invstd, int R, int C, int ldx, int lda,
                                                        int ldz, double* __restrict__ partial) {
    __shared__ double red[2][4][64];
    const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int r0 = blockIdx.x * ST_ROWS, r1 = min(R, r0 + ST_ROWS);
    for (int cg = 0; cg * 64 < C; ++cg) {
        const int c = cg * 64 + lane;
        double s0 = 0.0, s1 = 0.0;
        if (c < C) {
            float mu = 0.f, is = 0.f;
            if (MODE == 1) { mu = mean[c]; is = invstd[c]; }
            for (int r = r0 + rg; r < r1; r += 4) {
                const float x = X[(size_t)r * ldx + c];
                if (MODE == 0) {
                    s0 += (double)x;
                    s1 += (double)x * (double)x;
                } else {
                    const float dy = Act[(size_t)r * lda + c] > 0.f ? x : 0.f;
                    const float xh = (Z[(size_t)r * ldz + c] - mu) * is;
                    s0 += (double)dy;
                    s1 += (double)dy * (double)xh;
                }
            }
        }
        red[0][rg][lane] = s0;
        red[1][rg][lane] = s1;
        __syncthreads();
        if (rg == 0 && c < C) {
            partial[((size_t)blockIdx.x * 2 + 0) * C + c] = ((red[0][0][lane] + red[0][1][lane]) + red[0][2][lane]) + red[0][3][lane];
            partial[((size_t)blockIdx.x * 2 + 1) * C + c] = ((red[1][0][lane] + red[1][1][lane]) + red[1][2][lane]) + red[1][3][lane];
        }
        __syncthreads();
    }
}

// Combine the partials in chunk order. MODE 0: mean, biased variance, invstd = 1/sqrt(var + eps).
//                                     MODE 1: the two sums themselves as float (out0 = sum dy, out1 = sum dy*xhat).
// What a training-mode BatchNorm does with fresh batch statistics besides normalising, folded into the kernel that forms them
// (31 layers x 4 five-microsecond launches per training step otherwise): the deferred activation's constants
// a = gamma * invstd, b = beta - mean * a (two separately rounded operations, as the element-wise torch expressions they
// replace) and nn.BatchNorm's bookkeeping (running statistics with the n / (n - 1) variance, batch counter). Null pointers
// switch the pieces off.
struct BnTail {
    const float* gamma; const float* beta; float* act_a; float* act_b;
    float* running_mean; float* running_var; long long* tracked; float momentum;
};
__device__ __forceinline__ void bn_tail(const BnTail& t, int c, float mean, float var, float invstd, int R) {
#pragma clang fp contract(off)      // HIP's __fmul_rn / __fsub_rn are plain operators: hipcc's default contraction would fuse them
    if (t.act_a) {
        const float a = t.gamma[c] * invstd;
        const float ma = mean * a;
        t.act_a[c] = a;
        t.act_b[c] = t.beta[c] - ma;
    }
    if (t.running_mean) {
        const double n = (double)R;
        const float unbias = (float)(n / (n > 1.0 ? n - 1.0 : 1.0));
        t.running_mean[c] = (1.f - t.momentum) * t.running_mean[c] + t.momentum * mean;
        t.running_var[c] = (1.f - t.momentum) * t.running_var[c] + t.momentum * (var * unbias);
    }
    if (c == 0 && t.tracked) *t.tracked += 1;
}

template <int MODE>
__global__ __launch_bounds__(256) void col_stats_finish_kernel(const double* __restrict__ partial, int nchunks, int C, int R,
                                                               float eps, float* __restrict__ out0, float* __restrict__ out1,
                                                               float* __restrict__ out2, BnTail tail) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    double s0 = 0.0, s1 = 0.0;
    for (int k = 0; k < nchunks; ++k) {
        s0 += partial[((size_t)k * 2 + 0) * C + c];
        s1 += partial[((size_t)k * 2 + 1) * C + c];
    }
    if (MODE == 0) {
        const double mu = s0 / (double)R;
        double var = s1 / (double)R - mu * mu;
        if (var < 0.0) var = 0.0;
        const float inv = (float)(1.0 / sqrt(var + (double)eps));
        out0[c] = (float)mu;
        out1[c] = (float)var;
        out2[c] = inv;
        bn_tail(tail, c, (float)mu, (float)var, inv, R);
    } else {
        out0[c] = (float)s0;
        out1[c] = (float)s1;
    }
}

// Vectorised forms for C % 4 == 0, 16-byte aligned rows (every conv OUTPUT of the shipped networks: 64 / 128 / 256
// channels). A workgroup covers ST4_ROWS rows; thread (rg, q) owns the channel quad q of the rows rg, rg + RG, ...
// (RG = 256 / quads-per-row threads side by side read whole rows, coalesced float4s); float64 accumulation; the RG
// row groups are combined through LDS in a fixed order.
constexpr int ST4_ROWS = 256;

template <int MODE>
__global__ __launch_bounds__(256) void col_stats4_kernel(const float* __restrict__ X, const float* __restrict__ Act,
                                                         const float* __restrict__ Z, const float* __restrict__ mean,
                                                         const float* __restrict__ invstd, int R, int C, int ldx, int lda,
                                                         int ldz, double* __restrict__ partial, const float* __restrict__ act_a,
                                                         const float* __restrict__ act_b) {
    extern __shared__ double red4[];                       // [256][8]
    const int Cq = C >> 2;                                  // quads per row
    const int span = Cq < 256 ? Cq : 256;                   // threads side by side on one row
    const int RG = 256 / span;                              // rows in flight per pass
    const int q0 = threadIdx.x % span, rg = threadIdx.x / span;
    const int r0 = blockIdx.x * ST4_ROWS, r1 = min(R, r0 + ST4_ROWS);
    for (int qb = 0; qb < Cq; qb += span) {                 // one pass unless C > 1024
        const int q = qb + q0;
        double s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
        if (q < Cq && rg < RG) {
            f32x4t mu = {0, 0, 0, 0}, is = {0, 0, 0, 0};
            f32x4t ga = {0, 0, 0, 0}, gb = {0, 0, 0, 0};
            if (MODE == 1) { mu = *reinterpret_cast<const f32x4t*>(mean + 4 * q); is = *reinterpret_cast<const f32x4t*>(invstd + 4 * q); }
            if (MODE == 1 && !Act) { ga = *reinterpret_cast<const f32x4t*>(act_a + 4 * q); gb = *reinterpret_cast<const f32x4t*>(act_b + 4 * q); }
            for (int r = r0 + rg; r < r1; r += RG) {
                const f32x4t x = *reinterpret_cast<const f32x4t*>(X + (size_t)r * ldx + 4 * q);
                if (MODE == 0) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) { s0[k] += (double)x[k]; s1[k] += (double)x[k] * (double)x[k]; }
                } else {
                    const f32x4t z = *reinterpret_cast<const f32x4t*>(Z + (size_t)r * ldz + 4 * q);
                    f32x4t a;
                    if (Act) a = *reinterpret_cast<const f32x4t*>(Act + (size_t)r * lda + 4 * q);
                    else {                           // the activation was never materialised: relu(z * ga + gb) > 0
#pragma unroll
                        for (int k = 0; k < 4; ++k) a[k] = __builtin_fmaf(z[k], ga[k], gb[k]);
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float dy = a[k] > 0.f ? x[k] : 0.f;
                        const float xh = (z[k] - mu[k]) * is[k];
                        s0[k] += (double)dy;
                        s1[k] += (double)dy * (double)xh;
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) { red4[threadIdx.x * 8 + k] = s0[k]; red4[threadIdx.x * 8 + 4 + k] = s1[k]; }
        __syncthreads();
        if (rg == 0 && q < Cq) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                double a = 0.0, b = 0.0;
                for (int g = 0; g < RG; ++g) { a += red4[(g * span + q0) * 8 + k]; b += red4[(g * span + q0) * 8 + 4 + k]; }
                partial[((size_t)blockIdx.x * 2 + 0) * C + 4 * q + k] = a;
                partial[((size_t)blockIdx.x * 2 + 1) * C + 4 * q + k] = b;
            }
        }
        __syncthreads();
    }
}

// Second level for many chunks: workgroup = one channel, 256 threads stride over the chunks, LDS tree in a fixed order.
template <int MODE>
__global__ __launch_bounds__(256) void col_stats_finish2_kernel(const double* __restrict__ partial, int nchunks, int C, int R,
                                                                float eps, float* __restrict__ out0, float* __restrict__ out1,
                                                                float* __restrict__ out2, BnTail tail) {
    __shared__ double t0[256], t1[256];
    const int c = blockIdx.x, t = threadIdx.x;
    double s0 = 0.0, s1 = 0.0;
    for (int k = t; k < nchunks; k += 256) {
        s0 += partial[((size_t)k * 2 + 0) * C + c];
        s1 += partial[((size_t)k * 2 + 1) * C + c];
    }
    t0[t] = s0; t1[t] = s1;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (t < w) { t0[t] += t0[t + w]; t1[t] += t1[t + w]; }
        __syncthreads();
    }
    if (t == 0) {
        if (MODE == 0) {
            const double mu = t0[0] / (double)R;
            double var = t1[0] / (double)R - mu * mu;
            if (var < 0.0) var = 0.0;
            const float inv = (float)(1.0 / sqrt(var + (double)eps));
            out0[c] = (float)mu; out1[c] = (float)var; out2[c] = inv;
            bn_tail(tail, c, (float)mu, (float)var, inv, R);
        } else {
            out0[c] = (float)t0[0]; out1[c] = (float)t1[0];
        }
    }
}

// The same combine (MODE 1's order) for a BatchNorm backward whose dz is formed by its CONSUMER: dbeta = sum dy, dgamma = sum dy * xhat and
// the three per-channel constants of  dz = c0 + c1 * (z - mean) + (mask ? k1 * g : 0)  (include/ptt_hip.h: ptt_bn_bwd_consts_f32)
__global__ __launch_bounds__(256) void bn_bwd_consts_kernel(const double* __restrict__ partial, int nchunks, int C, int R,
                                                            const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                            float* __restrict__ dbeta, float* __restrict__ dgamma,
                                                            float* __restrict__ k1, float* __restrict__ c0, float* __restrict__ c1) {
    __shared__ double t0[256], t1[256];
    const int c = blockIdx.x, t = threadIdx.x;
    double s0 = 0.0, s1 = 0.0;
    for (int k = t; k < nchunks; k += 256) {
        s0 += partial[((size_t)k * 2 + 0) * C + c];
        s1 += partial[((size_t)k * 2 + 1) * C + c];
    }
    t0[t] = s0; t1[t] = s1;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (t < w) { t0[t] += t0[t + w]; t1[t] += t1[t + w]; }
        __syncthreads();
    }
    if (t == 0) {
        const float db = (float)t0[0], dg = (float)t1[0];
        dbeta[c] = db; dgamma[c] = dg;
        const double is = (double)invstd[c], kk = (double)gamma[c] * is, rinv = 1.0 / (double)R;
        k1[c] = (float)kk;
        c1[c] = (float)(-kk * ((double)dg * rinv * is));            // coefficient of (z - mean)
        c0[c] = (float)(-kk * ((double)db * rinv));
    }
}

// SyncBatchNorm pieces: the two column sums themselves in float64 (sums[c], sums[C + c]), combined in the same fixed order
// as col_stats_finish2_kernel, so that ranks can add their sums before the statistics are formed ...
__global__ __launch_bounds__(256) void col_sums_finish_kernel(const double* __restrict__ partial, int nchunks, int C,
                                                              double* __restrict__ sums, double rows) {   // rows < 0: no count slot
    __shared__ double t0[256], t1[256];
    const int c = blockIdx.x, t = threadIdx.x;
    double s0 = 0.0, s1 = 0.0;
    for (int k = t; k < nchunks; k += 256) {
        s0 += partial[((size_t)k * 2 + 0) * C + c];
        s1 += partial[((size_t)k * 2 + 1) * C + c];
    }
    t0[t] = s0; t1[t] = s1;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (t < w) { t0[t] += t0[t + w]; t1[t] += t1[t + w]; }
        __syncthreads();
    }
    if (t == 0) {
        sums[c] = t0[0]; sums[C + c] = t1[0];
        if (c == 0 && rows >= 0.0) sums[2 * C] = rows;
    }
}
// ... and mean / biased variance / invstd from (possibly all-reduced) sums over `count` rows
__global__ __launch_bounds__(256) void bn_finish_sums_kernel(const double* __restrict__ sums, int C, float eps,
                                                             float* __restrict__ mean, float* __restrict__ var,
                                                             float* __restrict__ invstd) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const double count = sums[2 * C];
    const double mu = sums[c] / count;
    double v = sums[C + c] / count - mu * mu;
    if (v < 0.0) v = 0.0;
    mean[c] = (float)mu; var[c] = (float)v; invstd[c] = (float)(1.0 / sqrt(v + (double)eps));
}

// element-wise kernels, vector form: a workgroup walks whole rows, thread = one channel quad (no index division)
__global__ __launch_bounds__(256) void bn_apply4_kernel(const float* __restrict__ Z, int ldz, const float* __restrict__ mean,
                                                        const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, int R, int C, int relu,
                                                        float* __restrict__ X, int ldx) {
    const int Cq = C >> 2, span = Cq < 256 ? Cq : 256, RG = 256 / span;
    const int q0 = threadIdx.x % span, rg = threadIdx.x / span;
    if (rg >= RG) return;
    for (int q = q0; q < Cq; q += span) {
        const f32x4t mu = *reinterpret_cast<const f32x4t*>(mean + 4 * q), is = *reinterpret_cast<const f32x4t*>(invstd + 4 * q);
        const f32x4t ga = *reinterpret_cast<const f32x4t*>(gamma + 4 * q), be = *reinterpret_cast<const f32x4t*>(beta + 4 * q);
        for (int r = blockIdx.x * RG + rg; r < R; r += gridDim.x * RG) {
            const f32x4t z = *reinterpret_cast<const f32x4t*>(Z + (size_t)r * ldz + 4 * q);
            f32x4t y;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                y[k] = (z[k] - mu[k]) * is[k] * ga[k] + be[k];
                if (relu) y[k] = fmaxf(y[k], 0.f);
            }
            *reinterpret_cast<f32x4t*>(X + (size_t)r * ldx + 4 * q) = y;
        }
    }
}

// POOLED: the incoming gradient is the POOLED one — dy[g * ns + k, c] = G[g, c] (row stride ldg) if k == arg[g, c] else 0,
// the backward of the max over the ns rows of group g; the (R, C) gradient tensor itself is never written or read.
template <bool POOLED>
__global__ __launch_bounds__(256) void bn_bwd_apply4_kernel(const float* __restrict__ G, int ldg, const float* __restrict__ Act,
                                                            int lda, const float* __restrict__ Z, int ldz,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd,
                                                            const float* __restrict__ gamma, const float* __restrict__ s0,
                                                            const float* __restrict__ s1, int R, int C, float* __restrict__ dZ,
                                                            int ldd, const float* __restrict__ act_a, const float* __restrict__ act_b,
                                                            float rinv_value, const double* __restrict__ count,
                                                            const int32_t* __restrict__ arg, int ns) {
    // 1 / rows the statistics were taken over: by value, or (SyncBatchNorm) from the all-reduced row count in device memory
    const int Cq = C >> 2, span = Cq < 256 ? Cq : 256, RG = 256 / span;
    const int q0 = threadIdx.x % span, rg = threadIdx.x / span;
    if (rg >= RG) return;
    const float rinv = count ? (float)(1.0 / *count) : rinv_value;
    for (int q = q0; q < Cq; q += span) {
        const f32x4t mu = *reinterpret_cast<const f32x4t*>(mean + 4 * q), is = *reinterpret_cast<const f32x4t*>(invstd + 4 * q);
        const f32x4t ga = *reinterpret_cast<const f32x4t*>(gamma + 4 * q);
        const f32x4t a0 = *reinterpret_cast<const f32x4t*>(s0 + 4 * q), a1 = *reinterpret_cast<const f32x4t*>(s1 + 4 * q);
        f32x4t ca = {0, 0, 0, 0}, cb = {0, 0, 0, 0};
        if (!Act) { ca = *reinterpret_cast<const f32x4t*>(act_a + 4 * q); cb = *reinterpret_cast<const f32x4t*>(act_b + 4 * q); }
        for (int r = blockIdx.x * RG + rg; r < R; r += gridDim.x * RG) {
            f32x4t g;
            if (POOLED) {
                const int grp = r / ns, k = r - grp * ns;
                const f32x4t gp = *reinterpret_cast<const f32x4t*>(G + (size_t)grp * ldg + 4 * q);
                const int4 ar = *reinterpret_cast<const int4*>(arg + (size_t)grp * C + 4 * q);
                g[0] = ar.x == k ? gp[0] : 0.f; g[1] = ar.y == k ? gp[1] : 0.f;
                g[2] = ar.z == k ? gp[2] : 0.f; g[3] = ar.w == k ? gp[3] : 0.f;
            } else {
                g = *reinterpret_cast<const f32x4t*>(G + (size_t)r * ldg + 4 * q);
            }
            const f32x4t z = *reinterpret_cast<const f32x4t*>(Z + (size_t)r * ldz + 4 * q);
            f32x4t a;
            if (Act) a = *reinterpret_cast<const f32x4t*>(Act + (size_t)r * lda + 4 * q);
            else {
#pragma unroll
                for (int k = 0; k < 4; ++k) a[k] = __builtin_fmaf(z[k], ca[k], cb[k]);
            }
            f32x4t d;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float dy = a[k] > 0.f ? g[k] : 0.f;
                const float xh = (z[k] - mu[k]) * is[k];
                d[k] = ga[k] * is[k] * (dy - a0[k] * rinv - xh * (a1[k] * rinv));
            }
            *reinterpret_cast<f32x4t*>(dZ + (size_t)r * ldd + 4 * q) = d;
        }
    }
}

// BatchNorm + ReLU backward sums when the gradient arrives POOLED: dy is non-zero only at the arg-max row of every
// (group, channel), so s0 = sum_g dy and s1 = sum_g dy * xhat run over G x C entries instead of R x C. One workgroup per
// chunk of PB_GROUPS groups, thread = channel (coalesced dP / arg rows; z is gathered at the arg-max row), float64 partials
// [chunk][2][C] in group order, combined by col_stats_finish2_kernel<1> / col_sums_finish_kernel as the dense form's.
constexpr int PB_GROUPS = 8;        // (64 until round 4: 24 workgroups for the vote aggregation's 3072 groups, 70 us of gathers each)
// groups per workgroup: PB_GROUPS, but never more chunks than ptt_bn_stats_workspace(R, C) holds (one per ST4_ROWS rows, at least
// STATS_MIN_CHUNKS); a small problem (the Conv1d stacks of the heads: 6144 rows, "groups" of one row) is cut into ~512 chunks
// anyway — 24 workgroups walking 256 dependent gathers each took 69 us
constexpr int STATS_MIN_CHUNKS = 512;
static inline int pool_bwd_groups_per_chunk(int G, int ns) {
    const int m = (ST4_ROWS + ns - 1) / ns;
    int per = m > PB_GROUPS ? m : PB_GROUPS;
    if ((G + per - 1) / per < STATS_MIN_CHUNKS / 2) per = (G + STATS_MIN_CHUNKS - 1) / STATS_MIN_CHUNKS;
    return per < 1 ? 1 : per;
}
__global__ __launch_bounds__(256) void pool_bwd_stats_kernel(const float* __restrict__ dP, int ldp, const int32_t* __restrict__ arg,
                                                             int G, int ns, const float* __restrict__ Z, int ldz,
                                                             const float* __restrict__ mean, const float* __restrict__ invstd,
                                                             const float* __restrict__ act_a, const float* __restrict__ act_b, int C,
                                                             double* __restrict__ partial, int per) {
    // thread = (group slot rg, channel): span channels side by side (coalesced dP / arg rows), the 256 / span group slots
    // walk the chunk's groups rg, rg + RG, ... with independent gathers in flight; slots combined through LDS in slot order
    __shared__ double red[2][256];
    const int span = C < 256 ? C : 256, RG = 256 / span;
    const int c0 = threadIdx.x % span, rg = threadIdx.x / span;
    const int g0 = blockIdx.x * per, g1 = min(G, g0 + per);
    for (int cb = 0; cb < C; cb += span) {
        const int c = cb + c0;
        double s0 = 0.0, s1 = 0.0;
        if (rg < RG && c < C) {
            const float mu = mean[c], is = invstd[c], a = act_a[c], b = act_b[c];
#pragma unroll 4
            for (int g = g0 + rg; g < g1; g += RG) {
                const int k = arg[(size_t)g * C + c];
                const float z = Z[((size_t)g * ns + k) * ldz + c];
                const float d = dP[(size_t)g * ldp + c];
                const float dy = __builtin_fmaf(z, a, b) > 0.f ? d : 0.f;
                const float xh = (z - mu) * is;
                s0 += (double)dy;
                s1 += (double)dy * (double)xh;
            }
        }
        red[0][threadIdx.x] = s0;
        red[1][threadIdx.x] = s1;
        __syncthreads();
        if (rg == 0 && c < C) {
            double t0 = 0.0, t1 = 0.0;
            for (int r = 0; r < RG; ++r) { t0 += red[0][r * span + c0]; t1 += red[1][r * span + c0]; }
            partial[((size_t)blockIdx.x * 2 + 0) * C + c] = t0;
            partial[((size_t)blockIdx.x * 2 + 1) * C + c] = t1;
        }
        __syncthreads();
    }
}

// nn.BatchNorm's training-mode bookkeeping in one launch: running_mean / running_var <- (1 - m) * running + m * batch
// (the variance unbiased by n / (n - 1), n = rows the statistics were taken over, in device memory: with SyncBatchNorm it
// is the all-reduced count), num_batches_tracked += 1 (torch/nn/modules/batchnorm.py; pytorch_utils.py:94-114 builds the units)
__global__ __launch_bounds__(256) void bn_running_update_kernel(const float* __restrict__ mean, const float* __restrict__ var,
                                                                const double* __restrict__ count, float momentum, int C,
                                                                float* __restrict__ running_mean, float* __restrict__ running_var,
                                                                long long* __restrict__ tracked) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c == 0 && tracked) *tracked += 1;
    if (c >= C) return;
    const double n = *count;
    const float unbias = (float)(n / (n > 1.0 ? n - 1.0 : 1.0));
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean[c];
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (var[c] * unbias);
}

// CosineSimAug's hoisted layer 0 in TRAINING mode (train_ops.xcorr_hoisted; p2b_xcoor.py:35-40): the only channel of the
// (B,260,n1,n2) fusion tensor that depends on the search point is the cosine, so the first convolution's output is
//   z0[b, j, i, :] = P[b, i, :] + cos[b, j, i] * w[:]          (j search point, i template point)
// forward: one pass that writes z0 (rows ordered (b, j, i)); backward: ONE pass over dz0 gives dP[b,i,:] = sum_j dz0,
// dcos[b,j,i] = <dz0[b,j,i,:], w> and the per-workgroup partials of dw[:] = sum dz0 * cos (summed in workgroup order).
// stats (optional, C <= 1024): the float64 column sums / sums of squares of the rows this workgroup writes, partial
// [workgroup][2][C] as sa_z0_rows_kernel's: layer 0's BatchNorm statistics without a pass of their own over z0
__global__ __launch_bounds__(256) void xcorr_z0_kernel(const float* __restrict__ P, const float* __restrict__ cosm,
                                                       const float* __restrict__ w, int n2, int n1, int C, long long total_rows,
                                                       float* __restrict__ z0, double* __restrict__ stats) {
    __shared__ double red[2][256][4];
    const int Cq = C >> 2, span = Cq < 256 ? Cq : 256, RG = 256 / span;
    const int q0 = threadIdx.x % span, rg = threadIdx.x / span;
    double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
    if (rg < RG) {
        for (int q = q0; q < Cq; q += span) {
            const f32x4t w4 = *reinterpret_cast<const f32x4t*>(w + 4 * q);
            for (long long r = (long long)blockIdx.x * RG + rg; r < total_rows; r += (long long)gridDim.x * RG) {
                const long long bj = r / n1;                       // (b, j)
                const int i = (int)(r - bj * n1);
                const long long b = bj / n2;
                const float cv = cosm[r];
                const f32x4t pv = *reinterpret_cast<const f32x4t*>(P + (b * n1 + i) * C + 4 * q);
                f32x4t o;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    o[k] = __builtin_fmaf(cv, w4[k], pv[k]);
                    if (stats) { s1[k] += (double)o[k]; s2[k] += (double)o[k] * (double)o[k]; }
                }
                *reinterpret_cast<f32x4t*>(z0 + r * C + 4 * q) = o;
            }
        }
    }
    if (!stats) return;
#pragma unroll
    for (int k = 0; k < 4; ++k) { red[0][threadIdx.x][k] = s1[k]; red[1][threadIdx.x][k] = s2[k]; }
    __syncthreads();
    if (rg == 0) {                                   // the row groups' sums in group order (span == Cq: one quad per thread)
        double* sp = stats + (size_t)blockIdx.x * 2 * C;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            double t1 = 0.0, t2 = 0.0;
            for (int g = 0; g < RG; ++g) { t1 += red[0][g * span + q0][k]; t2 += red[1][g * span + q0][k]; }
            sp[4 * q0 + k] = t1; sp[C + 4 * q0 + k] = t2;
        }
    }
}

// one wave per (b, i): walks the n2 search points; lane = channel quad (C <= 256)
__global__ __launch_bounds__(256) void xcorr_z0_bwd_kernel(const float* __restrict__ dz0, const float* __restrict__ cosm,
                                                           const float* __restrict__ w, int B, int n2, int n1, int C,
                                                           float* __restrict__ dP, float* __restrict__ dcos,
                                                           float* __restrict__ dw_partial) {
    __shared__ float red[4][256];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int bi = blockIdx.x * 4 + wv;                        // flat (b, i)
    const int Cq = C >> 2;
    const bool on = lane < Cq && bi < B * n1;
    f32x4t accP = {0.f, 0.f, 0.f, 0.f}, accW = {0.f, 0.f, 0.f, 0.f};
    if (bi < B * n1) {
        const int b = bi / n1, i = bi - b * n1;
        f32x4t w4 = {0.f, 0.f, 0.f, 0.f};
        if (on) w4 = *reinterpret_cast<const f32x4t*>(w + 4 * lane);
        int j = 0;
        for (; j + 3 < n2; j += 4) {                       // four rows in flight; the sums keep the order of the single-row loop
            f32x4t v[4];
            float cv[4], dot[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long long r = ((long long)b * n2 + j + u) * n1 + i;
                v[u] = f32x4t{0.f, 0.f, 0.f, 0.f};
                if (on) v[u] = *reinterpret_cast<const f32x4t*>(dz0 + r * C + 4 * lane);
                cv[u] = cosm[r];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                dot[u] = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) { accP[k] += v[u][k]; accW[k] = __builtin_fmaf(v[u][k], cv[u], accW[k]); dot[u] = __builtin_fmaf(v[u][k], w4[k], dot[u]); }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1)
#pragma unroll
                for (int u = 0; u < 4; ++u) dot[u] += __shfl_xor(dot[u], off, 64);
            if (lane < 4) dcos[((long long)b * n2 + j + lane) * n1 + i] = lane == 0 ? dot[0] : lane == 1 ? dot[1] : lane == 2 ? dot[2] : dot[3];
        }
        for (; j < n2; ++j) {
            const long long r = ((long long)b * n2 + j) * n1 + i;
            f32x4t v = {0.f, 0.f, 0.f, 0.f};
            if (on) v = *reinterpret_cast<const f32x4t*>(dz0 + r * C + 4 * lane);
            const float cv = cosm[r];
            float dot = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) { accP[k] += v[k]; accW[k] = __builtin_fmaf(v[k], cv, accW[k]); dot = __builtin_fmaf(v[k], w4[k], dot); }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) dot += __shfl_xor(dot, off, 64);
            if (lane == 0) dcos[r] = dot;
        }
        if (on) *reinterpret_cast<f32x4t*>(dP + (size_t)bi * C + 4 * lane) = accP;
    }
    // dw: the four waves' sums in wave order -> one partial row per workgroup
#pragma unroll
    for (int k = 0; k < 4; ++k) red[wv][4 * lane + k] = on ? accW[k] : 0.f;
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) dw_partial[(size_t)blockIdx.x * C + c] = ((red[0][c] + red[1][c]) + red[2][c]) + red[3][c];
}

// The same pass when dz0 does not exist yet: layer 0's BatchNorm + ReLU backward APPLIED ON THE FLY (round 4). G = the gradient
// w.r.t. relu(BatchNorm(z0)); z0 is recomputed from P / cos / w exactly as the forward pass formed it (one fma), so neither z0 nor
// dz0 is read or written: dz0 = gamma * invstd * (dy - s0 / R - xhat * s1 / R), dy = G where z0 * a + b > 0 (bn_bwd_apply4_kernel's
// arithmetic, bit for bit). Replaces the apply pass (read G, read z0, write dz0) + the pass above (read dz0): 4 tensor passes -> 1.
__global__ __launch_bounds__(256) void xcorr_z0_bnbwd_kernel(const float* __restrict__ G, const float* __restrict__ P,
                                                             const float* __restrict__ cosm, const float* __restrict__ w,
                                                             const float* __restrict__ mean, const float* __restrict__ invstd,
                                                             const float* __restrict__ gamma, const float* __restrict__ s0,
                                                             const float* __restrict__ s1, const float* __restrict__ act_a,
                                                             const float* __restrict__ act_b, float rinv, int B, int n2, int n1, int C,
                                                             float* __restrict__ dP, float* __restrict__ dcos,
                                                             float* __restrict__ dw_partial) {
    __shared__ float red[4][256];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int bi = blockIdx.x * 4 + wv;                        // flat (b, i)
    const int Cq = C >> 2;
    const bool on = lane < Cq && bi < B * n1;
    f32x4t accP = {0.f, 0.f, 0.f, 0.f}, accW = {0.f, 0.f, 0.f, 0.f};
    if (bi < B * n1) {
        const int b = bi / n1, i = bi - b * n1;
        f32x4t w4 = {0.f, 0.f, 0.f, 0.f}, pv = w4, mu = w4, is = w4, k0 = w4, a0 = w4, a1 = w4, ca = w4, cb = w4;
        if (on) {
            w4 = *reinterpret_cast<const f32x4t*>(w + 4 * lane);
            pv = *reinterpret_cast<const f32x4t*>(P + (size_t)bi * C + 4 * lane);
            mu = *reinterpret_cast<const f32x4t*>(mean + 4 * lane);
            is = *reinterpret_cast<const f32x4t*>(invstd + 4 * lane);
            k0 = *reinterpret_cast<const f32x4t*>(gamma + 4 * lane);
            a0 = *reinterpret_cast<const f32x4t*>(s0 + 4 * lane);
            a1 = *reinterpret_cast<const f32x4t*>(s1 + 4 * lane);
            ca = *reinterpret_cast<const f32x4t*>(act_a + 4 * lane);
            cb = *reinterpret_cast<const f32x4t*>(act_b + 4 * lane);
        }
        auto dz_of = [&](const f32x4t& g, float cv) {
            f32x4t d;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float z = __builtin_fmaf(cv, w4[k], pv[k]);                      // xcorr_z0_kernel's z0
                const float dy = __builtin_fmaf(z, ca[k], cb[k]) > 0.f ? g[k] : 0.f;   // bn_bwd_apply4_kernel from here on
                const float xh = (z - mu[k]) * is[k];
                d[k] = k0[k] * is[k] * (dy - a0[k] * rinv - xh * (a1[k] * rinv));
            }
            return d;
        };
        int j = 0;
        for (; j + 3 < n2; j += 4) {
            f32x4t g[4];
            float cv[4], dot[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long long r = ((long long)b * n2 + j + u) * n1 + i;
                g[u] = f32x4t{0.f, 0.f, 0.f, 0.f};
                if (on) g[u] = *reinterpret_cast<const f32x4t*>(G + r * C + 4 * lane);
                cv[u] = cosm[r];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                f32x4t v = {0.f, 0.f, 0.f, 0.f};
                if (on) v = dz_of(g[u], cv[u]);
                dot[u] = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) { accP[k] += v[k]; accW[k] = __builtin_fmaf(v[k], cv[u], accW[k]); dot[u] = __builtin_fmaf(v[k], w4[k], dot[u]); }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1)
#pragma unroll
                for (int u = 0; u < 4; ++u) dot[u] += __shfl_xor(dot[u], off, 64);
            if (lane < 4) dcos[((long long)b * n2 + j + lane) * n1 + i] = lane == 0 ? dot[0] : lane == 1 ? dot[1] : lane == 2 ? dot[2] : dot[3];
        }
        for (; j < n2; ++j) {
            const long long r = ((long long)b * n2 + j) * n1 + i;
            const float cv = cosm[r];
            f32x4t v = {0.f, 0.f, 0.f, 0.f};
            if (on) v = dz_of(*reinterpret_cast<const f32x4t*>(G + r * C + 4 * lane), cv);
            float dot = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) { accP[k] += v[k]; accW[k] = __builtin_fmaf(v[k], cv, accW[k]); dot = __builtin_fmaf(v[k], w4[k], dot); }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) dot += __shfl_xor(dot, off, 64);
            if (lane == 0) dcos[r] = dot;
        }
        if (on) *reinterpret_cast<f32x4t*>(dP + (size_t)bi * C + 4 * lane) = accP;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) red[wv][4 * lane + k] = on ? accW[k] : 0.f;
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) dw_partial[(size_t)blockIdx.x * C + c] = ((red[0][c] + red[1][c]) + red[2][c]) + red[3][c];
}

static inline bool vec4_ok(const void* p, int ld, int C) {
    return (C & 3) == 0 && (ld & 3) == 0 && (reinterpret_cast<uintptr_t>(p) & 15) == 0;
}

// X = relu((Z - mean) * invstd * gamma + beta), element-wise over (R, C) rows; relu == 0: no activation.
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ Z, int ldz, const float* __restrict__ mean,
                                                       const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, int R, int C, int relu,
                                                       float* __restrict__ X, int ldx) {
    const size_t total = (size_t)R * C;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int r = (int)(e / C), c = (int)(e - (size_t)r * C);
        float y = (Z[(size_t)r * ldz + c] - mean[c]) * invstd[c] * gamma[c] + beta[c];
        if (relu) y = fmaxf(y, 0.f);
        X[(size_t)r * ldx + c] = y;
    }
}

// dZ = gamma * invstd * (dy - s0 / R - xhat * s1 / R), dy = g masked by the ReLU (act > 0) — BatchNorm backward in
// training mode (batch statistics depend on the input).
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ G, int ldg, const float* __restrict__ Act,
                                                           int lda, const float* __restrict__ Z, int ldz,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ s0,
                                                           const float* __restrict__ s1, int R, int C, int relu,
                                                           float* __restrict__ dZ, int ldd) {
    const size_t total = (size_t)R * C;
    const float rinv = 1.0f / (float)R;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int r = (int)(e / C), c = (int)(e - (size_t)r * C);
        float dy = G[(size_t)r * ldg + c];
        if (relu && !(Act[(size_t)r * lda + c] > 0.f)) dy = 0.f;
        const float xh = (Z[(size_t)r * ldz + c] - mean[c]) * invstd[c];
        dZ[(size_t)r * ldd + c] = gamma[c] * invstd[c] * (dy - s0[c] * rinv - xh * (s1[c] * rinv));
    }
}

// out[g, c] = max over the ns consecutive rows of group g; arg[g, c] = the first row that attains it.
__global__ __launch_bounds__(256) void pool_rows_kernel(const float* __restrict__ X, int ldx, int G, int ns, int C,
                                                        float* __restrict__ out, int ldo, int32_t* __restrict__ arg,
                                                        const float* __restrict__ act_a, const float* __restrict__ act_b) {
    const size_t total = (size_t)G * C;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int g = (int)(e / C), c = (int)(e - (size_t)g * C);
        const float* p = X + (size_t)g * ns * ldx + c;
        const float sa = act_a ? act_a[c] : 1.f, sb = act_a ? act_b[c] : 0.f;     // optional relu(x * a + b) on the fly
        float m = act_a ? fmaxf(__builtin_fmaf(p[0], sa, sb), 0.f) : p[0];
        int a = 0;
        for (int k = 1; k < ns; ++k) {
            float v = p[(size_t)k * ldx];
            if (act_a) v = fmaxf(__builtin_fmaf(v, sa, sb), 0.f);
            if (v > m) { m = v; a = k; }
        }
        out[(size_t)g * ldo + c] = m;
        arg[(size_t)g * C + c] = a;
    }
}

// The max-pool of relu(y * a + b) over groups of rows from the per-group extrema the GEMM's epilogue took (ptt_rows_gemm_pool_f32):
// relu(a y + b) is monotone in y — increasing for a > 0, decreasing for a < 0 — so its maximum over a group is at the group's
// largest / smallest y, and the first row holding that extremum is the first arg-max (a == 0: every row ties, row 0). Same
// value and the same gradient routing as pool_rows_kernel (an arg-max whose activation is 0 gets no gradient either way).
__global__ __launch_bounds__(256) void pool_select_kernel(const float* __restrict__ pmax, const float* __restrict__ pmin,
                                                          const int32_t* __restrict__ amax, const int32_t* __restrict__ amin,
                                                          const float* __restrict__ act_a, const float* __restrict__ act_b, size_t total,
                                                          int C, float* __restrict__ out, int32_t* __restrict__ arg) {
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int c = (int)(e % C);
        const float a = act_a[c], b = act_b[c];
        const float y = a > 0.f ? pmax[e] : pmin[e];
        out[e] = fmaxf(__builtin_fmaf(y, a, b), 0.f);
        arg[e] = a > 0.f ? amax[e] : (a < 0.f ? amin[e] : 0);
    }
}

// dX[g*ns + k, c] = dOut[g, c] if k == arg[g, c] else 0 (every element of dX is written).
__global__ __launch_bounds__(256) void pool_rows_bwd_kernel(const float* __restrict__ dOut, int ldo, const int32_t* __restrict__ arg,
                                                            int G, int ns, int C, float* __restrict__ dX, int ldx) {
    const size_t total = (size_t)G * ns * C;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t row = e / C;
        const int c = (int)(e - row * C);
        const int g = (int)(row / ns), k = (int)(row - (size_t)g * ns);
        dX[row * ldx + c] = (arg[(size_t)g * C + c] == k) ? dOut[(size_t)g * ldo + c] : 0.f;
    }
}

// ------------------------------------------------------------------------------------------
// Weight gradient of a row-wise linear layer: dW[o, i] = sum_r dZ[r, o] * X[r, i]  (and db[o] = sum_r dZ[r, o]).
// A GEMM whose reduction axis is the ROW axis (hundreds of thousands of rows, <= 512 x 512 outputs): split over row
// chunks, one workgroup = one 128 x 128 output block of one chunk, 4 waves x (2 x 2) MFMA tiles of 32 x 32,
// v_mfma_f32_32x32x2_f32 (exact fp32). Both operands are k-major in memory (row r holds all channels), which is the
// MFMA's native operand order for this product: lane (k = lane >> 5, m = lane & 31) reads dZ[r0 + 2j + k][o0 + m] —
// 32 consecutive floats, conflict-free from the LDS tile. Partials go to a workspace and are summed in chunk order.
// ------------------------------------------------------------------------------------------
constexpr int WG_KC = 32;            // rows per staged sub-chunk
constexpr int WG_LD = 132;           // LDS row stride (floats)
constexpr int WG_ROWS = 4096;        // rows per workgroup at most (fewer when that leaves CUs idle)
constexpr int WG_MIN_ROWS = 64;         // (256 until round 4: a 6144-row layer was 96 workgroups of 8 stages, 28 us on a third of the CUs)

template <bool VEC>   // VEC: ld % 4 == 0 and P 16-byte aligned -> whole float4s (channels past cmax are zeroed afterwards)
__device__ __forceinline__ void wg_fetch(const float* __restrict__ P, int ld, int r0, int rmax, int c0, int cmax, int t,
                                         f32x4t (&st)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = t + i * 256;                  // float4 slot in the [32][128] block
        const int r = r0 + (e >> 5), c = c0 + ((e & 31) << 2);
        f32x4t v = {0.f, 0.f, 0.f, 0.f};
        if (r < rmax) {
            const float* src = P + (size_t)r * ld + c;
            if (VEC && c + 3 < cmax) {
                v = *reinterpret_cast<const f32x4t*>(src);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (c + q < cmax) v[q] = src[q];
            }
        }
        st[i] = v;
    }
}
__device__ __forceinline__ void wg_stage(float* S, int t, const f32x4t (&st)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = t + i * 256;
        *reinterpret_cast<f32x4t*>(S + (e >> 5) * WG_LD + ((e & 31) << 2)) = st[i];
    }
}

// relu(x * a + b) on the staged X block (training: the layer input = previous layer's BatchNorm + ReLU, never materialised)
__device__ __forceinline__ void wg_act(f32x4t (&st)[4], const float* __restrict__ xa, const float* __restrict__ xb, int c0, int cmax,
                                       int t) {
    const int c = c0 + ((t & 31) << 2);
    if (c + 3 < cmax) {
        const f32x4t a4 = *reinterpret_cast<const f32x4t*>(xa + c), b4 = *reinterpret_cast<const f32x4t*>(xb + c);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) st[i][q] = fmaxf(__builtin_fmaf(st[i][q], a4[q], b4[q]), 0.f);
    }
}

// NS > 1 (round 3): a layer with at most 64 output and / or input channels (SA0: 64 -> 64, 64 -> 128 over 786k rows) fills
// one or two of the block's four 64 x 64 quadrants; instead of three (two) waves multiplying zero padding — 4x (2x) the MFMA
// work, which made these memory-bound shapes compute-bound at 222 us — the waves that share a quadrant split the staged rows
// (NS = 2 or 4 ways) and their accumulators are added through LDS in wave order at the end.
template <bool VZ, bool VX, int NS = 1>
__global__ __launch_bounds__(256, 2) void linear_wgrad_kernel(const float* __restrict__ dZ, int ldz, const float* __restrict__ X,
                                                              int ldx, int R, int Cout, int Cin, int nbi, int chunk_rows,
                                                              float* __restrict__ partial, const float* __restrict__ xa,
                                                              const float* __restrict__ xb, int narrow_o = 0, int narrow_i = 0) {
    extern __shared__ __attribute__((aligned(16))) float wg_smem[];         // As[2][32 x 132] | Bs[2][32 x 132]: 67.6 KB
    constexpr int WG_BUF = WG_KC * WG_LD;
#define As(b) (wg_smem + (b) * WG_BUF)
#define Bs(b) (wg_smem + (2 + (b)) * WG_BUF)
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, half = lane >> 5, col = lane & 31;
    const int bo = blockIdx.x / nbi, bi = blockIdx.x - bo * nbi;       // 128-wide output-channel / input-channel block
    const int o0 = bo * 128, i0 = bi * 128;
    const int r_begin = blockIdx.y * chunk_rows, r_end = min(R, r_begin + chunk_rows);
    int wo = (w >> 1) * 64, wi = (w & 1) * 64;                         // this wave's 64 x 64 quadrant
    int sidx = 0;                                                      // NS > 1: which share of the staged rows
    if constexpr (NS > 1) {
        if (narrow_o) { sidx = w >> 1; wo = 0; }
        if (narrow_i) { sidx = sidx * 2 + (w & 1); wi = 0; }
        sidx = __builtin_amdgcn_readfirstlane(sidx);
    }
    f32x16t acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    f32x4t sa[4], sb[4];
    wg_fetch<VZ>(dZ, ldz, r_begin, r_end, o0, Cout, t, sa);
    wg_fetch<VX>(X, ldx, r_begin, r_end, i0, Cin, t, sb);
    if (xa) wg_act(sb, xa, xb, i0, Cin, t);
    wg_stage(As(0), t, sa);
    wg_stage(Bs(0), t, sb);
    __syncthreads();
    int buf = 0;
    for (int r0 = r_begin; r0 < r_end; r0 += WG_KC) {
        const bool more = r0 + WG_KC < r_end;
        if (more) {
            wg_fetch<VZ>(dZ, ldz, r0 + WG_KC, r_end, o0, Cout, t, sa);
            wg_fetch<VX>(X, ldx, r0 + WG_KC, r_end, i0, Cin, t, sb);
            if (xa) wg_act(sb, xa, xb, i0, Cin, t);
        }
        const float* A = As(buf) + (half + sidx * (WG_KC / NS)) * WG_LD + wo + col;
        const float* B = Bs(buf) + (half + sidx * (WG_KC / NS)) * WG_LD + wi + col;
        // the fragments of step j + 1 are requested before the MFMAs of step j (two register sets; see wgrad2_kernel, gemm_ops.hip)
        float fa[2][2], fb[2][2];
        fa[0][0] = A[0]; fa[0][1] = A[32]; fb[0][0] = B[0]; fb[0][1] = B[32];
#pragma unroll
        for (int j = 0; j < WG_KC / 2 / NS; ++j) {
            const int s = j & 1;
            if (j + 1 < WG_KC / 2 / NS) {
                fa[s ^ 1][0] = A[2 * (j + 1) * WG_LD]; fa[s ^ 1][1] = A[2 * (j + 1) * WG_LD + 32];
                fb[s ^ 1][0] = B[2 * (j + 1) * WG_LD]; fb[s ^ 1][1] = B[2 * (j + 1) * WG_LD + 32];
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s][0], fb[s][0], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s][0], fb[s][1], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s][1], fb[s][0], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s][1], fb[s][1], acc[1][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (more) {
            wg_stage(As(buf ^ 1), t, sa);
            wg_stage(Bs(buf ^ 1), t, sb);
        }
        __syncthreads();
        buf ^= 1;
    }
#undef As
#undef Bs
    if constexpr (NS > 1) {
        // the shares of a quadrant, added in wave order (the loop's last barrier has released the staging buffers). The
        // accumulators are only STORED here and the sum is formed from the LDS copies: accumulators that are also written after the
        // loop get copied to vector registers wholesale (256 VGPRs + 20 - 24 B of scratch per lane until round 4)
        float* slot = wg_smem + w * 4096;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) slot[((a * 2 + b) * 16 + r) * 64 + lane] = acc[a][b][r];
        __syncthreads();
        if (sidx != 0) return;
        const int step = (NS == 4) ? 1 : (narrow_i ? 1 : 2);           // the waves that share this wave's quadrant
        float* P = partial + (size_t)blockIdx.y * Cout * Cin;
        for (int a = 0; a < 2; ++a)
            for (int b = 0; b < 2; ++b) {
                const int ci = i0 + wi + b * 32 + col;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int e = ((a * 2 + b) * 16 + r) * 64 + lane;
                    float v = slot[e];
                    for (int k = 1; k < NS; ++k) v += wg_smem[(w + k * step) * 4096 + e];
                    const int co = o0 + wo + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (co < Cout && ci < Cin) P[(size_t)co * Cin + ci] = v;
                }
            }
        return;
    }
    // C/D layout: column (input channel) = lane & 31, row (output channel) = (reg & 3) + 8 * (reg >> 2) + 4 * half
    float* P = partial + (size_t)blockIdx.y * Cout * Cin;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int ci = i0 + wi + b * 32 + col;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = o0 + wo + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (co < Cout && ci < Cin) P[(size_t)co * Cin + ci] = acc[a][b][r];
            }
        }
}

// the same for rows that are not float4-addressable (C % 4, row stride % 4 or an unaligned base: the 3 + 256 = 259-channel rows of
// the vote layer): thread = (row group, column), scalar loads, the same chunking and combination order
__global__ __launch_bounds__(256) void colsum_partial_scalar_kernel(const float* __restrict__ X, int ldx, int R, int C, int rows_per_chunk,
                                                                    float* __restrict__ partial) {
    __shared__ float red[4][64];
    const int c = blockIdx.y * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
    const int r0 = blockIdx.x * rows_per_chunk, r1 = min(R, r0 + rows_per_chunk);
    float acc = 0.f;
    if (c < C)
        for (int r = r0 + rg; r < r1; r += 4) acc += X[(size_t)r * ldx + c];
    red[rg][threadIdx.x & 63] = acc;
    __syncthreads();
    if (rg == 0 && c < C) partial[(size_t)blockIdx.x * C + c] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

// Weight gradient of a layer with at most 4 INPUT channels (the relative-coordinate terms of the hoisted layer 0, fc_delta[0]:
// K = 3): dW[c][k] = sum_r dZ[r][c] * X[r][k]. On the MFMA kernel above that is a 128 x 128 tile with 125 idle columns and two
// staged operands; it is a memory-bound pass over dZ: thread = (row slot, channel quad), the K coordinates of a row are a
// wave-uniform load, partial sums per 256-row chunk combined through LDS in slot order, chunks summed by wgrad_finish_kernel.
constexpr int WK_ROWS = 256;
template <int K>
__global__ __launch_bounds__(256) void wgrad_smallk_kernel(const float* __restrict__ dZ, int ldz, const float* __restrict__ X, int ldx,
                                                           int R, int C, float* __restrict__ partial) {
    __shared__ float red[256][4 * K];
    const int Cq = C >> 2, span = Cq < 256 ? Cq : 256, RG = 256 / span;
    const int q0 = threadIdx.x % span, rg = threadIdx.x / span;
    const int r0 = blockIdx.x * WK_ROWS, r1 = min(R, r0 + WK_ROWS);
    for (int qb = 0; qb < Cq; qb += span) {
        const int q = qb + q0;
        float acc[4][K];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < K; ++k) acc[j][k] = 0.f;
        if (q < Cq && rg < RG) {
            for (int r = r0 + rg; r < r1; r += RG) {
                const f32x4t g = *reinterpret_cast<const f32x4t*>(dZ + (size_t)r * ldz + 4 * q);
                float xv[K];
#pragma unroll
                for (int k = 0; k < K; ++k) xv[k] = X[(size_t)r * ldx + k];
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int k = 0; k < K; ++k) acc[j][k] = __builtin_fmaf(g[j], xv[k], acc[j][k]);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < K; ++k) red[threadIdx.x][j * K + k] = acc[j][k];
        __syncthreads();
        if (rg == 0 && q < Cq) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    float t = 0.f;
                    for (int g2 = 0; g2 < RG; ++g2) t += red[g2 * span + q0][j * K + k];
                    partial[((size_t)blockIdx.x * C + 4 * q + j) * K + k] = t;
                }
        }
        __syncthreads();
    }
}

// dW = sum over chunks (fixed order); optionally accumulates into dW (beta = 1) for parameters used more than once.
// Workgroup = 32 consecutive outputs x 8 chunk groups: group g adds the chunks g, g + 8, ... in order, the 8 group sums are
// then added in order — a fixed summation tree, and 8x the loads in flight of a one-thread-per-output loop.
// Column sums of a (rows, C) tensor — the bias gradient of a row-wise layer (sum over the rows of dY; torch's reduce takes 27 us for
// the 6144-row layers of a step). One workgroup per (row chunk, block of 64 channel quads): 4 row lanes stride the chunk's rows with
// coalesced float4 loads, combined through LDS in lane order; the chunk partials are summed in chunk order by wgrad_finish_kernel
// (deterministic, as every reduction of the training step).
constexpr int CS_ROWS = 128;          // rows per chunk at least
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ X, int ldx, int R, int C, int rows_per_chunk,
                                                             float* __restrict__ partial) {
    __shared__ f32x4t red[4][64];
    const int q = blockIdx.y * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
    const int r0 = blockIdx.x * rows_per_chunk, r1 = min(R, r0 + rows_per_chunk);
    f32x4t acc = {0.f, 0.f, 0.f, 0.f};
    if (4 * q < C) {
        int r = r0 + rg;
        for (; r + 12 < r1; r += 16) {
            const f32x4t a = *reinterpret_cast<const f32x4t*>(X + (size_t)r * ldx + 4 * q);
            const f32x4t b = *reinterpret_cast<const f32x4t*>(X + (size_t)(r + 4) * ldx + 4 * q);
            const f32x4t c = *reinterpret_cast<const f32x4t*>(X + (size_t)(r + 8) * ldx + 4 * q);
            const f32x4t d = *reinterpret_cast<const f32x4t*>(X + (size_t)(r + 12) * ldx + 4 * q);
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = (((acc[k] + a[k]) + b[k]) + c[k]) + d[k];
        }
        for (; r < r1; r += 4) {
            const f32x4t a = *reinterpret_cast<const f32x4t*>(X + (size_t)r * ldx + 4 * q);
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] += a[k];
        }
    }
    red[rg][threadIdx.x & 63] = acc;
    __syncthreads();
    if (rg == 0 && 4 * q < C) {
        f32x4t t = red[0][threadIdx.x];
#pragma unroll
        for (int g = 1; g < 4; ++g)
#pragma unroll
            for (int k = 0; k < 4; ++k) t[k] += red[g][threadIdx.x][k];
        *reinterpret_cast<f32x4t*>(partial + (size_t)blockIdx.x * C + 4 * q) = t;
    }
}

// OUT = outputs per workgroup (32, 8 or 4): many chunks over few outputs (the K = 3 gradients: 3072 chunks x 192 outputs) take
// more groups per output — 74 us -> a few for that shape.
template <int OUT>
__global__ __launch_bounds__(256) void wgrad_finish_kernel(const float* __restrict__ partial, int nchunks, size_t n, int accumulate,
                                                           float* __restrict__ dW) {
    constexpr int G = 256 / OUT;
    __shared__ float part[G][OUT];
    const int lane = threadIdx.x % OUT, g = threadIdx.x / OUT;
    for (size_t base = (size_t)blockIdx.x * OUT; base < n; base += (size_t)gridDim.x * OUT) {
        const size_t e = base + lane;
        float s = 0.f;
        if (e < n)
            for (int k = g; k < nchunks; k += G) s += partial[(size_t)k * n + e];
        part[g][lane] = s;
        __syncthreads();
        if (g == 0 && e < n) {
            float tot = part[0][lane];
            for (int q = 1; q < G; ++q) tot += part[q][lane];
            dW[e] = accumulate ? dW[e] + tot : tot;
        }
        __syncthreads();
    }
}

// Every parameter gradient of a training step in ONE launch (ptt_grad_finish_f32). The weight-gradient kernels above leave their
// row-chunk partials where they are (ptt_linear_wgrad*_partials_f32, ptt_colsum_partials_f32) instead of each being followed by
// its own wgrad_finish_kernel launch — 75 launches of 5 - 10 us per step, most of them a few KB of output — and by autograd's
// add of the two branches' gradients of a shared weight. A SEGMENT is one destination (a parameter, or a column slice of one,
// inside the flat gradient buffer) with the JOBS that contribute to it, in the order the backward pass issued them; chunk k of
// the concatenated chunk list goes to thread group k % G, groups are folded in group order, the result is added to what the
// destination holds: a fixed summation tree per (job list), bit-reproducible run to run. blocks[2 w] = segment of workgroup w,
// blocks[2 w + 1] = its first output unit (a unit = 4 elements of a `vec` segment, else 1).
__global__ __launch_bounds__(256) void grad_finish_kernel(const ptt_grad_segment* __restrict__ segs, const ptt_grad_job* __restrict__ jobs,
                                                          const int32_t* __restrict__ blocks, float* __restrict__ flat) {
    __shared__ f32x4t part[256];
    const ptt_grad_segment sg = segs[blocks[2 * blockIdx.x]];
    const int OUT = sg.out, G = 256 / OUT;
    const int lane = threadIdx.x % OUT, g = threadIdx.x / OUT;
    const int e = (blocks[2 * blockIdx.x + 1] + lane) * (sg.vec ? 4 : 1);
    f32x4t s = {0.f, 0.f, 0.f, 0.f};
    if (e < sg.n) {
        int kbase = 0;
        for (int j = 0; j < sg.njobs; ++j) {
            const ptt_grad_job jb = jobs[sg.job0 + j];
            int c = g - kbase % G;
            if (c < 0) c += G;
            if (sg.vec) {
                const float* pe = jb.partial + e;
                const size_t step = (size_t)G * sg.n;
                for (; c + 3 * G < jb.nchunks; c += 4 * G) {     // four loads in flight, summed in chunk order
                    const float* p0 = pe + (size_t)c * sg.n;
                    const f32x4t v0 = *reinterpret_cast<const f32x4t*>(p0), v1 = *reinterpret_cast<const f32x4t*>(p0 + step),
                                 v2 = *reinterpret_cast<const f32x4t*>(p0 + 2 * step), v3 = *reinterpret_cast<const f32x4t*>(p0 + 3 * step);
#pragma unroll
                    for (int x = 0; x < 4; ++x) s[x] = (((s[x] + v0[x]) + v1[x]) + v2[x]) + v3[x];
                }
                for (; c < jb.nchunks; c += G) {
                    const f32x4t v = *reinterpret_cast<const f32x4t*>(pe + (size_t)c * sg.n);
#pragma unroll
                    for (int x = 0; x < 4; ++x) s[x] += v[x];
                }
            } else {
                for (; c < jb.nchunks; c += G) s[0] += jb.partial[(size_t)c * sg.n + e];
            }
            kbase += jb.nchunks;
        }
    }
    part[g * OUT + lane] = s;
    __syncthreads();
    if (g == 0 && e < sg.n) {
        f32x4t tot = part[lane];
        for (int q = 1; q < G; ++q)
#pragma unroll
            for (int x = 0; x < 4; ++x) tot[x] += part[q * OUT + lane][x];
        const int row = e / sg.cols, col = e - row * sg.cols;
        float* d = flat + sg.dst + (long long)row * sg.ld + col;
        if (sg.vec) {
            f32x4t o = *reinterpret_cast<const f32x4t*>(d);
#pragma unroll
            for (int x = 0; x < 4; ++x) o[x] += tot[x];
            *reinterpret_cast<f32x4t*>(d) = o;
        } else {
            *d += tot[0];
        }
    }
}

// out[b, e, :] = src[b, idx[b, e], :] over point-major rows (float4 quads; C % 4 == 0): the forward of grouping once the
// first MLP layer has been evaluated per POINT (the training-mode layer-0 hoist, ptt_amd/train_ops.py).
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, const int32_t* __restrict__ idx, int N,
                                                          int E, int C, float* __restrict__ out) {
    const int Cq = C >> 2, span = Cq < 256 ? Cq : 256, RG = 256 / span;
    const int q0 = threadIdx.x % span, rg = threadIdx.x / span;
    if (rg >= RG) return;
    const int b = blockIdx.y;
    const float* sb = src + (size_t)b * N * C;
    float* ob = out + (size_t)b * E * C;
    const int32_t* ib = idx + (size_t)b * E;
    for (int e = blockIdx.x * RG + rg; e < E; e += gridDim.x * RG) {
        const int n = ib[e];
        for (int q = q0; q < Cq; q += span)
            *reinterpret_cast<f32x4t*>(ob + (size_t)e * C + 4 * q) = *reinterpret_cast<const f32x4t*>(sb + (size_t)n * C + 4 * q);
    }
}

// The whole front of a hoisted SA level in training mode (train_ops.sa_level_hoisted) in one pass: per (centre m, slot k) row
//   rel = (xyz[b, idx[b,m,k]] - new_xyz[b,m]) (* 1 / radius)        QueryAndGroup, pointnet2_utils.py:350-354; torch divides a
//                                                                    tensor by a host scalar as a product with its float reciprocal
//   z0  = term[b, idx[b,m,k], :] + Wx . rel                          layer 0 of the SharedMLP, its feature half hoisted per POINT
// (term NULL: a level without point features, z0 = Wx . rel is the whole first convolution). rel rows are written too: the
// weight gradient of Wx needs them. Replaces group(xyz) / subtract / divide / permute-copy / gather_rows / a K = 3 GEMM with
// the gathered rows as residual — seven launches and two extra passes over the (rows, C0) tensor.
__global__ __launch_bounds__(256) void sa_z0_rows_kernel(const float* __restrict__ xyz, const float* __restrict__ new_xyz,
                                                         const int32_t* __restrict__ idx, const float* __restrict__ term,
                                                         const float* __restrict__ wx, int ldw, int N, int M, int ns, int C,
                                                         float rmul, float* __restrict__ z0, float* __restrict__ rel_out,
                                                         double* __restrict__ stats) {
#pragma clang fp contract(off)      // the subtraction / scaling are the reference's separately rounded element-wise steps
    // stats (optional, C <= 1024): the float64 column sums / sums of squares of the rows this workgroup writes, partial
    // [workgroup][2][C] (workgroup = blockIdx.y * gridDim.x + blockIdx.x), the format ptt_bn_finish_partials_* combine in order:
    // the BatchNorm statistics of layer 0 without a pass of their own over z0
    __shared__ double red[2][256][4];
    const int Cq = C >> 2, span = Cq < 256 ? Cq : 256, RG = 256 / span;
    const int q0 = threadIdx.x % span, rg = threadIdx.x / span;
    const int b = blockIdx.y, E = M * ns;
    const float* xb = xyz + (size_t)b * N * 3;
    const float* cb = new_xyz + (size_t)b * M * 3;
    const float* tb = term ? term + (size_t)b * N * C : nullptr;
    const int32_t* ib = idx + (size_t)b * E;
    float* zb = z0 + (size_t)b * E * C;
    float* rb = rel_out + (size_t)b * E * 3;
    double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
    if (rg < RG) {
        for (int e = blockIdx.x * RG + rg; e < E; e += gridDim.x * RG) {
            const int n = ib[e], m = e / ns;
            const float rx = (xb[3 * n + 0] - cb[3 * m + 0]) * rmul, ry = (xb[3 * n + 1] - cb[3 * m + 1]) * rmul,
                        rz = (xb[3 * n + 2] - cb[3 * m + 2]) * rmul;
            if (q0 == 0) { rb[3 * e + 0] = rx; rb[3 * e + 1] = ry; rb[3 * e + 2] = rz; }
            for (int q = q0; q < Cq; q += span) {
                f32x4t v = tb ? *reinterpret_cast<const f32x4t*>(tb + (size_t)n * C + 4 * q) : f32x4t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float* w = wx + (size_t)(4 * q + j) * ldw;                 // Wx (C,3), row stride ldw
                    const float y = __builtin_fmaf(w[2], rz, __builtin_fmaf(w[1], ry, w[0] * rx));
                    v[j] = v[j] + y;
                    if (stats) { s1[j] += (double)v[j]; s2[j] += (double)v[j] * (double)v[j]; }
                }
                *reinterpret_cast<f32x4t*>(zb + (size_t)e * C + 4 * q) = v;
            }
        }
    }
    if (!stats) return;
#pragma unroll
    for (int j = 0; j < 4; ++j) { red[0][threadIdx.x][j] = s1[j]; red[1][threadIdx.x][j] = s2[j]; }
    __syncthreads();
    if (rg == 0) {                                   // the row groups' sums in group order (span == Cq here: one quad per thread)
        double* sp = stats + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 2 * C;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double t1 = 0.0, t2 = 0.0;
            for (int g = 0; g < RG; ++g) { t1 += red[0][g * span + q0][j]; t2 += red[1][g * span + q0][j]; }
            sp[4 * q0 + j] = t1; sp[C + 4 * q0 + j] = t2;
        }
    }
}

// Its backward, deterministic: out[b, n, :] = sum of g[b, e, :] over the entries e with idx[b, e] == n, in ASCENDING e
// (order / start = the CSR scatter_csr_kernel builds). Thread = (point n, channel quad): coalesced row reads.
// negate: out = minuend - sum (minuend (B, N, C), or 0 if it is null) instead of the sum.
__global__ __launch_bounds__(256) void scatter_rows_det_kernel(const float* __restrict__ g, const int32_t* __restrict__ order,
                                                               const int32_t* __restrict__ start, int N, int E, int C,
                                                               float* __restrict__ out, const float* __restrict__ minuend, int negate) {
    const int Cq = C >> 2, span = Cq < 256 ? Cq : 256, RG = 256 / span;
    const int q0 = threadIdx.x % span, rg = threadIdx.x / span;
    if (rg >= RG) return;
    const int b = blockIdx.y;
    const float* gb = g + (size_t)b * E * C;
    const int32_t* ob = order + (size_t)b * E;
    const int32_t* st = start + (size_t)b * (N + 1);
    for (int n = blockIdx.x * RG + rg; n < N; n += gridDim.x * RG) {
        const int k0 = st[n], k1 = st[n + 1];
        for (int q = q0; q < Cq; q += span) {
            f32x4t acc = {0.f, 0.f, 0.f, 0.f};
            int k = k0;
            for (; k + 3 < k1; k += 4) {                 // four rows in flight, added in entry order
                const int e0 = ob[k], e1 = ob[k + 1], e2 = ob[k + 2], e3 = ob[k + 3];
                const f32x4t v0 = *reinterpret_cast<const f32x4t*>(gb + (size_t)e0 * C + 4 * q);
                const f32x4t v1 = *reinterpret_cast<const f32x4t*>(gb + (size_t)e1 * C + 4 * q);
                const f32x4t v2 = *reinterpret_cast<const f32x4t*>(gb + (size_t)e2 * C + 4 * q);
                const f32x4t v3 = *reinterpret_cast<const f32x4t*>(gb + (size_t)e3 * C + 4 * q);
#pragma unroll
                for (int x = 0; x < 4; ++x) acc[x] = (((acc[x] + v0[x]) + v1[x]) + v2[x]) + v3[x];
            }
            for (; k < k1; ++k) {
                const f32x4t v = *reinterpret_cast<const f32x4t*>(gb + (size_t)ob[k] * C + 4 * q);
                acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3];
            }
            if (negate) {
                f32x4t m = {0.f, 0.f, 0.f, 0.f};
                if (minuend) m = *reinterpret_cast<const f32x4t*>(minuend + ((size_t)b * N + n) * C + 4 * q);
#pragma unroll
                for (int x = 0; x < 4; ++x) acc[x] = m[x] - acc[x];
            }
            *reinterpret_cast<f32x4t*>(out + ((size_t)b * N + n) * C + 4 * q) = acc;
        }
    }
}

// Layer 0 of a hoisted SA level, backward, from the gradient G w.r.t. relu(BatchNorm(z0)) as the input-gradient GEMM of layer 1 left
// it (round 4): ONE pass in row order forms dz0 row by row (bn_bwd_apply4_kernel's arithmetic on G and the stored z0), accumulates
// the K = 3 weight gradient d_wx[c, :] = sum over rows of dz0[row, c] * rel[row, :] (per-workgroup partials, summed in workgroup
// order) and writes dz0 only when somebody else needs it (dz_out: a level with point features, whose row scatter reads it; may
// alias G). Replaces the apply pass + wgrad_smallk_kernel's pass over dz0; a level without point features moves 2 tensors instead
// of 4. (A form that also did the row scatter in CSR order was 1.7x SLOWER inside the step than in isolation: ball-query padding
// makes a few points the neighbour of hundreds of centres, and the thread that owns such a bin walks it alone.)
__global__ __launch_bounds__(256) void sa_z0_bnbwd_kernel(const float* __restrict__ G, const float* __restrict__ Z0,
                                                          const float* __restrict__ rel, const float* __restrict__ mean,
                                                          const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                          const float* __restrict__ s0, const float* __restrict__ s1,
                                                          const float* __restrict__ act_a, const float* __restrict__ act_b, float rinv,
                                                          long long R, int C, int rows_per_wg, float* dz_out,
                                                          float* __restrict__ dwx_partial) {
    __shared__ float red[256][12];
    const int Cq = C >> 2, span = Cq < 256 ? Cq : 256, RG = 256 / span;          // C <= 1024: one channel quad per thread
    const int q = threadIdx.x % span, rg = threadIdx.x / span;
    float wacc[4][3];
#pragma unroll
    for (int x = 0; x < 4; ++x) wacc[x][0] = wacc[x][1] = wacc[x][2] = 0.f;
    if (rg < RG) {
        const f32x4t mu = *reinterpret_cast<const f32x4t*>(mean + 4 * q), is = *reinterpret_cast<const f32x4t*>(invstd + 4 * q);
        const f32x4t ga = *reinterpret_cast<const f32x4t*>(gamma + 4 * q);
        const f32x4t a0 = *reinterpret_cast<const f32x4t*>(s0 + 4 * q), a1 = *reinterpret_cast<const f32x4t*>(s1 + 4 * q);
        const f32x4t ca = *reinterpret_cast<const f32x4t*>(act_a + 4 * q), cb = *reinterpret_cast<const f32x4t*>(act_b + 4 * q);
        auto dz_of = [&](const f32x4t& g, const f32x4t& z) {
            f32x4t d;
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const float dy = __builtin_fmaf(z[x], ca[x], cb[x]) > 0.f ? g[x] : 0.f;
                const float xh = (z[x] - mu[x]) * is[x];
                d[x] = ga[x] * is[x] * (dy - a0[x] * rinv - xh * (a1[x] * rinv));
            }
            return d;
        };
        // the workgroup's rows r0 .. r1: row group rg takes rows r0 + rg, r0 + rg + RG, ... (a wave reads whole consecutive rows)
        const long long r0 = (long long)blockIdx.x * rows_per_wg, r1 = r0 + rows_per_wg < R ? r0 + rows_per_wg : R;
        long long r = r0 + rg;
        for (; r + 3LL * RG < r1; r += 4LL * RG) {       // four rows in flight, consumed in row order
            f32x4t g[4], z[4];
            float rl[4][3];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long long e = r + (long long)u * RG;
                g[u] = *reinterpret_cast<const f32x4t*>(G + e * C + 4 * q);
                z[u] = *reinterpret_cast<const f32x4t*>(Z0 + e * C + 4 * q);
                rl[u][0] = rel[e * 3 + 0]; rl[u][1] = rel[e * 3 + 1]; rl[u][2] = rel[e * 3 + 2];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const f32x4t d = dz_of(g[u], z[u]);
                if (dz_out) *reinterpret_cast<f32x4t*>(dz_out + (r + (long long)u * RG) * C + 4 * q) = d;
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    wacc[x][0] = __builtin_fmaf(d[x], rl[u][0], wacc[x][0]);
                    wacc[x][1] = __builtin_fmaf(d[x], rl[u][1], wacc[x][1]);
                    wacc[x][2] = __builtin_fmaf(d[x], rl[u][2], wacc[x][2]);
                }
            }
        }
        for (; r < r1; r += RG) {
            const f32x4t d = dz_of(*reinterpret_cast<const f32x4t*>(G + r * C + 4 * q), *reinterpret_cast<const f32x4t*>(Z0 + r * C + 4 * q));
            if (dz_out) *reinterpret_cast<f32x4t*>(dz_out + r * C + 4 * q) = d;
            const float l0 = rel[r * 3 + 0], l1 = rel[r * 3 + 1], l2 = rel[r * 3 + 2];
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                wacc[x][0] = __builtin_fmaf(d[x], l0, wacc[x][0]);
                wacc[x][1] = __builtin_fmaf(d[x], l1, wacc[x][1]);
                wacc[x][2] = __builtin_fmaf(d[x], l2, wacc[x][2]);
            }
        }
    }
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int j = 0; j < 3; ++j) red[threadIdx.x][x * 3 + j] = wacc[x][j];
    __syncthreads();
    if (rg == 0) {                                       // the row groups' sums in group order
        float* P = dwx_partial + (size_t)blockIdx.x * C * 3;
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                float t = 0.f;
                for (int g = 0; g < RG; ++g) t += red[g * span + q][x * 3 + j];
                P[(4 * q + x) * 3 + j] = t;
            }
    }
}

// ------------------------------------------------------------------------------------------
// Point-Transformer block in TRAINING mode: the element-wise chains around its GEMMs over the per-(point, neighbour)
// tensors (B,N,k,D) (transformer_block/variants.py:156-163), each as ONE pass. Thread = (point, channel quad); the k
// neighbours are a loop in registers, so the softmax over neighbours needs no cross-thread traffic.
//   pair_input:   t[b,i,j,:]  = q[b,i,:] - kf[b, knn[b,i,j], :] + pos[b,i,j,:]                          (:160 argument)
//   attn_fwd:     attn[b,i,j,:] = softmax_j(a[b,i,j,:] * scale);  res[b,i,:] = sum_j attn * (vf[b,knn] + pos)   (:161-163)
//   attn_bwd:     given dres: dvp[b,i,j,:] = attn * dres;  da = attn * (dres.vp_j - sum_j' attn_j' dres.vp_j') * scale
// ------------------------------------------------------------------------------------------
template <int KN>
__global__ __launch_bounds__(256) void pair_input_kernel(const float* __restrict__ q, const float* __restrict__ kf,
                                                         const int32_t* __restrict__ knn, const float* __restrict__ pos, int N,
                                                         int D, long long total_pts, float* __restrict__ t, int ldq, int ldk) {
    const int Dq = D >> 2;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total_pts * Dq; e += (long long)gridDim.x * 256) {
        const long long pt = e / Dq;
        const int c = (int)(e - pt * Dq) * 4;
        const long long b = pt / N;
        const f32x4t qv = *reinterpret_cast<const f32x4t*>(q + pt * ldq + c);
#pragma unroll 4
        for (int j = 0; j < KN; ++j) {
            const int n = knn[pt * KN + j];
            const f32x4t kv = *reinterpret_cast<const f32x4t*>(kf + (b * N + n) * ldk + c);
            const f32x4t pv = *reinterpret_cast<const f32x4t*>(pos + (pt * KN + j) * D + c);
            f32x4t o;
#pragma unroll
            for (int x = 0; x < 4; ++x) o[x] = (qv[x] - kv[x]) + pv[x];
            *reinterpret_cast<f32x4t*>(t + (pt * KN + j) * D + c) = o;
        }
    }
}

template <int KN>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const float* __restrict__ a, const float* __restrict__ vf,
                                                       const int32_t* __restrict__ knn, const float* __restrict__ pos, int N, int D,
                                                       long long total_pts, float scale, float* __restrict__ attn,
                                                       float* __restrict__ res, int ldv) {
    const int Dq = D >> 2;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total_pts * Dq; e += (long long)gridDim.x * 256) {
        const long long pt = e / Dq;
        const int c = (int)(e - pt * Dq) * 4;
        const long long b = pt / N;
        f32x4t av[KN];
        f32x4t m = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
#pragma unroll
        for (int j = 0; j < KN; ++j) {
            av[j] = *reinterpret_cast<const f32x4t*>(a + (pt * KN + j) * D + c);
#pragma unroll
            for (int x = 0; x < 4; ++x) { av[j][x] *= scale; m[x] = fmaxf(m[x], av[j][x]); }
        }
        f32x4t sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < KN; ++j)
#pragma unroll
            for (int x = 0; x < 4; ++x) { av[j][x] = __expf(av[j][x] - m[x]); sum[x] += av[j][x]; }
        f32x4t inv, acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int x = 0; x < 4; ++x) inv[x] = 1.0f / sum[x];
#pragma unroll
        for (int j = 0; j < KN; ++j) {
            const int n = knn[pt * KN + j];
            const f32x4t vv = *reinterpret_cast<const f32x4t*>(vf + (b * N + n) * ldv + c);
            const f32x4t pv = *reinterpret_cast<const f32x4t*>(pos + (pt * KN + j) * D + c);
            f32x4t w;
#pragma unroll
            for (int x = 0; x < 4; ++x) { w[x] = av[j][x] * inv[x]; acc[x] += w[x] * (vv[x] + pv[x]); }
            if (attn) *reinterpret_cast<f32x4t*>(attn + (pt * KN + j) * D + c) = w;
        }
        *reinterpret_cast<f32x4t*>(res + pt * D + c) = acc;
    }
}

template <int KN>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const float* __restrict__ attn, const float* __restrict__ vf,
                                                       const int32_t* __restrict__ knn, const float* __restrict__ pos,
                                                       const float* __restrict__ dres, int N, int D, long long total_pts, float scale,
                                                       float* __restrict__ da, float* __restrict__ dvp) {
    const int Dq = D >> 2;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total_pts * Dq; e += (long long)gridDim.x * 256) {
        const long long pt = e / Dq;
        const int c = (int)(e - pt * Dq) * 4;
        const long long b = pt / N;
        const f32x4t g = *reinterpret_cast<const f32x4t*>(dres + pt * D + c);
        f32x4t w[KN], dat[KN];
        f32x4t s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < KN; ++j) {
            const int n = knn[pt * KN + j];
            w[j] = *reinterpret_cast<const f32x4t*>(attn + (pt * KN + j) * D + c);
            const f32x4t vv = *reinterpret_cast<const f32x4t*>(vf + (b * N + n) * D + c);
            const f32x4t pv = *reinterpret_cast<const f32x4t*>(pos + (pt * KN + j) * D + c);
            f32x4t o;
#pragma unroll
            for (int x = 0; x < 4; ++x) { dat[j][x] = g[x] * (vv[x] + pv[x]); s[x] += w[j][x] * dat[j][x]; o[x] = w[j][x] * g[x]; }
            *reinterpret_cast<f32x4t*>(dvp + (pt * KN + j) * D + c) = o;
        }
#pragma unroll
        for (int j = 0; j < KN; ++j) {
            f32x4t o;
#pragma unroll
            for (int x = 0; x < 4; ++x) o[x] = w[j][x] * (dat[j][x] - s[x]) * scale;
            *reinterpret_cast<f32x4t*>(da + (pt * KN + j) * D + c) = o;
        }
    }
}

void launch_wgrad_finish(const float* partial, int nchunks, size_t n, int accumulate, float* dW, hipStream_t s) {
    const int out = nchunks > 1024 ? 4 : nchunks > 256 ? 8 : 32;       // a function of the chunk count only: fixed summation tree
    size_t fgrid = (n + out - 1) / out;
    if (fgrid > 4096) fgrid = 4096;
    if (out == 4) hipLaunchKernelGGL(wgrad_finish_kernel<4>, dim3((unsigned)fgrid), dim3(256), 0, s, partial, nchunks, n, accumulate, dW);
    else if (out == 8) hipLaunchKernelGGL(wgrad_finish_kernel<8>, dim3((unsigned)fgrid), dim3(256), 0, s, partial, nchunks, n, accumulate, dW);
    else hipLaunchKernelGGL(wgrad_finish_kernel<32>, dim3((unsigned)fgrid), dim3(256), 0, s, partial, nchunks, n, accumulate, dW);
}

static inline int ew_grid(size_t total) {
    size_t g = (total + 255) / 256;
    return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

}  // namespace ptt

using namespace ptt;

extern "C" size_t ptt_bn_stats_workspace(int R, int C) {
    if (R <= 0 || C <= 0) return 0;
    const int chunks = (R + ST4_ROWS - 1) / ST4_ROWS;                                       // the finer of the two chunkings
    return (size_t)(chunks > STATS_MIN_CHUNKS ? chunks : STATS_MIN_CHUNKS) * 2 * (size_t)C * sizeof(double);
}

static int bn_tail_from(const ptt_bn_train_tail* t, int C, BnTail* out, const char* who) {
    *out = BnTail{};
    if (!t) return PTT_OK;
    if ((t->act_a != nullptr) != (t->act_b != nullptr) || (t->act_a && (!t->gamma || !t->beta)))
        return fail(PTT_EINVAL, "%s: the activation constants need gamma, beta, act_a and act_b together", who);
    if ((t->running_mean != nullptr) != (t->running_var != nullptr))
        return fail(PTT_EINVAL, "%s: running_mean and running_var go together", who);
    if (t->running_mean && !(t->momentum >= 0.f && t->momentum <= 1.f)) return fail(PTT_EINVAL, "%s: momentum=%g", who, (double)t->momentum);
    (void)C;
    *out = BnTail{t->gamma, t->beta, t->act_a, t->act_b, t->running_mean, t->running_var,
                  reinterpret_cast<long long*>(t->num_batches_tracked), t->momentum};
    return PTT_OK;
}

extern "C" int ptt_bn_stats_train_f32(const float* X, int R, int C, int ldx, float eps, float* mean, float* var, float* invstd,
                                      void* ws, size_t ws_bytes, const ptt_bn_train_tail* tail, ptt_stream_t stream) {
    if (R <= 0 || C <= 0 || ldx < C) return fail(PTT_EINVAL, "ptt_bn_stats_f32: R=%d C=%d ldx=%d", R, C, ldx);
    BnTail bt;
    if (int rc = bn_tail_from(tail, C, &bt, "ptt_bn_stats_train_f32")) return rc;
    if (!X || !mean || !var || !invstd) return fail(PTT_EINVAL, "ptt_bn_stats_f32: null pointer");
    if (!ws || ws_bytes < ptt_bn_stats_workspace(R, C)) return fail(PTT_EWORKSPACE, "ptt_bn_stats_f32: workspace too small");
    hipStream_t s = as_stream(stream);
    if (vec4_ok(X, ldx, C)) {
        const int nchunks = (R + ST4_ROWS - 1) / ST4_ROWS;
        hipLaunchKernelGGL((col_stats4_kernel<0>), dim3(nchunks), dim3(256), 256 * 8 * sizeof(double), s, X, nullptr, nullptr, nullptr,
                           nullptr, R, C, ldx, 0, 0, static_cast<double*>(ws), nullptr, nullptr);
        hipLaunchKernelGGL((col_stats_finish2_kernel<0>), dim3(C), dim3(256), 0, s, static_cast<const double*>(ws), nchunks, C, R, eps,
                           mean, var, invstd, bt);
        return check_launch("col_stats4_kernel");
    }
    const int nchunks = (R + ST_ROWS - 1) / ST_ROWS;
    hipLaunchKernelGGL((col_stats_kernel<0>), dim3(nchunks), dim3(256), 0, s, X, nullptr, nullptr, nullptr, nullptr, R, C, ldx, 0, 0,
                       static_cast<double*>(ws));
    hipLaunchKernelGGL((col_stats_finish_kernel<0>), dim3((C + 255) / 256), dim3(256), 0, s, static_cast<const double*>(ws), nchunks,
                       C, R, eps, mean, var, invstd, bt);
    return check_launch("col_stats_kernel");
}

extern "C" int ptt_bn_stats_f32(const float* X, int R, int C, int ldx, float eps, float* mean, float* var, float* invstd,
                                void* ws, size_t ws_bytes, ptt_stream_t stream) {
    return ptt_bn_stats_train_f32(X, R, C, ldx, eps, mean, var, invstd, ws, ws_bytes, nullptr, stream);
}

extern "C" int ptt_bn_apply_f32(const float* Z, int ldz, const float* mean, const float* invstd, const float* gamma,
                                const float* beta, int R, int C, int relu, float* X, int ldx, ptt_stream_t stream) {
    if (R <= 0 || C <= 0) return fail(PTT_EINVAL, "ptt_bn_apply_f32: R=%d C=%d", R, C);
    if (!Z || !mean || !invstd || !gamma || !beta || !X) return fail(PTT_EINVAL, "ptt_bn_apply_f32: null pointer");
    if (vec4_ok(Z, ldz, C) && vec4_ok(X, ldx, C) && vec4_ok(mean, 4, 4) && vec4_ok(invstd, 4, 4) && vec4_ok(gamma, 4, 4) &&
        vec4_ok(beta, 4, 4)) {
        const int Cq = C >> 2, RG = 256 / (Cq < 256 ? Cq : 256);
        int grid = (R + RG * 8 - 1) / (RG * 8);
        if (grid > 16384) grid = 16384;
        hipLaunchKernelGGL(bn_apply4_kernel, dim3(grid), dim3(256), 0, as_stream(stream), Z, ldz, mean, invstd, gamma, beta, R, C, relu,
                           X, ldx);
        return check_launch("bn_apply4_kernel");
    }
    hipLaunchKernelGGL(bn_apply_kernel, dim3(ew_grid((size_t)R * C)), dim3(256), 0, as_stream(stream), Z, ldz, mean, invstd, gamma,
                       beta, R, C, relu, X, ldx);
    return check_launch("bn_apply_kernel");
}

extern "C" int ptt_bn_bwd_f32(const float* G, int ldg, const float* Act, int lda, const float* Z, int ldz, const float* mean,
                              const float* invstd, const float* gamma, int R, int C, int relu, float* dZ, int ldd,
                              float* dgamma, float* dbeta, void* ws, size_t ws_bytes, const float* act_scale,
                              const float* act_shift, ptt_stream_t stream) {
    if (R <= 0 || C <= 0) return fail(PTT_EINVAL, "ptt_bn_bwd_f32: R=%d C=%d", R, C);
    if (!G || !Z || !mean || !invstd || !gamma || !dZ || !dgamma || !dbeta || (relu && !Act && !(act_scale && act_shift)))
        return fail(PTT_EINVAL, "ptt_bn_bwd_f32: null pointer");
    if (!ws || ws_bytes < ptt_bn_stats_workspace(R, C)) return fail(PTT_EWORKSPACE, "ptt_bn_bwd_f32: workspace too small");
    hipStream_t s = as_stream(stream);
    if (relu && vec4_ok(G, ldg, C) && (Act ? vec4_ok(Act, lda, C) : (vec4_ok(act_scale, 4, 4) && vec4_ok(act_shift, 4, 4))) &&
        vec4_ok(Z, ldz, C) && vec4_ok(dZ, ldd, C) && vec4_ok(mean, 4, 4) &&
        vec4_ok(invstd, 4, 4) && vec4_ok(gamma, 4, 4) && vec4_ok(dgamma, 4, 4) && vec4_ok(dbeta, 4, 4)) {
        const int nch = (R + ST4_ROWS - 1) / ST4_ROWS;
        hipLaunchKernelGGL((col_stats4_kernel<1>), dim3(nch), dim3(256), 256 * 8 * sizeof(double), s, G, Act, Z, mean, invstd, R, C, ldg,
                           lda, ldz, static_cast<double*>(ws), act_scale, act_shift);
        hipLaunchKernelGGL((col_stats_finish2_kernel<1>), dim3(C), dim3(256), 0, s, static_cast<const double*>(ws), nch, C, R, 0.f, dbeta,
                           dgamma, nullptr, BnTail{});
        const int Cq = C >> 2, RG = 256 / (Cq < 256 ? Cq : 256);
        int grid = (R + RG * 8 - 1) / (RG * 8);
        if (grid > 16384) grid = 16384;
        hipLaunchKernelGGL((bn_bwd_apply4_kernel<false>), dim3(grid), dim3(256), 0, s, G, ldg, Act, lda, Z, ldz, mean, invstd, gamma, dbeta,
                           dgamma, R, C, dZ, ldd, act_scale, act_shift, 1.0f / (float)R, nullptr, nullptr, 1);
        return check_launch("bn_bwd4_kernels");
    }
    if (!Act) return fail(PTT_EUNSUPPORTED, "ptt_bn_bwd_f32: the mask-from-z form needs C %% 4 == 0 and 16-byte aligned rows");
    const int nchunks = (R + ST_ROWS - 1) / ST_ROWS;
    // without a ReLU every position passes: the mask test reads G itself against 0 only when relu is set, so pass an
    // always-positive stand-in through Act == G is NOT valid; MODE 1 with relu == 0 uses Act = nullptr guarded below
    if (relu) {
        hipLaunchKernelGGL((col_stats_kernel<1>), dim3(nchunks), dim3(256), 0, s, G, Act, Z, mean, invstd, R, C, ldg, lda, ldz,
                           static_cast<double*>(ws));
    } else {
        return fail(PTT_EUNSUPPORTED, "ptt_bn_bwd_f32: relu == 0 is not instantiated (every SharedMLP unit has a ReLU)");
    }
    hipLaunchKernelGGL((col_stats_finish_kernel<1>), dim3((C + 255) / 256), dim3(256), 0, s, static_cast<const double*>(ws), nchunks,
                       C, R, 0.f, dbeta, dgamma, nullptr, BnTail{});
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_grid((size_t)R * C)), dim3(256), 0, s, G, ldg, Act, lda, Z, ldz, mean, invstd,
                       gamma, dbeta, dgamma, R, C, relu, dZ, ldd);
    return check_launch("bn_bwd_kernels");
}

// ---- SyncBatchNorm pieces (vector form only: C % 4 == 0, 16-byte aligned rows — every conv output of the shipped nets)
extern "C" int ptt_bn_sums_f64(const float* X, int R, int C, int ldx, double* sums, void* ws, size_t ws_bytes,
                               ptt_stream_t stream) {
    if (R <= 0 || C <= 0 || ldx < C) return fail(PTT_EINVAL, "ptt_bn_sums_f64: R=%d C=%d ldx=%d", R, C, ldx);
    if (!X || !sums) return fail(PTT_EINVAL, "ptt_bn_sums_f64: null pointer");
    if (!ws || ws_bytes < ptt_bn_stats_workspace(R, C)) return fail(PTT_EWORKSPACE, "ptt_bn_sums_f64: workspace too small");
    if (!vec4_ok(X, ldx, C)) return fail(PTT_EUNSUPPORTED, "ptt_bn_sums_f64: needs C %% 4 == 0 and 16-byte aligned rows");
    hipStream_t s = as_stream(stream);
    const int nchunks = (R + ST4_ROWS - 1) / ST4_ROWS;
    hipLaunchKernelGGL((col_stats4_kernel<0>), dim3(nchunks), dim3(256), 256 * 8 * sizeof(double), s, X, nullptr, nullptr, nullptr,
                       nullptr, R, C, ldx, 0, 0, static_cast<double*>(ws), nullptr, nullptr);
    hipLaunchKernelGGL(col_sums_finish_kernel, dim3(C), dim3(256), 0, s, static_cast<const double*>(ws), nchunks, C, sums, (double)R);
    return check_launch("ptt_bn_sums_f64");
}

extern "C" int ptt_bn_finish_f64(const double* sums, int C, float eps, float* mean, float* var, float* invstd,
                                 ptt_stream_t stream) {
    if (C <= 0) return fail(PTT_EINVAL, "ptt_bn_finish_f64: C=%d", C);
    if (!sums || !mean || !var || !invstd) return fail(PTT_EINVAL, "ptt_bn_finish_f64: null pointer");
    hipLaunchKernelGGL(bn_finish_sums_kernel, dim3((C + 255) / 256), dim3(256), 0, as_stream(stream), sums, C, eps, mean, var,
                       invstd);
    return check_launch("bn_finish_sums_kernel");
}

extern "C" int ptt_bn_bwd_sums_f64(const float* G, int ldg, const float* Act, int lda, const float* Z, int ldz, const float* mean,
                                   const float* invstd, int R, int C, double* sums, void* ws, size_t ws_bytes,
                                   const float* act_scale, const float* act_shift, ptt_stream_t stream) {
    if (R <= 0 || C <= 0) return fail(PTT_EINVAL, "ptt_bn_bwd_sums_f64: R=%d C=%d", R, C);
    if (!G || !Z || !mean || !invstd || !sums || (!Act && !(act_scale && act_shift)))
        return fail(PTT_EINVAL, "ptt_bn_bwd_sums_f64: null pointer");
    if (!ws || ws_bytes < ptt_bn_stats_workspace(R, C)) return fail(PTT_EWORKSPACE, "ptt_bn_bwd_sums_f64: workspace too small");
    if (!(vec4_ok(G, ldg, C) && (Act ? vec4_ok(Act, lda, C) : (vec4_ok(act_scale, 4, 4) && vec4_ok(act_shift, 4, 4))) &&
          vec4_ok(Z, ldz, C) && vec4_ok(mean, 4, 4) && vec4_ok(invstd, 4, 4)))
        return fail(PTT_EUNSUPPORTED, "ptt_bn_bwd_sums_f64: needs C %% 4 == 0 and 16-byte aligned rows");
    hipStream_t s = as_stream(stream);
    const int nch = (R + ST4_ROWS - 1) / ST4_ROWS;
    hipLaunchKernelGGL((col_stats4_kernel<1>), dim3(nch), dim3(256), 256 * 8 * sizeof(double), s, G, Act, Z, mean, invstd, R, C, ldg,
                       lda, ldz, static_cast<double*>(ws), act_scale, act_shift);
    hipLaunchKernelGGL(col_sums_finish_kernel, dim3(C), dim3(256), 0, s, static_cast<const double*>(ws), nch, C, sums, -1.0);
    return check_launch("ptt_bn_bwd_sums_f64");
}

extern "C" int ptt_bn_bwd_apply_f32(const float* G, int ldg, const float* Act, int lda, const float* Z, int ldz, const float* mean,
                                    const float* invstd, const float* gamma, const float* sum_dy, const float* sum_dy_xhat,
                                    const double* count, int R, int C, float* dZ, int ldd, const float* act_scale,
                                    const float* act_shift, ptt_stream_t stream) {
    if (R <= 0 || C <= 0) return fail(PTT_EINVAL, "ptt_bn_bwd_apply_f32: R=%d C=%d", R, C);
    if (!G || !Z || !mean || !invstd || !gamma || !sum_dy || !sum_dy_xhat || !count || !dZ || (!Act && !(act_scale && act_shift)))
        return fail(PTT_EINVAL, "ptt_bn_bwd_apply_f32: null pointer");
    if (!(vec4_ok(G, ldg, C) && (Act ? vec4_ok(Act, lda, C) : (vec4_ok(act_scale, 4, 4) && vec4_ok(act_shift, 4, 4))) &&
          vec4_ok(Z, ldz, C) && vec4_ok(dZ, ldd, C) && vec4_ok(mean, 4, 4) && vec4_ok(invstd, 4, 4) && vec4_ok(gamma, 4, 4) &&
          vec4_ok(sum_dy, 4, 4) && vec4_ok(sum_dy_xhat, 4, 4)))
        return fail(PTT_EUNSUPPORTED, "ptt_bn_bwd_apply_f32: needs C %% 4 == 0 and 16-byte aligned rows");
    const int Cq = C >> 2, RG = 256 / (Cq < 256 ? Cq : 256);
    int grid = (R + RG * 8 - 1) / (RG * 8);
    if (grid > 16384) grid = 16384;
    hipLaunchKernelGGL((bn_bwd_apply4_kernel<false>), dim3(grid), dim3(256), 0, as_stream(stream), G, ldg, Act, lda, Z, ldz, mean, invstd,
                       gamma, sum_dy, sum_dy_xhat, R, C, dZ, ldd, act_scale, act_shift, 0.f, count, nullptr, 1);
    return check_launch("bn_bwd_apply4_kernel");
}

// ---- the same backward when the gradient arrives pooled (the last layer of a SharedMLP + max-pool stage)
static int pooled_args_ok(const char* what, const float* dP, int ldp, const int32_t* arg, int ns, const float* Z, int ldz, const float* mean,
                          const float* invstd, int R, int C, const float* act_scale, const float* act_shift) {
    if (R <= 0 || C <= 0 || ns <= 0 || R % ns || ldp < C || ldz < C) return fail(PTT_EINVAL, "%s: R=%d C=%d ns=%d ldp=%d ldz=%d", what, R, C, ns, ldp, ldz);
    if (!dP || !arg || !Z || !mean || !invstd || !act_scale || !act_shift) return fail(PTT_EINVAL, "%s: null pointer", what);
    if (!(vec4_ok(dP, ldp, C) && vec4_ok(arg, C, C) && vec4_ok(Z, ldz, C) && vec4_ok(mean, 4, 4) && vec4_ok(invstd, 4, 4) &&
          vec4_ok(act_scale, 4, 4) && vec4_ok(act_shift, 4, 4)))
        return fail(PTT_EUNSUPPORTED, "%s: needs C %% 4 == 0 and 16-byte aligned rows", what);
    return PTT_OK;
}

extern "C" int ptt_bn_bwd_pooled_f32(const float* dPooled, int ldp, const int32_t* arg, int ns, const float* Z, int ldz,
                                     const float* mean, const float* invstd, const float* gamma, int R, int C, float* dZ, int ldd,
                                     float* dgamma, float* dbeta, void* ws, size_t ws_bytes, const float* act_scale,
                                     const float* act_shift, ptt_stream_t stream) {
    if (int rc = pooled_args_ok("ptt_bn_bwd_pooled_f32", dPooled, ldp, arg, ns, Z, ldz, mean, invstd, R, C, act_scale, act_shift)) return rc;
    if (!gamma || !dZ || !dgamma || !dbeta || !vec4_ok(dZ, ldd, C) || !vec4_ok(gamma, 4, 4) || !vec4_ok(dgamma, 4, 4) || !vec4_ok(dbeta, 4, 4))
        return fail(PTT_EINVAL, "ptt_bn_bwd_pooled_f32: bad output / gamma pointer");
    if (!ws || ws_bytes < ptt_bn_stats_workspace(R, C)) return fail(PTT_EWORKSPACE, "ptt_bn_bwd_pooled_f32: workspace too small");
    hipStream_t s = as_stream(stream);
    const int G = R / ns, per = pool_bwd_groups_per_chunk(G, ns), nch = (G + per - 1) / per;
    hipLaunchKernelGGL(pool_bwd_stats_kernel, dim3(nch), dim3(256), 0, s, dPooled, ldp, arg, G, ns, Z, ldz, mean, invstd, act_scale,
                       act_shift, C, static_cast<double*>(ws), per);
    hipLaunchKernelGGL((col_stats_finish2_kernel<1>), dim3(C), dim3(256), 0, s, static_cast<const double*>(ws), nch, C, R, 0.f, dbeta,
                       dgamma, nullptr, BnTail{});
    const int Cq = C >> 2, RG = 256 / (Cq < 256 ? Cq : 256);
    int grid = (R + RG * 8 - 1) / (RG * 8);
    if (grid > 16384) grid = 16384;
    hipLaunchKernelGGL((bn_bwd_apply4_kernel<true>), dim3(grid), dim3(256), 0, s, dPooled, ldp, nullptr, 0, Z, ldz, mean, invstd, gamma,
                       dbeta, dgamma, R, C, dZ, ldd, act_scale, act_shift, 1.0f / (float)R, nullptr, arg, ns);
    return check_launch("ptt_bn_bwd_pooled_f32");
}

extern "C" int ptt_bn_bwd_pooled_sums_f64(const float* dPooled, int ldp, const int32_t* arg, int ns, const float* Z, int ldz,
                                          const float* mean, const float* invstd, int R, int C, double* sums, void* ws,
                                          size_t ws_bytes, const float* act_scale, const float* act_shift, ptt_stream_t stream) {
    if (int rc = pooled_args_ok("ptt_bn_bwd_pooled_sums_f64", dPooled, ldp, arg, ns, Z, ldz, mean, invstd, R, C, act_scale, act_shift)) return rc;
    if (!sums) return fail(PTT_EINVAL, "ptt_bn_bwd_pooled_sums_f64: null pointer");
    if (!ws || ws_bytes < ptt_bn_stats_workspace(R, C)) return fail(PTT_EWORKSPACE, "ptt_bn_bwd_pooled_sums_f64: workspace too small");
    hipStream_t s = as_stream(stream);
    const int G = R / ns, per = pool_bwd_groups_per_chunk(G, ns), nch = (G + per - 1) / per;
    hipLaunchKernelGGL(pool_bwd_stats_kernel, dim3(nch), dim3(256), 0, s, dPooled, ldp, arg, G, ns, Z, ldz, mean, invstd, act_scale,
                       act_shift, C, static_cast<double*>(ws), per);
    hipLaunchKernelGGL(col_sums_finish_kernel, dim3(C), dim3(256), 0, s, static_cast<const double*>(ws), nch, C, sums, -1.0);
    return check_launch("ptt_bn_bwd_pooled_sums_f64");
}

extern "C" int ptt_bn_bwd_pooled_apply_f32(const float* dPooled, int ldp, const int32_t* arg, int ns, const float* Z, int ldz,
                                           const float* mean, const float* invstd, const float* gamma, const float* sum_dy,
                                           const float* sum_dy_xhat, const double* count, int R, int C, float* dZ, int ldd,
                                           const float* act_scale, const float* act_shift, ptt_stream_t stream) {
    if (int rc = pooled_args_ok("ptt_bn_bwd_pooled_apply_f32", dPooled, ldp, arg, ns, Z, ldz, mean, invstd, R, C, act_scale, act_shift)) return rc;
    if (!gamma || !sum_dy || !sum_dy_xhat || !count || !dZ || !vec4_ok(dZ, ldd, C) || !vec4_ok(gamma, 4, 4) || !vec4_ok(sum_dy, 4, 4) ||
        !vec4_ok(sum_dy_xhat, 4, 4))
        return fail(PTT_EINVAL, "ptt_bn_bwd_pooled_apply_f32: bad pointer");
    const int Cq = C >> 2, RG = 256 / (Cq < 256 ? Cq : 256);
    int grid = (R + RG * 8 - 1) / (RG * 8);
    if (grid > 16384) grid = 16384;
    hipLaunchKernelGGL((bn_bwd_apply4_kernel<true>), dim3(grid), dim3(256), 0, as_stream(stream), dPooled, ldp, nullptr, 0, Z, ldz, mean,
                       invstd, gamma, sum_dy, sum_dy_xhat, R, C, dZ, ldd, act_scale, act_shift, 0.f, count, arg, ns);
    return check_launch("bn_bwd_apply4_kernel<pooled>");
}

extern "C" int ptt_pool_rows_f32(const float* X, int ldx, int G, int ns, int C, float* out, int ldo, int32_t* arg,
                                 const float* act_scale, const float* act_shift, ptt_stream_t stream) {
    if (G <= 0 || ns <= 0 || C <= 0) return fail(PTT_EINVAL, "ptt_pool_rows_f32: G=%d ns=%d C=%d", G, ns, C);
    if (!X || !out || !arg) return fail(PTT_EINVAL, "ptt_pool_rows_f32: null pointer");
    hipLaunchKernelGGL(pool_rows_kernel, dim3(ew_grid((size_t)G * C)), dim3(256), 0, as_stream(stream), X, ldx, G, ns, C, out, ldo, arg,
                       act_scale, act_scale ? act_shift : nullptr);
    return check_launch("pool_rows_kernel");
}

extern "C" int ptt_pool_select_f32(const float* pmax, const float* pmin, const int32_t* amax, const int32_t* amin, const float* act_scale,
                                   const float* act_shift, int G, int C, float* out, int32_t* arg, ptt_stream_t stream) {
    if (G <= 0 || C <= 0) return fail(PTT_EINVAL, "ptt_pool_select_f32: G=%d C=%d", G, C);
    if (!pmax || !pmin || !amax || !amin || !act_scale || !act_shift || !out || !arg) return fail(PTT_EINVAL, "ptt_pool_select_f32: null pointer");
    hipLaunchKernelGGL(pool_select_kernel, dim3(ew_grid((size_t)G * C)), dim3(256), 0, as_stream(stream), pmax, pmin, amax, amin, act_scale,
                       act_shift, (size_t)G * C, C, out, arg);
    return check_launch("pool_select_kernel");
}

extern "C" int ptt_pool_rows_bwd_f32(const float* dOut, int ldo, const int32_t* arg, int G, int ns, int C, float* dX, int ldx,
                                     ptt_stream_t stream) {
    if (G <= 0 || ns <= 0 || C <= 0) return fail(PTT_EINVAL, "ptt_pool_rows_bwd_f32: G=%d ns=%d C=%d", G, ns, C);
    if (!dOut || !arg || !dX) return fail(PTT_EINVAL, "ptt_pool_rows_bwd_f32: null pointer");
    hipLaunchKernelGGL(pool_rows_bwd_kernel, dim3(ew_grid((size_t)G * ns * C)), dim3(256), 0, as_stream(stream), dOut, ldo, arg, G, ns,
                       C, dX, ldx);
    return check_launch("pool_rows_bwd_kernel");
}

// rows per workgroup: aim at ~768 workgroups (3 per CU; 2 are resident) in the launch, between 512 and 4096 rows, multiple of 32
static int wgrad_chunk_rows(int R, int Cout, int Cin) {
    const int blocks = ((Cout + 127) / 128) * ((Cin + 127) / 128);
    int want = (768 + blocks - 1) / blocks;                  // row chunks wanted
    int rows = (R + want - 1) / want;
    rows = (rows + WG_KC - 1) / WG_KC * WG_KC;
    if (rows < WG_MIN_ROWS) rows = WG_MIN_ROWS;
    if (rows > WG_ROWS) rows = WG_ROWS;
    return rows;
}

extern "C" int ptt_gather_rows_f32(const float* src, const int32_t* idx, int B, int N, int E, int C, float* out,
                                   ptt_stream_t stream) {
    if (B < 0 || N <= 0 || E < 0 || C <= 0 || (C & 3)) return fail(PTT_EINVAL, "ptt_gather_rows_f32: B=%d N=%d E=%d C=%d (C %% 4)", B, N, E, C);
    if (B == 0 || E == 0) return PTT_OK;
    if (!src || !idx || !out || ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(out)) & 15))
        return fail(PTT_EINVAL, "ptt_gather_rows_f32: null or unaligned pointer");
    const int Cq = C >> 2, RG = 256 / (Cq < 256 ? Cq : 256);
    int gx = (E + RG * 4 - 1) / (RG * 4);
    if (gx > 4096) gx = 4096;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(gx, B), dim3(256), 0, as_stream(stream), src, idx, N, E, C, out);
    return check_launch("gather_rows_kernel");
}

// with statistics the grid is also the number of partial sums the finishing pass walks per channel: about 4096 over the batch
// (16 workgroups per CU) instead of one workgroup per 4 row groups
static inline int sa_z0_grid(int B, int M, int ns, int C, bool stats) {
    const int E = M * ns, Cq = C >> 2, RG = 256 / (Cq < 256 ? Cq : 256);
    int gx = (E + RG * 4 - 1) / (RG * 4);
    if (gx > 4096) gx = 4096;
    if (stats) { const int cap = (4096 + B - 1) / B; if (gx > cap) gx = cap; }
    return gx;
}
extern "C" int ptt_sa_z0_rows_stat_chunks(int B, int M, int ns, int C) {
    if (B <= 0 || M <= 0 || ns <= 0 || C <= 0 || (C & 3) || C > 1024) return 0;
    return sa_z0_grid(B, M, ns, C, true) * B;
}
static int sa_z0_rows_launch(const float* xyz, const float* new_xyz, const int32_t* idx, const float* term, const float* wx,
                             int ldw, int B, int N, int M, int ns, int C, float radius, int normalize_xyz, float* z0,
                             float* rel_rows, double* stats, ptt_stream_t stream) {
    if (B < 0 || N <= 0 || M <= 0 || ns <= 0 || C <= 0 || (C & 3) || !(radius > 0.f) || ldw < 3)
        return fail(PTT_EINVAL, "ptt_sa_z0_rows_f32: B=%d N=%d M=%d ns=%d C=%d (C %% 4) radius=%g", B, N, M, ns, C, (double)radius);
    if ((long long)M * ns > 0x7fffffffLL / 4) return fail(PTT_EUNSUPPORTED, "ptt_sa_z0_rows_f32: M * ns = %lld rows per cloud", (long long)M * ns);
    if (stats && C > 1024) return fail(PTT_EUNSUPPORTED, "ptt_sa_z0_rows_stats_f32: C=%d (at most 1024 channels)", C);
    if (B == 0) return PTT_OK;
    if (!xyz || !new_xyz || !idx || !wx || !z0 || !rel_rows ||
        ((reinterpret_cast<uintptr_t>(z0) | reinterpret_cast<uintptr_t>(term)) & 15))
        return fail(PTT_EINVAL, "ptt_sa_z0_rows_f32: null or unaligned pointer");
    hipLaunchKernelGGL(sa_z0_rows_kernel, dim3(sa_z0_grid(B, M, ns, C, stats != nullptr), B), dim3(256), 0, as_stream(stream), xyz, new_xyz, idx, term, wx, ldw,
                       N, M, ns, C, normalize_xyz ? 1.0f / radius : 1.0f, z0, rel_rows, stats);
    return check_launch("sa_z0_rows_kernel");
}
extern "C" int ptt_sa_z0_rows_f32(const float* xyz, const float* new_xyz, const int32_t* idx, const float* term, const float* wx,
                                  int ldw, int B, int N, int M, int ns, int C, float radius, int normalize_xyz, float* z0,
                                  float* rel_rows, ptt_stream_t stream) {
    return sa_z0_rows_launch(xyz, new_xyz, idx, term, wx, ldw, B, N, M, ns, C, radius, normalize_xyz, z0, rel_rows, nullptr, stream);
}
extern "C" int ptt_sa_z0_rows_stats_f32(const float* xyz, const float* new_xyz, const int32_t* idx, const float* term, const float* wx,
                                        int ldw, int B, int N, int M, int ns, int C, float radius, int normalize_xyz, float* z0,
                                        float* rel_rows, double* stats_partial, size_t partial_elems, ptt_stream_t stream) {
    if (!stats_partial || partial_elems < (size_t)ptt_sa_z0_rows_stat_chunks(B, M, ns, C) * 2 * (size_t)C)
        return fail(PTT_EWORKSPACE, "ptt_sa_z0_rows_stats_f32: the partial sums need %d x 2 x %d doubles", ptt_sa_z0_rows_stat_chunks(B, M, ns, C), C);
    return sa_z0_rows_launch(xyz, new_xyz, idx, term, wx, ldw, B, N, M, ns, C, radius, normalize_xyz, z0, rel_rows, stats_partial, stream);
}

static int scatter_rows_run(const float* g, const int32_t* order, const int32_t* start, int B, int N, int E, int C, float* out,
                            const float* minuend, int negate, ptt_stream_t stream) {
    if (B < 0 || N <= 0 || E < 0 || C <= 0 || (C & 3)) return fail(PTT_EINVAL, "ptt_scatter_rows_csr_f32: B=%d N=%d E=%d C=%d", B, N, E, C);
    if (B == 0) return PTT_OK;
    if (!g || !order || !start || !out || ((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(minuend)) & 15))
        return fail(PTT_EINVAL, "ptt_scatter_rows_csr_f32: null or unaligned pointer");
    const int Cq = C >> 2, RG = 256 / (Cq < 256 ? Cq : 256);
    int gx = (N + RG - 1) / RG;
    if (gx > 4096) gx = 4096;
    hipLaunchKernelGGL(scatter_rows_det_kernel, dim3(gx, B), dim3(256), 0, as_stream(stream), g, order, start, N, E, C, out, minuend, negate);
    return check_launch("scatter_rows_det_kernel");
}
extern "C" int ptt_scatter_rows_csr_f32(const float* g, const int32_t* order, const int32_t* start, int B, int N, int E, int C,
                                        float* out, ptt_stream_t stream) {
    return scatter_rows_run(g, order, start, B, N, E, C, out, nullptr, 0, stream);
}
extern "C" int ptt_scatter_rows_csr_sub_f32(const float* g, const int32_t* order, const int32_t* start, int B, int N, int E, int C,
                                            const float* minuend, float* out, ptt_stream_t stream) {
    return scatter_rows_run(g, order, start, B, N, E, C, out, minuend, 1, stream);
}

static int pt_train_check(const char* what, int B, int N, int k, int D, const void* p0, const void* p1, const void* p2, const void* p3) {
    if (B < 0 || N <= 0 || D <= 0 || (D & 3)) return fail(PTT_EINVAL, "%s: B=%d N=%d D=%d (D %% 4)", what, B, N, D);
    if (k != 16) return fail(PTT_EUNSUPPORTED, "%s: k=%d (16 neighbours is instantiated)", what, k);
    if (B > 0 && (!p0 || !p1 || !p2 || !p3)) return fail(PTT_EINVAL, "%s: null pointer", what);
    return PTT_OK;
}

extern "C" int ptt_pt_pair_input_f32(const float* q, const float* kf, const int32_t* knn, const float* pos, int B, int N, int k,
                                     int D, float* t, ptt_stream_t stream) {
    if (int rc = pt_train_check("ptt_pt_pair_input_f32", B, N, k, D, q, kf, knn, pos)) return rc;
    if (B == 0) return PTT_OK;
    if (!t) return fail(PTT_EINVAL, "ptt_pt_pair_input_f32: null pointer");
    const long long pts = (long long)B * N;
    hipLaunchKernelGGL((pair_input_kernel<16>), dim3(ew_grid((size_t)pts * (D >> 2))), dim3(256), 0, as_stream(stream), q, kf, knn, pos, N,
                       D, pts, t, D, D);
    return check_launch("pair_input_kernel");
}

// the same two passes for INFERENCE at a handful of frames (the per-layer form of the Point-Transformer block,
// ptt_amd/models/transformer_block/variants.py): q / k / v are column slices of the stacked (B,N,3D) projection (row strides
// ldq / ldk / ldv), and the attention tensor is written only when asked for
extern "C" int ptt_pt_pair_input_ld_f32(const float* q, int ldq, const float* kf, int ldk, const int32_t* knn, const float* pos, int B,
                                        int N, int k, int D, float* t, ptt_stream_t stream) {
    if (int rc = pt_train_check("ptt_pt_pair_input_ld_f32", B, N, k, D, q, kf, knn, pos)) return rc;
    if (ldq < D || ldk < D || (ldq & 3) || (ldk & 3)) return fail(PTT_EINVAL, "ptt_pt_pair_input_ld_f32: ldq=%d ldk=%d", ldq, ldk);
    if (B == 0) return PTT_OK;
    if (!t) return fail(PTT_EINVAL, "ptt_pt_pair_input_ld_f32: null pointer");
    const long long pts = (long long)B * N;
    hipLaunchKernelGGL((pair_input_kernel<16>), dim3(ew_grid((size_t)pts * (D >> 2))), dim3(256), 0, as_stream(stream), q, kf, knn, pos, N,
                       D, pts, t, ldq, ldk);
    return check_launch("pair_input_kernel");
}

extern "C" int ptt_pt_attn_fwd_ld_f32(const float* a, const float* vf, int ldv, const int32_t* knn, const float* pos, int B, int N, int k,
                                      int D, float scale, float* attn, float* res, ptt_stream_t stream) {
    if (int rc = pt_train_check("ptt_pt_attn_fwd_ld_f32", B, N, k, D, a, vf, knn, pos)) return rc;
    if (ldv < D || (ldv & 3)) return fail(PTT_EINVAL, "ptt_pt_attn_fwd_ld_f32: ldv=%d", ldv);
    if (B == 0) return PTT_OK;
    if (!res) return fail(PTT_EINVAL, "ptt_pt_attn_fwd_ld_f32: null pointer");
    const long long pts = (long long)B * N;
    hipLaunchKernelGGL((attn_fwd_kernel<16>), dim3(ew_grid((size_t)pts * (D >> 2))), dim3(256), 0, as_stream(stream), a, vf, knn, pos, N, D,
                       pts, scale, attn, res, ldv);
    return check_launch("attn_fwd_kernel");
}

extern "C" int ptt_pt_attn_train_fwd_f32(const float* a, const float* vf, const int32_t* knn, const float* pos, int B, int N, int k,
                                         int D, float scale, float* attn, float* res, ptt_stream_t stream) {
    if (int rc = pt_train_check("ptt_pt_attn_train_fwd_f32", B, N, k, D, a, vf, knn, pos)) return rc;
    if (B == 0) return PTT_OK;
    if (!attn || !res) return fail(PTT_EINVAL, "ptt_pt_attn_train_fwd_f32: null pointer");
    const long long pts = (long long)B * N;
    hipLaunchKernelGGL((attn_fwd_kernel<16>), dim3(ew_grid((size_t)pts * (D >> 2))), dim3(256), 0, as_stream(stream), a, vf, knn, pos, N, D,
                       pts, scale, attn, res, D);
    return check_launch("attn_fwd_kernel");
}

extern "C" int ptt_pt_attn_train_bwd_f32(const float* attn, const float* vf, const int32_t* knn, const float* pos, const float* dres,
                                         int B, int N, int k, int D, float scale, float* da, float* dvp, ptt_stream_t stream) {
    if (int rc = pt_train_check("ptt_pt_attn_train_bwd_f32", B, N, k, D, attn, vf, knn, pos)) return rc;
    if (B == 0) return PTT_OK;
    if (!dres || !da || !dvp) return fail(PTT_EINVAL, "ptt_pt_attn_train_bwd_f32: null pointer");
    const long long pts = (long long)B * N;
    hipLaunchKernelGGL((attn_bwd_kernel<16>), dim3(ew_grid((size_t)pts * (D >> 2))), dim3(256), 0, as_stream(stream), attn, vf, knn, pos,
                       dres, N, D, pts, scale, da, dvp);
    return check_launch("attn_bwd_kernel");
}

static inline bool wgrad_smallk_ok(int Cout, int Cin) { return Cin >= 1 && Cin <= 4 && (Cout & 3) == 0; }

static inline int colsum_rows_per_chunk(int R, int C) {
    const int colblocks = (C + 255) / 256;
    int chunks = 512 / colblocks;                       // about two workgroups per CU in all
    int rows = (R + chunks - 1) / chunks;
    if (rows < CS_ROWS) rows = CS_ROWS;
    return (rows + 3) / 4 * 4;
}
extern "C" size_t ptt_colsum_workspace(int R, int C) {
    if (R <= 0 || C <= 0) return 0;
    const int rows = colsum_rows_per_chunk(R, C);
    return (size_t)((R + rows - 1) / rows) * (size_t)C * sizeof(float);
}
// out == nullptr: the chunk partials [nch][C] stay in the workspace (ptt_colsum_partials_f32), *nchunks_out = nch
static int colsum_run(const float* X, int R, int C, int ldx, float* out, void* ws, size_t ws_bytes, int* nchunks_out, ptt_stream_t stream) {
    if (R <= 0 || C <= 0 || ldx < C) return fail(PTT_EINVAL, "ptt_colsum_f32: R=%d C=%d ldx=%d", R, C, ldx);
    if (!X || (!out && !nchunks_out)) return fail(PTT_EINVAL, "ptt_colsum_f32: null pointer");
    const int rows = colsum_rows_per_chunk(R, C), nch = (R + rows - 1) / rows;
    hipStream_t s2 = as_stream(stream);
    if ((nch > 1 || !out) && (!ws || ws_bytes < ptt_colsum_workspace(R, C))) return fail(PTT_EWORKSPACE, "ptt_colsum_f32: workspace too small");
    float* part = (nch == 1 && out) ? out : static_cast<float*>(ws);
    const bool vec = !(C & 3) && !(ldx & 3) && !((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(part)) & 15);
    if (vec) hipLaunchKernelGGL(colsum_partial_kernel, dim3(nch, (C / 4 + 63) / 64), dim3(256), 0, s2, X, ldx, R, C, rows, part);
    else hipLaunchKernelGGL(colsum_partial_scalar_kernel, dim3(nch, (C + 63) / 64), dim3(256), 0, s2, X, ldx, R, C, rows, part);
    if (nch > 1 && out) launch_wgrad_finish(part, nch, (size_t)C, 0, out, s2);
    if (nchunks_out) *nchunks_out = nch;
    return check_launch("colsum_partial_kernel");
}
extern "C" int ptt_colsum_f32(const float* X, int R, int C, int ldx, float* out, void* ws, size_t ws_bytes, ptt_stream_t stream) {
    if (!out) return fail(PTT_EINVAL, "ptt_colsum_f32: null pointer");
    return colsum_run(X, R, C, ldx, out, ws, ws_bytes, nullptr, stream);
}
extern "C" int ptt_colsum_partials_f32(const float* X, int R, int C, int ldx, void* ws, size_t ws_bytes, int* nchunks, ptt_stream_t stream) {
    if (!nchunks) return fail(PTT_EINVAL, "ptt_colsum_partials_f32: null pointer");
    return colsum_run(X, R, C, ldx, nullptr, ws, ws_bytes, nchunks, stream);
}

extern "C" size_t ptt_linear_wgrad_workspace(int R, int Cout, int Cin) {
    if (R <= 0 || Cout <= 0 || Cin <= 0) return 0;
    if (wgrad_smallk_ok(Cout, Cin)) return (size_t)((R + WK_ROWS - 1) / WK_ROWS) * (size_t)Cout * Cin * sizeof(float);
    const int rows = wgrad_stream_ok(R, Cout, Cin, Cout, Cin) ? wgrad_stream_rows(R) : wgrad_chunk_rows(R, Cout, Cin);   // the finer chunking
    return (size_t)((R + rows - 1) / rows) * (size_t)Cout * Cin * sizeof(float);
}

// dW == nullptr: the chunk partials [nchunks][Cout * Cin] stay in the workspace (ptt_linear_wgrad_partials_f32)
static int linear_wgrad_run(const float* dZ, int ldz, const float* X, int ldx, int R, int Cout, int Cin, float* dW,
                            int accumulate, void* ws, size_t ws_bytes, const float* x_scale, const float* x_shift,
                            int* nchunks_out, ptt_stream_t stream) {
    if (x_scale && (!x_shift || (Cin & 3) || ((reinterpret_cast<uintptr_t>(x_scale) | reinterpret_cast<uintptr_t>(x_shift)) & 15)))
        return fail(PTT_EINVAL, "ptt_linear_wgrad_f32: the input transform needs Cin %% 4 == 0 and 16-byte aligned scale / shift");
    if (R <= 0 || Cout <= 0 || Cin <= 0 || ldz < Cout || ldx < Cin)
        return fail(PTT_EINVAL, "ptt_linear_wgrad_f32: R=%d Cout=%d Cin=%d ldz=%d ldx=%d", R, Cout, Cin, ldz, ldx);
    if (!dZ || !X || (!dW && !nchunks_out)) return fail(PTT_EINVAL, "ptt_linear_wgrad_f32: null pointer");
    if (!ws || ws_bytes < ptt_linear_wgrad_workspace(R, Cout, Cin))
        return fail(PTT_EWORKSPACE, "ptt_linear_wgrad_f32: workspace too small");
    if (wgrad_smallk_ok(Cout, Cin) && !x_scale && (ldz & 3) == 0 && (reinterpret_cast<uintptr_t>(dZ) & 15) == 0) {
        const int nch = (R + WK_ROWS - 1) / WK_ROWS;
        hipStream_t s2 = as_stream(stream);
        float* part = static_cast<float*>(ws);
        if (Cin == 1) hipLaunchKernelGGL((wgrad_smallk_kernel<1>), dim3(nch), dim3(256), 0, s2, dZ, ldz, X, ldx, R, Cout, part);
        else if (Cin == 2) hipLaunchKernelGGL((wgrad_smallk_kernel<2>), dim3(nch), dim3(256), 0, s2, dZ, ldz, X, ldx, R, Cout, part);
        else if (Cin == 3) hipLaunchKernelGGL((wgrad_smallk_kernel<3>), dim3(nch), dim3(256), 0, s2, dZ, ldz, X, ldx, R, Cout, part);
        else hipLaunchKernelGGL((wgrad_smallk_kernel<4>), dim3(nch), dim3(256), 0, s2, dZ, ldz, X, ldx, R, Cout, part);
        if (dW) launch_wgrad_finish(part, nch, (size_t)Cout * Cin, accumulate, dW, s2);
        if (nchunks_out) *nchunks_out = nch;
        return check_launch("wgrad_smallk_kernel");
    }
    if (wgrad_stream_ok(R, Cout, Cin, ldz, ldx)) {       // narrow layers over many rows: the streaming form (wgrad_stream.hip)
        const int rows = wgrad_stream_rows(R), nch = (R + rows - 1) / rows;
        hipStream_t s2 = as_stream(stream);
        if (int rc = launch_wgrad_stream(dZ, ldz, X, ldx, R, Cout, Cin, rows, static_cast<float*>(ws), x_scale, x_shift, s2)) return rc;
        if (dW) launch_wgrad_finish(static_cast<const float*>(ws), nch, (size_t)Cout * Cin, accumulate, dW, s2);
        if (nchunks_out) *nchunks_out = nch;
        return check_launch("wgrad_stream_kernel");
    }
    const int rows = wgrad_chunk_rows(R, Cout, Cin);
    const int nchunks = (R + rows - 1) / rows;
    const int nbo = (Cout + 127) / 128, nbi = (Cin + 127) / 128;
    hipStream_t s = as_stream(stream);
    const int lds = 4 * WG_KC * WG_LD * (int)sizeof(float);
    const bool vz = (ldz & 3) == 0 && (reinterpret_cast<uintptr_t>(dZ) & 15) == 0;
    const bool vx = (ldx & 3) == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0;
    const int narrow_o = (Cout <= 64), narrow_i = (Cin <= 64);
    const int ns = (narrow_o ? 2 : 1) * (narrow_i ? 2 : 1);
#define PTT_WGRAD_CASE(VZ, VX)                                                                                        \
    if (vz == VZ && vx == VX) {                                                                                       \
        if (ns == 1) {                                                                                                \
            if (int rc = set_lds_limit(reinterpret_cast<const void*>(linear_wgrad_kernel<VZ, VX, 1>), lds)) return rc; \
            hipLaunchKernelGGL((linear_wgrad_kernel<VZ, VX, 1>), dim3(nbo * nbi, nchunks), dim3(256), lds, s, dZ, ldz, X, ldx, R, Cout, \
                               Cin, nbi, rows, static_cast<float*>(ws), x_scale, x_shift, 0, 0);                      \
        } else if (ns == 2) {                                                                                         \
            if (int rc = set_lds_limit(reinterpret_cast<const void*>(linear_wgrad_kernel<VZ, VX, 2>), lds)) return rc; \
            hipLaunchKernelGGL((linear_wgrad_kernel<VZ, VX, 2>), dim3(nbo * nbi, nchunks), dim3(256), lds, s, dZ, ldz, X, ldx, R, Cout, \
                               Cin, nbi, rows, static_cast<float*>(ws), x_scale, x_shift, narrow_o, narrow_i);        \
        } else {                                                                                                      \
            if (int rc = set_lds_limit(reinterpret_cast<const void*>(linear_wgrad_kernel<VZ, VX, 4>), lds)) return rc; \
            hipLaunchKernelGGL((linear_wgrad_kernel<VZ, VX, 4>), dim3(nbo * nbi, nchunks), dim3(256), lds, s, dZ, ldz, X, ldx, R, Cout, \
                               Cin, nbi, rows, static_cast<float*>(ws), x_scale, x_shift, narrow_o, narrow_i);        \
        }                                                                                                             \
    }
    PTT_WGRAD_CASE(true, true) PTT_WGRAD_CASE(true, false) PTT_WGRAD_CASE(false, true) PTT_WGRAD_CASE(false, false)
#undef PTT_WGRAD_CASE
    if (dW) launch_wgrad_finish(static_cast<const float*>(ws), nchunks, (size_t)Cout * Cin, accumulate, dW, s);
    if (nchunks_out) *nchunks_out = nchunks;
    return check_launch("linear_wgrad_kernel");
}
extern "C" int ptt_linear_wgrad_f32(const float* dZ, int ldz, const float* X, int ldx, int R, int Cout, int Cin, float* dW,
                                    int accumulate, void* ws, size_t ws_bytes, const float* x_scale, const float* x_shift,
                                    ptt_stream_t stream) {
    if (!dW) return fail(PTT_EINVAL, "ptt_linear_wgrad_f32: null pointer");
    return linear_wgrad_run(dZ, ldz, X, ldx, R, Cout, Cin, dW, accumulate, ws, ws_bytes, x_scale, x_shift, nullptr, stream);
}
extern "C" int ptt_linear_wgrad_partials_f32(const float* dZ, int ldz, const float* X, int ldx, int R, int Cout, int Cin, void* ws,
                                             size_t ws_bytes, const float* x_scale, const float* x_shift, int* nchunks,
                                             ptt_stream_t stream) {
    if (!nchunks) return fail(PTT_EINVAL, "ptt_linear_wgrad_partials_f32: null pointer");
    return linear_wgrad_run(dZ, ldz, X, ldx, R, Cout, Cin, nullptr, 0, ws, ws_bytes, x_scale, x_shift, nchunks, stream);
}

extern "C" int ptt_grad_finish_f32(const ptt_grad_segment* segments_device, const ptt_grad_job* jobs_device, const int32_t* blocks_device,
                                   int n_blocks, float* flat, ptt_stream_t stream) {
    if (n_blocks < 0) return fail(PTT_EINVAL, "ptt_grad_finish_f32: n_blocks=%d", n_blocks);
    if (n_blocks == 0) return PTT_OK;
    if (!segments_device || !jobs_device || !blocks_device || !flat || (reinterpret_cast<uintptr_t>(flat) & 15))
        return fail(PTT_EINVAL, "ptt_grad_finish_f32: null table or a gradient buffer that is not 16-byte aligned");
    hipLaunchKernelGGL(grad_finish_kernel, dim3(n_blocks), dim3(256), 0, as_stream(stream), segments_device, jobs_device, blocks_device, flat);
    return check_launch("grad_finish_kernel");
}

extern "C" int ptt_bn_finish_partials_train_f32(const double* partial, int chunks, int C, int R, float eps, float* mean, float* var,
                                                float* invstd, const ptt_bn_train_tail* tail, ptt_stream_t stream) {
    if (chunks <= 0 || C <= 0 || R <= 0 || !partial || !mean || !var || !invstd)
        return fail(PTT_EINVAL, "ptt_bn_finish_partials_f32: chunks=%d C=%d R=%d", chunks, C, R);
    BnTail bt;
    if (int rc = bn_tail_from(tail, C, &bt, "ptt_bn_finish_partials_train_f32")) return rc;
    hipLaunchKernelGGL((col_stats_finish2_kernel<0>), dim3(C), dim3(256), 0, as_stream(stream), partial, chunks, C, R, eps, mean, var,
                       invstd, bt);
    return check_launch("col_stats_finish2_kernel");
}

extern "C" int ptt_bn_finish_partials_f32(const double* partial, int chunks, int C, int R, float eps, float* mean, float* var,
                                          float* invstd, ptt_stream_t stream) {
    return ptt_bn_finish_partials_train_f32(partial, chunks, C, R, eps, mean, var, invstd, nullptr, stream);
}

extern "C" int ptt_bn_sums_partials_f64(const double* partial, int chunks, int C, int R, double* sums, ptt_stream_t stream) {
    if (chunks <= 0 || C <= 0 || R <= 0 || !partial || !sums) return fail(PTT_EINVAL, "ptt_bn_sums_partials_f64: chunks=%d C=%d R=%d", chunks, C, R);
    hipLaunchKernelGGL(col_sums_finish_kernel, dim3(C), dim3(256), 0, as_stream(stream), partial, chunks, C, sums, (double)R);
    return check_launch("col_sums_finish_kernel");
}

extern "C" int ptt_bn_update_running_f32(const float* mean, const float* var, const double* count, float momentum, int C,
                                         float* running_mean, float* running_var, int64_t* num_batches_tracked, ptt_stream_t stream) {
    if (C <= 0 || !mean || !var || !count || !running_mean || !running_var) return fail(PTT_EINVAL, "ptt_bn_update_running_f32: C=%d or null pointer", C);
    hipLaunchKernelGGL(bn_running_update_kernel, dim3((C + 255) / 256), dim3(256), 0, as_stream(stream), mean, var, count, momentum, C,
                       running_mean, running_var, reinterpret_cast<long long*>(num_batches_tracked));
    return check_launch("bn_running_update_kernel");
}

static inline int xcorr_z0_grid(long long rows, int C, bool stats) {
    const int Cq = C >> 2, RG = 256 / (Cq < 256 ? Cq : 256);
    long long grid = (rows + RG * 8 - 1) / (RG * 8);
    const long long cap = stats ? 4096 : 16384;           // with statistics the grid is also the number of partial sums
    return (int)(grid > cap ? cap : grid);
}
static int xcorr_z0_launch(const float* P, const float* cos_t, const float* w_sim, int B, int n2, int n1, int C, float* z0, double* stats,
                           ptt_stream_t stream) {
    if (B <= 0 || n2 <= 0 || n1 <= 0 || C <= 0 || (C & 3)) return fail(PTT_EINVAL, "ptt_xcorr_z0_f32: B=%d n2=%d n1=%d C=%d", B, n2, n1, C);
    if (!P || !cos_t || !w_sim || !z0 || !vec4_ok(P, C, C) || !vec4_ok(z0, C, C) || !vec4_ok(w_sim, 4, 4))
        return fail(PTT_EINVAL, "ptt_xcorr_z0_f32: null or misaligned pointer");
    const long long rows = (long long)B * n2 * n1;
    hipLaunchKernelGGL(xcorr_z0_kernel, dim3((unsigned)xcorr_z0_grid(rows, C, stats != nullptr)), dim3(256), 0, as_stream(stream), P, cos_t,
                       w_sim, n2, n1, C, rows, z0, stats);
    return check_launch("xcorr_z0_kernel");
}
extern "C" int ptt_xcorr_z0_f32(const float* P, const float* cos_t, const float* w_sim, int B, int n2, int n1, int C, float* z0,
                                ptt_stream_t stream) {
    return xcorr_z0_launch(P, cos_t, w_sim, B, n2, n1, C, z0, nullptr, stream);
}
extern "C" int ptt_xcorr_z0_stat_chunks(int B, int n2, int n1, int C) {
    if (B <= 0 || n2 <= 0 || n1 <= 0 || C <= 0 || (C & 3) || C > 1024) return 0;
    return xcorr_z0_grid((long long)B * n2 * n1, C, true);
}
extern "C" int ptt_xcorr_z0_stats_f32(const float* P, const float* cos_t, const float* w_sim, int B, int n2, int n1, int C, float* z0,
                                      double* stats_partial, size_t partial_elems, ptt_stream_t stream) {
    const int chunks = ptt_xcorr_z0_stat_chunks(B, n2, n1, C);
    if (!chunks) return fail(PTT_EUNSUPPORTED, "ptt_xcorr_z0_stats_f32: B=%d n2=%d n1=%d C=%d (C %% 4 == 0, at most 1024 channels)", B, n2, n1, C);
    if (!stats_partial || partial_elems < (size_t)chunks * 2 * (size_t)C)
        return fail(PTT_EWORKSPACE, "ptt_xcorr_z0_stats_f32: the partial sums need %d x 2 x %d doubles", chunks, C);
    return xcorr_z0_launch(P, cos_t, w_sim, B, n2, n1, C, z0, stats_partial, stream);
}

extern "C" size_t ptt_xcorr_z0_bwd_workspace(int B, int n1, int C) {
    if (B <= 0 || n1 <= 0 || C <= 0) return 0;
    return (size_t)((B * n1 + 3) / 4) * C * sizeof(float);
}

extern "C" int ptt_xcorr_z0_bwd_f32(const float* dz0, const float* cos_t, const float* w_sim, int B, int n2, int n1, int C, float* dP,
                                    float* dcos, float* dw, void* ws, size_t ws_bytes, ptt_stream_t stream) {
    if (B <= 0 || n2 <= 0 || n1 <= 0 || C <= 0 || (C & 3) || C > 256)
        return fail(PTT_EINVAL, "ptt_xcorr_z0_bwd_f32: B=%d n2=%d n1=%d C=%d (C %% 4 == 0, C <= 256)", B, n2, n1, C);
    if (!dz0 || !cos_t || !w_sim || !dP || !dcos || !dw || !vec4_ok(dz0, C, C) || !vec4_ok(dP, C, C) || !vec4_ok(w_sim, 4, 4))
        return fail(PTT_EINVAL, "ptt_xcorr_z0_bwd_f32: null or misaligned pointer");
    if (!ws || ws_bytes < ptt_xcorr_z0_bwd_workspace(B, n1, C)) return fail(PTT_EWORKSPACE, "ptt_xcorr_z0_bwd_f32: workspace too small");
    hipStream_t s = as_stream(stream);
    const int nwg = (B * n1 + 3) / 4;
    hipLaunchKernelGGL(xcorr_z0_bwd_kernel, dim3(nwg), dim3(256), 0, s, dz0, cos_t, w_sim, B, n2, n1, C, dP, dcos, static_cast<float*>(ws));
    launch_wgrad_finish(static_cast<const float*>(ws), nwg, (size_t)C, 0, dw, s);
    return check_launch("xcorr_z0_bwd_kernel");
}

// CosineSimAug's layer 0 backward in ONE pass over the gradient of its activated output (training): the BatchNorm backward sums come
// from the partials of the GEMM that produced G (ptt_rows_gemm_bnbwd_f32) -> dgamma, dbeta; then xcorr_z0_bnbwd_kernel.
extern "C" int ptt_xcorr_z0_bnbwd_f32(const double* partial, int chunks, const float* G, const float* P, const float* cos_t,
                                      const float* w_sim, const float* mean, const float* invstd, const float* gamma,
                                      const float* act_scale, const float* act_shift, int B, int n2, int n1, int C, float* dP, float* dcos,
                                      float* dw, float* dgamma, float* dbeta, void* ws, size_t ws_bytes, ptt_stream_t stream) {
    if (B <= 0 || n2 <= 0 || n1 <= 0 || C <= 0 || (C & 3) || C > 256 || chunks <= 0)
        return fail(PTT_EINVAL, "ptt_xcorr_z0_bnbwd_f32: B=%d n2=%d n1=%d C=%d chunks=%d (C %% 4 == 0, C <= 256)", B, n2, n1, C, chunks);
    if (!partial || !G || !P || !cos_t || !w_sim || !mean || !invstd || !gamma || !act_scale || !act_shift || !dP || !dcos || !dw || !dgamma ||
        !dbeta || !vec4_ok(G, C, C) || !vec4_ok(P, C, C) || !vec4_ok(dP, C, C) || !vec4_ok(w_sim, 4, 4) || !vec4_ok(mean, 4, 4) ||
        !vec4_ok(invstd, 4, 4) || !vec4_ok(gamma, 4, 4) || !vec4_ok(dgamma, 4, 4) || !vec4_ok(dbeta, 4, 4) || !vec4_ok(act_scale, 4, 4) ||
        !vec4_ok(act_shift, 4, 4))
        return fail(PTT_EINVAL, "ptt_xcorr_z0_bnbwd_f32: null or misaligned pointer");
    if (!ws || ws_bytes < ptt_xcorr_z0_bwd_workspace(B, n1, C)) return fail(PTT_EWORKSPACE, "ptt_xcorr_z0_bnbwd_f32: workspace too small");
    hipStream_t s = as_stream(stream);
    const long long R = (long long)B * n2 * n1;
    hipLaunchKernelGGL((col_stats_finish2_kernel<1>), dim3(C), dim3(256), 0, s, partial, chunks, C, (int)R, 0.f, dbeta, dgamma, nullptr, BnTail{});
    const int nwg = (B * n1 + 3) / 4;
    hipLaunchKernelGGL(xcorr_z0_bnbwd_kernel, dim3(nwg), dim3(256), 0, s, G, P, cos_t, w_sim, mean, invstd, gamma, dbeta, dgamma, act_scale,
                       act_shift, 1.0f / (float)R, B, n2, n1, C, dP, dcos, static_cast<float*>(ws));
    launch_wgrad_finish(static_cast<const float*>(ws), nwg, (size_t)C, 0, dw, s);
    return check_launch("xcorr_z0_bnbwd_kernel");
}

// rows per workgroup of sa_z0_bnbwd_kernel: ~2048 workgroups (and partial sums of d_wx) in all, a multiple of 64 rows
static inline int sa_z0_bnbwd_rows(long long R) { return (int)(((R + 2047) / 2048 + 63) / 64 * 64); }
extern "C" size_t ptt_sa_z0_bnbwd_workspace(long long R, int C) {
    if (R <= 0 || C <= 0) return 0;
    const int rows = sa_z0_bnbwd_rows(R);
    return (size_t)((R + rows - 1) / rows) * C * 3 * sizeof(float);
}
// Layer 0 of a hoisted SA level, backward, from the gradient G (R, C) of its ACTIVATED output and the BatchNorm-backward partial
// sums the GEMM that produced G took (ptt_rows_gemm_bnbwd_f32): dgamma / dbeta, then sa_z0_bnbwd_kernel.
extern "C" int ptt_sa_z0_bnbwd_f32(const double* partial, int chunks, const float* G, const float* Z0, const float* rel_rows,
                                   const float* mean, const float* invstd, const float* gamma, const float* act_scale,
                                   const float* act_shift, long long R, int C, float* dz_out, float* dwx, float* dgamma, float* dbeta,
                                   void* ws, size_t ws_bytes, ptt_stream_t stream) {
    if (R <= 0 || R > 0x7fffffffLL || C <= 0 || (C & 3) || C > 1024 || chunks <= 0)
        return fail(PTT_EINVAL, "ptt_sa_z0_bnbwd_f32: R=%lld C=%d chunks=%d (C %% 4 == 0, C <= 1024)", R, C, chunks);
    if (!partial || !G || !Z0 || !rel_rows || !mean || !invstd || !gamma || !act_scale || !act_shift || !dgamma || !dbeta ||
        !vec4_ok(G, C, C) || !vec4_ok(Z0, C, C) || (dz_out && !vec4_ok(dz_out, C, C)) || !vec4_ok(mean, 4, 4) || !vec4_ok(invstd, 4, 4) ||
        !vec4_ok(gamma, 4, 4) || !vec4_ok(dgamma, 4, 4) || !vec4_ok(dbeta, 4, 4) || !vec4_ok(act_scale, 4, 4) || !vec4_ok(act_shift, 4, 4))
        return fail(PTT_EINVAL, "ptt_sa_z0_bnbwd_f32: null or misaligned pointer");
    if (!ws || ws_bytes < ptt_sa_z0_bnbwd_workspace(R, C)) return fail(PTT_EWORKSPACE, "ptt_sa_z0_bnbwd_f32: workspace too small");
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL((col_stats_finish2_kernel<1>), dim3(C), dim3(256), 0, s, partial, chunks, C, (int)R, 0.f, dbeta, dgamma, nullptr, BnTail{});
    const int rows = sa_z0_bnbwd_rows(R), nwg = (int)((R + rows - 1) / rows);
    hipLaunchKernelGGL(sa_z0_bnbwd_kernel, dim3(nwg), dim3(256), 0, s, G, Z0, rel_rows, mean, invstd, gamma, dbeta, dgamma, act_scale, act_shift,
                       1.0f / (float)R, R, C, rows, dz_out, static_cast<float*>(ws));
    if (dwx) launch_wgrad_finish(static_cast<const float*>(ws), nwg, (size_t)C * 3, 0, dwx, s);     // else: the caller sums the partials
    return check_launch("sa_z0_bnbwd_kernel");
}

// BatchNorm(train) + ReLU backward when the two sums were already taken by the GEMM that produced the gradient
// (ptt_rows_gemm_bnbwd_f32): combine its float64 partials in chunk order (-> dbeta, dgamma), then dz
extern "C" int ptt_bn_bwd_from_partials_f32(const double* partial, int chunks, const float* G, int ldg, const float* Z, int ldz,
                                            const float* mean, const float* invstd, const float* gamma, int R, int C, float* dZ, int ldd,
                                            float* dgamma, float* dbeta, const float* act_scale, const float* act_shift,
                                            ptt_stream_t stream) {
    if (R <= 0 || C <= 0 || chunks <= 0) return fail(PTT_EINVAL, "ptt_bn_bwd_from_partials_f32: R=%d C=%d chunks=%d", R, C, chunks);
    if (!partial || !G || !Z || !mean || !invstd || !gamma || !dZ || !dgamma || !dbeta || !act_scale || !act_shift)
        return fail(PTT_EINVAL, "ptt_bn_bwd_from_partials_f32: null pointer");
    if (!(vec4_ok(G, ldg, C) && vec4_ok(Z, ldz, C) && vec4_ok(dZ, ldd, C) && vec4_ok(mean, 4, 4) && vec4_ok(invstd, 4, 4) &&
          vec4_ok(gamma, 4, 4) && vec4_ok(dgamma, 4, 4) && vec4_ok(dbeta, 4, 4) && vec4_ok(act_scale, 4, 4) && vec4_ok(act_shift, 4, 4)))
        return fail(PTT_EUNSUPPORTED, "ptt_bn_bwd_from_partials_f32: needs C %% 4 == 0 and 16-byte aligned rows");
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL((col_stats_finish2_kernel<1>), dim3(C), dim3(256), 0, s, partial, chunks, C, R, 0.f, dbeta, dgamma, nullptr, BnTail{});
    const int Cq = C >> 2, RG = 256 / (Cq < 256 ? Cq : 256);
    int grid = (R + RG * 8 - 1) / (RG * 8);
    if (grid > 16384) grid = 16384;
    hipLaunchKernelGGL((bn_bwd_apply4_kernel<false>), dim3(grid), dim3(256), 0, s, G, ldg, nullptr, 0, Z, ldz, mean, invstd, gamma, dbeta,
                       dgamma, R, C, dZ, ldd, act_scale, act_shift, 1.0f / (float)R, nullptr, nullptr, 1);
    return check_launch("ptt_bn_bwd_from_partials_f32");
}

extern "C" int ptt_bn_bwd_sums_partials_f64(const double* partial, int chunks, int C, double* sums, ptt_stream_t stream) {
    if (chunks <= 0 || C <= 0 || !partial || !sums) return fail(PTT_EINVAL, "ptt_bn_bwd_sums_partials_f64: chunks=%d C=%d", chunks, C);
    hipLaunchKernelGGL(col_sums_finish_kernel, dim3(C), dim3(256), 0, as_stream(stream), partial, chunks, C, sums, -1.0);
    return check_launch("col_sums_finish_kernel");
}

extern "C" int ptt_bn_bwd_consts_f32(const double* partial, int chunks, const float* mean, const float* invstd, const float* gamma, int R,
                                     int C, float* dgamma, float* dbeta, float* k1, float* c0, float* c1, ptt_stream_t stream) {
    if (R <= 0 || C <= 0 || chunks <= 0) return fail(PTT_EINVAL, "ptt_bn_bwd_consts_f32: R=%d C=%d chunks=%d", R, C, chunks);
    if (!partial || !mean || !invstd || !gamma || !dgamma || !dbeta || !k1 || !c0 || !c1) return fail(PTT_EINVAL, "ptt_bn_bwd_consts_f32: null pointer");
    hipLaunchKernelGGL(bn_bwd_consts_kernel, dim3(C), dim3(256), 0, as_stream(stream), partial, chunks, C, R, invstd, gamma, dbeta, dgamma, k1, c0, c1);
    return check_launch("bn_bwd_consts_kernel");
}

extern "C" int ptt_bn_bwd_pooled_consts_f32(const float* dPooled, int ldp, const int32_t* arg, int ns, const float* Z, int ldz,
                                            const float* mean, const float* invstd, const float* gamma, int R, int C, float* dgamma,
                                            float* dbeta, float* k1, float* c0, float* c1, void* ws, size_t ws_bytes,
                                            const float* act_scale, const float* act_shift, ptt_stream_t stream) {
    if (int rc = pooled_args_ok("ptt_bn_bwd_pooled_consts_f32", dPooled, ldp, arg, ns, Z, ldz, mean, invstd, R, C, act_scale, act_shift)) return rc;
    if (!gamma || !dgamma || !dbeta || !k1 || !c0 || !c1) return fail(PTT_EINVAL, "ptt_bn_bwd_pooled_consts_f32: null pointer");
    if (!ws || ws_bytes < ptt_bn_stats_workspace(R, C)) return fail(PTT_EWORKSPACE, "ptt_bn_bwd_pooled_consts_f32: workspace too small");
    hipStream_t s = as_stream(stream);
    const int G = R / ns, per = pool_bwd_groups_per_chunk(G, ns), nch = (G + per - 1) / per;
    hipLaunchKernelGGL(pool_bwd_stats_kernel, dim3(nch), dim3(256), 0, s, dPooled, ldp, arg, G, ns, Z, ldz, mean, invstd, act_scale,
                       act_shift, C, static_cast<double*>(ws), per);
    hipLaunchKernelGGL(bn_bwd_consts_kernel, dim3(C), dim3(256), 0, s, static_cast<const double*>(ws), nch, C, R, invstd, gamma, dbeta, dgamma,
                       k1, c0, c1);
    return check_launch("ptt_bn_bwd_pooled_consts_f32");
}
