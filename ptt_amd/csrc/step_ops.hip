// N3, the ends of the training step that are not matrix work: the four tracking losses with their gradients
// (reference ptt/models/voting_heads/centroids_voting_head.py:29-62, box_voting_head.py:33-66 and the proposal labels of
// box_voting_head.py:96-104) and gradient clipping + the Adam update (tools/train_utils/train_utils.py:47-51 with
// tools/train_utils/optimization/__init__.py:12-14). In stock torch these are ~150 and ~25 launches of a few hundred elements
// each; here they are two launches each, with every reduction in a fixed order (bit-reproducible run to run).
#include <math.h>
#include "common.h"

namespace ptt {

// ------------------------------------------------------------------------------------------------------------------------------
// Losses. nn.BCEWithLogitsLoss(pos_weight): l = (1 - y) x + (1 + (pw - 1) y) (log1p(exp(-|x|)) + max(-x, 0));
// nn.SmoothL1Loss (beta = 1): 0.5 d^2 below 1, |d| - 0.5 above.
// ------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float bce_logits(float x, float y, float pw) {
    const float w = 1.f + (pw - 1.f) * y;
    return (1.f - y) * x + w * (log1pf(expf(-fabsf(x))) + fmaxf(-x, 0.f));
}
__device__ __forceinline__ float bce_logits_grad(float x, float y, float pw) {
    const float w = 1.f + (pw - 1.f) * y;
    const float s = 1.f / (1.f + expf(-x));
    return w * s - pw * y;                                   // = (1 - y) - w (1 - sigmoid(x))
}
__device__ __forceinline__ float smooth_l1(float d) { const float a = fabsf(d); return a < 1.f ? 0.5f * d * d : a - 0.5f; }
__device__ __forceinline__ float smooth_l1_grad(float d) { return fabsf(d) < 1.f ? d : (d > 0.f ? 1.f : -1.f); }

__device__ __forceinline__ float seed_label(const ptt_track_loss_desc& d, int i) {
    if (!d.search_inds) return d.cls_label[i];
    const int b = i / d.N;
    // the torch.gather this replaces raises on an index outside the cloud; here it must at least never leave the buffer
    const long long k = d.search_inds[i];
    return d.cls_label[(size_t)b * d.Ns + (size_t)(k < 0 ? 0 : (k >= d.Ns ? d.Ns - 1 : k))];
}
// box_voting_head.py:96-101: label = dist < 0.3, mask = dist < 0.3 or dist > 0.6, dist = sqrt(|centre - gt centre|^2 + 1e-6)
__device__ __forceinline__ void proposal_label(const ptt_track_loss_desc& d, int j, float& label, float& mask) {
#pragma clang fp contract(off)
    const int b = j / d.M;
    const float* c = d.centres + (size_t)j * 3;
    const float* r = d.reg_label + (size_t)b * d.ld_reg;
    const float dx = c[0] - r[0], dy = c[1] - r[1], dz = c[2] - r[2];
    const float dist = sqrtf(((dx * dx + dy * dy) + dz * dz) + 1e-6f);
    label = dist < 0.3f ? 1.f : 0.f;
    mask = (dist < 0.3f || dist > 0.6f) ? 1.f : 0.f;
}

constexpr int LOSS_THREADS = 1024, LOSS_SUMS = 7;
// out[0] = total, out[1..4] = the four un-weighted losses (seed classification, seed vote regression, proposal score,
// proposal box regression), out[5..7] = sum of the seed labels, of the proposal masks, of the proposal labels
__global__ __launch_bounds__(LOSS_THREADS) void track_losses_kernel(ptt_track_loss_desc d, float* __restrict__ out, float* __restrict__ total) {
    __shared__ double red[LOSS_SUMS][LOSS_THREADS / 64];
    double acc[LOSS_SUMS] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    const float pw_s = d.pos_weight_seed[0], pw_b = d.pos_weight_box[0];
    const int seeds = d.B * d.N, props = d.B * d.M;
    for (int i = threadIdx.x; i < seeds; i += LOSS_THREADS) {
        const int b = i / d.N;
        const float y = seed_label(d, i);
        const float* r = d.reg_label + (size_t)b * d.ld_reg;
        const float* v = d.votes + (size_t)i * 3;
        acc[0] += (double)bce_logits(d.seed_cls[i], y, pw_s);
        acc[1] += (double)y;
        const float m3 = ((smooth_l1(v[0] - r[0]) + smooth_l1(v[1] - r[1])) + smooth_l1(v[2] - r[2])) / 3.f;
        acc[2] += (double)(m3 * y);
    }
    for (int j = threadIdx.x; j < props; j += LOSS_THREADS) {
        const int b = j / d.M;
        float label, mask;
        proposal_label(d, j, label, mask);
        const float* p = d.box_data + (size_t)j * 5;
        const float* r = d.reg_label + (size_t)b * d.ld_reg;
        acc[3] += (double)mask;
        acc[4] += (double)label;
        acc[5] += (double)(bce_logits(p[4], label, pw_b) * mask);
        const float m4 = (((smooth_l1(p[0] - r[0]) + smooth_l1(p[1] - r[1])) + smooth_l1(p[2] - r[2])) + smooth_l1(p[3] - r[3])) / 4.f;
        acc[6] += (double)(m4 * label);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int s = 0; s < LOSS_SUMS; ++s) {
        double v = acc[s];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);      // fixed tree inside the wave
        if (lane == 0) red[s][wave] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t[LOSS_SUMS];
        for (int s = 0; s < LOSS_SUMS; ++s) {
            double v = 0.0;
            for (int w = 0; w < LOSS_THREADS / 64; ++w) v += red[s][w];           // waves in order
            t[s] = v;
        }
        const float seed_cls = (float)(t[0] / (double)(seeds > 0 ? seeds : 1));
        const float seed_reg = (float)t[2] / ((float)t[1] + 1e-6f);
        const float box_cls = (float)t[5] / ((float)t[3] + 1e-6f);
        const float box_reg = (float)t[6] / ((float)t[4] + 1e-6f);
        out[0] = (seed_cls * d.w_seed_cls + seed_reg * d.w_seed_reg) + (box_cls * d.w_box_cls + box_reg * d.w_box_reg);
        if (total) total[0] = out[0];
        out[1] = seed_cls; out[2] = seed_reg; out[3] = box_cls; out[4] = box_reg;
        out[5] = (float)t[1]; out[6] = (float)t[3]; out[7] = (float)t[4];
    }
}

// gradients of out[0] w.r.t. the seed scores (B,N), the votes (B,N,3) and the proposal rows (B,M,5), times the upstream scalar
__global__ __launch_bounds__(256) void track_losses_bwd_kernel(ptt_track_loss_desc d, const float* __restrict__ sums,
                                                               const float* __restrict__ upstream, float* __restrict__ d_seed_cls,
                                                               float* __restrict__ d_votes, float* __restrict__ d_box) {
    const float g = upstream ? upstream[0] : 1.f;
    const int seeds = d.B * d.N, props = d.B * d.M;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < seeds) {
        const int b = i / d.N;
        const float y = seed_label(d, i);
        d_seed_cls[i] = g * d.w_seed_cls * bce_logits_grad(d.seed_cls[i], y, d.pos_weight_seed[0]) / (float)seeds;
        const float k = g * d.w_seed_reg * y / (sums[5] + 1e-6f) / 3.f;
        const float* r = d.reg_label + (size_t)b * d.ld_reg;
        const float* v = d.votes + (size_t)i * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) d_votes[(size_t)i * 3 + c] = k * smooth_l1_grad(v[c] - r[c]);
    }
    if (i < props) {
        const int b = i / d.M;
        float label, mask;
        proposal_label(d, i, label, mask);
        const float* p = d.box_data + (size_t)i * 5;
        const float* r = d.reg_label + (size_t)b * d.ld_reg;
        const float k = g * d.w_box_reg * label / (sums[7] + 1e-6f) / 4.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) d_box[(size_t)i * 5 + c] = k * smooth_l1_grad(p[c] - r[c]);
        d_box[(size_t)i * 5 + 4] = g * d.w_box_cls * mask / (sums[6] + 1e-6f) * bce_logits_grad(p[4], label, d.pos_weight_box[0]);
    }
}

static int loss_desc_check(const char* what, const ptt_track_loss_desc* d) {
    if (!d) return fail(PTT_EINVAL, "%s: null descriptor", what);
    if (d->B <= 0 || d->N <= 0 || d->M <= 0 || d->ld_reg < 4 || (d->search_inds && d->Ns <= 0))
        return fail(PTT_EINVAL, "%s: B=%d N=%d M=%d Ns=%d ld_reg=%d", what, d->B, d->N, d->M, d->Ns, d->ld_reg);
    if ((long long)d->B * d->N > (1 << 24) || (long long)d->B * d->M > (1 << 24))
        return fail(PTT_EUNSUPPORTED, "%s: %lld seeds / %lld proposals (at most 2^24 each)", what, (long long)d->B * d->N, (long long)d->B * d->M);
    if (!d->seed_cls || !d->cls_label || !d->votes || !d->reg_label || !d->box_data || !d->centres || !d->pos_weight_seed || !d->pos_weight_box)
        return fail(PTT_EINVAL, "%s: null pointer", what);
    return PTT_OK;
}

// ------------------------------------------------------------------------------------------------------------------------------
// Gradient clipping + Adam over a table of tensors. Pass 1: per workgroup the float64 sum of squares of its slice of the
// gradients. Pass 2: every workgroup combines the partial sums in order (the same value everywhere), forms
// clip = min(1, max_norm / (norm + 1e-6)) as torch.nn.utils.clip_grad_norm_ does, and updates its slice:
//   g' = clip g (+ wd p);  m = b1 m + (1 - b1) g';  v = b2 v + (1 - b2) g'^2;  p -= (lr / (1 - b1^t)) m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// Workgroup w of the grid works on tensor tensor_of[w], elements [first[w], first[w] + ADAM_CHUNK).
// ------------------------------------------------------------------------------------------------------------------------------
constexpr int ADAM_CHUNK = 4096;        // elements per workgroup (256 threads x 16)

__global__ __launch_bounds__(256) void grad_sqsum_kernel(const ptt_adam_tensor* __restrict__ tensors, const int32_t* __restrict__ tensor_of,
                                                         const int64_t* __restrict__ first, double* __restrict__ partial) {
    __shared__ double red[4];
    const ptt_adam_tensor t = tensors[tensor_of[blockIdx.x]];
    const int64_t e0 = first[blockIdx.x], e1 = e0 + ADAM_CHUNK < t.n ? e0 + ADAM_CHUNK : t.n;
    double s = 0.0;
    for (int64_t e = e0 + threadIdx.x; e < e1; e += 256) { const double g = (double)t.grad[e]; s += g * g; }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// DEV: the hyper-parameters are read from device memory when the launch RUNS (a captured launch replays with the values of the
// step it is replayed for); the arithmetic is the same either way.
template <bool DEV>
__global__ __launch_bounds__(256) void adam_update_kernel(const ptt_adam_tensor* __restrict__ tensors, const int32_t* __restrict__ tensor_of,
                                                          const int64_t* __restrict__ first, const double* __restrict__ partial, int n_partial,
                                                          ptt_adam_hyper h_value, const ptt_adam_hyper* __restrict__ h_device,
                                                          float* __restrict__ norm_out) {
#pragma clang fp contract(off)
    const ptt_adam_hyper h = DEV ? *h_device : h_value;
    __shared__ double red[4];
    __shared__ float clip_s;
    float clip = 1.f;
    if (h.max_norm > 0.f) {
        double s = 0.0;
        for (int k = threadIdx.x; k < n_partial; k += 256) s += partial[k];
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            const float norm = (float)sqrt((red[0] + red[1]) + (red[2] + red[3]));
            const float c = h.max_norm / (norm + 1e-6f);
            clip_s = c < 1.f ? c : 1.f;
            if (blockIdx.x == 0 && norm_out) norm_out[0] = norm;
        }
        __syncthreads();
        clip = clip_s;
    }
    const ptt_adam_tensor t = tensors[tensor_of[blockIdx.x]];
    const int64_t e0 = first[blockIdx.x], e1 = e0 + ADAM_CHUNK < t.n ? e0 + ADAM_CHUNK : t.n;
    for (int64_t e = e0 + threadIdx.x; e < e1; e += 256) {
        float g = t.grad[e] * clip;
        float p = t.param[e];
        if (h.weight_decay != 0.f) g = g + h.weight_decay * p;
        const float m0 = t.exp_avg[e], w = h.one_minus_beta1;                               // torch: exp_avg.lerp_(grad, 1 - beta1),
        const float m = w < 0.5f ? m0 + w * (g - m0) : g - (g - m0) * (1.f - w);         // in ATen's two-sided lerp form
        const float v = t.exp_avg_sq[e] * h.beta2 + (h.one_minus_beta2 * g) * g;            // mul_(beta2).addcmul_(g, g, 1 - beta2)
        const float denom = sqrtf(v) / h.bias2_sqrt + h.eps;
        t.exp_avg[e] = m;
        t.exp_avg_sq[e] = v;
        t.param[e] = p - h.step_size * (m / denom);                                      // addcdiv_(exp_avg, denom, value=-step_size)
        if (h.write_clipped) t.grad[e] = t.grad[e] * clip;                               // leave .grad as clip_grad_norm_ would
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// The cosine map of CosineSimAug in training (reference ptt/models/similarity_modules/p2b_xcoor.py:35-42 through
// nn.CosineSimilarity: cos(s_j, t_i) = s_j . t_i / (max(|s_j|, eps) max(|t_i|, eps))): the map itself is a batched matrix product
// of UNIT rows; these two kernels are everything around it. Wave = one point, lanes over channels.
//   unit_rows:     u = x / max(|x|, eps) as point-major rows, nrm = max(|x|, eps) (negative when the clamp is active: no
//                  gradient flows through a clamped norm)
//   cos_bwd_rows:  dx = (A - (sum_i G_i cos_i) u) / nrm with A = sum_i G_i u'_i (the caller's matrix product), written in the
//                  input's own layout
// ------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void unit_rows_kernel(const float* __restrict__ x, long long sb, long long sn, long long sc, int n, int C,
                                                        long long points, float eps, float* __restrict__ unit, float* __restrict__ nrm) {
    const int lane = threadIdx.x & 63;
    const long long p = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= points) return;
    const long long b = p / n, j = p - b * n;
    const float* row = x + b * sb + j * sn;
    float ss = 0.f;
    for (int c = lane; c < C; c += 64) { const float v = row[c * sc]; ss = __builtin_fmaf(v, v, ss); }
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off, 64);
    const float norm = sqrtf(ss), d = fmaxf(norm, eps);
    for (int c = lane; c < C; c += 64) unit[p * C + c] = row[c * sc] / d;
    if (lane == 0) nrm[p] = norm < eps ? -d : d;
}

__global__ __launch_bounds__(256) void cos_bwd_rows_kernel(const float* __restrict__ A, const float* __restrict__ unit,
                                                           const float* __restrict__ nrm, const float* __restrict__ G,
                                                           const float* __restrict__ cosm, long long map_sb, long long own, long long other,
                                                           int m, int n, int C, long long points, float* __restrict__ dx, long long sb,
                                                           long long sn, long long sc) {
    const int lane = threadIdx.x & 63;
    const long long p = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= points) return;
    const long long b = p / n, j = p - b * n;
    const float d = nrm[p];
    float r = 0.f;
    if (d > 0.f) {                                                         // the projection term exists only where the norm is not clamped
        const float* g = G + b * map_sb + j * own;
        const float* c = cosm + b * map_sb + j * own;
        for (int i = lane; i < m; i += 64) r = __builtin_fmaf(g[i * other], c[i * other], r);
        for (int off = 32; off > 0; off >>= 1) r += __shfl_xor(r, off, 64);
    }
    const float inv = 1.f / fabsf(d);
    float* out = dx + b * sb + j * sn;
    for (int c = lane; c < C; c += 64) out[c * sc] = (A[p * C + c] - r * unit[p * C + c]) * inv;
}

}  // namespace ptt

using namespace ptt;

extern "C" int ptt_track_losses_f32(const ptt_track_loss_desc* d, float* out8, float* total, ptt_stream_t stream) {
    if (int rc = loss_desc_check("ptt_track_losses_f32", d)) return rc;
    if (!out8) return fail(PTT_EINVAL, "ptt_track_losses_f32: null output");
    hipLaunchKernelGGL(track_losses_kernel, dim3(1), dim3(LOSS_THREADS), 0, as_stream(stream), *d, out8, total);
    return check_launch("track_losses_kernel");
}

extern "C" int ptt_track_losses_bwd_f32(const ptt_track_loss_desc* d, const float* out8, const float* upstream, float* d_seed_cls,
                                        float* d_votes, float* d_box_data, ptt_stream_t stream) {
    if (int rc = loss_desc_check("ptt_track_losses_bwd_f32", d)) return rc;
    if (!out8 || !d_seed_cls || !d_votes || !d_box_data) return fail(PTT_EINVAL, "ptt_track_losses_bwd_f32: null pointer");
    const int n = d->B * (d->N > d->M ? d->N : d->M);
    hipLaunchKernelGGL(track_losses_bwd_kernel, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream), *d, out8, upstream, d_seed_cls,
                       d_votes, d_box_data);
    return check_launch("track_losses_bwd_kernel");
}

extern "C" int ptt_adam_chunk_elems(void) { return ADAM_CHUNK; }

extern "C" int ptt_adam_clip_step_f32(const ptt_adam_tensor* tensors_device, const int32_t* chunk_tensor_device,
                                      const int64_t* chunk_first_device, int n_chunks, const ptt_adam_hyper* hyper, double* partial,
                                      size_t partial_elems, float* norm_out, ptt_stream_t stream) {
    if (n_chunks < 0 || !hyper) return fail(PTT_EINVAL, "ptt_adam_clip_step_f32: n_chunks=%d", n_chunks);
    if (n_chunks == 0) return PTT_OK;
    if (!tensors_device || !chunk_tensor_device || !chunk_first_device) return fail(PTT_EINVAL, "ptt_adam_clip_step_f32: null table");
    if (!(hyper->bias2_sqrt > 0.f) || !(hyper->eps >= 0.f)) return fail(PTT_EINVAL, "ptt_adam_clip_step_f32: bias2_sqrt=%g eps=%g", (double)hyper->bias2_sqrt, (double)hyper->eps);
    hipStream_t s = as_stream(stream);
    if (hyper->max_norm > 0.f) {
        if (!partial || partial_elems < (size_t)n_chunks) return fail(PTT_EWORKSPACE, "ptt_adam_clip_step_f32: %d partial sums needed", n_chunks);
        hipLaunchKernelGGL(grad_sqsum_kernel, dim3(n_chunks), dim3(256), 0, s, tensors_device, chunk_tensor_device, chunk_first_device, partial);
        if (int rc = check_launch("grad_sqsum_kernel")) return rc;
    }
    hipLaunchKernelGGL(adam_update_kernel<false>, dim3(n_chunks), dim3(256), 0, s, tensors_device, chunk_tensor_device, chunk_first_device,
                       partial, n_chunks, *hyper, (const ptt_adam_hyper*)nullptr, norm_out);
    return check_launch("adam_update_kernel");
}

extern "C" int ptt_adam_clip_step_dev_f32(const ptt_adam_tensor* tensors_device, const int32_t* chunk_tensor_device,
                                          const int64_t* chunk_first_device, int n_chunks, const ptt_adam_hyper* hyper_device, int clip,
                                          double* partial, size_t partial_elems, float* norm_out, ptt_stream_t stream) {
    if (n_chunks < 0 || !hyper_device) return fail(PTT_EINVAL, "ptt_adam_clip_step_dev_f32: n_chunks=%d", n_chunks);
    if (n_chunks == 0) return PTT_OK;
    if (!tensors_device || !chunk_tensor_device || !chunk_first_device) return fail(PTT_EINVAL, "ptt_adam_clip_step_dev_f32: null table");
    hipStream_t s = as_stream(stream);
    if (clip) {
        if (!partial || partial_elems < (size_t)n_chunks) return fail(PTT_EWORKSPACE, "ptt_adam_clip_step_dev_f32: %d partial sums needed", n_chunks);
        hipLaunchKernelGGL(grad_sqsum_kernel, dim3(n_chunks), dim3(256), 0, s, tensors_device, chunk_tensor_device, chunk_first_device, partial);
        if (int rc = check_launch("grad_sqsum_kernel")) return rc;
    }
    hipLaunchKernelGGL(adam_update_kernel<true>, dim3(n_chunks), dim3(256), 0, s, tensors_device, chunk_tensor_device, chunk_first_device,
                       partial, n_chunks, ptt_adam_hyper{}, hyper_device, norm_out);
    return check_launch("adam_update_kernel");
}

extern "C" int ptt_unit_rows_f32(const float* x, long long sb, long long sn, long long sc, int B, int n, int C, float eps, float* unit,
                                 float* nrm, ptt_stream_t stream) {
    if (B <= 0 || n <= 0 || C <= 0 || !(eps > 0.f)) return fail(PTT_EINVAL, "ptt_unit_rows_f32: B=%d n=%d C=%d eps=%g", B, n, C, (double)eps);
    if (!x || !unit || !nrm) return fail(PTT_EINVAL, "ptt_unit_rows_f32: null pointer");
    const long long points = (long long)B * n;
    hipLaunchKernelGGL(unit_rows_kernel, dim3((unsigned)((points + 3) / 4)), dim3(256), 0, as_stream(stream), x, sb, sn, sc, n, C, points, eps,
                       unit, nrm);
    return check_launch("unit_rows_kernel");
}

extern "C" int ptt_cos_bwd_rows_f32(const float* A, const float* unit, const float* nrm, const float* G, const float* cosm, long long map_sb,
                                    long long own, long long other, int m, int B, int n, int C, float* dx, long long sb, long long sn,
                                    long long sc, ptt_stream_t stream) {
    if (B <= 0 || n <= 0 || C <= 0 || m <= 0) return fail(PTT_EINVAL, "ptt_cos_bwd_rows_f32: B=%d n=%d C=%d m=%d", B, n, C, m);
    if (!A || !unit || !nrm || !G || !cosm || !dx) return fail(PTT_EINVAL, "ptt_cos_bwd_rows_f32: null pointer");
    const long long points = (long long)B * n;
    hipLaunchKernelGGL(cos_bwd_rows_kernel, dim3((unsigned)((points + 3) / 4)), dim3(256), 0, as_stream(stream), A, unit, nrm, G, cosm, map_sb,
                       own, other, m, n, C, points, dx, sb, sn, sc);
    return check_launch("cos_bwd_rows_kernel");
}
