// Row jobs: up to PTT_ROW_JOBS_MAX independent row-wise fp32-MFMA layers in ONE launch (include/ptt_hip.h,
// ptt_row_jobs_f32), built for the launch chain of ONE tracklet frame (B = 1, the reference's own tracking mode:
// tools/eval_utils/eval_tracking_utils.py:140-152). At 128 - 2048 rows a layer is 17 - 1070 MFLOP: a few microseconds of
// matrix work behind ~5 us of launch latency, ~45 of them in a dependent chain. What this kernel does about it:
//   * K is split over the waves of a workgroup. 8 waves own a 32-row x (32 cw)-column tile, cw in {1, 2, 4}, and 8 / cw wave
//     groups each take a contiguous slice of the K-blocks; the partial accumulators meet in LDS. A 128 x 512 x 512 layer is
//     64 workgroups whose waves issue 32 MFMAs each (0.9 us) instead of 16 workgroups whose SIMDs issue 256 (7 us).
//   * several jobs per launch: the workgroups of a launch are dealt to the jobs of a by-value table (no device-side table to
//     upload, graph-capturable), so independent layers (q|k|v beside fc_delta, cla_layer beside vote_layer) share one launch.
//   * the element-wise launches around the layers are folded into the staging of the A operand (concatenated inputs,
//     fc_delta[0] + ReLU, q_i - k_j + pos_ij) and into the epilogue (sigmoid, two-part residuals and outputs, the softmax
//     over the 16 neighbours + weighted sum of the Point-Transformer block, which is local to a 32-row accumulator tile).
// Arithmetic: v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulation). The K axis is cut into EIGHT slices whatever
// the tile shape; every slice is accumulated from zero in K order and the eight sums are combined as one fixed binary tree
// (rj_body), so a row's bits depend on the row and the weights only — not on the launch size, the tile shape it selects or
// the jobs it shares a launch with — and are within fp32 rounding of the un-split kernels (linear_small_kernel, mfma_ops.hip).
#include <math.h>
#include "common.h"
#include "mfma_common.h"

namespace ptt {

struct RowJobDev {
    ptt_row_job j;
    int NT, nkb, cw, ks, hb, ldk, cg, blk0, vecx, vecp;
};
struct RowJobsParams { RowJobDev j[PTT_ROW_JOBS_MAX]; int n; };

template <int PROS, int EPIS, int PD>
__device__ __forceinline__ void rj_body(const RowJobDev& D, float* smem) {
    const ptt_row_job& J = D.j;
    const int t = threadIdx.x, lane = t & 63, half = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int cw = D.cw, ks = D.ks, hb = D.hb, ldk = D.ldk;
    const int cwi = w % cw, ksi = w / cw;
    const int b = (int)blockIdx.x - D.blk0;
    const int cgi = b % D.cg, rti = b / D.cg;
    const int row0 = rti * 32;
    const int ct = cgi * cw + cwi;
    const bool has_ct = ct < D.NT;
    float* Xs = smem;                        // [32][ldk]
    float* Rs = smem + 32 * ldk;             // [(ks-1) * cw][16 registers][64 lanes]

    // ---- the wave's first PD weight fragments: requested before anything else
    const __amdgpu_buffer_rsrc_t wr = weight_rsrc(J.Wpacked);
    const int wvoff = ((has_ct ? ct : 0) * 64 + lane) * 16;
    const int wkstep = D.NT * 1024;
    const int last = D.nkb - 1;
    f32x4 bw[PD];
#pragma unroll
    for (int i = 0; i < PD; ++i) bw[i] = weight_load(wr, wvoff, min(ksi * (8 / ks) * hb + i, last) * wkstep);

    // ---- stage the A tile: thread (tr, tc) owns rows tr, tr + 16 and the float4 column slots tc + 32 i
    {
        const int tr = t >> 5, tc = t & 31;
        const int qpr = (8 * hb * 8) >> 2;                      // float4 slots per row (zero beyond K)
        const int gr0 = row0 + tr, gr1 = row0 + tr + 16;
        if ((PROS & 1) && J.prologue == 0) {
            // float4 slots wholly inside X's columns come as one vector load, slots beyond K are zero; the few slots that hold
            // the end of X / the columns of X2 (or everything, when X is not 16-byte aligned) are filled element-wise below
            const int K1v = D.vecx ? (J.K1 & ~3) : 0, Kz = (J.K + 3) & ~3;
            const float* x0 = J.X + (size_t)gr0 * J.ldx;
            const float* x1 = J.X + (size_t)gr1 * J.ldx;
            const long long xm = J.Xmax ? J.Xmax - J.X : 0;      // element distance to the second operand of the maximum
            for (int i0 = 0; i0 < qpr; i0 += 128) {
                f32x4 v[4][2];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c = (i0 + tc + 32 * i) << 2;
                    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                    v[i][0] = (c < K1v && gr0 < J.rows) ? *reinterpret_cast<const f32x4*>(x0 + c) : z;
                    v[i][1] = (c < K1v && gr1 < J.rows) ? *reinterpret_cast<const f32x4*>(x1 + c) : z;
                    if (xm) {
                        const f32x4 m0 = (c < K1v && gr0 < J.rows) ? *reinterpret_cast<const f32x4*>(x0 + xm + c) : z;
                        const f32x4 m1 = (c < K1v && gr1 < J.rows) ? *reinterpret_cast<const f32x4*>(x1 + xm + c) : z;
#pragma unroll
                        for (int q = 0; q < 4; ++q) { v[i][0][q] = fmaxf(v[i][0][q], m0[q]); v[i][1][q] = fmaxf(v[i][1][q], m1[q]); }
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int s = i0 + tc + 32 * i, c = s << 2;
                    if (s < qpr && (c < K1v || c >= Kz)) {
                        *reinterpret_cast<f32x4*>(Xs + tr * ldk + c) = v[i][0];
                        *reinterpret_cast<f32x4*>(Xs + (tr + 16) * ldk + c) = v[i][1];
                    }
                }
            }
            const int tw = Kz - K1v;                             // columns [K1v, Kz): element-wise, zero beyond K
#pragma unroll 1
            for (int e = t; e < 32 * tw; e += 512) {
                const int r = e / tw, cc = K1v + (e - r * tw), gr = row0 + r;
                float val = 0.f;
                if (gr < J.rows) {
                    if (cc < J.K1) val = J.X[(size_t)gr * J.ldx + cc];
                    else if (cc < J.K) val = J.X2[(size_t)gr * J.ldx2 + (cc - J.K1)];
                }
                Xs[r * ldk + cc] = val;
            }
        } else if ((PROS & 2) && J.prologue == 1) {
            // A[r, c] = relu(fc_delta[0](rel[r])): products summed x, y, z as the K = 3 MFMA layer sums them, then the bias
            float r0[3] = {0.f, 0.f, 0.f}, r1[3] = {0.f, 0.f, 0.f};
            if (gr0 < J.rows) { r0[0] = J.rel[(size_t)gr0 * 3]; r0[1] = J.rel[(size_t)gr0 * 3 + 1]; r0[2] = J.rel[(size_t)gr0 * 3 + 2]; }
            if (gr1 < J.rows) { r1[0] = J.rel[(size_t)gr1 * 3]; r1[1] = J.rel[(size_t)gr1 * 3 + 1]; r1[2] = J.rel[(size_t)gr1 * 3 + 2]; }
            for (int s = tc; s < qpr; s += 32) {
                const int c = s << 2;
                f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (c + q < J.K) {
                        const f32x4 wv = *reinterpret_cast<const f32x4*>(J.w1 + (size_t)(c + q) * 4);
                        const float a0 = __builtin_fmaf(r0[2], wv[2], __builtin_fmaf(r0[1], wv[1], r0[0] * wv[0])) + wv[3];
                        const float a1 = __builtin_fmaf(r1[2], wv[2], __builtin_fmaf(r1[1], wv[1], r1[0] * wv[0])) + wv[3];
                        o0[q] = gr0 < J.rows ? fmaxf(a0, 0.f) : 0.f;
                        o1[q] = gr1 < J.rows ? fmaxf(a1, 0.f) : 0.f;
                    }
                }
                *reinterpret_cast<f32x4*>(Xs + tr * ldk + c) = o0;
                *reinterpret_cast<f32x4*>(Xs + (tr + 16) * ldk + c) = o1;
            }
        } else if ((PROS & 8) && J.prologue == 3) {
            // A[(c, j), :] = act0(term[n] + wx . rel), n = idx[c, j], rel = (xyz[n] - centre[c]) (/ radius): the arithmetic of
            // sa_gather_rows (mfma_ops.hip) — true division, the three products added to the term as one fma chain x, y, z
            const bool ok0 = gr0 < J.rows, ok1 = gr1 < J.rows;
            const int c0 = ok0 ? gr0 / J.ns : 0, c1 = ok1 ? gr1 / J.ns : 0;
            const size_t f0 = (size_t)(c0 / J.M) * J.N + (ok0 ? J.idx[gr0] : 0), f1 = (size_t)(c1 / J.M) * J.N + (ok1 ? J.idx[gr1] : 0);
            float r0[3], r1[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                r0[a] = J.xyz[f0 * 3 + a] - J.centres[(size_t)c0 * 3 + a];
                r1[a] = J.xyz[f1 * 3 + a] - J.centres[(size_t)c1 * 3 + a];
                if (J.normalize_xyz) { r0[a] /= J.radius; r1[a] /= J.radius; }
            }
            const float* t0 = J.X + f0 * J.ldx;
            const float* t1 = J.X + f1 * J.ldx;
            const float lo = J.pro_relu ? 0.f : -__builtin_inff();
            for (int i0 = 0; i0 < qpr; i0 += 128) {
                f32x4 va[4][2], wv[4][3];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c = (i0 + tc + 32 * i) << 2;
                    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                    const bool in = c < J.K;
                    va[i][0] = (in && ok0) ? *reinterpret_cast<const f32x4*>(t0 + c) : z;
                    va[i][1] = (in && ok1) ? *reinterpret_cast<const f32x4*>(t1 + c) : z;
#pragma unroll
                    for (int a = 0; a < 3; ++a) wv[i][a] = in ? *reinterpret_cast<const f32x4*>(J.wx + (size_t)a * J.K + c) : z;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int s = i0 + tc + 32 * i, c = s << 2;
                    if (s < qpr) {
                        f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = o0;
                        if (c < J.K) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const float y0 = __builtin_fmaf(wv[i][2][q], r0[2], __builtin_fmaf(wv[i][1][q], r0[1], __builtin_fmaf(wv[i][0][q], r0[0], va[i][0][q])));
                                const float y1 = __builtin_fmaf(wv[i][2][q], r1[2], __builtin_fmaf(wv[i][1][q], r1[1], __builtin_fmaf(wv[i][0][q], r1[0], va[i][1][q])));
                                o0[q] = ok0 ? fmaxf(y0, lo) : 0.f;
                                o1[q] = ok1 ? fmaxf(y1, lo) : 0.f;
                            }
                        }
                        *reinterpret_cast<f32x4*>(Xs + tr * ldk + c) = o0;
                        *reinterpret_cast<f32x4*>(Xs + (tr + 16) * ldk + c) = o1;
                    }
                }
            }
        } else if ((PROS & 4) && J.prologue == 2) {
            // A[(i, j), c] = (q_i[c] - k_{knn(i, j)}[c]) + pos_{ij}[c]   (K % 4 == 0, 16-byte aligned rows: checked by the host)
            const int p0 = gr0 >> 4, p1 = gr1 >> 4;
            const bool ok0 = gr0 < J.rows, ok1 = gr1 < J.rows;
            const int n0 = ok0 ? J.knn[gr0] : 0, n1 = ok1 ? J.knn[gr1] : 0;
            const float* q0 = J.qkv + (size_t)p0 * J.ldq + J.q_off;
            const float* q1 = J.qkv + (size_t)p1 * J.ldq + J.q_off;
            const float* k0 = J.qkv + ((size_t)(p0 / J.N) * J.N + n0) * J.ldq + J.k_off;
            const float* k1 = J.qkv + ((size_t)(p1 / J.N) * J.N + n1) * J.ldq + J.k_off;
            const float* s0 = J.pos + (size_t)gr0 * J.ldp;
            const float* s1 = J.pos + (size_t)gr1 * J.ldp;
            for (int i0 = 0; i0 < qpr; i0 += 128) {
                f32x4 qa[4][2], ka[4][2], pa[4][2];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c = (i0 + tc + 32 * i) << 2;
                    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                    const bool in = c < J.K;
                    qa[i][0] = (in && ok0) ? *reinterpret_cast<const f32x4*>(q0 + c) : z;
                    ka[i][0] = (in && ok0) ? *reinterpret_cast<const f32x4*>(k0 + c) : z;
                    pa[i][0] = (in && ok0) ? *reinterpret_cast<const f32x4*>(s0 + c) : z;
                    qa[i][1] = (in && ok1) ? *reinterpret_cast<const f32x4*>(q1 + c) : z;
                    ka[i][1] = (in && ok1) ? *reinterpret_cast<const f32x4*>(k1 + c) : z;
                    pa[i][1] = (in && ok1) ? *reinterpret_cast<const f32x4*>(s1 + c) : z;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int s = i0 + tc + 32 * i;
                    if (s < qpr) {
                        f32x4 o0, o1;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            o0[q] = (qa[i][0][q] - ka[i][0][q]) + pa[i][0][q];
                            o1[q] = (qa[i][1][q] - ka[i][1][q]) + pa[i][1][q];
                        }
                        *reinterpret_cast<f32x4*>(Xs + tr * ldk + (s << 2)) = o0;
                        *reinterpret_cast<f32x4*>(Xs + (tr + 16) * ldk + (s << 2)) = o1;
                    }
                }
            }
        }
    }
    // the epilogue's per-column constants and residual rows are requested now: behind the K loop they would be one more
    // exposed round trip to memory at the end of a kernel that is a chain of round trips
    const int col = ct * 32 + (lane & 31);
    const bool has_col = has_ct && col < J.Cout;
    float sc = 1.f, sh = 0.f, resv[16];
    const bool plain_epi = !((EPIS & 1) && J.epilogue == 1) && ksi == 0 && has_col;
    if (plain_epi) {
        if (J.scale) sc = J.scale[col];
        if (J.shift) sh = J.shift[col];
        const float* rp = nullptr;
        int ldr = 0;
        if (col >= J.res_split) { if (J.res) { rp = J.res + (col - J.res_split); ldr = J.ldr; } }
        else if (J.res2) { rp = J.res2 + col; ldr = J.ldr2; }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int gr = row0 + tile_row(r, half);
            resv[r] = (rp && gr < J.rows) ? rp[(size_t)gr * ldr] : 0.f;
        }
    }
    __syncthreads();

    // ---- this wave's K-blocks: 8 / ks consecutive SLICES of hb K-blocks (hb a multiple of PD), PD weight fragments in flight.
    // The sum over K is ALWAYS formed as the same tree over the eight slice sums s0..s7 (each accumulated from zero in K order):
    //     ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7))
    // whichever way the slices are dealt to waves (ks = 8, 4 or 2), so a row's result does not depend on the tile shape the
    // launch size selects — a frame gives the same bits alone and in a batch of three.
    const int nsl = 8 / ks;                                     // slices of this wave
    const int kb0 = ksi * nsl * hb;
    const float* arow = Xs + (lane & 31) * ldk + 4 * half + kb0 * 8;
    f32x16 acc, pair, total;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; pair[r] = 0.f; total[r] = 0.f; }
    int in_slice = 0, sl = 0;
    for (int k = 0; k < nsl * hb; k += PD) {
#pragma unroll
        for (int i = 0; i < PD; ++i) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(arow + (k + i) * 8);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], bw[i][q], acc, 0, 0, 0);
            bw[i] = weight_load(wr, wvoff, min(kb0 + k + i + PD, last) * wkstep);   // past the wave's share: a valid fragment, unused
        }
        in_slice += PD;
        if (in_slice == hb) {                                   // a slice sum is complete (wave-uniform)
            in_slice = 0;
            if (sl & 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) pair[r] += acc[r];
                if (sl == 1) total = pair;
                else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) total[r] += pair[r];
                }
            } else pair = acc;
            ++sl;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        }
    }
    if (nsl == 1) total = pair;
    if (ksi > 0) {
        float* dst = Rs + (size_t)(((ksi - 1) * cw + cwi) * 16) * 64 + lane;
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[r * 64] = total[r];
    }
    __syncthreads();
    if (ksi > 0 || !has_ct) return;
    {
        const float* src = Rs + (size_t)(cwi * 16) * 64 + lane;     // wave group g's partial: src + (g - 1) * cw * 1024
        const int gs = cw * 1024;
        if (ks == 2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = total[r] + src[r * 64];
        } else if (ks == 4) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = (total[r] + src[r * 64]) + (src[gs + r * 64] + src[2 * gs + r * 64]);
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                acc[r] = ((total[r] + src[r * 64]) + (src[gs + r * 64] + src[2 * gs + r * 64])) +
                         ((src[3 * gs + r * 64] + src[4 * gs + r * 64]) + (src[5 * gs + r * 64] + src[6 * gs + r * 64]));
        }
    }

    if ((EPIS & 1) && J.epilogue == 1) {
        // rows 0-15 of the tile are the 16 neighbours of point row0 / 16, rows 16-31 those of the next point: registers
        // 8p .. 8p+7 of a lane hold eight of point p's neighbours, lane ^ 32 the other eight. The per-column bias of the
        // layer is the same for all neighbours and cancels in the softmax: it is not read.
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int pt = (row0 >> 4) + p;
            if (pt * 16 >= J.rows) break;
            const size_t cloud_row0 = (size_t)(pt / J.N) * J.N;
            float a[8], m = -3.0e38f;
#pragma unroll
            for (int q = 0; q < 8; ++q) { a[q] = acc[8 * p + q] * J.sm_scale; m = fmaxf(m, a[q]); }
            m = max_halves(m);
            float sum = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) { a[q] = __expf(a[q] - m); sum += a[q]; }
            sum = add_halves(sum);
            const float inv = 1.0f / sum;
            float o = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int jn = tile_row(8 * p + q, half) - 16 * p;
                const int n = J.knn[(size_t)pt * 16 + jn];
                const float vv = has_col ? J.qkv[(cloud_row0 + n) * J.ldq + J.v_off + col] : 0.f;
                const float pv = has_col ? J.pos[((size_t)pt * 16 + jn) * J.ldp + col] : 0.f;
                o += (a[q] * inv) * (vv + pv);
            }
            o = add_halves(o);
            if (half == 0 && has_col) J.out[(size_t)pt * J.ldo + col] = o;
        }
        return;
    }
    if (!has_col) return;
    if ((EPIS & 2) && J.epilogue == 2) {
        // max over the ns = 16 (two centres per tile) or 32 (one) neighbour rows; rows past the end do not take part
        float best[2] = {-__builtin_inff(), -__builtin_inff()};
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int gr = row0 + tile_row(r, half);
            const float y = gr < J.rows ? acc[r] * sc + sh : -__builtin_inff();
            best[r >> 3] = fmaxf(best[r >> 3], y);
        }
        if (J.ns == 32) best[0] = fmaxf(best[0], best[1]);
        best[0] = max_halves(best[0]);
        best[1] = max_halves(best[1]);
        if (half == 0) {
            const int npc = J.ns == 32 ? 1 : 2;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int c = row0 / J.ns + p;
                if (p < npc && c * J.ns < J.rows) {
                    float y = best[p];
                    if (J.act == 1) y = fmaxf(y, 0.f);
                    else if (J.act == 2) y = 1.0f / (1.0f + __expf(-y));
                    J.out[(size_t)c * J.ldo + col] = y;
                }
            }
        }
        return;
    }
    float* op = col >= J.out_split ? J.out + (col - J.out_split) + J.out_col0 : J.out2 + col;
    const int ldo = col >= J.out_split ? J.ldo : J.ldo2;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int gr = row0 + tile_row(r, half);
        if (gr >= J.rows) continue;
        float y = acc[r] * sc + sh;
        if (J.raw) J.raw[(size_t)gr * J.ldraw + col] = y;
        if (J.act == 1) y = fmaxf(y, 0.f);
        else if (J.act == 2) y = 1.0f / (1.0f + __expf(-y));
        op[(size_t)gr * ldo] = y + resv[r];
    }
}

// PROS: bit p set = some job of the launch has prologue p; EPIS: bit e - 1 set = some job has epilogue e; PD: weight fragments in flight per
// wave. A launch gets the instantiation that holds only what its jobs use: the all-purpose kernel is ~50 KB of code, and a
// 5-us kernel pays for every instruction-cache line it has to pull in.
template <int PROS, int EPIS, int PD>
__global__ __launch_bounds__(512, 1) void rowjobs_kernel(RowJobsParams P) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int ji = 0;
#pragma unroll
    for (int i = 1; i < PTT_ROW_JOBS_MAX; ++i)
        if (i < P.n && (int)blockIdx.x >= P.j[i].blk0) ji = i;
    rj_body<PROS, EPIS, PD>(P.j[ji], smem);
}

}  // namespace ptt

using namespace ptt;

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" int ptt_row_jobs_f32(const ptt_row_job* jobs, int n_jobs, ptt_stream_t stream) {
    if (!jobs || n_jobs < 1 || n_jobs > PTT_ROW_JOBS_MAX) return fail(PTT_EINVAL, "ptt_row_jobs_f32: n_jobs=%d (1..%d)", n_jobs, PTT_ROW_JOBS_MAX);
    RowJobsParams P;
    P.n = 0;
    int blocks = 0, lds = 0;
    for (int i = 0; i < n_jobs; ++i) {
        const ptt_row_job& j = jobs[i];
        if (j.rows < 0 || j.K <= 0 || j.K > 1024 || j.Cout <= 0)
            return fail(PTT_EINVAL, "ptt_row_jobs_f32: job %d rows=%d K=%d Cout=%d", i, j.rows, j.K, j.Cout);
        if (j.rows == 0) continue;
        if (!j.Wpacked || !j.out) return fail(PTT_EINVAL, "ptt_row_jobs_f32: job %d: null weights / output", i);
        if (j.prologue == 0) {
            if (!j.X || j.K1 <= 0 || j.K1 > j.K || j.ldx < j.K1 || (j.K1 < j.K && (!j.X2 || j.ldx2 < j.K - j.K1)))
                return fail(PTT_EINVAL, "ptt_row_jobs_f32: job %d: K=%d K1=%d ldx=%d ldx2=%d", i, j.K, j.K1, j.ldx, j.ldx2);
            if (j.Xmax && (j.K1 != j.K || (j.K & 3) || (j.ldx & 3) || !aligned16(j.X) || !aligned16(j.Xmax)))
                return fail(PTT_EINVAL, "ptt_row_jobs_f32: job %d: Xmax needs one input of K %% 4 == 0 channels in 16-byte aligned rows", i);
        } else if (j.prologue == 1) {
            if (!j.rel || !j.w1 || !aligned16(j.w1)) return fail(PTT_EINVAL, "ptt_row_jobs_f32: job %d: prologue 1 needs rel and a 16-byte aligned w1", i);
        } else if (j.prologue == 2) {
            if (!j.qkv || !j.knn || !j.pos || j.N <= 0 || (j.K & 3) || (j.ldq & 3) || (j.ldp & 3) || (j.q_off & 3) || (j.k_off & 3) ||
                !aligned16(j.qkv) || !aligned16(j.pos) || (j.rows & 15))
                return fail(PTT_EINVAL, "ptt_row_jobs_f32: job %d: prologue 2 needs q|k|v, knn, pos, 16 rows per point and 16-byte aligned rows", i);
        } else if (j.prologue == 3) {
            if (!j.X || !j.idx || !j.xyz || !j.centres || !j.wx || j.N <= 0 || j.M <= 0 || j.ns <= 0 || (j.K & 3) || (j.ldx & 3) ||
                !aligned16(j.X) || !aligned16(j.wx) || (j.normalize_xyz && !(j.radius > 0.f)))
                return fail(PTT_EINVAL, "ptt_row_jobs_f32: job %d: prologue 3 needs term rows, idx, xyz, centres, wx (16-byte aligned, K %% 4 == 0)", i);
        } else return fail(PTT_EINVAL, "ptt_row_jobs_f32: job %d: prologue %d", i, j.prologue);
        if (j.epilogue == 1) {
            if (!j.qkv || !j.knn || !j.pos || j.N <= 0 || (j.rows & 15) || j.ldo < j.Cout || j.ldp < j.Cout)
                return fail(PTT_EINVAL, "ptt_row_jobs_f32: job %d: epilogue 1 needs q|k|v, knn, pos and 16 rows per point", i);
        } else if (j.epilogue == 2) {
            if ((j.ns != 16 && j.ns != 32) || (j.rows % j.ns) || j.ldo < j.Cout || j.act < 0 || j.act > 2)
                return fail(PTT_EINVAL, "ptt_row_jobs_f32: job %d: epilogue 2 pools ns = 16 or 32 rows per centre (ns=%d rows=%d)", i, j.ns, j.rows);
        } else if (j.epilogue == 0) {
            if (j.out_split < 0 || j.out_split > j.Cout || (j.out_split > 0 && !j.out2) || j.res_split < 0 || j.res_split > j.Cout ||
                j.act < 0 || j.act > 2)
                return fail(PTT_EINVAL, "ptt_row_jobs_f32: job %d: out_split=%d res_split=%d act=%d", i, j.out_split, j.res_split, j.act);
        } else return fail(PTT_EINVAL, "ptt_row_jobs_f32: job %d: epilogue %d", i, j.epilogue);
        RowJobDev& D = P.j[P.n];
        D.j = j;
        D.NT = (j.Cout + 31) / 32;
        D.nkb = (j.K + 7) / 8;
        const int rt = (j.rows + 31) / 32;
        int cw = j.col_tiles;
        if (cw != 1 && cw != 2 && cw != 4) {
            // the widest tile that still gives every CU a workgroup; below that, the most workgroups
            cw = 1;
            if (rt * ((D.NT + 3) / 4) >= 224) cw = 4;
            else if (rt * ((D.NT + 1) / 2) >= 224) cw = 2;
        }
        D.cw = cw; D.ks = 8 / cw;
        D.hb = ((D.nkb + 7) / 8 + 3) / 4 * 4;        // K-blocks per slice: 8 slices whatever the tile shape (see the kernel)
        D.ldk = 8 * D.hb * 8 + 4;
        D.cg = (D.NT + cw - 1) / cw;
        D.blk0 = blocks;
        D.vecx = (j.prologue == 0 && (j.ldx & 3) == 0 && aligned16(j.X)) ? 1 : 0;
        D.vecp = 0;
        blocks += rt * D.cg;
        const int need = (32 * D.ldk + (D.ks - 1) * cw * 16 * 64) * (int)sizeof(float);
        if (need > lds) lds = need;
        ++P.n;
    }
    if (P.n == 0) return PTT_OK;
    if (lds > 160 * 1024) return fail(PTT_EUNSUPPORTED, "ptt_row_jobs_f32: %d bytes of LDS per workgroup", lds);
    int pros = 0, epi1 = 0, pd = 8;
    for (int i = 0; i < P.n; ++i) {
        pros |= 1 << P.j[i].j.prologue;
        epi1 |= P.j[i].j.epilogue ? 1 << (P.j[i].j.epilogue - 1) : 0;
        if (P.j[i].hb & 7) pd = 4;
    }
    hipStream_t s = as_stream(stream);
    int rc = PTT_OK;
    bool done = false;
#define PTT_RJ_CASE(PR, EP, PDV)                                                                                         \
    if (!done && (pros & ~(PR)) == 0 && (epi1 & ~(EP)) == 0 && pd == (PDV)) {                                               \
        if ((rc = set_lds_limit(reinterpret_cast<const void*>(rowjobs_kernel<PR, EP, PDV>), lds))) return rc;           \
        hipLaunchKernelGGL((rowjobs_kernel<PR, EP, PDV>), dim3(blocks), dim3(512), lds, s, P);                          \
        done = true;                                                                                                    \
    }
    PTT_RJ_CASE(1, 0, 4) PTT_RJ_CASE(1, 0, 8)                 // plain layers (K < 512 / K >= 512)
    PTT_RJ_CASE(3, 0, 4) PTT_RJ_CASE(3, 0, 8)                 // q|k|v beside fc_delta
    PTT_RJ_CASE(4, 0, 8)                                      // fc_gamma[0] on the pair input
    PTT_RJ_CASE(1, 1, 8)                                      // fc_gamma[2] + softmax / weighted sum
    PTT_RJ_CASE(8, 0, 4) PTT_RJ_CASE(1, 2, 4)                 // a set-abstraction level: grouped layer 1, last layer + max-pool
    PTT_RJ_CASE(15, 3, 4) PTT_RJ_CASE(15, 3, 8)               // anything else
#undef PTT_RJ_CASE
    return check_launch("rowjobs_kernel");
}
