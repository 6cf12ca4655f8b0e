// Row GEMMs of the TRAINING step (N3) on exact-fp32 MFMA, gfx950:
//   rows_gemm_kernel   Y[R, N] = act_in(X)[R, K] . W^T (+ bias, ReLU, residual; optional column statistics of Y)
//                      forward and input gradient of every 1x1 convolution / nn.Linear over 10^4 - 10^6 rows
//                      (pytorch_utils.py:12-36 SharedMLP, transformer_block/variants.py:154-165, voting heads)
//   wgrad2_kernel      dW[N, K] = dZ^T[N, R] . X[R, K]      the weight gradients of the same layers
// The inference linear kernel (mfma_ops.hip: linear_kernel) launches one short-lived workgroup per 32-row tile: fetch,
// barrier, GEMM, epilogue in sequence, overlap only between the two workgroups of a CU. At 10^5 rows that leaves the
// matrix pipe 35-45 % idle (35-112 TFLOP/s against hipBLASLt's 55-138 on the same shapes, scripts/rows_gemm_bench.py).
// Here a PERSISTENT workgroup walks (row tile, K chunk) units: the global loads of unit u+1 are issued before unit u's
// first MFMAs and written to the other LDS buffer inside the second half of unit u's MFMA stream (with the deferred
// BatchNorm+ReLU of the producing layer applied in registers on the way), weight fragments stream from L2 two K-blocks
// ahead and across unit boundaries, ONE workgroup barrier per unit (LDS traffic only: global loads stay in flight across
// it). Same operand layouts as mfma_ops.hip: packed weights [K/8][N/32][64 lanes][4], LDS row stride == 4 (mod 8) floats.
#include <atomic>
#include <math.h>
#include <stdlib.h>
#include <type_traits>
#include "common.h"

namespace ptt {

typedef float g32x16 __attribute__((ext_vector_type(16)));
typedef float g32x4 __attribute__((ext_vector_type(4)));
typedef int gi32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int gu32x4 __attribute__((__vector_size__(4 * sizeof(unsigned int))));

namespace {

__device__ __forceinline__ __amdgpu_buffer_rsrc_t g_rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);   // raw, 32-bit offsets
}
__device__ __forceinline__ g32x4 g_load4(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(g32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ float g_load1(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ void g_store1(float v, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), r, voff, soff, 0);
}
// workgroup barrier that orders LDS traffic only (global loads / stores stay in flight across it)
__device__ __forceinline__ void g_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ int g_tile_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }
// lane <-> lane^32 sum on v_permlane32_swap (inline asm: hipcc folds the builtin when both operands are equal)
__device__ __forceinline__ float g_add_halves(float v) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return a + b;
}
// (value, row) of lane ^ 32 beside this lane's: lo = the lower half-wave's pair in both halves, hi = the upper half-wave's
__device__ __forceinline__ void g_swap_pair(float v, int i, float& lo_v, int& lo_i, float& hi_v, int& hi_i) {
    float a = v, b = v;
    int c = i, d = i;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(c), "+v"(d));
    lo_v = a; hi_v = b; lo_i = c; hi_i = d;
}
// the dispatcher places block b on XCD b % 8: consecutive LOGICAL blocks share an XCD (and its L2)
__device__ __forceinline__ int g_logical_block() {
    const int bid = blockIdx.x, per = (int)gridDim.x >> 3;
    if (bid >= (per << 3)) return bid;
    return (bid & 7) * per + (bid >> 3);
}

}  // namespace

struct RowsGemmParams {
    const float* X; const float* Wp; const float* bias; const float* residual; float* out;
    const float* in_a; const float* in_b;       // optional x <- relu(x * in_a[k] + in_b[k]) while the A operand is staged
    const float* mask; int ldm;                 // optional out <- mask > 0 ? out : 0 (the ReLU backward of the layer this
                                                // GEMM is the input gradient of; mask = that layer's output), before the statistics
    const float* bz; const float* bmean; const float* binv; const float* ba; const float* bb; int ldbz;
                                                // optional (with stats): the launch is the INPUT GRADIENT g of a layer whose input
                                                // is relu(BatchNorm(bz)) — its epilogue also takes the BatchNorm backward sums of
                                                // that producing layer, sum dy and sum dy * xhat (dy = g where bz * ba + bb > 0,
                                                // xhat = (bz - bmean) * binv), instead of the statistics of the output
    double* stats;                              // optional [chunks][2][N] partial column sums / sums of squares of Y
    int rows, K, ldx, N, NT, relu, ldr, ldo, ntiles, G, ncg, nchunks;
    // POOL instantiations: per group of POOL consecutive rows and column the largest and the smallest y with the FIRST row (inside
    // the group) that holds it — what the max-pool of relu(BatchNorm(y)) needs once the batch statistics (which this same launch
    // sums) are known: the sign of gamma * invstd picks max or min (ptt_pool_select_f32)
    float* pmax; float* pmin; int* amax; int* amin;
    // AIN instantiations: the A operand is NOT read from memory as it stands, it is the BatchNorm + ReLU backward of the layer this
    // launch is the input gradient of, formed while the rows are staged (the pass that used to write it is gone):
    //   dz = c0 + c1 * (z - mean) + (z * in_a + in_b > 0 [and the row is its group's arg-max] ? k1 * g : 0)
    // AIN 2: X = the dense gradient g (rows, K); AIN 16 / 32 / 64: X = z itself and the gradient is POOLED over groups of AIN rows,
    // tg (rows / AIN, K) with the arg-max row of every (group, channel) in targ. tz / ldtz = z for AIN 2. a_out (optional): dz
    // written out once (by the workgroups of column group 0) for the weight gradient of the same layer.
    const float* tz; int ldtz; const float* tk1; const float* tc0; const float* tc1; const float* tmu;
    const float* tg; int ldtg; const int* targ; float* a_out; int lda_out;
    // GS instantiations (with `residual`): the product is ALSO written without the residual (plain), and summed per group of 16
    // consecutive rows and column (gsum)
    float* gsum; float* plain; int ldgs, ldpl;
};

// WR x WC waves (WR * WC = 4): wave (wr, wc) owns row tiles wr*RT .. wr*RT+RT-1 of the workgroup's 32*RT*WR rows and the
// column tiles cg*WC*CT + wc + WC*u, u < CT, of its column group; KC input channels per staged chunk.
// EXP (timing experiments of a -DPTT_GEMM_DEV build only, wrong results): 1 no output stores, 2 no row fetch / staging,
// 4 every weight fragment from K-block 0 (L1-resident), 8 no barriers
#ifndef PTT_RG_PD
#define PTT_RG_PD 3          // weight fragments requested this many K-blocks ahead
#endif
// BNB: the statistics are the BatchNorm backward sums of the producing layer (p.bz ...), not those of the output
// GS 16: the launch also sums its outputs (after the residual) over groups of 16 consecutive rows (ptt_rows_gemm_rsum16_f32)
template <int WR, int RT, int CT, int KC, bool STATS, bool ACT, int EXP = 0, bool BNB = false, int POOL = 0, int AIN = 0, int GS = 0>
__global__ __launch_bounds__(256, 2) void rows_gemm_kernel(RowsGemmParams p) {
    static_assert(GS == 0 || (GS == 16 && !STATS && !ACT && !BNB && POOL == 0 && AIN == 0), "group sums: a plain GEMM with a residual");
    static_assert(!BNB || (STATS && !ACT), "the backward-sums epilogue is a statistics epilogue of a plain input gradient");
    static_assert(AIN == 0 || AIN == 2 || AIN == 16 || AIN == 32 || AIN == 64, "A-operand form");
    static_assert(AIN == 0 || (!ACT && POOL == 0), "the BatchNorm-backward A operand excludes the deferred-activation one");
    static_assert(POOL == 0 || (STATS && !BNB && (POOL == 16 || POOL == 32 || (POOL == 64 && RT % 2 == 0))), "pooled statistics epilogue");
    constexpr int WC = 4 / WR, TR = 32 * RT * WR, NKB = KC / 8, LDK = KC + 4, BUF = TR * LDK, QPR = KC / 4;
    constexpr int SLOTS = TR * QPR / 256, PD = PTT_RG_PD;
    constexpr int WI = SLOTS < NKB / 2 ? SLOTS : NKB / 2;       // K-blocks (the last ones of a unit) that carry a staging piece
    constexpr int PIECES = SLOTS / WI;
    static_assert(SLOTS % WI == 0 && PIECES >= 1 && PD >= 1 && PD <= NKB, "unit shape");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int t = threadIdx.x, lane = t & 63, half = lane >> 5, col = lane & 31;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = w / WC, wc = w % WC;
    const int lb = g_logical_block();
    const int cg = lb % p.ncg, g = lb / p.ncg;
    if (g >= p.ntiles) return;
    const int ntw = (p.ntiles - g + p.G - 1) / p.G;             // this workgroup's tiles: g, g + G, ...
    const int nunits = ntw * p.nchunks;

    // X through a descriptor whose size ends with the last row: rows past the end read as zeros (the row part of an
    // address is in the VGPR offset, which the range check covers; the chunk part in the scalar offset never leaves a row)
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X), 0,
                                                                         ((p.rows - 1) * p.ldx + p.K) * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = g_rsrc(p.Wp), ro = g_rsrc(p.out);
    const int ct0 = cg * (WC * CT) + wc;
    const int wvoff = (ct0 * 64 + lane) * 16;                   // this lane's byte offset inside a K-block row of fragments
    const int wkstep = p.NT * 1024;                             // bytes per K-block of the packed weights
    // staging slots of this thread: slot i = float4 (row r0 + i * RSTEP, channel quad q) of the unit's [TR][KC] block
    constexpr int RSTEP = 256 / QPR;
    const int q = t % QPR, r0 = t / QPR;
    const int lds_slot = (r0 * LDK + 4 * q);                    // + i * RSTEP * LDK floats
    const int ldx4 = p.ldx * 4;
    const int slot_off = r0 * ldx4 + q * 16;                    // + i * RSTEP * ldx4 + tile * TR * ldx4 bytes
    const int tile_bytes = TR * ldx4;

    g32x16 acc[RT][CT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int u = 0; u < CT; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rt][u][r] = 0.f;
    double dsum[CT], dsq[CT];
#pragma unroll
    for (int u = 0; u < CT; ++u) { dsum[u] = 0.0; dsq[u] = 0.0; }

    g32x4 st[SLOTS];
    g32x4 ta = {1.f, 1.f, 1.f, 1.f}, tb = {0.f, 0.f, 0.f, 0.f};
    // AIN: z beside the gradient (AIN 2) / the pooled gradient and the arg-max of the row's group beside z (AIN 3), the five
    // per-channel constants of the chunk, and where the staged dz goes when this workgroup writes it out
    constexpr bool PIN = AIN >= 16;                             // pooled gradient
    constexpr int TS = AIN == 2 ? SLOTS : 1, NG = PIN ? TR / AIN : 1, SPG = SLOTS / NG;      // slots of a thread per group
    static_assert(!PIN || (TR % AIN == 0 && AIN % (256 / QPR) == 0 && SLOTS % NG == 0), "a thread's slots fall into whole groups");
    g32x4 sz[TS], sg[NG];
    gi32x4 sarg[NG];
    g32x4 tk1 = {0.f, 0.f, 0.f, 0.f}, tc0 = tk1, tc1 = tk1, tmu = tk1;
    int aout_off = 0;
    const bool write_a = AIN != 0 && p.a_out != nullptr && cg == 0;
    const __amdgpu_buffer_rsrc_t rtz = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(AIN == 2 ? p.tz : p.X), 0,
                                                                          AIN == 2 ? ((p.rows - 1) * p.ldtz + p.K) * 4 : 0, 0x00020000);
    const int groups = PIN ? p.rows / (PIN ? AIN : 1) : 1;
    const __amdgpu_buffer_rsrc_t rtg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(PIN ? p.tg : p.X), 0,
                                                                          PIN ? ((groups - 1) * p.ldtg + p.K) * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rta = __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(PIN ? p.targ : nullptr), 0,
                                                                          PIN ? groups * p.K * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rao = __builtin_amdgcn_make_buffer_rsrc(AIN ? p.a_out : nullptr, 0,
                                                                          (AIN && p.a_out) ? ((p.rows - 1) * p.lda_out + p.K) * 4 : 0, 0x00020000);
    auto fetch = [&](int tile, int c) {
        // tiles past the last one (the prefetch of the final unit): any offset beyond the descriptor reads zeros
        const int tcl = tile < p.ntiles ? tile : p.ntiles;
        const int tb_off = tcl * tile_bytes + slot_off;
#pragma unroll
        for (int i = 0; i < SLOTS; ++i) st[i] = g_load4(rx, tb_off + i * (RSTEP * ldx4), c * (KC * 4));
        if (ACT || AIN) {
            ta = *reinterpret_cast<const g32x4*>(p.in_a + c * KC + 4 * q);
            tb = *reinterpret_cast<const g32x4*>(p.in_b + c * KC + 4 * q);
        }
        if constexpr (AIN != 0) {
            tk1 = *reinterpret_cast<const g32x4*>(p.tk1 + c * KC + 4 * q);
            tc0 = *reinterpret_cast<const g32x4*>(p.tc0 + c * KC + 4 * q);
            tc1 = *reinterpret_cast<const g32x4*>(p.tc1 + c * KC + 4 * q);
            tmu = *reinterpret_cast<const g32x4*>(p.tmu + c * KC + 4 * q);
            // row part AND chunk part in the vector offset, no scalar offset: a 128-bit buffer store WITH an SGPR offset showed the
            // "store data overwritten by the next VALU instruction" hazard on gfx950 (element 1 of a quad, lanes 12-15 of every 16,
            // one launch in a few) that hipcc only guards against for stores WITHOUT one (scripts/probes/fused_bnbwd_probe.py);
            // the range check still drops rows past the end (lda_out >= K)
            aout_off = (tcl * TR + r0) * (p.lda_out * 4) + q * 16 + c * (KC * 4);
        }
        if constexpr (AIN == 2) {
            const int z_off = (tcl * TR + r0) * (p.ldtz * 4) + q * 16;
#pragma unroll
            for (int i = 0; i < SLOTS; ++i) sz[i] = g_load4(rtz, z_off + i * (RSTEP * p.ldtz * 4), c * (KC * 4));
        }
        if constexpr (PIN) {        // slot i lies in group i / SPG of the tile (RSTEP divides the group size, r0 < RSTEP)
#pragma unroll
            for (int gq = 0; gq < NG; ++gq) {
                const int grp = tcl * NG + gq;
                sg[gq] = g_load4(rtg, grp * (p.ldtg * 4) + q * 16, c * (KC * 4));
                sarg[gq] = __builtin_bit_cast(gi32x4, g_load4(rta, grp * (p.K * 4) + q * 16, c * (KC * 4)));
            }
        }
    };
    auto stage = [&](float* buf, int i) {                       // deferred BatchNorm + ReLU of the producing layer, LDS write
        g32x4 v = st[i];
        if (ACT) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = fmaxf(__builtin_fmaf(v[k], ta[k], tb[k]), 0.f);
        }
        if constexpr (AIN == 2) {
            const g32x4 z = sz[i];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float t = __builtin_fmaf(tc1[k], z[k] - tmu[k], tc0[k]);
                v[k] = __builtin_fmaf(z[k], ta[k], tb[k]) > 0.f ? __builtin_fmaf(tk1[k], v[k], t) : t;
            }
        }
        if constexpr (PIN) {
            const g32x4 z = v, gp = sg[i / SPG];
            const gi32x4 ar = sarg[i / SPG];
            const int srow = r0 + (i % SPG) * RSTEP;            // this slot's row inside its group
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float t = __builtin_fmaf(tc1[k], z[k] - tmu[k], tc0[k]);
                v[k] = (ar[k] == srow && __builtin_fmaf(z[k], ta[k], tb[k]) > 0.f) ? __builtin_fmaf(tk1[k], gp[k], t) : t;
            }
        }
        *reinterpret_cast<g32x4*>(buf + lds_slot + i * (RSTEP * LDK)) = v;
        if constexpr (AIN != 0) {
            if (write_a)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(gu32x4, v), rao, aout_off + i * (RSTEP * p.lda_out * 4), 0, 0);
        }
    };

    // bias / ReLU branch-free: a missing bias reads any valid address and is replaced by 0, no ReLU = floor -inf
    const bool has_bias = p.bias != nullptr;
    const __amdgpu_buffer_rsrc_t rb = g_rsrc(has_bias ? p.bias : p.Wp);
    const __amdgpu_buffer_rsrc_t rr = g_rsrc(p.residual ? p.residual : (p.mask ? p.mask : (p.bz ? p.bz : p.X)));
    const int ldr = p.residual ? p.ldr : (p.mask ? p.ldm : p.ldbz);
    const float rfloor = p.relu ? 0.f : -__builtin_inff();
    auto epilogue = [&](int row_w, auto full_c, auto mode_c) {      // mode 0: plain, 1: + residual, 2: masked by `mask` > 0,
                                                                    // 3: BatchNorm backward sums of the producing layer
        constexpr bool FULL = decltype(full_c)::value;
        constexpr int MODE = decltype(mode_c)::value;
        constexpr bool RES = MODE != 0;
#pragma unroll
        for (int u = 0; u < CT; ++u) {
            const int cn = (ct0 + u * WC) * 32 + col;
            const float bl = g_load1(rb, cn * 4, 0);
            const float bv = has_bias ? bl : 0.f;
            const int obase = ((row_w + 4 * half) * p.ldo + cn) * 4, rbase = ((row_w + 4 * half) * ldr + cn) * 4;
            // column statistics without cancellation: sums of (y - K) and (y - K)^2 in float32 around a per-lane reference value
            // K (this lane's first value: within the column's spread), un-shifted in float64 — sum y = s + n K,
            // sum y^2 = q + 2 K s + n K^2. (Plain float32 sums of y^2 lose mean^2 / var digits in E[y^2] - mean^2: a BatchNorm over
            // a nearly constant channel then amplifies 1e-7 into 1e-4.)
            const float kref = MODE == 3 ? 0.f : acc[0][u][0] + bv;     // the lane's first row: if that one is past the end, all of its rows are
            float s = 0.f, sq = 0.f;
            int nrows = 0;
            float p64max = 0.f, p64min = 0.f;
            int p64imax = 0, p64imin = 0;
            float cm = 0.f, ci = 0.f, ca = 0.f, cb = 0.f;
            if (MODE == 3) { cm = p.bmean[cn]; ci = p.binv[cn]; ca = p.ba[cn]; cb = p.bb[cn]; }
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                float rv[16];
                if (RES) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int dr = rt * 32 + (r & 3) + 8 * (r >> 2);
                        const int grc = FULL ? 0 : ((row_w + 4 * half + dr < p.rows) ? 0 : (p.rows - 1 - (row_w + 4 * half + dr)));
                        rv[r] = g_load1(rr, rbase + (dr + grc) * (ldr * 4), 0);        // rows past the end: clamped, never stored
                    }
                }
                float pv_max[2] = {-__builtin_inff(), -__builtin_inff()}, pv_min[2] = {__builtin_inff(), __builtin_inff()};
                int pi_max[2] = {0, 0}, pi_min[2] = {0, 0};
                float gs[2] = {0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dr = rt * 32 + (r & 3) + 8 * (r >> 2);
                    const bool in = FULL || row_w + 4 * half + dr < p.rows;
                    float y = acc[rt][u][r] + bv;
                    if (MODE == 2) y = rv[r] > 0.f ? y : 0.f;
                    if (POOL > 0 && MODE == 0) {                // rows ascend with r inside a lane: strict comparisons keep the first
                        const int gq = POOL == 16 ? (r >> 3) : 0;
                        const int rowg = (dr + 4 * half) % (POOL > 0 ? POOL : 1);
                        if (in && y > pv_max[gq]) { pv_max[gq] = y; pi_max[gq] = rowg; }
                        if (in && y < pv_min[gq]) { pv_min[gq] = y; pi_min[gq] = rowg; }
                    }
                    if (STATS && MODE == 3) {                   // sum dy, sum dy * xhat of the layer this gradient flows into
                        const float z = rv[r];
                        const float dyv = (in && __builtin_fmaf(z, ca, cb) > 0.f) ? y : 0.f;
                        s += dyv; sq = __builtin_fmaf(dyv, (z - cm) * ci, sq);
                    } else if (STATS) {                         // rows past the end take no part
                        const float d = in ? y - kref : 0.f;
                        s += d; sq = __builtin_fmaf(d, d, sq);
                        if (!FULL) nrows += in ? 1 : 0;
                    }
                    y = fmaxf(y, rfloor);
                    if (GS == 16 && MODE == 1) {                // a lane's registers 0-7 / 8-15: the tile's two 16-row groups
                        gs[r >> 3] += in ? y : 0.f;
                        if (in) p.plain[(size_t)(row_w + 4 * half + dr) * p.ldpl + cn] = y;
                    }
                    if (MODE == 1) y += rv[r];
                    if (!(EXP & 1) || r == 15)
                        if (in) g_store1(y, ro, obase + dr * (p.ldo * 4), 0);
                    acc[rt][u][r] = 0.f;
                }
                if constexpr (GS == 16 && MODE == 1) {
#pragma unroll
                    for (int gq = 0; gq < 2; ++gq) {            // the other half-wave holds the group's other eight rows
                        const float tot = g_add_halves(gs[gq]);
                        const int row_g = row_w + rt * 32 + gq * 16;
                        if (half == 0 && row_g < p.rows) p.gsum[(size_t)(row_g >> 4) * p.ldgs + cn] = tot;
                    }
                }
                if constexpr (POOL > 0 && MODE == 0) {
                    // the other half-wave holds the group's other rows; a 64-row group spans two row tiles of this wave
                    constexpr int NG = POOL == 16 ? 2 : 1;
#pragma unroll
                    for (int gq = 0; gq < NG; ++gq) {
                        float lv, hv; int li, hi;
                        g_swap_pair(pv_max[gq], pi_max[gq], lv, li, hv, hi);
                        const bool th = hv > lv || (hv == lv && hi < li);
                        float mx = th ? hv : lv; int mxi = th ? hi : li;
                        g_swap_pair(pv_min[gq], pi_min[gq], lv, li, hv, hi);
                        const bool tl = hv < lv || (hv == lv && hi < li);
                        float mn = tl ? hv : lv; int mni = tl ? hi : li;
                        if (POOL == 64) {
                            if ((rt & 1) == 0) { p64max = mx; p64min = mn; p64imax = mxi; p64imin = mni; continue; }
                            // second tile of the group: its rows come later, so ties stay with the first tile
                            if (!(mx > p64max)) { mx = p64max; mxi = p64imax; }
                            if (!(mn < p64min)) { mn = p64min; mni = p64imin; }
                        }
                        const int row_g = row_w + rt * 32 - (POOL == 64 ? 32 : 0) + gq * 16;      // first row of the group
                        if (half == 0 && row_g < p.rows) {
                            const size_t o = (size_t)(row_g / POOL) * p.N + cn;
                            p.pmax[o] = mx; p.pmin[o] = mn; p.amax[o] = mxi; p.amin[o] = mni;
                        }
                    }
                }
            }
            if (STATS && MODE == 3) {
                dsum[u] += (double)s;
                dsq[u] += (double)sq;
            } else if (STATS) {
                const double K = (double)kref, S = (double)s, n = FULL ? 16.0 * RT : (double)nrows;
                dsum[u] += S + n * K;
                dsq[u] += (double)sq + 2.0 * K * S + n * K * K;
            }
        }
    };

    // ---- prologue: the first unit is staged in the open ----
    int tile = g, c = 0;
    fetch(tile, c);
    g32x4 wq[PD][CT];                                           // the weights of the current unit's first PD K-blocks
#pragma unroll
    for (int k = 0; k < PD; ++k)
#pragma unroll
        for (int u = 0; u < CT; ++u) wq[k][u] = g_load4(rw, wvoff, k * wkstep + u * (WC * 1024));
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) stage(smem, i);
    g_lds_barrier();

    for (int un = 0; un < nunits; ++un) {
        int c_n = c + 1, tile_n = tile;
        if (c_n == p.nchunks) { c_n = 0; tile_n = tile + p.G; }
        float* cur = smem + (un & 1) * BUF;
        float* nxt = smem + ((un + 1) & 1) * BUF;
        const int wk_cur = c * (NKB * wkstep), wk_nxt = c_n * (NKB * wkstep);
        const float* arow = cur + (wr * RT * 32 + (lane & 31)) * LDK + 4 * half;
        g32x4 av[2][RT];
        g32x4 wv[NKB + PD][CT];
#pragma unroll
        for (int k = 0; k < PD; ++k)
#pragma unroll
            for (int u = 0; u < CT; ++u) wv[k][u] = wq[k][u];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) av[0][rt] = *reinterpret_cast<const g32x4*>(arow + rt * 32 * LDK);
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            {   // weight fragments PD K-blocks ahead; past the end of this unit they are the next unit's first ones
                const int kq = kb + PD;
                const int base = (EXP & 4) ? 0 : (kq < NKB ? wk_cur + kq * wkstep : wk_nxt + (kq - NKB) * wkstep);
#pragma unroll
                for (int u = 0; u < CT; ++u) wv[kq][u] = g_load4(rw, wvoff, base + u * (WC * 1024));
            }
            if (kb + 1 < NKB) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
                    av[(kb + 1) & 1][rt] = *reinterpret_cast<const g32x4*>(arow + rt * 32 * LDK + (kb + 1) * 8);
            }
            if (kb == 0 && !(EXP & 2)) fetch(tile_n, c_n);      // the next unit's rows: in flight under this unit's MFMAs
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int u = 0; u < CT; ++u)
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt)
                        acc[rt][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kb & 1][rt][j], wv[kb][u][j], acc[rt][u], 0, 0, 0);
            if (kb >= NKB - WI && !(EXP & 2)) {                 // the tail of the stream: the row loads have landed
#pragma unroll
                for (int k = 0; k < PIECES; ++k) stage(nxt, (kb - (NKB - WI)) * PIECES + k);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int k = 0; k < PD; ++k)
#pragma unroll
            for (int u = 0; u < CT; ++u) wq[k][u] = wv[NKB + k][u];
        if (c == p.nchunks - 1) {
            // ---- epilogue of a row tile (in the open: the co-resident workgroup owns the matrix pipe meanwhile) ----
            const int row_w = tile * TR + wr * RT * 32;
            typedef std::integral_constant<int, 0> M0;
            typedef std::integral_constant<int, 1> M1;
            typedef std::integral_constant<int, 2> M2;
            typedef std::integral_constant<int, 3> M3;
            if constexpr (BNB) {
                if (row_w + RT * 32 <= p.rows) epilogue(row_w, std::true_type(), M3());
                else epilogue(row_w, std::false_type(), M3());
            } else if (row_w + RT * 32 <= p.rows) {
                if (p.residual) epilogue(row_w, std::true_type(), M1());
                else if (p.mask) epilogue(row_w, std::true_type(), M2());
                else epilogue(row_w, std::true_type(), M0());
            } else {
                if (p.residual) epilogue(row_w, std::false_type(), M1());
                else if (p.mask) epilogue(row_w, std::false_type(), M2());
                else epilogue(row_w, std::false_type(), M0());
            }
        }
        if (!(EXP & 8)) g_lds_barrier();
        tile = tile_n; c = c_n;
    }
    if (STATS) {                                                // the two half-waves hold different rows of the same columns
        double* sp = p.stats + (size_t)(g * WR + wr) * 2 * p.N;
#pragma unroll
        for (int u = 0; u < CT; ++u) {
            const double a = dsum[u] + __shfl_xor(dsum[u], 32, 64), b = dsq[u] + __shfl_xor(dsq[u], 32, 64);
            const int cn = (ct0 + u * WC) * 32 + col;
            if (half == 0) { sp[cn] = a; sp[p.N + cn] = b; }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Weight gradient dW[o, i] = sum_r dZ[r, o] * X[r, i]: the reduction axis is the ROW axis. linear_wgrad_kernel
// (train_ops.hip) gives a workgroup a 128 x 128 output block, so a 256 x 256 gradient reads both operands twice
// (32 flop per byte: HBM-bound at ~half the matrix peak). Here 8 waves own up to 256 x 256 outputs (wave tile
// 128 x 64 = 8 accumulator tiles), each operand row is read once, 32-row sub-chunks double-buffered in LDS (one
// barrier each). Partials per row chunk go to the workspace and are summed in chunk order by wgrad_finish_kernel.
// ------------------------------------------------------------------------------------------
// EXP (timing experiments of a -DPTT_GEMM_DEV build only, wrong results): 1 no staging writes in the loop, 2 no row requests in the
// loop, 4 no barriers, 8 the fragments of step 0 for every step, 16 no partial stores
template <int TN, int TK, int EXP = 0>   // wave tile (32 TN) x (32 TK); waves 2 (N) x 4 (K); block BN = 64 TN, BK = 128 TK
__global__ __launch_bounds__(512, 1) void wgrad2_kernel(const float* __restrict__ dZ, int ldz, const float* __restrict__ X, int ldx,
                                                       int R, int Cout, int Cin, int nbk, int chunk_rows,
                                                       float* __restrict__ partial, const float* __restrict__ xa,
                                                       const float* __restrict__ xb) {
    constexpr int BN = 64 * TN, BK = 128 * TK, LDA = BN + 4, LDB = BK + 4, RS = 32;
    constexpr int QA = BN / 4, QB = BK / 4, SA = RS * QA / 512, SB = RS * QB / 512;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const As = smem;                         // [2][RS][LDA]
    float* const Bs = smem + 2 * RS * LDA;          // [2][RS][LDB]
    const int t = threadIdx.x, lane = t & 63, half = lane >> 5, col = lane & 31;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6), wn = w >> 2, wk = w & 3;
    // the output tiles of ONE row chunk read the same rows of dZ / X (a 512 x 512 gradient in 256 x 256 tiles: every operand
    // quad twice): consecutive LOGICAL blocks share an XCD, so the tiles of a chunk meet in that XCD's L2 instead of each
    // fetching its operands from the fabric (profiles/r04z_pmc_train_gemm: 805 MB read per launch against 402 MB compulsory)
    const int ntile = (Cout / BN) * nbk;
    const int lb = g_logical_block();
    const int tile_id = lb % ntile, chunk = lb / ntile;
    const int bo = tile_id / nbk, bi = tile_id - bo * nbk;
    const int o0 = bo * BN, i0 = bi * BK;
    const int r_begin = chunk * chunk_rows, r_end = min(R, r_begin + chunk_rows);
    const __amdgpu_buffer_rsrc_t rz = g_rsrc(dZ), rxx = g_rsrc(X);
    const int qa = t % QA, ra = t / QA, qb = t % QB, rb = t / QB;
    constexpr int RSA = 512 / QA, RSB = 512 / QB;
    g32x4 ta = {1.f, 1.f, 1.f, 1.f}, tb = {0.f, 0.f, 0.f, 0.f};
    if (xa) { ta = *reinterpret_cast<const g32x4*>(xa + i0 + 4 * qb); tb = *reinterpret_cast<const g32x4*>(xb + i0 + 4 * qb); }

    g32x16 acc[TN][TK];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TK; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    // Rows in flight: sub-chunk k + 2 is requested at the top of iteration k into (na, nb) and handed over to (sa, sb) at the END of
    // that iteration — a whole sub-chunk (16 MFMA steps, ~7 us) to arrive — from where iteration k + 1 stages it. With one register
    // set (request at the top of the iteration that stages the rows from its 8th step on) the launch ran 1.6x slower on operands
    // that come from HBM than on operands the producing GEMM left in the Infinity Cache: it was waiting for its rows.
    g32x4 sa[SA], sb[SB], na[SA], nb[SB];
    int va = 0, vb = 0, nva = 0, nvb = 0;
    auto fetch = [&](int rs0) {
        nva = 0; nvb = 0;
#pragma unroll
        for (int i = 0; i < SA; ++i) {
            const int row = rs0 + ra + i * RSA;
            const int rc = row < r_end ? row : r_end - 1;
            if (row < r_end) nva |= 1 << i;
            na[i] = g_load4(rz, (rc * ldz + o0 + 4 * qa) * 4, 0);
        }
#pragma unroll
        for (int i = 0; i < SB; ++i) {
            const int row = rs0 + rb + i * RSB;
            const int rc = row < r_end ? row : r_end - 1;
            if (row < r_end) nvb |= 1 << i;
            nb[i] = g_load4(rxx, (rc * ldx + i0 + 4 * qb) * 4, 0);
        }
    };
    auto hand_over = [&]() {
#pragma unroll
        for (int i = 0; i < SA; ++i) sa[i] = na[i];
#pragma unroll
        for (int i = 0; i < SB; ++i) sb[i] = nb[i];
        va = nva; vb = nvb;
    };
    // staging piece i < SA + SB of the fetched sub-chunk: one float4 of dZ (i < SA) or of X (deferred activation applied)
    auto stage_piece = [&](int buf, int i) {
        if (i < SA) {
            g32x4 v = sa[i];
            if (!((va >> i) & 1)) v = g32x4{0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<g32x4*>(As + buf * (RS * LDA) + (ra + i * RSA) * LDA + 4 * qa) = v;
        } else {
            const int k2 = i - SA;
            g32x4 v = sb[k2];
            if (xa) {
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = fmaxf(__builtin_fmaf(v[k], ta[k], tb[k]), 0.f);
            }
            if (!((vb >> k2) & 1)) v = g32x4{0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<g32x4*>(Bs + buf * (RS * LDB) + (rb + k2 * RSB) * LDB + 4 * qb) = v;
        }
    };
    constexpr int NP = SA + SB;                                 // pieces per sub-chunk (4 .. 8), one behind each of the last j-steps
    static_assert(NP <= RS / 2, "one staging piece per MFMA step");
    fetch(r_begin);
    hand_over();
    fetch(r_begin + RS);                                        // past the end: clamped rows, zeroed when staged
#pragma unroll
    for (int i = 0; i < NP; ++i) stage_piece(0, i);
    hand_over();
    __syncthreads();
    int buf = 0;
    for (int rs0 = r_begin; rs0 < r_end; rs0 += RS) {
        if (!(EXP & 2)) fetch(rs0 + 2 * RS);
        const float* A = As + buf * (RS * LDA) + half * LDA + wn * (32 * TN) + col;
        const float* B = Bs + buf * (RS * LDB) + half * LDB + wk * (32 * TK) + col;
        // the operand fragments of step j + 1 are requested BEFORE the MFMAs of step j (two register sets): left to itself hipcc
        // placed each step's LDS reads right in front of its MFMAs, behind an s_waitcnt — the matrix pipe drained for an LDS round
        // trip every four MFMAs (profiles/r05p_pmc_train_gemm: 0.83 busy, 21 % of the wave cycles waiting)
        float av[2][TN], bv[2][TK];
        auto frag = [&](int j, int s) {
#pragma unroll
            for (int a = 0; a < TN; ++a) av[s][a] = A[2 * j * LDA + a * 32];
#pragma unroll
            for (int b = 0; b < TK; ++b) bv[s][b] = B[2 * j * LDB + b * 32];
        };
        frag(0, 0);
#pragma unroll
        for (int j = 0; j < RS / 2; ++j) {
            if (j + 1 < RS / 2) frag((EXP & 8) ? 0 : j + 1, (j + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
                for (int b = 0; b < TK; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j & 1][a], bv[j & 1][b], acc[a][b], 0, 0, 0);
            // the next sub-chunk goes to the OTHER buffer (free since the barrier that ended the previous iteration), one
            // piece behind each of the last MFMA steps: its global loads were issued RS/2 - NP steps ago
            if (!(EXP & 1) && j >= RS / 2 - NP) stage_piece(buf ^ 1, j - (RS / 2 - NP));
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!(EXP & 2)) hand_over();
        if (!(EXP & 4)) g_lds_barrier();
        buf ^= 1;
    }
    // C/D layout: column (input channel) = lane & 31, row (output channel) = (reg & 3) + 8 (reg >> 2) + 4 half
    float* P = partial + (size_t)chunk * Cout * Cin;
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TK; ++b) {
            const int ci = i0 + wk * (32 * TK) + b * 32 + col;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = o0 + wn * (32 * TN) + a * 32 + g_tile_row(r, half);
                if (!(EXP & 16) || r == 15) P[(size_t)co * Cin + ci] = acc[a][b][r];
            }
        }
}

// the fixed-order sum over row chunks (train_ops.hip)
void launch_wgrad_finish(const float* partial, int nchunks, size_t n, int accumulate, float* dW, hipStream_t s);

// Compute units of the CURRENT device, cached per device ordinal (a process may drive several device models; the launch shape —
// and with it the K-summation order of a 256-column layer — follows the device the launch goes to: results are bit-reproducible per
// device model, DESIGN.md section 2 "Batch dependence").
static int cu_count() {
    constexpr int MAX_DEV = 64;
    static std::atomic<int> cached[MAX_DEV];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return 256;
    int n = cached[dev].load(std::memory_order_relaxed);
    if (n == 0) {
        int v = 0;
        n = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
        cached[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

// launch geometry of ptt_rows_gemm_f32 (shared by the workspace query)
struct RowsGemmGeom { int WR, RT, CT, KC, TR, ncg, ntiles, G, chunks; bool ok; };
static RowsGemmGeom rows_gemm_geom(int rows, int K, int N) {
    RowsGemmGeom g{};
    g.ok = false;
    if (rows <= 0 || K <= 0 || N <= 0) return g;
    if (N % 128 == 0 && K % 128 == 0) {
        // 256 columns per workgroup (two accumulator tiles per row tile: each staged row feeds twice the MFMAs) — unless the launch is
        // too small to give every CU its two workgroups that way (the 6144-row layers of the transformer blocks and the heads are
        // 96 row tiles: 96 or 192 workgroups on 256 CUs): then 128 columns per workgroup, twice the workgroups. Measured on the
        // training step (round 5): always 256 columns 20.18 ms, 128 below one workgroup per CU 19.78, below two 19.70, below four 19.79.
        // (32-row tiles for launches that still leave CUs without a workgroup — 6144 rows x 256 columns is 192 workgroups: 19.34 against
        // 19.46 ms when every such launch takes them, but the statistics epilogues then change their partial-sum layout with the
        // launch size; for the plain launches alone 19.40 against 19.43: not kept)
        const int tiles64 = (rows + 63) / 64;
        g.WR = 1; g.RT = 2; g.KC = 128;
        g.CT = (N % 256 == 0 && (long long)tiles64 * (N / 256) >= 2 * cu_count()) ? 2 : 1;
    }
    else if (N % 128 == 0 && K % 64 == 0) { g.WR = 1; g.RT = 4; g.CT = 1; g.KC = 64; }
    else if (N % 64 == 0 && K % 64 == 0) { g.WR = 2; g.RT = 2; g.CT = 1; g.KC = 64; }
    else return g;
    const int WC = 4 / g.WR;
    g.TR = 32 * g.RT * g.WR;
    g.ncg = N / (32 * WC * g.CT);
    g.ntiles = (rows + g.TR - 1) / g.TR;
    int cap = 2 * cu_count() / g.ncg;                   // two workgroups per CU resident
    if (cap >= 8) cap &= ~7;
    if (cap < 1) cap = 1;
    g.G = g.ntiles < cap ? g.ntiles : cap;
    g.chunks = g.G * g.WR;
    g.ok = true;
    return g;
}

}  // namespace ptt

using namespace ptt;

extern "C" int ptt_rows_gemm_supported(int rows, int K, int N, int ldx, int ldo) {
    const RowsGemmGeom g = rows_gemm_geom(rows, K, N);
    if (!g.ok) return 0;
    if ((long long)rows * ldx >= (1LL << 29) || (long long)rows * ldo >= (1LL << 29) || (ldx & 3)) return 0;
    return 1;
}

extern "C" int ptt_rows_gemm_stat_chunks(int rows, int K, int N) {
    const RowsGemmGeom g = rows_gemm_geom(rows, K, N);
    return g.ok ? g.chunks : 0;
}

struct BnBwdArgs { const float* z; int ldz; const float* mean; const float* invstd; const float* a; const float* b; };
struct PoolArgs { float* pmax; float* pmin; int32_t* amax; int32_t* amin; int ns; };
struct GroupSumArgs { float* gsum; int ldgs; float* plain; int ldpl; };
static int rows_gemm_launch(const float* X, int rows, int K, int ldx, const float* in_scale, const float* in_shift,
                            const float* Wpacked, int N, const float* bias, int relu, const float* residual, int ldr,
                            const float* mask, int ldm, float* out, int ldo, double* stats, size_t stats_elems, ptt_stream_t stream,
                            const BnBwdArgs* bn = nullptr, const PoolArgs* pool = nullptr, const ptt_bn_bwd_input* ain = nullptr,
                            const GroupSumArgs* gs = nullptr);

extern "C" int ptt_rows_gemm_rsum16_supported(int rows, int K, int N, int ldx) {
    if (!ptt_rows_gemm_supported(rows, K, N, ldx, N) || rows % 16) return 0;
    const RowsGemmGeom g = rows_gemm_geom(rows, K, N);
    return (g.WR == 1 && g.RT == 2 && g.KC == 128) ? 1 : 0;
}

// plain = X @ W^T, out = plain + residual, gsum = the column sums of plain per group of 16 consecutive rows, from one epilogue: the
// input gradient dt of the pair layer of a Point-Transformer block, the gradient of pos_enc (dt + the aggregation's, which rides in
// as the residual) and the query gradient (dt summed over a point's 16 neighbours) — ptt_amd/train_ops.py: _AttnCore.
extern "C" int ptt_rows_gemm_rsum16_f32(const float* X, int rows, int K, int ldx, const float* Wpacked, int N, const float* residual,
                                        int ldr, float* out, int ldo, float* plain, int ldp, float* gsum, int ldg,
                                        ptt_stream_t stream) {
    if (!residual || !plain || !gsum || ldp < N || ldg < N) return fail(PTT_EINVAL, "ptt_rows_gemm_rsum16_f32: null pointer or ldp / ldg < N");
    if ((long long)rows * ldp >= (1LL << 31)) return fail(PTT_EUNSUPPORTED, "ptt_rows_gemm_rsum16_f32: rows * ldp >= 2^31");
    if (!ptt_rows_gemm_rsum16_supported(rows, K, N, ldx))
        return fail(PTT_EUNSUPPORTED, "ptt_rows_gemm_rsum16_f32: rows=%d K=%d N=%d (whole groups of 16 rows, K %% 128 == 0, N %% 128 == 0)", rows, K, N);
    const GroupSumArgs gs{gsum, ldg, plain, ldp};
    return rows_gemm_launch(X, rows, K, ldx, nullptr, nullptr, Wpacked, N, nullptr, 0, residual, ldr, nullptr, 0, out, ldo, nullptr, 0, stream,
                            nullptr, nullptr, nullptr, &gs);
}

extern "C" int ptt_rows_gemm_bnbwd_fused_supported(int rows, int K, int N, int ns) {
    if (!ptt_rows_gemm_supported(rows, K, N, K, N) || (ns != 0 && ns != 16 && ns != 32 && ns != 64) || (ns > 0 && rows % ns)) return 0;
    const RowsGemmGeom g = rows_gemm_geom(rows, K, N);
    if (g.WR == 1 && g.RT == 4) return 0;            // the 128-row x 64-channel tile has no registers left for a second operand stream
    return (long long)rows * K < (1LL << 29) ? 1 : 0;
}

// The input gradient of a layer whose OWN BatchNorm + ReLU backward is formed while its rows are staged (in), with the backward
// sums of the layer below out of the epilogue (as ptt_rows_gemm_bnbwd_f32): g_below = dz @ W^T, dz never read from memory.
extern "C" int ptt_rows_gemm_bnbwd_fused_f32(const ptt_bn_bwd_input* in, int rows, int K, const float* Wpacked, int N, const float* Z,
                                             int ldz, const float* mean, const float* invstd, const float* act_scale,
                                             const float* act_shift, float* out, int ldo, double* sums_partial, size_t partial_elems,
                                             ptt_stream_t stream) {
    if (!in || !in->z || !in->g || !in->k1 || !in->c0 || !in->c1 || !in->mean || !in->act_a || !in->act_b || in->ldz < K || in->ldg < K ||
        (in->arg != nullptr) != (in->ns > 0) || (in->dz_out && in->ldd < K))
        return fail(PTT_EINVAL, "ptt_rows_gemm_bnbwd_fused_f32: bad input descriptor");
    if (!Z || !mean || !invstd || !act_scale || !act_shift || !sums_partial || ldz < N)
        return fail(PTT_EINVAL, "ptt_rows_gemm_bnbwd_fused_f32: null pointer or ldz=%d < N=%d", ldz, N);
    if (!ptt_rows_gemm_bnbwd_fused_supported(rows, K, N, in->ns) || (in->ldz & 3) || (in->ldg & 3) || (in->ldd & 3) ||
        (long long)rows * in->ldz >= (1LL << 29) || (long long)rows * in->ldg >= (1LL << 29) || (long long)rows * in->ldd >= (1LL << 29) ||
        (long long)rows * ldz >= (1LL << 29))
        return fail(PTT_EUNSUPPORTED, "ptt_rows_gemm_bnbwd_fused_f32: rows=%d K=%d N=%d ns=%d", rows, K, N, in->ns);
    const uintptr_t al = reinterpret_cast<uintptr_t>(in->z) | reinterpret_cast<uintptr_t>(in->g) | reinterpret_cast<uintptr_t>(in->k1) |
                         reinterpret_cast<uintptr_t>(in->c0) | reinterpret_cast<uintptr_t>(in->c1) | reinterpret_cast<uintptr_t>(in->mean) | reinterpret_cast<uintptr_t>(in->act_a) |
                         reinterpret_cast<uintptr_t>(in->act_b) | reinterpret_cast<uintptr_t>(in->arg) | reinterpret_cast<uintptr_t>(in->dz_out);
    if (al & 15) return fail(PTT_EINVAL, "ptt_rows_gemm_bnbwd_fused_f32: 16-byte aligned rows and constants expected");
    const BnBwdArgs bn{Z, ldz, mean, invstd, act_scale, act_shift};
    // the kernel's X operand: the dense gradient, or (pooled) z itself
    const float* X = in->ns > 0 ? in->z : in->g;
    const int ldx = in->ns > 0 ? in->ldz : in->ldg;
    return rows_gemm_launch(X, rows, K, ldx, nullptr, nullptr, Wpacked, N, nullptr, 0, nullptr, 0, nullptr, 0, out, ldo, sums_partial,
                            partial_elems, stream, &bn, nullptr, in);
}

extern "C" int ptt_rows_gemm_bnbwd_f32(const float* X, int rows, int K, int ldx, const float* Wpacked, int N, const float* Z, int ldz,
                                       const float* mean, const float* invstd, const float* act_scale, const float* act_shift,
                                       float* out, int ldo, double* sums_partial, size_t partial_elems, ptt_stream_t stream) {
    if (!Z || !mean || !invstd || !act_scale || !act_shift || !sums_partial || ldz < N)
        return fail(PTT_EINVAL, "ptt_rows_gemm_bnbwd_f32: null pointer or ldz=%d < N=%d", ldz, N);
    if ((long long)rows * ldz >= (1LL << 29)) return fail(PTT_EUNSUPPORTED, "ptt_rows_gemm_bnbwd_f32: rows * ldz >= 2^29");
    const BnBwdArgs bn{Z, ldz, mean, invstd, act_scale, act_shift};
    return rows_gemm_launch(X, rows, K, ldx, nullptr, nullptr, Wpacked, N, nullptr, 0, nullptr, 0, nullptr, 0, out, ldo, sums_partial,
                            partial_elems, stream, &bn);
}

extern "C" int ptt_rows_gemm_pool_supported(int rows, int K, int N, int ldx, int ns) {
    if (!ptt_rows_gemm_supported(rows, K, N, ldx, N) || ns <= 0 || rows % ns) return 0;
    const RowsGemmGeom g = rows_gemm_geom(rows, K, N);
    return ((g.WR == 1 && g.RT == 2 && g.KC == 128 && (ns == 16 || ns == 32 || ns == 64)) ||
            (g.WR == 1 && g.RT == 4 && g.CT == 1 && g.KC == 64 && ns == 32)) ? 1 : 0;
}

extern "C" int ptt_rows_gemm_pool_f32(const float* X, int rows, int K, int ldx, const float* in_scale, const float* in_shift,
                                      const float* Wpacked, int N, float* out, int ldo, double* stats, size_t stats_elems, int ns,
                                      float* pmax, float* pmin, int32_t* amax, int32_t* amin, ptt_stream_t stream) {
    const PoolArgs pool{pmax, pmin, amax, amin, ns};
    return rows_gemm_launch(X, rows, K, ldx, in_scale, in_shift, Wpacked, N, nullptr, 0, nullptr, 0, nullptr, 0, out, ldo, stats,
                            stats_elems, stream, nullptr, &pool);
}

extern "C" int ptt_rows_gemm_f32(const float* X, int rows, int K, int ldx, const float* in_scale, const float* in_shift,
                                 const float* Wpacked, int N, const float* bias, int relu, const float* residual, int ldr,
                                 float* out, int ldo, double* stats, size_t stats_elems, ptt_stream_t stream) {
    return rows_gemm_launch(X, rows, K, ldx, in_scale, in_shift, Wpacked, N, bias, relu, residual, ldr, nullptr, 0, out, ldo, stats,
                            stats_elems, stream);
}

extern "C" int ptt_rows_gemm_masked_f32(const float* X, int rows, int K, int ldx, const float* Wpacked, int N, const float* mask,
                                        int ldm, float* out, int ldo, double* stats, size_t stats_elems, ptt_stream_t stream) {
    if (!mask || ldm < N) return fail(PTT_EINVAL, "ptt_rows_gemm_masked_f32: mask=%p ldm=%d", (const void*)mask, ldm);
    return rows_gemm_launch(X, rows, K, ldx, nullptr, nullptr, Wpacked, N, nullptr, 0, nullptr, 0, mask, ldm, out, ldo, stats,
                            stats_elems, stream);
}

static int rows_gemm_launch(const float* X, int rows, int K, int ldx, const float* in_scale, const float* in_shift,
                            const float* Wpacked, int N, const float* bias, int relu, const float* residual, int ldr,
                            const float* mask, int ldm, float* out, int ldo, double* stats, size_t stats_elems, ptt_stream_t stream,
                            const BnBwdArgs* bn, const PoolArgs* pool, const ptt_bn_bwd_input* ain, const GroupSumArgs* gs) {
    if (rows < 0 || K <= 0 || N <= 0 || ldx < K || ldo < N || (residual && ldr < N))
        return fail(PTT_EINVAL, "ptt_rows_gemm_f32: rows=%d K=%d N=%d ldx=%d ldo=%d ldr=%d", rows, K, N, ldx, ldo, ldr);
    if (rows == 0) return PTT_OK;
    if (!X || !Wpacked || !out) return fail(PTT_EINVAL, "ptt_rows_gemm_f32: null pointer");
    if (!ptt_rows_gemm_supported(rows, K, N, ldx, ldo))
        return fail(PTT_EUNSUPPORTED, "ptt_rows_gemm_f32: rows=%d K=%d N=%d ldx=%d (needs K %% 64 == 0, N %% 64 == 0, ldx %% 4 == 0, "
                                      "rows * ld < 2^29)", rows, K, N, ldx);
    if ((reinterpret_cast<uintptr_t>(X) & 15) || (in_scale && ((reinterpret_cast<uintptr_t>(in_scale) | reinterpret_cast<uintptr_t>(in_shift)) & 15)))
        return fail(PTT_EINVAL, "ptt_rows_gemm_f32: X / in_scale / in_shift must be 16-byte aligned");
    if ((in_scale == nullptr) != (in_shift == nullptr)) return fail(PTT_EINVAL, "ptt_rows_gemm_f32: in_scale and in_shift go together");
    const RowsGemmGeom g = rows_gemm_geom(rows, K, N);
    if (bn && (!stats || bias || residual || mask || in_scale)) return fail(PTT_EINVAL, "ptt_rows_gemm_bnbwd_f32: plain input gradient only");
    if (pool && (!stats || bias || residual || mask || bn || !in_scale || relu || !pool->pmax || !pool->pmin || !pool->amax || !pool->amin ||
                 rows % pool->ns))
        return fail(PTT_EINVAL, "ptt_rows_gemm_pool_f32: a statistics launch with a deferred-activation input, whole groups of rows");
    if (pool && !((g.WR == 1 && g.RT == 2 && g.KC == 128 && (pool->ns == 16 || pool->ns == 32 || pool->ns == 64)) ||
                  (g.WR == 1 && g.RT == 4 && g.CT == 1 && g.KC == 64 && pool->ns == 32)))
        return fail(PTT_EUNSUPPORTED, "ptt_rows_gemm_pool_f32: K=%d N=%d ns=%d is not an instantiated shape (ptt_rows_gemm_pool_supported)", K, N, pool->ns);
    if (stats && (bias || stats_elems < (size_t)g.chunks * 2 * N))
        return fail(PTT_EINVAL, "ptt_rows_gemm_f32: statistics need bias == NULL and %zu doubles of workspace", (size_t)g.chunks * 2 * N);
    if (mask && (long long)rows * ldm >= (1LL << 29)) return fail(PTT_EUNSUPPORTED, "ptt_rows_gemm_masked_f32: rows * ldm >= 2^29");
    if (residual && (long long)rows * ldr >= (1LL << 29)) return fail(PTT_EUNSUPPORTED, "ptt_rows_gemm_f32: rows * ldr >= 2^29");
    RowsGemmParams p;
    p.bz = bn ? bn->z : nullptr; p.ldbz = bn ? bn->ldz : 0;
    p.bmean = bn ? bn->mean : nullptr; p.binv = bn ? bn->invstd : nullptr; p.ba = bn ? bn->a : nullptr; p.bb = bn ? bn->b : nullptr;
    p.mask = mask; p.ldm = ldm;
    p.X = X; p.Wp = Wpacked; p.bias = bias; p.residual = residual; p.out = out; p.in_a = in_scale; p.in_b = in_shift;
    p.stats = stats; p.rows = rows; p.K = K; p.ldx = ldx; p.N = N; p.NT = N / 32; p.relu = relu; p.ldr = ldr; p.ldo = ldo;
    p.ntiles = g.ntiles; p.G = g.G; p.ncg = g.ncg; p.nchunks = K / g.KC;
    p.pmax = pool ? pool->pmax : nullptr; p.pmin = pool ? pool->pmin : nullptr; p.amax = pool ? pool->amax : nullptr; p.amin = pool ? pool->amin : nullptr;
    p.tz = nullptr; p.ldtz = 0; p.tk1 = p.tc0 = p.tc1 = p.tmu = nullptr; p.tg = nullptr; p.ldtg = 0; p.targ = nullptr; p.a_out = nullptr; p.lda_out = 0;
    if (ain) {
        if (!bn) return fail(PTT_EINVAL, "ptt_rows_gemm_f32: the fused BatchNorm-backward operand comes with the backward-sums epilogue");
        p.tz = ain->z; p.ldtz = ain->ldz; p.tk1 = ain->k1; p.tc0 = ain->c0; p.tc1 = ain->c1; p.tmu = ain->mean; p.in_a = ain->act_a; p.in_b = ain->act_b;
        p.tg = ain->g; p.ldtg = ain->ldg; p.targ = ain->arg; p.a_out = ain->dz_out; p.lda_out = ain->ldd;
    }
    p.gsum = gs ? gs->gsum : nullptr; p.plain = gs ? gs->plain : nullptr; p.ldgs = gs ? gs->ldgs : 0; p.ldpl = gs ? gs->ldpl : 0;
    const int lds = 2 * g.TR * (g.KC + 4) * (int)sizeof(float);
    const dim3 grid(g.G * g.ncg);
    hipStream_t s = as_stream(stream);
    int rc = PTT_OK;
    if (gs) {
        if (!residual || bias || relu || mask || stats || bn || pool || ain || in_scale)
            return fail(PTT_EINVAL, "ptt_rows_gemm_rsum16_f32: a plain GEMM with a residual");
#define PTT_RG_GS(CT_)                                                                                                  \
        if (g.WR == 1 && g.RT == 2 && g.CT == CT_ && g.KC == 128) {                                                     \
            if ((rc = set_lds_limit(reinterpret_cast<const void*>(rows_gemm_kernel<1, 2, CT_, 128, false, false, 0, false, 0, 0, 16>), lds))) return rc; \
            hipLaunchKernelGGL((rows_gemm_kernel<1, 2, CT_, 128, false, false, 0, false, 0, 0, 16>), grid, dim3(256), lds, s, p); \
        }
        PTT_RG_GS(2) PTT_RG_GS(1)
#undef PTT_RG_GS
        return check_launch("rows_gemm_kernel(rsum16)");
    }
#ifdef PTT_GEMM_DEV
    if (const char* e = getenv("PTT_RG_EXP")) {
        const int x = atoi(e);
        if (g.WR == 1 && g.RT == 2 && g.CT == 2 && g.KC == 128 && x && !in_scale) {
#define PTT_RG_EXP_CASE(X)                                                                                              \
            if (x == X) {                                                                                               \
                if ((rc = set_lds_limit(reinterpret_cast<const void*>(rows_gemm_kernel<1, 2, 2, 128, false, false, X>), lds))) return rc; \
                hipLaunchKernelGGL((rows_gemm_kernel<1, 2, 2, 128, false, false, X>), grid, dim3(256), lds, s, p);      \
                return check_launch("rows_gemm_kernel(exp)");                                                           \
            }
            PTT_RG_EXP_CASE(1) PTT_RG_EXP_CASE(2) PTT_RG_EXP_CASE(3) PTT_RG_EXP_CASE(4) PTT_RG_EXP_CASE(7) PTT_RG_EXP_CASE(8) PTT_RG_EXP_CASE(15)
#undef PTT_RG_EXP_CASE
        }
    }
#endif
    if (pool) {
#define PTT_RG_POOL(WR_, RT_, CT_, KC_, NS_)                                                                            \
        if (g.WR == WR_ && g.RT == RT_ && g.CT == CT_ && g.KC == KC_ && pool->ns == NS_) {                              \
            if ((rc = set_lds_limit(reinterpret_cast<const void*>(rows_gemm_kernel<WR_, RT_, CT_, KC_, true, true, 0, false, NS_>), lds))) return rc; \
            hipLaunchKernelGGL((rows_gemm_kernel<WR_, RT_, CT_, KC_, true, true, 0, false, NS_>), grid, dim3(256), lds, s, p); \
        }
        PTT_RG_POOL(1, 2, 2, 128, 16) PTT_RG_POOL(1, 2, 2, 128, 32) PTT_RG_POOL(1, 2, 2, 128, 64) PTT_RG_POOL(1, 4, 1, 64, 32)
        PTT_RG_POOL(1, 2, 1, 128, 16) PTT_RG_POOL(1, 2, 1, 128, 32) PTT_RG_POOL(1, 2, 1, 128, 64)      // launches too small for 256 columns per workgroup
#undef PTT_RG_POOL
        return check_launch("rows_gemm_kernel(pool)");
    }
#define PTT_RG_LAUNCH(WR_, RT_, CT_, KC_, ST_, AC_)                                                                     \
    {                                                                                                                   \
        if ((rc = set_lds_limit(reinterpret_cast<const void*>(rows_gemm_kernel<WR_, RT_, CT_, KC_, ST_, AC_>), lds))) return rc; \
        hipLaunchKernelGGL((rows_gemm_kernel<WR_, RT_, CT_, KC_, ST_, AC_>), grid, dim3(256), lds, s, p);               \
    }
#define PTT_RG_AIN(WR_, RT_, CT_, KC_, AIN_)                                                                            \
    {                                                                                                                   \
        if ((rc = set_lds_limit(reinterpret_cast<const void*>(rows_gemm_kernel<WR_, RT_, CT_, KC_, true, false, 0, true, 0, AIN_>), lds))) return rc; \
        hipLaunchKernelGGL((rows_gemm_kernel<WR_, RT_, CT_, KC_, true, false, 0, true, 0, AIN_>), grid, dim3(256), lds, s, p); \
    }
#define PTT_RG_AINS(WR_, RT_, CT_, KC_)                                                                                 \
        if (bn && ain && ain->ns == 16) PTT_RG_AIN(WR_, RT_, CT_, KC_, 16)                                              \
        else if (bn && ain && ain->ns == 32) PTT_RG_AIN(WR_, RT_, CT_, KC_, 32)                                         \
        else if (bn && ain && ain->ns == 64) PTT_RG_AIN(WR_, RT_, CT_, KC_, 64)                                         \
        else if (bn && ain) PTT_RG_AIN(WR_, RT_, CT_, KC_, 2)                                                           \
        else
#define PTT_RG_NO_AINS(WR_, RT_, CT_, KC_)
#define PTT_RG_CASE(WR_, RT_, CT_, KC_, AINS_)                                                                          \
    if (g.WR == WR_ && g.RT == RT_ && g.CT == CT_ && g.KC == KC_) {                                                     \
        AINS_(WR_, RT_, CT_, KC_)                                                                                       \
        if (bn) {                                                                                                  \
            if ((rc = set_lds_limit(reinterpret_cast<const void*>(rows_gemm_kernel<WR_, RT_, CT_, KC_, true, false, 0, true>), lds))) return rc; \
            hipLaunchKernelGGL((rows_gemm_kernel<WR_, RT_, CT_, KC_, true, false, 0, true>), grid, dim3(256), lds, s, p); \
        } else if (stats && in_scale) PTT_RG_LAUNCH(WR_, RT_, CT_, KC_, true, true)                                     \
        else if (stats) PTT_RG_LAUNCH(WR_, RT_, CT_, KC_, true, false)                                                  \
        else if (in_scale) PTT_RG_LAUNCH(WR_, RT_, CT_, KC_, false, true)                                               \
        else PTT_RG_LAUNCH(WR_, RT_, CT_, KC_, false, false)                                                            \
    }
    PTT_RG_CASE(1, 2, 2, 128, PTT_RG_AINS) PTT_RG_CASE(1, 2, 1, 128, PTT_RG_AINS) PTT_RG_CASE(1, 4, 1, 64, PTT_RG_NO_AINS)
    PTT_RG_CASE(2, 2, 1, 64, PTT_RG_AINS)
#undef PTT_RG_CASE
#undef PTT_RG_AINS
#undef PTT_RG_NO_AINS
#undef PTT_RG_AIN
#undef PTT_RG_LAUNCH
    return check_launch("rows_gemm_kernel");
}

// ---- weight gradient, large blocks ----
namespace ptt {
#ifndef PTT_WG2_MIN_ROWS
#define PTT_WG2_MIN_ROWS 768      // rows per chunk at least
#endif
struct Wgrad2Geom { int TN, TK, BN, BK, nbo, nbk, chunk_rows, nchunks; bool ok; };
static Wgrad2Geom wgrad2_geom(int R, int Cout, int Cin) {
    Wgrad2Geom g{};
    g.ok = false;
    if (R < 2048 || Cout % 128 || Cin % 128) return g;
    // the largest block that still fills the chip with row chunks of >= 768 rows (one workgroup per CU, one round)
    const int ncu = cu_count();
    // block shapes, largest first. (Round 5, scripts/wgrad_bench.py: in isolation the 128 x 256 block is 5-8 % faster than 256 x 256
    // at >= 98304 rows — 107 against 99 TFLOP/s — but inside the training step, where the launch follows a GEMM that leaves its
    // operands in the Infinity Cache, preferring it cost 0.24 ms per step: 20.42 against 20.18 ms.)
    const int cand[4][2] = {{4, 2}, {4, 1}, {2, 2}, {2, 1}};
    int first = 0;
#ifdef PTT_GEMM_DEV
    if (const char* e = getenv("PTT_WG2_FIRST")) first = atoi(e);       // dev: skip the larger blocks (tile-shape experiments)
#endif
    for (int k = first; k < 4; ++k) {
        const int TN = cand[k][0], TK = cand[k][1];
        if (Cout % (64 * TN) || Cin % (128 * TK)) continue;
        const int blocks = (Cout / (64 * TN)) * (Cin / (128 * TK));
        if (blocks > ncu) continue;
        int nch = ncu / blocks;                                 // floor: never a second round of workgroups
        int rows = (R + nch - 1) / nch;
        if (rows < PTT_WG2_MIN_ROWS) rows = PTT_WG2_MIN_ROWS;
        rows = (rows + 31) & ~31;
        const int nchunks = (R + rows - 1) / rows;
        if (nchunks * blocks * 4 < ncu * 3 && k < 3) continue;  // under 3/4 of the CUs busy: try a smaller block
        if (nchunks * blocks * 2 < ncu) break;                  // still under half: the 128 x 128-block kernel of round 2
        g.TN = TN; g.TK = TK; g.BN = 64 * TN; g.BK = 128 * TK; g.nbo = Cout / g.BN; g.nbk = Cin / g.BK;
        g.chunk_rows = rows; g.nchunks = nchunks; g.ok = true;
        return g;
    }
    return g;
}
}  // namespace ptt

extern "C" size_t ptt_linear_wgrad2_workspace(int R, int Cout, int Cin) {
    const Wgrad2Geom g = wgrad2_geom(R, Cout, Cin);
    return g.ok ? (size_t)g.nchunks * Cout * Cin * sizeof(float) : 0;
}

// dW == nullptr: the row-chunk partials [nchunks][Cout * Cin] stay in the workspace (ptt_linear_wgrad2_partials_f32)
static int linear_wgrad2_run(const float* dZ, int ldz, const float* X, int ldx, int R, int Cout, int Cin, float* dW,
                             int accumulate, void* ws, size_t ws_bytes, const float* x_scale, const float* x_shift,
                             int* nchunks_out, ptt_stream_t stream) {
    const Wgrad2Geom g = wgrad2_geom(R, Cout, Cin);
    if (!g.ok || (ldz & 3) || (ldx & 3) || ((reinterpret_cast<uintptr_t>(dZ) | reinterpret_cast<uintptr_t>(X)) & 15) ||
        (long long)R * ldz >= (1LL << 29) || (long long)R * ldx >= (1LL << 29))
        return fail(PTT_EUNSUPPORTED, "ptt_linear_wgrad2_f32: R=%d Cout=%d Cin=%d ldz=%d ldx=%d (needs R >= 2048, Cout %% 128 == 0, "
                                      "Cin %% 128 == 0, 16-byte aligned rows)", R, Cout, Cin, ldz, ldx);
    if (ldz < Cout || ldx < Cin || !dZ || !X || (!dW && !nchunks_out)) return fail(PTT_EINVAL, "ptt_linear_wgrad2_f32: bad argument");
    if (x_scale && (!x_shift || ((reinterpret_cast<uintptr_t>(x_scale) | reinterpret_cast<uintptr_t>(x_shift)) & 15)))
        return fail(PTT_EINVAL, "ptt_linear_wgrad2_f32: the input transform needs 16-byte aligned scale / shift");
    if (!ws || ws_bytes < ptt_linear_wgrad2_workspace(R, Cout, Cin)) return fail(PTT_EWORKSPACE, "ptt_linear_wgrad2_f32: workspace too small");
    hipStream_t s = as_stream(stream);
    const int lds = 2 * 32 * (g.BN + 4 + g.BK + 4) * (int)sizeof(float);
    const dim3 grid(g.nbo * g.nbk * g.nchunks);
    int rc = PTT_OK;
#define PTT_WG2_CASE(TN_, TK_)                                                                                          \
    if (g.TN == TN_ && g.TK == TK_) {                                                                                   \
        if ((rc = set_lds_limit(reinterpret_cast<const void*>(wgrad2_kernel<TN_, TK_>), lds))) return rc;               \
        hipLaunchKernelGGL((wgrad2_kernel<TN_, TK_>), grid, dim3(512), lds, s, dZ, ldz, X, ldx, R, Cout, Cin, g.nbk, g.chunk_rows, \
                           static_cast<float*>(ws), x_scale, x_shift);                                                  \
    }
#ifdef PTT_GEMM_DEV
    if (const char* e = getenv("PTT_WG2_EXP")) {
        const int x = atoi(e);
#define PTT_WG2_EXP_CASE(X)                                                                                             \
        if (x == X && g.TN == 4 && g.TK == 2) {                                                                         \
            if ((rc = set_lds_limit(reinterpret_cast<const void*>(wgrad2_kernel<4, 2, X>), lds))) return rc;            \
            hipLaunchKernelGGL((wgrad2_kernel<4, 2, X>), grid, dim3(512), lds, s, dZ, ldz, X_, ldx, R, Cout, Cin, g.nbk, g.chunk_rows, \
                               static_cast<float*>(ws), x_scale, x_shift);                                              \
            return check_launch("wgrad2_kernel(exp)");                                                                  \
        }
        const float* X_ = X;
        PTT_WG2_EXP_CASE(1) PTT_WG2_EXP_CASE(2) PTT_WG2_EXP_CASE(3) PTT_WG2_EXP_CASE(4) PTT_WG2_EXP_CASE(7) PTT_WG2_EXP_CASE(8)
        PTT_WG2_EXP_CASE(15) PTT_WG2_EXP_CASE(16) PTT_WG2_EXP_CASE(31)
#undef PTT_WG2_EXP_CASE
    }
#endif
    PTT_WG2_CASE(4, 2) PTT_WG2_CASE(4, 1) PTT_WG2_CASE(2, 2) PTT_WG2_CASE(2, 1)
#undef PTT_WG2_CASE
    if (dW) launch_wgrad_finish(static_cast<const float*>(ws), g.nchunks, (size_t)Cout * Cin, accumulate, dW, s);
    if (nchunks_out) *nchunks_out = g.nchunks;
    return check_launch("wgrad2_kernel");
}
extern "C" int ptt_linear_wgrad2_f32(const float* dZ, int ldz, const float* X, int ldx, int R, int Cout, int Cin, float* dW,
                                     int accumulate, void* ws, size_t ws_bytes, const float* x_scale, const float* x_shift,
                                     ptt_stream_t stream) {
    if (!dW) return fail(PTT_EINVAL, "ptt_linear_wgrad2_f32: bad argument");
    return linear_wgrad2_run(dZ, ldz, X, ldx, R, Cout, Cin, dW, accumulate, ws, ws_bytes, x_scale, x_shift, nullptr, stream);
}
extern "C" int ptt_linear_wgrad2_partials_f32(const float* dZ, int ldz, const float* X, int ldx, int R, int Cout, int Cin, void* ws,
                                              size_t ws_bytes, const float* x_scale, const float* x_shift, int* nchunks,
                                              ptt_stream_t stream) {
    if (!nchunks) return fail(PTT_EINVAL, "ptt_linear_wgrad2_partials_f32: null pointer");
    return linear_wgrad2_run(dZ, ldz, X, ldx, R, Cout, Cin, nullptr, 0, ws, ws_bytes, x_scale, x_shift, nchunks, stream);
}
