// fp32-MFMA kernels of the hot path (gfx950): weight packing, row-wise linear layers, the
// fused set-abstraction level (group -> normalise -> SharedMLP -> max-pool) and the fused
// per-(point,neighbour) part of the Point-Transformer block.
//
// Common structure
//   * one workgroup = 4 waves (256 threads), two workgroups per CU (<= 80 KB LDS each), so
//     the MFMA pipe of a SIMD always has a second wave to run while the first one gathers,
//     writes an epilogue or waits at a barrier;
//   * the activation tile X[rows][K] lives in LDS (row stride ldk == 4 mod 8 floats, which
//     makes the ds_read_b128 A-fragment reads bank-conflict free) and is overwritten in
//     place by each layer's output — grouped / per-pair tensors never reach HBM;
//   * weights are pre-packed once into MFMA B-fragment order [K/8][Cout/32][lane][4], so a
//     wave fetches a fragment with ONE coalesced 1 KiB global_load_dwordx4 straight from
//     L2 (weights are <= 1 MiB per layer and shared by every workgroup);
//   * v_mfma_f32_32x32x2_f32: exact fp32 (bitwise an fmaf chain), 64 cycles per issue.
//     A K-block of 8 input channels is 4 MFMAs; lane half h (lane>>5) feeds channels
//     8*kb + 4*h + j to MFMA j, so each lane's 4 operands are one 16-byte load.
//   * C/D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
//     a tile's 32 rows are the neighbours of one centre (or of two, 16 each), so max-pool
//     and the softmax over neighbours are register-local plus one lane^32 exchange.
// Two kernels leave that common structure (round 2): sa_stream_kernel keeps everything that is not a GEMM inside the
// waves' own MFMA streams (persistent workgroups, two LDS tiles), and sa_lds_kernel chains its layers through the
// accumulator layout itself (transposed products, activations never leave the registers). DESIGN.md lessons 8, 12-14.
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "common.h"
#include "mfma_common.h"

#ifndef PTT_PAIR_PF
#define PTT_PAIR_PF 1
#endif
#ifndef PTT_GATHER_BATCH     // feature rows a wave keeps in flight while grouping
#define PTT_GATHER_BATCH 16
#endif
#ifndef PTT_SA_WAVES        // waves per SIMD the SA chain kernel is register-budgeted for
#define PTT_SA_WAVES 2
#endif
// GEMM loop flavours (gemm_core's PF): 0 = two register sets pinned with sched_barrier (best for the SA
// chains and the linear kernel, measured), 1 = one-block prefetch scheduled by hipcc (best for the pair
// kernel), 2 = two blocks in flight (slower everywhere: the L2->CU path saturates).

namespace ptt {

// Per-phase cycle stamps of the chained kernels (scripts/sa_phases.py, scripts/pair_phases.py): only in -DPTT_DEV
// builds; a release kernel has neither the pointer test nor the store.
#ifdef PTT_DEV
#define PTT_STAMP(i) do { if (p.dbg && threadIdx.x == 0 && blockIdx.x < 4096) p.dbg[blockIdx.x * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define PTT_STAMP(i) do { } while (0)
#endif

// Two workgroups share each CU (and each SIMD's MFMA pipe). Launched together with identical work
// they run in lockstep, so their non-MFMA phases (gather, epilogue, softmax) coincide and the matrix
// pipe idles. The workgroup that got the SECOND LDS allocation of its CU (HW_REG_LDS_ALLOC.LDS_BASE
// != 0) in the FIRST wave of workgroups therefore starts `quanta` x ~8k cycles late; later workgroups
// inherit the offset from the slot they replace. Placement only changes speed, never results.
__device__ __forceinline__ void stagger_second_slot(int first_wave_blocks, int quanta) {
    if ((int)blockIdx.x < first_wave_blocks && quanta > 0) {
        const unsigned alloc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 6);   // HW_REG_LDS_ALLOC
        const unsigned lds_base = alloc & 0xffu, lds_size = (alloc >> 12) & 0x1ffu;
        int slot = lds_size ? (int)(lds_base / lds_size) : 0;                                // 0,1,2,.. on this CU
        if (lds_base != 0 && slot == 0) slot = 1;                                            // field granularities differ
        for (int i = 0; i < slot * quanta; ++i) __builtin_amdgcn_s_sleep(127);
    }
}

// XCD-aware workgroup numbering (speed only, never correctness). The dispatcher is observed to place block b
// on XCD b % 8, and each XCD has its own 4 MiB L2: with the plain numbering the workgroups of one frame are dealt
// round-robin over all 8 XCDs, so every XCD pulls every frame's feature / q|k|v rows through its own L2. Renumbered,
// the blocks an XCD receives are CONSECUTIVE logical workgroups = whole frames, whose rows (each used by ~16
// neighbourhoods) hit in that XCD's L2. Measured on the pair kernel (profiles/README.md): fabric traffic per
// launch 443 -> 88 MB (55 MB compulsory), L2 hit rate 95.6 -> 99.1 %, kernel time -2 %.
// -DPTT_XCD_REMAP=0 restores the plain numbering.
#ifndef PTT_LINEAR_PF
#define PTT_LINEAR_PF 0      // the hand-pinned two-register-set K loop: 5 % faster than the rotating prefetch on the linear
#endif                      // kernel's short launches (0.272 -> 0.259 ms per step; the pair kernel prefers form 1)
#ifndef PTT_XCD_REMAP
#define PTT_XCD_REMAP 1
#endif
__device__ __forceinline__ int logical_block() {
#if PTT_XCD_REMAP
    const int bid = blockIdx.x, per = (int)gridDim.x >> 3;
    if (bid >= (per << 3)) return bid;              // the last grid % 8 blocks keep their number
    return (bid & 7) * per + (bid >> 3);
#else
    return blockIdx.x;
#endif
}

// ------------------------------------------------------------------------------------------
// weight packing
// ------------------------------------------------------------------------------------------
// W element (output col, input k) of batch b lives at W[b * sb + col * so + k * sk]: a plain (Cout,K) weight is
// so = K, sk = 1; a weight given TRANSPOSED in memory ((K,Cout) rows, e.g. the values of an attention block used as the
// B operand of P.V) is so = 1, sk = row stride. blockIdx.y = batch.
__global__ void pack_weight_kernel(const float* __restrict__ W, int Cout, int K, int NT, int rot, size_t total,
                                   float* __restrict__ P, long long so, long long sk, long long sb) {
    const float* Wb = W + (size_t)blockIdx.y * sb;
    float* Pb = P + (size_t)blockIdx.y * total;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(e & 3);
        const int lane = (int)((e >> 2) & 63);
        const size_t tile = e >> 8;
        const int ct = (int)(tile % NT);
        const int kb = (int)(tile / NT);
        const int col = ct * 32 + (lane & 31);
        const int k = kb * 8 + 4 * (lane >> 5) + j;
        // packed position k holds original input channel (k + rot) mod K
        Pb[e] = (col < Cout && k < K) ? Wb[(size_t)col * so + (size_t)((k + rot) % K) * sk] : 0.0f;
    }
}

// Many weights in one launch (a training step re-packs all of them after every optimiser update): blockIdx.y = job.
__global__ __launch_bounds__(256) void pack_weights_kernel(const ptt_pack_job* __restrict__ jobs, float* __restrict__ arena) {
    const ptt_pack_job jb = jobs[blockIdx.y];
    const int NT = (jb.Cout + 31) / 32;
    const size_t total = (size_t)NT * (size_t)((jb.K + 7) / 8) * 256;
    float* Pb = arena + jb.out_offset;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(e & 3);
        const int lane = (int)((e >> 2) & 63);
        const size_t tile = e >> 8;
        const int ct = (int)(tile % NT);
        const int kb = (int)(tile / NT);
        const int col = ct * 32 + (lane & 31);
        const int k = kb * 8 + 4 * (lane >> 5) + j;
        Pb[e] = (col < jb.Cout && k < jb.K) ? jb.W[(long long)col * jb.stride_out + (long long)k * jb.stride_k] : 0.0f;
    }
}

// One K-block: 4 * RT * CT MFMAs on the first CT column tiles of acc (ACT >= CT columns wide).
// Weight fragments are fetched with raw buffer loads: address = descriptor base (SGPRs) + per-lane byte offset (one
// VGPR, constant for the whole GEMM) + wave-uniform byte offset (SGPR, advanced by the scalar unit). The flat
// global_load form made hipcc recompute a 64-bit VGPR address per fragment (v_add_co / v_addc pairs): ~10 vector-ALU
// instructions per K-block that steal issue time from the fp32 MFMAs sharing the SIMD (DESIGN.md lesson 8).
// -DPTT_BUFFER_WEIGHTS=0 restores the flat loads.
#ifndef PTT_BUFFER_WEIGHTS
#define PTT_BUFFER_WEIGHTS 1
#endif
template <int RT, int CT, int ACT>
__device__ __forceinline__ void gemm_mfma_block(const f32x4 (&a)[RT], const f32x4 (&b)[CT], f32x16 (&acc)[RT][ACT]) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int u = 0; u < CT; ++u)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
                acc[rt][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[rt][j], b[u][j], acc[rt][u], 0, 0, 0);
}

// ------------------------------------------------------------------------------------------
// The shared GEMM core: acc[rt][u] += X[rt-th 32 rows][0:8*nkb] * W[:, column tile ct0 + 4*u], u < CT.
// A from LDS (one ds_read_b128 per row tile per K-block), B from the packed global weights with a
// one-block register prefetch that hipcc unrolls and interleaves with counted vmcnt waits.
// CT is a COMPILE-TIME tile count on purpose: a run-time "how many of my column tiles exist" test
// puts the loads under control flow, and the waitcnt pass then falls back to vmcnt(0) before every
// MFMA group, which drains the prefetch (measured 2-4x slower K-loops).
// (A hand-pinned two-register-set pipeline with sched_barrier(0) was measured 4-6 % SLOWER than this
//  form: the loop is bound by the L2->CU fetch path, not by load placement — see DESIGN.md.)
template <int RT, int CT, int ACT, int CTS = 4, int PF = 1>   // CTS: distance (in column tiles) between this wave's tiles
__device__ __forceinline__ void gemm_core(const float* Xs, int ldk, int nkb, const f32x4* __restrict__ Wp, int NT,
                                          int ct0, int lane, f32x16 (&acc)[RT][ACT], const f32x4* pre = nullptr) {
    // `pre` (PF 0 and 1): the first K-block's CT weight fragments, already requested by the caller — issued
    // before the previous layer's epilogue and barriers so that a layer does not start with an exposed L2 round trip
    static_assert(CT <= ACT, "accumulator array too narrow");
    const int row = lane & 31, half = lane >> 5;
    const float* arow = Xs + row * ldk + 4 * half;
    const f32x4* bp = Wp + (size_t)ct0 * 64 + lane;
    const size_t bstep = (size_t)NT * 64;
#if PTT_BUFFER_WEIGHTS
    const __amdgpu_buffer_rsrc_t wr = weight_rsrc(Wp);
    const int wvoff = (ct0 * 64 + lane) * 16;             // this lane's byte offset inside a K-block row of fragments
    const int wkstep = NT * 1024;                          // bytes per K-block
#define PTT_WFRAG(KB, U) weight_load(wr, wvoff, (KB) * wkstep + (U) * CTS * 1024)
#else
#define PTT_WFRAG(KB, U) (bp[(size_t)(KB) * bstep + (size_t)(U) * CTS * 64])
#endif

    if constexpr (PF == 0) {                     // two register sets, order pinned with sched_barrier
        f32x4 a0[RT], a1[RT], b0[CT], b1[CT];
#define PTT_LOAD_BLOCK(A, Bv, KB)                                                             \
        {                                                                                     \
            _Pragma("unroll") for (int u = 0; u < CT; ++u) Bv[u] = PTT_WFRAG(KB, u);          \
            _Pragma("unroll") for (int rt = 0; rt < RT; ++rt)                                 \
                A[rt] = *reinterpret_cast<const f32x4*>(arow + rt * 32 * ldk + (KB) * 8);     \
        }
        if (pre) {
#pragma unroll
            for (int u = 0; u < CT; ++u) b0[u] = pre[u];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) a0[rt] = *reinterpret_cast<const f32x4*>(arow + rt * 32 * ldk);
        } else {
            PTT_LOAD_BLOCK(a0, b0, 0)
        }
        int kb = 0;
#pragma unroll 1
        for (; kb + 2 < nkb; kb += 2) {
            PTT_LOAD_BLOCK(a1, b1, kb + 1)
            __builtin_amdgcn_sched_barrier(0);
            gemm_mfma_block<RT, CT, ACT>(a0, b0, acc);
            __builtin_amdgcn_sched_barrier(0);
            PTT_LOAD_BLOCK(a0, b0, kb + 2)
            __builtin_amdgcn_sched_barrier(0);
            gemm_mfma_block<RT, CT, ACT>(a1, b1, acc);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (kb + 1 < nkb) {
            PTT_LOAD_BLOCK(a1, b1, kb + 1)
            __builtin_amdgcn_sched_barrier(0);
            gemm_mfma_block<RT, CT, ACT>(a0, b0, acc);
            __builtin_amdgcn_sched_barrier(0);
            gemm_mfma_block<RT, CT, ACT>(a1, b1, acc);
        } else {
            gemm_mfma_block<RT, CT, ACT>(a0, b0, acc);
        }
#undef PTT_LOAD_BLOCK
    } else if constexpr (PF == 1) {
        f32x4 bcur[CT], bnxt[CT];
#pragma unroll
        for (int u = 0; u < CT; ++u) { bcur[u] = pre ? pre[u] : PTT_WFRAG(0, u); bnxt[u] = bcur[u]; }
        // the last block is peeled so that the loop body has NO conditional load: with a load under control
        // flow (run-time nkb) the waitcnt pass gives up counting and waits vmcnt(0) before every MFMA group
        for (int kb = 0; kb + 1 < nkb; ++kb) {
#pragma unroll
            for (int u = 0; u < CT; ++u) bnxt[u] = PTT_WFRAG(kb + 1, u);
            f32x4 a[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) a[rt] = *reinterpret_cast<const f32x4*>(arow + rt * 32 * ldk + kb * 8);
            gemm_mfma_block<RT, CT, ACT>(a, bcur, acc);
#pragma unroll
            for (int u = 0; u < CT; ++u) bcur[u] = bnxt[u];
        }
        {
            f32x4 a[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) a[rt] = *reinterpret_cast<const f32x4*>(arow + rt * 32 * ldk + (nkb - 1) * 8);
            gemm_mfma_block<RT, CT, ACT>(a, bcur, acc);
        }
    } else {                                     // two K-blocks of weights in flight (needs nkb % 3 == 1 handling below)
        f32x4 b0[CT], b1[CT], b2[CT];
#pragma unroll
        for (int u = 0; u < CT; ++u) {
            b0[u] = bp[(size_t)u * CTS * 64];
            b1[u] = (nkb > 1) ? bp[bstep + (size_t)u * CTS * 64] : b0[u];
            b2[u] = b0[u];
        }
        int kb = 0;
        for (; kb + 3 <= nkb; kb += 3) {
#define PTT_STAGE(CUR, FILL, OFF)                                                                         \
            {                                                                                             \
                if (kb + (OFF) + 2 < nkb) {                                                               \
                    const f32x4* bn = bp + (size_t)(kb + (OFF) + 2) * bstep;                              \
                    _Pragma("unroll") for (int u = 0; u < CT; ++u) FILL[u] = bn[(size_t)u * CTS * 64];    \
                }                                                                                         \
                f32x4 a[RT];                                                                              \
                _Pragma("unroll") for (int rt = 0; rt < RT; ++rt)                                         \
                    a[rt] = *reinterpret_cast<const f32x4*>(arow + rt * 32 * ldk + (kb + (OFF)) * 8);     \
                gemm_mfma_block<RT, CT, ACT>(a, CUR, acc);                                                \
            }
            PTT_STAGE(b0, b2, 0)
            PTT_STAGE(b1, b0, 1)
            PTT_STAGE(b2, b1, 2)
        }
        // tail (nkb % 3): blocks kb, kb+1 are already in b0, b1
        if (kb < nkb) {
            f32x4 a[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) a[rt] = *reinterpret_cast<const f32x4*>(arow + rt * 32 * ldk + kb * 8);
            gemm_mfma_block<RT, CT, ACT>(a, b0, acc);
            if (kb + 1 < nkb) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) a[rt] = *reinterpret_cast<const f32x4*>(arow + rt * 32 * ldk + (kb + 1) * 8);
                gemm_mfma_block<RT, CT, ACT>(a, b1, acc);
            }
        }
#undef PTT_STAGE
    }
#undef PTT_WFRAG
}

// Run the core on however many of this wave's (up to CT) column tiles exist: a wave-uniform
// dispatch to compile-time tile counts.
template <int RT, int CT, int PFM>
__device__ __forceinline__ void gemm_tiles(const float* Xs, int ldk, int nkb, const f32x4* __restrict__ Wp, int NT,
                                           int ct0, int nvalid, int lane, f32x16 (&acc)[RT][CT],
                                           const f32x4* pre = nullptr) {
    if (nvalid >= CT) gemm_core<RT, CT, CT, 4, PFM>(Xs, ldk, nkb, Wp, NT, ct0, lane, acc, pre);
    else if constexpr (CT > 1) {
        if (nvalid == 1) gemm_core<RT, 1, CT, 4, PFM>(Xs, ldk, nkb, Wp, NT, ct0, lane, acc, pre);
        else if constexpr (CT > 2) {
            if (nvalid == 2) gemm_core<RT, 2, CT, 4, PFM>(Xs, ldk, nkb, Wp, NT, ct0, lane, acc, pre);
            else if (nvalid == 3) gemm_core<RT, 3, CT, 4, PFM>(Xs, ldk, nkb, Wp, NT, ct0, lane, acc, pre);
        }
    }
}

// Workgroup barrier that orders LDS traffic only: unlike __syncthreads() it does not wait for outstanding
// global loads (vmcnt), so weight fragments requested for the NEXT layer stay in flight across it.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// First K-block of a GEMM's weights for this wave's CT column tiles w, w+4, ... (all exist).
template <int CT>
__device__ __forceinline__ void prefetch_first_block_full(const float* Wp, int w, int lane, f32x4 (&pre)[CT]) {
    const f32x4* bp = reinterpret_cast<const f32x4*>(Wp) + (size_t)w * 64 + lane;
#pragma unroll
    for (int u = 0; u < CT; ++u) pre[u] = bp[(size_t)u * 4 * 64];
}

// What a wave requests ahead of a layer of the chained kernels: the first K-block of the weights of its (up to 2)
// column tiles w, w+4, and — when the layer has no separate scale (BatchNorm folded into the packed weights) — the
// per-column shift, which then INITIALISES the accumulators instead of being added in the epilogue.
struct SaPre { f32x4 w[2]; float sh[2]; };
__device__ __forceinline__ void prefetch_first_block(const float* Wp, const float* scale, const float* shift, int NT, int w,
                                                     int lane, SaPre& pre) {
    const f32x4* bp = reinterpret_cast<const f32x4*>(Wp) + (size_t)w * 64 + lane;
    pre.sh[0] = pre.sh[1] = 0.f;
    if (w < NT) pre.w[0] = bp[0];
    if (w + 4 < NT) pre.w[1] = bp[4 * 64];
    if (shift && !scale) {
        // buffer loads (descriptor in SGPRs + one 32-bit lane offset): a flat load's loop-invariant 64-bit address was
        // hoisted into a VGPR pair per tile and spilled in the kernels that sit at the 256-register bound
        const __amdgpu_buffer_rsrc_t rs = weight_rsrc(shift);
        const int voff = (w * 32 + (lane & 31)) * (int)sizeof(float);
        if (w < NT) pre.sh[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, 0, 0));
        if (w + 4 < NT) pre.sh[1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, 4 * 32 * (int)sizeof(float), 0));
    }
}

template <int RT, int CT>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[RT][CT]) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int u = 0; u < CT; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rt][u][r] = 0.f;
}

// ------------------------------------------------------------------------------------------
// Linear: out[rows, Cout] = act(X[rows, K] @ W^T * scale + shift) (+ residual)
// A workgroup owns RT*32 rows x 256 columns (wave w: column tiles w and w+4 of its group, all
// row tiles, so every weight fragment feeds RT MFMAs). X is staged through two LDS buffers in
// K-chunks of 128 channels with float4 loads; the next chunk is fetched into registers while the
// current one is being multiplied and written to the other buffer afterwards (one barrier per chunk). grid = (ceil(rows/(32*RT)), ceil(Cout/256)).
// ------------------------------------------------------------------------------------------
struct LinearParams {
    const float* X; const float* Wp; const float* scale; const float* shift; const float* residual; float* out;
    int rows, K, ldx, Cout, relu, ldr, ldo, nkb, NT, vec_ok;
    long long xb, wb, ob, rb;       // per-batch element strides (blockIdx.z = batch; all 0 for a plain launch)
    const float* in_a; const float* in_b;   // optional input transform x <- relu(x * in_a[k] + in_b[k]) applied while the A
                                            // operand is staged (training: the previous layer's BatchNorm + ReLU, never materialised)
};

constexpr int LIN_KC = 128;          // channels per staged chunk
constexpr int LIN_LDK = LIN_KC + 4;  // LDS row stride (== 4 mod 8)

// VEC: K % 4 == 0, ldx % 4 == 0 and X 16-byte aligned -> one float4 per slot, bounds by slot.
template <int RT, bool VEC>
__device__ __forceinline__ void lin_fetch(const LinearParams& p, int row0, int k0, int t, f32x4 (&st)[RT * 4]) {
#pragma unroll
    for (int i = 0; i < RT * 4; ++i) {
        const int e = t + i * 256;               // float4 slot inside the [RT*32][32] chunk
        const int r = e >> 5, c = k0 + ((e & 31) << 2);
        const int gr = row0 + r;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (VEC) {
            if (gr < p.rows && c < p.K) v = *reinterpret_cast<const f32x4*>(p.X + (size_t)gr * p.ldx + c);
        } else if (gr < p.rows) {
            const float* src = p.X + (size_t)gr * p.ldx + c;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (c + q < p.K) v[q] = src[q];
        }
        st[i] = v;
    }
    if (p.in_a) {                                // same channel quad for every slot of this thread (e & 31 == t & 31)
        const int c = k0 + ((t & 31) << 2);
        if (c < p.K) {
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(p.in_a + c), b4 = *reinterpret_cast<const f32x4*>(p.in_b + c);
#pragma unroll
            for (int i = 0; i < RT * 4; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) st[i][q] = fmaxf(__builtin_fmaf(st[i][q], a4[q], b4[q]), 0.f);
        }
    }
}

template <int RT>
__device__ __forceinline__ void lin_stage(float* Xs, int t, const f32x4 (&st)[RT * 4]) {
#pragma unroll
    for (int i = 0; i < RT * 4; ++i) {
        const int e = t + i * 256;
        *reinterpret_cast<f32x4*>(Xs + (e >> 5) * LIN_LDK + ((e & 31) << 2)) = st[i];
    }
}

template <int RT, bool VEC, int CT>   // a workgroup owns RT*32 rows x CT*128 columns
__global__ __launch_bounds__(256, 2) void linear_kernel(LinearParams p) {
    if (blockIdx.z) {               // batched launch: this workgroup's operands
        p.X += (size_t)blockIdx.z * p.xb; p.Wp += (size_t)blockIdx.z * p.wb; p.out += (size_t)blockIdx.z * p.ob;
        if (p.residual) p.residual += (size_t)blockIdx.z * p.rb;
    }
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;                                    // 2 x [RT*32][LIN_LDK]
    constexpr int BUF = RT * 32 * LIN_LDK;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int row0 = blockIdx.x * 32 * RT;
    const int ctbase = blockIdx.y * (4 * CT) + w;        // this wave's first column tile, the next is +4
    int nvalid = 0;
#pragma unroll
    for (int u = 0; u < CT; ++u)
        if (ctbase + 4 * u < p.NT) nvalid = u + 1;
    const int nchunks = (p.nkb * 8 + LIN_KC - 1) / LIN_KC;
    const size_t bstep = (size_t)p.NT * 64;

    f32x4 st[RT * 4];
    f32x16 acc[RT][CT];
    zero_acc(acc);
    lin_fetch<RT, VEC>(p, row0, 0, t, st);
    lin_stage<RT>(Xs, t, st);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        float* cur = Xs + (c & 1) * BUF;
        if (c + 1 < nchunks) lin_fetch<RT, VEC>(p, row0, (c + 1) * LIN_KC, t, st);
        const int nkb_c = min(LIN_KC / 8, p.nkb - c * (LIN_KC / 8));
        gemm_tiles<RT, CT, PTT_LINEAR_PF>(cur, LIN_LDK, nkb_c, reinterpret_cast<const f32x4*>(p.Wp) + (size_t)c * (LIN_KC / 8) * bstep,
                          p.NT, ctbase, nvalid, lane, acc);
        if (c + 1 < nchunks) {
            lin_stage<RT>(Xs + ((c + 1) & 1) * BUF, t, st);   // the other buffer: last read in chunk c-1
            __syncthreads();
        }
    }
    if (nvalid == 0) return;

    const int half = lane >> 5;
#pragma unroll
    for (int u = 0; u < CT; ++u) {
        if (u >= nvalid) break;
        const int col = (ctbase + 4 * u) * 32 + (lane & 31);
        if (col >= p.Cout) continue;
        const float sc = p.scale ? p.scale[col] : 1.f;
        const float sh = p.shift ? p.shift[col] : 0.f;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gr = row0 + rt * 32 + tile_row(r, half);
                if (gr >= p.rows) continue;
                float y = acc[rt][u][r] * sc + sh;
                if (p.relu) y = fmaxf(y, 0.f);
                if (p.residual) y += p.residual[(size_t)gr * p.ldr + col];
                p.out[(size_t)gr * p.ldo + col] = y;
            }
    }
}

// ------------------------------------------------------------------------------------------
// The same layer for SHORT launches (a handful of frames: <= 8192 rows, K <= 520) — one tracklet frame runs ~30 of them
// back to back, 128 - 2048 rows each, and linear_kernel spends them waiting: its K loop restarts per 128-channel chunk (an
// exposed L2 round trip + a barrier every 16 K-blocks) with one weight fragment in flight per wave. Here a workgroup of
// 8 waves owns 32 rows x 128 columns: the whole X tile is staged once (one barrier), the two wave groups split the K axis
// in halves (half the dependent MFMA chain; partial sums meet in LDS), and every wave keeps LS_PD weight fragments in
// flight (4 registers each: a ring, refilled right behind the MFMAs that consumed a slot; at a handful of workgroups per
// launch the fragments come from HBM / Infinity Cache, ~1 us away: with 4 in flight a K-block cost 0.27 us, 4x its MFMA time).
// ------------------------------------------------------------------------------------------
constexpr int LS_PD = 8;
__global__ __launch_bounds__(512, 1) void linear_small_kernel(LinearParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int t = threadIdx.x, lane = t & 63, half = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6), ks = w >> 2, cw = w & 3;
    const int row0 = blockIdx.x * 32;
    const int ct = blockIdx.y * 4 + cw;                    // this wave's column tile
    const bool has_ct = ct < p.NT;
    // K-blocks of the two halves, padded to whole rings: [0, hb) and [hb, 2 hb); the X tile is zero beyond K
    const int hb = ((p.nkb + 1) / 2 + LS_PD - 1) / LS_PD * LS_PD;
    const int ldk = 2 * hb * 8 + 4;                         // == 4 (mod 8)
    float* Xs = smem;                                       // [32][ldk]
    float* Rs = smem + 32 * ldk;                            // [4 column tiles][16 registers][64 lanes] partial sums of the second half
    {   // stage the tile: float4 slots, zero beyond the rows / channels that exist
        const int qpr = 2 * hb * 2;                         // float4 slots per row (8 channels = 2 slots per K-block)
        for (int e = t; e < 32 * qpr; e += 512) {
            const int r = e / qpr, c = (e - r * qpr) << 2;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (row0 + r < p.rows && c < p.K) v = *reinterpret_cast<const f32x4*>(p.X + (size_t)(row0 + r) * p.ldx + c);
            *reinterpret_cast<f32x4*>(Xs + r * ldk + c) = v;
        }
    }
    const __amdgpu_buffer_rsrc_t wr = weight_rsrc(p.Wp);
    const int wvoff = ((has_ct ? ct : 0) * 64 + lane) * 16;
    const int wkstep = p.NT * 1024;
    const int kb0 = ks * hb, last = p.nkb - 1;
    f32x4 b[LS_PD];
#pragma unroll
    for (int i = 0; i < LS_PD; ++i) b[i] = weight_load(wr, wvoff, min(kb0 + i, last) * wkstep);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    __syncthreads();
    const float* arow = Xs + (lane & 31) * ldk + 4 * half + kb0 * 8;
    for (int k = 0; k < hb; k += LS_PD) {
#pragma unroll
        for (int i = 0; i < LS_PD; ++i) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(arow + (k + i) * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[i][j], acc, 0, 0, 0);
            b[i] = weight_load(wr, wvoff, min(kb0 + k + i + LS_PD, last) * wkstep);     // past the end: a valid fragment times zeros
        }
    }
    if (ks == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) Rs[(cw * 16 + r) * 64 + lane] = acc[r];
    }
    __syncthreads();
    if (ks == 1 || !has_ct) return;
    const int col = ct * 32 + (lane & 31);
    if (col >= p.Cout) return;
    const float sc = p.scale ? p.scale[col] : 1.f;
    const float sh = p.shift ? p.shift[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int gr = row0 + tile_row(r, half);
        if (gr >= p.rows) continue;
        float y = (acc[r] + Rs[(cw * 16 + r) * 64 + lane]) * sc + sh;
        if (p.relu) y = fmaxf(y, 0.f);
        if (p.residual) y += p.residual[(size_t)gr * p.ldr + col];
        p.out[(size_t)gr * p.ldo + col] = y;
    }
}

// ------------------------------------------------------------------------------------------
// Fused set-abstraction level. A workgroup owns 64 grouped rows = 64/NS centres.
// ------------------------------------------------------------------------------------------
struct SaLayerDev { const float* Wp; const float* scale; const float* shift; int Cin, Cout, relu, nkb, NT; };
struct SaParams {
    const float* xyz; const float* new_xyz; const int32_t* idx; const float* feat; float* out;
    long long fsb, fsc, fsn, osb, osc, osm;
    int B, N, M, C, use_xyz, normalize, n_layers, ldk, K0, first_wave, stagger, vec_gather;
    int tiles, chunk;     // sa_stream_kernel: 64-row tiles in the launch, consecutive tiles per (persistent) workgroup
    int hoist, l0_relu;   // layer 0 hoisted: feat = per-point term (B,N,C), wx = (3,C) weights of the relative coordinates
    const float* wx;
    float radius;
    long long* dbg;   // dev only (PTT_DEBUG_STAMPS)
    SaLayerDev L[PTT_SA_MAX_LAYERS];
};

// ------------------------------------------------------------------------------------------
// A stack of up to four 1x1 convolutions (+ folded BatchNorm, + ReLU) over point rows in ONE launch: the heads'
// Conv1d stacks (vote_layer 259 -> 256 -> 256 -> 259 with the input as residual, cla_layer 256 -> 256 -> 256 -> 1,
// refine_layer ... -> 5) and CosineSimAug's two trailing convolutions. A workgroup owns 32 rows and all columns; the
// activations stay in its LDS tile between layers. One launch instead of one ptt_linear_f32 launch per layer: at
// 128 - 6144 rows each of those is launch- and tail-bound (14 us at one frame, 27 us at 48).
// Inner layers: Cout <= 256 (two column tiles per wave); last layer: Cout <= 384, any width (padded tiles are not
// stored); K <= 264.
// ------------------------------------------------------------------------------------------
struct RowsMlpParams {
    const float* X; const float* residual; float* out;
    int rows, K0, ldx, ldr, ldo, n_layers, ldk, vec_in;
    SaLayerDev L[PTT_SA_MAX_LAYERS];
};

__global__ __launch_bounds__(256, 2) void rows_mlp_kernel(RowsMlpParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;                                       // [32][ldk]
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, half = lane >> 5;
    const int row0 = blockIdx.x * 32;
    f32x4 pw[3];
    float psh[3];
    auto prefetch = [&](const SaLayerDev& L) {               // first K-block of this wave's column tiles + their shifts
        const f32x4* bp = reinterpret_cast<const f32x4*>(L.Wp) + (size_t)w * 64 + lane;
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            psh[u] = 0.f;
            if (w + 4 * u < L.NT) {
                pw[u] = bp[(size_t)u * 4 * 64];
                const int col = (w + 4 * u) * 32 + (lane & 31);
                if (L.shift && !L.scale && col < L.Cout) psh[u] = L.shift[col];
            }
        }
    };
    prefetch(p.L[0]);
    // ---- stage the 32 input rows, zero-padded to the K-block boundary (rows past the end: zeros) ----
    {
        const int Kpad = p.L[0].nkb * 8;
        if (p.vec_in) {
            const int nq = Kpad >> 2;
            for (int e = t; e < 32 * nq; e += 256) {
                const int r = e / nq, c4 = (e - r * nq) * 4;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (row0 + r < p.rows && c4 < p.K0) v = *reinterpret_cast<const f32x4*>(p.X + (size_t)(row0 + r) * p.ldx + c4);
                *reinterpret_cast<f32x4*>(Xs + r * p.ldk + c4) = v;
            }
        } else {
            for (int e = t; e < 32 * Kpad; e += 256) {
                const int r = e / Kpad, c = e - r * Kpad;
                Xs[r * p.ldk + c] = (row0 + r < p.rows && c < p.K0) ? p.X[(size_t)(row0 + r) * p.ldx + c] : 0.f;
            }
        }
    }
    lds_barrier();
    for (int l = 0; l < p.n_layers; ++l) {
        const SaLayerDev& L = p.L[l];
        const bool last = l == p.n_layers - 1, affine = L.scale != nullptr;
        f32x16 acc[1][3];
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][u][r] = psh[u];
        int nvalid = 0;
#pragma unroll
        for (int u = 0; u < 3; ++u)
            if (w + 4 * u < L.NT) nvalid = u + 1;
        if (nvalid) gemm_tiles<1, 3, 0>(Xs, p.ldk, L.nkb, reinterpret_cast<const f32x4*>(L.Wp), L.NT, w, nvalid, lane, acc, pw);
        if (!last) { prefetch(p.L[l + 1]); lds_barrier(); }     // every wave has read this layer's input
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            if (u >= nvalid) break;
            const int col = (w + 4 * u) * 32 + (lane & 31);
            float sc = 1.f, sh = 0.f;
            if (affine && col < L.Cout) { sc = L.scale[col]; sh = L.shift ? L.shift[col] : 0.f; }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float y = affine ? acc[0][u][r] * sc + sh : acc[0][u][r];
                if (L.relu) y = fmaxf(y, 0.f);
                const int row = tile_row(r, half);
                if (!last) {
                    Xs[row * p.ldk + col] = (col < L.Cout) ? y : 0.f;       // padded columns feed the next layer as zeros
                } else if (col < L.Cout && row0 + row < p.rows) {
                    if (p.residual) y += p.residual[(size_t)(row0 + row) * p.ldr + col];
                    p.out[(size_t)(row0 + row) * p.ldo + col] = y;
                }
            }
        }
        if (!last) lds_barrier();
    }
}

// ------------------------------------------------------------------------------------------
// Grouping for the SA kernels: the calling wave fills ROWS consecutive rows of an LDS tile with
// [ neighbour features (C) | (xyz_nbr - centre)(/radius) (3) | zeros up to Kpad0 ]
// (features FIRST so that point-major rows land 16-byte aligned; layer 0's weight is packed with
// the matching rotation, see ptt_pack_weight_rot_f32). Memory-level parallelism is the point:
// all index loads, then all feature loads of the wave's rows are in flight together — a
// row-at-a-time loop serialises ~16 L2 round trips and was HALF of the kernel's time.
// ------------------------------------------------------------------------------------------
template <int NS, int ROWS>
__device__ __forceinline__ void sa_gather_rows(const SaParams& p, float* Xt, int row0, int centre0, int lane) {
    static_assert(ROWS == 8 || ROWS == 16 || ROWS == 32, "rows per wave");
    constexpr int GB = ROWS < PTT_GATHER_BATCH ? ROWS : PTT_GATHER_BATCH;
    const int total_centres = p.B * p.M;
    const int Kpad0 = p.L[0].nkb * 8;
    int b_l = 0, n_l = 0;
    float dx = 0.f, dy = 0.f, dz = 0.f;
    if (lane < ROWS) {
        const int r = row0 + lane;
        int c = centre0 + r / NS;
        if (c >= total_centres) c = total_centres - 1;
        b_l = c / p.M;
        n_l = p.idx[(size_t)c * NS + (r % NS)];
        if (p.use_xyz) {
            const size_t flat = (size_t)b_l * p.N + n_l;
            dx = p.xyz[flat * 3 + 0] - p.new_xyz[(size_t)c * 3 + 0];
            dy = p.xyz[flat * 3 + 1] - p.new_xyz[(size_t)c * 3 + 1];
            dz = p.xyz[flat * 3 + 2] - p.new_xyz[(size_t)c * 3 + 2];
            if (p.normalize) { dx /= p.radius; dy /= p.radius; dz /= p.radius; }
            if (!p.hoist) {
                float* x = Xt + r * p.ldk + p.C;
                x[0] = dx; x[1] = dy; x[2] = dz;
            }
        }
        for (int c2 = p.K0; c2 < Kpad0; ++c2) Xt[r * p.ldk + c2] = 0.f;
    }
    if (p.C == 0) return;
    if (p.hoist == 2) {
        // Hoisted layer 0 with 128 channels: a row is 32 float4, so ONE wave instruction handles TWO rows (lanes 0-31 /
        // 32-63). Row index, cloud and relative coordinates reach the lanes by ds_bpermute; the per-point terms come
        // through a buffer descriptor with one 32-bit offset per lane. ~17 instructions per row pair instead of ~36
        // for two single rows — in this phase the wave gets about one issue slot per MFMA of the co-resident
        // workgroup (DESIGN.md lesson 8), so the instruction count IS the duration of the phase.
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        constexpr int NP = ROWS / 2;
        const int sub = lane >> 5, q = lane & 31;
        const __amdgpu_buffer_rsrc_t rf = weight_rsrc(p.feat);
        const f32x4 wx0 = *reinterpret_cast<const f32x4*>(p.wx + q * 4);
        const f32x4 wx1 = *reinterpret_cast<const f32x4*>(p.wx + 128 + q * 4);
        const f32x4 wx2 = *reinterpret_cast<const f32x4*>(p.wx + 256 + q * 4);
        int off[NP];
        float rx[NP], ry[NP], rz[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int r = 2 * i + sub;
            const int nn = __shfl(n_l, r, 64), bb = __shfl(b_l, r, 64);
            off[i] = ((bb * p.N + nn) * 128 + q * 4) * (int)sizeof(float);
            rx[i] = __shfl(dx, r, 64); ry[i] = __shfl(dy, r, 64); rz[i] = __shfl(dz, r, 64);
        }
        f32x4 v[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) v[i] = weight_load(rf, off[i], 0);
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const f32x2 rx2 = {rx[i], rx[i]}, ry2 = {ry[i], ry[i]}, rz2 = {rz[i], rz[i]};
#pragma unroll
            for (int j = 0; j < 4; j += 2) {
                f32x2 y = {v[i][j], v[i][j + 1]};
                y = __builtin_elementwise_fma(f32x2{wx0[j], wx0[j + 1]}, rx2, y);
                y = __builtin_elementwise_fma(f32x2{wx1[j], wx1[j + 1]}, ry2, y);
                y = __builtin_elementwise_fma(f32x2{wx2[j], wx2[j + 1]}, rz2, y);
                v[i][j] = p.l0_relu ? fmaxf(y[0], 0.f) : y[0];
                v[i][j + 1] = p.l0_relu ? fmaxf(y[1], 0.f) : y[1];
            }
            *reinterpret_cast<f32x4*>(Xt + (row0 + 2 * i + sub) * p.ldk + q * 4) = v[i];
        }
        return;
    }
    if (p.vec_gather) {                                  // point-major rows: one float4 per lane per row
        const int nq = p.C >> 2;
        f32x4 wx0 = {0.f, 0.f, 0.f, 0.f}, wx1 = wx0, wx2 = wx0;
        if (p.hoist && lane < nq) {
            wx0 = *reinterpret_cast<const f32x4*>(p.wx + lane * 4);
            wx1 = *reinterpret_cast<const f32x4*>(p.wx + p.C + lane * 4);
            wx2 = *reinterpret_cast<const f32x4*>(p.wx + 2 * p.C + lane * 4);
        }
#pragma unroll
        for (int base = 0; base < ROWS; base += GB) {
            f32x4 v[GB];
#pragma unroll
            for (int i = 0; i < GB; ++i) {
                const int bb = __builtin_amdgcn_readlane(b_l, base + i), nn = __builtin_amdgcn_readlane(n_l, base + i);
                const float* src = p.feat + (long long)bb * p.fsb + (long long)nn * p.fsn;
                if (lane < nq) v[i] = *reinterpret_cast<const f32x4*>(src + lane * 4);
            }
#pragma unroll
            for (int i = 0; i < GB; ++i) {
                if (p.hoist) {            // layer 0 of the MLP: per-point term + Wx . rel, activation — straight into the tile
                    const float rx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dx), base + i));
                    const float ry = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dy), base + i));
                    const float rz = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dz), base + i));
                    // packed fp32 FMAs (v_pk_fma_f32: two channels per instruction): 6 instead of 12 per row
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
                    const f32x2 rx2 = {rx, rx}, ry2 = {ry, ry}, rz2 = {rz, rz};
#pragma unroll
                    for (int j = 0; j < 4; j += 2) {
                        f32x2 y = {v[i][j], v[i][j + 1]};
                        y = __builtin_elementwise_fma(f32x2{wx0[j], wx0[j + 1]}, rx2, y);
                        y = __builtin_elementwise_fma(f32x2{wx1[j], wx1[j + 1]}, ry2, y);
                        y = __builtin_elementwise_fma(f32x2{wx2[j], wx2[j + 1]}, rz2, y);
                        v[i][j] = p.l0_relu ? fmaxf(y[0], 0.f) : y[0];
                        v[i][j + 1] = p.l0_relu ? fmaxf(y[1], 0.f) : y[1];
                    }
                }
                if (lane < nq) *reinterpret_cast<f32x4*>(Xt + (row0 + base + i) * p.ldk + lane * 4) = v[i];
            }
        }
    } else {                                             // any strides: 64/ROWS lane groups stride over the channels
        constexpr int G = 64 / ROWS;
        const int r = lane % ROWS, g = lane / ROWS;
        const int bb = __shfl(b_l, r, 64), nn = __shfl(n_l, r, 64);
        const float* src = p.feat + (long long)bb * p.fsb + (long long)nn * p.fsn;
        float* dst = Xt + (row0 + r) * p.ldk;
        int c = g;
        for (; c + 15 * G < p.C; c += 16 * G) {
            float v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = src[(long long)(c + i * G) * p.fsc];
#pragma unroll
            for (int i = 0; i < 16; ++i) dst[c + i * G] = v[i];
        }
        for (; c < p.C; c += G) dst[c] = src[(long long)c * p.fsc];
    }
}

template <int NS, int CT, int RT = 2>
__device__ __forceinline__ void sa_layer(const SaParams& p, const SaLayerDev& L, bool last, float* Xs, int lane, int w,
                                         int centre0, int ncentres, SaPre& pre, const SaLayerDev* Lnext) {
    static_assert(NS != 64 || RT == 2, "a 64-neighbour centre spans two row tiles");
    // Everything outside the MFMA loop costs matrix time (fp32 MFMA and the vector ALU are one resource, DESIGN.md
    // lesson 8), so the epilogue is kept to one instruction per value where the layer allows it: with the BatchNorm
    // scale folded into the packed weights (L.scale == NULL) the shift starts the accumulators, an inner layer is
    // max(acc, 0) + one LDS write per value, and the last layer pools FIRST and applies the ReLU to the pooled value
    // (max and ReLU commute).
    const bool affine = L.scale != nullptr;              // wave-uniform: separate scale (and shift) in the epilogue
    f32x16 acc[RT][CT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int u = 0; u < CT; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rt][u][r] = pre.sh[u];          // 0 unless the shift rides in the accumulator
    int nvalid = 0;
#pragma unroll
    for (int u = 0; u < CT; ++u)
        if (w + 4 * u < L.NT) nvalid = u + 1;
    gemm_tiles<RT, CT, 0>(Xs, p.ldk, L.nkb, reinterpret_cast<const f32x4*>(L.Wp), L.NT, w, nvalid, lane, acc, pre.w);
    if (Lnext) prefetch_first_block(Lnext->Wp, Lnext->scale, Lnext->shift, Lnext->NT, w, lane, pre);   // in flight across the epilogue + barriers
    if (!last) lds_barrier();
    // every wave has finished reading this layer's input tile (the last layer writes no LDS)

    const int half = lane >> 5;
#pragma unroll
    for (int u = 0; u < CT; ++u) {
        if (u >= nvalid) break;
        const int col = (w + 4 * u) * 32 + (lane & 31);  // < Cout: the host admits only Cout % 32 == 0
        float sc = 1.f, sh = 0.f;
        if (affine) { sc = L.scale[col]; sh = L.shift ? L.shift[col] : 0.f; }
        float m64 = -__builtin_inff();                 // NS == 64: one centre spans both row tiles
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            float y[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) y[r] = affine ? acc[rt][u][r] * sc + sh : acc[rt][u][r];
            if (!last) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    Xs[(rt * 32 + tile_row(r, half)) * p.ldk + col] = L.relu ? fmaxf(y[r], 0.f) : y[r];
            } else if (NS == 64) {
                float m = y[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) m = fmaxf(m, y[r]);
                m64 = fmaxf(m64, max_halves(m));
                if (rt == RT - 1 && half == 0 && ncentres > 0) {
                    const int b = centre0 / p.M, mm = centre0 - b * p.M;
                    p.out[b * p.osb + col * p.osc + mm * p.osm] = L.relu ? fmaxf(m64, 0.f) : m64;
                }
            } else if (NS == 32) {
                float m = y[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) m = fmaxf(m, y[r]);
                m = max_halves(m);
                if (L.relu) m = fmaxf(m, 0.f);
                const int c = centre0 + rt;
                if (half == 0 && rt < ncentres) {
                    const int b = c / p.M, mm = c - b * p.M;
                    p.out[b * p.osb + col * p.osc + mm * p.osm] = m;
                }
            } else {  // NS == 16: rows 0..15 (regs 0..7) and rows 16..31 (regs 8..15) are two centres
                float m0 = y[0], m1 = y[8];
#pragma unroll
                for (int r = 1; r < 8; ++r) { m0 = fmaxf(m0, y[r]); m1 = fmaxf(m1, y[8 + r]); }
                m0 = max_halves(m0);
                m1 = max_halves(m1);
                if (L.relu) { m0 = fmaxf(m0, 0.f); m1 = fmaxf(m1, 0.f); }
                if (half == 0) {
                    const int ca = rt * 2, cb = rt * 2 + 1;
                    if (ca < ncentres) {
                        const int c = centre0 + ca; const int b = c / p.M, mm = c - b * p.M;
                        p.out[b * p.osb + col * p.osc + mm * p.osm] = m0;
                    }
                    if (cb < ncentres) {
                        const int c = centre0 + cb; const int b = c / p.M, mm = c - b * p.M;
                        p.out[b * p.osb + col * p.osc + mm * p.osm] = m1;
                    }
                }
            }
        }
    }
    if (!last) lds_barrier();
}

template <int NS, int RT>
__global__ __launch_bounds__(256, PTT_SA_WAVES) void sa_fused_kernel(SaParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;                                      // [32*RT][ldk]
    constexpr int CPW = 32 * RT / NS;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int total_centres = p.B * p.M;
    const int centre0 = logical_block() * CPW;
    const int ncentres = min(CPW, total_centres - centre0);
    stagger_second_slot(p.first_wave, p.stagger);
    PTT_STAMP(0);

    SaPre pre;
    pre.w[0] = pre.w[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    prefetch_first_block(p.L[0].Wp, p.L[0].scale, p.L[0].shift, p.L[0].NT, w, lane, pre);      // layer 0's first weight block rides along the gather

    // ---- group: neighbour features + relative (normalised) coordinates -> X; wave w fills 8*RT consecutive rows ----
    sa_gather_rows<NS, 8 * RT>(p, Xs, w * 8 * RT, centre0, lane);
    lds_barrier();

    PTT_STAMP(1);
    for (int l = 0; l < p.n_layers; ++l) {
        const SaLayerDev& L = p.L[l];
        const bool last = (l == p.n_layers - 1);
        const int ctw = (L.NT + 3) >> 2;
        const SaLayerDev* Ln = last ? nullptr : &p.L[l + 1];
        if (ctw <= 1) sa_layer<NS, 1, RT>(p, L, last, Xs, lane, w, centre0, ncentres, pre, Ln);
        else sa_layer<NS, 2, RT>(p, L, last, Xs, lane, w, centre0, ncentres, pre, Ln);
        PTT_STAMP(2 + l);
    }
}

// ------------------------------------------------------------------------------------------
// Streaming form of the fused set-abstraction level for the two levels that carry most of the SA work (SA1, SA2 of
// both branches: hoisted layer 0 with 128 channels, then 128 -> 128 -> 256, 32 neighbours).
//
// Why: fp32 MFMA and the vector ALU are one SIMD resource, and a wave that issues vector-ALU / LDS instructions while
// ANOTHER wave streams MFMAs on the same SIMD gets roughly one issue slot per MFMA (DESIGN.md lesson 8): the gather of
// sa_fused_kernel (~150 instructions per wave) costs ~17k cycles beside the co-resident workgroup's GEMM although it
// is ~1k cycles of work. Instructions of the SAME wave issue in order at full rate between its own MFMAs. So here a
// workgroup is persistent (it walks `chunk` consecutive 64-row tiles), the LDS tile is double-buffered, and every wave
// gathers ITS 16 rows of tile n+1 inside its own MFMA stream of tile n's last (longest) GEMM: per pair of K-blocks one
// row pair — a broadcast ds_read of (row offset, rel), one buffer load, 12 FMAs + 4 max, one ds_write_b128.
// ------------------------------------------------------------------------------------------
constexpr int SAS_LDK = 132;                 // 128 channels + 4 (== 4 mod 8: conflict-free A reads)
constexpr int SAS_TILE = 64 * SAS_LDK;

// one row pair of the wave's 16 rows: request (rows 2i, 2i+1 -> lanes 0-31 / 32-63, four channels per lane) ...
__device__ __forceinline__ f32x4 sas_pair_load(const float* meta, __amdgpu_buffer_rsrc_t rf, int w, int i, int sub, int q,
                                               f32x4& m) {
    m = *reinterpret_cast<const f32x4*>(meta + (w * 16 + 2 * i + sub) * 4);
    return weight_load(rf, __builtin_bit_cast(int, m[0]) + q * 16, 0);
}
// ... and finish: h0 = relu(term + Wx . rel) -> the tile (scalar FMAs: v_pk_fma_f32 beside MFMAs is an anti-lever,
// MI355X_MICROARCH.md "price of one filler beside MFMAs")
__device__ __forceinline__ void sas_pair_store(float* X, int w, int i, int sub, int q, f32x4 v, const f32x4& m, const f32x4& wx0,
                                               const f32x4& wx1, const f32x4& wx2, float floor) {   // floor: 0 = ReLU, -inf = none
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float y = __builtin_fmaf(wx0[j], m[1], v[j]);
        y = __builtin_fmaf(wx1[j], m[2], y);
        y = __builtin_fmaf(wx2[j], m[3], y);
        v[j] = fmaxf(y, floor);
    }
    *reinterpret_cast<f32x4*>(X + (16 * w + 2 * i + sub) * SAS_LDK + q * 4) = v;
}

// a centre index with its (cloud, index inside the cloud), advanced without divisions: wave-uniform, scalar ALU only
struct SasCentre {
    int c, b, m;
    __device__ __forceinline__ void advance(int k, int M) {
        c += k; m += k;
        while (m >= M) { m -= M; ++b; }
    }
};

// max over the 32 neighbour rows of one 32 x 32 accumulator tile (16 registers x 2 half-waves), + shift, ReLU
__device__ __forceinline__ float sas_pool(const f32x16& a, float sh, int relu) {
    float mx = a[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, a[r]);
    mx = max_halves(mx) + sh;
    return relu ? fmaxf(mx, 0.f) : mx;
}

#ifndef PTT_SAS_INTERLEAVE
#define PTT_SAS_INTERLEAVE 0    // 1: one gather instruction behind each MFMA (measured 3.5 % SLOWER than the block behind the last MFMA)
#endif
#ifndef PTT_SAS_EXP
#define PTT_SAS_EXP 0    // timing experiments only (wrong results): 1 no barriers, 2 no gather, 4 no h1 writes, 8 no pool, 16 no meta
#endif
template <int NS>
__global__ __launch_bounds__(256, 2) void sa_stream_kernel(SaParams p) {
    static_assert(NS == 32, "one centre per 32-row tile");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const P = smem;                                 // h0 of the current tile (gathered during the previous tile's layer 2)
    float* const Q = smem + SAS_TILE;                      // h1 of the current tile
    float* const meta = smem + 2 * SAS_TILE;               // [4 waves][16 rows][off, dx, dy, dz]
    const int t = threadIdx.x, lane = t & 63, half = lane >> 5, sub = half, q = lane & 31;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int tile0 = logical_block() * p.chunk;
    const int ntiles = min(p.chunk, p.tiles - tile0);
    if (ntiles <= 0) return;
    stagger_second_slot(p.first_wave, p.stagger);
    const SaLayerDev& L1 = p.L[0];
    const SaLayerDev& L2 = p.L[1];
    const __amdgpu_buffer_rsrc_t rf = weight_rsrc(p.feat);
    const __amdgpu_buffer_rsrc_t wr1 = weight_rsrc(L1.Wp), wr2 = weight_rsrc(L2.Wp);
    const int wvoff = (w * 64 + lane) * 16;                // this lane's byte offset inside a K-block row of fragments
    constexpr int WK1 = 4 * 1024, WK2 = 8 * 1024;          // bytes per K-block of the packed weights (NT = 4 / 8)
    const f32x4 wx0 = *reinterpret_cast<const f32x4*>(p.wx + q * 4);
    const f32x4 wx1 = *reinterpret_cast<const f32x4*>(p.wx + 128 + q * 4);
    const f32x4 wx2 = *reinterpret_cast<const f32x4*>(p.wx + 256 + q * 4);
    const int col = lane & 31;
    const float sh1 = L1.shift ? L1.shift[w * 32 + col] : 0.f;
    const float sh2a = L2.shift ? L2.shift[w * 32 + col] : 0.f;
    const float sh2b = L2.shift ? L2.shift[(w + 4) * 32 + col] : 0.f;
    f32x16 sh1v;                                           // layer 1's shift as the C operand of its first MFMAs
#pragma unroll
    for (int r = 0; r < 16; ++r) sh1v[r] = sh1;
    const int total_centres = p.B * p.M;
    const float floor0 = p.l0_relu ? 0.f : -__builtin_inff();
    const float rdiv = p.normalize ? p.radius : 1.0f;      // x / 1 is exact: one code path, no branch in the MFMA stream
    const int M = p.M, N = p.N;
    const int osb = (int)p.osb, osm = (int)p.osm;
    const int ocol0 = (w * 32 + col) * (int)p.osc, ocol1 = ((w + 4) * 32 + col) * (int)p.osc;

    // the wave's own centre of a tile (rows 16w .. 16w+15 belong to centre 2*tile + (w >> 1)) and the tile's first centre
    SasCentre own, first;
    {
        const int c0 = tile0 * 2;
        first.c = c0; first.b = c0 / M; first.m = c0 - first.b * M;
        own = first;
        own.advance(w >> 1, M);
    }
    // index -> (relative coordinates, byte offset of the neighbour's per-point term row) of the wave's 16 rows, all
    // 64 lanes redundantly (lane & 15): no exec-mask change, so the pieces can sit inside an MFMA stream
    auto meta_index = [&](const SasCentre& ce) -> int {
        const int c = ce.c < total_centres ? ce.c : total_centres - 1;
        return p.idx[(size_t)c * 32 + 16 * (w & 1) + (lane & 15)];
    };
    struct MetaXyz { float x, y, z, cx, cy, cz; int flat; };
    auto meta_fetch = [&](const SasCentre& ce, int n) -> MetaXyz {
        const bool in = ce.c < total_centres;
        const int c = in ? ce.c : total_centres - 1;
        const int b = in ? ce.b : p.B - 1;
        MetaXyz r;
        r.flat = b * N + n;
        r.x = p.xyz[(size_t)r.flat * 3 + 0]; r.y = p.xyz[(size_t)r.flat * 3 + 1]; r.z = p.xyz[(size_t)r.flat * 3 + 2];
        r.cx = p.new_xyz[(size_t)c * 3 + 0]; r.cy = p.new_xyz[(size_t)c * 3 + 1]; r.cz = p.new_xyz[(size_t)c * 3 + 2];
        return r;
    };
    auto meta_store = [&](const MetaXyz& r) {
        const float dx = (r.x - r.cx) / rdiv, dy = (r.y - r.cy) / rdiv, dz = (r.z - r.cz) / rdiv;
        const int off = r.flat * (128 * (int)sizeof(float));
        *reinterpret_cast<f32x4*>(meta + (w * 16 + (lane & 15)) * 4) = f32x4{__builtin_bit_cast(float, off), dx, dy, dz};
    };

    // ---- prologue: the first tile is gathered in the open ----
    f32x4 pre1[2] = {weight_load(wr1, wvoff, 0), weight_load(wr1, wvoff, WK1)};   // layer 1's first two weight blocks
    {
        const int n = meta_index(own);
        meta_store(meta_fetch(own, n));
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            f32x4 m;
            const f32x4 v = sas_pair_load(meta, rf, w, i, sub, q, m);
            sas_pair_store(P, w, i, sub, q, v, m, wx0, wx1, wx2, floor0);
        }
    }

    f32x16 acc[2][2];                                      // layer 2's accumulators: pooled inside the NEXT tile's layer 1
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rt][u][r] = 0.f;
    float pv[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    SasCentre prev = first;                                // first centre of the tile whose accumulators are pending
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    auto store_pooled = [&](const SasCentre& f) {          // the two centres of a tile: scalar index arithmetic
        SasCentre c1 = f;
        c1.advance(1, M);
        if (half == 0) {
            if (f.c < total_centres) {
                float* o = p.out + (f.b * osb + f.m * osm);
                o[ocol0] = pv[0][0]; o[ocol1] = pv[0][1];
            }
            if (c1.c < total_centres) {
                float* o = p.out + (c1.b * osb + c1.m * osm);
                o[ocol0] = pv[1][0]; o[ocol1] = pv[1][1];
            }
        }
    };

    for (int it = 0; it < ntiles; ++it) {
        const bool more = it + 1 < ntiles;                               // last tile: re-gathers itself (branch-free loop)
        SasCentre own_next = own;
        if (more) own_next.advance(2, M);
        if (!(PTT_SAS_EXP & 1)) lds_barrier();                           // A: P is complete; every wave is done with Q
        const int nn = meta_index(own_next);                             // index load in flight under the first K-blocks
        MetaXyz mx;

        // ---- layer 1: 128 -> 128, this wave's column tile w, both row tiles; inside its MFMA stream the max-pool of
        // the PREVIOUS tile's layer-2 accumulators and the (index -> coordinates -> meta) chain of the NEXT tile ----
        f32x16 acc1[2];
        {
            const float* arow = P + (lane & 31) * SAS_LDK + 4 * half;
            // weight fragments two K-blocks ahead (an L2 round trip is longer than the 8 MFMAs of one K-block), A one ahead
            f32x4 a0[2], a1[2], bc[2], bn[2];
#define SAS_A1(A, KB)                                                                           \
            {                                                                                   \
                A[0] = *reinterpret_cast<const f32x4*>(arow + (KB) * 8);                        \
                A[1] = *reinterpret_cast<const f32x4*>(arow + 32 * SAS_LDK + (KB) * 8);         \
            }
            bc[0] = pre1[0]; bc[1] = pre1[1];
            SAS_A1(a0, 0)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (i < 7) { bn[0] = weight_load(wr1, wvoff, (2 * i + 2) * WK1); bn[1] = weight_load(wr1, wvoff, (2 * i + 3) * WK1); }
                SAS_A1(a1, 2 * i + 1)
                if (i == 4 && !(PTT_SAS_EXP & 16)) mx = meta_fetch(own_next, nn);
                __builtin_amdgcn_sched_barrier(0);
                if (i == 0) {
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt) acc1[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[rt][0], bc[0][0], sh1v, 0, 0, 0);
#pragma unroll
                    for (int j = 1; j < 4; ++j)
#pragma unroll
                        for (int rt = 0; rt < 2; ++rt)
                            acc1[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[rt][j], bc[0][j], acc1[rt], 0, 0, 0);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int rt = 0; rt < 2; ++rt)
                            acc1[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[rt][j], bc[0][j], acc1[rt], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (i < 7) SAS_A1(a0, 2 * i + 2)
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt)
                        acc1[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[rt][j], bc[1][j], acc1[rt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (i < 4 && !(PTT_SAS_EXP & 8)) pv[i >> 1][i & 1] = sas_pool(acc[i >> 1][i & 1], (i & 1) ? sh2b : sh2a, L2.relu);
                if (i == 6 && !(PTT_SAS_EXP & 16)) meta_store(mx);
                __builtin_amdgcn_sched_barrier(0);
                bc[0] = bn[0]; bc[1] = bn[1];
            }
#undef SAS_A1
        }
        f32x4 pre2[2][2];                                                // layer 2's first two K-blocks ride across the epilogue
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            pre2[k][0] = weight_load(wr2, wvoff, k * WK2);
            pre2[k][1] = weight_load(wr2, wvoff, k * WK2 + 4 * 1024);
        }
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r = (PTT_SAS_EXP & 4) ? 15 : 0; r < 16; ++r)
                Q[(rt * 32 + tile_row(r, half)) * SAS_LDK + w * 32 + col] = fmaxf(acc1[rt][r], 0.f);
        if (it > 0) store_pooled(prev);
        if (!(PTT_SAS_EXP & 1)) lds_barrier();                           // C: h1 is complete; every wave is done with P

        // ---- layer 2: 128 -> 256 (column tiles w, w+4), with the gather of the next tile inside the MFMA stream ----
        {
            const float* arow = Q + (lane & 31) * SAS_LDK + 4 * half;
            f32x4 a0[2], a1[2], bc[2][2], bn[2][2];
#define SAS_A2(A, KB)                                                                           \
            {                                                                                   \
                A[0] = *reinterpret_cast<const f32x4*>(arow + (KB) * 8);                        \
                A[1] = *reinterpret_cast<const f32x4*>(arow + 32 * SAS_LDK + (KB) * 8);         \
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) { bc[k][0] = pre2[k][0]; bc[k][1] = pre2[k][1]; }
            SAS_A2(a0, 0)
            // (row offset, rel) of row pair i is read one iteration ahead: the feature load that depends on it must not
            // put an LDS round trip in front of the iteration's MFMAs
            f32x4 mc = *reinterpret_cast<const f32x4*>(meta + (w * 16 + sub) * 4), mn = mc;
#pragma unroll
            for (int i = 0; i < 8; ++i) {                                // K-blocks 2i, 2i+1 and row pair i of the next tile
                if (i < 7) {
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        bn[k][0] = weight_load(wr2, wvoff, (2 * i + 2 + k) * WK2);
                        bn[k][1] = weight_load(wr2, wvoff, (2 * i + 2 + k) * WK2 + 4 * 1024);
                    }
                } else {                                                 // the next tile's first layer-1 weights
                    pre1[0] = weight_load(wr1, wvoff, 0);
                    pre1[1] = weight_load(wr1, wvoff, WK1);
                }
                SAS_A2(a1, 2 * i + 1)
                f32x4 v = wx0;
                if (!(PTT_SAS_EXP & 2)) v = weight_load(rf, __builtin_bit_cast(int, mc[0]) + q * 16, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (i == 0) {
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int rt = 0; rt < 2; ++rt)
                            acc[rt][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[rt][0], bc[0][u][0], zero16, 0, 0, 0);
#pragma unroll
                    for (int j = 1; j < 4; ++j)
#pragma unroll
                        for (int u = 0; u < 2; ++u)
#pragma unroll
                            for (int rt = 0; rt < 2; ++rt)
                                acc[rt][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[rt][j], bc[0][u][j], acc[rt][u], 0, 0, 0);
                } else {
                    gemm_mfma_block<2, 2, 2>(a0, bc[0], acc);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (i < 7) {
                    SAS_A2(a0, 2 * i + 2)
                    mn = *reinterpret_cast<const f32x4*>(meta + (w * 16 + 2 * (i + 1) + sub) * 4);
                }
                __builtin_amdgcn_sched_barrier(0);
                gemm_mfma_block<2, 2, 2>(a1, bc[1], acc);
                if (!(PTT_SAS_EXP & 2)) sas_pair_store(P, w, i, sub, q, v, mc, wx0, wx1, wx2, floor0);
#if PTT_SAS_INTERLEAVE
                // one gather instruction behind each MFMA instead of a block of 17 behind the last one
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
                    __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);   // VALU
                }
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);       // the ds_write of the row pair
#endif
                __builtin_amdgcn_sched_barrier(0);
                mc = mn;
#pragma unroll
                for (int k = 0; k < 2; ++k) { bc[k][0] = bn[k][0]; bc[k][1] = bn[k][1]; }
            }
#undef SAS_A2
        }
        prev = first;
        if (more) { first.advance(2, M); own = own_next; }
    }
    // ---- the last tile's max over the 32 neighbours (ReLU after the pool), one centre per row tile ----
#pragma unroll
    for (int i = 0; i < 4; ++i) pv[i >> 1][i & 1] = sas_pool(acc[i >> 1][i & 1], (i & 1) ? sh2b : sh2a, L2.relu);
    store_pooled(prev);
}

// ------------------------------------------------------------------------------------------
// N1: the P2B cosine-similarity feature augmentation (CosineSimAug.forward, similarity_modules/
// p2b_xcoor.py:25-46) as one fused kernel on the same layer chain.
//   fusion[b, :, i, j] = [ cos(template_i, search_j) | template_xyz_i (3) | template_feat_i (C) ]
//   out[b, :, j]       = max_i  SharedMLP(fusion)[b, :, i, j]
// Only the similarity channel depends on j, so layer 0 is split algebraically:
//   W0 . fusion = w_sim * cos_ij + P[b, i, :],   P = W0[:, 1:] . [xyz_i ; feat_i]   (per template point, once
// per frame, on the linear kernel). A workgroup owns one search point j = 64 (i) rows: it computes the 64
// cosines, builds relu(bn0(w_sim * cos + P)) straight into the LDS tile and runs the remaining layers with the
// max over the 64 template points in registers. The (B,260,64,128) fusion tensor never exists.
// ------------------------------------------------------------------------------------------
struct XcorrParams {
    const float* P; const float* wsim; const float* scale0; const float* shift0;
    float* sim_out;
    const float* cos_t;   // (B,Ns,Nt) cosine map from cos_map_kernel
    int C0, Nt;
    SaParams sa;     // B, M (= Ns), out strides, ldk, layers (the remaining SharedMLP layers), stagger
    // split form (a handful of frames): the features of the cosine phase and the pair workspace
    const float* sfeat; const float* tfeat; long long s_sb, s_sn, t_sb, t_sn; int C; float eps;
    long long out_sh;               // element offset of the second half's output
};

// Cosine map of one frame batch: cos_t[b][j][i] = <templ_i, search_j> / (max(|templ_i|, eps) * max(|search_j|, eps))
// (torch.nn.functional.cosine_similarity, p2b_xcoor.py:35-36). One wave per search point, lane i = template point.
// 4 MFLOP per frame: a separate 10-us launch instead of a phase of every xcorr workgroup.
struct CosParams {
    const float* sfeat; const float* tfeat; float* cos_t;
    long long s_sb, s_sn, s_sc, t_sb, t_sn, t_sc;
    int B, Ns, Nt, C;
    float eps;
};
__global__ __launch_bounds__(256) void cos_map_kernel(CosParams q) {
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);       // flat search point
    if (j >= q.B * q.Ns) return;
    const int b = j / q.Ns, jj = j - b * q.Ns;
    const float* s = q.sfeat + (long long)b * q.s_sb + (long long)jj * q.s_sn;
    for (int i0 = 0; i0 < q.Nt; i0 += 64) {
        const int i = i0 + lane;
        const float* a = q.tfeat + (long long)b * q.t_sb + (long long)(i < q.Nt ? i : q.Nt - 1) * q.t_sn;
        float dot = 0.f, na = 0.f, ns = 0.f;
        if (q.t_sc == 1 && q.s_sc == 1 && (q.C & 3) == 0 && ((q.t_sn | q.t_sb | q.s_sn | q.s_sb) & 3) == 0 &&
            ((reinterpret_cast<uintptr_t>(q.tfeat) | reinterpret_cast<uintptr_t>(q.sfeat)) & 15) == 0) {
            for (int c = 0; c < q.C; c += 4) {
                const f32x4 av = *reinterpret_cast<const f32x4*>(a + c), sv = *reinterpret_cast<const f32x4*>(s + c);
#pragma unroll
                for (int k = 0; k < 4; ++k) { dot += av[k] * sv[k]; na += av[k] * av[k]; ns += sv[k] * sv[k]; }
            }
        } else {
            for (int c = 0; c < q.C; ++c) {
                const float av = a[(long long)c * q.t_sc], sv = s[(long long)c * q.s_sc];
                dot += av * sv; na += av * av; ns += sv * sv;
            }
        }
        if (i < q.Nt) q.cos_t[((long long)b * q.Ns + jj) * q.Nt + i] = dot / (fmaxf(sqrtf(na), q.eps) * fmaxf(sqrtf(ns), q.eps));
    }
}

// SPLIT (a handful of frames: B * Ns <= 512 search points would be as many workgroups, half the chip at one frame): TWO
// workgroups per search point, each with 32 of the template points (RT = 1), the cosines formed here instead of by
// cos_map_kernel (one launch less on a latency-bound chain). Each half writes ITS maximum (out + h * out_sh); the consumer
// takes the element-wise maximum of the two while it stages its operand (ptt_row_job.Xmax) — no atomics, no counters, and
// relu(max(a, b)) = max(relu(a), relu(b)).
template <int RT, bool SPLIT>
__global__ __launch_bounds__(256, 2) void xcorr_fused_kernel(XcorrParams q) {
    constexpr int ROWS = 32 * RT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const SaParams& p = q.sa;
    float* Xs = smem;                        // [ROWS][ldk]
    float* simv = smem + ROWS * p.ldk;       // [ROWS] cosine of the current template points with this search point
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, half = lane >> 5;
    const int npts = p.B * p.M;
    const int j = SPLIT ? (int)blockIdx.x % npts : logical_block();           // flat search point b*Ns + jj
    const int hsel = SPLIT ? (int)blockIdx.x / npts : 0;                        // which half of the template points (same XCD: npts % 8 == 0)
    const int b = j / p.M, jj = j - b * p.M;
    stagger_second_slot(p.first_wave, p.stagger);
    SaPre pre;
    pre.w[0] = pre.w[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    // running max over the template axis of this wave's (up to 2) column tiles: the template points are walked in
    // chunks of 64 rows (Nt = 64: one chunk; 512 template seeds of the stress configuration: eight)
    float run[2] = {-__builtin_inff(), -__builtin_inff()};
    const int nlast = p.n_layers - 1;
    const SaLayerDev& LL = p.L[nlast];

    const int ibeg = SPLIT ? hsel * (q.Nt >> 1) : 0, iend = SPLIT ? ibeg + (q.Nt >> 1) : q.Nt;
    for (int i0 = ibeg; i0 < iend; i0 += ROWS) {
        prefetch_first_block(p.L[0].Wp, p.L[0].scale, p.L[0].shift, p.L[0].NT, w, lane, pre);
        if (i0 != ibeg) __syncthreads();     // the previous chunk's last GEMM has read Xs / simv
        if constexpr (SPLIT) {
            // cosines of this chunk's 32 template points with the search point: 8 lanes per template point, 4 channels per
            // lane and step (unit channel stride, C % 4 == 0: checked by the host), folded with three xor-shuffles
            const int i = t >> 3, sub = t & 7;
            const float* a = q.tfeat + (long long)b * q.t_sb + (long long)(i0 + i) * q.t_sn;
            const float* sv_ = q.sfeat + (long long)b * q.s_sb + (long long)jj * q.s_sn;
            float dot = 0.f, na = 0.f, ns = 0.f;
            for (int c = sub * 4; c < q.C; c += 32) {
                const f32x4 av = *reinterpret_cast<const f32x4*>(a + c), sv = *reinterpret_cast<const f32x4*>(sv_ + c);
#pragma unroll
                for (int k = 0; k < 4; ++k) { dot += av[k] * sv[k]; na += av[k] * av[k]; ns += sv[k] * sv[k]; }
            }
#pragma unroll
            for (int m = 1; m < 8; m <<= 1) { dot += __shfl_xor(dot, m, 64); na += __shfl_xor(na, m, 64); ns += __shfl_xor(ns, m, 64); }
            if (sub == 0) simv[i] = dot / (fmaxf(sqrtf(na), q.eps) * fmaxf(sqrtf(ns), q.eps));
        } else {
            // the cosine map of the whole batch is computed once by cos_map_kernel (ptt_cosine_map_f32): 64 values to fetch
            // here instead of ~340 vector-ALU / load instructions per wave in a phase that runs beside another workgroup's
            // MFMA stream (and its address registers pushed this kernel over the 256-VGPR bound)
            if (t < ROWS) {
                const float cs = q.cos_t[(unsigned)((b * p.M + jj) * q.Nt + i0 + t)];
                simv[t] = cs;
                if (q.sim_out) q.sim_out[(unsigned)((b * q.Nt + i0 + t) * p.M + jj)] = cs;
            }
        }
        __syncthreads();

        // ---- layer 0: relu(bn0(w_sim * cos_i + P[b,i,:])) -> X ----
        const int nq = q.C0 >> 2;
        if (!q.scale0 && !q.shift0 && (q.C0 & 3) == 0 && (256 % nq) == 0) {
            // BatchNorm already folded into P and w_sim by the caller: relu(P'_i + w' * cos_i), four channels per lane
            // (one 16-byte load, two packed FMAs, four max, one 16-byte LDS write). A thread keeps ONE channel quad and
            // walks rows i, i + 256/nq, ...: no integer division per element (a run-time divisor is ~25 vector-ALU
            // instructions, paid in matrix time beside the other workgroup's GEMM) and w_sim's quad is loaded once.
            // (C0 / 4 not a power of two: the scalar form below.)
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            const int c4 = t % nq, step = 256 / nq;
            const f32x4 w4 = *reinterpret_cast<const f32x4*>(q.wsim + c4 * 4);
            const float* prow = q.P + ((size_t)(b * q.Nt + i0) * q.C0 + c4 * 4);
            for (int i = t / nq; i < ROWS; i += step) {
                const f32x4 pv = *reinterpret_cast<const f32x4*>(prow + (unsigned)(i * q.C0));
                const float cs = simv[i];
                const f32x2 c2 = {cs, cs};
                f32x2 lo = __builtin_elementwise_fma(f32x2{w4[0], w4[1]}, c2, f32x2{pv[0], pv[1]});
                f32x2 hi = __builtin_elementwise_fma(f32x2{w4[2], w4[3]}, c2, f32x2{pv[2], pv[3]});
                *reinterpret_cast<f32x4*>(Xs + i * p.ldk + c4 * 4) =
                    f32x4{fmaxf(lo[0], 0.f), fmaxf(lo[1], 0.f), fmaxf(hi[0], 0.f), fmaxf(hi[1], 0.f)};
            }
        } else {
            for (int c = t; c < q.C0; c += 256) {
                const float wsim = q.wsim[c], sc = q.scale0 ? q.scale0[c] : 1.f, sh = q.shift0 ? q.shift0[c] : 0.f;
                const float* pr = q.P + ((long long)b * q.Nt + i0) * q.C0 + c;
#pragma unroll 8
                for (int i = 0; i < ROWS; ++i) {
                    const float v = (pr[(long long)i * q.C0] + wsim * simv[i]) * sc + sh;
                    Xs[i * p.ldk + c] = fmaxf(v, 0.f);
                }
            }
        }
        __syncthreads();

        for (int l = 0; l < nlast; ++l) {
            const SaLayerDev& L = p.L[l];
            const int ctw = (L.NT + 3) >> 2;
            if (ctw <= 1) sa_layer<32 * RT, 1, RT>(p, L, false, Xs, lane, w, j, 1, pre, &p.L[l + 1]);
            else sa_layer<32 * RT, 2, RT>(p, L, false, Xs, lane, w, j, 1, pre, &p.L[l + 1]);
        }
        // ---- last layer: GEMM, then the max over this chunk's 64 template rows into the running max ----
        {
            const bool affine = LL.scale != nullptr;
            f32x16 acc[RT][2];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[rt][u][r] = pre.sh[u];
            int nvalid = 0;
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (w + 4 * u < LL.NT) nvalid = u + 1;
            gemm_tiles<RT, 2, 0>(Xs, p.ldk, LL.nkb, reinterpret_cast<const f32x4*>(LL.Wp), LL.NT, w, nvalid, lane, acc, pre.w);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (u >= nvalid) break;
                const int col = (w + 4 * u) * 32 + (lane & 31);
                float sc = 1.f, sh = 0.f;
                if (affine) { sc = LL.scale[col]; sh = LL.shift ? LL.shift[col] : 0.f; }
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    float m = affine ? acc[rt][u][0] * sc + sh : acc[rt][u][0];
#pragma unroll
                    for (int r = 1; r < 16; ++r) m = fmaxf(m, affine ? acc[rt][u][r] * sc + sh : acc[rt][u][r]);
                    run[u] = fmaxf(run[u], max_halves(m));
                }
            }
        }
    }
    if (half == 0) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (w + 4 * u >= LL.NT) break;
            const int col = (w + 4 * u) * 32 + (lane & 31);
            p.out[(SPLIT ? hsel * q.out_sh : 0) + b * p.osb + col * p.osc + jj * p.osm] = LL.relu ? fmaxf(run[u], 0.f) : run[u];
        }
    }
}

// ------------------------------------------------------------------------------------------
// Wave-private variant of the fused set-abstraction level, for levels whose whole weight set is
// small enough to live in L1/L2 (SA0: 3->64->64->128 = 50 KB): every wave owns 32 grouped rows
// (32/NS centres) and ALL output columns, its activation tile is private in LDS, so the layer
// chain runs without a single workgroup barrier and all four waves do the same amount of work
// (the column-split kernel leaves half the waves idle when a layer has only two column tiles).
// ------------------------------------------------------------------------------------------
template <int NS, int CT, bool AFFINE, int RT>
__device__ __forceinline__ void sa_wave_layer(const SaParams& p, const SaLayerDev& L, bool last, float* Xw, int lane,
                                              int centre0, int ncentres) {
    // same epilogue economy as sa_layer: shift in the accumulator when there is no separate scale (AFFINE false),
    // ReLU after the pool. (The shift is fetched here, not a layer ahead: the other waves of the SIMD cover the
    // round trip, and the extra live registers would spill at the VGPR bound of this kernel.)
    f32x16 acc[RT][CT];
#pragma unroll
    for (int u = 0; u < CT; ++u) {
        const float s0 = (!AFFINE && L.shift) ? L.shift[u * 32 + (lane & 31)] : 0.f;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rt][u][r] = s0;
    }
    gemm_core<RT, CT, CT, 1>(Xw, p.ldk, L.nkb, reinterpret_cast<const f32x4*>(L.Wp), L.NT, 0, lane, acc);
    __builtin_amdgcn_wave_barrier();
    const int half = lane >> 5;
#pragma unroll
    for (int u = 0; u < CT; ++u) {
        const int col = u * 32 + (lane & 31);
        float sc = 1.f, sv = 0.f;
        if constexpr (AFFINE) { sc = L.scale[col]; sv = L.shift ? L.shift[col] : 0.f; }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            float y[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) y[r] = AFFINE ? acc[rt][u][r] * sc + sv : acc[rt][u][r];
            if (!last) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    Xw[(rt * 32 + tile_row(r, half)) * p.ldk + col] = L.relu ? fmaxf(y[r], 0.f) : y[r];
            } else if (NS == 32) {
                float m = y[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) m = fmaxf(m, y[r]);
                m = max_halves(m);
                if (L.relu) m = fmaxf(m, 0.f);
                if (half == 0 && rt < ncentres) {
                    const int c = centre0 + rt; const int b = c / p.M, mm = c - b * p.M;
                    p.out[b * p.osb + col * p.osc + mm * p.osm] = m;
                }
            } else {
                float m0 = y[0], m1 = y[8];
#pragma unroll
                for (int r = 1; r < 8; ++r) { m0 = fmaxf(m0, y[r]); m1 = fmaxf(m1, y[8 + r]); }
                m0 = max_halves(m0);
                m1 = max_halves(m1);
                if (L.relu) { m0 = fmaxf(m0, 0.f); m1 = fmaxf(m1, 0.f); }
                if (half == 0) {
                    if (2 * rt < ncentres) {
                        const int c = centre0 + 2 * rt; const int b = c / p.M, mm = c - b * p.M;
                        p.out[b * p.osb + col * p.osc + mm * p.osm] = m0;
                    }
                    if (2 * rt + 1 < ncentres) {
                        const int c = centre0 + 2 * rt + 1; const int b = c / p.M, mm = c - b * p.M;
                        p.out[b * p.osb + col * p.osc + mm * p.osm] = m1;
                    }
                }
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// ------------------------------------------------------------------------------------------
// SA0 (no point features: rows are [rel.x rel.y rel.z], 3 -> 64 -> 64 -> 128, 32 neighbours): the activations never
// leave the registers. A PERSISTENT workgroup of 8 waves per CU holds the packed weights of layers 1 and 2 (48 KB) in
// LDS; every wave owns whole centres (32 grouped rows x all channels) and chains the layers TRANSPOSED:
//     layers 0, 1:  H^T[cout][row] = W[cout][k] . H_in^T[k][row]    weights as the MFMA A operand, rows on the lanes
//     layer  2   :  Y[row][cout]   = H[row][k]  . W^T[k][cout]      activations as the A operand, channels on the lanes
// The 32x32 accumulator layout (lane = row, half-wave h, register r <-> channel (r&3) + 8(r>>2) + 4h) IS the operand
// layout of the next MFMA step for K index pair (channel(r,0), channel(r,1)), and the packed weight fragment of K-block
// 4t + (r>>2), element r&3, holds exactly those two input channels — so a layer's output registers feed the next layer's
// MFMAs directly (after an in-register ReLU): no LDS round trip, no ds_write, no barrier after the weight load. The last
// layer is taken un-transposed so that the max over the 32 neighbours is a max over REGISTERS again, not over lanes.
// Layer 0 is two MFMA steps on (dx, dy) / (dz, 1) with its shift in the fourth K slot; layer 1's shift is one extra
// step against the constant (1, 0); layer 2's shift is added after the pool.
// (The earlier form kept a private [32][68] LDS tile per wave: ~250 epilogue instructions per 200 MFMAs, each paid in
//  matrix time beside the co-resident wave's MFMA stream — 0.71 of peak.)
// ------------------------------------------------------------------------------------------
#ifndef PTT_SAL_WGS
#define PTT_SAL_WGS 1        // workgroups per CU
#endif
#ifndef PTT_SAL_WAVES
#define PTT_SAL_WAVES 12     // waves per workgroup (three per SIMD: 148 VGPRs)
#endif
__global__ __launch_bounds__(64 * PTT_SAL_WAVES) __attribute__((amdgpu_waves_per_eu(PTT_SAL_WAVES * PTT_SAL_WGS / 4)))
void sa_lds_kernel(SaParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* W1 = smem;                                        // [8][2][256]   64 x 64
    float* W2 = smem + 4096;                                 // [8][4][256]   64 x 128
    const int t = threadIdx.x, lane = t & 63, half = lane >> 5, col = lane & 31;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    {
        const f32x4* g1 = reinterpret_cast<const f32x4*>(p.L[1].Wp);
        const f32x4* g2 = reinterpret_cast<const f32x4*>(p.L[2].Wp);
        f32x4* s1 = reinterpret_cast<f32x4*>(W1);
        f32x4* s2 = reinterpret_cast<f32x4*>(W2);
        const int nthr = (int)blockDim.x;                    // 12 waves for full launches, 4 for a handful of frames (see the host)
        for (int i = t; i < 1024; i += nthr) s1[i] = g1[i];
        for (int i = t; i < 2048; i += nthr) s2[i] = g2[i];
    }
    // per-lane constants, parked in LDS (10 registers less: four waves per SIMD fit without a spill). Layer 0's weights
    // and shift as A operands: lane (channel c, half h) supplies W0[c][h] for step 0 and (h ? shift0[c] : W0[c][2]) for
    // step 1; layer 1's shift for its extra step; layer 2's for after the pool
    float* cst = smem + 4096 + 8192 + w * 640 + lane;        // [wave][10][64]
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const f32x4 f = reinterpret_cast<const f32x4*>(p.L[0].Wp)[u * 64 + col];     // (w_x, w_y, w_z, 0) of channel 32u + col
        const float s0 = p.L[0].shift ? p.L[0].shift[u * 32 + col] : 0.f;
        cst[(0 + u) * 64] = half ? f[1] : f[0];
        cst[(2 + u) * 64] = half ? s0 : f[2];
        cst[(4 + u) * 64] = p.L[1].shift ? p.L[1].shift[u * 32 + col] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) cst[(6 + u) * 64] = p.L[2].shift ? p.L[2].shift[u * 32 + col] : 0.f;
    const float one_zero = half ? 0.f : 1.f, one_hi = 1.f;
    __syncthreads();
    const int total = p.B * p.M, M = p.M, N = p.N;           // one tile per centre
    const int gw = logical_block() * ((int)blockDim.x >> 6) + w;
    const int c0 = gw * p.chunk, c1 = min(total, c0 + p.chunk);
    if (c0 >= c1) return;
    const float rdiv = p.normalize ? p.radius : 1.0f;        // x / 1 is exact: one code path
    const int osb = (int)p.osb, osm = (int)p.osm, osc = (int)p.osc;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float* a1p = W1 + lane * 4;
    const float* b2p = W2 + lane * 4;
    const __amdgpu_buffer_rsrc_t rx = weight_rsrc(p.xyz);

    SasCentre ce;
    ce.c = c0; ce.b = c0 / M; ce.m = c0 - ce.b * M;
    // every lane: grouped row `col` of the tile (both half-waves hold the same rows and supply different K indices)
    int n_cur = p.idx[(size_t)c0 * 32 + col];
    float px, py, pz, cx, cy, cz;
    {
        const size_t flat = (size_t)ce.b * N + n_cur;
        px = p.xyz[flat * 3]; py = p.xyz[flat * 3 + 1]; pz = p.xyz[flat * 3 + 2];
        cx = p.new_xyz[(size_t)c0 * 3]; cy = p.new_xyz[(size_t)c0 * 3 + 1]; cz = p.new_xyz[(size_t)c0 * 3 + 2];
    }
    // Loads are pinned where their latency is covered (sched_barrier: left alone, the scheduler sinks the next tile's
    // index load to its first use and waits vmcnt(0) in the middle of layer 2, and issues each weight fragment's
    // ds_read right in front of the MFMAs that consume it)
    for (int c = c0; c < c1; ++c) {
        SasCentre nx = ce;                                   // the last tile re-requests itself (branch-free loop body)
        if (c + 1 < c1) nx.advance(1, M);
        const int n_next = p.idx[(size_t)nx.c * 32 + col];   // in flight under layers 0 and 1
        f32x4 wa[2], wn[2];                                  // layer 1's weight fragments: current / next K-block
#pragma unroll
        for (int u = 0; u < 2; ++u) wa[u] = *reinterpret_cast<const f32x4*>(a1p + u * 256);
        const float dx = (px - cx) / rdiv, dy = (py - cy) / rdiv, dz = (pz - cz) / rdiv;
        __builtin_amdgcn_sched_barrier(0);
        // ---- layer 0: 3 -> 64 (transposed), K = (dx, dy | dz, 1) ----
        f32x16 h0[2];
        {
            const float k0 = half ? dy : dx, k1 = half ? one_hi : dz;
#pragma unroll
            for (int u = 0; u < 2; ++u) h0[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(cst[(0 + u) * 64], k0, zero16, 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 2; ++u) h0[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(cst[(2 + u) * 64], k1, h0[u], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r) h0[u][r] = fmaxf(h0[u][r], 0.f);
        }
        // ---- layer 1: 64 -> 64 (transposed) ----
        f32x16 h1[2];
        f32x4 wb[4], wm[4];                                  // layer 2's weight fragments
        {
#pragma unroll
            for (int u = 0; u < 2; ++u) h1[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(cst[(4 + u) * 64], one_zero, zero16, 0, 0, 0);
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) {
                if (kb < 7) {
#pragma unroll
                    for (int u = 0; u < 2; ++u) wn[u] = *reinterpret_cast<const f32x4*>(a1p + ((kb + 1) * 2 + u) * 256);
                } else {
#pragma unroll
                    for (int u = 0; u < 4; ++u) wb[u] = *reinterpret_cast<const f32x4*>(b2p + u * 256);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int u = 0; u < 2; ++u)
                        h1[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[u][j], h0[kb >> 2][(kb & 3) * 4 + j], h1[u], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 2; ++u) wa[u] = wn[u];
            }
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r) h1[u][r] = fmaxf(h1[u][r], 0.f);
        }
        // the next tile's coordinates: in flight under layer 2
        {   // 32-bit offsets on a buffer descriptor: with 64-bit index arithmetic the sign extension of n_next is hoisted
            // to the load and the wave waits vmcnt(0) at the top of the tile
            const int off = (nx.b * N + n_next) * 12;
            px = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, off, 0, 0));
            py = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, off + 4, 0, 0));
            pz = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, off + 8, 0, 0));
            cx = p.new_xyz[(size_t)nx.c * 3]; cy = p.new_xyz[(size_t)nx.c * 3 + 1]; cz = p.new_xyz[(size_t)nx.c * 3 + 2];
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- layer 2: 64 -> 128 (rows back on the M axis), max over the 32 neighbours, shift, ReLU after the pool ----
        {
            f32x16 acc[4];
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) {
                if (kb < 7) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) wm[u] = *reinterpret_cast<const f32x4*>(b2p + ((kb + 1) * 4 + u) * 256);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(h1[kb >> 2][(kb & 3) * 4 + j], wb[u][j],
                                                                       (kb == 0 && j == 0) ? zero16 : acc[u], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 4; ++u) wb[u] = wm[u];
            }
            float* o = p.out + (ce.b * osb + ce.m * osm);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float mx = sas_pool(acc[u], cst[(6 + u) * 64], p.L[2].relu);
                if (half == 0) o[(u * 32 + col) * osc] = mx;
            }
        }
        ce = nx;
    }
}

// RT row tiles (32 * RT grouped rows) per wave; only RT = 1 is launched (see the host entry point).
template <int NS, int RT>
__global__ __launch_bounds__(256, 2) void sa_wave_kernel(SaParams p) {   // 2 waves per SIMD: at 3 the 168-VGPR cap spilled 92 B per lane
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int CPW = 32 * RT / NS;                        // centres per wave
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    float* Xw = smem + w * 32 * RT * p.ldk;                  // this wave's private [32*RT][ldk] tile
    const int total_centres = p.B * p.M;
    const int centre0 = (logical_block() * 4 + w) * CPW;
    if (centre0 >= total_centres) return;
    const int ncentres = min(CPW, total_centres - centre0);

#pragma unroll
    for (int rt = 0; rt < RT; ++rt) sa_gather_rows<NS, 32>(p, Xw, rt * 32, centre0, lane);
    __builtin_amdgcn_wave_barrier();

    for (int l = 0; l < p.n_layers; ++l) {
        const SaLayerDev& L = p.L[l];
        const bool last = (l == p.n_layers - 1);
#define PTT_WAVE_LAYER(CTV)                                                                         \
        { if (L.scale) sa_wave_layer<NS, CTV, true, RT>(p, L, last, Xw, lane, centre0, ncentres);       \
          else sa_wave_layer<NS, CTV, false, RT>(p, L, last, Xw, lane, centre0, ncentres); }
        if (L.NT == 1) PTT_WAVE_LAYER(1)
        else if (L.NT == 2) PTT_WAVE_LAYER(2)
        else PTT_WAVE_LAYER(4)
#undef PTT_WAVE_LAYER
    }
}

// ------------------------------------------------------------------------------------------
// Point-Transformer pair kernel. A workgroup owns 2 points x 16 neighbours = 32 pair rows
// and all D = 512 channels; wave w owns column tiles w, w+4, w+8, w+12.
// ------------------------------------------------------------------------------------------
struct AttnParams {
    const float* xyz; const float* rel; const int32_t* knn; const float* qkv; const float* Wd1p;
    const float* Wd2p; const float* bd2; const float* Wg1p; const float* bg1; const float* Wg2p; const float* bg2;
    float* res; float* attn;
    const int32_t* order;   // optional: the flat point tile slot s works on (a permutation inside every cloud), NULL = s
    int BN, N, first_wave, stagger;
    long long* dbg;   // dev only: per-phase s_memtime stamps (PTT_DEBUG_STAMPS), NULL in production
};

template <int D>
__global__ __launch_bounds__(256, 2) void pt_attn_pair_kernel(AttnParams p) {
    constexpr int KNN = 16, NT = D / 32, CT = NT / 4, LDK = D + 4, NKB = D / 8;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;                                       // [32][LDK]
    int* nb = reinterpret_cast<int*>(smem + 32 * LDK);      // [32] flat neighbour row (b*N + n)
    constexpr int LDR = 12;                                 // [rel.x rel.y rel.z 1 | 0 0 0 0] + pad (stride = 4 mod 8)
    float* relt = smem + 32 * LDK + 32;                     // [32][LDR]: the A operand of fc_delta[0]
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, half = lane >> 5;
    const int slot0 = logical_block() * 2;                   // the tile's two point slots
    const int npts = min(2, p.BN - slot0);
    // the points behind the slots (ptt_spatial_order_f32: neighbours in space next to each other in launch order); both lie in
    // the same cloud (N is even, the order permutes inside clouds)
    const int s0 = slot0 < p.BN ? slot0 : p.BN - 1, s1 = slot0 + 1 < p.BN ? slot0 + 1 : p.BN - 1;
    const int pt0 = p.order ? p.order[s0] : s0, pt1 = p.order ? p.order[s1] : s1;
    f32x4 pre[CT];
    prefetch_first_block_full<CT>(p.Wd1p, w, lane, pre);    // fc_delta[0]'s only weight block: requested first
    stagger_second_slot(p.first_wave, p.stagger);
    PTT_STAMP(0);

    if (t < 32) {
        const int pt = (t >> 4) ? pt1 : pt0;
        const int b = pt / p.N;
        const int n = p.knn[(size_t)pt * KNN + (t & 15)];
        const int flat = b * p.N + n;
        nb[t] = n * (3 * D * (int)sizeof(float));      // byte offset of the neighbour's q|k|v row inside its cloud
        f32x4 r4;
        if (p.rel) {                                   // precomputed by the kNN kernel: no index -> xyz dependency
            const float* rl = p.rel + ((size_t)pt * KNN + (t & 15)) * 3;
            r4 = f32x4{rl[0], rl[1], rl[2], 1.f};
        } else {
            r4 = f32x4{p.xyz[(size_t)pt * 3 + 0] - p.xyz[(size_t)flat * 3 + 0],
                       p.xyz[(size_t)pt * 3 + 1] - p.xyz[(size_t)flat * 3 + 1],
                       p.xyz[(size_t)pt * 3 + 2] - p.xyz[(size_t)flat * 3 + 2], 1.f};
        }
        *reinterpret_cast<f32x4*>(relt + t * LDR) = r4;
        *reinterpret_cast<f32x4*>(relt + t * LDR + 4) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    lds_barrier();

    int cols[CT];
#pragma unroll
    for (int u = 0; u < CT; ++u) cols[u] = (w + 4 * u) * 32 + (lane & 31);

    // fc_delta[0] + ReLU: h = relu([rel 1] . [W | b]^T) as ONE K-block of MFMAs (K = 4, zero-padded to 8) instead of
    // ~500 vector-ALU instructions per wave — next to the other workgroup's MFMA stream those crawl (DESIGN.md lesson 8)
    {
        f32x16 h[1][CT];
        zero_acc(h);
        gemm_core<1, CT, CT, 4, 1>(relt, LDR, 1, reinterpret_cast<const f32x4*>(p.Wd1p), NT, w, lane, h, pre);
        prefetch_first_block_full<CT>(p.Wd2p, w, lane, pre);    // fc_delta[2]'s first weight block
#pragma unroll
        for (int u = 0; u < CT; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) Xs[tile_row(r, half) * LDK + cols[u]] = fmaxf(h[0][u][r], 0.f);
    }
    lds_barrier();

    PTT_STAMP(1);
    // ---- delta = fc_delta[2](h) ----
    f32x16 delta[1][CT];
    zero_acc(delta);
    gemm_core<1, CT, CT, 4, PTT_PAIR_PF>(Xs, LDK, NKB, reinterpret_cast<const f32x4*>(p.Wd2p), NT, w, lane, delta, pre);
    prefetch_first_block_full<CT>(p.Wg1p, w, lane, pre);    // next GEMM's first block: in flight across the epilogue
#pragma unroll
    for (int u = 0; u < CT; ++u) {
        const float bb = p.bd2[cols[u]];
#pragma unroll
        for (int r = 0; r < 16; ++r) delta[0][u][r] += bb;
    }
    PTT_STAMP(2);
    // Gathers of neighbour k / v rows: raw buffer loads on a descriptor based at the cloud's first q|k|v row. The
    // per-(row, lane) byte offset is ONE 32-bit VGPR per tile row; channel group and the k / v column block are
    // immediates or an SGPR — a flat 64-bit address per load costs 3-4 vector-ALU instructions, 64 loads per phase.
    const int cloud = pt0 / p.N;
    const __amdgpu_buffer_rsrc_t rq = weight_rsrc(p.qkv + (size_t)cloud * p.N * 3 * D);
    int nrow[16];  // byte offset of (neighbour row, this lane's first column) for each of this lane's 16 tile rows
#pragma unroll
    for (int r = 0; r < 16; ++r) nrow[r] = nb[tile_row(r, half)] + (w * 32 + (lane & 31)) * (int)sizeof(float);

    lds_barrier();  // all waves done with h
    // t = (q_i - k_j) + delta  -> X
    {
        const int pa = pt0, pb = (npts > 1) ? pt1 : pt0;
#pragma unroll
        for (int u = 0; u < CT; ++u) {
            const float qa = p.qkv[(size_t)pa * 3 * D + cols[u]];
            const float qb = p.qkv[(size_t)pb * 3 * D + cols[u]];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float kv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                    rq, nrow[r] + (D + u * 128) * (int)sizeof(float), 0, 0));
                const float q = (r < 8) ? qa : qb;
                Xs[tile_row(r, half) * LDK + cols[u]] = (q - kv) + delta[0][u][r];
            }
        }
    }
    lds_barrier();

    PTT_STAMP(3);
    // ---- g = relu(fc_gamma[0](t)) -> X ----
    {
        f32x16 acc[1][CT];
        zero_acc(acc);
        gemm_core<1, CT, CT, 4, PTT_PAIR_PF>(Xs, LDK, NKB, reinterpret_cast<const f32x4*>(p.Wg1p), NT, w, lane, acc, pre);
        prefetch_first_block_full<CT>(p.Wg2p, w, lane, pre);
        PTT_STAMP(4);
        lds_barrier();
#pragma unroll
        for (int u = 0; u < CT; ++u) {
            const float bb = p.bg1[cols[u]];
#pragma unroll
            for (int r = 0; r < 16; ++r) Xs[tile_row(r, half) * LDK + cols[u]] = fmaxf(acc[0][u][r] + bb, 0.f);
        }
        lds_barrier();
    }

    // ---- a = fc_gamma[2](g); softmax over the 16 neighbours; res = sum attn * (v + delta) ----
    f32x16 acc[1][CT];
    zero_acc(acc);
    PTT_STAMP(5);
    gemm_core<1, CT, CT, 4, PTT_PAIR_PF>(Xs, LDK, NKB, reinterpret_cast<const f32x4*>(p.Wg2p), NT, w, lane, acc, pre);
    PTT_STAMP(6);
    // softmax_j((a_j + b) / sqrt(D)) over the 16 neighbours of a point: the bias b is the same for every neighbour, so
    // it cancels (fc_gamma[2].bias is never read); 1/sqrt(D) and log2(e) are one constant inside exp2; the weighted sum
    // is normalised once at the end. Fewer vector-ALU instructions next to the other workgroup's MFMA stream.
    const float kexp = 1.4426950408889634f / sqrtf((float)D);
    // all 64 neighbour values of this lane are requested before any softmax arithmetic: one L2 round trip
    // instead of eight (the gathers, not the math, were the length of this phase)
    float vv[CT][16];
#pragma unroll
    for (int u = 0; u < CT; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            vv[u][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                rq, nrow[r] + u * 128 * (int)sizeof(float), 2 * D * (int)sizeof(float), 0));
#pragma unroll
    for (int u = 0; u < CT; ++u) {
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
            float s[8];
            float m = acc[0][u][pp * 8];
#pragma unroll
            for (int r = 1; r < 8; ++r) m = fmaxf(m, acc[0][u][pp * 8 + r]);
            m = max_halves(m);
            float sum = 0.f, o = 0.f;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int rr = pp * 8 + r;
                s[r] = __builtin_amdgcn_exp2f((acc[0][u][rr] - m) * kexp);
                sum += s[r];
                o += s[r] * (vv[u][rr] + delta[0][u][rr]);
            }
            sum = add_halves(sum);
            o = add_halves(o);
            const float rsum = __builtin_amdgcn_rcpf(sum);
            if (p.attn && pp < npts) {
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int row = tile_row(pp * 8 + r, half);  // = pp*16 + j
                    p.attn[((size_t)(pp ? pt1 : pt0) * KNN + (row & 15)) * D + cols[u]] = s[r] * rsum;
                }
            }
            if (half == 0 && pp < npts) p.res[(size_t)(pp ? pt1 : pt0) * D + cols[u]] = o * rsum;
        }
    }
    PTT_STAMP(7);
}

// softmax(scale * x) along each row of a (rows, n) matrix, in place: one wave per row (the N x N scores of the dense
// attention variant, n <= a few thousand). Row maximum and sum over the wave with DPP reductions.
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ X, long long rows, int n, int ld, float scale) {
    const int lane = threadIdx.x & 63;
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    float* x = X + r * ld;
    float m = -3.0e38f;
    for (int c = lane; c < n; c += 64) m = fmaxf(m, x[c] * scale);
    m = wave_max_f32(m);
    float sum = 0.f;
    for (int c = lane; c < n; c += 64) {
        const float e = __expf(x[c] * scale - m);
        x[c] = e;
        sum += e;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    const float inv = 1.0f / sum;
    for (int c = lane; c < n; c += 64) x[c] *= inv;
}

}  // namespace ptt

using namespace ptt;

extern "C" int ptt_softmax_rows_f32(float* X, int64_t rows, int n, int ld, float scale, ptt_stream_t stream) {
    if (rows < 0 || n <= 0 || ld < n) return fail(PTT_EINVAL, "ptt_softmax_rows_f32: rows=%lld n=%d ld=%d", (long long)rows, n, ld);
    if (rows == 0) return PTT_OK;
    if (!X) return fail(PTT_EINVAL, "ptt_softmax_rows_f32: null pointer");
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, as_stream(stream), X, (long long)rows, n, ld,
                       scale);
    return check_launch("softmax_rows_kernel");
}

extern "C" size_t ptt_packed_weight_elems(int Cout, int K) {
    if (Cout <= 0 || K <= 0) return 0;
    const size_t NT = (size_t)(Cout + 31) / 32, NKB = (size_t)(K + 7) / 8;
    return NT * NKB * 256;
}

extern "C" int ptt_pack_weight_rot_f32(const float* W, int Cout, int K, int rot, float* packed, ptt_stream_t stream) {
    if (Cout <= 0 || K <= 0 || rot < 0 || rot >= K)
        return fail(PTT_EINVAL, "ptt_pack_weight_rot_f32: Cout=%d K=%d rot=%d", Cout, K, rot);
    if (!W || !packed) return fail(PTT_EINVAL, "ptt_pack_weight_rot_f32: null pointer");
    const size_t total = ptt_packed_weight_elems(Cout, K);
    const int NT = (Cout + 31) / 32;
    size_t g = (total + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)g), dim3(256), 0, as_stream(stream), W, Cout, K, NT, rot, total,
                       packed, (long long)K, 1LL, 0LL);
    return check_launch("pack_weight_kernel");
}

extern "C" int ptt_pack_weight_strided_f32(const float* W, int Cout, int K, int64_t stride_out, int64_t stride_k, int batch,
                                           int64_t stride_batch, float* packed, ptt_stream_t stream) {
    if (Cout <= 0 || K <= 0 || batch < 0) return fail(PTT_EINVAL, "ptt_pack_weight_strided_f32: Cout=%d K=%d batch=%d", Cout, K, batch);
    if (batch == 0) return PTT_OK;
    if (!W || !packed) return fail(PTT_EINVAL, "ptt_pack_weight_strided_f32: null pointer");
    const size_t total = ptt_packed_weight_elems(Cout, K);
    const int NT = (Cout + 31) / 32;
    size_t g = (total + 255) / 256;
    if (g > 1024) g = 1024;
    hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)g, (unsigned)batch), dim3(256), 0, as_stream(stream), W, Cout, K, NT, 0,
                       total, packed, (long long)stride_out, (long long)stride_k, (long long)stride_batch);
    return check_launch("pack_weight_kernel");
}

extern "C" int ptt_pack_weights_f32(const ptt_pack_job* jobs_device, int n_jobs, float* arena, ptt_stream_t stream) {
    if (n_jobs < 0) return fail(PTT_EINVAL, "ptt_pack_weights_f32: n_jobs=%d", n_jobs);
    if (n_jobs == 0) return PTT_OK;
    if (!jobs_device || !arena) return fail(PTT_EINVAL, "ptt_pack_weights_f32: null pointer");
    if (n_jobs > 65535) return fail(PTT_EUNSUPPORTED, "ptt_pack_weights_f32: n_jobs=%d exceeds one launch (65535)", n_jobs);
    hipLaunchKernelGGL(pack_weights_kernel, dim3(32, (unsigned)n_jobs), dim3(256), 0, as_stream(stream), jobs_device, arena);
    return check_launch("pack_weights_kernel");
}

extern "C" int ptt_pack_weight_f32(const float* W, int Cout, int K, float* packed, ptt_stream_t stream) {
    return ptt_pack_weight_rot_f32(W, Cout, K, 0, packed, stream);
}

static int linear_launch(const float* X, int rows, int K, int ldx, const float* Wpacked, int Cout, const float* scale,
                         const float* shift, int relu, const float* residual, int ldr, float* out, int ldo, int batch,
                         long long xb, long long wb, long long ob, long long rb, ptt_stream_t stream,
                         const float* in_a = nullptr, const float* in_b = nullptr);

extern "C" int ptt_linear_act_in_f32(const float* X, int rows, int K, int ldx, const float* in_scale, const float* in_shift,
                                     const float* Wpacked, int Cout, float* out, int ldo, ptt_stream_t stream) {
    if (!in_scale || !in_shift || (K & 3) || (ldx & 3) || ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(in_scale) |
                                                             reinterpret_cast<uintptr_t>(in_shift)) & 15))
        return fail(PTT_EINVAL, "ptt_linear_act_in_f32: needs K %% 4 == 0, ldx %% 4 == 0 and 16-byte aligned X / in_scale / in_shift");
    return linear_launch(X, rows, K, ldx, Wpacked, Cout, nullptr, nullptr, 0, nullptr, 0, out, ldo, 1, 0, 0, 0, 0, stream, in_scale,
                         in_shift);
}

extern "C" int ptt_linear_f32(const float* X, int rows, int K, int ldx, const float* Wpacked, int Cout,
                              const float* scale, const float* shift, int relu, const float* residual, int ldr,
                              float* out, int ldo, ptt_stream_t stream) {
    return linear_launch(X, rows, K, ldx, Wpacked, Cout, scale, shift, relu, residual, ldr, out, ldo, 1, 0, 0, 0, 0, stream);
}

extern "C" int ptt_linear_batched_f32(const float* X, int rows, int K, int ldx, int64_t x_batch_stride, const float* Wpacked,
                                      int64_t w_batch_stride, int Cout, const float* scale, const float* shift, int relu,
                                      const float* residual, int ldr, int64_t r_batch_stride, float* out, int ldo,
                                      int64_t o_batch_stride, int batch, ptt_stream_t stream) {
    if (batch < 0 || batch > 65535) return fail(PTT_EINVAL, "ptt_linear_batched_f32: batch=%d", batch);
    if (batch == 0) return PTT_OK;
    return linear_launch(X, rows, K, ldx, Wpacked, Cout, scale, shift, relu, residual, ldr, out, ldo, batch, x_batch_stride,
                         w_batch_stride, o_batch_stride, r_batch_stride, stream);
}

static int linear_launch(const float* X, int rows, int K, int ldx, const float* Wpacked, int Cout, const float* scale,
                         const float* shift, int relu, const float* residual, int ldr, float* out, int ldo, int batch,
                         long long xb, long long wb, long long ob, long long rb, ptt_stream_t stream, const float* in_a,
                         const float* in_b) {
    if (rows < 0 || K <= 0 || Cout <= 0 || ldx < K || ldo < Cout || (residual && ldr < Cout))
        return fail(PTT_EINVAL, "ptt_linear_f32: rows=%d K=%d Cout=%d ldx=%d ldo=%d ldr=%d", rows, K, Cout, ldx, ldo,
                    ldr);
    if (rows == 0) return PTT_OK;
    if (!X || !Wpacked || !out) return fail(PTT_EINVAL, "ptt_linear_f32: null pointer");
    LinearParams p;
    p.X = X; p.Wp = Wpacked; p.scale = scale; p.shift = shift; p.residual = residual; p.out = out;
    p.rows = rows; p.K = K; p.ldx = ldx; p.Cout = Cout; p.relu = relu; p.ldr = ldr; p.ldo = ldo;
    p.nkb = (K + 7) / 8; p.NT = (Cout + 31) / 32;
    p.vec_ok = ((ldx & 3) == 0 && (xb & 3) == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0) ? 1 : 0;
    p.xb = xb; p.wb = wb; p.ob = ob; p.rb = rb;
    p.in_a = in_a; p.in_b = in_b;
    // Tile choice (measured on all six GEMM shapes of the path, scripts/kernel_bench.py SWEEP_LINEAR=1): the smallest
    // tile, 32 rows x 128 columns, wins everywhere (qkv 93 vs 75 TFLOP/s for 64x256): these launches are only
    // 1-10 GFLOP, so workgroup count (>= 4 per CU, fine-grained tails) matters more than weight reuse per workgroup.
    const bool vec = p.vec_ok && (K & 3) == 0;
    hipStream_t s = as_stream(stream);
    if (dev_switches().linear_small && vec && batch == 1 && !in_a && rows <= 8192 && K <= 520 && K >= 32) {
        const int hb = ((p.nkb + 1) / 2 + LS_PD - 1) / LS_PD * LS_PD;
        const int lds = (32 * (2 * hb * 8 + 4) + 4 * 16 * 64) * (int)sizeof(float);
        if (int rc = set_lds_limit(reinterpret_cast<const void*>(linear_small_kernel), lds)) return rc;
        hipLaunchKernelGGL(linear_small_kernel, dim3((rows + 31) / 32, (p.NT + 3) / 4), dim3(512), lds, s, p);
        return check_launch("linear_small_kernel");
    }
    const int rt64 = (rows + 63) / 64, rt32 = (rows + 31) / 32, cg256 = (p.NT + 7) / 8, cg128 = (p.NT + 3) / 4;
    int RT = dev_switches().linear_rt, CT = dev_switches().linear_ct;
    // the row GEMMs of the training step (10^5-10^6 rows): 32 x 256 tiles reuse every A tile for twice the columns —
    // 93 vs 85 TFLOP/s at 393k x 128 -> 256, 106 vs 95 at 256 -> 256 (scripts/rows_gemm_bench.py); inference launches
    // (<= 24576 rows) keep the 32 x 128 tile that wins there
    if (rows >= 32768 && Cout >= 256 && RT == 1 && CT == 1) CT = 2;
    const int lds = 2 * (RT * 32) * LIN_LDK * (int)sizeof(float);
    const dim3 grid(RT == 2 ? rt64 : rt32, CT == 2 ? cg256 : cg128, batch);
    int rc = PTT_OK;
#define PTT_LIN_CASE(R, V, C)                                                                                  \
    if (RT == R && vec == V && CT == C) {                                                                      \
        if ((rc = set_lds_limit(reinterpret_cast<const void*>(linear_kernel<R, V, C>), lds))) return rc;       \
        hipLaunchKernelGGL((linear_kernel<R, V, C>), grid, dim3(256), lds, s, p);                              \
    }
    PTT_LIN_CASE(1, true, 1) PTT_LIN_CASE(1, true, 2) PTT_LIN_CASE(1, false, 1) PTT_LIN_CASE(1, false, 2)
#ifdef PTT_DEV      // 64-row tiles are a sweep option only (PTT_LINEAR_TILE=21|22): the 64 x 256 one spills 80 bytes per lane
    PTT_LIN_CASE(2, true, 1) PTT_LIN_CASE(2, true, 2) PTT_LIN_CASE(2, false, 1) PTT_LIN_CASE(2, false, 2)
#endif
#undef PTT_LIN_CASE
    return check_launch("linear_kernel");
}

extern "C" int ptt_rows_mlp_f32(const float* X, int rows, int K, int ldx, const ptt_sa_layer* layers, int n_layers,
                                const float* residual, int ldr, float* out, int ldo, ptt_stream_t stream) {
    if (rows < 0 || K <= 0 || ldx < K || n_layers < 1 || n_layers > PTT_SA_MAX_LAYERS || !layers)
        return fail(PTT_EINVAL, "ptt_rows_mlp_f32: rows=%d K=%d ldx=%d layers=%d", rows, K, ldx, n_layers);
    if (rows == 0) return PTT_OK;
    if (!X || !out) return fail(PTT_EINVAL, "ptt_rows_mlp_f32: null pointer");
    if (K > 264) return fail(PTT_EUNSUPPORTED, "ptt_rows_mlp_f32: K=%d (at most 264 input channels)", K);
    RowsMlpParams p;
    p.X = X; p.residual = residual; p.out = out; p.rows = rows; p.K0 = K; p.ldx = ldx; p.ldr = ldr; p.ldo = ldo;
    p.n_layers = n_layers;
    int maxk = 0, cin = K;
    for (int l = 0; l < n_layers; ++l) {
        const ptt_sa_layer& s = layers[l];
        const bool last = l == n_layers - 1;
        if (s.Cin != cin) return fail(PTT_EINVAL, "ptt_rows_mlp_f32: layer %d Cin=%d, expected %d", l, s.Cin, cin);
        if (s.Cout <= 0 || s.Cout > (last ? 384 : 256))
            return fail(PTT_EUNSUPPORTED, "ptt_rows_mlp_f32: layer %d Cout=%d (inner layers <= 256, the last <= 384)", l, s.Cout);
        if (!s.Wpacked) return fail(PTT_EINVAL, "ptt_rows_mlp_f32: layer %d has no weights", l);
        SaLayerDev& L = p.L[l];
        L.Wp = s.Wpacked; L.scale = s.scale; L.shift = s.shift; L.Cin = s.Cin; L.Cout = s.Cout; L.relu = s.relu;
        L.nkb = (s.Cin + 7) / 8; L.NT = (s.Cout + 31) / 32;
        if (L.nkb * 8 > maxk) maxk = L.nkb * 8;
        if (!last && L.NT * 32 > maxk) maxk = L.NT * 32;      // an inner layer writes whole (zero-padded) column tiles
        cin = s.Cout;
    }
    if (ldo < cin || (residual && ldr < cin)) return fail(PTT_EINVAL, "ptt_rows_mlp_f32: ldo=%d ldr=%d Cout=%d", ldo, ldr, cin);
    p.ldk = maxk + 4;
    p.vec_in = ((K & 3) == 0 && (ldx & 3) == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0) ? 1 : 0;
    const int lds = 32 * p.ldk * (int)sizeof(float);
    int rc = set_lds_limit(reinterpret_cast<const void*>(rows_mlp_kernel), lds);
    if (rc) return rc;
    hipLaunchKernelGGL(rows_mlp_kernel, dim3((rows + 31) / 32), dim3(256), lds, as_stream(stream), p);
    return check_launch("rows_mlp_kernel");
}

extern "C" int ptt_sa_fused_fwd_f32(const ptt_sa_desc* d, ptt_stream_t stream) {
    if (!d) return fail(PTT_EINVAL, "ptt_sa_fused_fwd_f32: null descriptor");
    if (d->B < 0 || d->N <= 0 || d->M < 0 || d->C < 0 || d->n_layers < 1 || d->n_layers > PTT_SA_MAX_LAYERS)
        return fail(PTT_EINVAL, "ptt_sa_fused_fwd_f32: B=%d N=%d M=%d C=%d layers=%d", d->B, d->N, d->M, d->C,
                    d->n_layers);
    if (d->nsample != 16 && d->nsample != 32 && d->nsample != 64)
        return fail(PTT_EUNSUPPORTED, "ptt_sa_fused_fwd_f32: nsample=%d (16, 32 and 64 are instantiated)", d->nsample);
    if (d->B == 0 || d->M == 0) return PTT_OK;
    const bool hoist = d->l0_point_term != nullptr;
    if (!d->xyz || !d->new_xyz || !d->idx || !d->out || (!hoist && d->C > 0 && !d->feat))
        return fail(PTT_EINVAL, "ptt_sa_fused_fwd_f32: null pointer");
    if (!d->use_xyz && d->C == 0) return fail(PTT_EINVAL, "ptt_sa_fused_fwd_f32: no xyz and no features");
    if (hoist) {
        const int c0 = d->l0_channels;
        if (!d->use_xyz || !d->l0_xyz_weight || c0 <= 0 || (c0 % 32) != 0 || c0 > 256 ||
            (reinterpret_cast<uintptr_t>(d->l0_point_term) & 15) != 0 ||
            (reinterpret_cast<uintptr_t>(d->l0_xyz_weight) & 15) != 0)
            return fail(PTT_EINVAL, "ptt_sa_fused_fwd_f32: hoisted layer 0 needs use_xyz, 16-byte aligned (B,N,C0) and "
                        "(3,C0) arrays and C0 a multiple of 32 <= 256 (C0=%d)", c0);
    }

    SaParams p;
    p.xyz = d->xyz; p.new_xyz = d->new_xyz; p.idx = d->idx; p.feat = d->feat; p.out = d->out;
    p.fsb = d->feat_sb; p.fsc = d->feat_sc; p.fsn = d->feat_sn;
    p.osb = d->out_sb; p.osc = d->out_sc; p.osm = d->out_sm;
    p.B = d->B; p.N = d->N; p.M = d->M; p.C = d->C; p.use_xyz = d->use_xyz ? 1 : 0;
    p.normalize = d->normalize_xyz ? 1 : 0; p.n_layers = d->n_layers; p.radius = d->radius;
    p.K0 = (d->use_xyz ? 3 : 0) + d->C;
    p.hoist = hoist ? 1 : 0; p.l0_relu = d->l0_relu ? 1 : 0; p.wx = d->l0_xyz_weight;
    if (hoist && d->l0_channels == 128 && (unsigned long long)d->B * d->N * 128ull * sizeof(float) < 0x7fffffffull &&
        !dev_switches().sa_gather1)
        p.hoist = 2;                                      // two rows per wave instruction, 32-bit buffer offsets
    if (hoist) {      // the tile starts as layer 0's output: rows of the per-point term, C0 channels
        p.feat = d->l0_point_term; p.C = d->l0_channels; p.K0 = p.C;
        p.fsc = 1; p.fsn = p.C; p.fsb = (long long)d->N * p.C;
    }
    int maxk = 0, cin = p.K0;
    for (int l = 0; l < d->n_layers; ++l) {
        const ptt_sa_layer& s = d->layers[l];
        if (s.Cin != cin) return fail(PTT_EINVAL, "ptt_sa_fused_fwd_f32: layer %d Cin=%d, expected %d", l, s.Cin, cin);
        if (s.Cout <= 0 || (s.Cout % 32) != 0 || s.Cout > 256)
            return fail(PTT_EUNSUPPORTED, "ptt_sa_fused_fwd_f32: layer %d Cout=%d must be a multiple of 32 <= 256", l,
                        s.Cout);
        if (!s.Wpacked) return fail(PTT_EINVAL, "ptt_sa_fused_fwd_f32: layer %d has no weights", l);
        SaLayerDev& L = p.L[l];
        L.Wp = s.Wpacked; L.scale = s.scale; L.shift = s.shift; L.Cin = s.Cin; L.Cout = s.Cout; L.relu = s.relu;
        L.nkb = (s.Cin + 7) / 8; L.NT = s.Cout / 32;
        if (L.nkb * 8 > maxk) maxk = L.nkb * 8;
        if (l + 1 < d->n_layers && s.Cout > maxk) maxk = s.Cout;
        cin = s.Cout;
    }
    {   // several kernels address these arrays with 32-bit byte offsets (buffer descriptors, scalar index arithmetic)
        const unsigned long long lim = 0x7fffffffull, B = (unsigned long long)d->B;
        if (B * d->N * 12ull >= lim || B * d->M * d->nsample * 4ull >= lim || B * d->M * (unsigned long long)cin * 4ull >= lim ||
            B * d->N * (unsigned long long)(p.C > 0 ? p.C : 1) * 4ull >= lim)
            return fail(PTT_EUNSUPPORTED, "ptt_sa_fused_fwd_f32: B=%d N=%d M=%d: an array of this launch exceeds 2 GiB "
                        "(32-bit offsets) — split the batch", d->B, d->N, d->M);
    }
    p.ldk = maxk + 4;
    p.vec_gather = (p.C > 0 && p.fsc == 1 && (p.C & 3) == 0 && p.C <= 256 && (p.fsn & 3) == 0 &&
                    (p.fsb & 3) == 0 && (reinterpret_cast<uintptr_t>(p.feat) & 15) == 0) ? 1 : 0;
    p.first_wave = 1024; p.stagger = dev_switches().sa_stagger;
    p.dbg = dev_switches().stamps;
    const int total_centres = d->B * d->M;
    hipStream_t s = as_stream(stream);
    int rc;
    // SA0 shape (no point features, 3 -> 64 -> 64 -> 128, 32 neighbours, BatchNorm scale folded): persistent workgroups
    // with the 50 KB of packed weights resident in LDS
    if (d->C == 0 && d->use_xyz && !hoist && d->nsample == 32 && d->n_layers == 3 && p.L[0].Cout == 64 && p.L[1].Cout == 64 &&
        p.L[2].Cout == 128 && !p.L[0].scale && !p.L[1].scale && !p.L[2].scale && p.L[0].relu && p.L[1].relu &&
        dev_switches().sa_lds) {
        const int waves = 256 * PTT_SAL_WAVES * PTT_SAL_WGS;   // resident waves of the device
        p.chunk = (total_centres + waves - 1) / waves;
        // short-lived workgroups (2 centres per wave) instead of one pass per wave: in the graphed step the CUs that also run
        // the next batch's FPS are slower, and dynamically scheduled workgroups go where the time is (3.18 -> 3.14 ms)
        if (dev_switches().sa_lds_chunk > 0 && p.chunk > dev_switches().sa_lds_chunk) p.chunk = dev_switches().sa_lds_chunk;
        // a handful of frames (one tracklet frame: 512 centres): with 12 waves per workgroup that is 43 workgroups whose SIMDs run
        // three centres' 198 MFMAs one after the other (16 us); four waves per workgroup put one centre on every SIMD of 128 CUs
        const int nw = total_centres * 3 <= 256 * PTT_SAL_WAVES ? 4 : PTT_SAL_WAVES;
        int wgs = (total_centres + p.chunk * nw - 1) / (p.chunk * nw);
        const int lds = (4096 + 8192 + PTT_SAL_WAVES * 640) * (int)sizeof(float);
        if ((rc = set_lds_limit(reinterpret_cast<const void*>(sa_lds_kernel), lds))) return rc;
        hipLaunchKernelGGL(sa_lds_kernel, dim3(wgs), dim3(64 * nw), lds, s, p);
        return check_launch("sa_lds_kernel");
    }
    // small weight set (fits L1/L2 comfortably) and <= 4 column tiles everywhere: barrier-free wave-private kernel
    size_t wbytes = 0;
    bool wave_ok = true;
    for (int l = 0; l < d->n_layers; ++l) {
        wbytes += (size_t)d->layers[l].Cin * d->layers[l].Cout * sizeof(float);
        const int nt = d->layers[l].Cout / 32;
        if (!(nt == 1 || nt == 2 || nt == 4)) wave_ok = false;
    }
    wave_ok = wave_ok && dev_switches().sa_wave;
    if (wave_ok && wbytes <= 64 * 1024 && d->nsample <= 32) {
        // one 32-row tile per wave; two (each weight fragment feeding two row tiles) was measured slower — 75 vs 98
        // TFLOP/s: the 128 accumulator registers of the 128-column layer spill and only two waves fit a SIMD
        const int lds = 4 * 32 * p.ldk * (int)sizeof(float);
        const int cpw = 32 / d->nsample, per_wg = 4 * cpw;
        const dim3 grid((total_centres + per_wg - 1) / per_wg);
        if (d->nsample == 32) {
            if ((rc = set_lds_limit(reinterpret_cast<const void*>(sa_wave_kernel<32, 1>), lds))) return rc;
            hipLaunchKernelGGL((sa_wave_kernel<32, 1>), grid, dim3(256), lds, s, p);
        } else {
            if ((rc = set_lds_limit(reinterpret_cast<const void*>(sa_wave_kernel<16, 1>), lds))) return rc;
            hipLaunchKernelGGL((sa_wave_kernel<16, 1>), grid, dim3(256), lds, s, p);
        }
        return check_launch("sa_wave_kernel");
    }
    // SA1 / SA2 shape (hoisted 128-channel layer 0, then 128 -> 128 -> 256 with the BatchNorm scale folded, 32
    // neighbours): persistent double-buffered kernel that gathers the next tile inside its own MFMA stream
    if (p.hoist == 2 && d->nsample == 32 && d->n_layers == 2 && p.L[0].Cin == 128 && p.L[0].Cout == 128 &&
        p.L[1].Cout == 256 && !p.L[0].scale && !p.L[1].scale && p.L[0].relu && dev_switches().sa_stream) {
        p.tiles = (total_centres + 1) / 2;
        int wgs = p.tiles < 512 ? p.tiles : 512;                 // 2 workgroups per CU x 256 CUs stay resident
        p.chunk = (p.tiles + wgs - 1) / wgs;
        if (dev_switches().sa_chunk > 0 && p.chunk > dev_switches().sa_chunk) p.chunk = dev_switches().sa_chunk;
        wgs = (p.tiles + p.chunk - 1) / p.chunk;
        const int lds = (2 * SAS_TILE + 4 * 16 * 4) * (int)sizeof(float);
        if ((rc = set_lds_limit(reinterpret_cast<const void*>(sa_stream_kernel<32>), lds))) return rc;
        hipLaunchKernelGGL((sa_stream_kernel<32>), dim3(wgs), dim3(256), lds, s, p);
        return check_launch("sa_stream_kernel");
    }
    int RT = dev_switches().sa_rt;                       // rows per workgroup = 32 * RT (2 unless a dev build says 1)
    if (d->nsample == 64) RT = 2;                        // one centre = two row tiles
    const int lds = 32 * RT * p.ldk * (int)sizeof(float);
    if (lds > 160 * 1024) return fail(PTT_EUNSUPPORTED, "ptt_sa_fused_fwd_f32: %d B of LDS per workgroup", lds);
    const int cpw = 32 * RT / d->nsample;
    const dim3 grid((total_centres + cpw - 1) / cpw);
#define PTT_SA_CASE(NSV, RTV)                                                                                   \
    if (d->nsample == NSV && RT == RTV) {                                                                       \
        if ((rc = set_lds_limit(reinterpret_cast<const void*>(sa_fused_kernel<NSV, RTV>), lds))) return rc;     \
        hipLaunchKernelGGL((sa_fused_kernel<NSV, RTV>), grid, dim3(256), lds, s, p);                            \
    }
    PTT_SA_CASE(32, 2) PTT_SA_CASE(32, 1) PTT_SA_CASE(16, 2) PTT_SA_CASE(16, 1) PTT_SA_CASE(64, 2)
#undef PTT_SA_CASE
    return check_launch("sa_fused_kernel");
}

extern "C" int ptt_xcorr_fused_fwd_f32(const ptt_xcorr_desc* d, ptt_stream_t stream) {
    if (!d) return fail(PTT_EINVAL, "ptt_xcorr_fused_fwd_f32: null descriptor");
    if (d->B < 0 || d->Ns <= 0 || d->C0 <= 0 || d->n_layers < 1 || d->n_layers > PTT_SA_MAX_LAYERS)
        return fail(PTT_EINVAL, "ptt_xcorr_fused_fwd_f32: B=%d Ns=%d C0=%d layers=%d", d->B, d->Ns, d->C0, d->n_layers);
    if (d->Nt <= 0 || (d->Nt % 64) != 0)
        return fail(PTT_EUNSUPPORTED, "ptt_xcorr_fused_fwd_f32: Nt=%d (the template seeds are walked in chunks of 64)", d->Nt);
    if (d->n_layers < 2) return fail(PTT_EUNSUPPORTED, "ptt_xcorr_fused_fwd_f32: needs at least two MFMA layers after layer 0");
    if ((d->C0 % 8) != 0 || d->C0 > 256) return fail(PTT_EUNSUPPORTED, "ptt_xcorr_fused_fwd_f32: C0=%d", d->C0);
    if (d->B == 0) return PTT_OK;
    if (!d->P || !d->w_sim || !d->out) return fail(PTT_EINVAL, "ptt_xcorr_fused_fwd_f32: null pointer");
    if ((unsigned long long)d->B * d->Ns * d->Nt >= 0x7fffffffull || (unsigned long long)d->B * d->Nt * d->C0 >= 0x7fffffffull)
        return fail(PTT_EUNSUPPORTED, "ptt_xcorr_fused_fwd_f32: B=%d Ns=%d Nt=%d: 32-bit element offsets — split the batch",
                    d->B, d->Ns, d->Nt);
    XcorrParams q;
    q.P = d->P; q.wsim = d->w_sim; q.scale0 = d->scale0;
    q.shift0 = d->shift0; q.sim_out = d->sim_out; q.cos_t = d->cos_t;
    q.C0 = d->C0; q.Nt = d->Nt;
    q.sfeat = q.tfeat = nullptr; q.s_sb = q.s_sn = q.t_sb = q.t_sn = 0; q.C = 0; q.eps = 0.f; q.out_sh = 0;
    SaParams& p = q.sa;
    memset(&p, 0, sizeof(p));
    p.out = d->out; p.osb = d->out_sb; p.osc = d->out_sc; p.osm = d->out_sn;
    p.B = d->B; p.M = d->Ns; p.n_layers = d->n_layers;
    int maxk = d->C0, cin = d->C0;
    for (int l = 0; l < d->n_layers; ++l) {
        const ptt_sa_layer& s = d->layers[l];
        if (s.Cin != cin) return fail(PTT_EINVAL, "ptt_xcorr_fused_fwd_f32: layer %d Cin=%d, expected %d", l, s.Cin, cin);
        if (s.Cout <= 0 || (s.Cout % 32) != 0 || s.Cout > 256)
            return fail(PTT_EUNSUPPORTED, "ptt_xcorr_fused_fwd_f32: layer %d Cout=%d", l, s.Cout);
        if (!s.Wpacked) return fail(PTT_EINVAL, "ptt_xcorr_fused_fwd_f32: layer %d has no weights", l);
        SaLayerDev& L = p.L[l];
        L.Wp = s.Wpacked; L.scale = s.scale; L.shift = s.shift; L.Cin = s.Cin; L.Cout = s.Cout; L.relu = s.relu;
        L.nkb = (s.Cin + 7) / 8; L.NT = s.Cout / 32;
        if (l + 1 < d->n_layers && s.Cout > maxk) maxk = s.Cout;
        cin = s.Cout;
    }
    p.ldk = ((maxk + 7) / 8) * 8 + 4;
    p.first_wave = 1024; p.stagger = 2;
    if (d->split) {
        // two workgroups per search point, cosines in the kernel: see xcorr_fused_kernel<1, true>
        if (!d->search_feat || !d->templ_feat || d->sim_out || (d->C & 3) || (d->s_sb & 3) || (d->s_sn & 3) ||
            (d->t_sb & 3) || (d->t_sn & 3) || ((reinterpret_cast<uintptr_t>(d->search_feat) | reinterpret_cast<uintptr_t>(d->templ_feat)) & 15) ||
            ((d->B * d->Ns) & 7) || d->C <= 0)
            return fail(PTT_EINVAL, "ptt_xcorr_fused_fwd_f32: the split form needs the features (unit channel stride, C %% 4 == 0, 16-byte "
                                    "aligned rows), B * Ns %% 8 == 0 and no sim_out");
        q.sfeat = d->search_feat; q.tfeat = d->templ_feat; q.s_sb = d->s_sb; q.s_sn = d->s_sn; q.t_sb = d->t_sb; q.t_sn = d->t_sn;
        q.C = d->C; q.eps = d->eps; q.out_sh = d->out_sh;
        const int lds = (32 * p.ldk + 32) * (int)sizeof(float);
        int rc = set_lds_limit(reinterpret_cast<const void*>(xcorr_fused_kernel<1, true>), lds);
        if (rc) return rc;
        hipLaunchKernelGGL((xcorr_fused_kernel<1, true>), dim3(2 * d->B * d->Ns), dim3(256), lds, as_stream(stream), q);
        return check_launch("xcorr_fused_kernel");
    }
    if (!d->cos_t) return fail(PTT_EINVAL, "ptt_xcorr_fused_fwd_f32: null pointer (cos_t comes from ptt_cosine_map_f32)");
    const int lds = (64 * p.ldk + 64) * (int)sizeof(float);
    int rc = set_lds_limit(reinterpret_cast<const void*>(xcorr_fused_kernel<2, false>), lds);
    if (rc) return rc;
    hipLaunchKernelGGL((xcorr_fused_kernel<2, false>), dim3(d->B * d->Ns), dim3(256), lds, as_stream(stream), q);
    return check_launch("xcorr_fused_kernel");
}

extern "C" int ptt_cosine_map_f32(const float* search_feat, int64_t s_sb, int64_t s_sn, int64_t s_sc, const float* templ_feat,
                                  int64_t t_sb, int64_t t_sn, int64_t t_sc, int B, int Ns, int Nt, int C, float eps,
                                  float* cos_t, ptt_stream_t stream) {
    if (B < 0 || Ns <= 0 || Nt <= 0 || C <= 0) return fail(PTT_EINVAL, "ptt_cosine_map_f32: B=%d Ns=%d Nt=%d C=%d", B, Ns, Nt, C);
    if (B == 0) return PTT_OK;
    if (!search_feat || !templ_feat || !cos_t) return fail(PTT_EINVAL, "ptt_cosine_map_f32: null pointer");
    CosParams q;
    q.sfeat = search_feat; q.tfeat = templ_feat; q.cos_t = cos_t;
    q.s_sb = s_sb; q.s_sn = s_sn; q.s_sc = s_sc; q.t_sb = t_sb; q.t_sn = t_sn; q.t_sc = t_sc;
    q.B = B; q.Ns = Ns; q.Nt = Nt; q.C = C; q.eps = eps;
    hipLaunchKernelGGL(cos_map_kernel, dim3((B * Ns + 3) / 4), dim3(256), 0, as_stream(stream), q);
    return check_launch("cos_map_kernel");
}

extern "C" int ptt_pt_attn_pair_f32(const ptt_attn_desc* d, ptt_stream_t stream) {
    if (!d) return fail(PTT_EINVAL, "ptt_pt_attn_pair_f32: null descriptor");
    if (d->B < 0 || d->N <= 0) return fail(PTT_EINVAL, "ptt_pt_attn_pair_f32: B=%d N=%d", d->B, d->N);
    if (d->D != 512 || d->k != 16)
        return fail(PTT_EUNSUPPORTED, "ptt_pt_attn_pair_f32: D=%d k=%d (D=512, k=16 is instantiated)", d->D, d->k);
    if (d->B == 0) return PTT_OK;
    if (!d->xyz || !d->knn || !d->qkv || !d->Wd1p || !d->Wd2p || !d->bd2 || !d->Wg1p || !d->bg1 ||
        !d->Wg2p || !d->res)
        return fail(PTT_EINVAL, "ptt_pt_attn_pair_f32: null pointer");
    AttnParams p;
    p.xyz = d->xyz; p.rel = d->rel; p.knn = d->knn; p.qkv = d->qkv; p.Wd1p = d->Wd1p; p.Wd2p = d->Wd2p; p.bd2 = d->bd2;
    p.Wg1p = d->Wg1p; p.bg1 = d->bg1; p.Wg2p = d->Wg2p; p.bg2 = d->bg2; p.res = d->res; p.attn = d->attn;
    p.order = d->order;
    p.BN = d->B * d->N; p.N = d->N;
    p.first_wave = 512; p.stagger = dev_switches().pair_stagger;
    p.dbg = dev_switches().stamps;
    if ((d->N & 1) != 0)
        return fail(PTT_EUNSUPPORTED, "ptt_pt_attn_pair_f32: N=%d must be even (a tile holds two points of one cloud)",
                    d->N);
    int lds = (32 * (512 + 4) + 32 + 32 * 12) * (int)sizeof(float);
    lds += dev_switches().pair_lds_pad;                               // dev: force 1 workgroup per CU
    int rc = set_lds_limit(reinterpret_cast<const void*>(pt_attn_pair_kernel<512>), lds);
    if (rc) return rc;
    hipLaunchKernelGGL((pt_attn_pair_kernel<512>), dim3((p.BN + 1) / 2), dim3(256), lds, as_stream(stream), p);
    return check_launch("pt_attn_pair_kernel");
}
