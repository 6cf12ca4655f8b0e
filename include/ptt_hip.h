/*
 * ptt_hip.h — C ABI of libptt_hip.so, the MI355X (gfx950) implementation of PTT's
 * per-frame point-feature hot path.
 *
 * Every entry point replaces one call the reference makes into the third-party CUDA
 * extension `pointnet2_ops._ext` (reference: ptt/models/backbones_3d/pointnet2/
 * pointnet2_utils.py:24) or one pure-PyTorch module forward on the hot path; the
 * reference call site each one stands in for is cited beside it.
 *
 * Conventions
 *   - plain C: raw DEVICE pointers, sizes, a hipStream_t passed as void*.
 *   - every function returns 0 (PTT_OK) or a negative PTT_E* code; nothing throws,
 *     nothing allocates, nothing synchronises the host. Kernels are enqueued on
 *     `stream` and the call returns immediately.
 *   - caller owns all buffers (outputs and workspaces); layouts are row-major and
 *     contiguous unless a stride argument says otherwise.
 *   - thread-safe and re-entrant: no global state beyond a thread-local last-error
 *     string.
 *   - clouds are (B, N, 3) fp32 "xyz"; indices are int32 as in the reference
 *     (pointnet2_utils.py:58-85, 265-294).
 */
#ifndef PTT_HIP_H
#define PTT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PTT_ABI_VERSION 21

enum {
    PTT_OK = 0,
    PTT_EINVAL = -1,        /* bad size / null pointer / inconsistent descriptor   */
    PTT_EUNSUPPORTED = -2,  /* shape outside what the kernels are instantiated for */
    PTT_ELAUNCH = -3,       /* hipLaunch / runtime error, see ptt_last_error_string */
    PTT_EWORKSPACE = -4     /* workspace too small                                  */
};

typedef void* ptt_stream_t; /* hipStream_t */

int ptt_version(void);
const char* ptt_error_name(int code);
const char* ptt_last_error_string(void);

/* ---------------------------------------------------------------------------------
 * F1  furthest point sampling
 * replaces _ext.furthest_point_sampling(xyz, npoint)        pointnet2_utils.py:78
 *   xyz  (B,N,3) f32   ->  idx_out (B,npoint) i32
 * idx[0]=0; running min-dist starts at 1e10; points with x*x+y*y+z*z <= 1e-3 are
 * neither updated nor selectable; d=(dx*dx+dy*dy)+dz*dz in fp32 without FMA
 * contraction; arg-max ties -> lowest index; no candidate -> 0.
 * ------------------------------------------------------------------------------- */
int ptt_fps_f32(const float* xyz, int B, int N, int npoint, int32_t* idx_out,
                ptt_stream_t stream);
/* The same call for clouds of any size (the reference's op has no limit; ptt_fps_f32 answers PTT_EUNSUPPORTED for
 * N > 16384 or npoint > 15360): the running min-distance in `workspace` (>= B*N floats), identical picks, slower. */
int ptt_fps_ws_f32(const float* xyz, int B, int N, int npoint, int32_t* idx_out, float* workspace,
                   size_t workspace_elems, ptt_stream_t stream);

/* ---------------------------------------------------------------------------------
 * F2  gather centres
 * replaces _ext.gather_points(features, idx)                pointnet2_utils.py:112
 *          _ext.gather_points_grad(grad_out, idx, N)        pointnet2_utils.py:118
 *   feat (B,C,N) f32, idx (B,M) i32 -> out (B,C,M);   out[b,c,j] = feat[b,c,idx[b,j]]
 *   grad: grad_feat (B,C,N) is zero-filled by the call, then += over all j.
 * ------------------------------------------------------------------------------- */
int ptt_gather_f32(const float* feat, const int32_t* idx, int B, int C, int N, int M,
                   float* out, ptt_stream_t stream);
int ptt_gather_grad_f32(const float* grad_out, const int32_t* idx, int B, int C, int N,
                        int M, float* grad_feat, ptt_stream_t stream);

/* Centres of one SA level in one launch (the xyz gather + transposes of pointnet2_modules.py:79-81 and
 * the int64 cast of :90): new_xyz[b,m,:] = xyz[b, idx[b,m], :]; idx == NULL selects the first M points
 * ('sequence' sampling, :70-71); idx64_out (B,M) receives the indices as int64, or NULL. */
int ptt_select_centres_f32(const float* xyz, const int32_t* idx, int B, int N, int M, float* new_xyz,
                           int64_t* idx64_out, ptt_stream_t stream);

/* ---------------------------------------------------------------------------------
 * F3  ball query  (centres first, as the reference calls it)
 * replaces _ext.ball_query(new_xyz, xyz, radius, nsample)   pointnet2_utils.py:287
 *   new_xyz (B,M,3), xyz (B,N,3) -> idx_out (B,M,nsample) i32
 * first `nsample` indices k in ascending order with (dx*dx+dy*dy)+dz*dz < r*r
 * (fp32, strict); unfilled slots repeat the first hit; no hit -> zeros.
 * ------------------------------------------------------------------------------- */
int ptt_ball_query_f32(const float* new_xyz, const float* xyz, int B, int M, int N,
                       float radius, int nsample, int32_t* idx_out, ptt_stream_t stream);

/* ptt_select_centres_f32 + ptt_ball_query_f32 of one SA level in one launch (same results): new_xyz (B,M,3), idx64_out
 * (B,M) or NULL, idx_out (B,M,nsample); sel (B,M) int32 sample indices or NULL for the first M points. */
int ptt_centres_ball_query_f32(const float* xyz, const int32_t* sel, int B, int N, int M, float radius, int nsample,
                               float* new_xyz, int64_t* idx64_out, int32_t* idx_out, ptt_stream_t stream);

/* ---------------------------------------------------------------------------------
 * F4  grouping
 * replaces _ext.group_points(features, idx)                 pointnet2_utils.py:237
 *          _ext.group_points_grad(grad_out, idx, N)         pointnet2_utils.py:257
 *   feat (B,C,N), idx (B,M,ns) -> out (B,C,M,ns);  out[b,c,j,k] = feat[b,c,idx[b,j,k]]
 * ------------------------------------------------------------------------------- */
int ptt_group_f32(const float* feat, const int32_t* idx, int B, int C, int N, int M,
                  int ns, float* out, ptt_stream_t stream);
int ptt_group_grad_f32(const float* grad_out, const int32_t* idx, int B, int C, int N,
                       int M, int ns, float* grad_feat, ptt_stream_t stream);

/* ---------------------------------------------------------------------------------
 * N3  deterministic scatter-add: the backward of gather_points (E = M) and group_points (E = M*nsample)
 * replaces _ext.gather_points_grad / _ext.group_points_grad   pointnet2_utils.py:118,257
 *   out[b,c,n] = sum of src[b,c,e] over the entries e with idx[b,e] == n, added in ascending e, i.e. the
 *   result of the sequential loop — bit-identical run to run (upstream's atomicAdd kernels are not).
 *   idx (B,E) i32 in [0,N); src (B,C,E); out (B,C,N); E <= 16384.
 *   workspace: ptt_scatter_add_det_workspace(B,N,E) bytes of device memory (PTT_EWORKSPACE if smaller).
 * ------------------------------------------------------------------------------- */
size_t ptt_scatter_add_det_workspace(int B, int N, int E);
int ptt_scatter_add_det_f32(const float* src, const int32_t* idx, int B, int C, int N, int E, float* out,
                            void* workspace, size_t workspace_bytes, ptt_stream_t stream);

/* ---------------------------------------------------------------------------------
 * T1  k nearest neighbours inside one cloud
 * replaces square_distance(xyz, xyz).argsort()[:, :, :k]    transformer_block/variants.py:150-151
 *   xyz (B,N,3) -> idx_out (B,N,k) i32, ascending by (squared distance, index);
 *   d = (dx*dx+dy*dy)+dz*dz in fp32 (model_utils/layer_utils.py:26). Requires k <= N.
 * ------------------------------------------------------------------------------- */
int ptt_knn_f32(const float* xyz, int B, int N, int k, int32_t* idx_out,
                ptt_stream_t stream);
/* Same, additionally writing rel_out (B,N,k,3) = xyz_i - xyz_neighbour, the input of fc_delta
 * (variants.py:158 `xyz[:, :, None] - knn_xyz`): the pair kernel then starts without a dependent
 * index -> coordinate round trip. rel_out may be NULL. */
int ptt_knn_rel_f32(const float* xyz, int B, int N, int k, int32_t* idx_out, float* rel_out,
                    ptt_stream_t stream);

/* ---------------------------------------------------------------------------------
 * Weight packing for the fp32-MFMA kernels.
 *   W (Cout,K) row-major (an nn.Linear / 1x1-conv weight)  ->  packed fragment order
 *   [K8][Cout32][64 lanes][4]  with K padded to a multiple of 8 and Cout to 32 (zeros).
 * ------------------------------------------------------------------------------- */
size_t ptt_packed_weight_elems(int Cout, int K);
int ptt_pack_weight_f32(const float* W, int Cout, int K, float* packed,
                        ptt_stream_t stream);
/* Same, with the input channels rotated: packed position k holds original channel (k+rot) mod K.
 * The fused SA kernel keeps each grouped row as [features | xyz] (16-byte aligned feature rows),
 * while the reference concatenates [xyz | features] (pointnet2_utils.py:359-361): pack the FIRST
 * SharedMLP layer of ptt_sa_fused_fwd_f32 with rot = 3 when use_xyz, rot = 0 otherwise. */
int ptt_pack_weight_rot_f32(const float* W, int Cout, int K, int rot, float* packed,
                            ptt_stream_t stream);

/* ---------------------------------------------------------------------------------
 * Row-wise linear layer on fp32 MFMA:  out = act(X @ W^T * scale + shift) (+ residual)
 * replaces nn.Linear / Conv1d(k=1) forwards on the hot path:
 *   TransformerBlock.fc1 / w_qs / w_ks / w_vs / fc2          variants.py:154-156,164
 *   PointNet2BackboneLight.cov_final                         pointnet2_backbone.py:46
 *   X (rows, K) with row stride ldx; out (rows, Cout) with row stride ldo;
 *   scale/shift per output channel (NULL = 1 / 0); residual (rows, Cout) stride ldr or NULL.
 * ------------------------------------------------------------------------------- */
int ptt_linear_f32(const float* X, int rows, int K, int ldx, const float* Wpacked, int Cout,
                   const float* scale, const float* shift, int relu, const float* residual,
                   int ldr, float* out, int ldo, ptt_stream_t stream);

/* ---------------------------------------------------------------------------------
 * F4+F5+F6+F7 fused: group -> (xyz - centre)/radius -> concat -> SharedMLP(eval BN
 * folded to scale/shift, ReLU) -> max over nsample.
 * replaces QueryAndGroup.forward                           pointnet2_utils.py:320-380
 *          SharedMLP forward (eval)                        pytorch_utils.py:12-36
 *          F.max_pool2d over nsample                       pointnet2_modules.py:85-88
 * Grouped tensors never reach HBM.
 * ------------------------------------------------------------------------------- */
#define PTT_SA_MAX_LAYERS 4

typedef struct ptt_sa_layer {
    const float* Wpacked; /* packed (Cout,Cin) conv weight; layer 0: rot = 3 if use_xyz      */
    const float* scale;   /* (Cout) gamma/sqrt(var+eps), or NULL = 1                   */
    const float* shift;   /* (Cout) beta - mean*scale (or conv bias), or NULL = 0      */
    int Cin;              /* input channels of this layer (first layer: 3*use_xyz + C) */
    int Cout;             /* multiple of 32                                            */
    int relu;
} ptt_sa_layer;

/* A stack of up to PTT_SA_MAX_LAYERS 1x1 convolutions (+ folded BatchNorm + ReLU) over point rows in one launch:
 * out = L_n(... L_1(X)) (+ residual). Replaces the eval-mode Conv1d stacks of the heads
 * (voting_heads/centroids_voting_head.py: vote_layer, cla_layer; voting_heads/box_voting_head.py: refine_layer) and the
 * two trailing convolutions of CosineSimAug (similarity_modules/p2b_xcoor.py:43-45), which the reference runs as one
 * cuDNN convolution + batch_norm + relu per layer. X (rows, K <= 264) with row stride ldx; inner layers Cout <= 256, the
 * last layer any Cout <= 384 (e.g. 1, 5, 259); layers[i].Wpacked from ptt_pack_weight_f32, scale NULL when the BatchNorm
 * scale is folded into the weights (one instruction less per value in the epilogue). */
int ptt_rows_mlp_f32(const float* X, int rows, int K, int ldx, const ptt_sa_layer* layers, int n_layers,
                     const float* residual, int ldr, float* out, int ldo, ptt_stream_t stream);

/* Point jobs (round 4): the centre selections + ball queries of several set-abstraction levels and one kNN in ONE launch — the
 * index ops between the level-0 furthest point sampling and the first SharedMLP of one tracklet frame
 * (pointnet2_modules.py:68-83 for three levels, transformer_block/variants.py:150-151). Every job reads the RAW clouds
 * xyz (B,Nraw,3): centre m of cloud b is raw[centre_sel[b][m]] (NULL: raw[m]), point k is raw[point_sel[b][k]] (NULL:
 * raw[k]), k < Npts; with 'sequence' sampling above level 0 both selections of every level are prefixes of the level-0
 * sample indices, so the levels do not depend on each other.
 *   kind 0  ball query: new_xyz (B,M,3), idx64_out (B,M) = the centres' raw indices or NULL, idx_out (B,M,nsample) = positions
 *           k inside the level's points — the results of ptt_centres_ball_query_f32 on the level's own point tensor;
 *   kind 1  kNN of the Npts <= 128 points among themselves (M == Npts): idx_out (B,Npts,nsample), rel_out (B,Npts,nsample,3)
 *           or NULL — the results of ptt_knn_rel_f32 on the points' own tensor. */
/* F3 for large clouds (round 5): the same results as ptt_ball_query_f32 / ptt_centres_ball_query_f32 through a uniform grid —
 * the cloud's points binned into cells of edge >= 1.001 radius, a centre tests the points of its 27 neighbouring cells only, the
 * first nsample hits IN INDEX ORDER restored by a bitmap (bit-identical to the sweep). workspace: ptt_ball_query_grid_workspace(B, N)
 * bytes, 16-byte aligned (PTT_EWORKSPACE if smaller); N <= 131072. Worth it from a few thousand points per cloud on. */
size_t ptt_ball_query_grid_workspace(int B, int N);
int ptt_ball_query_grid_f32(const float* new_xyz, const float* xyz, int B, int M, int N, float radius, int nsample,
                            int32_t* idx_out, void* workspace, size_t workspace_bytes, ptt_stream_t stream);
int ptt_centres_ball_query_grid_f32(const float* xyz, const int32_t* sel, int B, int N, int M, float radius, int nsample,
                                    float* new_xyz, int64_t* idx64_out, int32_t* idx_out, void* workspace, size_t workspace_bytes,
                                    ptt_stream_t stream);
/* One launch for a small FPS-sampled set-abstraction level whose centres then meet in a TransformerBlock (vote_aggregation +
 * the box head's transformer at one tracklet frame, box_voting_head.py:75-86): ptt_fps_f32 + ptt_centres_ball_query_f32 +
 * ptt_knn_rel_f32 (of the CENTRES among themselves) with identical results. xyz (B,N<=256,3) -> inds (B,M) i32, inds64 (B,M)
 * or NULL, new_xyz (B,M,3), idx (B,M,nsample), knn (B,M,k) + rel (B,M,k,3) (k = 0: neither). M <= 128. */
int ptt_fps_ball_knn_f32(const float* xyz, int B, int N, int M, float radius, int nsample, int k, int32_t* inds,
                         int64_t* inds64, float* new_xyz, int32_t* idx, int32_t* knn, float* rel, ptt_stream_t stream);
#define PTT_POINT_JOBS_MAX 4
typedef struct ptt_point_job {
    const float* xyz; const int32_t* centre_sel; const int32_t* point_sel;
    float* new_xyz; int64_t* idx64_out; int32_t* idx_out; float* rel_out;
    int32_t kind, sel_ld, B, Nraw, Npts, M, nsample;
    float radius;
} ptt_point_job;
int ptt_point_jobs_f32(const ptt_point_job* jobs, int n_jobs, ptt_stream_t stream);

/* ---------------------------------------------------------------------------------
 * Row jobs (round 4): up to PTT_ROW_JOBS_MAX independent row-wise layers in ONE launch, each
 *     out = act(A @ W^T * scale + shift) (+ residual)
 * with the operand A formed while it is staged and the result consumed in the epilogue — the launch
 * chain of ONE tracklet frame (tools/eval_utils/eval_tracking_utils.py:140-152: B = 1) is ~45 dependent
 * launches of 128-2048 rows, each a few microseconds of work behind ~5 us of launch latency, so what
 * counts is how many launches are on the chain. One job replaces one of
 *   nn.Linear / Conv1d(k=1) forwards (transformer_block/variants.py:154-164; voting_heads/
 *     centroids_voting_head.py:83-94 cla_layer / vote_layer; box_voting_head.py:88-90 refine_layer;
 *     similarity_modules/p2b_xcoor.py:43-45; pointnet2_backbone.py:46),
 * and the element-wise code around them:
 *   prologue 0  A = [X | X2]: the columns [0,K1) from X (row stride ldx), [K1,K) from X2 (ldx2) — the
 *               torch.cat((xyz, feats)) in front of vote_layer (centroids_voting_head.py:86-90) and of
 *               CosineSimAug's per-template-point term without the copy (weights packed with the
 *               matching input-channel rotation);
 *   prologue 1  A[r,:] = relu(fc_delta[0](rel[r,:]))            variants.py:158 (first Linear + ReLU; w1 (K,4) =
 *               [wx wy wz bias] per channel), so that fc_delta is one launch;
 *   prologue 2  A[(i,j),:] = q_i - k[knn_ij] + pos_ij            variants.py:160 (argument of fc_gamma);
 *   epilogue 1  res_i = sum_j softmax_j(y_ij * sm_scale) * (v[knn_ij] + pos_ij)   variants.py:161-163 — the 32 rows
 *               of a tile are the 16 neighbours of two points, so the softmax over neighbours is local to the
 *               accumulator tile (the per-column bias cancels and is not read); out (points, Cout).
 *   prologue 3  A[(c,j),:] = act0(term[idx_cj,:] + Wx . (xyz[idx_cj] - centre_c)(/radius)): the grouped input of a
 *               set-abstraction level whose first convolution is hoisted to one row per POINT (ptt_sa_desc.l0_point_term;
 *               QueryAndGroup + SharedMLP layer 0, pointnet2_utils.py:320-361, pytorch_utils.py:12-36): X = term
 *               (B*N rows, ldx), idx (B,M,ns) from the ball query, wx (3,K), act0 = ReLU when pro_relu;
 *   epilogue 2  out[c,:] = act(max_j y_cj): the max-pool over the ns = 16 or 32 neighbours of a centre
 *               (F.max_pool2d, pointnet2_modules.py:85-87), local to the 32-row accumulator tile; out (B*M, Cout).
 *               Prologue 3 -> layer 1, then a plain layer with epilogue 2 = vote_aggregation (box_voting_head.py:75-79) in two
 *               launches of 1024 rows over the chip, where the fused SA kernel is 16 workgroups at one frame.
 *   act         0 none, 1 ReLU, 2 sigmoid (`raw`, if given, receives the value before the activation:
 *               pred_centroids_cls beside its sigmoid, centroids_voting_head.py:84,92-94)
 *   residual    out column c >= res_split adds res[row*ldr + c - res_split], c < res_split adds
 *               res2[row*ldr2 + c] (either may be NULL): `vote_in + vote_layer(vote_in)` with vote_in = cat(xyz, feats)
 *               (centroids_voting_head.py:90), `offsets[:, 0:3] + centres` (box_voting_head.py:91);
 *   output      column c >= out_split goes to out[row*ldo + c - out_split + out_col0], c < out_split to
 *               out2[row*ldo2 + c]: the slices / concatenations behind the heads (votes vs. votes_feats).
 * A workgroup of 8 waves owns a 32-row x (32 * col_tiles)-column tile and splits K over 8 / col_tiles wave groups
 * (partial sums meet in LDS): a 128-row layer is 64-192 workgroups with a 1-2 us MFMA chain each instead of 16 with 7 us.
 * Wpacked from ptt_pack_weight_f32 (K <= 1024). Jobs of one launch must not depend on each other.
 * ------------------------------------------------------------------------------- */
#define PTT_ROW_JOBS_MAX 4
typedef struct ptt_row_job {
    const float* X; const float* X2;
    const float* Xmax;                                 /* prologue 0, optional: A = max(X, Xmax) element-wise (same rows / stride as X; needs
                                                          K1 == K, K % 4 == 0, 16-byte aligned rows): the two halves of the split ptt_xcorr_fused_fwd_f32 */
    const float* Wpacked; const float* scale; const float* shift;
    const float* res; const float* res2;
    float* out; float* out2; float* raw;
    const float* rel; const float* w1;                 /* prologue 1: (rows,3), (K,4) */
    const float* qkv; const int32_t* knn; const float* pos;   /* prologue 2 / epilogue 1: (points, ldq) rows holding q | k | v at
                                                          column offsets q_off / k_off / v_off, (points,16) neighbour indices
                                                          inside the point's cloud of N points, (rows, ldp) positional term */
    int32_t rows, K, K1, ldx, ldx2, Cout;
    int32_t act, res_split, ldr, ldr2, out_split, out_col0, ldo, ldo2, ldraw;
    int32_t prologue, epilogue, ldq, q_off, k_off, v_off, ldp, N;
    float sm_scale;
    int32_t col_tiles;                                 /* 0: the library picks 1, 2 or 4 by launch size */
    const int32_t* idx; const float* xyz; const float* centres; const float* wx;   /* prologue 3 */
    float radius;
    int32_t ns, M, normalize_xyz, pro_relu;            /* prologue 3 / epilogue 2: neighbours per centre, centres per cloud */
} ptt_row_job;
int ptt_row_jobs_f32(const ptt_row_job* jobs, int n_jobs, ptt_stream_t stream);

typedef struct ptt_sa_desc {
    const float* xyz;     /* (B,N,3)                                                   */
    const float* new_xyz; /* (B,M,3) centres                                           */
    const int32_t* idx;   /* (B,M,nsample) from ptt_ball_query_f32                     */
    const float* feat;    /* point features or NULL when C == 0                        */
    int64_t feat_sb, feat_sc, feat_sn; /* element strides of feat[b][c][n]             */
    float* out;           /* pooled features                                           */
    int64_t out_sb, out_sc, out_sm;    /* element strides of out[b][c][m]              */
    int B, N, M, nsample, C;
    float radius;
    int use_xyz;          /* prepend the 3 relative coordinates (reference default)    */
    int normalize_xyz;    /* divide them by radius (pointnet2_utils.py:353-354)        */
    int n_layers;
    ptt_sa_layer layers[PTT_SA_MAX_LAYERS];
    /* Optional: layer 0 of the SharedMLP hoisted out of the grouped stage. The first Conv2d is linear in
     * [rel ; f_n], so  layer0(centre i, neighbour n) = act(T[b][n][:] + Wx . rel(i,n))  with
     *   T  = scale0 * (W0[:,3:] . f_n) + shift0     one row per POINT (ptt_linear_f32), (B,N,C0) contiguous
     *   Wx = (scale0 * W0[:, 0:3])^T                (3,C0)
     * which moves 2*C*C0 flop per grouped row (N*... rows instead of M*nsample) out of the kernel.
     * When l0_point_term != NULL: use_xyz must be 1, feat/C are ignored and `layers` holds the REMAINING layers
     * (layers[0].Cin == l0_channels). NULL = layer 0 runs inside the kernel on the gathered rows. */
    const float* l0_point_term;
    const float* l0_xyz_weight;
    int l0_channels;
    int l0_relu;
} ptt_sa_desc;

int ptt_sa_fused_fwd_f32(const ptt_sa_desc* d, ptt_stream_t stream);

/* ---------------------------------------------------------------------------------
 * N1  P2B cosine-similarity feature augmentation, fused
 * replaces CosineSimAug.forward up to the max over the template axis
 *                                              similarity_modules/p2b_xcoor.py:25-42
 *   fusion[b,:,i,j] = [cos(templ_i, search_j) | templ_xyz_i | templ_feat_i]  (never materialised)
 *   out[b,:,j] = max_i SharedMLP(fusion)[b,:,i,j]
 * Layer 0 is split: W0.fusion = w_sim*cos_ij + P[b,i,:], with P = W0[:,1:].[xyz_i;feat_i] computed
 * once per template point by ptt_linear_f32. `layers` are the REMAINING SharedMLP layers.
 * The template seeds are walked in chunks of 64 with a running max (Nt % 64 == 0); the cosine map comes from
 * ptt_cosine_map_f32 (below).
 * ------------------------------------------------------------------------------- */
typedef struct ptt_xcorr_desc {
    const float* cos_t;                 /* (B,Ns,Nt) cosine map cos(search_j, templ_i) from ptt_cosine_map_f32 (one launch
                                           per batch: computing it inside every workgroup cost ~340 vector-ALU / load
                                           instructions per wave beside the other workgroup's MFMA stream)             */
    const float* P;                     /* (B,Nt,C0) contiguous, see above                   */
    const float* w_sim;                 /* (C0) W0[:,0]                                      */
    const float* scale0;                /* (C0) folded BN of layer 0 (NULL = 1 / 0)          */
    const float* shift0;
    float* out;                         /* out[b][c][j], element strides below               */
    int64_t out_sb, out_sc, out_sn;
    float* sim_out;                     /* optional (B,Nt,Ns) copy of the cosine map in the reference's orientation, or NULL */
    int B, Ns, Nt, C0;
    int n_layers;
    ptt_sa_layer layers[PTT_SA_MAX_LAYERS];
    /* Split form, for a handful of frames (round 4; B * Ns search points would be as many workgroups: half the chip at one
     * frame): split != 0 launches TWO workgroups per search point, each with half of the template points, and forms the
     * cosines inside the kernel (cos_t is not read, ptt_cosine_map_f32 not needed). Half h writes its maxima to
     * out + h * out_sh (same strides); the maximum over the template axis is the element-wise maximum of the two, taken by
     * the consumer (ptt_row_job.Xmax). search_feat[b][j][:], templ_feat[b][i][:] with unit channel stride, element strides
     * s_sb / s_sn / t_sb / t_sn, C channels (C % 4 == 0), eps of F.cosine_similarity; B * Ns % 8 == 0; sim_out NULL. */
    int32_t split; int64_t out_sh;
    const float* search_feat; const float* templ_feat;
    int64_t s_sb, s_sn, t_sb, t_sn;
    int C; float eps;
} ptt_xcorr_desc;

int ptt_xcorr_fused_fwd_f32(const ptt_xcorr_desc* d, ptt_stream_t stream);

/* cos_t[b][j][i] = <templ_i, search_j> / (max(|templ_i|, eps) * max(|search_j|, eps))   (F.cosine_similarity, p2b_xcoor.py:35-36)
 * search_feat[b][n][c] / templ_feat[b][i][c] with element strides; cos_t (B,Ns,Nt) contiguous. */
int ptt_cosine_map_f32(const float* search_feat, int64_t s_sb, int64_t s_sn, int64_t s_sc, const float* templ_feat,
                       int64_t t_sb, int64_t t_sn, int64_t t_sc, int B, int Ns, int Nt, int C, float eps,
                       float* cos_t, ptt_stream_t stream);

/* ---------------------------------------------------------------------------------
 * T2..T6 fused per-(point,neighbour) part of the Point-Transformer block:
 *   delta = fc_delta(xyz_i - xyz_j);  a = fc_gamma(q_i - k_j + delta);
 *   attn = softmax_j(a / sqrt(D));    res_i = sum_j attn * (v_j + delta)
 * replaces TransformerBlock.forward lines                  variants.py:158-163
 *   qkv (B,N,3*D) = [q | k | v] rows (from ptt_linear_f32 with the stacked weight),
 *   knn (B,N,k) i32. D = 512 and k = 16 are the instantiated configuration.
 *   res (B,N,D); attn (B,N,k,D) or NULL (the heads discard it: centroids_voting_head.py:76).
 * ------------------------------------------------------------------------------- */
typedef struct ptt_attn_desc {
    const float* xyz;      /* (B,N,3) */
    const float* rel;      /* (B,N,k,3) from ptt_knn_rel_f32, or NULL (computed from xyz and knn) */
    const int32_t* knn;    /* (B,N,k) */
    const float* qkv;      /* (B,N,3*D) */
    const float* Wd1p;     /* packed (ptt_pack_weight_f32, K = 4) [fc_delta[0].weight (D,3) | fc_delta[0].bias (D)]:
                              layer 0 of the position encoding runs as one MFMA K-block on rows [rel.xyz 1]   */
    const float* Wd2p;     /* packed fc_delta[2].weight */
    const float* bd2;
    const float* Wg1p;     /* packed fc_gamma[0].weight */
    const float* bg1;
    const float* Wg2p;     /* packed fc_gamma[2].weight */
    const float* bg2;      /* fc_gamma[2].bias: accepted, never read — a per-channel constant over the
                              neighbours cancels in softmax_j                                               */
    float* res;            /* (B,N,D) */
    float* attn;           /* (B,N,k,D) or NULL */
    int B, N, k, D;
    const int32_t* order;  /* (B,N) from ptt_spatial_order_f32, or NULL: which flat point (b*N + n) launch slot s works on. A
                              permutation inside every cloud; results do not depend on it (L2 locality of the k | v gathers) */
} ptt_attn_desc;

int ptt_pt_attn_pair_f32(const ptt_attn_desc* d, ptt_stream_t stream);
/* order (B,N) i32: the points of every cloud along a Morton curve through the cloud's bounding box (flat indices b*N + n,
 * ties by index: a permutation inside each cloud), N <= 8192. Neighbours in space become neighbours in launch order. */
int ptt_spatial_order_f32(const float* xyz, int B, int N, int32_t* order, ptt_stream_t stream);

/* ---------------------------------------------------------------------------------
 * N4  device-side pre/post-processing of the sequential tracking loop
 * (tools/eval_utils/eval_tracking_utils.py:140-229,266-274). A tracklet's clouds stay resident in HBM; per frame
 * the host supplies the float64 box quantities of the previous result (a few dozen numbers) and receives one
 * (x, y, z, theta, score) row.
 *
 * ptt_crop_compact_f32 — one job = crop_center_pc (ptt/datasets/kitti/kitti_tracking_utils.py:300-339):
 *   keep points with lo1 < p < hi1 (crop_pc :281-298 in the cloud's frame; float32 point vs float64 bound, strict);
 *   p <- float32(p + trans) per axis (PointCloud.translate :44-46); p <- float32(rot . p) (PointCloud.rotate :48-49,
 *   float64 dot product); keep points with lo2 < p < hi2 (the second crop_pc, in the box frame). Survivors are
 *   written to `out` as (count, 3) float32 rows IN THEIR ORIGINAL ORDER; *count receives their number (rows beyond
 *   `capacity` are counted, not written). The job array lives in DEVICE memory (the caller uploads it).
 *   label_out != NULL (the training crop, crop_center_pc with gt_box, :308-312,322): label_out[r] = 1 where survivor r lies inside
 *   the ground-truth box — get_label_by_box (:238-272) evaluated on the first crop's points in the CLOUD's frame, i.e.
 *   p' = float32(lrot . float32(p + ltrans)), llo < p' < lhi with the ground-truth box's own translation / rotation / bounds —
 *   carried through the second crop exactly as crop_pc carries `label` (:295-296).
 * ------------------------------------------------------------------------------- */
typedef struct ptt_crop_job {
    const float* points;        /* (3, n_points) float32: row 0 = x, row 1 = y, row 2 = z (PointCloud.points) */
    int64_t ld;                 /* elements between rows */
    double lo1[3], hi1[3];
    double trans[3];
    double rot[9];              /* row-major 3x3 */
    double lo2[3], hi2[3];
    float* out;                 /* (capacity, 3) */
    int32_t* count;
    int32_t n_points;
    int32_t capacity;
    uint8_t* label_out;         /* (capacity) or NULL: the survivors' labels against the ground-truth box ... */
    double ltrans[3];           /* ... whose frame is p' = lrot . (p + ltrans), bounds llo < p' < lhi */
    double lrot[9];
    double llo[3], lhi[3];
} ptt_crop_job;

int ptt_crop_compact_f32(const ptt_crop_job* jobs_device, int n_jobs, ptt_stream_t stream);
/* The same crops with the job table in HOST memory, passed by value in the kernel arguments (n_jobs <=
 * PTT_CROP_JOBS_BY_VALUE_MAX): one tracklet's frame is two crops, and a per-frame host-to-device copy of the table in front
 * of the launch costs more than the launch. The table is read before the call returns. */
#define PTT_CROP_JOBS_BY_VALUE_MAX 8
int ptt_crop_compact_host_f32(const ptt_crop_job* jobs_host, int n_jobs, ptt_stream_t stream);

/* ptt_regularize_f32 — one job = regularize_pc(pc, input_size, istrain=False) (kitti_tracking_utils.py:342-367) on the
 * concatenation of up to PTT_MAX_SEGMENTS compacted crops (get_model :219-236 concatenates the crops of several
 * frames): n = total rows; n <= 2 -> all-zero cloud; n == input_size -> copied through; otherwise row i of `out` is
 * point idx[i], idx = np.random.randint(0, n, input_size) drawn right after set_manual_seed(1), i.e. the 32-bit
 * outputs of MT19937(seed 1) masked to the smallest 2^k - 1 >= n - 1 and skipped while > n - 1. `draws` (DEVICE) holds
 * the generator's first n_draws outputs: fill a host buffer with ptt_mt19937_fill(1, ...) and upload it once;
 * n_draws >= 4 * input_size + 1024 always suffices (acceptance probability > 1/2). info (2 x int32, may be NULL)
 * receives n and the number of generator outputs consumed (0 when nothing was drawn). */
#define PTT_MAX_SEGMENTS 4
typedef struct ptt_regularize_job {
    const float* seg[PTT_MAX_SEGMENTS];          /* (seg_count, 3) float32 each */
    const int32_t* seg_count[PTT_MAX_SEGMENTS];  /* device scalars written by ptt_crop_compact_f32 */
    int32_t seg_capacity[PTT_MAX_SEGMENTS];
    float* out;                                  /* (input_size, 3) */
    int32_t* info;
    int32_t n_seg;
    int32_t input_size;
} ptt_regularize_job;

int ptt_regularize_f32(const ptt_regularize_job* jobs_device, int n_jobs, const uint32_t* draws, int n_draws,
                       ptt_stream_t stream);
/* ptt_crop_compact_f32 + ptt_regularize_f32 of one frame in ONE launch: workgroup w runs crop job w and then resampling job w
 * (whose segments are that crop's output and, for a template, crops of earlier launches) — identical results. crop_jobs may
 * live in device memory or in PINNED host memory (read through unified addressing: the launch can sit in a hipGraph whose
 * table the host rewrites between replays); reg_jobs_device as for ptt_regularize_f32. */
int ptt_crop_regularize_f32(const ptt_crop_job* crop_jobs, const ptt_regularize_job* reg_jobs_device, int n_jobs,
                            const uint32_t* draws, int n_draws, ptt_stream_t stream);

/* First n outputs of MT19937 seeded with init_genrand(seed) (what np.random.seed(seed) followed by 32-bit draws
 * yields) into a HOST buffer — the one entry point that takes a host pointer. */
int ptt_mt19937_fill(uint32_t seed, uint32_t* out_host, int n);

/* Host-side float64 box arithmetic of the tracking loop (HOST pointers, no device work): what the reference does per
 * frame through `Box` / pyquaternion (kitti_tracking_utils.py:68-160,186-216,300-339), batched over n tracklets so
 * that a step of 48 interleaved tracklets costs microseconds of host time. Quaternions are (w, x, y, z).
 *
 * ptt_track_crop_bounds — the float64 quantities of crop_center_pc(pc, box, offset=offset, scale=scale) per box
 *   (lo1/hi1: crop_pc with 2*offset, 4*scale in the cloud's frame; trans = -center; rot = R^T; lo2/hi2: crop_pc of the
 *   carried-along box with offset + extra2[i] (extra2 NULL = 0; gt_box.wlh[1]*0.6 for the search crop, :321) and scale)
 *   written into jobs_host[i * job_stride] (the other fields are left untouched).
 * ptt_track_box_by_offset — boxes[i] <- get_box_by_offset(boxes[i], offsets[i*offset_stride .. +4), use_z) for every i
 *   with active[i] != 0 (active NULL = all). offsets are the model's float32 outputs (x, y, z, theta in degrees); the
 *   angle is formed as the float32 product the reference forms. An x / y offset larger than the box is redrawn from
 *   U(-1, 1) (:205-208) using the two next 32-bit outputs of MT19937(seed 1) at position rng_pos[i] (the state numpy's
 *   global generator has after regularize_pc's resampling consumed rng_pos[i] outputs); rng_pos[i] advances by 2 per
 *   draw; offsets are updated in place with the values used. */
typedef struct ptt_track_box {
    double center[3];
    double wlh[3];
    double quat[4];
} ptt_track_box;

int ptt_track_crop_bounds(const ptt_track_box* boxes, int n, double offset, double scale, const double* extra2,
                          ptt_crop_job* jobs_host, int job_stride);
int ptt_track_box_by_offset(ptt_track_box* boxes, int n, float* offsets, int offset_stride, int use_z,
                            const int32_t* active, int64_t* rng_pos);
/* The host side of post_process for one step in one call: per tracklet the first arg-max of the scores out of the (n,P,5) read-back
 * (P == 1: rows already selected) -> est_out (n,5); rng_pos[i] <- the draw count of this frame's resampling (info (n,2,2) int32 =
 * (points, draws used) of search / template: the template's if it resampled, else the search's; a negative count = the draw table
 * ran out: PTT_EINVAL); then ptt_track_box_by_offset(boxes, est_out, ...). */
int ptt_track_select_update(const float* proposals_host, int P, const int32_t* info_host, ptt_track_box* boxes, int n, int use_z,
                            const int32_t* active, int64_t* rng_pos, float* est_out);

/* ptt_select_box_f32 — post_process (eval_tracking_utils.py:266-274): for every frame b the row of
 * pred_box_data (B,P,5) with the largest score (column 4; first one among equals, as np.argmax) -> out (B,5);
 * idx_out (B) receives its index, or NULL. */
int ptt_select_box_f32(const float* pred_box_data, int B, int P, float* out, int32_t* idx_out, ptt_stream_t stream);

/* ---------------------------------------------------------------------------------
 * N3  training step of the shared-MLP stages on hand-written kernels. The reference's SharedMLP in train mode is
 * [1x1 conv (no bias) -> BatchNorm with batch statistics -> ReLU] x 3 followed by a max over the neighbour axis
 * (pytorch_utils.py:12-36,94-114; pointnet2_modules.py:84-88; similarity_modules/p2b_xcoor.py:39-41), and
 * loss.backward() (tools/train_utils/train_utils.py:47-48) runs their backward. Activations are (R, C) ROWS (one row
 * per (centre, neighbour) position, channels contiguous): the convolution and its input gradient are ptt_linear_f32,
 * the rest is below. Every reduction runs in a fixed order (float64 partials per 2048 / 4096-row chunk, combined in
 * chunk order): bit-reproducible.
 *
 * ptt_bn_stats_f32      per-channel batch statistics of X (R,C): mean, BIASED variance, invstd = 1/sqrt(var + eps)
 *                       (BatchNorm2d forward in train mode; the caller updates running_mean / running_var).
 * ptt_bn_apply_f32      X = relu?((Z - mean) * invstd * gamma + beta)
 * ptt_bn_bwd_f32        backward of BatchNorm(train) + ReLU: dy = G where Act > 0 else 0; dbeta = sum dy;
 *                       dgamma = sum dy * xhat; dZ = gamma * invstd * (dy - dbeta / R - xhat * dgamma / R).
 * ptt_pool_rows_f32     out[g,c] = max_k X[g*ns + k, c], arg = first arg-max   (F.max_pool2d over the neighbour axis)
 * ptt_pool_rows_bwd_f32 dX[g*ns + k, c] = dOut[g,c] if k == arg[g,c] else 0
 * ptt_linear_wgrad_f32  dW[o,i] (+)= sum_r dZ[r,o] * X[r,i] on fp32 MFMA (the weight gradient of a 1x1 convolution /
 *                       nn.Linear over R rows), split over 4096-row chunks.
 * ------------------------------------------------------------------------------- */
size_t ptt_bn_stats_workspace(int R, int C);
int ptt_bn_stats_f32(const float* X, int R, int C, int ldx, float eps, float* mean, float* var, float* invstd,
                     void* workspace, size_t workspace_bytes, ptt_stream_t stream);
int ptt_bn_apply_f32(const float* Z, int ldz, const float* mean, const float* invstd, const float* gamma,
                     const float* beta, int R, int C, int relu, float* X, int ldx, ptt_stream_t stream);
int ptt_bn_bwd_f32(const float* G, int ldg, const float* Act, int lda, const float* Z, int ldz, const float* mean,
                   const float* invstd, const float* gamma, int R, int C, int relu, float* dZ, int ldd,
                   float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, const float* act_scale,
                   const float* act_shift, ptt_stream_t stream);
/* SyncBatchNorm (tools/train_tracking.py:133-134, --sync_bn -> nn.SyncBatchNorm.convert_sync_batchnorm): the same
 * statistics split where the ranks exchange them. Forward: ptt_bn_sums_f64 -> all-reduce(sums, row count) ->
 * ptt_bn_finish_f64. Backward: ptt_bn_bwd_sums_f64 (sum dy, sum dy * xhat; a rank's dbeta / dgamma are its LOCAL sums,
 * as torch's SyncBatchNorm keeps them) -> all-reduce -> ptt_bn_bwd_apply_f32 with the global sums and the global count.
 * Forward sums = 2 * C + 1 doubles: sum, sum of squares, and in the last slot the row count R — so one all-reduce carries
 * everything and the global count never visits the host; ptt_bn_finish_f64 and ptt_bn_bwd_apply_f32 (count = a device
 * pointer to that slot) read it from device memory. Backward sums = 2 * C doubles. Fixed combination order; C % 4 == 0,
 * 16-byte aligned rows. */
int ptt_bn_sums_f64(const float* X, int R, int C, int ldx, double* sums, void* workspace, size_t workspace_bytes,
                    ptt_stream_t stream);
int ptt_bn_finish_f64(const double* sums, int C, float eps, float* mean, float* var, float* invstd,
                      ptt_stream_t stream);
int ptt_bn_bwd_sums_f64(const float* G, int ldg, const float* Act, int lda, const float* Z, int ldz, const float* mean,
                        const float* invstd, int R, int C, double* sums, void* workspace, size_t workspace_bytes,
                        const float* act_scale, const float* act_shift, ptt_stream_t stream);
int ptt_bn_bwd_apply_f32(const float* G, int ldg, const float* Act, int lda, const float* Z, int ldz, const float* mean,
                         const float* invstd, const float* gamma, const float* sum_dy, const float* sum_dy_xhat,
                         const double* count, int R, int C, float* dZ, int ldd, const float* act_scale, const float* act_shift,
                         ptt_stream_t stream);
int ptt_pool_rows_f32(const float* X, int ldx, int G, int ns, int C, float* out, int ldo, int32_t* arg,
                      const float* act_scale, const float* act_shift, ptt_stream_t stream);
int ptt_pool_rows_bwd_f32(const float* dOut, int ldo, const int32_t* arg, int G, int ns, int C, float* dX, int ldx,
                          ptt_stream_t stream);
/* Point-Transformer block in training mode (transformer_block/variants.py:156-163): the element-wise chains around its
 * GEMMs over the (B,N,k,D) per-(point, neighbour) tensors, one pass each (k = 16, D % 4 == 0; q / kf / vf are (B,N,D)
 * point rows, knn (B,N,k) int32, pos / a / t / attn / da / dvp (B,N,k,D)):
 *   ptt_pt_pair_input_f32      t = q_i - kf[knn_ij] + pos_ij                                  (argument of fc_gamma, :160)
 *   ptt_pt_attn_train_fwd_f32  attn = softmax_j(a * scale); res_i = sum_j attn_ij * (vf[knn_ij] + pos_ij)      (:161-163)
 *   ptt_pt_attn_train_bwd_f32  from dres: dvp_ij = attn_ij * dres_i (gradient of v[knn] and of pos),
 *                              da_ij = attn_ij * (dres_i.vp_ij - sum_j' attn_ij' dres_i.vp_ij') * scale          */
int ptt_pt_pair_input_f32(const float* q, const float* kf, const int32_t* knn, const float* pos, int B, int N, int k, int D,
                          float* t, ptt_stream_t stream);
int ptt_pt_attn_train_fwd_f32(const float* a, const float* vf, const int32_t* knn, const float* pos, int B, int N, int k, int D,
                              float scale, float* attn, float* res, ptt_stream_t stream);
int ptt_pt_attn_train_bwd_f32(const float* attn, const float* vf, const int32_t* knn, const float* pos, const float* dres, int B,
                              int N, int k, int D, float scale, float* da, float* dvp, ptt_stream_t stream);

/* The same two passes with row strides for q / k / v (column slices of the stacked (B,N,3D) projection) and an optional attention
 * output (attn may be NULL): the per-layer INFERENCE form of the block for a handful of frames, where the fused pair kernel's
 * one workgroup per two points leaves most CUs idle (one tracklet frame: 64 workgroups) — ptt_amd/models/transformer_block. */
int ptt_pt_pair_input_ld_f32(const float* q, int ldq, const float* kf, int ldk, const int32_t* knn, const float* pos, int B, int N,
                             int k, int D, float* t, ptt_stream_t stream);
int ptt_pt_attn_fwd_ld_f32(const float* a, const float* vf, int ldv, const int32_t* knn, const float* pos, int B, int N, int k, int D,
                           float scale, float* attn, float* res, ptt_stream_t stream);

/* Grouping of point-major rows and its deterministic backward (the training-mode layer-0 hoist: the first MLP layer's
 * feature half is evaluated once per point, then gathered per (centre, neighbour) row):
 *   ptt_gather_rows_f32       out[b,e,:] = src[b, idx[b,e], :]        src (B,N,C), idx (B,E) -> out (B,E,C); C % 4 == 0
 *   ptt_scatter_csr_i32       order (B,E) / start (B,N+1): the entries of every cloud sorted by (idx, e) — a stable counting sort
 *                             up to 2048 bins (any E), a bitonic network in LDS above (E <= 16384)
 *   ptt_scatter_rows_csr_f32  out[b,n,:] = sum of g[b,e,:] over idx[b,e] == n in ascending e (fixed order) */
int ptt_gather_rows_f32(const float* src, const int32_t* idx, int B, int N, int E, int C, float* out, ptt_stream_t stream);
int ptt_scatter_csr_i32(const int32_t* idx, int B, int N, int E, int32_t* order, int32_t* start, ptt_stream_t stream);
int ptt_scatter_rows_csr_f32(const float* g, const int32_t* order, const int32_t* start, int B, int N, int E, int C,
                             float* out, ptt_stream_t stream);
/* The last layer of a SharedMLP + max-pool stage in training (pytorch_utils.py:12-36 + F.max_pool2d, pointnet2_modules.py:84-88)
 * without the pooling pass over its (rows, N) output: ptt_rows_gemm_f32 with statistics whose epilogue ALSO takes, per group of
 * `ns` consecutive rows and per column, the largest and the smallest output with the first row (inside the group) holding it —
 * pmax / pmin (rows / ns, N) float, amax / amin (rows / ns, N) int32. Once the batch statistics (summed by the same launch) give
 * the BatchNorm's a = gamma * invstd and b = beta - mean * a, ptt_pool_select_f32 forms the pooled relu(a y + b) and its arg-max:
 * relu(a y + b) is monotone in y, the sign of a picks max or min. in_scale / in_shift (the producing layer's deferred activation)
 * are required; shapes: ptt_rows_gemm_pool_supported (K, N multiples of 128 with ns 16 / 32 / 64, or K % 64 == 0, N % 128 == 0 with ns 32). */
int ptt_rows_gemm_pool_supported(int rows, int K, int N, int ldx, int ns);
int ptt_rows_gemm_pool_f32(const float* X, int rows, int K, int ldx, const float* in_scale, const float* in_shift,
                           const float* Wpacked, int N, float* out, int ldo, double* stats, size_t stats_elems, int ns,
                           float* pmax, float* pmin, int32_t* amax, int32_t* amin, ptt_stream_t stream);
int ptt_pool_select_f32(const float* pmax, const float* pmin, const int32_t* amax, const int32_t* amin, const float* act_scale,
                        const float* act_shift, int G, int C, float* out, int32_t* arg, ptt_stream_t stream);
/* out[c] = sum over the R rows of X[r][c] — the bias gradient of a row-wise layer (nn.Linear / Conv1d(k=1) backward,
 * transformer_block/variants.py:154-164 in training) — in a fixed order: bit-reproducible. C % 4 == 0, 16-byte aligned rows. */
size_t ptt_colsum_workspace(int R, int C);
int ptt_colsum_f32(const float* X, int R, int C, int ldx, float* out, void* workspace, size_t workspace_bytes, ptt_stream_t stream);
size_t ptt_linear_wgrad_workspace(int R, int Cout, int Cin);
int ptt_linear_wgrad_f32(const float* dZ, int ldz, const float* X, int ldx, int R, int Cout, int Cin, float* dW,
                         int accumulate, void* workspace, size_t workspace_bytes, const float* x_scale, const float* x_shift,
                         ptt_stream_t stream);
/* Deferred activation (training): a layer's output relu(z * a[c] + b[c]) (a = gamma * invstd, b = beta - mean * a) is
 * never written; its consumers apply it while they load z — ptt_linear_act_in_f32 (the next convolution),
 * ptt_linear_wgrad_f32's x_scale / x_shift (that convolution's weight gradient), ptt_pool_rows_f32's act_scale /
 * act_shift (the max-pool), ptt_bn_bwd_f32 with Act == NULL (the ReLU mask is z * a + b > 0). NULL scales = no transform. */
int ptt_linear_act_in_f32(const float* X, int rows, int K, int ldx, const float* in_scale, const float* in_shift,
                          const float* Wpacked, int Cout, float* out, int ldo, ptt_stream_t stream);

/* ---------------------------------------------------------------------------------
 * T-opt  the dense scaled-dot-product variant TransformerBlockSTD (transformer_block/variants.py:29-40): the only
 * literal Q.K^T / attn.V of the reference, as batched fp32-MFMA GEMMs.
 *   ptt_pack_weight_strided_f32  packs `batch` matrices given by strides into MFMA B-fragment order: element
 *                                (output o, input k) of batch b at W[b*stride_batch + o*stride_out + k*stride_k].
 *                                K of a batch (rows of the q|k|v buffer) -> the B operand of Q.K^T; V + delta read
 *                                transposed (stride_out 1, stride_k = row stride) -> the B operand of attn.V.
 *   ptt_linear_batched_f32       ptt_linear_f32 over `batch` independent (X, Wpacked, out[, residual]) sets.
 *   ptt_softmax_rows_f32         in-place softmax(scale * x) along the rows of a (rows, n) matrix (`attn / sqrt(d)`, :35).
 * ------------------------------------------------------------------------------- */
int ptt_pack_weight_strided_f32(const float* W, int Cout, int K, int64_t stride_out, int64_t stride_k, int batch,
                                int64_t stride_batch, float* packed, ptt_stream_t stream);
int ptt_linear_batched_f32(const float* X, int rows, int K, int ldx, int64_t x_batch_stride, const float* Wpacked,
                           int64_t w_batch_stride, int Cout, const float* scale, const float* shift, int relu,
                           const float* residual, int ldr, int64_t r_batch_stride, float* out, int ldo,
                           int64_t o_batch_stride, int batch, ptt_stream_t stream);
int ptt_softmax_rows_f32(float* X, int64_t rows, int n, int ld, float scale, ptt_stream_t stream);

/* ---------------------------------------------------------------------------------
 * N3  the row GEMMs of the training step at 10^4 - 10^6 rows (round 3): a persistent, software-pipelined form of
 * ptt_linear_f32 / ptt_linear_wgrad_f32 for the forward, input-gradient and weight-gradient GEMMs of every 1x1
 * convolution and nn.Linear the training step runs over (centre, neighbour) / (point, neighbour) rows:
 *   SharedMLP convolutions and their backward      pytorch_utils.py:12-36 (loss.backward(): tools/train_utils/train_utils.py:47-48)
 *   TransformerBlock fc1, w_qs/w_ks/w_vs, fc_delta[2], fc_gamma[0], fc_gamma[2], fc2       transformer_block/variants.py:154-165
 *   the Conv1d stacks of the heads                 voting_heads/centroids_voting_head.py:15-21, box_voting_head.py:25
 * ptt_rows_gemm_f32:  out[r, :] = relu?( act_in(X[r, :]) . W^T + bias ) (+ residual[r, :]),  act_in(x) = relu(x * in_scale + in_shift)
 *   when in_scale / in_shift are given (the producing layer's BatchNorm + ReLU, never materialised), else x.
 *   Wpacked from ptt_pack_weight_f32 (Cout = N, K). Needs K % 64 == 0, N % 64 == 0, ldx % 4 == 0, 16-byte aligned X
 *   (ptt_rows_gemm_supported says so; otherwise use ptt_linear_f32).
 *   stats != NULL (then bias must be NULL): also the per-channel sums and sums of squares of `out` over the rows — the
 *   BatchNorm batch statistics of a convolution output without a second pass over it — as float64 partials
 *   stats[chunk][2][N], chunk < ptt_rows_gemm_stat_chunks(rows, K, N) (a pure function of the shape: the summation order
 *   is fixed); ptt_bn_finish_partials_f32 / ptt_bn_sums_partials_f64 combine them in chunk order.
 * ptt_linear_wgrad2_f32: ptt_linear_wgrad_f32 with up to 256 x 256 outputs per workgroup (every operand row read once);
 *   needs R >= 2048, Cout % 128 == 0, Cin % 128 == 0; same partial-sum scheme (fixed order), own workspace size. */
int ptt_rows_gemm_supported(int rows, int K, int N, int ldx, int ldo);
int ptt_rows_gemm_stat_chunks(int rows, int K, int N);
int ptt_rows_gemm_f32(const float* X, int rows, int K, int ldx, const float* in_scale, const float* in_shift,
                      const float* Wpacked, int N, const float* bias, int relu, const float* residual, int ldr,
                      float* out, int ldo, double* stats, size_t stats_elems, ptt_stream_t stream);
/* the input gradient of a Linear + ReLU layer's successor, masked by that ReLU: out = (mask > 0) ? X . W^T : 0, mask = the
 * ReLU's output (rows, N), row stride ldm; statistics (optional) are those of the MASKED output, whose column sums are
 * the bias gradient of the masked layer (nn.Sequential(Linear, ReLU, Linear): fc_delta / fc_gamma, variants.py:139-148). */
int ptt_rows_gemm_masked_f32(const float* X, int rows, int K, int ldx, const float* Wpacked, int N, const float* mask, int ldm,
                             float* out, int ldo, double* stats, size_t stats_elems, ptt_stream_t stream);
/* The input gradient g = dZ_next . W_next of a layer whose INPUT is relu(BatchNorm(Z)) (Z = the producing layer's convolution
 * output, deferred activation act_scale / act_shift): the GEMM's epilogue also takes that producing layer's BatchNorm backward
 * sums, sum dy and sum dy * xhat with dy = g where Z * act_scale + act_shift > 0, as float64 partials [chunk][2][N] — the pass
 * over (g, Z) that ptt_bn_bwd_f32 starts with disappears. ptt_bn_bwd_from_partials_f32 combines the partials (dbeta, dgamma) and
 * writes dZ; ptt_bn_bwd_sums_partials_f64 is the SyncBatchNorm form of the combine (all-reduce, then ptt_bn_bwd_apply_f32). */
int ptt_rows_gemm_bnbwd_f32(const float* X, int rows, int K, int ldx, const float* Wpacked, int N, const float* Z, int ldz,
                            const float* mean, const float* invstd, const float* act_scale, const float* act_shift, float* out,
                            int ldo, double* sums_partial, size_t partial_elems, ptt_stream_t stream);
int ptt_bn_bwd_from_partials_f32(const double* partial, int chunks, const float* G, int ldg, const float* Z, int ldz, const float* mean,
                                 const float* invstd, const float* gamma, int R, int C, float* dZ, int ldd, float* dgamma, float* dbeta,
                                 const float* act_scale, const float* act_shift, ptt_stream_t stream);
int ptt_bn_bwd_sums_partials_f64(const double* partial, int chunks, int C, double* sums, ptt_stream_t stream);
/* Round 5: the BatchNorm + ReLU backward of a layer APPLIED BY ITS CONSUMER. With dbeta = sum dy, dgamma = sum dy * xhat
 *   dz = gamma * invstd * (dy - dbeta / R - xhat * dgamma / R) = c0 + c1 * (z - mean) + (mask ? k1 * g : 0)
 * (k1 = gamma * invstd, c1 = -k1 * invstd * dgamma / R, c0 = -k1 * dbeta / R): three per-channel
 * constants instead of a pass that reads g and z and writes dz (ptt_bn_bwd_from_partials_f32 / ptt_bn_bwd_pooled_f32's last
 * launch, 1.5 ms of a 21-ms training step). ptt_bn_bwd_consts_f32 combines the partials of ptt_rows_gemm_bnbwd_f32 (dense
 * gradient), ptt_bn_bwd_pooled_consts_f32 sums over a POOLED gradient first (as ptt_bn_bwd_pooled_f32); both write dgamma,
 * dbeta and the constants. ptt_rows_gemm_bnbwd_fused_f32 = ptt_rows_gemm_bnbwd_f32 whose A operand is that dz, formed while
 * its rows are staged, and (dz_out) written out once for the layer's weight gradient. */
typedef struct ptt_bn_bwd_input {
    const float* g; int ldg;            /* dense: the gradient w.r.t. the layer's activated output (R,K); pooled: (R / ns, K) */
    const int32_t* arg; int ns;         /* pooled: arg-max row inside every group of ns rows, (R / ns, K) contiguous; NULL, 0: dense */
    const float* z; int ldz;            /* the layer's convolution output (R,K) */
    const float* k1; const float* c0; const float* c1; const float* mean;
    const float* act_a; const float* act_b;     /* the ReLU mask: z * act_a + act_b > 0 */
    float* dz_out; int ldd;             /* optional (R,K): dz written by the launch */
} ptt_bn_bwd_input;
int ptt_bn_bwd_consts_f32(const double* partial, int chunks, const float* mean, const float* invstd, const float* gamma, int R, int C,
                          float* dgamma, float* dbeta, float* k1, float* c0, float* c1, ptt_stream_t stream);
int ptt_bn_bwd_pooled_consts_f32(const float* dPooled, int ldp, const int32_t* arg, int ns, const float* Z, int ldz, const float* mean,
                                 const float* invstd, const float* gamma, int R, int C, float* dgamma, float* dbeta, float* k1,
                                 float* c0, float* c1, void* workspace, size_t workspace_bytes, const float* act_scale,
                                 const float* act_shift, ptt_stream_t stream);
int ptt_rows_gemm_bnbwd_fused_supported(int rows, int K, int N, int ns);
int ptt_rows_gemm_bnbwd_fused_f32(const ptt_bn_bwd_input* in, int rows, int K, const float* Wpacked, int N, const float* Z, int ldz,
                                  const float* mean, const float* invstd, const float* act_scale, const float* act_shift, float* out,
                                  int ldo, double* sums_partial, size_t partial_elems, ptt_stream_t stream);
int ptt_bn_finish_partials_f32(const double* partial, int chunks, int C, int R, float eps, float* mean, float* var, float* invstd,
                               ptt_stream_t stream);
int ptt_bn_sums_partials_f64(const double* partial, int chunks, int C, int R, double* sums, ptt_stream_t stream);
/* BatchNorm(train) + ReLU backward of the LAST layer of a SharedMLP + max-pool stage, with the gradient still POOLED:
 * dy[g*ns + k, c] = dPooled[g, c] if k == arg[g, c] else 0 (F.max_pool2d's backward, pointnet2_modules.py:84-88) is formed
 * on the fly — ptt_pool_rows_bwd_f32's (R, C) tensor is never written, the two sums run over G x C entries instead of R x C.
 * The ReLU mask is z * act_scale + act_shift > 0 (deferred activation). _sums / _apply: the SyncBatchNorm split, as
 * ptt_bn_bwd_sums_f64 / ptt_bn_bwd_apply_f32. Workspace: ptt_bn_stats_workspace(R, C). */
int ptt_bn_bwd_pooled_f32(const float* dPooled, int ldp, const int32_t* arg, int ns, const float* Z, int ldz, const float* mean,
                          const float* invstd, const float* gamma, int R, int C, float* dZ, int ldd, float* dgamma, float* dbeta,
                          void* workspace, size_t workspace_bytes, const float* act_scale, const float* act_shift, ptt_stream_t stream);
int ptt_bn_bwd_pooled_sums_f64(const float* dPooled, int ldp, const int32_t* arg, int ns, const float* Z, int ldz, const float* mean,
                               const float* invstd, int R, int C, double* sums, void* workspace, size_t workspace_bytes,
                               const float* act_scale, const float* act_shift, ptt_stream_t stream);
int ptt_bn_bwd_pooled_apply_f32(const float* dPooled, int ldp, const int32_t* arg, int ns, const float* Z, int ldz, const float* mean,
                                const float* invstd, const float* gamma, const float* sum_dy, const float* sum_dy_xhat,
                                const double* count, int R, int C, float* dZ, int ldd, const float* act_scale,
                                const float* act_shift, ptt_stream_t stream);
/* nn.BatchNorm's training-mode bookkeeping in ONE launch (running statistics with momentum, unbiased variance, batch counter);
 * count = the rows the statistics were taken over, a float64 in device memory (SyncBatchNorm: the all-reduced count). */
int ptt_bn_update_running_f32(const float* mean, const float* var, const double* count, float momentum, int C, float* running_mean,
                              float* running_var, int64_t* num_batches_tracked, ptt_stream_t stream);
/* CosineSimAug's first convolution in training mode with its similarity channel split off (p2b_xcoor.py:35-40):
 * z0[b,j,i,:] = P[b,i,:] + cos_t[b,j,i] * w_sim[:]  (rows ordered (b, j, i): search point j, template point i), and its
 * backward in one pass over dz0: dP[b,i,:] = sum_j dz0, dcos[b,j,i] = <dz0[b,j,i,:], w_sim>, dw[:] = sum dz0 * cos_t
 * (fixed summation order). C % 4 == 0; the backward needs C <= 256. */
int ptt_xcorr_z0_f32(const float* P, const float* cos_t, const float* w_sim, int B, int n2, int n1, int C, float* z0, ptt_stream_t stream);
/* the same launch also sums z0's BatchNorm statistics: float64 partials (chunks, 2, C), chunks = ptt_xcorr_z0_stat_chunks(...) */
int ptt_xcorr_z0_stat_chunks(int B, int n2, int n1, int C);
int ptt_xcorr_z0_stats_f32(const float* P, const float* cos_t, const float* w_sim, int B, int n2, int n1, int C, float* z0,
                           double* stats_partial, size_t partial_elems, ptt_stream_t stream);
size_t ptt_xcorr_z0_bwd_workspace(int B, int n1, int C);
/* ptt_xcorr_z0_bwd_f32 when dz0 does not exist yet: G = the gradient w.r.t. relu(BatchNorm(z0)) as ptt_rows_gemm_bnbwd_f32 left it,
 * with that launch's partial sums; layer 0's BatchNorm + ReLU backward is applied while G is read and z0 is recomputed from
 * P / cos / w (neither z0 nor dz0 is read or written). Also returns dgamma / dbeta of that BatchNorm. Same workspace. */
int ptt_xcorr_z0_bnbwd_f32(const double* partial, int chunks, const float* G, const float* P, const float* cos_t, const float* w_sim,
                           const float* mean, const float* invstd, const float* gamma, const float* act_scale, const float* act_shift,
                           int B, int n2, int n1, int C, float* dP, float* dcos, float* dw, float* dgamma, float* dbeta, void* workspace,
                           size_t workspace_bytes, ptt_stream_t stream);
int ptt_xcorr_z0_bwd_f32(const float* dz0, const float* cos_t, const float* w_sim, int B, int n2, int n1, int C, float* dP, float* dcos,
                         float* dw, void* workspace, size_t workspace_bytes, ptt_stream_t stream);
/* Round 3, launch consolidation of the training step (the step is bound by device time, and ~250 of its launches were
 * five-microsecond bookkeeping kernels):
 *   ptt_bn_train_tail   what a training-mode BatchNorm layer does with fresh batch statistics besides normalising
 *     (torch/nn/modules/batchnorm.py; the reference builds the layers in pytorch_utils.py:94-114), written by the SAME
 *     launch that forms the statistics: act_a = gamma * invstd and act_b = beta - mean * act_a (the deferred activation's
 *     constants, each operation rounded separately), running_mean / running_var updated with `momentum` and the n / (n - 1)
 *     variance (n = R), *num_batches_tracked += 1. Null members switch a piece off; a NULL tail = the plain entry point.
 *   ptt_pack_weights_f32  packs n_jobs weights (views by element strides: W[col * stride_out + k * stride_k], so a
 *     transposed pack is a stride swap) into one arena in ONE launch; jobs live in DEVICE memory (a training step re-packs
 *     every weight after each optimiser update: the table is built once, the launch repeats). out_offset: float offset
 *     inside `arena`, a multiple of 4; each job writes ptt_packed_weight_elems(Cout, K) floats.
 *   ptt_sa_z0_rows_f32  the front of one SA level in training mode with layer 0 hoisted, one pass: for row (b, m, k),
 *     n = idx[b,m,k]: rel = (xyz[b,n] - new_xyz[b,m]) (* float(1 / radius) when normalize_xyz, as torch divides by a host
 *     scalar) — QueryAndGroup,
 *     pointnet2_utils.py:350-354 — and z0 = term[b,n,:] + Wx . rel, the first SharedMLP convolution (pytorch_utils.py:12-36;
 *     channel order xyz first, pointnet2_utils.py:359-361) with its feature half `term` (B,N,C) evaluated once per point;
 *     term NULL: a level without point features. z0 (B*M*ns, C) rows, rel_rows (B*M*ns, 3) (the weight gradient of Wx
 *     (C,3), row stride ldw >= 3, needs them). C % 4 == 0. No gradient w.r.t. the coordinates is provided: callers with
 *     learnable coordinates (the box head's vote aggregation) keep ptt_group_f32 + ptt_gather_rows_f32. */
int ptt_sa_z0_rows_f32(const float* xyz, const float* new_xyz, const int32_t* idx, const float* term, const float* wx, int ldw, int B,
                       int N, int M, int ns, int C, float radius, int normalize_xyz, float* z0, float* rel_rows, ptt_stream_t stream);
/* The same launch also summing the BatchNorm statistics of layer 0 (the float64 column sums / sums of squares of z0) as
 * partials (chunks, 2, C), chunks = ptt_sa_z0_rows_stat_chunks(B, M, ns, C), for ptt_bn_finish_partials_f32 (or its _train form) — no
 * statistics pass over z0. C <= 1024. */
int ptt_sa_z0_rows_stat_chunks(int B, int M, int ns, int C);
int ptt_sa_z0_rows_stats_f32(const float* xyz, const float* new_xyz, const int32_t* idx, const float* term, const float* wx, int ldw, int B,
                             int N, int M, int ns, int C, float radius, int normalize_xyz, float* z0, float* rel_rows,
                             double* stats_partial, size_t partial_elems, ptt_stream_t stream);
typedef struct ptt_bn_train_tail {
    const float* gamma; const float* beta;      /* (C) affine parameters, needed for act_a / act_b */
    float* act_a; float* act_b;                 /* (C) out, or both NULL */
    float* running_mean; float* running_var;    /* (C) updated in place, or both NULL */
    int64_t* num_batches_tracked;               /* incremented, or NULL */
    float momentum;
} ptt_bn_train_tail;
int ptt_bn_stats_train_f32(const float* X, int R, int C, int ldx, float eps, float* mean, float* var, float* invstd,
                           void* workspace, size_t workspace_bytes, const ptt_bn_train_tail* tail, ptt_stream_t stream);
int ptt_bn_finish_partials_train_f32(const double* partial, int chunks, int C, int R, float eps, float* mean, float* var,
                                     float* invstd, const ptt_bn_train_tail* tail, ptt_stream_t stream);
typedef struct ptt_pack_job {
    const float* W;
    int64_t out_offset, stride_out, stride_k;
    int32_t Cout, K;
} ptt_pack_job;
int ptt_pack_weights_f32(const ptt_pack_job* jobs_device, int n_jobs, float* arena, ptt_stream_t stream);
size_t ptt_linear_wgrad2_workspace(int R, int Cout, int Cin);
int ptt_linear_wgrad2_f32(const float* dZ, int ldz, const float* X, int ldx, int R, int Cout, int Cin, float* dW,
                          int accumulate, void* workspace, size_t workspace_bytes, const float* x_scale, const float* x_shift,
                          ptt_stream_t stream);

/* The four tracking losses of a training step with their gradients, in two launches (reference
 * ptt/models/voting_heads/centroids_voting_head.py:29-62: BCEWithLogits (mean) over the seed scores + masked smooth-L1 of the votes;
 * box_voting_head.py:33-66: masked BCEWithLogits of the proposal scores + label-weighted smooth-L1 of the 4 box numbers; the
 * proposal labels / masks of box_voting_head.py:96-104 are formed in the kernels from the proposal centres).
 *   seed_cls (B,N) logits; cls_label (B,Ns) per search POINT with search_inds (B,N) int64 the seeds' point indices, or
 *   search_inds NULL and cls_label (B,N) per seed; votes (B,N,3); reg_label (B, ld_reg >= 4): gt centre + angle;
 *   box_data (B,M,5) = (x, y, z, angle, score logit); centres (B,M,3); pos_weight_*: device scalars (the modules' buffers).
 * out8: [0] the weighted total, [1..4] the un-weighted losses (seed cls, seed reg, proposal cls, proposal reg), [5..7] the sums
 * of the seed labels / proposal masks / proposal labels (the backward's denominators). Reductions in float64, fixed order.
 * ptt_track_losses_bwd_f32: d total / d (seed_cls, votes, box_data) times upstream[0] (device scalar; NULL: 1). */
typedef struct ptt_track_loss_desc {
    const float* seed_cls;
    const float* cls_label;
    const int64_t* search_inds;
    const float* votes;
    const float* reg_label;
    const float* box_data;
    const float* centres;
    const float* pos_weight_seed;
    const float* pos_weight_box;
    int32_t B, N, Ns, M, ld_reg;
    float w_seed_cls, w_seed_reg, w_box_cls, w_box_reg;
} ptt_track_loss_desc;
int ptt_track_losses_f32(const ptt_track_loss_desc* desc, float* out8, float* total /* optional: out8[0] once more */, ptt_stream_t stream);
int ptt_track_losses_bwd_f32(const ptt_track_loss_desc* desc, const float* out8, const float* upstream, float* d_seed_cls,
                             float* d_votes, float* d_box_data, ptt_stream_t stream);
/* torch.nn.utils.clip_grad_norm_(params, max_norm) + torch.optim.Adam.step() (tools/train_utils/train_utils.py:48-51,
 * optimization/__init__.py:12-14) over a device table of tensors in two launches. The caller cuts every tensor into chunks of
 * ptt_adam_chunk_elems() elements: chunk_tensor[w] = table index, chunk_first[w] = first element. hyper: step_size =
 * lr / (1 - beta1^t), bias2_sqrt = sqrt(1 - beta2^t); max_norm <= 0: no clipping (partial may be NULL); write_clipped: also
 * scale .grad in place as clip_grad_norm_ leaves it. partial: n_chunks doubles of workspace; norm_out (optional): the total norm. */
typedef struct ptt_adam_tensor {
    float* param;
    float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    int64_t n;
} ptt_adam_tensor;
typedef struct ptt_adam_hyper {
    float beta1, beta2, one_minus_beta1, one_minus_beta2 /* formed in double by the caller, as torch does */, eps, step_size, bias2_sqrt,
          weight_decay, max_norm;
    int32_t write_clipped;
} ptt_adam_hyper;
int ptt_adam_chunk_elems(void);
int ptt_adam_clip_step_f32(const ptt_adam_tensor* tensors_device, const int32_t* chunk_tensor_device, const int64_t* chunk_first_device,
                           int n_chunks, const ptt_adam_hyper* hyper, double* partial, size_t partial_elems, float* norm_out,
                           ptt_stream_t stream);
/* The same two launches with the hyper-parameters read from DEVICE memory when the update launch runs: the form a captured
 * training step (hipGraph) replays, the host writing step_size / bias2_sqrt (and a scheduler's lr) of the step into hyper_device
 * before each replay. clip != 0: the norm pass is launched and hyper_device->max_norm (> 0) clips; clip == 0: hyper_device->max_norm
 * must be <= 0. Arithmetic identical to ptt_adam_clip_step_f32. */
int ptt_adam_clip_step_dev_f32(const ptt_adam_tensor* tensors_device, const int32_t* chunk_tensor_device, const int64_t* chunk_first_device,
                               int n_chunks, const ptt_adam_hyper* hyper_device, int clip, double* partial, size_t partial_elems,
                               float* norm_out, ptt_stream_t stream);
/* The two element-wise ends of CosineSimAug's cosine map in training (p2b_xcoor.py:35-42 via nn.CosineSimilarity); the map
 * itself is a batched product of unit rows.
 *   ptt_unit_rows_f32     x addressed as x[b * sb + j * sn + c * sc] (any layout) -> unit (B,n,C) rows x / max(|x|, eps) and
 *                         nrm (B,n) = max(|x|, eps), stored NEGATIVE where the clamp is active
 *   ptt_cos_bwd_rows_f32  dx[b,j,:] = (A[b,j,:] - (sum_i G[b,j,i] cos[b,j,i]) unit[b,j,:]) / |nrm[b,j]| (no projection term where
 *                         nrm < 0), A (B,n,C) = the caller's product of G with the OTHER side's unit rows; G / cos addressed as
 *                         [b * map_sb + j * own + i * other], i < m; dx written as dx[b * sb + j * sn + c * sc] */
int ptt_unit_rows_f32(const float* x, long long sb, long long sn, long long sc, int B, int n, int C, float eps, float* unit, float* nrm,
                      ptt_stream_t stream);
int ptt_cos_bwd_rows_f32(const float* A, const float* unit, const float* nrm, const float* G, const float* cosm, long long map_sb,
                         long long own, long long other, int m, int B, int n, int C, float* dx, long long sb, long long sn, long long sc,
                         ptt_stream_t stream);
/* Layer 0 of a hoisted SA level, backward, in one pass in row order over the gradient G (R, C) of its ACTIVATED output (training):
 * the BatchNorm + ReLU backward is applied row by row from G, the stored z0 and the sums of `partial` (ptt_rows_gemm_bnbwd_f32's),
 * dwx (C,3) = dz0^T rel_rows is accumulated on the way, and dz0 is written only if dz_out != NULL (a level with point features:
 * its row scatter reads it; dz_out may alias G). Also dgamma / dbeta of that BatchNorm. dwx NULL: the partial sums of dwx,
 * [workspace bytes / (12 C)][C][3], stay in the workspace for ptt_grad_finish_f32. */
size_t ptt_sa_z0_bnbwd_workspace(long long R, int C);
int ptt_sa_z0_bnbwd_f32(const double* partial, int chunks, const float* G, const float* Z0, const float* rel_rows, const float* mean,
                        const float* invstd, const float* gamma, const float* act_scale, const float* act_shift, long long R, int C,
                        float* dz_out, float* dwx, float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, ptt_stream_t stream);

/* The parameter gradients of a training step finished by ONE launch (reference tools/train_utils/train_utils.py:47-49:
 * loss.backward() has every .grad complete before clip_grad_norm_ reads it; tools/train_tracking.py:158-159: the gradient that
 * DistributedDataParallel averages over ranks). The *_partials entries are ptt_linear_wgrad_f32 / ptt_linear_wgrad2_f32 /
 * ptt_colsum_f32 without their finishing launch: the row-chunk partial sums [*nchunks][Cout * Cin] (resp. [*nchunks][C]) stay in
 * `workspace` (same size queries). ptt_grad_finish_f32 then adds, for every SEGMENT (one destination inside the flat float32
 * gradient buffer `flat`: n elements as rows of `cols`, row stride `ld`, first element `dst`; 16-byte aligned pieces when `vec`),
 * the chunks of its jobs [job0, job0 + njobs) in a fixed order; `out` (a power of two, 4 ... 256) = outputs per workgroup, the
 * other 256 / out thread groups split the chunks. blocks_device: pairs (segment, first output unit) per workgroup; a unit is
 * 4 elements of a vec segment, else 1. Every job of a segment holds partials of that segment's n. Tables live on the device. */
typedef struct ptt_grad_job {
    const float* partial;
    int32_t nchunks, reserved;
} ptt_grad_job;
typedef struct ptt_grad_segment {
    int64_t dst;
    int32_t n, cols, ld, job0, njobs, out, vec, reserved;
} ptt_grad_segment;
int ptt_linear_wgrad_partials_f32(const float* dZ, int ldz, const float* X, int ldx, int R, int Cout, int Cin, void* workspace,
                                  size_t workspace_bytes, const float* x_scale, const float* x_shift, int* nchunks, ptt_stream_t stream);
int ptt_linear_wgrad2_partials_f32(const float* dZ, int ldz, const float* X, int ldx, int R, int Cout, int Cin, void* workspace,
                                   size_t workspace_bytes, const float* x_scale, const float* x_shift, int* nchunks, ptt_stream_t stream);
int ptt_colsum_partials_f32(const float* X, int R, int C, int ldx, void* workspace, size_t workspace_bytes, int* nchunks,
                            ptt_stream_t stream);
int ptt_grad_finish_f32(const ptt_grad_segment* segments_device, const ptt_grad_job* jobs_device, const int32_t* blocks_device,
                        int n_blocks, float* flat, ptt_stream_t stream);

/* The backward pass of a Point-Transformer block's attention core (variants.py:158-163 under autograd) with three passes folded away:
 *   ptt_rows_gemm_rsum16_f32     plain = X @ W^T, out = plain + residual and, per group of 16 consecutive rows (a point's 16
 *                                neighbours), gsum[g, :] = the column sums of plain over the group — one epilogue. With X = the gradient
 *                                of fc_gamma's hidden layer and residual = the aggregation's gradient of (v + pos_enc): plain is the
 *                                gradient of the pair input, out the gradient of pos_enc (its two consumers' sum, which autograd would
 *                                form with a pass of its own) and gsum the gradient of q (a reduction pass).
 *                                ptt_rows_gemm_rsum16_supported: K % 128 == 0, N % 128 == 0, rows % 16 == 0.
 *   ptt_scatter_rows_csr_sub_f32 ptt_scatter_rows_csr_f32 with out = minuend - (the sums), minuend NULL = 0: the gradient of k is
 *                                MINUS the scatter of the pair input's gradient (a negation pass). */
int ptt_rows_gemm_rsum16_supported(int rows, int K, int N, int ldx);
int ptt_rows_gemm_rsum16_f32(const float* X, int rows, int K, int ldx, const float* Wpacked, int N, const float* residual, int ldr,
                             float* out, int ldo, float* plain, int ldp, float* gsum, int ldg, ptt_stream_t stream);
int ptt_scatter_rows_csr_sub_f32(const float* g, const int32_t* order, const int32_t* start, int B, int N, int E, int C,
                                 const float* minuend, float* out, ptt_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PTT_HIP_H */
