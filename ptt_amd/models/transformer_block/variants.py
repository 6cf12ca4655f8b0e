"""Mirror of ptt/models/transformer_block/variants.py: TransformerBlock (:127-165, the variant every
shipped config selects) and TransformerBlockSTD (:12-40).

TransformerBlock is Point-Transformer *vector* attention over the k nearest neighbours:
    delta_ij = fc_delta(xyz_i - xyz_j)
    attn_ij  = softmax_j( fc_gamma(q_i - k_j + delta_ij) / sqrt(d_model) )      (per channel)
    res_i    = fc2( sum_j attn_ij * (v_j + delta_ij) ) + features_i
Parameter names (fc1, fc2, fc_delta.{0,2}, fc_gamma.{0,2}, w_qs, w_ks, w_vs) are the reference's,
so its checkpoints load unchanged.

Eval mode on a HIP device runs four kernels: kNN, the q|k|v projection with fc1 folded in, the fused pair
kernel (fc_delta[2], fc_gamma[0], fc_gamma[2] on fp32 MFMA + softmax + weighted sum, with every
(B,N,k,d_model) intermediate kept in LDS/registers) and fc2+residual.

Return contract = the reference's (variants.py:165): `forward(xyz, features) -> (res, attn)` with attn the (B,N,k,d_model)
attention tensor — ALWAYS, unless the CALLER opts out with `want_attn=False` (second value None; the tensor is then never
written to HBM: 100 MB per 48 frames of 128 points). Who opts out: this build's own callers, which keep `[0]` only exactly as
the reference's heads do (centroids_voting_head.py:76, box_voting_head.py:86) — ptt_amd's CentroidVotingHead / BoxVotingHead
and hot_path.FrameHotPath. Anyone else calling the block as the reference does gets what the reference returns.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import ops, train_ops
from ..model_utils import index_points, square_distance


# at most this many points (B * N) take the per-layer inference form of TransformerBlock instead of the fused pair kernel
PER_LAYER_MAX_POINTS = ops.ONE_FRAME_MAX_POINTS


def _rows2d(layers, x):
    """layers(x) for Linear stacks acting on the last dim, evaluated on the flattened (rows, C) view. Same numbers;
    on ROCm the autograd backward of nn.Linear over a 4-D (B,N,k,C) input takes a GEMM path that runs at 1-4 TFLOP/s
    (scripts/probes/linear_train_probe.py: 14 ms vs 1.6 ms for forward + backward of one 512x512 layer at B=48)."""
    return layers(x.reshape(-1, x.shape[-1])).reshape(*x.shape[:-1], -1)


SPATIAL_ORDER_MIN_POINTS = 512       # clouds from this size on are handed to the pair kernel in Morton order


class TransformerBlock(nn.Module):
    def __init__(self, d_points, d_model, k, **kwargs) -> None:
        super().__init__()
        self.fc1 = nn.Linear(d_points, d_model)
        self.fc2 = nn.Linear(d_model, d_points)
        self.fc_delta = nn.Sequential(nn.Linear(3, d_model), nn.ReLU(), nn.Linear(d_model, d_model))
        self.fc_gamma = nn.Sequential(nn.Linear(d_model, d_model), nn.ReLU(), nn.Linear(d_model, d_model))
        self.w_qs = nn.Linear(d_model, d_model, bias=False)
        self.w_ks = nn.Linear(d_model, d_model, bias=False)
        self.w_vs = nn.Linear(d_model, d_model, bias=False)
        self.k = k
        self.d_model = d_model
        self.d_points = d_points
        self._cache = None

    # ---------------------------------------------------------------- fused-path parameters
    def _fusable(self, xyz, features):
        """Eval mode on a HIP device, the instantiated shape (d_model 512, k 16, an even number of points) and no
        autograd graph being recorded; an eval-mode HIP call that is turned away says so once (ops.note_unfused)."""
        if self.training or not xyz.is_cuda:
            return False
        name = 'TransformerBlock(d_model=%d, k=%d)' % (self.d_model, self.k)
        if ops.autograd_recording(self, xyz, features):
            return ops.note_unfused(name, 'autograd is recording (wrap inference in torch.no_grad())')
        if self.d_model != 512 or self.k != 16:
            return ops.note_unfused(name, 'ptt_pt_attn_pair_f32 instantiates d_model 512, k 16')
        if xyz.shape[1] % 2 != 0 or xyz.shape[1] < self.k:
            return ops.note_unfused(name, 'needs an even number of points >= k (got %d)' % xyz.shape[1])
        if features.dtype != torch.float32 or xyz.dtype != torch.float32:
            return ops.note_unfused(name, 'inputs must be float32')
        return True

    def _params(self):
        ts = [self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias, self.w_qs.weight, self.w_ks.weight,
              self.w_vs.weight, self.fc_delta[0].weight, self.fc_delta[0].bias, self.fc_delta[2].weight,
              self.fc_delta[2].bias, self.fc_gamma[0].weight, self.fc_gamma[0].bias, self.fc_gamma[2].weight,
              self.fc_gamma[2].bias]
        key = tuple((t.data_ptr(), t._version) for t in ts)
        if self._cache is not None and self._cache[0] == key:
            return self._cache[1]
        with torch.no_grad():
            f = lambda t: t.detach().float().contiguous()
            # fc1 has no activation and only feeds w_qs / w_ks / w_vs (variants.py:155-156), so the two linear maps
            # are one: [q|k|v] = (W_qkv W_1) f + W_qkv b_1 — 256 -> 1536 instead of 256 -> 512 -> 1536 (product in f64)
            wqkv = torch.cat([self.w_qs.weight, self.w_ks.weight, self.w_vs.weight], 0).double()
            P = dict(
                qkv=ops.pack_weight((wqkv @ self.fc1.weight.double()).float().contiguous()),
                qkv_b=(wqkv @ self.fc1.bias.double()).float().contiguous(),
                wd1=ops.pack_delta0(self.fc_delta[0].weight, self.fc_delta[0].bias),
                wd2=ops.pack_weight(self.fc_delta[2].weight), bd2=f(self.fc_delta[2].bias),
                wg1=ops.pack_weight(self.fc_gamma[0].weight), bg1=f(self.fc_gamma[0].bias),
                wg2=ops.pack_weight(self.fc_gamma[2].weight), bg2=f(self.fc_gamma[2].bias),
                fc2=ops.pack_weight(self.fc2.weight), fc2_b=f(self.fc2.bias),
                wd1_lin=ops.pack_weight(self.fc_delta[0].weight), bd1=f(self.fc_delta[0].bias),
                w1b=torch.cat((f(self.fc_delta[0].weight), f(self.fc_delta[0].bias)[:, None]), 1).contiguous())
        ops.publish_params(self.fc1.weight.device)
        self._cache = (key, P)
        return P

    # xyz: b x n x 3, features: b x n x f
    def forward(self, xyz, features, knn=None, want_attn=True):
        """-> (res (B,N,d_points), attn (B,N,k,d_model)), as the reference (variants.py:149-165). Extensions of its signature:
        `knn` (used on the fused path only): (knn_idx (B,N,k) int32, rel (B,N,k,3)) of `xyz` already formed by the caller — the
        backbone computes the seeds' neighbours beside its ball queries; `want_attn=False`: the caller keeps `[0]` only, attn is
        None and never leaves the chip."""
        if self._fusable(xyz, features):
            P = self._params()
            D = self.d_model
            xyz = xyz.contiguous()
            if knn is not None and knn[0].shape == (xyz.shape[0], xyz.shape[1], self.k):
                knn_idx, rel = knn
            else:
                knn_idx, rel = ops.knn(xyz, self.k, want_rel=True)
            if xyz.shape[0] * xyz.shape[1] <= PER_LAYER_MAX_POINTS and not want_attn:
                # a handful of frames (one tracklet frame: 128 / 64 points): the fused pair kernel's one workgroup per two
                # points is a 64-workgroup launch of three chained 512 x 512 GEMMs (106 us on 64 of the 256 CUs). Per layer,
                # each GEMM over the (point, neighbour) rows fills the chip, and the element-wise steps between them ride in
                # the GEMMs' operand staging / epilogue (ptt_row_jobs_f32): kNN, then FOUR launches —
                #   [q|k|v projection  ||  fc_delta with its first Linear + ReLU formed while the A tile is staged]
                #   fc_gamma[0] on q_i - k_j + pos_ij gathered while the A tile is staged
                #   fc_gamma[2] with the softmax over the 16 neighbours and the weighted sum as its epilogue
                #   fc2 + residual
                # instead of the nine of round 3 (a launch of this chain costs 7 - 15 us, mostly latency).
                B, N = xyz.shape[0], xyz.shape[1]
                dev = xyz.device
                qkv = torch.empty((B, N, 3 * D), dtype=torch.float32, device=dev)
                pos = torch.empty((B * N * self.k, D), dtype=torch.float32, device=dev)
                g = torch.empty_like(pos)
                res = torch.empty((B, N, D), dtype=torch.float32, device=dev)
                out = torch.empty((B, N, self.d_points), dtype=torch.float32, device=dev)
                # (the long job first: its workgroups take the CUs at once, the projection's short ones fill in behind them)
                ops.row_jobs([ops.row_job(P['wd2'], D, prologue=1, rel=rel.view(-1, 3), w1=P['w1b'], K=D, shift=P['bd2'], out=pos),
                              ops.row_job(P['qkv'], 3 * D, x=features, shift=P['qkv_b'], out=qkv)])
                ops.row_jobs([ops.row_job(P['wg1'], D, prologue=2, qkv=qkv, knn=knn_idx.view(-1, self.k), pos=pos, q_off=0, k_off=D,
                                          N=N, K=D, shift=P['bg1'], act=1, out=g)])
                ops.row_jobs([ops.row_job(P['wg2'], D, x=g, epilogue=1, qkv=qkv, knn=knn_idx.view(-1, self.k), pos=pos, v_off=2 * D,
                                          N=N, sm_scale=1.0 / np.sqrt(D), out=res)])
                ops.row_jobs([ops.row_job(P['fc2'], self.d_points, x=res, shift=P['fc2_b'], res=features, out=out)])
                return out, None
            qkv = ops.linear(features, P['qkv'], 3 * D, None, P['qkv_b'])
            if xyz.shape[0] * xyz.shape[1] <= PER_LAYER_MAX_POINTS:
                # the same per-layer form with the (B,N,k,D) attention tensor written out
                pairs = rel.view(-1, 3)
                h = ops.linear(pairs, P['wd1_lin'], D, None, P['bd1'], relu=True)
                pos = ops.linear(h, P['wd2'], D, None, P['bd2']).view(xyz.shape[0], xyz.shape[1], self.k, D)
                t = ops.pt_pair_input_qkv(qkv, knn_idx, pos, D)
                g = ops.linear(t.view(-1, D), P['wg1'], D, None, P['bg1'], relu=True)
                a = ops.linear(g, P['wg2'], D, None, P['bg2']).view_as(t)
                res, attn = ops.pt_attn_fwd_qkv(a, qkv, knn_idx, pos, D, 1.0 / np.sqrt(D), want_attn)
            else:
                # many points per cloud: the workgroups take them along a space-filling curve, so that those running together
                # gather k | v rows of the same neighbourhood out of L2 (FPS order = far apart: ~10x the compulsory HBM reads
                # at 2048 points); the k | v rows of a 128-point cloud fit L2 whole
                order = ops.spatial_order(xyz) if SPATIAL_ORDER_MIN_POINTS <= xyz.shape[1] <= 8192 else None
                res, attn = ops.pt_attn_pair(xyz, knn_idx, qkv, P['wd1'], P['wd2'], P['bd2'], P['wg1'],
                                             P['bg1'], P['wg2'], P['bg2'], D, want_attn, rel=rel, order=order)
            res = ops.linear(res, P['fc2'], self.d_points, None, P['fc2_b'], False, features)
            return res, attn

        if train_ops.pt_block_usable(self, xyz, features):
            # training mode on a HIP device: the GEMMs on the persistent row GEMM / weight-gradient kernels, the element-wise
            # chains over the (B,N,k,D) tensors as one hand-written pass each (ptt_amd/train_ops.py: _PairInput,
            # _AttnAggregate); the neighbour gathers of k and v happen inside those passes and their gradients come back
            # through the deterministic row scatter-add. kNN: the HIP kernel (ascending (distance, index), a stable
            # refinement of the reference's argsort).
            knn_idx, rel = ops.knn(xyz.contiguous(), self.k, want_rel=True)          # rel = xyz_i - xyz_j (:158)
            if xyz.requires_grad:                                                     # the box head's proposals carry grad
                rel = train_ops._KnnRel.apply(xyz, knn_idx, rel)
            # every Linear on the row kernels (train_ops._RowsLinear / _RowsMlp2): bias, ReLU, the residual and the ReLU
            # backward live in GEMM epilogues; the K = 3 layer fc_delta[0] and its 512 x 3 weight gradient on the linear /
            # weight-gradient kernels
            x = train_ops.rows_linear(self.fc1, features)
            q, kf, vf = (train_ops.rows_linear(m, x) for m in (self.w_qs, self.w_ks, self.w_vs))
            pos_enc = train_ops.rows_mlp2(self.fc_delta, rel)                          # (B,N,k,D)
            # the backward scatters of k and v run over the same neighbour indices: their (bin, entry) order is formed once
            order, start = ops.scatter_csr(knn_idx.view(knn_idx.shape[0], -1), knn_idx.shape[1])
            if train_ops.ATTN_CORE:
                # pair input, fc_gamma, softmax and aggregate as one autograd function (its backward folds three passes over the
                # (B,N,k,D) tensors into GEMM / scatter epilogues: train_ops._AttnCore)
                res, attn = train_ops.attn_core(self.fc_gamma, q, kf, vf, knn_idx, pos_enc, 1.0 / np.sqrt(self.d_model), order, start)
            else:
                t = train_ops._PairInput.apply(q, kf, knn_idx, pos_enc, order, start)
                a = train_ops.rows_mlp2(self.fc_gamma, t)
                res, attn = train_ops._AttnAggregate.apply(a, vf, knn_idx, pos_enc, 1.0 / np.sqrt(self.d_model), order, start)
            res = train_ops.rows_linear(self.fc2, res, residual=features)
            return res, attn

        # CPU: the reference's op sequence in stock torch (variants.py:149-165)
        dists = square_distance(xyz, xyz)
        knn_idx = dists.argsort()[:, :, :self.k]
        knn_xyz = index_points(xyz, knn_idx)
        pre = features
        x = self.fc1(features)
        q, k, v = self.w_qs(x), index_points(self.w_ks(x), knn_idx), index_points(self.w_vs(x), knn_idx)
        pos_enc = _rows2d(self.fc_delta, xyz[:, :, None] - knn_xyz)
        attn = _rows2d(self.fc_gamma, q[:, :, None] - k + pos_enc)
        attn = F.softmax(attn / np.sqrt(k.size(-1)), dim=-2)
        # reference: torch.einsum('bmnf,bmnf->bmf', attn, v + pos_enc) (variants.py:163). einsum lowers this to
        # B*N*512 batched (1x16)@(16x1) GEMMs — 164 ms of a 235 ms training step on ROCm; a product + sum over the
        # neighbour axis is the same contraction
        res = (attn * (v + pos_enc)).sum(dim=2)
        res = self.fc2(res) + pre
        return res, attn


class TransformerBlockSTD(nn.Module):
    """Dense scaled-dot-product variant (variants.py:12-40): softmax(q k^T / sqrt(d)) @ (v + fc_delta(xyz)) — the only
    literal Q.K^T / attn.V form in the reference (no shipped config selects it).

    Eval mode on a HIP device runs it as fp32-MFMA GEMMs on the linear kernel: the stacked q|k|v projection with fc1
    folded in (as TransformerBlock does), fc_delta as two row-wise layers whose second one adds v (residual) so that
    v + pos_enc never needs its own pass, then per frame  S = Q K^T  (K packed as the B operand straight from the q|k|v
    buffer, ptt_pack_weight_strided_f32 + ptt_linear_batched_f32), an in-place row softmax with the 1/sqrt(d) scale
    (ptt_softmax_rows_f32),  O = attn (V + delta)  (the values packed TRANSPOSED), and fc2 + residual."""

    def __init__(self, d_points, d_model, k, **kwargs) -> None:
        super().__init__()
        self.fc1 = nn.Linear(d_points, d_model)
        self.fc2 = nn.Linear(d_model, d_points)
        self.fc_delta = nn.Sequential(nn.Linear(3, d_model), nn.ReLU(), nn.Linear(d_model, d_model))
        self.w_qs = nn.Linear(d_model, d_model, bias=False)
        self.w_ks = nn.Linear(d_model, d_model, bias=False)
        self.w_vs = nn.Linear(d_model, d_model, bias=False)
        self.k = k
        self.d_model = d_model
        self.d_points = d_points
        self._cache = None

    def _fusable(self, xyz, features):
        if self.training or not xyz.is_cuda:
            return False
        name = 'TransformerBlockSTD(d_model=%d)' % self.d_model
        if ops.autograd_recording(self, xyz, features):
            return ops.note_unfused(name, 'autograd is recording (wrap inference in torch.no_grad())')
        if features.dtype != torch.float32 or xyz.dtype != torch.float32:
            return ops.note_unfused(name, 'inputs must be float32')
        return True

    def _params(self):
        ts = [self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias, self.w_qs.weight, self.w_ks.weight,
              self.w_vs.weight, self.fc_delta[0].weight, self.fc_delta[0].bias, self.fc_delta[2].weight, self.fc_delta[2].bias]
        key = tuple((t.data_ptr(), t._version) for t in ts)
        if self._cache is not None and self._cache[0] == key:
            return self._cache[1]
        with torch.no_grad():
            f = lambda t: t.detach().float().contiguous()
            wqkv = torch.cat([self.w_qs.weight, self.w_ks.weight, self.w_vs.weight], 0).double()
            P = dict(qkv=ops.pack_weight((wqkv @ self.fc1.weight.double()).float().contiguous()),
                     qkv_b=(wqkv @ self.fc1.bias.double()).float().contiguous(),
                     wd1=ops.pack_weight(self.fc_delta[0].weight), bd1=f(self.fc_delta[0].bias),
                     wd2=ops.pack_weight(self.fc_delta[2].weight), bd2=f(self.fc_delta[2].bias),
                     fc2=ops.pack_weight(self.fc2.weight), fc2_b=f(self.fc2.bias))
        ops.publish_params(self.fc1.weight.device)
        self._cache = (key, P)
        return P

    def forward(self, xyz, features, knn=None, want_attn=True):
        # `knn`: the callers hand every block the neighbour table formed beside their sampling; full attention has no use for it.
        # `want_attn`: accepted for the callers' sake; the (B,N,N) attention matrix exists either way and is always returned
        if self._fusable(xyz, features):
            P, D = self._params(), self.d_model
            B, N, _ = xyz.shape
            qkv = ops.linear(features, P['qkv'], 3 * D, None, P['qkv_b'])                       # (B,N,3D)
            h = ops.linear(xyz.contiguous(), P['wd1'], D, None, P['bd1'], relu=True)            # relu(fc_delta[0](xyz))
            vd = ops.linear(h, P['wd2'], D, None, P['bd2'], False, qkv[:, :, 2 * D:])           # v + fc_delta(xyz)
            kp = ops.pack_weight_strided(qkv[:, :, D:2 * D], N, D, 3 * D, 1, B, N * 3 * D)      # K rows as the B operand
            attn = ops.linear_batched(qkv[:, :, 0:D], kp, N)                                    # Q K^T  (B,N,N)
            ops.softmax_rows_(attn, 1.0 / np.sqrt(D))
            vp = ops.pack_weight_strided(vd, D, N, 1, D, B, N * D)                              # (V + delta)^T as the B operand
            res = ops.linear_batched(attn, vp, D)                                               # attn (V + delta)
            res = ops.linear(res, P['fc2'], self.d_points, None, P['fc2_b'], False, features)
            return res, attn
        pre = features
        x = self.fc1(features)
        q, k, v = self.w_qs(x), self.w_ks(x), self.w_vs(x)
        attn = F.softmax(q @ k.transpose(1, 2) / np.sqrt(k.size(-1)), dim=-1)
        res = attn @ (v + self.fc_delta(xyz))
        res = self.fc2(res) + pre
        return res, attn
