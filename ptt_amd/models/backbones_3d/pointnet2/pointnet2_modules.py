"""Mirror of ptt/models/backbones_3d/pointnet2/pointnet2_modules.py: PointnetSAModuleVotes (:22-90).

Same constructor, attributes (`grouper`, `mlp_module`, `sample_method`), state_dict keys and
`forward(xyz, features, npoint, inds=None) -> (new_xyz, new_features, inds_int64)` contract.

Two execution paths, chosen per call:
  * eval mode on a HIP device (inference / tracking): sample -> ball query -> ONE fused kernel
    (ptt_sa_fused_fwd_f32: group, centre-subtract, /radius, concat, 3x[1x1 conv + folded BN +
    ReLU] on fp32 MFMA, max over nsample). The grouped (B,C,M,ns) tensors the reference
    materialises (pointnet2_utils.py:351-361) never exist. Output features are stored
    point-major (B,M,C) and returned as the (B,C,M) transposed view, which is also the layout
    the next level gathers from with coalesced loads.
  * training mode (BatchNorm needs batch statistics): the reference's op sequence on the HIP
    ops + stock torch conv/BN, with autograd through gather/group.
"""
import os
from typing import List

import torch
import torch.nn as nn
import torch.nn.functional as F

from .... import ops, train_ops
from . import pointnet2_utils
from . import pytorch_utils as pt_utils


class PointnetSAModuleVotes(nn.Module):
    def __init__(self, *, mlp: List[int], radius: float = None, nsample: int = None, bn: bool = True,
                 use_xyz: bool = True, normalize_xyz: bool = False, sample_uniformly: bool = False,
                 sample_method='fps'):
        super().__init__()
        self.radius = radius
        self.nsample = nsample
        self.mlp_module = None
        self.use_xyz = use_xyz
        self.normalize_xyz = normalize_xyz
        self.sample_uniformly = sample_uniformly
        self.grouper = pointnet2_utils.QueryAndGroup(radius, nsample, use_xyz=use_xyz, ret_grouped_xyz=True,
                                                     normalize_xyz=normalize_xyz, sample_uniformly=sample_uniformly,
                                                     ret_unique_cnt=False)
        mlp_spec = mlp
        if use_xyz and len(mlp_spec) > 0:
            mlp_spec[0] += 3        # in place on the caller's list, exactly as the reference does (:51-53)
        self.mlp_module = pt_utils.SharedMLP(mlp_spec, bn=bn)
        self.sample_method = sample_method
        self._fused_cache = None    # (key, [(wpacked, scale, shift, cin, cout, relu), ...])
        self._arange_cache = None
        self.centres_knn_k = 0      # > 0: a caller that runs a TransformerBlock on this level's centres (the box head) wants their
        self.centres_knn = None     #      kNN; at one frame it is formed in the sampling launch and left here (knn_idx, rel)

    # ------------------------------------------------------------------ sampling (reference :63-77)
    def _sample(self, xyz, features, npoint):
        if self.sample_method == 'fps':
            return pointnet2_utils.furthest_point_sample(xyz, npoint)
        if self.sample_method in ('rs', 'sequence'):
            # the same (B, npoint) index table every call: built once per shape (callers only read it)
            cache = self.__dict__.setdefault('_sequence_cache', {})
            key = (xyz.size(0), int(npoint), xyz.device)
            if key not in cache:
                cache[key] = torch.arange(npoint, dtype=torch.int32, device=xyz.device).repeat(xyz.size(0), 1)
            return cache[key]
        if self.sample_method == 'ffps':
            raise NotImplementedError("sample_method 'ffps' needs furthest_point_sampling_with_dist, which the "
                                      "reference's extension never provided (tools/cfgs/kitti_models/ptt.yaml:42)")
        raise NotImplementedError(self.sample_method)

    # ------------------------------------------------------------------ fused-path parameters
    def _fusable(self, xyz, features):
        """Eval mode on a HIP device, a shape ptt_sa_fused_fwd_f32 instantiates, and no autograd graph being recorded
        (the kernels have none: with gradients enabled and anything requiring them the call stays on the
        differentiable stock-torch path, as the reference's eval mode is). An eval-mode HIP call that is turned away
        says so once (ops.note_unfused)."""
        if self.training or not xyz.is_cuda:
            return False
        name = 'PointnetSAModuleVotes(nsample=%s, mlp=%s)' % (self.nsample, [u.conv.weight.shape[0] for u in self.mlp_module
                                                                              if hasattr(u, 'conv')])
        if ops.autograd_recording(self, xyz, features):
            return ops.note_unfused(name, 'autograd is recording (wrap inference in torch.no_grad())')
        if self.sample_uniformly or self.nsample not in (16, 32, 64):
            return ops.note_unfused(name, 'nsample must be 16, 32 or 64 and sample_uniformly off')
        if xyz.dtype != torch.float32 or (features is not None and features.dtype != torch.float32):
            return ops.note_unfused(name, 'inputs must be float32')
        if len(self.mlp_module) > 4:
            return ops.note_unfused(name, 'more than 4 SharedMLP layers')
        for unit in self.mlp_module:
            conv = getattr(unit, 'conv', None)
            if conv is None or conv.kernel_size != (1, 1) or conv.weight.shape[0] % 32 != 0 or conv.weight.shape[0] > 256:
                return ops.note_unfused(name, 'layer widths must be multiples of 32, at most 256, 1x1 convolutions')
            if list(unit._modules.keys())[0] != 'conv':     # pre-activation units are not folded
                return ops.note_unfused(name, 'pre-activation units')
        return True

    def train(self, mode=True):
        # a train-mode forward updates the BatchNorm running statistics through raw pointers on some torch builds
        # (no _version bump): drop the folded / packed parameters whenever the mode changes
        self._fused_cache = None
        return super().train(mode)

    def _fused_params(self, device):
        tensors = []
        for unit in self.mlp_module:
            tensors.append(unit.conv.weight)
            if unit.conv.bias is not None:
                tensors.append(unit.conv.bias)
            if hasattr(unit, 'normlayer'):
                bn = unit.normlayer.bn
                tensors += [bn.weight, bn.bias, bn.running_mean, bn.running_var]
                if bn.num_batches_tracked is not None:
                    tensors.append(bn.num_batches_tracked)     # bumped by every train-mode forward
        key = (str(device),) + tuple((t.data_ptr(), t._version) for t in tensors)
        if self._fused_cache is not None and self._fused_cache[0] == key:
            return self._fused_cache[1]
        layers = []
        with torch.no_grad():
            for li, unit in enumerate(self.mlp_module):
                w = unit.conv.weight
                cout, cin = w.shape[0], w.shape[1]
                rot = 3 if (li == 0 and self.use_xyz and cin > 3) else 0     # kernel row layout is [features | xyz]
                scale = shift = None
                if hasattr(unit, 'normlayer'):
                    bn = unit.normlayer.bn
                    scale = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).float().contiguous()
                    shift = (bn.bias - bn.running_mean * scale).float().contiguous()
                    if unit.conv.bias is not None:
                        shift = (shift + unit.conv.bias * scale).contiguous()
                elif unit.conv.bias is not None:
                    shift = unit.conv.bias.detach().float().contiguous()
                # the BatchNorm scale is folded into the packed weights: the kernels then start their accumulators at
                # `shift` and the epilogue is a bare ReLU (every vector-ALU instruction outside the MFMA loop costs
                # matrix time on gfx950). raw_scale stays available for the hoisted layer 0.
                wq = w if scale is None else w * scale.view(-1, 1, 1, 1)
                layers.append((ops.pack_weight(wq, rot), None, shift, cin, cout, hasattr(unit, 'activation'), scale))
            # Layer 0 is linear in [rel ; f_n]: its feature half is evaluated once per POINT (N rows on the linear
            # kernel) instead of once per (centre, neighbour) row; the kernel adds the 3 relative-coordinate terms.
            hoist = None
            w0 = self.mlp_module[0].conv.weight
            if (self.use_xyz and len(layers) >= 2 and w0.shape[1] > 3 and layers[0][4] <= 256
                    and os.environ.get('PTT_SA_HOIST', '1') != '0'):
                w2 = w0.reshape(w0.shape[0], w0.shape[1]).float()
                scale0 = layers[0][6]
                wx = w2[:, 0:3] if scale0 is None else w2[:, 0:3] * scale0[:, None]
                # the BatchNorm scale folded into both halves of layer 0 (no per-value scale in the linear launch's epilogue)
                wf = w2[:, 3:] if scale0 is None else w2[:, 3:] * scale0[:, None]
                hoist = (ops.pack_weight(wf.contiguous()), wx.t().contiguous(), layers[0][4], layers[0][5])
        ops.publish_params(device)
        self._fused_cache = (key, (layers, hoist))
        return layers, hoist

    # ------------------------------------------------------------------ forward (reference :57-90)
    def forward(self, xyz: torch.Tensor, features: torch.Tensor, npoint: int, inds: torch.Tensor = None, pre=None):
        """`pre` (an extension of the reference signature, eval mode on a HIP device only): (new_xyz, idx, inds64) of this level
        already computed by the caller — the backbone forms the ball queries of all three levels of a branch in one launch
        when it runs one tracklet frame (ops.sa_levels_point_jobs)."""
        fused = self._fusable(xyz, features)
        if pre is not None and not fused:
            pre = None
        self.centres_knn = None
        if (fused and pre is None and inds is None and self.sample_method == 'fps' and xyz.shape[1] <= 256 and npoint <= 128
                and xyz.shape[0] * npoint * self.nsample <= ops.ONE_FRAME_MAX_SA_ROWS and self.centres_knn_k <= npoint):
            # a handful of frames (vote_aggregation at one tracklet frame): sampling, centre selection, ball query and the
            # centres' kNN in ONE launch (ptt_fps_ball_knn_f32) instead of three
            xyz = xyz.contiguous()
            inds, inds64, new_xyz, idx, self.centres_knn = ops.fps_ball_knn(xyz, npoint, self.radius, self.nsample, self.centres_knn_k)
            pre = (new_xyz, idx, inds64)
        prefix = inds is None and self.sample_method in ('rs', 'sequence')   # centres = the first npoint points
        if pre is not None:
            pass
        elif fused and prefix:
            # 'sequence' indices are constants of (B, npoint): build them once, not four tiny kernels per call
            key = (xyz.size(0), npoint, str(xyz.device))
            if self._arange_cache is None:
                self._arange_cache = {}
            if key not in self._arange_cache:          # search and template branches alternate (B, npoint)
                self._arange_cache[key] = torch.arange(npoint, dtype=torch.int64, device=xyz.device).repeat(
                    xyz.size(0), 1)
                ops.publish_params(xyz.device)
            inds64 = self._arange_cache[key]
        elif inds is None:
            inds = self._sample(xyz, features, npoint)
        else:
            assert inds.shape[1] == npoint
            inds = inds.to(torch.int32)

        if fused:
            xyz = xyz.contiguous()
            if pre is not None:
                new_xyz, idx, inds64 = pre
            elif prefix:                                 # centres + ball query in one launch
                new_xyz, _, idx = ops.centres_ball_query(xyz, None, npoint, self.radius, self.nsample)
            else:
                new_xyz, inds64, idx = ops.centres_ball_query(xyz, inds.contiguous(), npoint, self.radius, self.nsample)
            layers, hoist = self._fused_params(xyz.device)
            if hoist is not None and features is not None:
                wf_packed, wx, c0, relu0 = hoist
                rows = features.transpose(1, 2)                       # (B,N,C): contiguous when point-major
                # rows with a padded stride (the one-frame head hands over 260-float rows) are read in place
                uniform = rows.stride(2) == 1 and rows.stride(0) == rows.shape[1] * rows.stride(1)
                term = ops.linear(rows if uniform else rows.contiguous(), wf_packed, c0, None, layers[0][2], relu=False)
                B, M, ns = xyz.shape[0], npoint, self.nsample
                if (B * M * ns <= ops.ONE_FRAME_MAX_SA_ROWS and ns in (16, 32) and len(layers) == 3 and c0 % 4 == 0 and c0 >= 192):
                    # a handful of frames (vote_aggregation at one tracklet frame: 64 centres x 16 neighbours): the fused SA
                    # kernel would be 16 workgroups with two chained 256 x 256 layers each (41 us); as two row-job launches
                    # the 1024 grouped rows spread over the chip: layer 1 on rows built while its A tile is staged (term[idx]
                    # + Wx . rel, ReLU), then layer 2 with the max over the neighbours as its epilogue
                    L1, L2 = layers[1], layers[2]
                    h = torch.empty((B * M * ns, L1[4]), dtype=torch.float32, device=xyz.device)
                    ops.row_jobs([ops.row_job(L1[0], L1[4], prologue=3, x=term, idx=idx, xyz=xyz, centres=new_xyz, wx=wx,
                                              radius=self.radius, ns=ns, M=M, N=xyz.shape[1], normalize_xyz=self.normalize_xyz,
                                              pro_relu=relu0, scale=L1[1], shift=L1[2], act=1 if L1[5] else 0, out=h)])
                    pooled = torch.empty((B, M, L2[4]), dtype=torch.float32, device=xyz.device)
                    ops.row_jobs([ops.row_job(L2[0], L2[4], x=h, epilogue=2, ns=ns, M=M, scale=L2[1], shift=L2[2],
                                              act=1 if L2[5] else 0, out=pooled)])
                    return new_xyz, pooled.transpose(1, 2), inds64
                new_features = ops.sa_fused_forward(xyz, new_xyz, idx, None, [L[:6] for L in layers[1:]], self.radius, True,
                                                    self.normalize_xyz, point_major_out=True, l0=(term, wx, relu0))
            else:
                new_features = ops.sa_fused_forward(xyz, new_xyz, idx, features, [L[:6] for L in layers], self.radius,
                                                    self.use_xyz, self.normalize_xyz, point_major_out=True)
            return new_xyz, new_features, inds64

        hoist = (self.use_xyz and not self.sample_uniformly and self.nsample * npoint <= 16384
                 and train_ops.usable(self.mlp_module, xyz if features is None else features)
                 and self.mlp_module[0].conv.weight.shape[0] % 4 == 0)
        if hoist and not xyz.requires_grad and inds.dtype == torch.int32:
            # training mode on a HIP device, fixed coordinates (the backbone): centres + ball query in one launch, then
            # layer 0 per (centre, neighbour) row in one more (train_ops.sa_level_hoisted -> ptt_sa_z0_rows_f32) — also for
            # the level without point features, whose first convolution is the three coordinate channels alone
            new_xyz, inds64, idx = ops.centres_ball_query(xyz.contiguous(), inds.contiguous(), npoint, self.radius, self.nsample)
            y = train_ops.sa_level_hoisted(xyz, new_xyz, features, idx, self.mlp_module, self.radius, self.normalize_xyz)
            return new_xyz, y, inds64
        xyz_flipped = xyz.transpose(1, 2).contiguous()
        new_xyz = pointnet2_utils.gather_operation(xyz_flipped, inds).transpose(1, 2).contiguous()
        if hoist and features is not None:
            # learnable coordinates (the box head's vote aggregation): layer 0 hoisted to one row per POINT, gradients to
            # the coordinates through the grouping / gather operators
            idx = pointnet2_utils.ball_query(self.radius, self.nsample, xyz, new_xyz)
            y = train_ops.sa_level_hoisted(xyz, new_xyz, features, idx, self.mlp_module, self.radius, self.normalize_xyz)
            return new_xyz, y, inds.to(torch.int64)
        grouped_features, grouped_xyz = self.grouper(xyz, new_xyz, features)      # (B,C,M,ns)
        if train_ops.usable(self.mlp_module, grouped_features):
            # training mode on a HIP device: SharedMLP (batch-statistics BatchNorm) + the max over the neighbours, forward
            # and backward, on the hand-written row kernels (ptt_amd/train_ops.py)
            return new_xyz, train_ops.shared_mlp_pool(grouped_features, self.mlp_module, pool_dim=3), inds.to(torch.int64)
        y = self.mlp_module(grouped_features)
        # reference: F.max_pool2d(y, kernel_size=[1, nsample]).squeeze(-1) (:85-88); max over the last axis routes the
        # gradient to one arg-max the same way and avoids torch's NCHW pooling kernel (6 ms per training step here)
        y = y.max(dim=3)[0]                                                       # (B,Cout,M)
        return new_xyz, y, inds.to(torch.int64)
