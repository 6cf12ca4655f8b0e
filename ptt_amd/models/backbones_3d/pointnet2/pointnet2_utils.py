"""Operator-level mirror of ptt/models/backbones_3d/pointnet2/pointnet2_utils.py.

Same public names, argument order, dtypes and autograd behaviour as the reference wrappers
(FurthestPointSampling :58-85, GatherOperation :88-122, GroupingOperation :214-262,
BallQuery :265-294, QueryAndGroup :297-380, GroupAll :383-429), but every op lands in
libptt_hip.so (gfx950 HIP) through ptt_amd.ops instead of the CUDA-only pointnet2_ops._ext.
"""
import torch
import torch.nn as nn
from torch.autograd import Function

from .... import ops


class _FurthestPointSampling(Function):
    @staticmethod
    def forward(ctx, xyz, npoint):
        return ops.furthest_point_sampling(xyz.contiguous(), npoint)

    @staticmethod
    def backward(ctx, grad=None):
        return None, None


class _GatherOperation(Function):
    @staticmethod
    def forward(ctx, features, idx):
        ctx.n = features.shape[2]
        ctx.save_for_backward(idx)
        return ops.gather_points(features.contiguous(), idx.contiguous())

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        return ops.gather_points_grad(grad_out.contiguous(), idx, ctx.n), None


class _GroupingOperation(Function):
    @staticmethod
    def forward(ctx, features, idx):
        ctx.n = features.shape[2]
        ctx.save_for_backward(idx)
        return ops.group_points(features.contiguous(), idx.contiguous())

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        return ops.group_points_grad(grad_out.contiguous(), idx, ctx.n), None


class _BallQuery(Function):
    @staticmethod
    def forward(ctx, radius, nsample, xyz, new_xyz):
        return ops.ball_query(new_xyz.contiguous(), xyz.contiguous(), radius, nsample)

    @staticmethod
    def backward(ctx, grad=None):
        return None, None, None, None


def _not_reached(name):
    def fn(*args, **kwargs):
        raise NotImplementedError(
            "%s: defined by the reference wrappers but never called anywhere in PTT; not provided" % name)
    return fn


furthest_point_sample = _FurthestPointSampling.apply          # (xyz (B,N,3), npoint) -> (B,npoint) int32
gather_operation = _GatherOperation.apply                      # (features (B,C,N), idx (B,M) int32) -> (B,C,M)
grouping_operation = _GroupingOperation.apply                  # (features (B,C,N), idx (B,M,ns) int32) -> (B,C,M,ns)
ball_query = _BallQuery.apply                                  # (radius, nsample, xyz, new_xyz) -> (B,M,ns) int32
furthest_point_sampling_with_dist = _not_reached("furthest_point_sampling_with_dist")
three_nn = _not_reached("three_nn")
three_interpolate = _not_reached("three_interpolate")


class QueryAndGroup(nn.Module):
    """Ball query + grouping + centre subtraction (+ /radius) + concat, as the reference's
    QueryAndGroup.forward (:320-380). This is the UNFUSED path (training, or callers that want
    the grouped tensor); PointnetSAModuleVotes uses the fused kernel in eval mode instead."""

    def __init__(self, radius, nsample, use_xyz=True, ret_grouped_xyz=False, normalize_xyz=False,
                 sample_uniformly=False, ret_unique_cnt=False):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz
        self.ret_grouped_xyz = ret_grouped_xyz
        self.normalize_xyz = normalize_xyz
        self.sample_uniformly = sample_uniformly
        self.ret_unique_cnt = ret_unique_cnt
        if ret_unique_cnt:
            assert sample_uniformly

    def _resample_uniformly(self, idx):
        # reference :339-348 — host-side loop, off in every shipped config
        unique_cnt = torch.zeros((idx.shape[0], idx.shape[1]))
        for b in range(idx.shape[0]):
            for r in range(idx.shape[1]):
                uniq = torch.unique(idx[b, r, :])
                n = uniq.shape[0]
                unique_cnt[b, r] = n
                pick = torch.randint(0, n, (self.nsample - n,), dtype=torch.long, device=uniq.device)
                idx[b, r, :] = torch.cat((uniq, uniq[pick]))
        return idx, unique_cnt

    def forward(self, xyz, new_xyz, features=None, return_idx=False):
        idx = ball_query(self.radius, self.nsample, xyz, new_xyz)
        unique_cnt = None
        if self.sample_uniformly:
            idx, unique_cnt = self._resample_uniformly(idx)

        grouped_xyz = grouping_operation(xyz.transpose(1, 2).contiguous(), idx)   # (B,3,M,ns)
        grouped_xyz -= new_xyz.transpose(1, 2).unsqueeze(-1)
        if self.normalize_xyz:
            grouped_xyz /= self.radius

        if features is not None:
            grouped_features = grouping_operation(features, idx)
            new_features = torch.cat([grouped_xyz, grouped_features], dim=1) if self.use_xyz else grouped_features
        else:
            assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
            new_features = grouped_xyz

        ret = [new_features]
        if self.ret_grouped_xyz:
            ret.append(grouped_xyz)
        if self.ret_unique_cnt:
            ret.append(unique_cnt)
        if len(ret) == 1:
            return ret[0]
        if return_idx:
            ret.append(idx)
        return tuple(ret)


class GroupAll(nn.Module):
    """Reference :383-429 — groups every point into one region (pure torch views)."""

    def __init__(self, use_xyz=True, ret_grouped_xyz=False):
        super().__init__()
        self.use_xyz = use_xyz
        self.ret_grouped_xyz = ret_grouped_xyz

    def forward(self, xyz, new_xyz, features=None):
        grouped_xyz = xyz.transpose(1, 2).unsqueeze(2)
        if features is not None:
            grouped_features = features.unsqueeze(2)
            new_features = torch.cat([grouped_xyz, grouped_features], dim=1) if self.use_xyz else grouped_features
        else:
            new_features = grouped_xyz
        return (new_features, grouped_xyz) if self.ret_grouped_xyz else new_features
