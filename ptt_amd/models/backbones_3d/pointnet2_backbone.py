"""Mirror of ptt/models/backbones_3d/pointnet2_backbone.py: PointNet2BackboneLight (:8-67).

Three set-abstraction levels applied to the search and the template cloud with shared
weights, a final 1x1 Conv1d (`cov_final`), and the composition of the per-level sample
indices back to the raw cloud. Attribute names (`SA_modules`, `cov_final`,
`num_point_features`) and the batch_dict key contract are the reference's. In eval mode on a
HIP device `cov_final` runs on the fp32-MFMA linear kernel straight from the point-major
features the last SA level produced.
"""
from typing import List

import torch
import torch.nn as nn

from ... import graph_policy, ops, train_ops
from .pointnet2 import pointnet2_modules


class PointNet2BackboneLight(nn.Module):
    def __init__(self, model_cfg, input_channels, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        input_channels -= 3
        self.num_points_each_layer = []
        self.SA_modules = nn.ModuleList()
        sa = self.model_cfg.SA_CONFIG
        for k in range(len(sa.RADIUS)):
            mlps = list(sa.MLPS[k])
            if k == 0:
                mlps[0] = input_channels
            self.SA_modules.append(pointnet2_modules.PointnetSAModuleVotes(
                radius=sa.RADIUS[k], nsample=sa.NSAMPLE[k], mlp=mlps,
                use_xyz=sa.get('USE_XYZ', True), normalize_xyz=sa.get('NORMALIZE_XYZ', True),
                sample_method=sa.SAMPLE_METHOD[k]))
        self.cov_final = nn.Conv1d(256, 256, kernel_size=1)
        self.num_point_features = sa.MLPS[-1][-1]
        self._cov_cache = None
        self.seed_knn = 16          # neighbours of the seeds' kNN formed beside the ball queries at one frame (0: not formed);
        #                             the tracker sets it to its centroid head's transformer's k
        self.overlap_branches = True
        self._side_stream = None

    @staticmethod
    def _break_up_pc(pc):
        xyz = pc[..., 0:3].contiguous()
        features = pc[..., 3:].contiguous() if pc.size(-1) > 3 else None   # as the reference (:38)
        return xyz, features

    def _cov_final(self, features):
        """features (B,256,M). Eval + HIP + point-major storage: MFMA linear; else stock Conv1d."""
        if not features.is_cuda:
            return self.cov_final(features)
        if self.training:
            if features.dtype != torch.float32:
                return self.cov_final(features)
            # training on a HIP device: the 1x1 convolution over point-major rows on the row kernels (forward, input and
            # weight gradient: train_ops._RowsLinear); the SA level's output already is (B,M,C) storage
            return train_ops.rows_linear(self.cov_final, features.transpose(1, 2)).transpose(1, 2)
        if ops.autograd_recording(self.cov_final, features) or features.dtype != torch.float32:
            ops.note_unfused('PointNet2BackboneLight.cov_final', 'autograd is recording or input is not float32')
            return self.cov_final(features)
        w, b = self.cov_final.weight, self.cov_final.bias
        key = (w.data_ptr(), w._version, b.data_ptr(), b._version)
        if self._cov_cache is None or self._cov_cache[0] != key:
            self._cov_cache = (key, ops.pack_weight(w), b.detach().float().contiguous())
            ops.publish_params(w.device)
        rows = features.transpose(1, 2)                        # (B,M,C); contiguous when point-major
        out = ops.linear(rows, self._cov_cache[1], w.shape[0], None, self._cov_cache[2])
        return out.transpose(1, 2)                              # (B,C,M) view

    def _arange64(self, xyz, n):
        key = (xyz.size(0), n, str(xyz.device))
        cache = self.__dict__.setdefault('_arange_cache', {})
        if key not in cache:
            cache[key] = torch.arange(n, dtype=torch.int64, device=xyz.device).repeat(xyz.size(0), 1)
            ops.publish_params(xyz.device)
        return cache[key]

    def branch_forward(self, pts, npoints: List, inds0=None, want_knn=0):
        """`inds0`: optional precomputed level-0 sample indices (B, npoints[0]) — the SA module accepts
        caller-supplied indices exactly as the reference's does (pointnet2_modules.py:60,76-77); a pipelined
        driver computes them for batch n+1 while batch n is in the dense kernels."""
        xyz, features = self._break_up_pc(pts)
        sa = self.model_cfg.SA_CONFIG
        one_frame = (not self.training and xyz.is_cuda and features is None and len(self.SA_modules) == 3
                     and xyz.shape[0] * npoints[0] <= 4 * ops.ONE_FRAME_MAX_POINTS
                     and self.SA_modules[0].sample_method == 'fps'
                     and all(m.sample_method in ('rs', 'sequence') for m in self.SA_modules[1:])
                     and npoints[0] >= npoints[1] >= npoints[2] and (not want_knn or npoints[2] <= 128)
                     and all(m._fusable(xyz, None) for m in self.SA_modules[:1]))
        knn = None
        if one_frame:
            # one tracklet frame: the three ball queries (and the kNN of the seeds, which the centroid head's transformer
            # needs) in ONE launch right behind the level-0 sampling — with 'sequence' sampling every level's centres and
            # points are prefixes of the level-0 sample, so nothing waits for the level below (ops.sa_levels_point_jobs)
            from .pointnet2 import pointnet2_utils
            xyz = xyz.contiguous()
            if inds0 is None:
                inds0 = pointnet2_utils.furthest_point_sample(xyz, npoints[0])
            inds0 = inds0.to(torch.int32).contiguous()
            levels, inds64, knn = ops.sa_levels_point_jobs(xyz, inds0, list(npoints), list(sa.RADIUS), list(sa.NSAMPLE),
                                                           knn_k=want_knn)
            seq = [None, self._arange64(xyz, npoints[1]), self._arange64(xyz, npoints[2])]
            xyz, features, inds0 = self.SA_modules[0](xyz=xyz, features=features, npoint=npoints[0], inds=inds0,
                                                      pre=(levels[0][0], levels[0][1], inds64))
            xyz, features, inds1 = self.SA_modules[1](xyz=xyz, features=features, npoint=npoints[1],
                                                      pre=(levels[1][0], levels[1][1], seq[1]))
            xyz, features, inds2 = self.SA_modules[2](xyz=xyz, features=features, npoint=npoints[2],
                                                      pre=(levels[2][0], levels[2][1], seq[2]))
        else:
            xyz, features, inds0 = self.SA_modules[0](xyz=xyz, features=features, npoint=npoints[0], inds=inds0)
            xyz, features, inds1 = self.SA_modules[1](xyz=xyz, features=features, npoint=npoints[1])
            xyz, features, inds2 = self.SA_modules[2](xyz=xyz, features=features, npoint=npoints[2])
        point_features = self._cov_final(features)
        assert inds1.dtype == inds2.dtype == torch.int64, 'index type must be int64, not {}'.format(inds2.dtype)
        if all(m.sample_method in ('rs', 'sequence') for m in self.SA_modules[1:]) and inds0.shape[1] >= npoints[2]:
            # levels 1 and 2 take the FIRST npoints of the level below: the composition is a prefix of level 0's indices
            # (same values as the two gathers of the reference, :48, without their launches)
            inds = inds0[:, :npoints[2]]
        else:
            inds = inds0.gather(1, inds1).gather(1, inds2)
        if want_knn:
            return xyz, point_features, inds, knn
        return xyz, point_features, inds

    def sample(self, search_points, template_points):
        """Level-0 furthest point sampling of both clouds -> (inds_search, inds_template) int32: the stage a throughput
        driver runs for the NEXT batch on a side stream (ptt_amd.hot_path.PipelinedHotPath)."""
        from .pointnet2 import pointnet2_utils
        sa = self.model_cfg.SA_CONFIG
        assert sa.SAMPLE_METHOD[0] == 'fps'
        return (pointnet2_utils.furthest_point_sample(search_points[..., 0:3].contiguous(), sa.NPOINTS_SEARCH[0]),
                pointnet2_utils.furthest_point_sample(template_points[..., 0:3].contiguous(), sa.NPOINTS_TEMPLATE[0]))

    def forward_branches(self, search_points, template_points, inds=None):
        """Both branches -> the six batch_dict entries of the reference's forward (:56-63). Eval mode on a HIP device:
        the template branch runs on a second stream — FPS is a latency-bound chain on B workgroups, so each branch's
        kernels fill the CUs the other's FPS leaves idle. `inds` = optional (search, template) level-0 FPS indices."""
        sa = self.model_cfg.SA_CONFIG
        i_s, i_t = inds if inds is not None else (None, None)
        if not (self.overlap_branches and search_points.is_cuda and not self.training):
            r = self.branch_forward(search_points, sa.NPOINTS_SEARCH, i_s, want_knn=self.seed_knn)
            t_seeds, t_feats, t_inds = self.branch_forward(template_points, sa.NPOINTS_TEMPLATE, i_t)
        else:
            if self._side_stream is None or self._side_stream.device != search_points.device:
                self._side_stream = torch.cuda.Stream(device=search_points.device)
            main = torch.cuda.current_stream(search_points.device)
            side = graph_policy.branch(main, self._side_stream)       # inside a capture the policy may keep the branch in line
            with torch.cuda.stream(side):
                t_seeds, t_feats, t_inds = self.branch_forward(template_points, sa.NPOINTS_TEMPLATE, i_t)
            r = self.branch_forward(search_points, sa.NPOINTS_SEARCH, i_s, want_knn=self.seed_knn)
            graph_policy.join(main, side)
            if side is not main:
                for t in (t_seeds, t_feats, t_inds, template_points):
                    t.record_stream(main)
        s_seeds, s_feats, s_inds = r[:3]
        s_knn = r[3] if len(r) > 3 else None
        out = {'search_seeds': s_seeds, 'search_feats': s_feats, 'search_inds': s_inds,
               'template_seeds': t_seeds, 'template_feats': t_feats, 'template_inds': t_inds}
        if s_knn is not None:
            out['search_seeds_knn'] = s_knn       # extension of the key contract: (knn_idx, rel) of the seeds for the centroid head
        return out

    def forward(self, batch_dict):
        # 'fps_inds' is an extension of the key contract: level-0 sample indices computed ahead by a pipelined driver
        batch_dict.update(self.forward_branches(batch_dict['search_points'], batch_dict['template_points'],
                                                batch_dict.pop('fps_inds', None)))
        batch_dict.pop('search_points')
        batch_dict.pop('template_points')
        return batch_dict
