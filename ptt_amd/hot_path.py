"""The per-frame hot path as one module: what bench.py times and smoke() checks.

One *frame* = one (search, template) cloud pair through
    PointNet2BackboneLight.forward   (3 SA levels x 2 branches + cov_final, pointnet2_backbone.py:52-67)
 -> TransformerBlock on the 128 search seeds          (centroids_voting_head.py:71-76)
 -> vote_aggregation SA level 128 -> 64, r=.3, ns=16  (box_voting_head.py:75-79)
 -> TransformerBlock on the 64 proposals              (box_voting_head.py:81-86)
with the constants of tools/cfgs/kitti_models/ptt.yaml:41-51,72-79,96-112.
The small Conv1d heads and CosineSimAug that sit between these stages in the full tracker are
outside the hot path; `bridge()` stands in for them with the same tensor shapes (votes = seeds, votes_feats =
cat(score, feats), (B, 1 + C, N) as centroids_voting_head.py:94 — handed over, as this build's head does, as a channel-major
view of point-major rows).
"""
import torch
import torch.nn as nn

from . import graph_policy
from .graph_policy import graph_mode, set_graph_mode          # noqa: F401 — the switch lives here for callers (graph_policy.py)
from .models.backbones_3d.pointnet2.pointnet2_modules import PointnetSAModuleVotes
from .models.backbones_3d.pointnet2_backbone import PointNet2BackboneLight
from .models.transformer_block import build_transformer


class AttrDict(dict):
    """dict with attribute access — the slice of EasyDict behaviour the model code uses."""

    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError:
            raise AttributeError(k)
        return v

    def __setattr__(self, k, v):
        self[k] = v

    @staticmethod
    def wrap(obj):
        if isinstance(obj, dict):
            return AttrDict({k: AttrDict.wrap(v) for k, v in obj.items()})
        return obj


def kitti_model_cfg():
    """MODEL section constants of tools/cfgs/kitti_models/ptt.yaml (nuScenes is identical)."""
    tb = dict(ENABLE=True, NAME='TransformerBlock', DIM_INPUT=256, DIM_MODEL=512, KNN=16, N_HEADS=1, N_LAYERS=1)
    return AttrDict.wrap(dict(
        BACKBONE_3D=dict(NAME='PointNet2BackboneLight', DEBUG=False, SA_CONFIG=dict(
            SAMPLE_METHOD=['fps', 'sequence', 'sequence'], USE_XYZ=True, NORMALIZE_XYZ=True,
            NPOINTS_SEARCH=[512, 256, 128], NPOINTS_TEMPLATE=[256, 128, 64], RADIUS=[0.3, 0.5, 0.7],
            NSAMPLE=[32, 32, 32], MLPS=[[0, 64, 64, 128], [128, 128, 128, 256], [256, 128, 128, 256]])),
        CENTROID_HEAD=dict(TRANSFORMER_BLOCK=dict(tb)),
        BOX_HEAD=dict(SA_CONFIG=dict(NPOINTS=64, RADIUS=0.3, NSAMPLE=16, MLPS=[257, 256, 256, 256], USE_XYZ=True,
                                     NORMALIZE_XYZ=True, SAMPLE_METHOD='fps'),
                      TRANSFORMER_BLOCK=dict(tb)),
    ))


class FrameHotPath(nn.Module):
    def __init__(self, model_cfg=None):
        super().__init__()
        cfg = model_cfg if model_cfg is not None else kitti_model_cfg()
        self.cfg = cfg
        self.backbone_3d = PointNet2BackboneLight(cfg.BACKBONE_3D, input_channels=3)
        self.centroid_transformer = build_transformer(cfg.CENTROID_HEAD.TRANSFORMER_BLOCK)
        sa = cfg.BOX_HEAD.SA_CONFIG
        self.vote_aggregation = PointnetSAModuleVotes(
            radius=sa.RADIUS, nsample=sa.NSAMPLE, mlp=list(sa.MLPS), use_xyz=sa.get('USE_XYZ', True),
            normalize_xyz=sa.get('NORMALIZE_XYZ', True), sample_method=sa.SAMPLE_METHOD)
        self.box_transformer = build_transformer(cfg.BOX_HEAD.TRANSFORMER_BLOCK)
        self.vote_aggregation.centres_knn_k = self.box_transformer.k
        self.npoints_box = sa.NPOINTS
        # FPS is a latency-bound chain on B workgroups; running the template branch on a second HIP
        # stream lets its kernels fill the CUs the search branch's FPS leaves idle (and vice versa).
        self.overlap_branches = True

    @staticmethod
    def bridge(seeds, feats_bnc):
        """Stand-in for the heads between the two transformers: votes (B,128,3), votes_feats (B,257,128) — a channel-major
        VIEW of point-major rows: the SA level that consumes it reads rows, so a channel-major copy here would be
        transposed straight back (two 6 MB copies per step)."""
        score = torch.sigmoid(feats_bnc[:, :, :1])
        return seeds, torch.cat((score, feats_bnc), dim=2).transpose(1, 2)

    def sample(self, search_points, template_points):
        """Level-0 furthest point sampling of both clouds -> (inds_search, inds_template) int32. Split out so
        that a driver can run it for the NEXT batch on a side stream (PipelinedHotPath)."""
        return self.backbone_3d.sample(search_points, template_points)

    def _backbone(self, search_points, template_points, inds=None):
        self.backbone_3d.overlap_branches = self.overlap_branches
        return self.backbone_3d.forward_branches(search_points, template_points, inds)

    def forward(self, search_points, template_points, inds=None):
        d = self._backbone(search_points, template_points, inds)
        seeds = d['search_seeds']
        fused = self.centroid_transformer(xyz=seeds, features=d['search_feats'].transpose(1, 2).contiguous(),
                                          knn=d.pop('search_seeds_knn', None), want_attn=False)[0]
        votes, votes_feats = self.bridge(seeds, fused)
        centres, prop_feats, _ = self.vote_aggregation(xyz=votes, features=votes_feats, npoint=self.npoints_box)
        box_feats = self.box_transformer(xyz=centres, features=prop_feats.transpose(1, 2).contiguous(),
                                         knn=self.vote_aggregation.centres_knn, want_attn=False)[0]
        d['centroid_feats'] = fused
        d['pred_box_center'] = centres
        d['box_feats'] = box_feats
        return d


class TrackerThroughput(object):
    """Adapter giving a full tracker (ptt_amd.models.trackers.PTT) the (sample, forward-with-indices) interface of
    FrameHotPath, so that GraphedHotPath / PipelinedHotPath can drive it:
        pipe = PipelinedHotPath(TrackerThroughput(tracker), search0, template0)"""

    def __init__(self, tracker):
        self.tracker = tracker

    def sample(self, search_points, template_points):
        return self.tracker.backbone_3d.sample(search_points, template_points)

    def __call__(self, search_points, template_points, inds=None):
        batch = {'search_points': search_points, 'template_points': template_points,
                 'batch_size': search_points.shape[0]}
        if inds is not None:
            batch['fps_inds'] = inds
        return self.tracker(batch)


class GraphedHotPath(object):
    """hipGraph replay of one hot-path step for fixed shapes (tracking and benchmarking run the same
    shapes every step): removes the per-kernel host launch cost and the gaps between ~60 dependent
    launches. Inputs are copied into static buffers; outputs are the captured tensors."""

    def __init__(self, model, search_points, template_points, warmup=3):
        """`model`: any callable (search (B,NS,3), template (B,NT,3)) -> outputs, e.g. FrameHotPath or a full tracker
        wrapped as lambda s, t: tracker({'search_points': s, 'template_points': t})."""
        self.model = model
        self.search = search_points.clone()
        self.template = template_points.clone()
        with torch.no_grad():
            s = torch.cuda.Stream(device=search_points.device)
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(warmup):                  # weight packing / LDS attributes happen here, not in capture
                    model(self.search, self.template)
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with graph_policy.capture_scope(), torch.cuda.graph(self.graph):
                self.out = model(self.search, self.template)

    def __call__(self, search_points=None, template_points=None):
        if search_points is not None:
            self.search.copy_(search_points, non_blocking=True)
        if template_points is not None:
            self.template.copy_(template_points, non_blocking=True)
        self.graph.replay()
        return self.out


class PipelinedHotPath(object):
    """Throughput mode: hipGraph replay in which the FPS of batch n+1 (a latency-bound chain that occupies
    only B of the 256 CUs) runs on a side stream while batch n is in the MFMA kernels. Batches are
    independent, all work of every batch is still executed — it is software pipelining across batches, so a
    call returns the outputs of the PREVIOUS call's inputs (`flush()` drains the last one).

        pipe = PipelinedHotPath(model, search0, template0)      # primes the pipeline with batch 0
        out0 = pipe(search1, template1)                         # results of batch 0, FPS of batch 1 in flight
        out1 = pipe(search2, template2) ...
    """

    def __init__(self, model, search_points, template_points, warmup=3):
        self.model = model
        dev = search_points.device
        self.cur = [search_points.clone(), template_points.clone()]       # batch n (dense stage input)
        self.nxt = [search_points.clone(), template_points.clone()]       # batch n+1 (sampling stage input)
        self.side = torch.cuda.Stream(device=dev)
        with torch.no_grad():
            s = torch.cuda.Stream(device=dev)
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(warmup):
                    inds = model.sample(*self.cur)
                    model(self.cur[0], self.cur[1], inds)
                self.inds_cur = [t.clone() for t in model.sample(*self.cur)]
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with graph_policy.capture_scope(), torch.cuda.graph(self.graph):
                main = torch.cuda.current_stream()
                side = graph_policy.branch(main, self.side)                # (the policy may keep the stage in line: graph_policy.py)
                with torch.cuda.stream(side):
                    inds_nxt = model.sample(*self.nxt)                     # stage A of batch n+1
                self.out = model(self.cur[0], self.cur[1], self.inds_cur)  # stage B of batch n
                graph_policy.join(main, side)
                # rotate: what was "next" becomes "current" for the following replay
                for dst, src in zip(self.inds_cur, inds_nxt):
                    dst.copy_(src)
                for dst, src in zip(self.cur, self.nxt):
                    dst.copy_(src)

    def __call__(self, next_search=None, next_template=None):
        if next_search is not None:
            self.nxt[0].copy_(next_search, non_blocking=True)
        if next_template is not None:
            self.nxt[1].copy_(next_template, non_blocking=True)
        self.graph.replay()
        return self.out

    def flush(self):
        return self.__call__()


class InterleavedHotPath(object):
    """Throughput mode with `ways` independent batches in flight: `ways` PipelinedHotPath graphs, each on its own HIP
    stream, replayed round-robin. The tail of one batch's kernels (the last workgroups of a launch, the launch gaps
    between ~55 dependent kernels) is filled by the other batch's kernels; every batch still executes every kernel.
    Measured on the KITTI-Car workload (scripts/probes/dual_pipeline_probe.py): 3.18 -> 3.11 ms per 48-frame step with two
    ways, nothing more with three or four.

        pipe = InterleavedHotPath(model, search0, template0)        # ways = 2
        out = pipe(search_k, template_k)     # enqueues batch k; returns the outputs of batch k - ways (None at first)

    The outputs belong to the stream in `pipe.last_stream`; make the consuming stream wait on it (or synchronise)."""

    def __init__(self, model, search_points, template_points, ways=2, warmup=3):
        dev = search_points.device
        self.ways = int(ways)
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(self.ways)]
        self.pipes = []
        for st in self.streams:
            st.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(st):
                self.pipes.append(PipelinedHotPath(model, search_points, template_points, warmup=warmup))
        torch.cuda.synchronize(dev)
        self.calls = 0
        self.last_stream = self.streams[0]

    def __call__(self, next_search=None, next_template=None):
        k = self.calls % self.ways
        self.calls += 1
        self.last_stream = self.streams[k]
        with torch.cuda.stream(self.streams[k]):
            return self.pipes[k](next_search, next_template)

    def flush(self):
        return [self.__call__() for _ in range(self.ways)]


def randomize_(module, seed=0):
    """Random-init weights incl. non-trivial BatchNorm running statistics (no checkpoints offline)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in module.modules():
            if isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d)):
                m.weight.copy_(torch.empty_like(m.weight, device='cpu').uniform_(0.5, 1.5, generator=g))
                m.bias.copy_(torch.empty_like(m.bias, device='cpu').normal_(0, 0.1, generator=g))
                m.running_mean.copy_(torch.empty_like(m.running_mean, device='cpu').normal_(0, 0.2, generator=g))
                m.running_var.copy_(torch.empty_like(m.running_var, device='cpu').uniform_(0.5, 1.5, generator=g))
            elif isinstance(m, (nn.Conv1d, nn.Conv2d, nn.Linear)):
                fan_in = m.weight[0].numel()
                m.weight.copy_(torch.empty_like(m.weight, device='cpu').normal_(0, (2.0 / fan_in) ** 0.5, generator=g))
                if m.bias is not None:
                    m.bias.copy_(torch.empty_like(m.bias, device='cpu').uniform_(-0.1, 0.1, generator=g))
    return module
