"""Mirror of ptt/models/voting_heads/box_voting_head.py: BoxVotingHead (:10-112).
Vote aggregation (a 4th set-abstraction level on the 128 votes: FPS 128->64, r=.3, ns=16, MLP [260,256,256,256]),
the second Point-Track-Transformer block and a small Conv1d stack that regresses (dx,dy,dz,dtheta,score) per
proposal. `vote_aggregation`, `refine_layer`, `transformer_block` are the checkpoint names."""
import torch

from ... import ops, train_ops
from ..backbones_3d.pointnet2 import pointnet2_modules
from ..backbones_3d.pointnet2 import pytorch_utils as layer_utils
from ..transformer_block import build_transformer
from .voting_head_template import VotingHeadTemplate


class BoxVotingHead(VotingHeadTemplate):
    def __init__(self, model_cfg, **kwargs):
        super().__init__(model_cfg)
        sa = self.model_cfg.SA_CONFIG
        self.vote_aggregation = pointnet2_modules.PointnetSAModuleVotes(
            radius=sa.RADIUS, nsample=sa.NSAMPLE, mlp=sa.MLPS,          # the cfg list itself, mutated like the reference (:19)
            use_xyz=sa.get('USE_XYZ', True), normalize_xyz=sa.get('NORMALIZE_XYZ', True),
            sample_method=sa.SAMPLE_METHOD)
        fc = self.model_cfg.FC
        self.refine_layer = (layer_utils.Seq(fc[0])
                             .conv1d(fc[1], bn=True)
                             .conv1d(fc[2], bn=True)
                             .conv1d(fc[3], activation=None))
        if self.model_cfg.TRANSFORMER_BLOCK.ENABLE:
            self.transformer_block = build_transformer(self.model_cfg.TRANSFORMER_BLOCK)
            self.vote_aggregation.centres_knn_k = self.transformer_block.k     # the proposals' kNN beside their sampling (one frame)

    # ------------------------------------------------------------------ losses (reference :33-66)
    def get_cls_layer_loss(self, forward_ret_dict):
        weights = self.model_cfg.LOSS_CONFIG.LOSS_WEIGHTS
        forward_ret_dict = self._proposal_labels(forward_ret_dict)
        mask = forward_ret_dict['mask']
        loss = self.cls_loss_func(forward_ret_dict['pred_boxes_cls'], forward_ret_dict['cls_label'])
        loss = torch.sum(loss * mask) / (torch.sum(mask) + 1e-6)
        tb_dict = {'boxes_cls_loss': loss.item()}
        return loss.float() * weights['boxes_cls_weight'], tb_dict

    def get_reg_layer_loss(self, forward_ret_dict):
        weights = self.model_cfg.LOSS_CONFIG.LOSS_WEIGHTS
        mask = self._proposal_labels(forward_ret_dict)['cls_label']
        pred = forward_ret_dict['pred_boxes_reg']
        target = forward_ret_dict['reg_label'][:, None, :].expand_as(pred)
        loss = self.reg_loss_func(pred, target)
        loss = (loss.mean(2) * mask).sum() / (mask.sum() + 1e-06)
        tb_dict = {'boxes_reg_loss': loss.item()}
        return loss.float() * weights['boxes_reg_weight'], tb_dict

    def get_loss(self, tb_dict=None):
        tb_dict = {} if tb_dict is None else tb_dict
        loss_cls, tb1 = self.get_cls_layer_loss(self.forward_ret_dict)
        loss_reg, tb2 = self.get_reg_layer_loss(self.forward_ret_dict)
        tb_dict.update(tb1)
        tb_dict.update(tb2)
        return (loss_cls + loss_reg).float(), tb_dict

    # ------------------------------------------------------------------ forward (reference :68-112)
    def _fusable(self, feats):
        if self.training:
            return train_ops.conv1d_stack_usable(self.refine_layer, feats)
        return layer_utils.rows_fusable(self.refine_layer, feats)

    def _one_frame(self, rows):
        return (rows.shape[0] * rows.shape[1] <= ops.ONE_FRAME_MAX_POINTS and len(self.refine_layer) >= 1
                and self.refine_layer[-1].conv.weight.shape[0] >= 3)

    def _train_labels(self, batch_dict, centres):
        """What the losses need (reference :96-110). The proposal labels and the score / box slices are formed when a loss asks
        for them (_proposal_labels): the one-launch loss path (train_ops.track_losses) forms them inside its kernel."""
        self.forward_ret_dict = {'pred_box_data': batch_dict['pred_box_data'], 'centres': centres, 'reg_label': batch_dict['reg_label']}

    @staticmethod
    def _proposal_labels(d):
        if 'mask' not in d:
            dist = torch.sqrt(torch.sum((d['centres'] - d['reg_label'][:, None, 0:3]) ** 2, dim=-1) + 1e-6)
            label = torch.zeros_like(dist, dtype=torch.float)
            mask = torch.zeros_like(label, dtype=torch.float)
            label[dist < 0.3] = 1
            mask[dist < 0.3] = 1
            mask[dist > 0.6] = 1
            d.update({'pred_boxes_cls': d['pred_box_data'][:, :, -1], 'pred_boxes_reg': d['pred_box_data'][:, :, :-1],
                      'mask': mask, 'cls_label': label})
        return d

    def forward(self, batch_dict):
        centres, feats, _ = self.vote_aggregation(xyz=batch_dict['pred_centroids_votes'],
                                                  features=batch_dict['votes_feats'],
                                                  npoint=self.model_cfg.SA_CONFIG.NPOINTS)
        if self._fusable(feats):
            # on a HIP device: stay on point-major rows (the SA output already is); eval: one folded-BN linear launch per
            # refine layer; training: the row kernels of ptt_amd/train_ops.py. The (B,M,5) result is produced directly in
            # the layout the caller wants
            rows = feats.transpose(1, 2)                                                         # (B,M,C)
            if hasattr(self, 'transformer_block'):
                rows = self.transformer_block(xyz=centres, features=rows.contiguous(), knn=self.vote_aggregation.centres_knn, want_attn=False)[0]
            if not self.training and self._one_frame(rows):
                # a handful of frames: one ptt_row_jobs_f32 launch per refine convolution, the last one adding the proposal
                # centres to its first three columns (reference :91) and writing pred_box_data directly
                L = layer_utils.rows_layers(self.refine_layer)
                B, M, _ = rows.shape
                x = rows.contiguous()
                for wp, cout, scale, shift, relu in L[:-1]:
                    h = torch.empty((B * M, cout), dtype=torch.float32, device=rows.device)
                    ops.row_jobs([ops.row_job(wp, cout, x=x, scale=scale, shift=shift, act=1 if relu else 0, out=h)])
                    x = h
                wp, cout, scale, shift, relu = L[-1]
                # pred_box_out (set by a driver: ptt_amd.tracklet_runner): a preallocated (B,M,5) buffer to write the proposals
                # into, so that its one read-back per frame needs no gathering copy
                boxes = getattr(self, 'pred_box_out', None)
                if boxes is None or tuple(boxes.shape) != (B, M, cout) or boxes.device != rows.device:
                    boxes = torch.empty((B, M, cout), dtype=torch.float32, device=rows.device)
                ops.row_jobs([ops.row_job(wp, cout, x=x, scale=scale, shift=shift, act=1 if relu else 0, res2=centres.contiguous(),
                                          res_split=3, out=boxes)])
                batch_dict['pred_box_center'] = centres
                batch_dict['pred_box_data'] = boxes
                return batch_dict
            if self.training:
                offsets = train_ops.conv1d_stack_rows(self.refine_layer, rows)                   # (B,M,5)
            else:
                offsets = layer_utils.rows_forward(self.refine_layer, rows)                      # (B,M,5)
            batch_dict['pred_box_center'] = centres
            off_xyz, off_rest = offsets.split([3, offsets.shape[-1] - 3], dim=2)                 # one split: its backward is one concatenation
            batch_dict['pred_box_data'] = torch.cat((off_xyz + centres, off_rest), dim=2)
            if self.training:
                self._train_labels(batch_dict, centres)
            return batch_dict
        if hasattr(self, 'transformer_block'):
            fused = self.transformer_block(xyz=centres, features=feats.transpose(1, 2).contiguous(), want_attn=False)[0]
            feats = fused.transpose(1, 2).contiguous()

        offsets = self.refine_layer(feats)                                                       # (B,5,M)
        boxes = torch.cat((offsets[:, 0:3, :] + centres.transpose(1, 2).contiguous(), offsets[:, 3:, :]), dim=1)
        batch_dict['pred_box_center'] = centres
        batch_dict['pred_box_data'] = boxes.transpose(1, 2).contiguous()                         # (B,M,5)

        if self.training:
            self._train_labels(batch_dict, centres)
        return batch_dict
