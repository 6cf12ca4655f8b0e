"""Mirror of ptt/models/voting_heads/centroids_voting_head.py: CentroidVotingHead (:9-109).
Seed-wise classification + vote regression on the 128 search seeds, preceded by the Point-Track-Transformer
block (the hot-path kernel sequence of ptt_amd.models.transformer_block). The two 3-layer Conv1d stacks are
tiny (0.05 GFLOP/frame): stock torch layers in training, folded-BN linear launches on point-major rows in eval mode
on a HIP device; names `cla_layer`, `vote_layer`, `transformer_block` are the checkpoint contract."""
import torch

from ... import ops, train_ops
from ..backbones_3d.pointnet2 import pytorch_utils as layer_utils
from ..transformer_block import build_transformer
from .voting_head_template import VotingHeadTemplate


def _conv_stack(channels):
    return (layer_utils.Seq(channels[0])
            .conv1d(channels[1], bn=True)
            .conv1d(channels[2], bn=True)
            .conv1d(channels[3], activation=None))


class CentroidVotingHead(VotingHeadTemplate):
    def __init__(self, model_cfg, **kwargs):
        super().__init__(model_cfg)
        self.cla_layer = _conv_stack(self.model_cfg.CLS_FC.CHANNELS)
        self.vote_layer = _conv_stack(self.model_cfg.REG_FC.CHANNELS)
        if self.model_cfg.TRANSFORMER_BLOCK.ENABLE:
            self.transformer_block = build_transformer(self.model_cfg.TRANSFORMER_BLOCK)

    # ------------------------------------------------------------------ losses (reference :29-62)
    @staticmethod
    def _seed_labels(d):
        """The seeds' labels, gathered from the per-point labels when first needed (the one-launch loss path of
        train_ops.track_losses gathers them itself)."""
        if 'cls_label' not in d:
            d['cls_label'] = d['cls_label_points'].gather(1, d['search_inds'])
        return d

    def get_cls_layer_loss(self, forward_ret_dict):
        weights = self.model_cfg.LOSS_CONFIG.LOSS_WEIGHTS
        forward_ret_dict = self._seed_labels(forward_ret_dict)
        loss = self.cls_loss_func(forward_ret_dict['pred_centroids_cls'].view(-1),
                                  forward_ret_dict['cls_label'].view(-1))
        tb_dict = {'centroids_cls_loss': loss.item()}
        return loss.float() * weights['centroids_cls_weight'], tb_dict

    def get_reg_layer_loss(self, forward_ret_dict):
        weights = self.model_cfg.LOSS_CONFIG.LOSS_WEIGHTS
        mask = self._seed_labels(forward_ret_dict)['cls_label']
        pred = forward_ret_dict['pred_centroids_votes']
        target = forward_ret_dict['reg_label'][:, None, :3].expand_as(pred)
        loss = self.reg_loss_func(pred, target)
        loss = (loss.mean(2) * mask).sum() / (mask.sum() + 1e-06)
        tb_dict = {'centroids_reg_loss': loss.item()}
        return loss.float() * weights['centroids_reg_weight'], tb_dict

    def get_loss(self, tb_dict=None):
        tb_dict = {} if tb_dict is None else tb_dict
        loss_cls, tb1 = self.get_cls_layer_loss(self.forward_ret_dict)
        loss_reg, tb2 = self.get_reg_layer_loss(self.forward_ret_dict)
        tb_dict.update(tb1)
        tb_dict.update(tb2)
        return (loss_cls + loss_reg).float(), tb_dict

    # ------------------------------------------------------------------ forward (reference :64-109)
    def _stack(self, seq, rows, residual=None):
        if self.training:
            return train_ops.conv1d_stack_rows(seq, rows, residual)
        return layer_utils.rows_forward(seq, rows, residual)

    def _forward_rows(self, batch_dict):
        """On a HIP device: the same computation on point-major rows — the transformer already works on (B,N,C) rows,
        the two Conv1d stacks run as linear layers over rows (eval: BatchNorm folded, one launch each; training: the row
        kernels of ptt_amd/train_ops.py, batch statistics out of the GEMM epilogues), and `votes_feats` is handed to the box
        head as the (B,1+C,N) view of (B,N,1+C) storage, which its grouping kernels gather coalesced. No layout copies."""
        seeds = batch_dict['search_seeds']                                        # (B,N,3)
        rows = batch_dict['cosine_feats'].transpose(1, 2)                         # (B,N,C)
        if hasattr(self, 'transformer_block'):
            rows = self.transformer_block(xyz=seeds, features=rows.contiguous(), want_attn=False)[0]
        with_xyz = torch.cat((seeds, rows), dim=2)                                # (B,N,3+C)
        cls_in = with_xyz if getattr(self.model_cfg, 'CLS_USE_SEARCH_XYZ', False) else rows
        cls_out = self._stack(self.cla_layer, cls_in).squeeze(-1)                 # (B,N)
        voted = self._stack(self.vote_layer, with_xyz, residual=with_xyz)
        voted_xyz, voted_feats = voted.split([3, voted.shape[-1] - 3], dim=2)     # one split: its backward is one concatenation
        batch_dict['pred_centroids_cls'] = cls_out.squeeze(0)
        batch_dict['pred_centroids_votes'] = voted_xyz.contiguous()               # (B,N,3)
        batch_dict['votes_feats'] = torch.cat((cls_out.sigmoid().unsqueeze(-1), voted_feats),
                                              dim=2).transpose(1, 2)              # (B,1+C,N) view
        if self.training:
            self.forward_ret_dict = {
                'pred_centroids_cls': batch_dict['pred_centroids_cls'],
                'pred_centroids_votes': batch_dict['pred_centroids_votes'],
                'cls_label_points': batch_dict['cls_label'], 'search_inds': batch_dict['search_inds'],
                'reg_label': batch_dict['reg_label'],
            }
        return batch_dict

    def _forward_one_frame(self, batch_dict):
        """The same head for a handful of frames (one tracklet frame, the reference's tracking mode): three launches of TWO
        jobs each — cla_layer's and vote_layer's i-th convolutions side by side (ptt_row_jobs_f32) — with the concatenations,
        the sigmoid, the residual and the output slices of reference :83-94 inside the jobs: vote_layer reads [feats | xyz]
        from two tensors, its last convolution adds (xyz | feats) back and writes votes and votes_feats[:, 1:], cla_layer's
        last convolution writes the raw scores and their sigmoid into votes_feats[:, 0]. Nine launches less per frame."""
        seeds = batch_dict['search_seeds'].contiguous()                          # (B,N,3)
        knn = batch_dict.pop('search_seeds_knn', None)                            # formed by the backbone beside its ball queries
        rows = batch_dict['cosine_feats'].transpose(1, 2).contiguous()            # (B,N,C): a view of point-major storage
        if hasattr(self, 'transformer_block'):
            rows = self.transformer_block(xyz=seeds, features=rows, knn=knn, want_attn=False)[0]
        B, N, C = rows.shape
        dev = rows.device
        cls_xyz = getattr(self.model_cfg, 'CLS_USE_SEARCH_XYZ', False)
        cla = layer_utils.rows_layers(self.cla_layer, 3 if cls_xyz else 0)
        vote = layer_utils.rows_layers(self.vote_layer, 3)
        new = lambda c: torch.empty((B * N, c), dtype=torch.float32, device=dev)
        hc, hv = [new(cla[0][1]), new(cla[1][1])], [new(vote[0][1]), new(vote[1][1])]
        ld = (1 + C + 3) // 4 * 4                                               # votes_feats rows padded to 16 bytes
        vfeats = torch.empty((B, N, ld), dtype=torch.float32, device=dev)
        votes = torch.empty((B, N, 3), dtype=torch.float32, device=dev)
        cls_raw = torch.empty((B * N, 1), dtype=torch.float32, device=dev)

        def job(L, x, out, x2=None, **kw):
            wp, cout, scale, shift, relu = L
            return ops.row_job(wp, cout, x=x, x2=x2, scale=scale, shift=shift, act=1 if relu else 0, out=out, **kw)

        ops.row_jobs([job(cla[0], rows, hc[0], seeds if cls_xyz else None), job(vote[0], rows, hv[0], seeds)])
        ops.row_jobs([job(cla[1], hc[0], hc[1]), job(vote[1], hv[0], hv[1])])
        wp, cout, scale, shift, _ = cla[2]
        ops.row_jobs([ops.row_job(wp, cout, x=hc[1], scale=scale, shift=shift, act=2, out=vfeats.view(B * N, ld)[:, 0:1], raw=cls_raw),
                      job(vote[2], hv[1], vfeats.view(B * N, ld), res=rows, res2=seeds, res_split=3, out_col0=1, out2=votes,
                          out_split=3)])
        batch_dict['pred_centroids_cls'] = cls_raw.view(B, N).squeeze(0)
        batch_dict['pred_centroids_votes'] = votes
        batch_dict['votes_feats'] = vfeats[:, :, :1 + C].transpose(1, 2)          # (B,1+C,N) view of point-major rows
        return batch_dict

    def _one_frame(self, feats):
        """At most ops.ONE_FRAME_MAX_POINTS seeds in the batch, the shipped stack shapes (three convolutions, vote_layer
        from 3 + C back to 3 + C, cla_layer to one score)."""
        if self.training or feats.shape[0] * feats.shape[2] > ops.ONE_FRAME_MAX_POINTS:
            return False
        C = feats.shape[1]
        cv, cc = [u.conv.weight.shape for u in self.vote_layer], [u.conv.weight.shape for u in self.cla_layer]
        cls_in = C + 3 if getattr(self.model_cfg, 'CLS_USE_SEARCH_XYZ', False) else C
        return (len(cv) == 3 and len(cc) == 3 and cv[0][1] == C + 3 and cv[2][0] == C + 3 and cc[0][1] == cls_in and cc[2][0] == 1
                and not hasattr(self.vote_layer[2], 'activation') and not hasattr(self.cla_layer[2], 'activation'))

    def _fusable(self, feats):
        if self.training:
            return (train_ops.conv1d_stack_usable(self.cla_layer, feats) and train_ops.conv1d_stack_usable(self.vote_layer, feats))
        return layer_utils.rows_fusable(self.cla_layer, feats) and layer_utils.rows_fusable(self.vote_layer, feats)

    def forward(self, batch_dict):
        if self._fusable(batch_dict['cosine_feats']):
            if self._one_frame(batch_dict['cosine_feats']):
                return self._forward_one_frame(batch_dict)
        batch_dict.pop('search_seeds_knn', None)             # the backbone's hand-over to the one-frame path only
        if self._fusable(batch_dict['cosine_feats']):
            return self._forward_rows(batch_dict)
        seeds_xyz = batch_dict['search_seeds'].transpose(1, 2).contiguous()      # (B,3,N)
        feats = batch_dict['cosine_feats']                                        # (B,C,N)

        if hasattr(self, 'transformer_block'):
            fused = self.transformer_block(xyz=batch_dict['search_seeds'],
                                           features=feats.transpose(1, 2).contiguous(), want_attn=False)[0]
            feats = fused.transpose(1, 2).contiguous()

        if getattr(self.model_cfg, 'CLS_USE_SEARCH_XYZ', False):
            feats = torch.cat((seeds_xyz, feats), dim=1)
            cls_out = self.cla_layer(feats).squeeze(1)
            vote_in = feats
        else:
            cls_out = self.cla_layer(feats).squeeze(1)
            vote_in = torch.cat((seeds_xyz, feats), dim=1)
        cls_pred = cls_out.squeeze(0)
        score = cls_out.sigmoid()
        voted = vote_in + self.vote_layer(vote_in)

        batch_dict['pred_centroids_cls'] = cls_pred
        batch_dict['pred_centroids_votes'] = voted[:, 0:3, :].transpose(1, 2).contiguous()       # (B,N,3)
        batch_dict['votes_feats'] = torch.cat((score.unsqueeze(1), voted[:, 3:, :]), dim=1)      # (B,1+C,N)

        if self.training:
            self.forward_ret_dict = {
                'pred_centroids_cls': batch_dict['pred_centroids_cls'],
                'pred_centroids_votes': batch_dict['pred_centroids_votes'],
                'cls_label': batch_dict['cls_label'].gather(1, batch_dict['search_inds']),
                'reg_label': batch_dict['reg_label'],
            }
        return batch_dict
