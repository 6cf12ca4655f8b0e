"""Mirror of ptt/models/similarity_modules/p2b_xcoor.py: CosineSimAug (:9-46), the P2B template/search feature
augmentation. Attribute names (`cosine`, `mlp`, `conv`) and state_dict keys are the reference's.

Reference op sequence: cosine map (B,64,128) -> expand/concat to a (B,260,64,128) tensor -> SharedMLP
[260,256,256,256] over all 8192 (template, search) pairs -> max over the template axis -> 2 x Conv1d.
Eval mode on a HIP device: only the similarity channel of the 260 depends on the search point, so layer 0
is split into a per-template-point part (one small MFMA linear) plus a rank-1 update inside the fused kernel
(ptt_xcorr_fused_fwd_f32), which also computes the cosines, runs the remaining layers on fp32 MFMA and takes
the max over the 64 template points in registers. The big fusion tensor never exists; layer-0 work drops from
8192 x 260 x 256 to 64 x 259 x 256 MACs per frame (a third of the module's FLOPs).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import ops, train_ops
from ..backbones_3d.pointnet2 import pytorch_utils as layer_utils


class CosineSimAug(nn.Module):
    def __init__(self, model_cfg):
        super().__init__()
        self.model_cfg = model_cfg
        self.cosine = nn.CosineSimilarity(dim=1)
        self.mlp = layer_utils.SharedMLP(self.model_cfg.MLP.CHANNELS, bn=self.model_cfg.MLP.BN)
        self.conv = (
            layer_utils.Seq(self.model_cfg.CONV.CHANNELS[0])
            .conv1d(self.model_cfg.CONV.CHANNELS[1], bn=self.model_cfg.CONV.BN)
            .conv1d(self.model_cfg.CONV.CHANNELS[2], activation=None)
        )
        self._cache = None

    # ------------------------------------------------------------------ fused-path parameters
    def _fusable(self, search_feats, template_feats):
        """Eval mode on a HIP device, float32 features, 64 template seeds, BatchNorm'd SharedMLP / first Conv1d (their
        folding is what `_params` implements) and no autograd graph being recorded; an eval-mode HIP call that is
        turned away says so once (ops.note_unfused)."""
        if self.training or not search_feats.is_cuda:
            return False
        name = 'CosineSimAug'
        if ops.autograd_recording(self, search_feats, template_feats):
            return ops.note_unfused(name, 'autograd is recording (wrap inference in torch.no_grad())')
        if search_feats.dtype != torch.float32 or template_feats.dtype != torch.float32:
            return ops.note_unfused(name, 'features must be float32')
        if template_feats.shape[-1] % 64 != 0:
            return ops.note_unfused(name, 'ptt_xcorr_fused_fwd_f32 walks the template seeds in chunks of 64 (got %d)'
                                    % template_feats.shape[-1])
        units = list(self.mlp)
        if len(units) < 3 or len(units) > 5:
            return ops.note_unfused(name, 'SharedMLP depth %d' % len(units))
        for u in units:
            if not hasattr(u, 'normlayer') or u.conv.weight.shape[0] % 32 or u.conv.weight.shape[0] > 256:
                return ops.note_unfused(name, 'SharedMLP needs BatchNorm and widths that are multiples of 32, <= 256')
        if len(self.conv) != 2 or not hasattr(self.conv[0], 'normlayer') or hasattr(self.conv[1], 'normlayer'):
            return ops.note_unfused(name, 'CONV must be [Conv1d+BN(+ReLU), Conv1d] (CONV.BN: True)')
        if units[0].conv.weight.shape[1] != template_feats.shape[1] + 4:
            return ops.note_unfused(name, 'first SharedMLP layer does not take 1 + 3 + C channels')
        return True

    def train(self, mode=True):
        self._cache = None          # BatchNorm running statistics may change without a _version bump in train mode
        return super().train(mode)

    @staticmethod
    def _fold(unit):
        bn = unit.normlayer.bn
        scale = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).float().contiguous()
        shift = (bn.bias - bn.running_mean * scale).float().contiguous()
        return scale, shift

    def _params(self):
        ts = list(self.state_dict().values())         # incl. num_batches_tracked: bumped by every train-mode forward
        key = tuple((t.data_ptr(), t._version) for t in ts)
        if self._cache is not None and self._cache[0] == key:
            return self._cache[1]
        with torch.no_grad():
            units = list(self.mlp)
            w0 = units[0].conv.weight.reshape(units[0].conv.weight.shape[0], -1)       # (C0, 1 + 3 + C)
            s0, t0 = self._fold(units[0])
            layers = []
            for u in units[1:]:
                sc, sh = self._fold(u)
                w = u.conv.weight
                # BatchNorm scale folded into the packed weights: the kernel starts its accumulators at the shift
                layers.append((ops.pack_weight(w * sc.view(-1, 1, 1, 1)), None, sh, w.shape[1], w.shape[0], True))
            # layer 0's BatchNorm is folded into its two halves: the per-template-point term comes out of the linear
            # kernel as s0 * (W0[:,1:] . [xyz;feat]) + t0, the similarity column as s0 * w_sim
            P = dict(w_sim=(w0[:, 0].float() * s0).contiguous(), w_rest=ops.pack_weight(w0[:, 1:]), c0=w0.shape[0],
                     w_rest_fx=ops.pack_weight(w0[:, 1:], 3),      # the same weights for rows laid out [feats | xyz]
                     scale0=s0, shift0=t0, layers=layers)
        ops.publish_params(self.conv[0].conv.weight.device)
        self._cache = (key, P)
        return P

    def forward(self, batch_dict):
        search_feats = batch_dict['search_feats']            # (B,f,n2)
        template_feats = batch_dict['template_feats']        # (B,f,n1)
        template_xyz = batch_dict['template_seeds']          # (B,n1,3)
        b, f, n2 = search_feats.shape
        n1 = template_feats.shape[-1]

        if self._fusable(search_feats, template_feats):
            P = self._params()
            trows = template_feats.transpose(1, 2)                                            # (B,n1,f)
            if b * n1 <= ops.ONE_FRAME_MAX_POINTS and trows.is_contiguous():
                # a handful of frames: the per-template-point term reads [feats | xyz] from the two tensors (no concatenation)
                pre = torch.empty((b, n1, P['c0']), dtype=torch.float32, device=trows.device)
                ops.row_jobs([ops.row_job(P['w_rest_fx'], P['c0'], x=trows, x2=template_xyz.contiguous(), scale=P['scale0'],
                                          shift=P['shift0'], out=pre)])
            else:
                rows = torch.cat((template_xyz, trows), dim=2)                                # (B,n1,3+f)
                pre = ops.linear(rows, P['w_rest'], P['c0'], P['scale0'], P['shift0'])        # (B,n1,C0), BN folded in
            if b * n2 <= ops.ONE_FRAME_MAX_POINTS:
                # a handful of frames: two workgroups per search point, cosines formed in the kernel (one launch less)
                fused, _ = ops.xcorr_fused(search_feats, template_feats, pre, P['w_sim'], None, None,
                                           P['layers'], eps=self.cosine.eps, split=True)      # (B,C,n2) view
            else:
                cos_t = ops.cosine_map(search_feats, template_feats, eps=self.cosine.eps)     # (B,n2,n1), one launch
                fused, _ = ops.xcorr_fused(search_feats, template_feats, pre, P['w_sim'], None, None,
                                           P['layers'], eps=self.cosine.eps, cos_t=cos_t)     # (B,C,n2) view
            if fused.dim() == 4:
                # the split form's two halves: their element-wise maximum (= the max over the template axis) is taken by the
                # first trailing convolution while it stages its operand
                L = layer_utils.rows_layers(self.conv)
                h0, h1 = fused[0].transpose(1, 2), fused[1].transpose(1, 2)                  # (B,n2,C) contiguous rows
                x = None
                for li, (wp, cout, scale, shift, relu) in enumerate(L):
                    y = torch.empty((b, n2, cout), dtype=torch.float32, device=h0.device)
                    ops.row_jobs([ops.row_job(wp, cout, x=h0 if li == 0 else x, xmax=h1 if li == 0 else None, scale=scale, shift=shift,
                                              act=1 if relu else 0, out=y)])
                    x = y
            else:
                y = layer_utils.rows_forward(self.conv, fused.transpose(1, 2))               # both convolutions, one launch
            batch_dict['cosine_feats'] = y.transpose(1, 2)                                    # (B,c,n2) view
            return batch_dict

        if (train_ops.usable(self.mlp, search_feats) and self.mlp[0].conv.weight.shape[1] == f + 4
                and self.mlp[0].conv.weight.shape[0] % 4 == 0 and self.mlp[0].conv.weight.shape[0] <= 256):   # ptt_xcorr_z0_bwd_f32: C0 <= 256
            # training on a HIP device: layer 0 split per template point + similarity term (train_ops.xcorr_hoisted),
            # the remaining SharedMLP layers and the max over the template axis on the row kernels
            fused = train_ops.xcorr_hoisted(search_feats, template_feats, template_xyz, self.mlp, self.cosine.eps)   # (B,C,n2) view
            if train_ops.conv1d_stack_usable(self.conv, fused):
                batch_dict['cosine_feats'] = train_ops.conv1d_stack_rows(self.conv, fused.transpose(1, 2)).transpose(1, 2)
            else:
                batch_dict['cosine_feats'] = self.conv(fused)
            return batch_dict
        sim_feat = self.cosine(template_feats.unsqueeze(-1).expand(b, f, n1, n2),
                               search_feats.unsqueeze(2).expand(b, f, n1, n2))
        template_xyz_ = template_xyz.transpose(1, 2).contiguous().unsqueeze(-1).expand(b, 3, n1, n2)
        fusion_feature = torch.cat((sim_feat.unsqueeze(1), template_xyz_), dim=1)
        fusion_feature = torch.cat((fusion_feature, template_feats.unsqueeze(-1).expand(b, f, n1, n2)), dim=1)
        if train_ops.usable(self.mlp, fusion_feature):         # training on a HIP device: hand-written row kernels
            batch_dict['cosine_feats'] = self.conv(train_ops.shared_mlp_pool(fusion_feature, self.mlp, pool_dim=2))
            return batch_dict
        fusion_feature = self.mlp(fusion_feature)
        fusion_feature = fusion_feature.max(dim=2)[0]        # = F.max_pool2d(., [n1, 1]).squeeze(2) (reference :41-42)
        batch_dict['cosine_feats'] = self.conv(fusion_feature)
        return batch_dict
