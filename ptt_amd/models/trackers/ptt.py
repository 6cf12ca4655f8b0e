"""Mirror of ptt/models/trackers/ptt.py: PTT (:15-60) — runs the module list over the batch dict; eval returns the
dict (keys documented at reference :22-39), training returns (ret_dict, tb_dict, disp_dict)."""
import torch.nn as nn

from ... import train_ops
from .tracker3d_template import Tracker3DTemplate


class PTT(Tracker3DTemplate):
    def __init__(self, model_cfg, num_class, dataset):
        super().__init__(model_cfg=model_cfg, num_class=num_class, dataset=dataset)
        self.module_list = self.build_networks()

    def forward(self, batch_dict):
        for module in self.module_list:
            batch_dict = module(batch_dict)
        if self.training:
            loss, tb_dict, disp_dict = self.get_training_loss()
            return {'loss': loss.float()}, tb_dict, disp_dict
        return batch_dict

    def _one_launch_losses(self):
        """On a HIP device with the shipped loss modules: the four losses (and, in the backward pass, their gradients) in one
        launch (ptt_track_losses_f32) instead of ~150 element-wise launches and four .item() synchronisations; the values in
        tb_dict / disp_dict are train_ops.LossValues numbers, fetched from the device when a logger first reads them."""
        c, b = self.centroid_voting_head, self.box_voting_head
        dc, db = c.forward_ret_dict or {}, b.forward_ret_dict or {}
        if not all(k in dc for k in ('cls_label_points', 'search_inds')) or not all(k in db for k in ('pred_box_data', 'centres')):
            return None
        for head, reduction in ((c, 'mean'), (b, 'none')):
            f, r = head.cls_loss_func, head.reg_loss_func
            if not (type(f) is nn.BCEWithLogitsLoss and f.reduction == reduction and f.weight is None and f.pos_weight is not None
                    and f.pos_weight.numel() == 1 and type(r) is nn.SmoothL1Loss and r.reduction == 'none' and r.beta == 1.0):
                return None
        t = (dc['pred_centroids_cls'], dc['pred_centroids_votes'], db['pred_box_data'], db['centres'], dc['cls_label_points'], dc['reg_label'],
             c.cls_loss_func.pos_weight, b.cls_loss_func.pos_weight)
        if dc['pred_centroids_cls'].dim() != 2 or db['pred_box_data'].shape[-1] != 5 or not train_ops.track_losses_usable(
                *t, search_inds=dc['search_inds'], seeds_shape=dc['pred_centroids_cls'].shape):
            return None
        wc, wb = c.model_cfg.LOSS_CONFIG.LOSS_WEIGHTS, b.model_cfg.LOSS_CONFIG.LOSS_WEIGHTS
        total, vals = train_ops.track_losses(t[0], t[1], t[2], t[3], t[4], dc['search_inds'], t[5], t[6], t[7],
                                             (wc['centroids_cls_weight'], wc['centroids_reg_weight'], wb['boxes_cls_weight'], wb['boxes_reg_weight']))
        tb_dict = {'centroids_cls_loss': vals[1], 'centroids_reg_loss': vals[2], 'boxes_cls_loss': vals[3], 'boxes_reg_loss': vals[4]}
        return total, tb_dict, dict(tb_dict)

    def get_training_loss(self):
        fused = self._one_launch_losses()
        if fused is not None:
            return fused
        disp_dict = {}
        loss_centroids, tb_dict = self.centroid_voting_head.get_loss()
        loss_boxes, tb_dict = self.box_voting_head.get_loss(tb_dict)
        disp_dict.update(tb_dict)
        return loss_centroids + loss_boxes, tb_dict, disp_dict
