"""Mirror of ptt/models/voting_heads/voting_head_template.py (:8-25): the two loss modules every head owns.
The reference calls .cuda() on them inside the constructor (:23,25); here they follow `model.to(device)`.
`cls_loss_func.pos_weight` is a registered buffer and therefore part of the checkpoint key set."""
import torch
import torch.nn as nn


class VotingHeadTemplate(nn.Module):
    def __init__(self, model_cfg):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_class = 1
        self.build_losses(self.model_cfg.LOSS_CONFIG)
        self.forward_ret_dict = None

    def build_losses(self, losses_cfg):
        pos_weight = torch.tensor([losses_cfg.CLS_LOSS_POS_WEIGHT], dtype=torch.float32)
        self.add_module('cls_loss_func', nn.BCEWithLogitsLoss(pos_weight=pos_weight,
                                                              reduction=losses_cfg.CLS_LOSS_REDUCTION))
        self.add_module('reg_loss_func', nn.SmoothL1Loss(reduction='none'))
