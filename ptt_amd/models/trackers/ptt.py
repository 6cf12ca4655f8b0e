"""Mirror of ptt/models/trackers/ptt.py: PTT (:15-60) — runs the module list over the batch dict; eval returns the
dict (keys documented at reference :22-39), training returns (ret_dict, tb_dict, disp_dict)."""
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

    def get_training_loss(self):
        disp_dict = {}
        loss_centroids, tb_dict = self.centroid_voting_head.get_loss()
        loss_boxes, tb_dict = self.box_voting_head.get_loss(tb_dict)
        disp_dict.update(tb_dict)
        return loss_centroids + loss_boxes, tb_dict, disp_dict
