"""Mirror of ptt/models/trackers/tracker3d_template.py: Tracker3DTemplate (:9-155) — assembles the tracker from
`module_topology` (backbone_3d -> similarity_module -> centroid_voting_head -> box_voting_head) and holds the
checkpoint helpers. Module attribute names are the state_dict prefixes of the reference's checkpoints."""
import os

import torch
import torch.nn as nn

from .. import backbones_3d, similarity_modules, voting_heads


class Tracker3DTemplate(nn.Module):
    def __init__(self, model_cfg, num_class, dataset):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_class = num_class
        self.dataset = dataset
        self.training = dataset.training
        self.class_names = dataset.class_names
        self.register_buffer('global_step', torch.LongTensor(1).zero_())
        self.module_topology = ['backbone_3d', 'similarity_module', 'centroid_voting_head', 'box_voting_head']

    @property
    def mode(self):
        return 'TRAIN' if self.training else 'TEST'

    def forward(self, **kwargs):
        raise NotImplementedError

    def update_global_step(self):
        self.global_step += 1

    def build_networks(self):
        info = {
            'module_list': [],
            'num_rawpoint_features': self.dataset.point_feature_encoder.num_point_features,
            'num_point_features': self.dataset.point_feature_encoder.num_point_features,
            'grid_size': self.dataset.grid_size,
            'point_cloud_range': self.dataset.point_cloud_range,
            'voxel_size': self.dataset.voxel_size,
        }
        for name in self.module_topology:
            module, info = getattr(self, 'build_%s' % name)(model_info_dict=info)
            self.add_module(name, module)
        return info['module_list']

    def build_backbone_3d(self, model_info_dict):
        cfg = self.model_cfg.get('BACKBONE_3D', None)
        if cfg is None:
            return None, model_info_dict
        module = backbones_3d.__all__[cfg.NAME](
            model_cfg=cfg, input_channels=model_info_dict['num_point_features'],
            grid_size=model_info_dict['grid_size'], voxel_size=model_info_dict['voxel_size'],
            point_cloud_range=model_info_dict['point_cloud_range'])
        model_info_dict['module_list'].append(module)
        model_info_dict['num_point_features'] = module.num_point_features
        return module, model_info_dict

    def build_similarity_module(self, model_info_dict):
        cfg = self.model_cfg.get('SIMILARITY_MODULE', None)
        if cfg is None:
            return None, model_info_dict
        module = similarity_modules.__all__[cfg.NAME](model_cfg=cfg)
        model_info_dict['module_list'].append(module)
        return module, model_info_dict

    def _build_head(self, key, model_info_dict):
        cfg = self.model_cfg.get(key, None)
        if cfg is None:
            return None, model_info_dict
        module = voting_heads.__all__[cfg.NAME](
            model_cfg=cfg, input_channels=model_info_dict['num_point_features'], num_class=1)
        model_info_dict['module_list'].append(module)
        return module, model_info_dict

    def build_centroid_voting_head(self, model_info_dict):
        return self._build_head('CENTROID_HEAD', model_info_dict)

    def build_box_voting_head(self, model_info_dict):
        return self._build_head('BOX_HEAD', model_info_dict)

    def post_processing(self, batch_dict):
        pass

    # ------------------------------------------------------------------ checkpoints (reference :96-155)
    def load_params_from_file(self, filename, logger, to_cpu=False):
        """Partial load: every checkpoint entry whose key AND shape match this model is taken."""
        if not os.path.isfile(filename):
            raise FileNotFoundError(filename)
        logger.info('==> Loading parameters from checkpoint %s to %s' % (filename, 'CPU' if to_cpu else 'GPU'))
        checkpoint = torch.load(filename, map_location=torch.device('cpu') if to_cpu else None)
        disk = checkpoint['model_state']
        if 'version' in checkpoint:
            logger.info('==> Checkpoint trained from version: %s' % checkpoint['version'])
        own = self.state_dict()
        taken = {k: v for k, v in disk.items() if k in own and own[k].shape == v.shape}
        own.update(taken)
        self.load_state_dict(own)
        for k in own:
            if k not in taken:
                logger.info('Not updated weight %s: %s' % (k, str(own[k].shape)))
        logger.info('==> Done (loaded %d/%d)' % (len(taken), len(own)))

    def load_params_with_optimizer(self, filename, to_cpu=False, optimizer=None, logger=None):
        if not os.path.isfile(filename):
            raise FileNotFoundError(filename)
        logger.info('==> Loading parameters from checkpoint %s to %s' % (filename, 'CPU' if to_cpu else 'GPU'))
        loc = torch.device('cpu') if to_cpu else None
        checkpoint = torch.load(filename, map_location=loc)
        self.load_state_dict(checkpoint['model_state'])
        if optimizer is not None:
            if checkpoint.get('optimizer_state') is not None:
                optimizer.load_state_dict(checkpoint['optimizer_state'])
            else:
                stem, ext = os.path.splitext(filename)
                side = '%s_optim%s' % (stem, ext)
                if os.path.exists(side):
                    optimizer.load_state_dict(torch.load(side, map_location=loc)['optimizer_state'])
        if 'version' in checkpoint:
            print('==> Checkpoint trained from version: %s' % checkpoint['version'])
        logger.info('==> Done')
        return checkpoint.get('it', 0.0), checkpoint.get('epoch', -1)
