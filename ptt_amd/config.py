"""Mirror of ptt/config.py (:16-85): attribute-access config dict, YAML loading with `_BASE_CONFIG_` includes and
`--set KEY VAL ...` overrides — without the easydict dependency. Also carries the MODEL section of the shipped
KITTI / nuScenes configs as Python constants (tools/cfgs/kitti_models/ptt.yaml:28-127; the nuScenes MODEL section
differs only by the absence of CLS_USE_SEARCH_XYZ), so the model can be built without the YAML."""
from ast import literal_eval
from pathlib import Path

import yaml


class EasyDict(dict):
    """dict with attribute access; nested dicts are converted on assignment (the easydict behaviour the code uses)."""

    def __init__(self, d=None, **kwargs):
        super().__init__()
        for k, v in dict(d or {}, **kwargs).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, EasyDict):
            v = EasyDict(v)
        elif isinstance(v, (list, tuple)):
            v = type(v)(EasyDict(x) if isinstance(x, dict) and not isinstance(x, EasyDict) else x for x in v)
        super().__setitem__(k, v)

    __setattr__ = __setitem__

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def update(self, other=None, **kwargs):
        for k, v in dict(other or {}, **kwargs).items():
            self[k] = v


def log_config_to_file(cfg, pre='cfg', logger=None):
    for key, val in cfg.items():
        if isinstance(val, EasyDict):
            logger.info('\n%s.%s = edict()' % (pre, key))
            log_config_to_file(val, pre=pre + '.' + key, logger=logger)
        else:
            logger.info('%s.%s: %s' % (pre, key, val))


def cfg_from_list(cfg_list, config):
    """`--set A.B.C value ...` overrides with the reference's typing rules (config.py:16-48)."""
    assert len(cfg_list) % 2 == 0
    for k, v in zip(cfg_list[0::2], cfg_list[1::2]):
        keys = k.split('.')
        d = config
        for sub in keys[:-1]:
            assert sub in d, 'NotFoundKey: %s' % sub
            d = d[sub]
        sub = keys[-1]
        assert sub in d, 'NotFoundKey: %s' % sub
        try:
            value = literal_eval(v)
        except Exception:
            value = v
        cur = d[sub]
        if type(value) != type(cur) and isinstance(cur, EasyDict):
            for item in value.split(','):
                ck, cv = item.split(':')
                cur[ck] = type(cur[ck])(cv)
        elif type(value) != type(cur) and isinstance(cur, list):
            items = value.split(',') if isinstance(value, str) else list(value)   # '3,4' literal_evals to a tuple
            d[sub] = [type(cur[0])(x) for x in items]
        else:
            assert type(value) == type(cur), 'type {} does not match original type {}'.format(type(value), type(cur))
            d[sub] = value


def _load_yaml(path):
    with open(path, 'r') as f:
        return yaml.safe_load(f)


def merge_new_config(config, new_config):
    if '_BASE_CONFIG_' in new_config:
        config.update(EasyDict(_load_yaml(new_config['_BASE_CONFIG_'])))
    for key, val in new_config.items():
        if not isinstance(val, dict):
            config[key] = val
            continue
        if key not in config:
            config[key] = EasyDict()
        merge_new_config(config[key], val)
    return config


def cfg_from_yaml_file(cfg_file, config):
    merge_new_config(config=config, new_config=_load_yaml(cfg_file))
    return config


def ptt_model_cfg(cls_use_search_xyz=False):
    """MODEL section of tools/cfgs/kitti_models/ptt.yaml as an EasyDict."""
    tb = dict(ENABLE=True, NAME='TransformerBlock', DIM_INPUT=256, DIM_MODEL=512, KNN=16, N_HEADS=1, N_LAYERS=1)
    return EasyDict(dict(
        NAME='PTT',
        BACKBONE_3D=dict(NAME='PointNet2BackboneLight', DEBUG=False, SA_CONFIG=dict(
            SAMPLE_METHOD=['fps', 'sequence', 'sequence'], USE_XYZ=True, NORMALIZE_XYZ=True,
            NPOINTS_SEARCH=[512, 256, 128], NPOINTS_TEMPLATE=[256, 128, 64], RADIUS=[0.3, 0.5, 0.7],
            NSAMPLE=[32, 32, 32], MLPS=[[0, 64, 64, 128], [128, 128, 128, 256], [256, 128, 128, 256]])),
        SIMILARITY_MODULE=dict(NAME='CosineSimAug', DEBUG=False, MLP=dict(CHANNELS=[260, 256, 256, 256], BN=True),
                               CONV=dict(CHANNELS=[256, 256, 256], BN=True)),
        CENTROID_HEAD=dict(NAME='CentroidVotingHead', DEBUG=False, CLS_USE_SEARCH_XYZ=cls_use_search_xyz,
                           CLS_FC=dict(CHANNELS=[256, 256, 256, 1]), REG_FC=dict(CHANNELS=[259, 256, 256, 259]),
                           TRANSFORMER_BLOCK=dict(tb),
                           LOSS_CONFIG=dict(CLS_LOSS='BinaryCrossEntropy', CLS_LOSS_REDUCTION='mean',
                                            CLS_LOSS_POS_WEIGHT=1.0, REG_LOSS='smooth-l1',
                                            LOSS_WEIGHTS={'centroids_cls_weight': 0.2, 'centroids_reg_weight': 1.0})),
        BOX_HEAD=dict(NAME='BoxVotingHead', DEBUG=False, FC=[256, 256, 256, 5],
                      SA_CONFIG=dict(NPOINTS=64, RADIUS=0.3, NSAMPLE=16, MLPS=[257, 256, 256, 256], USE_XYZ=True,
                                     NORMALIZE_XYZ=True, SAMPLE_METHOD='fps'),
                      TRANSFORMER_BLOCK=dict(tb),
                      LOSS_CONFIG=dict(CLS_LOSS='BinaryCrossEntropy', CLS_LOSS_REDUCTION='none',
                                       CLS_LOSS_POS_WEIGHT=2.0, REG_LOSS='smooth-l1',
                                       LOSS_WEIGHTS={'boxes_cls_weight': 1.5, 'boxes_reg_weight': 0.2})),
    ))


class StubDataset(object):
    """The attributes build_network reads from the dataset object (tracker3d_template.py:15-17,38-44), for building a
    model without a dataset (benchmarks, tests)."""

    class _Encoder(object):
        num_point_features = 3

    def __init__(self, training=False, class_names=('Car',)):
        self.training = training
        self.class_names = list(class_names)
        self.point_feature_encoder = self._Encoder()
        self.grid_size = None
        self.point_cloud_range = None
        self.voxel_size = None


cfg = EasyDict()
cfg.ROOT_DIR = (Path(__file__).resolve().parent / '../').resolve()
cfg.LOCAL_RANK = 0
