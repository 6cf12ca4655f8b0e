from .ptt import PTT

__all__ = {'PTT': PTT}


def build_tracker(model_cfg, num_class, dataset):
    return __all__[model_cfg.NAME](model_cfg=model_cfg, num_class=num_class, dataset=dataset)
