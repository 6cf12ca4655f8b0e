"""Mirror of ptt/models/transformer_block/__init__.py: registry + build_transformer (:7-27)."""
from .variants import TransformerBlock, TransformerBlockSTD

__all__ = {
    'TransformerBlock': TransformerBlock,
    'TransformerBlockSTD': TransformerBlockSTD,
}

_NOT_ON_HOT_PATH = ('MulTransformerBlock', 'TransformerBlockALL', 'TransformerBlockBackbone', 'TransformerBlockCosine',
                    'TransformerBlockMLP', 'TransformerBlockOffset', 'CrossAttentionBlock')


def build_transformer(model_cfg):
    """model_cfg: NAME, DIM_INPUT, DIM_MODEL, KNN, N_HEADS, N_LAYERS (heads/layers are swallowed by
    **kwargs exactly as in the reference, transformer_block/__init__.py:20-27)."""
    name = model_cfg.NAME
    if name not in __all__:
        if name in _NOT_ON_HOT_PATH:
            raise NotImplementedError(
                "%s exists in the reference but is selected by no shipped config (SURVEY.md §2 row 5/6); "
                "only TransformerBlock / TransformerBlockSTD are provided" % name)
        raise KeyError(name)
    return __all__[name](d_points=model_cfg.DIM_INPUT, d_model=model_cfg.DIM_MODEL, k=model_cfg.KNN,
                         heads=model_cfg.N_HEADS, layers=model_cfg.N_LAYERS)
