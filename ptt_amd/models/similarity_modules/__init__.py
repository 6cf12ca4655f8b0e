from .p2b_xcoor import CosineSimAug

__all__ = {'CosineSimAug': CosineSimAug}
