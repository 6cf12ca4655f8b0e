from .pointnet2_backbone import PointNet2BackboneLight

__all__ = {"PointNet2BackboneLight": PointNet2BackboneLight}
