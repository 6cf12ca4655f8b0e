from .layer_utils import index_points, square_distance

__all__ = ["square_distance", "index_points"]
