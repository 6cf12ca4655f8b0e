"""Mirror of ptt/models/model_utils/layer_utils.py:12-40 (square_distance, index_points).

Plain torch; used by the training-mode (unfused) transformer path. The fused inference path
replaces both with ptt_knn_f32 and in-kernel gathers.
"""
import torch


def square_distance(src, dst):
    """(B,N,C),(B,M,C) -> (B,N,M) squared distances in difference form (layer_utils.py:26)."""
    diff = src.unsqueeze(2) - dst.unsqueeze(1)
    return (diff * diff).sum(-1)


def index_points(points, idx):
    """points (B,N,C), idx (B,S[,K]) int64 -> (B,S[,K],C) (layer_utils.py:29-40)."""
    shape = idx.shape
    flat = idx.reshape(shape[0], -1)
    out = torch.gather(points, 1, flat.unsqueeze(-1).expand(-1, -1, points.shape[-1]))
    return out.reshape(*shape, points.shape[-1])
