"""`pointnet2_ops._ext` on libptt_hip.so: the entry points the reference calls, same names and argument
order (pointnet2_utils.py:48,78,112,118,145,182,204,237,257,287). Ops PTT never reaches raise."""
from ptt_amd.ops import (ball_query, furthest_point_sampling, furthest_point_sampling_with_dist, gather_points,
                         gather_points_grad, group_points, group_points_grad, three_interpolate,
                         three_interpolate_grad, three_nn)

__all__ = ["ball_query", "furthest_point_sampling", "furthest_point_sampling_with_dist", "gather_points",
           "gather_points_grad", "group_points", "group_points_grad", "three_interpolate",
           "three_interpolate_grad", "three_nn"]
