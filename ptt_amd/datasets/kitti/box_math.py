"""Float64 box / quaternion arithmetic of the tracking loop, vectorised over a leading batch axis (48 interleaved
tracklets update 48 boxes per step; one numpy call each instead of a Python loop).

Restates, in array form, the few operations the reference performs through `pyquaternion.Quaternion` and
`kitti_tracking_utils.Box` (ptt/datasets/kitti/kitti_tracking_utils.py:68-160, 186-216): unit quaternions (w, x, y, z),
Hamilton product, inverse, rotation matrix, matrix -> quaternion (the branch-by-largest-diagonal method pyquaternion
documents), box corners, and `get_box_by_offset`. pyquaternion itself is not installed in this image; its published
algorithms are what is restated.
"""
import numpy as np


def q_from_axis_angle(axis, angle):
    """(..,3), (..) -> (..,4)."""
    axis = np.asarray(axis, np.float64)
    angle = np.asarray(angle, np.float64)
    mag_sq = (axis * axis).sum(-1, keepdims=True)
    axis = np.where(np.abs(1.0 - mag_sq) > 1e-12, axis / np.sqrt(mag_sq), axis)
    half = angle[..., None] / 2.0
    return np.concatenate([np.cos(half), axis * np.sin(half)], -1)


def q_mul(a, b):
    """Hamilton product (..,4) x (..,4)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    aw, ax, ay, az = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    bw, bx, by, bz = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack([aw * bw - ax * bx - ay * by - az * bz,
                     ax * bw + aw * bx - az * by + ay * bz,
                     ay * bw + az * bx + aw * by - ax * bz,
                     az * bw - ay * bx + ax * by + aw * bz], -1)


def q_inverse(q):
    q = np.asarray(q, np.float64)
    ss = (q * q).sum(-1, keepdims=True)
    return q * np.array([1.0, -1.0, -1.0, -1.0]) / ss


def q_normalised(q):
    q = np.asarray(q, np.float64)
    n = np.sqrt((q * q).sum(-1, keepdims=True))
    return np.where((np.abs(1.0 - n * n) < 1e-14) | (n == 0), q, q / np.where(n == 0, 1.0, n))


def q_rotation_matrix(q):
    """(..,4) -> (..,3,3): the lower-right 3x3 of Q(q) . Qbar(q)^T for the normalised quaternion."""
    q = q_normalised(q)
    w, x, y, z = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    r = np.empty(q.shape[:-1] + (3, 3), np.float64)
    r[..., 0, 0] = x * x + w * w - z * z - y * y
    r[..., 0, 1] = x * y - w * z - z * w + y * x
    r[..., 0, 2] = x * z + w * y + z * x + y * w
    r[..., 1, 0] = y * x + z * w + w * z + x * y
    r[..., 1, 1] = y * y - z * z + w * w - x * x
    r[..., 1, 2] = y * z + z * y - w * x - x * w
    r[..., 2, 0] = z * x - y * w + x * z - w * y
    r[..., 2, 1] = z * y + y * z + x * w + w * x
    r[..., 2, 2] = z * z - y * y - x * x + w * w
    return r


def q_from_matrix(R):
    """(..,3,3) rotation matrices -> (..,4), branching on the diagonal as pyquaternion's trace method does."""
    R = np.asarray(R, np.float64)
    m = np.swapaxes(R, -1, -2)                        # the method is written for the row-vector convention
    m00, m11, m22 = m[..., 0, 0], m[..., 1, 1], m[..., 2, 2]
    c1 = m22 < 0
    c2 = m00 > m11
    c3 = m00 < -m11
    t_a = 1 + m00 - m11 - m22
    q_a = np.stack([m[..., 1, 2] - m[..., 2, 1], t_a, m[..., 0, 1] + m[..., 1, 0], m[..., 2, 0] + m[..., 0, 2]], -1)
    t_b = 1 - m00 + m11 - m22
    q_b = np.stack([m[..., 2, 0] - m[..., 0, 2], m[..., 0, 1] + m[..., 1, 0], t_b, m[..., 1, 2] + m[..., 2, 1]], -1)
    t_c = 1 - m00 - m11 + m22
    q_c = np.stack([m[..., 0, 1] - m[..., 1, 0], m[..., 2, 0] + m[..., 0, 2], m[..., 1, 2] + m[..., 2, 1], t_c], -1)
    t_d = 1 + m00 + m11 + m22
    q_d = np.stack([t_d, m[..., 1, 2] - m[..., 2, 1], m[..., 2, 0] - m[..., 0, 2], m[..., 0, 1] - m[..., 1, 0]], -1)
    sel_a, sel_b = c1 & c2, c1 & ~c2
    sel_c = ~c1 & c3
    t = np.where(sel_a, t_a, np.where(sel_b, t_b, np.where(sel_c, t_c, t_d)))
    q = np.where(sel_a[..., None], q_a, np.where(sel_b[..., None], q_b, np.where(sel_c[..., None], q_c, q_d)))
    return q * (0.5 / np.sqrt(t))[..., None]


_SX = np.array([1, 1, 1, 1, -1, -1, -1, -1], np.float64)
_SY = np.array([1, -1, -1, 1, 1, -1, -1, 1], np.float64)
_SZ = np.array([1, 1, -1, -1, 1, 1, -1, -1], np.float64)


def corners(center, wlh, R):
    """Box.corners (:140-158): (..,3), (..,3), (..,3,3) -> (..,3,8)."""
    w, l, h = wlh[..., 0:1], wlh[..., 1:2], wlh[..., 2:3]
    local = np.stack([l / 2 * _SX, w / 2 * _SY, h / 2 * _SZ], -2)           # (..,3,8)
    return np.matmul(R, local) + np.asarray(center, np.float64)[..., None]


def box_rotate(center, quat, q):
    """Box.rotate (:127-130): centre and orientation (velocity is not tracked)."""
    R = q_rotation_matrix(q)
    return np.einsum('...ij,...j->...i', R, center), q_mul(q, quat)


def get_box_by_offset(center, wlh, quat, offset, use_z=False, uniform=None):
    """kitti_tracking_utils.get_box_by_offset (:186-216), batched: boxes (B,3)/(B,3)/(B,4), offset (B,4) = x, y, z,
    theta in degrees. `uniform()` supplies the replacement np.random.uniform(-1, 1) draws of :205-208 (called once per
    triggered condition, x before y, box by box). Returns (center, quat) of the moved boxes and the offsets as used."""
    center = np.array(center, np.float64)
    wlh = np.asarray(wlh, np.float64)
    quat = np.asarray(quat, np.float64)
    offset = np.array(offset)                         # caller's dtype (float32 model outputs): the angle below is then a
    #                                                   float32 product and a redrawn offset rounds to float32, as in
    #                                                   the reference (numpy >= 2 scalar promotion)
    R0 = q_rotation_matrix(quat)
    rot_quat = q_from_matrix(R0)
    trans = center.copy()
    c = center - trans
    c, q = box_rotate(c, quat, q_inverse(rot_quat))
    angle = offset[..., -1] * np.pi / 180
    zaxis = np.broadcast_to(np.array([0.0, 0.0, 1.0]), c.shape)
    c, q = box_rotate(c, q, q_from_axis_angle(zaxis, angle))
    big_x = offset[..., 0] > wlh[..., 0]
    big_y = offset[..., 1] > np.minimum(wlh[..., 1], 2)
    if big_x.any() or big_y.any():
        if uniform is None:
            uniform = lambda: np.random.uniform(-1, 1)
        flat_off = offset.reshape(-1, offset.shape[-1])
        for b in np.nonzero((big_x | big_y).reshape(-1))[0]:
            if big_x.reshape(-1)[b]:
                flat_off[b, 0] = uniform()
            if big_y.reshape(-1)[b]:
                flat_off[b, 1] = uniform()
        offset = flat_off.reshape(offset.shape)
    step = np.stack([offset[..., 0], offset[..., 1], offset[..., 2] if use_z else np.zeros_like(offset[..., 2])], -1)
    c = c + step.astype(np.float64)
    c, q = box_rotate(c, q, rot_quat)
    c = c + trans
    return c, q, offset


def crop_bounds(center, wlh, quat, offset, scale, extra2=0.0):
    """The float64 quantities of one crop_center_pc (:300-339) per box, batched:
      lo1/hi1  crop_pc(pc, sample_box, offset=2*offset, scale=4*scale) in the cloud's frame (:301, :281-298)
      trans, rot   new_pc.translate(-center); new_pc.rotate(R^T)   (:313-317)
      lo2/hi2  crop_pc(new_pc, new_box, offset=offset + extra2, scale=scale) in the box frame (:320-330), where new_box
               is the sample box carried through the same translate / rotate (its corners come from the composed
               quaternion, as in the reference, not from an assumed identity)
    extra2 = gt_box.wlh[1] * 0.6 for the search crop (:321), 0 for template crops (:330)."""
    center = np.asarray(center, np.float64)
    wlh = np.asarray(wlh, np.float64)
    R = q_rotation_matrix(quat)
    c1 = corners(center, wlh * (4 * scale), R)
    hi1 = c1.max(-1) + 2 * offset
    lo1 = c1.min(-1) - 2 * offset
    rot = np.swapaxes(R, -1, -2)
    trans = -center
    nb_center = center + trans
    nb_center, nb_quat = box_rotate(nb_center, quat, q_from_matrix(rot))
    c2 = corners(nb_center, wlh * scale, q_rotation_matrix(nb_quat))
    off2 = offset + np.asarray(extra2, np.float64)
    hi2 = c2.max(-1) + np.asarray(off2)[..., None]
    lo2 = c2.min(-1) - np.asarray(off2)[..., None]
    return dict(lo1=lo1, hi1=hi1, trans=trans, rot=rot, lo2=lo2, hi2=hi2)
