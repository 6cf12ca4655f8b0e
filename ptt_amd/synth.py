"""Seeded synthetic (search, template) clouds shaped like the reference's inputs.

The reference resamples every cropped cloud WITH REPLACEMENT to a fixed size
(ptt/datasets/kitti/kitti_tracking_utils.py:342-367, `regularize_pc`), so exact duplicate
points are common and an empty crop becomes an all-zero cloud (:359-362). The generator
reproduces that distribution without any dataset.
"""
import numpy as np

SEARCH_BOX = np.array([4.8, 3.4, 1.6], np.float32)
TEMPLATE_BOX = np.array([2.4, 1.0, 0.9], np.float32)
CAR_SIGMA = np.array([1.6, 0.7, 0.5], np.float32)
PED_SIGMA = np.array([0.3, 0.3, 0.6], np.float32)


def _unique_points(rs, k, box, sigma, frac_uniform):
    ku = int(round(k * frac_uniform))
    uni = (rs.random_sample((ku, 3)).astype(np.float32) * 2.0 - 1.0) * box
    obj = rs.standard_normal((k - ku, 3)).astype(np.float32) * sigma
    obj = np.clip(obj, -box, box)
    return np.concatenate([uni, obj], 0).astype(np.float32)


def cloud(rs, n, k_unique, box, sigma, frac_uniform=0.7):
    """k_unique distinct points resampled with replacement to n (regularize_pc:351-353)."""
    if k_unique <= 0:
        return np.zeros((n, 3), np.float32)          # regularize_pc:359-362
    pts = _unique_points(rs, k_unique, box, sigma, frac_uniform)
    if k_unique >= n:
        return pts[:n].copy()
    sel = rs.randint(0, k_unique, size=n)
    return pts[sel]


def frames(seed, B, NS, NT, K_s=600, K_t=300, kind="car", zero_clouds=0):
    """Returns search (B,NS,3), template (B,NT,3) float32 arrays.
    kind='car' (config 2), 'ped' (config 3: sparse, heavy duplication), 'dense' (config 5: K=N)."""
    rs = np.random.RandomState(seed)
    s = np.empty((B, NS, 3), np.float32)
    t = np.empty((B, NT, 3), np.float32)
    for b in range(B):
        if kind == "ped":
            s[b] = cloud(rs, NS, K_s, SEARCH_BOX, PED_SIGMA, 0.2)
            t[b] = cloud(rs, NT, K_t, TEMPLATE_BOX, PED_SIGMA, 0.0)
        elif kind == "dense":
            s[b] = cloud(rs, NS, NS, SEARCH_BOX, CAR_SIGMA, 1.0)
            t[b] = cloud(rs, NT, NT, TEMPLATE_BOX, CAR_SIGMA, 1.0)
        else:
            s[b] = cloud(rs, NS, K_s, SEARCH_BOX, CAR_SIGMA, 0.7)
            t[b] = cloud(rs, NT, K_t, TEMPLATE_BOX, CAR_SIGMA, 0.0)
    for b in range(min(zero_clouds, B)):
        s[B - 1 - b] = 0.0
        t[B - 1 - b] = 0.0
    return s, t


def tracklet(seed, n_frames, n_obj=(100, 700), n_bg=(1500, 4000)):
    """A synthetic tracklet for the sequential tracking loop (tools/eval_utils/eval_tracking_utils.py:77-152): per frame
    a (3, N_i) float32 cloud — a box-shaped object cluster moving along a smooth path inside uniform background clutter
    within +-12 m (the reference pre-crops every scan to a neighbourhood of the ground-truth box,
    kitti_dataset_tracking.py:304-310) — and its ground-truth box as (center (3), wlh (3), quaternion (w,x,y,z)).
    Returns (clouds, boxes)."""
    rs = np.random.RandomState(seed)
    wlh = np.array([1.6 + 0.2 * rs.rand(), 3.9 + 0.5 * rs.rand(), 1.5])
    centers = np.cumsum(np.c_[rs.uniform(0.3, 0.9, n_frames), rs.uniform(-0.2, 0.2, n_frames),
                              rs.uniform(-0.03, 0.03, n_frames)], 0) + np.array([8.0, 2.0, -0.7])
    yaws = rs.uniform(-1, 1) + np.cumsum(rs.uniform(-0.05, 0.05, n_frames))
    clouds, boxes = [], []
    for i in range(n_frames):
        c, s_ = np.cos(yaws[i]), np.sin(yaws[i])
        Rz = np.array([[c, -s_, 0], [s_, c, 0], [0, 0, 1]])
        no, nb = int(rs.randint(*n_obj)), int(rs.randint(*n_bg))
        obj = (rs.uniform(-0.5, 0.5, (no, 3)) * np.array([wlh[1], wlh[0], wlh[2]])) @ Rz.T + centers[i]
        bg = rs.uniform(-1, 1, (nb, 3)) * np.array([12.0, 12.0, 2.0]) + centers[i]
        pts = np.concatenate([obj, bg], 0)[rs.permutation(no + nb)].astype(np.float32)
        clouds.append(np.ascontiguousarray(pts.T))
        boxes.append((centers[i].copy(), wlh.copy(), np.array([np.cos(yaws[i] / 2), 0.0, 0.0, np.sin(yaws[i] / 2)])))
    return clouds, boxes
