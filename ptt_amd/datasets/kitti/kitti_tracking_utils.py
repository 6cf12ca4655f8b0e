"""Mirror of the part of ptt/datasets/kitti/kitti_tracking_utils.py the sequential tracking loop calls
(tools/eval_utils/eval_tracking_utils.py:8-13 imports PointCloud, get_box_by_offset, and uses crop_center_pc,
regularize_pc, get_model through `kitti_utils`): same names and argument meaning.

What differs is WHERE the work happens. `PointCloud.points` is a (3, N) float32 tensor resident on the HIP device;
`crop_center_pc` / `get_model` / `regularize_pc` run the device kernels of ptt_amd/csrc/track_ops.hip
(ptt_crop_compact_f32, ptt_regularize_f32) and return device-resident results. Only box arithmetic — a few dozen
float64 operations per frame — stays on the host (box_math.py). These functions are the drop-in, one-call-at-a-time
form (each call that must know a point COUNT synchronises, as the reference's numpy code implicitly does); the
tracking loop itself should use ptt_amd.tracklet_runner.TrackletRunner, which fuses a whole frame into two launches
plus the model graph and reads back one row per tracklet.

There is no CPU fallback: a host array is moved to the device, never processed on the host.
"""
import copy

import numpy as np
import torch

from ... import ops
from . import box_math as bm


class Quaternion(object):
    """The slice of pyquaternion.Quaternion the reference uses: Quaternion(matrix=R), Quaternion(axis=a, angle=t)
    (or radians= / degrees=), Quaternion(w, x, y, z), q1 * q2, .inverse, .rotation_matrix, .elements, .axis, .radians,
    .degrees. Unit-quaternion arithmetic in float64 (box_math.py)."""

    def __init__(self, *args, **kw):
        if 'matrix' in kw:
            self.q = bm.q_from_matrix(np.asarray(kw['matrix'], np.float64))
        elif 'axis' in kw:
            ang = kw.get('angle', kw.get('radians', None))
            if ang is None:
                ang = np.deg2rad(kw.get('degrees', 0.0))
            self.q = bm.q_from_axis_angle(np.asarray(kw['axis'], np.float64), np.float64(ang))
        elif 'array' in kw:
            self.q = np.array(kw['array'], np.float64)
        elif len(args) == 4:
            self.q = np.array(args, np.float64)
        elif len(args) == 1:
            a = args[0]
            self.q = np.array(a.q if isinstance(a, Quaternion) else a, np.float64)
        elif not args:
            self.q = np.array([1.0, 0.0, 0.0, 0.0])
        else:
            raise ValueError("unsupported Quaternion constructor arguments")

    def __mul__(self, other):
        return Quaternion(array=bm.q_mul(self.q, other.q))

    @property
    def inverse(self):
        return Quaternion(array=bm.q_inverse(self.q))

    @property
    def rotation_matrix(self):
        return bm.q_rotation_matrix(self.q)

    @property
    def elements(self):
        return self.q

    @property
    def radians(self):
        q = bm.q_normalised(self.q)
        n = np.linalg.norm(q[1:])
        return float(((2.0 * np.arctan2(n, q[0])) + np.pi) % (2 * np.pi) - np.pi)

    @property
    def degrees(self):
        return float(np.rad2deg(self.radians))

    @property
    def axis(self):
        v = bm.q_normalised(self.q)[1:]
        n = np.linalg.norm(v)
        return v / n if n > 1e-14 else np.array([0.0, 0.0, 0.0])

    def __repr__(self):
        return "Quaternion(%r, %r, %r, %r)" % tuple(self.q.tolist())


class Box(object):
    """Box (:68-160): center (3), wlh (3), orientation; the attributes the tracking loop reads and writes."""

    def __init__(self, center, size, orientation, label=np.nan, score=np.nan, velocity=(np.nan, np.nan, np.nan), name=None):
        assert not np.any(np.isnan(center)) and not np.any(np.isnan(size))
        assert len(center) == 3 and len(size) == 3
        self.center = np.array(center, np.float64)
        self.wlh = np.array(size, np.float64)
        self.orientation = orientation
        self.label = int(label) if not np.isnan(label) else label
        self.score = float(score) if not np.isnan(score) else score
        self.velocity = np.array(velocity)
        self.name = name

    @property
    def rotation_matrix(self):
        return self.orientation.rotation_matrix

    def translate(self, x):
        self.center += x

    def rotate(self, quaternion):
        self.center = np.dot(quaternion.rotation_matrix, self.center)
        self.orientation = quaternion * self.orientation
        self.velocity = np.dot(quaternion.rotation_matrix, self.velocity)

    def corners(self, wlh_factor=1.0):
        return bm.corners(self.center, self.wlh * wlh_factor, self.rotation_matrix)

    def bottom_corners(self):
        return self.corners()[:, [2, 3, 7, 6]]

    def __repr__(self):
        return "Box(center=%s, wlh=%s, q=%s)" % (self.center.tolist(), self.wlh.tolist(), self.orientation.q.tolist())


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("ptt.datasets.kitti.kitti_tracking_utils runs on the HIP device; none is visible "
                           "(there is no CPU fallback)")
    return torch.device('cuda', torch.cuda.current_device())


class PointCloud(object):
    """(3, N) float32 points resident on the device (PointCloud, :9-65)."""

    def __init__(self, points):
        if isinstance(points, np.ndarray):
            points = torch.from_numpy(np.ascontiguousarray(points[0:3], dtype=np.float32)).to(_device())
        elif not points.is_cuda:
            points = points.to(_device())
        if points.shape[0] > 3:
            points = points[0:3]
        self.points = points.to(torch.float32).contiguous()

    def nbr_points(self):
        return int(self.points.shape[1])


def get_box_by_offset(box, offset, use_z=False):
    """:186-216 for one box; `offset` (x, y, z, theta in degrees) is updated in place when the reference would redraw
    it (:205-208, from numpy's global generator — the same generator is used here)."""
    off = np.asarray(offset)
    off = off.astype(np.float32 if off.dtype == np.float32 else np.float64)      # the caller's precision (the loop hands over float32)
    c, q, used = bm.get_box_by_offset(box.center[None], box.wlh[None], box.orientation.q[None], off[None], use_z)
    try:
        offset[0], offset[1] = used[0, 0], used[0, 1]
    except TypeError:
        pass
    new_box = copy.deepcopy(box)
    new_box.center = c[0]
    new_box.orientation = Quaternion(array=q[0])
    return new_box


def _crop_jobs(pcs, params, outs, counts):
    """One ptt_crop_job per (cloud, bounds): `params` = box_math.crop_bounds over the same leading axis."""
    n = len(pcs)
    jobs = np.zeros(n, ops.CROP_JOB)
    for i, pc in enumerate(pcs):
        jobs['points'][i] = pc.points.data_ptr()
        jobs['ld'][i] = pc.points.stride(0)
        jobs['n_points'][i] = pc.points.shape[1]
        jobs['out'][i] = outs[i].data_ptr()
        jobs['capacity'][i] = outs[i].shape[0]
        jobs['count'][i] = counts[i:i + 1].data_ptr()
    for k in ('lo1', 'hi1', 'trans', 'lo2', 'hi2'):
        jobs[k] = params[k]
    jobs['rot'] = params['rot'].reshape(n, 9)
    return jobs


def _crop_on_device(pcs, boxes, offset, scale, extra2, label_boxes=None, label_offset=0.0, label_scale=1.0):
    """crop_center_pc of every (pc, box) pair in ONE launch -> (list of (cap,3) device buffers, counts int32 (n,)); with
    label_boxes (one ground-truth box per pair) also the survivors' labels against those boxes (get_label_by_box :238-272
    carried through the second crop), as a list of (cap,) uint8 device buffers."""
    dev = pcs[0].points.device
    center = np.stack([b.center for b in boxes])
    wlh = np.stack([b.wlh for b in boxes])
    quat = np.stack([b.orientation.q for b in boxes])
    params = bm.crop_bounds(center, wlh, quat, offset, scale, extra2)
    outs = [torch.empty((max(1, pc.nbr_points()), 3), dtype=torch.float32, device=dev) for pc in pcs]
    counts = torch.zeros(len(pcs), dtype=torch.int32, device=dev)
    jobs = _crop_jobs(pcs, params, outs, counts)
    labels = None
    if label_boxes is not None:
        lp = bm.crop_bounds(np.stack([b.center for b in label_boxes]), np.stack([b.wlh for b in label_boxes]),
                            np.stack([b.orientation.q for b in label_boxes]), label_offset, label_scale, 0.0)
        labels = [torch.zeros((o.shape[0],), dtype=torch.uint8, device=dev) for o in outs]
        jobs['ltrans'], jobs['llo'], jobs['lhi'] = lp['trans'], lp['lo2'], lp['hi2']
        jobs['lrot'] = lp['rot'].reshape(len(pcs), 9)
        for i, lab in enumerate(labels):
            jobs['label_out'][i] = lab.data_ptr()
    ops.crop_compact(ops.upload_jobs(jobs), len(pcs))
    return (outs, counts) if labels is None else (outs, counts, labels)


def crop_center_pc(pc, sample_box, gt_box=None, sample_offsets=None, offset=0.0, scale=1.0, normalize=False,
                   visual_handle=None, refine_box=True):
    """:300-339. Returns the cropped cloud in the sample box's frame (device-resident). With `gt_box` also the per-point
    labels (:308-312 get_label_by_box on the first crop, carried through the second crop :322; a device float64 0/1 tensor,
    the dtype get_label_by_box :269-271 returns — the same launch forms them) and the regression target label_reg (:325-329)."""
    extra2 = gt_box.wlh[1] * 0.6 if gt_box is not None else 0.0
    new_label = None
    if gt_box is not None:
        outs, counts, labels = _crop_on_device([pc], [sample_box], offset, scale, extra2, [gt_box],
                                               offset if refine_box else 0.0, scale if refine_box else 1.0)
    else:
        outs, counts = _crop_on_device([pc], [sample_box], offset, scale, extra2)
    n = int(counts.cpu()[0])
    if gt_box is not None:
        new_label = labels[0][:n].to(torch.float64)
    new_pc = PointCloud.__new__(PointCloud)
    new_pc.points = outs[0][:n].t().contiguous()
    if normalize:
        wlh = sample_box.wlh
        new_pc.points = new_pc.points / torch.tensor([[wlh[1]], [wlh[0]], [wlh[2]]], dtype=torch.float32,
                                                     device=new_pc.points.device)
    if gt_box is None:
        return new_pc
    label_reg = None
    if sample_offsets is not None:
        rot = np.transpose(sample_box.rotation_matrix)
        g = np.dot(rot, gt_box.center - sample_box.center)
        label_reg = np.array([g[0], g[1], g[2], -sample_offsets[-1]])
    return new_pc, new_label, label_reg


def get_model(PCs, boxes, offset=0., scale=1.0, normalize=False, visual_handle=None):
    """:219-236: the crops of several (cloud, box) pairs concatenated in order (device-resident)."""
    if len(PCs) == 0:
        return PointCloud(torch.ones((3, 0), dtype=torch.float32, device=_device()))
    outs, counts = _crop_on_device(list(PCs), list(boxes), offset, scale, 0.0)
    ns = counts.cpu().tolist()
    parts = [o[:n] for o, n in zip(outs, ns) if n > 0]
    pts = torch.cat(parts, 0) if parts else torch.ones((0, 3), dtype=torch.float32, device=outs[0].device)
    new_pc = PointCloud.__new__(PointCloud)
    new_pc.points = pts.t().contiguous()
    return new_pc


def regularize_pc(pc, input_size, ratio=1, label=None, reg=None, istrain=True):
    """:342-367: a fixed-size (input_size, 3) float32 cloud, resampled with replacement; returns a device tensor.
    istrain=False (the tracking loop): the index stream np.random.randint yields right after set_manual_seed(1), reproduced on
    the device (ptt_regularize_f32), numpy's global generator left as the reference leaves it. istrain=True (the data
    loader's form, :349-353 without the reseed): the indices are drawn here from numpy's running global generator, exactly as
    the reference draws them, and gathered on the device. With `label` (crop_center_pc's per-point labels, a device tensor)
    -> (points, label resampled with the same indices (:354-355; zeros for an (almost) empty crop :361-362), reg), float64
    labels in every branch and for both values of istrain."""
    if label is not None and input_size <= 0:
        return pc.points.t().contiguous(), label, reg
    if input_size > 0 and (istrain or label is not None):
        size = int(input_size) // int(ratio)
        rows = pc.points.t().contiguous()
        n = rows.shape[0]
        with_label = lambda pts, lab: pts if label is None else (pts, lab, reg)
        if n <= 2:
            return with_label(torch.zeros((size, 3), dtype=torch.float32, device=rows.device),
                              torch.zeros((size,), dtype=torch.float64, device=rows.device))
        if label is not None:
            label = label.to(torch.float64)                    # one dtype in every branch (the reference's: float64 0/1)
        if n == size:
            return with_label(rows, label)
        if not istrain:
            # the evaluation form with labels (:348-355): the same draws the tracking loop's form below reproduces on the
            # device, taken on the host here because the labels ride on the indices; numpy's global generator is left
            # seeded with 1 and advanced by exactly these draws, as the reference leaves it
            np.random.seed(1)
        idx = torch.from_numpy(np.random.randint(low=0, high=n, size=size, dtype=np.int64)).to(rows.device)
        return with_label(rows.index_select(0, idx), label.index_select(0, idx) if label is not None else None)
    if input_size <= 0:
        return pc.points.t().contiguous()
    size = int(input_size) // int(ratio)
    dev = pc.points.device
    rows = pc.points.t().contiguous()                          # (n,3) rows, as ptt_crop_compact_f32 writes them
    n = rows.shape[0]
    if n == 0:
        rows = torch.zeros((1, 3), dtype=torch.float32, device=dev)
    count = torch.tensor([n], dtype=torch.int32, device=dev)
    out = torch.empty((size, 3), dtype=torch.float32, device=dev)
    jobs = np.zeros(1, ops.REGULARIZE_JOB)
    jobs['seg'][0, 0] = rows.data_ptr()
    jobs['seg_count'][0, 0] = count.data_ptr()
    jobs['seg_capacity'][0, 0] = max(n, 1)
    jobs['out'][0] = out.data_ptr()
    jobs['n_seg'][0] = 1
    jobs['input_size'][0] = size
    info = torch.zeros(2, dtype=torch.int32, device=dev)
    jobs['info'][0] = info.data_ptr()
    ops.regularize(ops.upload_jobs(jobs), 1, ops.mt19937_draws(dev, max(8192, 4 * size + 1024)))
    # the reference reseeds numpy's GLOBAL generator here (set_manual_seed(1), :349-350) and draws from it (:351-353);
    # get_box_by_offset later redraws implausible offsets from that same generator (:205-208). Leave it in the state the
    # reference would: seeded with 1 and advanced by exactly the draws the resampling consumed (reported by the kernel).
    if n > 2 and n != size:
        used = int(info.cpu()[1])
        if used < 0:
            raise RuntimeError("ptt_regularize_f32 ran out of pre-drawn MT19937 outputs")
        np.random.seed(1)
        np.random.randint(low=0, high=n, size=size, dtype=np.int64)
    return out
