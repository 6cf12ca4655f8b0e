"""TEST INFRASTRUCTURE — CPU restatement (numpy) of the pre/post-processing of the reference's sequential tracking
loop, the checker for the N4 device path. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module; the product never does.

Restated, function by function, from /root/reference:
  crop_pc            ptt/datasets/kitti/kitti_tracking_utils.py:281-298
  crop_center_pc     :300-339 (points; with a ground-truth box also the per-point labels of get_label_by_box :238-272)
  get_model          :219-236
  regularize_pc      :342-367 with istrain=False (set_manual_seed(1) then np.random.randint)
  get_box_by_offset  :186-216
  Box.corners/rotate/translate :122-158, PointCloud.translate/rotate :44-49
  post_process       tools/eval_utils/eval_tracking_utils.py:266-274
  test_frame flow    tools/eval_utils/eval_tracking_utils.py:140-229 (track())

Pinned by tests/golden/G12_tracking_pre_post.npz: outputs of the reference's OWN functions, imported in the build
container (tests/golden/make_golden.py). The reference needs `pyquaternion`, which this image lacks; the generator
supplies a stand-in restating pyquaternion's published formulas (the same formulas as `_Quat` below), so quantities
that pass through quaternion algebra are pinned to "reference code + restated pyquaternion", and the crop / resample
fixtures additionally carry the rotation matrices as data. Arithmetic note: numpy >= 2 (NEP 50) evaluates
`float32_array + float64_scalar` and `float32_array < float64_scalar` in float64 — that is what this container runs
and what is restated; under numpy 1.x the reference itself would round the bounds / translation to float32 first.
"""
import copy

import numpy as np


class _Quat(object):
    """Unit-quaternion operations as pyquaternion documents them (w, x, y, z)."""

    def __init__(self, q):
        self.q = np.array(q, np.float64)

    @staticmethod
    def from_axis_angle(axis, angle):
        axis = np.array(axis, np.float64)
        mag_sq = np.dot(axis, axis)
        if abs(1.0 - mag_sq) > 1e-12:
            axis = axis / np.sqrt(mag_sq)
        theta = float(angle) / 2.0                    # pyquaternion uses math.cos / math.sin: float64 of the given angle
        return _Quat(np.hstack(([np.cos(theta)], axis * np.sin(theta))))

    @staticmethod
    def from_matrix(R):
        m = np.array(R, np.float64).conj().transpose()
        if m[2, 2] < 0:
            if m[0, 0] > m[1, 1]:
                t = 1 + m[0, 0] - m[1, 1] - m[2, 2]
                q = [m[1, 2] - m[2, 1], t, m[0, 1] + m[1, 0], m[2, 0] + m[0, 2]]
            else:
                t = 1 - m[0, 0] + m[1, 1] - m[2, 2]
                q = [m[2, 0] - m[0, 2], m[0, 1] + m[1, 0], t, m[1, 2] + m[2, 1]]
        else:
            if m[0, 0] < -m[1, 1]:
                t = 1 - m[0, 0] - m[1, 1] + m[2, 2]
                q = [m[0, 1] - m[1, 0], m[2, 0] + m[0, 2], m[1, 2] + m[2, 1], t]
            else:
                t = 1 + m[0, 0] + m[1, 1] + m[2, 2]
                q = [t, m[1, 2] - m[2, 1], m[2, 0] - m[0, 2], m[0, 1] - m[1, 0]]
        return _Quat(np.array(q, np.float64) * (0.5 / np.sqrt(t)))

    def _q_matrix(self):
        q = self.q
        return np.array([[q[0], -q[1], -q[2], -q[3]], [q[1], q[0], -q[3], q[2]],
                         [q[2], q[3], q[0], -q[1]], [q[3], -q[2], q[1], q[0]]])

    def _q_bar_matrix(self):
        q = self.q
        return np.array([[q[0], -q[1], -q[2], -q[3]], [q[1], q[0], q[3], -q[2]],
                         [q[2], -q[3], q[0], q[1]], [q[3], q[2], -q[1], q[0]]])

    def mul(self, other):
        return _Quat(np.dot(self._q_matrix(), other.q))

    @property
    def inverse(self):
        ss = np.dot(self.q, self.q)
        return _Quat(np.hstack((self.q[0:1], -self.q[1:4])) / ss)

    @property
    def rotation_matrix(self):
        n = np.sqrt(np.dot(self.q, self.q))
        if not abs(1.0 - n * n) < 1e-14 and n > 0:
            self.q = self.q / n
        return np.dot(self._q_matrix(), self._q_bar_matrix().conj().transpose())[1:][:, 1:]


class RefBox(object):
    """center (3), wlh (3), orientation quaternion — Box (:68-160) reduced to what the tracking loop touches."""

    def __init__(self, center, wlh, quat):
        self.center = np.array(center, np.float64)
        self.wlh = np.array(wlh, np.float64)
        self.quat = _Quat(quat.q if isinstance(quat, _Quat) else quat)

    @property
    def rotation_matrix(self):
        return self.quat.rotation_matrix

    def translate(self, x):
        self.center = self.center + x

    def rotate(self, q):
        self.center = np.dot(q.rotation_matrix, self.center)
        self.quat = q.mul(self.quat)

    def corners(self):
        w, l, h = self.wlh
        x = l / 2 * np.array([1, 1, 1, 1, -1, -1, -1, -1])
        y = w / 2 * np.array([1, -1, -1, 1, 1, -1, -1, 1])
        z = h / 2 * np.array([1, 1, -1, -1, 1, 1, -1, -1])
        c = np.dot(self.rotation_matrix, np.vstack((x, y, z)))
        c[0, :] = c[0, :] + self.center[0]
        c[1, :] = c[1, :] + self.center[1]
        c[2, :] = c[2, :] + self.center[2]
        return c


def crop_pc(points, box, offset=0, scale=1.0):
    """points (3,n) float32 -> the columns strictly inside the axis-aligned extent of the scaled box's corners."""
    tmp = copy.deepcopy(box)
    tmp.wlh = tmp.wlh * scale
    c = tmp.corners()
    maxi = np.max(c, 1) + offset
    mini = np.min(c, 1) - offset
    close = (points[0, :] > mini[0]) & (points[0, :] < maxi[0])
    close &= (points[1, :] > mini[1]) & (points[1, :] < maxi[1])
    close &= (points[2, :] > mini[2]) & (points[2, :] < maxi[2])
    return points[:, close]


def _inside(points, box, offset, scale):
    tmp = copy.deepcopy(box)
    tmp.wlh = tmp.wlh * scale
    c = tmp.corners()
    maxi = np.max(c, 1) + offset
    mini = np.min(c, 1) - offset
    close = (points[0, :] > mini[0]) & (points[0, :] < maxi[0])
    close &= (points[1, :] > mini[1]) & (points[1, :] < maxi[1])
    close &= (points[2, :] > mini[2]) & (points[2, :] < maxi[2])
    return close


def get_label_by_box(points, box, offset=0.0, scale=1.0):
    """:238-272: which columns of the (3,n) float32 cloud lie strictly inside `box` (scaled, grown by offset), tested in the
    box's own frame: a float32 copy of the cloud translated by -centre and rotated by the transposed rotation."""
    tmp = copy.deepcopy(box)
    pts = np.asarray(points, np.float32).copy()
    rot_mat = np.transpose(tmp.rotation_matrix)
    trans = -tmp.center
    for i in range(3):
        pts[i, :] = pts[i, :] + trans[i]
    tmp.translate(trans)
    pts[:3, :] = np.dot(rot_mat, pts[:3, :])
    tmp.rotate(_Quat.from_matrix(rot_mat))
    return _inside(pts, tmp, offset, scale)


def crop_center_pc_labels(points, sample_box, gt_box, offset=0.0, scale=1.0, refine_box=True):
    """crop_center_pc with a ground-truth box (:300-339): -> ((3,m) float32 cloud, (m,) bool labels)."""
    points = np.asarray(points, np.float32)
    first = _inside(points, sample_box, 2 * offset, 4 * scale)
    pts = points[:, first].copy()
    label = get_label_by_box(pts, gt_box, offset if refine_box else 0.0, scale if refine_box else 1.0)
    new_box = copy.deepcopy(sample_box)
    rot_mat = np.transpose(new_box.rotation_matrix)
    trans = -new_box.center
    for i in range(3):
        pts[i, :] = pts[i, :] + trans[i]
    new_box.translate(trans)
    pts[:3, :] = np.dot(rot_mat, pts[:3, :])
    new_box.rotate(_Quat.from_matrix(rot_mat))
    close = _inside(pts, new_box, offset + gt_box.wlh[1] * 0.6, 1 * scale)
    return pts[:, close], label[close]


def crop_center_pc(points, sample_box, gt_wlh1=None, offset=0.0, scale=1.0):
    """(3,n) float32 cloud -> (3,m) float32 cloud in the sample box's frame. gt_wlh1 = gt_box.wlh[1] when the caller
    passes a gt_box (the search crop), else None (template crops)."""
    pts = crop_pc(np.asarray(points, np.float32), sample_box, offset=2 * offset, scale=4 * scale).copy()
    new_box = copy.deepcopy(sample_box)
    rot_mat = np.transpose(new_box.rotation_matrix)
    trans = -new_box.center
    for i in range(3):
        pts[i, :] = pts[i, :] + trans[i]                 # float64 sum stored into the float32 array
    new_box.translate(trans)
    pts[:3, :] = np.dot(rot_mat, pts[:3, :])             # float64 product stored into the float32 array
    new_box.rotate(_Quat.from_matrix(rot_mat))
    if gt_wlh1 is not None:
        return crop_pc(pts, new_box, offset=offset + gt_wlh1 * 0.6, scale=1 * scale)
    return crop_pc(pts, new_box, offset=offset, scale=scale)


def get_model(clouds, boxes, offset=0.0, scale=1.0):
    if len(clouds) == 0:
        return np.ones((3, 0))
    points = np.ones((clouds[0].shape[0], 0))
    for pc, box in zip(clouds, boxes):
        cropped = crop_center_pc(pc, box, None, offset=offset, scale=scale)
        if cropped.shape[1] > 0:
            points = np.concatenate([points, cropped], axis=1)
    return points


def regularize_pc(points, input_size):
    """-> (input_size, 3) float32 (istrain=False form)."""
    pc = np.array(points, dtype=np.float32)
    if pc.shape[1] > 2:
        if pc.shape[1] != int(input_size):
            np.random.seed(1)                            # set_manual_seed(1) reseeds numpy's GLOBAL generator, which
            #                                              get_box_by_offset's redraw (:205-208) later draws from
            idx = np.random.randint(low=0, high=pc.shape[1], size=int(input_size), dtype=np.int64)
            pc = pc[:, idx]
        return pc.reshape((3, int(input_size))).T
    return np.zeros((3, int(input_size)), np.float32).T


def get_box_by_offset(box, offset, use_z=False, uniform=None):
    offset = np.array(offset)        # keeps the caller's dtype: the tracking loop passes float32 model outputs, so the
    #                                  angle below is a float32 product (NEP 50) and a redrawn offset rounds to float32
    rot_quat = _Quat.from_matrix(box.rotation_matrix)
    trans = np.array(box.center)
    new_box = copy.deepcopy(box)
    new_box.translate(-trans)
    new_box.rotate(rot_quat.inverse)
    new_box.rotate(_Quat.from_axis_angle([0, 0, 1], offset[-1] * np.pi / 180))
    uniform = uniform or (lambda: np.random.uniform(-1, 1))
    if offset[0] > new_box.wlh[0]:
        offset[0] = uniform()
    if offset[1] > min(new_box.wlh[1], 2):
        offset[1] = uniform()
    new_box.translate(np.array([offset[0], offset[1], offset[2] if use_z else 0]))
    new_box.rotate(rot_quat)
    new_box.translate(trans)
    return new_box


def post_process(pred_box_data):
    """(P,5) -> (offset (4), score): np.argmax over the scores, lowest index among equal maxima."""
    idx = pred_box_data[:, 4].argmax()
    return pred_box_data[idx, 0:4], pred_box_data[idx, 4]


def prepare_frame(clouds, gt_boxes, results, i, search_size=1024, template_size=512, offset=0.0, scale=1.25,
                  model_offset=0.0, model_scale=1.25):
    """prepare_search + prepare_template of frame i (REF_BOX previous_result, SHAPE_AGGREGATION firstandprevious):
    -> search (search_size,3), template (template_size,3) float32."""
    ref_box = results[-1]
    cand = crop_center_pc(clouds[i], ref_box, gt_boxes[i].wlh[1], offset=offset, scale=scale)
    search = regularize_pc(cand, search_size)
    model = get_model([clouds[0], clouds[i - 1]], [results[0], results[i - 1]], offset=model_offset, scale=model_scale)
    template = regularize_pc(model, template_size)
    return search, template


def track(clouds, gt_boxes, infer, use_z=True, **kw):
    """TrackingEvaluator.test_batch for one tracklet: `infer(search (1,S,3), template (1,T,3)) -> pred_box_data (P,5)`.
    Returns the result boxes (frame 0 = its ground-truth box)."""
    results = [copy.deepcopy(gt_boxes[0])]
    for i in range(1, len(clouds)):
        search, template = prepare_frame(clouds, gt_boxes, results, i, **kw)
        off, _ = post_process(np.asarray(infer(search[None], template[None])))
        results.append(get_box_by_offset(results[-1], off, use_z))
    return results
