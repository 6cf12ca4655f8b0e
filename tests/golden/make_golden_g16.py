"""Fixture G16 — the per-point labels of the REFERENCE's training crop (run in the build container against /root/reference,
like make_golden.py; only arrays are committed).

crop_center_pc with a ground-truth box (ptt/datasets/kitti/kitti_tracking_utils.py:300-339, the data loader's call at
kitti_dataset_tracking.py:129-138) returns, beside the cropped cloud, which of its points lie inside the ground-truth box
(get_label_by_box :238-272 on the first crop, carried through the second crop :322) and the regression target; regularize_pc
(:342-367) then resamples points and labels with the same indices. G16 = those outputs on the six-frame synthetic tracklet of
fixture G12 (its clouds and boxes are read back from G12_tracking_pre_post.npz), for the shipped search-area settings
(offset 0, scale 1.25), for a grown box (offset 0.3, scale 1.0), with refine_box False, and for an empty crop — each also
checked here against the repo's restatement (oracle.tracking_ref.crop_center_pc_labels).

    python tests/golden/make_golden_g16.py        # writes tests/golden/G16_crop_labels.npz
"""
import importlib.util
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tests.golden import make_golden as MG          # noqa: E402


def main():
    MG._install_stubs()
    sys.path.insert(0, MG.REF)                   # `ptt` = the reference's package (the repo root holds a drop-in alias of that name)
    from pyquaternion import Quaternion as PQ
    spec = importlib.util.spec_from_file_location("ref_kitti_tracking_utils", os.path.join(MG.REF, "ptt/datasets/kitti/kitti_tracking_utils.py"))
    ref_ku = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_ku)
    from oracle import tracking_ref as TR
    g12 = np.load(os.path.join(HERE, "G12_tracking_pre_post.npz"))
    T, wlh = int(g12["n_frames"]), g12["wlh"]
    box = lambda nm, i: ref_ku.Box(g12["%s_center_%d" % (nm, i)], wlh, PQ(array=g12["%s_quat_%d" % (nm, i)]))
    tb = lambda bx: TR.RefBox(bx.center, bx.wlh, bx.orientation.elements)
    out = {"n_frames": T, "settings": np.array([[0.0, 1.25, 1], [0.3, 1.0, 1], [0.2, 1.25, 0]])}     # offset, scale, refine_box
    n_pos = 0
    for i in range(1, T):
        cloud = g12["cloud_%d" % i]
        for k, (offset, scale, refine) in enumerate(out["settings"]):
            offs = np.array([0.1 * i, -0.05, 0.0, 2.0 + i], np.float32)
            pc, label, reg = ref_ku.crop_center_pc(ref_ku.PointCloud(cloud.copy()), box("ref", i), box("gt", i), sample_offsets=offs,
                                                   offset=float(offset), scale=float(scale), refine_box=bool(refine))
            o_pts, o_label = TR.crop_center_pc_labels(cloud, tb(box("ref", i)), tb(box("gt", i)), float(offset), float(scale), bool(refine))
            assert np.array_equal(o_pts, pc.points) and np.array_equal(o_label, label), (i, k)
            out["points_%d_%d" % (i, k)] = np.asarray(pc.points, np.float32)
            out["label_%d_%d" % (i, k)] = np.asarray(label, np.bool_)
            out["reg_%d_%d" % (i, k)] = np.asarray(reg, np.float64)
            out["offsets_%d_%d" % (i, k)] = offs
            n_pos += int(label.sum())
            if k == 0:
                # the data loader's resampling (:349-355): indices from numpy's global generator, seeded here
                np.random.seed(4000 + i)
                pts, cls, _ = ref_ku.regularize_pc(pc, 1024, label=label, reg=reg)
                out["reg_points_%d" % i], out["reg_label_%d" % i] = np.asarray(pts, np.float32), np.asarray(cls)
                # the evaluation form with labels (:348-355, istrain=False: set_manual_seed(1) in front of the draws), and
                # the first draw of numpy's global generator after it (the state the call leaves behind)
                np.random.seed(77)
                pts, cls, _ = ref_ku.regularize_pc(pc, 1024, label=label, reg=reg, istrain=False)
                out["eval_reg_points_%d" % i], out["eval_reg_label_%d" % i] = np.asarray(pts, np.float32), np.asarray(cls)
                out["eval_next_draw_%d" % i] = np.asarray(np.random.randint(0, 1 << 30, size=4, dtype=np.int64))
    far = ref_ku.Box(g12["far_center"], wlh, PQ(array=g12["far_quat"]))
    pc, label, reg = ref_ku.crop_center_pc(ref_ku.PointCloud(g12["cloud_1"].copy()), far, box("gt", 1), sample_offsets=np.zeros(4, np.float32),
                                           offset=0.0, scale=1.25)
    assert pc.points.shape[1] == 0 and label.shape[0] == 0
    pts, cls, _ = ref_ku.regularize_pc(pc, 1024, label=label, reg=reg)
    out["empty_reg_points"], out["empty_reg_label"] = np.asarray(pts, np.float32), np.asarray(cls)
    np.savez_compressed(os.path.join(HERE, "G16_crop_labels.npz"), **out)
    print("G16 written: %d crops, %d positive labels; oracle == reference bitwise" % ((T - 1) * 3, n_pos))


if __name__ == "__main__":
    main()
