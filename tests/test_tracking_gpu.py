"""N4 on the GPU: the device-side pre/post-processing of the sequential tracking loop against fixture G12 (outputs of
the reference's own kitti_tracking_utils functions) and against the oracle's restatement of the whole loop."""
import os

import numpy as np
import pytest
import torch

from oracle import tracking_ref as TR
from ptt_amd import ops, synth
from ptt_amd.datasets.kitti import box_math as bm

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _g():
    return np.load(os.path.join(GOLD, "G12_tracking_pre_post.npz"))


def _kbox(g, kind, i):
    from ptt.datasets.kitti.kitti_tracking_utils import Box, Quaternion
    return Box(g["%s_center_%d" % (kind, i)], g["wlh"], Quaternion(array=g["%s_quat_%d" % (kind, i)]))


def test_G12_mirror_module_on_device_equals_the_reference(dev):
    """ptt.datasets.kitti.kitti_tracking_utils (device-backed mirror): crop_center_pc, get_model, regularize_pc
    bit-identical to the reference's outputs on every frame of the fixture tracklet."""
    import ptt.datasets.kitti.kitti_tracking_utils as ku
    g = _g()
    T = int(g["n_frames"])
    pcs = [ku.PointCloud(g["cloud_%d" % i]) for i in range(T)]
    for i in range(1, T):
        cand, _, _ = ku.crop_center_pc(pcs[i], _kbox(g, "ref", i), _kbox(g, "gt", i), offset=0.0, scale=1.25)
        np.testing.assert_array_equal(cand.points.cpu().numpy(), g["search_crop_%d" % i])
        np.testing.assert_array_equal(ku.regularize_pc(cand, 1024, istrain=False).cpu().numpy(), g["search_%d" % i])
        model = ku.get_model([pcs[0], pcs[i - 1]], [_kbox(g, "gt", 0), _kbox(g, "ref", i)], offset=0.0, scale=1.25)
        np.testing.assert_array_equal(model.points.cpu().numpy(), g["model_crop_%d" % i])
        np.testing.assert_array_equal(ku.regularize_pc(model, 512, istrain=False).cpu().numpy(), g["template_%d" % i])
    from ptt.datasets.kitti.kitti_tracking_utils import Box, Quaternion
    far = Box(g["far_center"], g["wlh"], Quaternion(array=g["far_quat"]))
    empty, _, _ = ku.crop_center_pc(pcs[1], far, _kbox(g, "gt", 1), offset=0.0, scale=1.25)
    assert empty.nbr_points() == 0
    assert float(ku.regularize_pc(empty, 1024, istrain=False).abs().max()) == 0.0         # :359-362 all-zero cloud


def test_G12_regularize_edge_cases(dev):
    """n = 0, 2 (zero cloud), 3, n == input_size (copied through), n just above a power of two, n >> size."""
    import ptt.datasets.kitti.kitti_tracking_utils as ku
    g = _g()
    for tag in ("n0", "n2", "n3", "n512", "n513", "n1024", "n1025", "n5000"):
        size = 512 if tag == "n512" else 1024
        got = ku.regularize_pc(ku.PointCloud(g["reg_in_" + tag]), size, istrain=False).cpu().numpy()
        np.testing.assert_array_equal(got, g["reg_out_" + tag], err_msg=tag)


@pytest.mark.parametrize("n", [3, 17, 100, 1023, 2049, 40000])
def test_resampling_index_stream_is_numpys(dev, n):
    """The gathered rows identify the indices: resample the cloud whose point k is (k, 0, 0) and compare with
    np.random.randint(0, n, 1024) after np.random.seed(1) (regularize_pc:349-353); info reports n and the draws used."""
    import ptt.datasets.kitti.kitti_tracking_utils as ku
    pts = np.zeros((3, n), np.float32)
    pts[0] = np.arange(n)
    got = ku.regularize_pc(ku.PointCloud(pts), 1024, istrain=False).cpu().numpy()
    np.random.seed(1)
    want = np.random.randint(low=0, high=n, size=1024, dtype=np.int64)
    np.testing.assert_array_equal(got[:, 0].astype(np.int64), want)


def test_regularize_pc_leaves_numpys_global_generator_where_the_reference_does(dev):
    """regularize_pc (istrain=False) reseeds numpy's GLOBAL generator with 1 and draws the resampling indices from it
    (kitti_tracking_utils.py:349-353); get_box_by_offset later redraws implausible offsets from that same generator
    (:205-208). The device-side mirror leaves the generator in exactly that state."""
    import ptt.datasets.kitti.kitti_tracking_utils as ku
    for n in (5, 700, 1024, 2):
        pts = np.zeros((3, n), np.float32)
        pts[0] = np.arange(n)
        np.random.seed(12345)
        before = np.random.get_state()[1].copy()
        ku.regularize_pc(ku.PointCloud(pts), 1024, istrain=False)
        got = np.random.uniform(-1, 1)
        if n > 2 and n != 1024:
            np.random.seed(1)
            np.random.randint(low=0, high=n, size=1024, dtype=np.int64)
        else:                                            # no resampling: the reference leaves the generator alone (:354-362)
            np.random.seed(12345)
            assert np.array_equal(before, np.random.get_state()[1])
        assert got == np.random.uniform(-1, 1), n


def test_regularize_pc_training_form_draws_from_numpys_running_generator(dev):
    """istrain=True (kitti_tracking_utils.py:349-353 without the reseed): the indices come from numpy's global generator in
    whatever state it is — the mirror draws them the same way and gathers on the device."""
    import ptt.datasets.kitti.kitti_tracking_utils as ku
    for n in (700, 3, 5000):
        pts = np.random.RandomState(n).standard_normal((3, n)).astype(np.float32)
        np.random.seed(777)
        got = ku.regularize_pc(ku.PointCloud(pts), 1024, istrain=True).cpu().numpy()
        np.random.seed(777)
        want = pts[:, np.random.randint(low=0, high=n, size=1024, dtype=np.int64)].T
        np.testing.assert_array_equal(got, want)
    same = np.random.RandomState(1).standard_normal((3, 1024)).astype(np.float32)
    np.testing.assert_array_equal(ku.regularize_pc(ku.PointCloud(same), 1024, istrain=True).cpu().numpy(), same.T)
    assert float(ku.regularize_pc(ku.PointCloud(same[:, :2]), 1024, istrain=True).abs().max()) == 0.0


def test_select_box_is_first_argmax(dev):
    rs = np.random.RandomState(3)
    x = rs.standard_normal((7, 64, 5)).astype(np.float32)
    x[1, 10, 4] = x[1, 40, 4] = 9.0                       # tie: the lower index wins
    x[2, :, 4] = 0.5                                      # all equal -> 0
    x[3, 63, 4] = 50.0
    out = ops.select_box(torch.from_numpy(x).to(dev)).cpu().numpy()
    for b in range(7):
        np.testing.assert_array_equal(out[b], x[b, x[b, :, 4].argmax()])


@pytest.mark.parametrize("batch,lengths", [(1, [6]), (3, [5, 3, 6]), (2, [4, 4, 3]), (6, [4, 3, 5, 2, 4])])
def test_tracklet_runner_equals_the_reference_loop(dev, batch, lengths):
    """TrackletRunner (clouds resident on the device, crop + resample + model graph + box selection per step, lockstep
    over `batch` tracklets, groups of tracklets when there are more than `batch`) against the oracle's restatement of
    TrackingEvaluator.test_batch driving the SAME tracker one frame at a time: every result box of every frame equal
    (the inputs the model sees are bit-identical, so are its outputs; the float64 box update agrees to 1e-9)."""
    from ptt_amd.config import StubDataset, ptt_model_cfg
    from ptt_amd.hot_path import randomize_
    from ptt_amd.models import build_network
    from ptt_amd.tracklet_runner import TrackletRunner
    tracker = randomize_(build_network(ptt_model_cfg(), 1, StubDataset()), seed=2).to(dev).eval()
    with torch.no_grad():                                  # small regression outputs, as a trained model's are:
        tracker.box_voting_head.refine_layer[-1].conv.weight.mul_(0.05)      # keeps the boxes on their objects
        tracker.box_voting_head.refine_layer[-1].conv.bias.mul_(0.05)
    tracklets = [synth.tracklet(100 + k, T) for k, T in enumerate(lengths)]
    runner = TrackletRunner(tracker, dev, batch=batch)
    got = runner.run(tracklets)

    def infer(search, template):
        with torch.no_grad():
            out = tracker({'search_points': torch.from_numpy(np.ascontiguousarray(search)).to(dev),
                           'template_points': torch.from_numpy(np.ascontiguousarray(template)).to(dev), 'batch_size': 1})
        return out['pred_box_data'][0].cpu().numpy()

    # batch 6: more than the "handful of tracklets" (ptt_amd.ops.CROP_JOBS_BY_VALUE_MAX / 2) — the job table is uploaded, the best
    # proposal is selected on the device and (B,5) rows come back; the model then runs its many-frame kernels, whose sums are
    # ordered differently from the one-frame chain the per-frame oracle loop takes (1e-5-level outputs): the boxes agree to 1e-4,
    # not bit for bit (a wrong slot, tracklet order or generator position would be off by decimetres)
    tol = 1e-9 if 2 * batch <= ops.CROP_JOBS_BY_VALUE_MAX else 1e-4
    n_moved = 0
    for (clouds, boxes), res in zip(tracklets, got):
        ref = TR.track(clouds, [TR.RefBox(*b) for b in boxes], infer, use_z=True)
        assert len(res) == len(ref) == len(clouds)
        for i, (r, o) in enumerate(zip(res, ref)):
            np.testing.assert_allclose(r[0], o.center, rtol=0, atol=tol, err_msg="frame %d centre" % i)
            np.testing.assert_allclose(bm.q_rotation_matrix(r[2]), o.rotation_matrix, rtol=0, atol=tol)
            n_moved += int(i > 0 and float(np.abs(r[0] - res[0][0]).max()) > 1e-6)
    assert n_moved > 0


def test_overlapped_runners_give_the_same_boxes(dev):
    """run_overlapped (two runners on two streams, their lockstep groups advancing alternately) == one runner, box for
    box: tracklets are independent, and a frame's result does not depend on which other frames share its batch."""
    from ptt_amd.config import StubDataset, ptt_model_cfg
    from ptt_amd.hot_path import randomize_
    from ptt_amd.models import build_network
    from ptt_amd.tracklet_runner import TrackletRunner, run_overlapped
    tracker = randomize_(build_network(ptt_model_cfg(), 1, StubDataset()), seed=4).to(dev).eval()
    tracklets = [synth.tracklet(500 + k, T) for k, T in enumerate([5, 3, 6, 4, 2, 5, 1])]
    single = TrackletRunner(tracker, dev, batch=3).run(tracklets)
    both = run_overlapped([TrackletRunner(tracker, dev, batch=2), TrackletRunner(tracker, dev, batch=2)], tracklets)
    assert len(both) == len(single) == len(tracklets)
    for a, b, (clouds, _) in zip(single, both, tracklets):
        assert len(a) == len(b) == len(clouds)
        for x, y in zip(a, b):
            np.testing.assert_array_equal(x[0], y[0])
            np.testing.assert_array_equal(x[2], y[2])


def test_G16_training_crop_labels_and_label_resampling_equal_the_reference(dev):
    """The data loader's crop (reference kitti_dataset_tracking.py:129-148): crop_center_pc with a ground-truth box returns the
    cropped cloud, the per-point labels (formed by the same launch) and the regression target; regularize_pc resamples points
    and labels with the same indices from numpy's running generator — all bit-identical to fixture G16 (the reference's outputs),
    for the shipped setting, a grown box, refine_box = False and an empty crop."""
    import ptt.datasets.kitti.kitti_tracking_utils as ku
    g, g16 = _g(), np.load(os.path.join(GOLD, "G16_crop_labels.npz"))
    pcs = [ku.PointCloud(g["cloud_%d" % i]) for i in range(int(g["n_frames"]))]
    for i in range(1, int(g16["n_frames"])):
        for k, (offset, scale, refine) in enumerate(g16["settings"]):
            pc, label, reg = ku.crop_center_pc(pcs[i], _kbox(g, "ref", i), _kbox(g, "gt", i), sample_offsets=g16["offsets_%d_%d" % (i, k)],
                                               offset=float(offset), scale=float(scale), refine_box=bool(refine))
            np.testing.assert_array_equal(pc.points.cpu().numpy(), g16["points_%d_%d" % (i, k)])
            assert label.dtype == torch.float64 and label.is_cuda             # get_label_by_box :269-271 returns float64 0/1
            np.testing.assert_array_equal(label.cpu().numpy(), g16["label_%d_%d" % (i, k)])
            np.testing.assert_allclose(reg, g16["reg_%d_%d" % (i, k)], rtol=0, atol=1e-12)
            if k == 0:
                np.random.seed(4000 + i)
                pts, cls, reg2 = ku.regularize_pc(pc, 1024, label=label, reg=reg)
                np.testing.assert_array_equal(pts.cpu().numpy(), g16["reg_points_%d" % i])
                np.testing.assert_array_equal(cls.cpu().numpy().astype(g16["reg_label_%d" % i].dtype), g16["reg_label_%d" % i])
                assert reg2 is reg and cls.dtype == torch.float64
                np.random.seed(77)                                            # the evaluation form reseeds with 1 itself (:349-350)
                pts, cls, _ = ku.regularize_pc(pc, 1024, label=label, reg=reg, istrain=False)
                np.testing.assert_array_equal(pts.cpu().numpy(), g16["eval_reg_points_%d" % i])
                np.testing.assert_array_equal(cls.cpu().numpy(), g16["eval_reg_label_%d" % i])
                np.testing.assert_array_equal(np.random.randint(0, 1 << 30, size=4, dtype=np.int64), g16["eval_next_draw_%d" % i])
    from ptt.datasets.kitti.kitti_tracking_utils import Box, Quaternion
    far = Box(g["far_center"], g["wlh"], Quaternion(array=g["far_quat"]))
    pc, label, reg = ku.crop_center_pc(pcs[1], far, _kbox(g, "gt", 1), sample_offsets=np.zeros(4, np.float32), offset=0.0, scale=1.25)
    assert pc.nbr_points() == 0 and label.numel() == 0
    pts, cls, _ = ku.regularize_pc(pc, 1024, label=label, reg=reg)
    np.testing.assert_array_equal(pts.cpu().numpy(), g16["empty_reg_points"])
    np.testing.assert_array_equal(cls.cpu().numpy(), g16["empty_reg_label"])
    assert cls.dtype == torch.float64                                         # the same dtype as the non-empty crops' labels
