"""N4 on the CPU: the oracle of the tracking loop's pre/post-processing against fixture G12 (outputs of the reference's
own crop_center_pc / get_model / regularize_pc / get_box_by_offset), and the product's HOST logic (float64 box math,
job tables) against the oracle. The device kernels are tested in tests/test_tracking_gpu.py."""
import os

import numpy as np

from oracle import tracking_ref as TR
from ptt_amd.datasets.kitti import box_math as bm

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _g():
    return np.load(os.path.join(GOLD, "G12_tracking_pre_post.npz"))


def _box(g, kind, i):
    return TR.RefBox(g["%s_center_%d" % (kind, i)], g["wlh"], g["%s_quat_%d" % (kind, i)])


def test_G12_oracle_crops_and_resampling_equal_the_reference():
    g = _g()
    T = int(g["n_frames"])
    clouds = [g["cloud_%d" % i] for i in range(T)]
    for i in range(1, T):
        cand = TR.crop_center_pc(clouds[i], _box(g, "ref", i), g["wlh"][1], 0.0, 1.25)
        np.testing.assert_array_equal(cand, g["search_crop_%d" % i])
        np.testing.assert_array_equal(TR.regularize_pc(cand, 1024), g["search_%d" % i])
        model = TR.get_model([clouds[0], clouds[i - 1]], [_box(g, "gt", 0), _box(g, "ref", i)], 0.0, 1.25)
        np.testing.assert_array_equal(model, g["model_crop_%d" % i])
        np.testing.assert_array_equal(TR.regularize_pc(model, 512), g["template_%d" % i])
        assert 2 < cand.shape[1] != 1024 and 2 < model.shape[1] != 512          # the resampling path is exercised
    for tag in ("n0", "n2", "n3", "n512", "n513", "n1024", "n1025", "n5000"):
        size = 512 if tag == "n512" else 1024
        np.testing.assert_array_equal(TR.regularize_pc(g["reg_in_" + tag], size), g["reg_out_" + tag], err_msg=tag)
    far = TR.RefBox(g["far_center"], g["wlh"], g["far_quat"])
    assert TR.crop_center_pc(clouds[1], far, g["wlh"][1], 0.0, 1.25).shape[1] == 0


def test_G12_box_update_oracle_and_product_host_math():
    """get_box_by_offset incl. the redraw branch (:205-208): oracle and the product's batched float64 math against the
    reference's outputs; the rotation matrices of the fixture boxes against both quaternion restatements."""
    g = _g()
    offs = g["gbo_offsets"]
    assert offs.dtype == np.float32
    box = _box(g, "ref", 2)
    for i in range(int(g["n_frames"])):
        for kind in ("gt", "ref"):
            q = g["%s_quat_%d" % (kind, i)]
            np.testing.assert_allclose(bm.q_rotation_matrix(q), g["%s_rot_%d" % (kind, i)], rtol=0, atol=1e-15)
            np.testing.assert_allclose(TR._Quat(q).rotation_matrix, g["%s_rot_%d" % (kind, i)], rtol=0, atol=1e-15)
    for use_z in (1, 0):
        for k in range(offs.shape[0]):
            want_c, want_q = g["gbo_center_%d_%d" % (use_z, k)], g["gbo_quat_%d_%d" % (use_z, k)]
            np.random.seed(77 + k)
            ob = TR.get_box_by_offset(box, offs[k].copy(), bool(use_z))
            np.testing.assert_allclose(ob.center, want_c, rtol=0, atol=1e-12)
            np.testing.assert_allclose(ob.quat.q, want_q, rtol=0, atol=1e-12)
            rs = np.random.RandomState(77 + k)
            c, q, used = bm.get_box_by_offset(box.center[None], box.wlh[None], box.quat.q[None], offs[k:k + 1].copy(),
                                              bool(use_z), uniform=lambda: rs.uniform(-1, 1))
            np.testing.assert_allclose(c[0], want_c, rtol=0, atol=1e-12)
            np.testing.assert_allclose(q[0], want_q, rtol=0, atol=1e-12)
            np.testing.assert_array_equal(used[0].astype(np.float32), g["gbo_used_%d_%d" % (use_z, k)])
    # batched call == per-box calls (no redraw among the first three offsets)
    c, q, _ = bm.get_box_by_offset(np.repeat(box.center[None], 3, 0), np.repeat(box.wlh[None], 3, 0),
                                   np.repeat(box.quat.q[None], 3, 0), offs[:3].copy(), True)
    for k in range(3):
        np.testing.assert_allclose(c[k], g["gbo_center_1_%d" % k], rtol=0, atol=1e-12)


def test_crop_bounds_match_the_oracles_box_arithmetic():
    """box_math.crop_bounds (what the device kernel receives) == the bounds the oracle's crop_pc derives, for the search
    crop (extra offset 0.6 * gt length) and the template crop."""
    g = _g()
    for i in range(1, int(g["n_frames"])):
        b = _box(g, "ref", i)
        p = bm.crop_bounds(b.center[None], b.wlh[None], b.quat.q[None], 0.0, 1.25, g["wlh"][1] * 0.6)
        import copy
        t = copy.deepcopy(b); t.wlh = t.wlh * 5.0
        c = t.corners()
        np.testing.assert_allclose(p["hi1"][0], c.max(1), rtol=0, atol=1e-12)
        np.testing.assert_allclose(p["lo1"][0], c.min(1), rtol=0, atol=1e-12)
        np.testing.assert_allclose(p["rot"][0], b.rotation_matrix.T, rtol=0, atol=1e-15)
        np.testing.assert_array_equal(p["trans"][0], -b.center)
        half = np.array([b.wlh[1], b.wlh[0], b.wlh[2]]) / 2 * 1.25 + g["wlh"][1] * 0.6
        np.testing.assert_allclose(p["hi2"][0], half, rtol=0, atol=1e-12)
        np.testing.assert_allclose(p["lo2"][0], -half, rtol=0, atol=1e-12)


def test_job_tables_mirror_the_c_structs():
    import ctypes
    from ptt_amd import _lib, ops
    for dt, st in ((ops.CROP_JOB, _lib.CropJob), (ops.REGULARIZE_JOB, _lib.RegularizeJob)):
        assert dt.itemsize == ctypes.sizeof(st)
        for name, _ in st._fields_:
            assert dt.fields[name][1] == getattr(st, name).offset, name
    draws = np.empty(64, np.uint32)
    assert _lib.lib().ptt_mt19937_fill(1, draws.ctypes.data, 64) == 0
    np.random.seed(1)
    np.testing.assert_array_equal(draws, np.random.randint(0, 2 ** 32, 64, dtype=np.uint32))


def test_c_host_box_math_equals_the_numpy_restatement():
    """ptt_track_crop_bounds / ptt_track_box_by_offset (the per-step host helpers TrackletRunner calls) against
    box_math.py on 48 random boxes, incl. the redraw branch with the generator position bookkeeping."""
    from ptt_amd import ops
    rs = np.random.RandomState(5)
    B = 48
    boxes = np.zeros(B, ops.TRACK_BOX)
    boxes['center'] = rs.standard_normal((B, 3)) * 5
    boxes['wlh'] = np.abs(rs.standard_normal((B, 3))) + 1
    q = rs.standard_normal((B, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    q[:24] = bm.q_from_axis_angle(np.tile([0, 0, 1.], (24, 1)), rs.uniform(-3, 3, 24))
    boxes['quat'] = q
    extra = np.abs(rs.standard_normal(B))
    jobs = np.zeros(2 * B, ops.CROP_JOB)
    ops.track_crop_bounds(boxes, 0.1, 1.25, extra, jobs[1::2], job_stride=2)
    p = bm.crop_bounds(boxes['center'], boxes['wlh'], boxes['quat'], 0.1, 1.25, extra)
    for k in ('lo1', 'hi1', 'trans', 'lo2', 'hi2'):
        np.testing.assert_allclose(jobs[k][1::2], p[k], rtol=0, atol=1e-13, err_msg=k)
        assert not jobs[k][0::2].any()
    np.testing.assert_allclose(jobs['rot'][1::2], p['rot'].reshape(B, 9), rtol=0, atol=1e-15)

    est = (rs.standard_normal((B, 5)) * 0.3).astype(np.float32)
    est[3, 0], est[5, 1] = 9.0, 7.0
    est[7, :2] = 5.0
    active = np.ones(B, np.int32)
    active[11] = 0
    got, e2, pos = boxes.copy(), est.copy(), np.full(B, 1500, np.int64)
    ops.track_box_by_offset(got, e2, True, active, pos)
    for b in range(B):
        if not active[b]:
            assert got[b] == boxes[b]
            continue
        r = np.random.RandomState(1)
        r.randint(0, 2 ** 32, 1500, dtype=np.uint32)            # numpy's generator after 1500 32-bit outputs
        c, qq, used = bm.get_box_by_offset(boxes['center'][b:b + 1], boxes['wlh'][b:b + 1], boxes['quat'][b:b + 1],
                                           est[b:b + 1, :4].copy(), True, uniform=lambda: r.uniform(-1, 1))
        np.testing.assert_allclose(got['center'][b], c[0], rtol=0, atol=1e-12)
        np.testing.assert_allclose(got['quat'][b], qq[0], rtol=0, atol=1e-14)
        np.testing.assert_array_equal(e2[b, :4], used[0].astype(np.float32))
    assert pos[3] == pos[5] == 1502 and pos[7] == 1504 and pos[0] == 1500


def test_G16_oracle_training_crop_labels_equal_the_reference():
    """crop_center_pc with a ground-truth box: cropped points and per-point labels (get_label_by_box carried through the second
    crop) of the oracle's restatement against the reference's outputs, three settings x five frames."""
    g, g16 = _g(), np.load(os.path.join(GOLD, "G16_crop_labels.npz"))
    pos = 0
    for i in range(1, int(g16["n_frames"])):
        for k, (offset, scale, refine) in enumerate(g16["settings"]):
            pts, label = TR.crop_center_pc_labels(g["cloud_%d" % i], _box(g, "ref", i), _box(g, "gt", i), float(offset), float(scale), bool(refine))
            np.testing.assert_array_equal(pts, g16["points_%d_%d" % (i, k)])
            np.testing.assert_array_equal(label, g16["label_%d_%d" % (i, k)])
            pos += int(label.sum())
            assert 0 < label.sum() < label.shape[0]                                # both classes present in every crop
    assert pos > 5000


def test_track_select_update_is_argmax_plus_generator_bookkeeping_plus_box_update():
    """ptt_track_select_update (a HOST function: runs here) against its three numpy / library steps: np.argmax of the scores (first
    among equals, first NaN), the generator position rule (template's draw count if it resampled, else the search's, else unchanged),
    ptt_track_box_by_offset; a negative draw count raises."""
    import pytest
    from ptt_amd import ops
    rs = np.random.RandomState(5)
    B, P = 6, 64
    prop = rs.standard_normal((B, P, 5)).astype(np.float32)
    prop[1, 10, 4] = prop[1, 40, 4] = 9.0                       # a tie: the first wins
    prop[2, 30, 4] = np.nan                                     # np.argmax returns the first NaN
    info = np.zeros((B, 2, 2), np.int32)
    info[:, 0, 1] = [100, 0, 7, 0, 50, 3]
    info[:, 1, 1] = [0, 0, 9, 20, 60, 0]
    def boxes():
        bx = np.zeros(B, ops.TRACK_BOX)
        bx['center'] = np.linspace(0, 1, B * 3).reshape(B, 3)
        bx['wlh'], bx['quat'][:, 0] = (1.7, 4.2, 1.5), 1.0
        return bx
    active = np.array([1, 1, 1, 0, 1, 1], np.int32)
    pos0 = np.arange(B).astype(np.int64) + 1000
    a, pa, est = boxes(), pos0.copy(), np.zeros((B, 5), np.float32)
    ops.track_select_update(prop, info, a, True, active, pa, est)
    best = np.argmax(prop[:, :, 4], axis=1)
    assert list(best[:3]) == [int(np.argmax(prop[0, :, 4])), 10, 30]
    ref_est = prop[np.arange(B), best].copy()
    used = np.where(info[:, 1, 1] > 0, info[:, 1, 1], info[:, 0, 1])
    pb = np.where(used > 0, used, pos0).astype(np.int64)
    b = boxes()
    ops.track_box_by_offset(b, ref_est, True, active, pb)
    np.testing.assert_array_equal(est, ref_est)
    np.testing.assert_array_equal(pa, pb)
    for k in ('center', 'quat', 'wlh'):
        np.testing.assert_array_equal(a[k], b[k])
    # rows already selected (the (B,5) read-back of a large batch)
    a2, pa2, est2 = boxes(), pos0.copy(), np.zeros((B, 5), np.float32)
    ops.track_select_update(np.ascontiguousarray(prop[np.arange(B), best]), info, a2, True, active, pa2, est2)
    np.testing.assert_array_equal(a2['center'], a['center'])
    info[4, 1, 1] = -1
    with pytest.raises(RuntimeError, match="ran out of pre-drawn MT19937 outputs"):
        ops.track_select_update(prop, info, boxes(), True, active, pos0.copy(), est)
