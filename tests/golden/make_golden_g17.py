"""Fixture G17 — what the REFERENCE itself can say about ball query and about farthest-point sampling near the origin (run in
the build container against /root/reference, like make_golden.py; only arrays are committed).

The arithmetic of `ball_query` / `furthest_point_sample` lives in upstream pointnet2_ops CUDA, absent from the image (SURVEY.md
8c). Two things ARE reference-held and pin the oracle (and through it the HIP kernels) from outside this build:

(A) WHICH points lie inside a ball. `square_distance` (ptt/models/model_utils/layer_utils.py:12-26) is the reference's own fp32
    difference-form distance. For every centre of the KITTI (M, N, radius, nsample) calls of SURVEY.md 8a F3 — the six backbone
    levels, the box head's, the north-star 2048-point level 0 and a 16384-point cloud (the uniform-grid form of the HIP ball
    query) — on car, pedestrian and all-zero clouds: the number of points k with square_distance(centre, p_k) < float32(r)^2 and
    the first `nsample` of them in ascending order (-1 padded). The oracle's / kernels' slots 0 .. min(count, nsample) - 1 must
    be exactly that list; what fills the REMAINING slots (the first hit; zeros when there is none) stays upstream's documented
    rule restated by this build — the one part of ball query no reference-held vector can pin.
(B) FPS with points inside the origin ball. Upstream skips points with |p|^2 <= 1e-3 (neither updated nor selectable). The
    reference's own numpy FPS (ptt/utils/common_utils.py:78-112, `fps_downsample`) has no such rule — but run on the cloud WITH
    THOSE POINTS REMOVED it must select the same points: expected indices = positions in the full cloud of the reference's
    picks among the kept points (point 0 kept: both start there).

    python tests/golden/make_golden_g17.py        # writes tests/golden/G17_ball_hits_origin_fps.npz, appends to GOLDEN_REPORT.txt
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tests.golden import make_golden as MG          # noqa: E402

# (name, N points, M centres, radius, nsample): SURVEY.md 8a F3
BALL_CASES = [("s0", 1024, 512, 0.3, 32), ("s1", 512, 256, 0.5, 32), ("s2", 256, 128, 0.7, 32), ("t0", 512, 256, 0.3, 32),
              ("t1", 256, 128, 0.5, 32), ("t2", 128, 64, 0.7, 32), ("box", 128, 64, 0.3, 16), ("s0_2048", 2048, 512, 0.3, 32),
              ("grid_16384", 16384, 1024, 0.3, 32)]


def main():
    MG._install_stubs()
    sys.path.insert(0, MG.REF)                   # `ptt` = the reference's package
    from ptt.models.model_utils.layer_utils import square_distance
    import ptt.utils.common_utils as ref_cu
    if not hasattr(np, "long"):
        np.long = np.int64                       # fps_downsample uses the removed alias (restored here only, as make_golden.py does)
    from oracle import index_ops as O
    from ptt_amd import synth

    out, n_centres, n_fps = {"ball_cases": np.array([c[0] for c in BALL_CASES])}, 0, 0
    rs = np.random.RandomState(1717)
    for name, N, M, r, ns in BALL_CASES:
        if N >= 16384:
            clouds = np.stack([synth.cloud(rs, N, N, synth.SEARCH_BOX, synth.CAR_SIGMA, 1.0),          # no duplicates (configs[4])
                               synth.cloud(rs, N, 600, synth.SEARCH_BOX, synth.CAR_SIGMA, 0.7)])       # heavy duplication
        else:
            clouds = np.stack([synth.cloud(rs, N, 600 if N >= 1024 else max(N // 2, 40), synth.SEARCH_BOX, synth.CAR_SIGMA, 0.7),
                               synth.cloud(rs, N, 60, synth.SEARCH_BOX, synth.PED_SIGMA, 0.2),
                               np.zeros((N, 3), np.float32)])
        clouds = np.ascontiguousarray(clouds, np.float32)
        sel = O.fps(clouds, M).astype(np.int32)                                                      # centres: inputs of the case
        centres = np.take_along_axis(clouds, sel[:, :, None].astype(np.int64), 1)
        d = square_distance(torch.from_numpy(centres), torch.from_numpy(clouds))                     # (B,M,N) fp32, the reference's
        assert d.dtype == torch.float32
        hit = (d < torch.tensor(np.float32(r) * np.float32(r))).numpy()
        cnt = hit.sum(-1).astype(np.int32)
        first = np.full(hit.shape[:2] + (ns,), -1, np.int32)
        for b in range(hit.shape[0]):
            for j in range(hit.shape[1]):
                k = np.flatnonzero(hit[b, j])[:ns]
                first[b, j, :k.size] = k
        # the oracle agrees on the pinned part (asserted here too; tests/test_oracle_cpu.py holds it to the committed arrays)
        mine = O.ball_query(centres, clouds, r, ns)
        for b in range(hit.shape[0]):
            for j in range(hit.shape[1]):
                c = min(int(cnt[b, j]), ns)
                assert np.array_equal(mine[b, j, :c], first[b, j, :c]), (name, b, j)
        out.update({"%s_xyz" % name: clouds, "%s_sel" % name: sel, "%s_count" % name: cnt, "%s_first" % name: first,
                    "%s_rn" % name: np.array([r, ns], np.float64)})
        n_centres += cnt.size

    # (B) origin-ball FPS
    fps_cases = []
    for ci, (n, k_unique, kind, n_origin) in enumerate(((1024, 600, "car", 40), (512, 300, "car", 25), (1024, 60, "ped", 100),
                                                         (256, 256, "dense", 8), (128, 40, "car", 30))):
        sig = synth.PED_SIGMA if kind == "ped" else synth.CAR_SIGMA
        pts = synth.cloud(rs, n, k_unique, synth.SEARCH_BOX, sig, 1.0 if kind == "dense" else 0.7)
        near = (pts * pts).sum(1) <= 2e-3
        pts[near] += np.float32(0.25)                                    # first: nothing near the origin by accident
        pos = rs.choice(np.arange(1, n), n_origin, replace=False)        # point 0 stays outside: FPS starts there
        tiny = rs.standard_normal((n_origin, 3)).astype(np.float32)
        tiny *= (rs.uniform(0.0, 0.03, (n_origin, 1)) / np.maximum(np.linalg.norm(tiny, axis=1, keepdims=True), 1e-9)).astype(np.float32)
        tiny[: n_origin // 3] = 0.0                                      # exact zeros (regularize_pc's padding) among them
        pts[pos] = tiny                                                  # |p| <= 0.03 < sqrt(1e-3) = 0.0316
        pts = np.ascontiguousarray(pts, np.float32)
        mag = (pts[:, 0] * pts[:, 0] + pts[:, 1] * pts[:, 1]) + pts[:, 2] * pts[:, 2]
        keep = np.flatnonzero(mag > np.float32(1e-3))
        assert keep[0] == 0 and keep.size == n - n_origin, (keep.size, n, n_origin)
        out["fps_cloud_%d" % ci] = pts
        for m in sorted({64, keep.size // 2, min(keep.size, n // 2 + 100)}):
            seed = next(sd for sd in range(100000) if np.random.RandomState(sd).randint(0, keep.size, (1,))[0] == 0)
            np.random.seed(seed)
            ref = np.asarray(ref_cu.fps_downsample(pts[keep], m, id=True)).reshape(-1)
            expect = keep[ref].astype(np.int32)
            mine = O.fps(pts[None], m)[0]
            assert np.array_equal(mine, expect), ("oracle FPS with origin-ball points != reference FPS on the kept points", ci, m,
                                                  int(np.argmax(mine != expect)))
            out["fps_idx_%d_%d" % (ci, m)] = expect
            fps_cases.append((ci, m))
            n_fps += 1
    out["fps_cases"] = np.array(fps_cases, np.int32)
    np.savez_compressed(os.path.join(HERE, "G17_ball_hits_origin_fps.npz"), **out)
    line = ("G17 ball-query hit sets under the reference's square_distance (layer_utils.py:12-26): %d centres over %d (M, N, r, nsample) calls "
            "on car / ped / zero clouds incl. a 16384-point cloud, oracle slots 0..min(count, nsample)-1 == reference's ascending in-ball "
            "list; FPS with origin-ball points == reference fps_downsample on the kept points, %d (cloud, npoint) cases"
            % (n_centres, len(BALL_CASES), n_fps))
    rep = os.path.join(HERE, "GOLDEN_REPORT.txt")
    lines = [l for l in open(rep).read().splitlines() if not l.startswith("G17 ")]
    with open(rep, "w") as fh:
        fh.write("\n".join(lines + [line]) + "\n")
    print(line)


if __name__ == "__main__":
    main()
