"""GPU parity of the index ops against the CPU oracle: bit-exact (int32 indices, fp32 copies).

Covers the reference's call sites pointnet2_utils.py:78 (FPS), :112/:118 (gather), :237/:257
(group), :287 (ball query) and variants.py:150-151 (kNN), on the input distribution the
reference produces (duplicates from resampling, all-zero clouds, points in the 1e-3 origin
ball, under-filled balls).
"""
import numpy as np
import pytest
import torch

from oracle import index_ops as O
from ptt_amd import ops, synth

pytestmark = pytest.mark.gpu


def _dev(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _clouds(seed, B, N, kind="car", K=None, zero=0):
    s, _ = synth.frames(seed, B, N, 64, K_s=K if K is not None else max(8, int(N * 0.6)), K_t=32, kind=kind,
                        zero_clouds=zero)
    return s


@pytest.mark.parametrize("N,npoint", [(64, 32), (128, 64), (200, 77), (256, 256), (512, 256), (1024, 512),
                                      (2048, 512), (4096, 1024), (8192, 512), (16384, 64),
                                      (4096, 4096), (3000, 2500), (4096, 3800)])   # index buffer + cloud copy around the 64 KB LDS line
def test_fps_matches_oracle(dev, N, npoint):
    xyz = _clouds(N, 3, N)
    got = ops.furthest_point_sampling(_dev(xyz, dev), npoint).cpu().numpy()
    ref = O.fps(xyz, npoint)
    assert got.dtype == np.int32
    np.testing.assert_array_equal(got, ref)


def test_fps_edge_cases(dev):
    rs = np.random.RandomState(7)
    B, N = 6, 1024
    xyz = _clouds(11, B, N, kind="ped", K=40)          # heavy duplication: FPS exhausts unique points
    xyz[1] = 0.0                                         # all-zero cloud (regularize_pc:359-362)
    xyz[2, :300] = rs.uniform(-0.015, 0.015, (300, 3))   # many points inside the 1e-3 origin ball
    xyz[3, 0] = 0.0                                      # start point itself is skipped
    xyz[4] = rs.uniform(-1, 1, (N, 3)).round(1)          # coarse grid => many exact distance ties
    got = ops.furthest_point_sampling(_dev(xyz, dev), 512).cpu().numpy()
    np.testing.assert_array_equal(got, O.fps(xyz, 512))
    assert (got[1] == 0).all()


@pytest.mark.parametrize("N,M,r,ns", [(1024, 512, 0.3, 32), (512, 256, 0.5, 32), (256, 128, 0.7, 32),
                                      (128, 64, 0.3, 16), (2048, 512, 0.3, 32), (100, 37, 0.4, 8),
                                      (16384, 256, 0.3, 32), (300, 64, 5.0, 64)])
def test_ball_query_matches_oracle(dev, N, M, r, ns):
    xyz = _clouds(N + M, 4, N)
    xyz[3] = 0.0
    centres = xyz[:, :M].copy()
    centres[0] += 100.0                                  # no hits at all => zeros
    got = ops.ball_query(_dev(centres, dev), _dev(xyz, dev), r, ns).cpu().numpy()
    ref = O.ball_query(centres, xyz, r, ns)
    np.testing.assert_array_equal(got, ref)
    assert (got[0] == 0).all()


def test_gather_group_and_grads(dev):
    rs = np.random.RandomState(3)
    B, C, N, M, ns = 3, 19, 257, 64, 16
    feat = rs.standard_normal((B, C, N)).astype(np.float32)
    idx1 = rs.randint(0, N, (B, M)).astype(np.int32)
    idx2 = rs.randint(0, N, (B, M, ns)).astype(np.int32)
    np.testing.assert_array_equal(ops.gather_points(_dev(feat, dev), _dev(idx1, dev)).cpu().numpy(),
                                  O.gather(feat, idx1))
    np.testing.assert_array_equal(ops.group_points(_dev(feat, dev), _dev(idx2, dev)).cpu().numpy(),
                                  O.group(feat, idx2))
    go1 = rs.standard_normal((B, C, M)).astype(np.float32)
    go2 = rs.standard_normal((B, C, M, ns)).astype(np.float32)
    # default: the deterministic scatter-add (entries added in ascending order = the oracle's sequential loop) => bit-exact
    np.testing.assert_array_equal(ops.gather_points_grad(_dev(go1, dev), _dev(idx1, dev), N).cpu().numpy(),
                                  O.gather_grad(go1, idx1, N))
    np.testing.assert_array_equal(ops.group_points_grad(_dev(go2, dev), _dev(idx2, dev), N).cpu().numpy(),
                                  O.group_grad(go2, idx2, N))


@pytest.mark.parametrize("B,C,N,M,ns", [(4, 131, 512, 256, 32), (2, 7, 2048, 512, 32), (3, 260, 128, 64, 16),
                                         (2, 5, 300, 37, 5), (1, 3, 16, 1, 1)])
def test_scatter_grads_deterministic_and_atomic(dev, B, C, N, M, ns):
    """Backward of group_points at the training shapes (and ragged ones): the deterministic path equals the oracle bit
    for bit, twice in a row; upstream's atomicAdd behaviour (ops.set_atomic_grads: LDS-accumulating kernel, global atomics
    beyond N = 16384) agrees to fp32 rounding. Heavy duplication: indices drawn from a quarter of the points."""
    rs = np.random.RandomState(B * 1000 + N)
    idx = rs.randint(0, max(1, N // 4), (B, M, ns)).astype(np.int32)
    idx[0, :, 0] = N - 1                                   # the last bin is used too
    go = rs.standard_normal((B, C, M, ns)).astype(np.float32)
    ref = O.group_grad(go, idx, N)
    a = ops.group_points_grad(_dev(go, dev), _dev(idx, dev), N).cpu().numpy()
    b = ops.group_points_grad(_dev(go, dev), _dev(idx, dev), N).cpu().numpy()
    np.testing.assert_array_equal(a, ref)
    np.testing.assert_array_equal(a, b)
    prev = ops.set_atomic_grads(True)
    try:
        c = ops.group_points_grad(_dev(go, dev), _dev(idx, dev), N).cpu().numpy()
        np.testing.assert_allclose(c, ref, rtol=1e-4, atol=1e-4)
        g1 = rs.standard_normal((B, C, M)).astype(np.float32)
        np.testing.assert_allclose(ops.gather_points_grad(_dev(g1, dev), _dev(idx[:, :, 0].copy(), dev), N).cpu().numpy(),
                                   O.gather_grad(g1, idx[:, :, 0].copy(), N), rtol=1e-4, atol=1e-4)
    finally:
        ops.set_atomic_grads(prev)


@pytest.mark.parametrize("N,k", [(64, 16), (128, 16), (100, 7), (512, 16), (2048, 16)])
def test_knn_matches_oracle(dev, N, k):
    xyz = _clouds(5 * N, 3, N, kind="ped", K=max(k, N // 3))   # duplicates => exact ties, broken by index
    xyz[2] = 0.0
    got = ops.knn(_dev(xyz, dev), k).cpu().numpy()
    np.testing.assert_array_equal(got, O.knn(xyz, k))


def test_cpu_tensors_are_rejected():
    with pytest.raises(RuntimeError):
        ops.furthest_point_sampling(torch.zeros(1, 8, 3), 4)
    with pytest.raises(RuntimeError):
        ops.ball_query(torch.zeros(1, 2, 3), torch.zeros(1, 8, 3), 0.3, 4)


def test_full_size_properties(dev):
    """Config-2 sizes (B=48, 2048 pts): properties that need no oracle."""
    s, _ = synth.frames(0, 48, 2048, 1024)
    xyz = _dev(s, dev)
    idx = ops.furthest_point_sampling(xyz, 512)
    assert int(idx.min()) >= 0 and int(idx.max()) < 2048 and (idx[:, 0] == 0).all()
    # K_s=600 unique points < 512? no: 600 >= 512, so FPS never repeats a coordinate
    sel = torch.gather(xyz, 1, idx.long()[..., None].expand(-1, -1, 3)).cpu().numpy()
    for b in range(0, 48, 7):
        assert len(np.unique(sel[b], axis=0)) == 512
    new_xyz = torch.gather(xyz, 1, idx.long()[..., None].expand(-1, -1, 3)).contiguous()
    bq = ops.ball_query(new_xyz, xyz, 0.3, 32).long()
    nb = torch.gather(xyz, 1, bq.reshape(48, -1)[..., None].expand(-1, -1, 3)).reshape(48, 512, 32, 3)
    d2 = ((nb - new_xyz[:, :, None]) ** 2).sum(-1)
    assert float(d2.max()) < 0.3 * 0.3 + 1e-6          # every returned neighbour is inside the ball
    first = bq[..., :1]
    assert (bq >= first).all()                          # first slot is the lowest-index hit


def test_randomised_shapes_against_oracle(dev):
    """40 random (B, N, npoint / M, radius, nsample / k) draws incl. tiny clouds, N not a multiple of the wave size,
    heavy duplication and coarse grids (exact ties): indices must stay bit-identical to the oracle."""
    rs = np.random.RandomState(2024)
    for trial in range(40):
        B = int(rs.randint(1, 5))
        N = int(rs.choice([1, 2, 3, 7, 31, 63, 64, 65, 100, 127, 129, 255, 257, 500, 777, 1023, 1025, 1500, 3000, 5000]))
        mode = trial % 4
        if mode == 0:
            xyz = rs.uniform(-3, 3, (B, N, 3)).astype(np.float32)
        elif mode == 1:
            base = rs.uniform(-3, 3, (B, max(1, N // 5), 3)).astype(np.float32)
            xyz = np.stack([base[b][rs.randint(0, base.shape[1], N)] for b in range(B)])          # duplicates
        elif mode == 2:
            xyz = rs.uniform(-1, 1, (B, N, 3)).round(1).astype(np.float32)                          # grid: exact ties
        else:
            xyz = rs.uniform(-0.05, 0.05, (B, N, 3)).astype(np.float32)                             # mostly inside the origin ball
        npoint = int(rs.randint(1, N + 1))
        t = _dev(xyz, dev)
        np.testing.assert_array_equal(ops.furthest_point_sampling(t, npoint).cpu().numpy(), O.fps(xyz, npoint),
                                      err_msg="fps trial %d N=%d npoint=%d mode=%d" % (trial, N, npoint, mode))
        M = int(rs.randint(1, min(N, 200) + 1))
        centres = xyz[:, rs.randint(0, N, M)] + rs.uniform(-0.1, 0.1, (B, M, 3)).astype(np.float32)
        r, ns = float(rs.choice([0.05, 0.3, 0.7, 2.0])), int(rs.choice([1, 4, 16, 32, 50]))
        np.testing.assert_array_equal(ops.ball_query(_dev(centres, dev), t, r, ns).cpu().numpy(),
                                      O.ball_query(centres, xyz, r, ns), err_msg="ball query trial %d" % trial)
        if N <= 1100:                                    # the O(N^2 k) oracle gets slow beyond this
            k = int(rs.randint(1, min(N, 20) + 1))
            np.testing.assert_array_equal(ops.knn(t, k).cpu().numpy(), O.knn(xyz, k), err_msg="knn trial %d" % trial)


def test_argument_errors_are_loud(dev):
    t = torch.zeros(1, 8, 3, device=dev)
    with pytest.raises(RuntimeError):
        ops.knn(t, 9)                                    # k > N
    with pytest.raises(RuntimeError):
        ops.ball_query(t, t.double(), 0.3, 4)            # dtype
    with pytest.raises(RuntimeError):
        ops.furthest_point_sampling(t.transpose(1, 2), 4)   # non-contiguous
    with pytest.raises(RuntimeError):
        ops.group_points(torch.zeros(1, 4, 8, device=dev), torch.zeros(1, 2, 3, device=dev), )   # idx dtype


@pytest.mark.parametrize("B,N,M,r,ns", [(24, 1024, 512, 0.3, 32), (20, 2048, 512, 0.3, 32), (48, 512, 256, 0.5, 32),
                                        (72, 130, 128, 0.7, 16), (40, 300, 252, 5.0, 64)])
def test_ball_query_four_centres_per_wave(dev, B, N, M, r, ns):
    """Launches of at least 8192 centres take the kernel in which a wave owns FOUR centres (a point is loaded once and
    tested against all four; a centre drops out when it has nsample hits): same indices, both entry points. Sparse clouds
    (duplicates), an all-zero cloud, a cloud whose first centres are far from everything, under-filled and over-full balls."""
    assert B * M >= 8192 and M % 4 == 0
    xyz = _clouds(7 * N + M, B, N, kind="ped" if N < 400 else "car", K=max(8, N // 4))
    xyz[3] = 0.0
    centres = xyz[:, :M].copy()
    centres[0, :6] += 100.0                              # centres 0-5 of cloud 0: no hits; 6, 7 share their wave with them
    got = ops.ball_query(_dev(centres, dev), _dev(xyz, dev), r, ns).cpu().numpy()
    np.testing.assert_array_equal(got, O.ball_query(centres, xyz, r, ns))
    assert (got[0, :6] == 0).all()
    xd = _dev(xyz, dev)
    sel = ops.furthest_point_sampling(xd, M)
    new_b, i64_b, idx_b = ops.centres_ball_query(xd, sel, M, r, ns)
    new_a, i64_a = ops.select_centres(xd, sel, M)
    assert torch.equal(new_a, new_b) and torch.equal(i64_a, i64_b)
    np.testing.assert_array_equal(idx_b.cpu().numpy(), O.ball_query(new_b.cpu().numpy(), xyz, r, ns))


@pytest.mark.parametrize("N,M,ns,with_sel", [(1024, 512, 32, True), (512, 256, 32, False), (128, 64, 16, True), (100, 7, 5, False)])
def test_centres_ball_query_equals_the_two_kernels(dev, N, M, ns, with_sel):
    """ptt_centres_ball_query_f32 (one launch per SA level) == ptt_select_centres_f32 followed by ptt_ball_query_f32, and
    through them the oracle."""
    xyz = _clouds(N + M, 4, N, kind="car", K=max(8, N // 3))
    xyz[3] = 0.0
    xd = _dev(xyz, dev)
    sel = ops.furthest_point_sampling(xd, M) if with_sel else None
    new_a, i64_a = ops.select_centres(xd, sel, M)
    idx_a = ops.ball_query(new_a, xd, 0.4, ns)
    new_b, i64_b, idx_b = ops.centres_ball_query(xd, sel, M, 0.4, ns)
    assert torch.equal(new_a, new_b) and torch.equal(idx_a, idx_b)
    assert (i64_a is None and i64_b is None) or torch.equal(i64_a, i64_b)
    np.testing.assert_array_equal(idx_b.cpu().numpy(), O.ball_query(new_b.cpu().numpy(), xyz, 0.4, ns))


def test_empty_batches_and_empty_sample_sets_are_no_ops(dev):
    """B = 0 (a rank whose shard is empty) and npoint / M = 0: every op returns a tensor of the right (empty) shape and
    launches nothing — the reference's ops behave the same way on empty inputs."""
    e = torch.zeros((0, 64, 3), device=dev)
    assert tuple(ops.furthest_point_sampling(e, 16).shape) == (0, 16)
    assert tuple(ops.ball_query(torch.zeros((0, 8, 3), device=dev), e, 0.3, 4).shape) == (0, 8, 4)
    assert tuple(ops.knn(e, 16).shape) == (0, 64, 16)
    new_xyz, i64, idx = ops.centres_ball_query(e, None, 8, 0.3, 4)
    assert tuple(new_xyz.shape) == (0, 8, 3) and tuple(idx.shape) == (0, 8, 4)
    x = torch.from_numpy(_clouds(3, 2, 64)).to(dev)
    assert tuple(ops.furthest_point_sampling(x, 0).shape) == (2, 0)
    assert tuple(ops.ball_query(x[:, :0].contiguous(), x, 0.3, 4).shape) == (2, 0, 4)
    assert tuple(ops.linear(torch.zeros((0, 32), device=dev), ops.pack_weight(torch.randn(64, 32, device=dev)), 64).shape) == (0, 64)
    torch.cuda.synchronize()
