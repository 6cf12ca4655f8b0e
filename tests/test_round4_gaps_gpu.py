"""Round-4 closures of the review's small parity gaps: the `pointnet2_ops._ext` drop-in shim called as the reference calls it,
a BatchNorm frozen inside a training model, the resampling draw table running out."""
import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import index_ops as O
from ptt_amd import ops, synth, train_ops
from ptt_amd.models.backbones_3d.pointnet2 import pytorch_utils as pt_utils

pytestmark = pytest.mark.gpu


def test_pointnet2_ops_ext_shim_with_the_reference_argument_order(dev):
    """`import pointnet2_ops._ext as _ext` (pointnet2_utils.py:24) resolves to this repo's shim; its six functions are called
    with the reference's argument order (pointnet2_utils.py:78,112,118,237,257,287) and checked against the CPU oracle."""
    import pointnet2_ops._ext as _ext
    rs = np.random.RandomState(4)
    s, _ = synth.frames(4, 2, 256, 64)
    xyz = torch.from_numpy(s).to(dev)
    # :78   _ext.furthest_point_sampling(xyz, npoint)
    inds = _ext.furthest_point_sampling(xyz, 64)
    assert inds.dtype == torch.int32 and np.array_equal(inds.cpu().numpy(), O.fps(s, 64))
    feats = torch.from_numpy(rs.standard_normal((2, 5, 256)).astype(np.float32)).to(dev)
    # :112  _ext.gather_points(features, idx)
    got = _ext.gather_points(feats, inds)
    np.testing.assert_array_equal(got.cpu().numpy(), O.gather(feats.cpu().numpy(), inds.cpu().numpy()))
    # :118  _ext.gather_points_grad(grad_out.contiguous(), idx, N)
    go = torch.from_numpy(rs.standard_normal((2, 5, 64)).astype(np.float32)).to(dev)
    gg = _ext.gather_points_grad(go.contiguous(), inds, 256)
    ref = torch.zeros(2, 5, 256).scatter_add_(2, inds.cpu().long()[:, None, :].expand(-1, 5, -1), go.cpu())
    np.testing.assert_allclose(gg.cpu().numpy(), ref.numpy(), atol=1e-6)
    # :287  _ext.ball_query(new_xyz, xyz, radius, nsample)   — centres FIRST
    new_xyz = torch.gather(xyz, 1, inds.long()[..., None].expand(-1, -1, 3)).contiguous()
    idx = _ext.ball_query(new_xyz, xyz, 0.5, 16)
    assert idx.dtype == torch.int32
    np.testing.assert_array_equal(idx.cpu().numpy(), O.ball_query(new_xyz.cpu().numpy(), s, 0.5, 16))
    # :237  _ext.group_points(features, idx)
    grouped = _ext.group_points(feats, idx)
    np.testing.assert_array_equal(grouped.cpu().numpy(), O.group(feats.cpu().numpy(), idx.cpu().numpy()))
    # :257  _ext.group_points_grad(grad_out.contiguous(), idx, N)
    g4 = torch.from_numpy(rs.standard_normal((2, 5, 64, 16)).astype(np.float32)).to(dev)
    gf = _ext.group_points_grad(g4.contiguous(), idx, 256)
    ref = torch.zeros(2, 5, 256).scatter_add_(2, idx.cpu().long().reshape(2, 1, -1).expand(-1, 5, -1), g4.cpu().reshape(2, 5, -1))
    np.testing.assert_allclose(gf.cpu().numpy(), ref.numpy(), atol=1e-5)
    # the ops PTT never reaches (pointnet2_utils.py:48,145,182,204) say so instead of computing something
    for name in ("three_nn", "three_interpolate", "three_interpolate_grad", "furthest_point_sampling_with_dist"):
        with pytest.raises(NotImplementedError):
            getattr(_ext, name)(xyz, xyz)
    # upstream's behaviour on a CPU tensor: an error, not a silent fallback
    with pytest.raises(RuntimeError):
        _ext.furthest_point_sampling(xyz.cpu(), 8)


def test_a_batchnorm_frozen_inside_a_training_model_takes_the_stock_path(dev):
    """model.train() with one BatchNorm put back in eval mode (fine-tuning with frozen statistics): the hand-written training
    path uses batch statistics, so such a stack must refuse it (train_ops.usable False), run on stock torch, normalise with the
    RUNNING statistics and leave them and num_batches_tracked untouched."""
    torch.manual_seed(0)
    mlp = pt_utils.SharedMLP([16, 32, 64], bn=True).to(dev).train()
    with torch.no_grad():
        for u in mlp:
            u.normlayer.bn.running_mean.normal_(0, 0.3)
            u.normlayer.bn.running_var.uniform_(0.5, 1.5)
    x = torch.randn(2, 16, 40, 8, device=dev)
    assert train_ops.usable(mlp, x)
    mlp[1].normlayer.bn.eval()
    assert not train_ops.usable(mlp, x)
    before = {k: v.clone() for k, v in mlp.state_dict().items()}
    y = train_ops.shared_mlp_pool(x, mlp, pool_dim=3) if train_ops.usable(mlp, x) else mlp(x).max(dim=3)[0]
    # layer 1 normalised with its running statistics: recompute it by hand from layer 0's (batch-statistics) output
    with torch.no_grad():
        h0 = mlp[0](x)
        bn = mlp[1].normlayer.bn
        z = mlp[1].conv(h0)
        ref = torch.relu((z - before['layer1.normlayer.bn.running_mean'][None, :, None, None])
                         / torch.sqrt(before['layer1.normlayer.bn.running_var'][None, :, None, None] + bn.eps)
                         * bn.weight[None, :, None, None] + bn.bias[None, :, None, None]).max(dim=3)[0]
    np.testing.assert_allclose(y.detach().cpu().numpy(), ref.cpu().numpy(), atol=1e-5, rtol=1e-5)
    after = mlp.state_dict()
    for k in ('layer1.normlayer.bn.running_mean', 'layer1.normlayer.bn.running_var', 'layer1.normlayer.bn.num_batches_tracked'):
        assert torch.equal(before[k], after[k]), k
    # the heads' Conv1d stacks gate the same way
    seq = pt_utils.Seq(8).conv1d(16, bn=True).conv1d(4, activation=None).to(dev).train()
    rows = torch.randn(2, 8, 30, device=dev)
    assert train_ops.conv1d_stack_usable(seq, rows)
    seq[0].normlayer.bn.eval()
    assert not train_ops.conv1d_stack_usable(seq, rows)


def test_usable_refuses_parameters_the_raw_pointer_path_cannot_take(dev):
    """float64 / non-contiguous BatchNorm parameters or buffers go to the stock path instead of raising in the middle of
    forward (the row kernels read them through raw pointers)."""
    mlp = pt_utils.SharedMLP([16, 32], bn=True).to(dev).train()
    x = torch.randn(2, 16, 40, 8, device=dev)
    assert train_ops.usable(mlp, x)
    mlp[0].normlayer.bn.double()
    assert not train_ops.usable(mlp, x)
    seq = pt_utils.Seq(8).conv1d(16, bn=True).conv1d(4, activation=None).to(dev).train()
    rows = torch.randn(2, 8, 30, device=dev)
    assert train_ops.conv1d_stack_usable(seq, rows)
    seq[0].conv.double()
    assert not train_ops.conv1d_stack_usable(seq, rows)


def test_running_out_of_resampling_draws_raises(dev):
    """regularize_pc draws indices by rejection from a pre-drawn MT19937 table; a table that is too short makes the kernel flag the
    cloud (draw count -1, NaN fill) and TrackletRunner raise — never a silently wrong cloud."""
    from ptt_amd.config import StubDataset, ptt_model_cfg
    from ptt_amd.hot_path import randomize_
    from ptt_amd.models import build_network
    from ptt_amd.tracklet_runner import TrackletRunner
    tracker = randomize_(build_network(ptt_model_cfg(), 1, StubDataset()), seed=0).to(dev).eval()
    runner = TrackletRunner(tracker, dev, batch=1)
    runner.draws = ops.mt19937_draws(dev, 64)            # 1024 points need at least 1024 draws
    clouds, boxes = synth.tracklet(9100, 3)
    with pytest.raises(RuntimeError, match="ran out of pre-drawn MT19937 outputs"):
        runner.run([(clouds, boxes)])
