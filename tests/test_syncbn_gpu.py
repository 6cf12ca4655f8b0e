"""--sync_bn (tools/train_tracking.py:133-134): the SharedMLP + max-pool stage of ptt_amd/train_ops.py with nn.SyncBatchNorm
units, on two ranks that each hold half of a batch, against ONE process that holds the whole batch with plain BatchNorm.
SyncBatchNorm's contract is exactly that equivalence: same outputs, same input gradients, parameter gradients that add up,
same running statistics. The two ranks share cuda:0 and talk over gloo (RCCL refuses two ranks on one device); the
statistics exchange itself is backend-agnostic (one all-reduce of 2C + 1 float64 per layer and direction)."""
import os
import sys
import tempfile

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SPEC = [8, 64, 128]
B, M, NS = 4, 16, 8


def _make(seed=3):
    from ptt_amd.models.backbones_3d.pointnet2.pytorch_utils import SharedMLP
    torch.manual_seed(seed)
    mlp = SharedMLP(list(SPEC), bn=True)
    with torch.no_grad():
        for m in mlp.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.1)
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.randn(B, SPEC[0], M, NS, generator=g)
    wgt = torch.randn(B, SPEC[-1], M, generator=g)
    return mlp, x, wgt


def _run(mlp, x, wgt):
    from ptt_amd import train_ops
    x = x.clone().requires_grad_(True)
    assert train_ops.usable(mlp, x)
    y = train_ops.shared_mlp_pool(x, mlp, pool_dim=3)
    (y * wgt).sum().backward()
    out = {"y": y.detach(), "dx": x.grad}
    for n, p in mlp.named_parameters():
        out["g." + n] = p.grad
    for n, b in mlp.named_buffers():
        if "running" in n:
            out["b." + n] = b
    return {k: v.detach().cpu().numpy() for k, v in out.items()}


def _rank_main(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    mlp, x, wgt = _make()
    mlp = torch.nn.SyncBatchNorm.convert_sync_batchnorm(mlp).to(dev).train()
    assert any(isinstance(m, torch.nn.SyncBatchNorm) for m in mlp.modules())
    per = B // world
    res = _run(mlp, x[rank * per:(rank + 1) * per].to(dev), wgt[rank * per:(rank + 1) * per].to(dev))
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), **res)
    dist.barrier()
    dist.destroy_process_group()


def test_sync_bn_two_ranks_equal_one_big_batch():
    import torch.multiprocessing as mp
    dev = torch.device("cuda:0")
    mlp, x, wgt = _make()
    ref = _run(mlp.to(dev).train(), x.to(dev), wgt.to(dev))
    port = 29600 + (os.getpid() % 200)
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_rank_main, args=(2, port, d), nprocs=2, join=True)
        r = [dict(np.load(os.path.join(d, "rank%d.npz" % k))) for k in range(2)]

    def close(a, b, what, tol=2e-5):
        scale = max(1e-3, float(np.abs(b).max()))
        assert float(np.abs(a - b).max()) <= tol * scale, (what, float(np.abs(a - b).max()), scale)

    close(np.concatenate([r[0]["y"], r[1]["y"]]), ref["y"], "output")
    close(np.concatenate([r[0]["dx"], r[1]["dx"]]), ref["dx"], "input gradient", 1e-4)
    for k in ref:
        if k.startswith("g."):
            close(r[0][k] + r[1][k], ref[k], k, 1e-4)          # the loss is a sum over all frames: local gradients add up
        if k.startswith("b."):
            close(r[0][k], ref[k], k)
            close(r[1][k], ref[k], k)
