"""BASELINE.json configs[3] (nuScenes-Car shaped data-parallel training of the full tracker) on the GPU.

1. One rank at the full workload: DataParallelTrainer.step on synthetic_train_batch(seed, 48) (K_s = 200 unique points):
   finite loss, every parameter has a finite gradient that is non-zero unless it is mathematically zero in the REFERENCE
   model too (fixture G10 names those: biases in front of a softmax over neighbours), and two runs from the same seed
   leave bit-identical parameters (fixed-order BatchNorm sums, deterministic scatter-adds, fixed-order weight gradients).
2. Two ranks (gloo, both on cuda:0 — RCCL refuses two ranks on one device; the driver's SCALE run covers RCCL): the full
   tracker with its gradient all-reduce (the flat gradient buffer's one collective, and DistributedDataParallel) ON THE
   HAND-WRITTEN ROW KERNELS (train_ops.usable / pt_block_usable asserted
   true on both ranks — on the CPU/gloo test they are bypassed), different batches per rank: every gradient equals the
   mean of two single-process GPU runs, both replicas hold identical parameters after clip + Adam; once more with
   --sync_bn (SyncBatchNorm statistics exchanged by the row kernels' float64 sums).
Reference: tools/train_tracking.py:133-134,158-159, tools/train_utils/train_utils.py:47-51."""
import os
import sys
import tempfile

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
SEED = 41


def _build(dev):
    from ptt_amd.config import StubDataset, ptt_model_cfg
    from ptt_amd.models import build_network
    from tests.util import fill_state_dict_
    return fill_state_dict_(build_network(ptt_model_cfg(), 1, StubDataset(training=True)), SEED).to(dev).train()


def _count_paths():
    """Wrap the two gates of the hand-written training path and count their answers."""
    from ptt_amd import train_ops
    seen = {"mlp_true": 0, "mlp_false": 0, "pt_true": 0, "pt_false": 0}
    u0, p0 = train_ops.usable, train_ops.pt_block_usable

    def usable(mlp, x):
        r = u0(mlp, x)
        seen["mlp_true" if r else "mlp_false"] += 1
        return r

    def pt_block_usable(block, xyz, f):
        r = p0(block, xyz, f)
        seen["pt_true" if r else "pt_false"] += 1
        return r

    train_ops.usable, train_ops.pt_block_usable = usable, pt_block_usable
    return seen, lambda: (setattr(train_ops, "usable", u0), setattr(train_ops, "pt_block_usable", p0))


def _zero_grad_keys():
    g = np.load(os.path.join(GOLD, "G10_train_step.npz"))
    return {str(k) for k, n in zip(g["grad_keys"], g["grad_norms"]) if n <= 1e-3}


def test_config3_full_batch_step_is_finite_complete_and_bit_reproducible(dev):
    from ptt_amd.train_step import DataParallelTrainer, synthetic_train_batch
    seen, restore = _count_paths()
    try:
        runs = []
        for _ in range(2):
            trainer = DataParallelTrainer(_build(dev), dev)
            batch = synthetic_train_batch(100, 48, dev)                      # K_s = 200: nuScenes-Car sparsity
            loss = trainer.forward_backward(batch)
            grads = {k: p.grad.detach().clone() for k, p in trainer.tracker.named_parameters() if p.grad is not None}
            trainer2 = DataParallelTrainer(_build(dev), dev)
            loss2 = trainer2.step(batch)
            params = {k: p.detach().clone() for k, p in trainer2.tracker.named_parameters()}
            runs.append((float(loss.detach()), float(loss2.detach()), grads, params))
    finally:
        restore()
    assert seen["mlp_true"] > 0 and seen["mlp_false"] == 0 and seen["pt_true"] > 0 and seen["pt_false"] == 0, seen
    loss, loss2, grads, params = runs[0]
    assert np.isfinite(loss) and loss == loss2
    zero_ok = _zero_grad_keys()
    assert len(grads) == 106 and len(zero_ok) <= 12, sorted(zero_ok)
    for k, gr in grads.items():
        assert bool(torch.isfinite(gr).all()), k
        if k not in zero_ok:
            assert float(gr.abs().max()) > 0.0, k
    for k in params:
        assert bool(torch.isfinite(params[k]).all()), k
    # bit-reproducible: the same seed gives the same loss, gradients and updated parameters
    differing = [k for k in grads if not torch.equal(grads[k], runs[1][2][k])]
    assert runs[1][0] == loss and not differing, ("gradients differ between two identical runs", differing[:8], len(differing))
    differing = [k for k in params if not torch.equal(params[k], runs[1][3][k])]
    assert not differing, ("parameters differ between two identical runs", differing[:8])


def _rank_main(rank, world, port, outdir, sync_bn, reducer):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    from ptt_amd.train_step import DataParallelTrainer, synthetic_train_batch
    seen, restore = _count_paths()
    trainer = DataParallelTrainer(_build(dev), dev, sync_bn=sync_bn, reducer=reducer)
    assert trainer.world == world and trainer.ranks_seen() == world and trainer.collective
    if reducer == "ddp":
        assert isinstance(trainer.model, torch.nn.parallel.DistributedDataParallel) and trainer.sink is None
    else:
        assert trainer.sink is not None and not trainer.ddp
    if sync_bn:
        assert any(isinstance(m, torch.nn.SyncBatchNorm) for m in trainer.tracker.modules())
    batch = synthetic_train_batch(500 + rank, 2, dev)
    loss = trainer.forward_backward(batch)
    out = {"loss": np.float64(float(loss.detach()))}
    for k, p in trainer.tracker.named_parameters():
        out["g." + k] = p.grad.detach().cpu().numpy()
    trainer.step(batch)
    for k, p in trainer.tracker.named_parameters():
        out["p." + k] = p.detach().cpu().numpy()
    restore()
    out["seen"] = np.array([seen["mlp_true"], seen["mlp_false"], seen["pt_true"], seen["pt_false"]])
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), **out)
    dist.barrier()
    dist.destroy_process_group()


def _two_ranks(sync_bn, reducer="flat"):
    import torch.multiprocessing as mp
    port = 29800 + (os.getpid() % 150) + (50 if sync_bn else 0) + (200 if reducer == "ddp" else 0)
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_rank_main, args=(2, port, d, sync_bn, reducer), nprocs=2, join=True)
        return [dict(np.load(os.path.join(d, "rank%d.npz" % k))) for k in range(2)]


def _single(dev, batch_seed, reducer="flat"):
    from ptt_amd.train_step import DataParallelTrainer, synthetic_train_batch
    trainer = DataParallelTrainer(_build(dev), dev, reducer=reducer)
    trainer.forward_backward(synthetic_train_batch(batch_seed, 2, dev))
    return {k: p.grad.detach().cpu().numpy() for k, p in trainer.tracker.named_parameters() if p.grad is not None}


@pytest.mark.parametrize("reducer", ["flat", "ddp"])
def test_config3_two_rank_ddp_on_the_row_kernels(dev, reducer):
    """reducer: the flat gradient buffer with its one all-reduce (the trainer's default on a HIP device), and DistributedDataParallel."""
    r = _two_ranks(sync_bn=False, reducer=reducer)
    for k in (0, 1):
        mlp_t, mlp_f, pt_t, pt_f = (int(v) for v in r[k]["seen"])
        assert mlp_t > 0 and mlp_f == 0 and pt_t > 0 and pt_f == 0, (k, r[k]["seen"])       # the hand-written path ran
    g0, g1 = _single(dev, 500, reducer), _single(dev, 501, reducer)
    gmax = max(float(np.abs(v).max()) for v in g0.values())
    worst = 0.0
    for k in g0:
        mean = (g0[k] + g1[k]) / 2
        scale = max(float(np.abs(mean).max()), 1e-3 * gmax)
        for rk in (0, 1):
            err = float(np.abs(r[rk]["g." + k] - mean).max()) / scale
            worst = max(worst, err)
            assert err < 1e-4, (k, rk, err)
        assert np.array_equal(r[0]["g." + k], r[1]["g." + k]), k            # both ranks hold the same reduced gradient
    print("two-rank DDP on the row kernels: worst gradient error vs the mean of two single-process runs %.2e" % worst)
    k = 'backbone_3d.SA_modules.1.mlp_module.layer0.conv.weight'
    assert float(np.abs(g0[k] - g1[k]).max()) > 0                            # the batches differ
    for k in r[0]:
        if k.startswith("p."):
            assert np.array_equal(r[0][k], r[1][k]), k                       # identical replicas after clip + Adam
    assert np.isfinite(r[0]["loss"]) and np.isfinite(r[1]["loss"]) and r[0]["loss"] != r[1]["loss"]


def test_config3_two_rank_ddp_with_sync_bn_on_the_row_kernels(dev):
    r = _two_ranks(sync_bn=True)
    plain = _two_ranks(sync_bn=False)
    for k in (0, 1):
        mlp_t, mlp_f, pt_t, pt_f = (int(v) for v in r[k]["seen"])
        assert mlp_t > 0 and mlp_f == 0 and pt_t > 0 and pt_f == 0, (k, r[k]["seen"])       # SyncBatchNorm units stay on the row kernels
    changed = 0
    for k in r[0]:
        if k.startswith("g."):
            assert np.isfinite(r[0][k]).all(), k
            assert np.array_equal(r[0][k], r[1][k]), k
            changed += int(not np.array_equal(r[0][k], plain[0][k]))
        if k.startswith("p."):
            assert np.array_equal(r[0][k], r[1][k]), k
    assert changed > 50, changed          # statistics over both ranks' rows: not the per-rank-statistics gradients
    assert np.isfinite(r[0]["loss"]) and np.isfinite(r[1]["loss"])


def _graph_rank_main(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    from ptt_amd.config import StubDataset, ptt_model_cfg
    from ptt_amd.models import build_network
    from ptt_amd.train_step import DataParallelTrainer, synthetic_train_batch

    def make(graph):
        torch.manual_seed(1000 + rank)                      # ranks that did NOT seed alike: the trainer broadcasts rank 0's state
        model = build_network(ptt_model_cfg(), 1, StubDataset(training=True)).to(dev).train()
        return DataParallelTrainer(model, dev, graph=graph)
    out = {}
    batches = [synthetic_train_batch(700 + 10 * k + rank, 2, dev) for k in range(3)]
    for name, graph in (("eager", False), ("graph", True)):
        tr = make(graph)
        assert tr.collective and tr.world == world
        for k in range(7):
            loss = tr.step(batches[k % 3])
        torch.cuda.synchronize()
        out[name + ".loss"] = np.float64(float(loss.detach()))
        out[name + ".graph_steps"] = np.int64(tr.graph_steps)
        out[name + ".two_graphs"] = np.int64(tr.captured is not None and tr.captured.second is not None)
        for k, v in tr.tracker.state_dict().items():
            out[name + ".s." + k] = v.detach().cpu().numpy()
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), **out)
    dist.barrier()
    dist.destroy_process_group()


def test_config3_two_rank_captured_step_equals_the_eager_step(dev):
    """The captured training step with a REAL exchange: two ranks (gloo, sharing cuda:0), differently seeded models (the flat reducer
    broadcasts rank 0's parameters and buffers at construction), different batches per rank, seven steps = three eager, the capture, four
    replays of [forward + backward + finish] | all-reduce | [x 1 / world, clip + Adam]. Parameters after the last step: bit-identical
    to the same run with graph=False, and identical on both ranks (the BatchNorm running statistics are per rank by design)."""
    import torch.multiprocessing as mp
    port = 30300 + (os.getpid() % 150)
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_graph_rank_main, args=(2, port, d), nprocs=2, join=True)
        r = [dict(np.load(os.path.join(d, "rank%d.npz" % k))) for k in range(2)]
    for k in (0, 1):
        assert int(r[k]["graph.graph_steps"]) == 4 and int(r[k]["graph.two_graphs"]) == 1 and int(r[k]["eager.graph_steps"]) == 0
        assert np.isfinite(r[k]["graph.loss"]) and r[k]["graph.loss"] == r[k]["eager.loss"]
        for key in r[k]:
            if key.startswith("graph.s."):
                assert np.array_equal(r[k][key], r[k]["eager.s." + key[len("graph.s."):]]), (k, key)
    params = [key for key in r[0] if key.startswith("graph.s.") and "running_" not in key and "num_batches" not in key]
    assert len(params) > 100
    for key in params:
        assert np.array_equal(r[0][key], r[1][key]), key
    assert r[0]["graph.loss"] != r[1]["graph.loss"]                           # the ranks saw different batches
