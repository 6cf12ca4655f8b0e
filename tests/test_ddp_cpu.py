"""BASELINE.json configs[3] on CPU: the data-parallel training step of the full tracker with `gloo`, world_size 2.

What is checked: after DataParallelTrainer.forward_backward on two ranks with DIFFERENT batches, every parameter
gradient on every rank equals the MEAN of the gradients two single-process runs produce on those two batches (the
all-reduce really averages; BatchNorm uses per-rank batch statistics, as the reference would without --sync_bn), and a
full step (clip + Adam) leaves both ranks with bit-identical parameters.

The product index ops refuse CPU tensors, so — in this test only — they are replaced by the oracle (tests may use it).
"""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_index_ops(put=setattr):
    """Swap the HIP index ops for the CPU oracle (test infrastructure; CPU tensors only). `put`: setattr in the
    spawned workers, monkeypatch.setattr in the pytest process (undone after the test)."""
    import ptt_amd.ops as ops
    from oracle import index_ops as O
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    c = lambda x: x.detach().contiguous().numpy()
    put(ops, "furthest_point_sampling", lambda xyz, n: t(O.fps(c(xyz), n)))
    put(ops, "gather_points", lambda f, i: t(O.gather(c(f), i.numpy())))
    put(ops, "gather_points_grad", lambda g_, i, n: t(O.gather_grad(c(g_), i.numpy(), n)))
    put(ops, "ball_query", lambda new_xyz, xyz, r, ns: t(O.ball_query(c(new_xyz), c(xyz), r, ns)))
    put(ops, "group_points", lambda f, i: t(O.group(c(f), i.numpy())))
    put(ops, "group_points_grad", lambda g_, i, n: t(O.group_grad(c(g_), i.numpy(), n)))


def _build(seed):
    from ptt_amd.config import StubDataset, ptt_model_cfg
    from ptt_amd.models import build_network
    from tests.util import fill_state_dict_
    return fill_state_dict_(build_network(ptt_model_cfg(), 1, StubDataset(training=True)), seed).train()


B = 2
SEED = 41


def _single_process_grads(batch_seed):
    from ptt_amd.train_step import DataParallelTrainer, synthetic_train_batch
    trainer = DataParallelTrainer(_build(SEED), "cpu")
    trainer.forward_backward(synthetic_train_batch(batch_seed, B, "cpu"))
    return {k: p.grad.numpy().copy() for k, p in trainer.tracker.named_parameters() if p.grad is not None}


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    _oracle_index_ops()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ptt_amd.train_step import DataParallelTrainer, synthetic_train_batch
    trainer = DataParallelTrainer(_build(SEED), "cpu")
    assert trainer.world == world and trainer.ranks_seen() == world
    batch = synthetic_train_batch(500 + rank, B, "cpu")
    loss = trainer.forward_backward(batch)
    # numpy copies: a torch tensor in a queue is passed by shared-memory handle, which dies with this process
    grads = {k: p.grad.numpy().copy() for k, p in trainer.tracker.named_parameters() if p.grad is not None}
    trainer.step(batch)                                     # clip + Adam on the averaged gradient
    params = {k: p.detach().numpy().copy() for k, p in trainer.tracker.named_parameters()}
    q.put((rank, float(loss.detach()), grads, params))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_ddp_gradients_are_the_mean_of_the_ranks(monkeypatch):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        rank, loss, grads, params = q.get(timeout=600)
        got[rank] = (loss, grads, params)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0

    _oracle_index_ops(monkeypatch.setattr)
    g0, g1 = _single_process_grads(500), _single_process_grads(501)
    assert set(g0) == set(got[0][1]) == set(got[1][1]) and len(g0) > 100
    worst = 0.0
    for k in g0:
        mean = (g0[k] + g1[k]) / 2
        scale = float(np.abs(mean).max()) + 1e-12
        for r in (0, 1):
            err = float(np.abs(got[r][1][k] - mean).max()) / scale
            worst = max(worst, err)
            assert err < 1e-5, (k, r, err)
        assert np.array_equal(got[0][1][k], got[1][1][k]), k       # both ranks hold the same reduced gradient
    # the two batches differ, so the mean is not either rank's own gradient
    k = 'backbone_3d.SA_modules.1.mlp_module.layer0.conv.weight'
    assert float(np.abs(g0[k] - g1[k]).max()) > 0
    # after clip + Adam both replicas hold identical parameters
    for k in got[0][2]:
        assert np.array_equal(got[0][2][k], got[1][2][k]), k
    assert np.isfinite(got[0][0]) and np.isfinite(got[1][0])


def _sync_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ptt_amd.train_step import broadcast_module_state
    torch.manual_seed(1000 + rank)                          # ranks that did NOT seed alike
    m = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.BatchNorm1d(5), torch.nn.Linear(5, 3))
    m.train()
    m(torch.randn(16, 7))                                   # running statistics and the batch counter move, differently per rank
    for _ in range(rank):
        m(torch.randn(16, 7))
    before = {k: v.numpy().copy() for k, v in m.state_dict().items()}
    broadcast_module_state(m, buffers_only=True)
    mid = {k: v.numpy().copy() for k, v in m.state_dict().items()}
    broadcast_module_state(m)
    after = {k: v.numpy().copy() for k, v in m.state_dict().items()}
    q.put((rank, before, mid, after))
    dist.barrier()
    dist.destroy_process_group()


def test_replicas_built_under_different_seeds_take_rank_zeros_parameters_and_buffers():
    """What the flat reducer does at construction in place of DistributedDataParallel's initial broadcast (train_step.
    DataParallelTrainer.sync_replicas): two gloo ranks seed differently; afterwards every parameter and buffer (the int64 batch
    counter included) is rank 0's."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sync_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        rank, before, mid, after = q.get(timeout=300)
        got[rank] = (before, mid, after)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    b0, b1 = got[0][0], got[1][0]
    assert not np.array_equal(b0["0.weight"], b1["0.weight"]) and b0["1.num_batches_tracked"] != b1["1.num_batches_tracked"]
    # buffers_only: the statistics are rank 0's, the parameters still each rank's own
    assert np.array_equal(got[1][1]["1.running_mean"], b0["1.running_mean"]) and got[1][1]["1.num_batches_tracked"] == b0["1.num_batches_tracked"]
    assert np.array_equal(got[1][1]["0.weight"], b1["0.weight"])
    for k in b0:                                            # everything: rank 0 unchanged, rank 1 == rank 0
        assert np.array_equal(got[0][2][k], b0[k]) and np.array_equal(got[1][2][k], b0[k]), k
