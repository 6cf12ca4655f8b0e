"""world_size-2 gloo test of the N>1 path on CPU: partition, per-rank work, result merge, timing reduce.
The per-item work here is the ORACLE (tests may use it); the product's data path has no collective."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ptt_amd import tracklet_shard as ts


def test_partition_matches_reference_sampler_semantics():
    assert ts.shard_indices(10, 0, 4) == [0, 4, 8]
    assert ts.shard_indices(10, 1, 4) == [1, 5, 9]
    assert ts.shard_indices(10, 2, 4) == [2, 6, 0]          # padded by wrap-around
    assert ts.shard_indices(10, 3, 4) == [3, 7, 1]
    parts = [ts.shard_indices(10, r, 4) for r in range(4)]
    assert ts.merge_results(parts, 10) == list(range(10))
    assert ts.shard_indices(0, 0, 2) == [] and ts.shard_indices(3, 0, 1) == [0, 1, 2]
    assert ts.shard_indices(1, 1, 2) == [0]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import index_ops as O
    from ptt_amd import synth
    n_tracklets = 5
    clouds, _ = synth.frames(3, n_tracklets, 256, 64)
    mine = ts.shard_indices(n_tracklets)
    local = [O.fps(clouds[i:i + 1], 32)[0].tolist() for i in mine]      # independent per-item work, no collective
    merged = ts.gather_results(local, n_tracklets)
    t = ts.max_over_ranks(1.0 + rank)
    if rank == 0:
        q.put((merged, t))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_shard_and_merge():
    from oracle import index_ops as O
    from ptt_amd import synth
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    merged, t = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    clouds, _ = synth.frames(3, 5, 256, 64)
    expect = [O.fps(clouds[i:i + 1], 32)[0].tolist() for i in range(5)]
    assert merged == expect
    assert t == 2.0
