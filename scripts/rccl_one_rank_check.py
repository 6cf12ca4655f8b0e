#!/usr/bin/env python
"""The multi-GPU training code on a ONE-GPU box: an `nccl` (= RCCL) process group with one rank, the full tracker on the
hand-written row kernels with its gradient all-reduce — the flat gradient buffer's single collective (the default on a HIP
device) and, once more, DistributedDataParallel — against the same model without any collective.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port P scripts/rccl_one_rank_check.py

With one participant the bucket all-reduce is an identity that still runs on the device through RCCL, and DDP divides the
gradient by world size 1: every gradient, and every parameter after clip + Adam, must equal the unwrapped run's BIT FOR BIT
(what tools/train_tracking.py:158-159 + ptt/utils/common_utils.py:275-289 set up in the reference). Prints one JSON line."""
import json
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptt_amd.config import StubDataset, ptt_model_cfg                     # noqa: E402
from ptt_amd.models import build_network                                 # noqa: E402
from ptt_amd.train_step import GRAD_ELEMS, DataParallelTrainer, synthetic_train_batch   # noqa: E402


def main():
    world, rank, local = int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    env_seen = {k: os.environ.get(k) for k in ("HSA_ENABLE_IPC_MODE_LEGACY", "NCCL_DEBUG", "HSA_FORCE_FINE_GRAIN_PCIE", "NCCL_P2P_DISABLE")}
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    one = torch.ones(1, device=dev)
    dist.all_reduce(one)                                               # RCCL communicator creation + one collective

    def run(force_ddp, steps, reducer="flat", graph=None):
        torch.manual_seed(1)
        model = build_network(ptt_model_cfg(), 1, StubDataset(training=True)).to(dev).train()
        trainer = DataParallelTrainer(model, dev, force_ddp=force_ddp, reducer=reducer, graph=graph)
        batch = synthetic_train_batch(100, 8, dev)
        trainer.forward_backward(batch)
        grads = {k: p.grad.detach().clone() for k, p in trainer.tracker.named_parameters() if p.grad is not None}
        for _ in range(steps):
            loss = trainer.step(batch)
        params = {k: p.detach().clone() for k, p in trainer.tracker.named_parameters()}
        return trainer, grads, params, float(loss)

    t0, g0, p0, l0 = run(False, 2)
    t1, g1, p1, l1 = run(True, 2)
    with t1.no_sync():                                                 # the exposure measurement's branch of bench.py
        t1.step(synthetic_train_batch(100, 8, dev))
    # DistributedDataParallel around the same model: its gradients are the per-weight finished ones (another summation order
    # than the flat buffer's single finishing launch), so they are compared to the unwrapped run of that form
    t2, g2, p2, l2 = run(False, 2, "ddp")
    t3, g3, p3, l3 = run(True, 2, "ddp")
    with t3.no_sync():
        t3.step(synthetic_train_batch(100, 8, dev))
    # the CAPTURED step (two hipGraphs around the eager all-reduce) on the one-rank group against the eager, unwrapped trainer:
    # seven steps = three eager ones, the capture, four replays
    t4, _, p4, l4 = run(False, 7, graph=False)
    t5, _, p5, l5 = run(True, 7, graph=True)
    gmax = max(float(v.abs().max()) for v in g0.values())
    flat_vs_ddp = max(float((g0[k] - g2[k]).abs().max()) / max(float(g2[k].abs().max()), 1e-3 * gmax) for k in g0)
    out = {"world": world, "ranks_seen": int(one.item()), "flat": t1.sink is not None and bool(t1.collective) and not t1.ddp,
           "ddp": bool(t3.ddp) and type(t3.model).__name__ == "DistributedDataParallel",
           "unwrapped_is_plain": not t0.ddp and not t0.collective and not t2.ddp, "grad_keys_equal": sorted(g0) == sorted(g1) == sorted(g2) == sorted(g3),
           "grads_bit_equal": all(torch.equal(g0[k], g1[k]) for k in g0) and all(torch.equal(g2[k], g3[k]) for k in g2),
           "params_bit_equal": all(torch.equal(p0[k], p1[k]) for k in p0) and all(torch.equal(p2[k], p3[k]) for k in p2),
           "flat_vs_ddp_max_rel": flat_vs_ddp,
           "n_grads": len(g1), "grad_bytes_allreduced_per_step": t1.grad_bytes_allreduced(), "expected_grad_bytes": GRAD_ELEMS * 4,
           "graph_captured": t5.captured is not None and t5.captured.second is not None and t5.graph_steps == 4 and t4.captured is None,
           "graph_params_bit_equal": all(torch.equal(p4[k], p5[k]) for k in p4) and l4 == l5,
           "loss_equal": l0 == l1 and l2 == l3, "loss": l1, "env": env_seen, "backend": dist.get_backend()}
    torch.cuda.synchronize()
    dist.destroy_process_group()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
