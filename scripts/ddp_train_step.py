#!/usr/bin/env python
"""BASELINE.json configs[3] as a stand-alone launcher: data-parallel training steps of the full tracker, one process
per GPU, the 19.6 MB gradient all-reduced in one DDP bucket over RCCL/xGMI (ptt_amd/train_step.py).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port 29533 \
        scripts/ddp_train_step.py --steps 20

is what the reference's scripts/train_ddp.sh:9 does for tools/train_tracking.py (whose own DDP wrap is dead code,
tools/train_tracking.py:63). `python bench.py --workload train --gpus N` runs the same step under the bench contract.
`--backend gloo --device cpu` is for the CPU tests only (the product index ops need a HIP device)."""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptt_amd.config import StubDataset, ptt_model_cfg                     # noqa: E402
from ptt_amd.models import build_network                                 # noqa: E402
from ptt_amd.train_step import GRAD_ELEMS, DataParallelTrainer, synthetic_train_batch   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=48, help="frames per rank per step")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL on ROCm) or gloo")
    ap.add_argument("--sync_bn", action="store_true", help="tools/train_tracking.py --sync_bn: batch statistics over all ranks")
    a = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", 1))
    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if a.backend == "nccl":
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    else:
        dev = torch.device("cpu")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if world > 1:
        kw = {"device_id": dev} if dev.type == "cuda" else {}
        dist.init_process_group(a.backend, rank=rank, world_size=world, **kw)
    torch.manual_seed(1)                                   # tools/train_tracking.py:73-79
    model = build_network(ptt_model_cfg(), 1, StubDataset(training=True)).to(dev).train()
    trainer = DataParallelTrainer(model, dev, sync_bn=a.sync_bn)
    batch = synthetic_train_batch(100 + rank, a.batch, dev)

    def sync():
        if dev.type == "cuda":
            torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for _ in range(a.warmup):
        trainer.step(batch)
    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = trainer.step(batch)
    sync()
    dt = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    seen = trainer.ranks_seen()
    if rank == 0:
        print(json.dumps({"metric": "training frames/sec (fwd+bwd+Adam, DDP gradient all-reduce)",
                          "value": round(a.batch * world * a.steps / dt.item(), 1), "n_gpus": world,
                          "ranks_seen": seen, "backend": a.backend,
                          "ms_per_step": round(dt.item() / a.steps * 1e3, 3), "loss": float(loss),
                          "grad_bytes_allreduced_per_step": GRAD_ELEMS * 4 if world > 1 else 0}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
