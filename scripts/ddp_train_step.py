#!/usr/bin/env python
"""BASELINE.json configs[3]: data-parallel training step of the full tracker, one process per GPU, gradient
all-reduce of the 4 903 113 fp32 parameters (19.6 MB, one DDP bucket) over RCCL/xGMI.

The reference's own DDP wrap is dead code (tools/train_tracking.py:63 forces dist_train=False), so this is the
launcher the build provides:  python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 \
    scripts/ddp_train_step.py --steps 20
Train mode runs the reference op sequence on the HIP ops (FPS, ball query, group/gather with autograd) and stock
torch conv/BN/linear layers; optimiser/clipping as tools/train_utils (Adam lr 1e-3 betas .5/.999 eps 1e-6, clip 10)."""
import argparse, os, sys, time, json
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptt_amd import synth
from ptt_amd.config import StubDataset, ptt_model_cfg
from ptt_amd.models import build_network


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=48)
    a = ap.parse_args()
    world, rank, lr = int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(lr)
    dev = torch.device("cuda", lr)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    torch.manual_seed(1)
    model = build_network(ptt_model_cfg(), 1, StubDataset(training=True)).to(dev).train()
    ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[lr], bucket_cap_mb=25)
    opt = torch.optim.Adam(ddp.parameters(), lr=1e-3, betas=(0.5, 0.999), eps=1e-6)
    B = a.batch
    s, t = synth.frames(100 + rank, B, 1024, 512, K_s=200, K_t=100)          # nuScenes-Car sparsity
    batch = lambda: {'search_points': torch.from_numpy(s).to(dev), 'template_points': torch.from_numpy(t).to(dev),
                     'batch_size': B, 'cls_label': (torch.rand(B, 1024, device=dev) > 0.7).float(),
                     'reg_label': torch.randn(B, 4, device=dev) * 0.3}

    def step():
        ret, _, _ = ddp(batch())
        loss = ret['loss'].mean()
        opt.zero_grad(set_to_none=True)
        loss.backward()                                   # DDP overlaps the bucketed all-reduce with backward
        torch.nn.utils.clip_grad_norm_(ddp.parameters(), 10)
        opt.step()
        return loss

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize(); dist.barrier(); t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize(); dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"metric": "training frames/sec (fwd+bwd+Adam, DDP over RCCL)", "value": round(B * world * a.steps / dt.item(), 1),
                          "n_gpus": world, "ms_per_step": round(dt.item() / a.steps * 1e3, 3), "loss": float(loss),
                          "grad_bytes_allreduced_per_step": 4903113 * 4}))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
