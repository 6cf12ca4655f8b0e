"""Data-parallel training step of the full tracker (BASELINE.json configs[3]): one process per GPU, the batch
sharded across ranks, ONE gradient all-reduce per step (4 903 113 fp32 values = 19.6 MB) over RCCL/xGMI.

On a HIP device the gradients of a step live in ONE flat buffer (train_ops.GradSink: every `.grad` is a view of it, filled by
one finishing launch after the backward pass) and the all-reduce is one collective over that buffer — what
DistributedDataParallel does with its single 25-MB bucket (all 19.6 MB become ready with the LAST gradient of the backward
pass, so there is nothing for it to overlap), without the bucket copies and the per-parameter hooks. `reducer="ddp"` (and
every CPU run: the gloo tests) wraps the model in DistributedDataParallel instead.

What it stands in for in the reference: tools/train_tracking.py:158-159 (the DistributedDataParallel wrap — dead
code there because :63 forces dist_train=False, so the reference's `--launcher pytorch` runs N unsynchronised
replicas) and tools/train_utils/train_utils.py:47-51 (model_func -> loss.backward() -> clip_grad_norm_(10) ->
optimizer.step()), with the optimiser of tools/cfgs/kitti_models/ptt.yaml:129-133 (Adam, lr 1e-3, betas 0.5/0.999,
eps 1e-6; ptt/optimization/__init__.py:12-14).

The same class runs on `gloo` + CPU tensors in the tests (world_size 2) — there the HIP index ops are replaced by
the tests' oracle, because the product ops refuse CPU tensors.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import synth
from .optim import ClipAdam

GRAD_ELEMS = 4903113            # parameters of the shipped PTT model


def synthetic_train_batch(seed, B, device, NS=1024, NT=512, K_s=200, K_t=100):
    """A training batch shaped like KittiTrackingDataset.get_train_items' default-collated output
    (ptt/datasets/kitti/kitti_dataset_tracking.py:60-107): search (B,NS,3), template (B,NT,3), cls_label (B,NS),
    reg_label (B,4). K_s = 200 unique points: nuScenes-Car sparsity. Seeded numpy, so every box
    and backend sees the same numbers."""
    s, t = synth.frames(seed, B, NS, NT, K_s=K_s, K_t=K_t)
    rs = np.random.RandomState(seed + 7919)
    cls = (rs.random_sample((B, NS)) > 0.7).astype(np.float32)
    reg = (rs.standard_normal((B, 4)) * 0.3).astype(np.float32)
    to = lambda a: torch.from_numpy(a).to(device)
    return {'search_points': to(s), 'template_points': to(t), 'batch_size': B, 'cls_label': to(cls), 'reg_label': to(reg)}


class DataParallelTrainer(object):
    """model (+ DDP when a process group with more than one rank is initialised) + Adam + gradient clipping.

        trainer = DataParallelTrainer(build_network(...).to(dev).train(), dev)
        loss = trainer.step(batch)          # forward, backward (all-reduce inside), clip, Adam
    """

    def __init__(self, model, device, lr=1e-3, betas=(0.5, 0.999), eps=1e-6, clip=10.0, bucket_cap_mb=25, sync_bn=False, force_ddp=False,
                 reducer=None):
        """force_ddp: reduce the gradients over the process group whenever one is initialised, a ONE-rank group included (the
        all-reduce then runs on the device with one participant: how a one-GPU box exercises the multi-GPU code).
        reducer: "flat" (HIP devices, the default there) = train_ops.GradSink + one all-reduce over its buffer; "ddp" =
        DistributedDataParallel (the default, and the only choice, off the HIP device)."""
        self.device = torch.device(device)
        self.world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        collective = self.world > 1 or (force_ddp and dist.is_available() and dist.is_initialized())
        if reducer is None:
            reducer = "flat" if self.device.type == 'cuda' else "ddp"
        if reducer not in ("flat", "ddp") or (reducer == "flat" and self.device.type != 'cuda'):
            raise ValueError("reducer: 'flat' (HIP device) or 'ddp'")
        self.reducer = reducer
        self.collective = collective
        self.ddp = collective and reducer == "ddp"
        if sync_bn and self.world > 1:
            # tools/train_tracking.py:133-134 (--sync_bn). The SharedMLP stages keep running on the hand-written row
            # kernels: their statistics are exchanged as 2C + 1 float64 sums per layer (ptt_amd/train_ops.py)
            model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)
        self.tracker = model
        if self.ddp:
            ids = [self.device.index] if self.device.type == 'cuda' else None
            # bucket_cap_mb 25 > 19.6 MB: the whole gradient is one bucket, one all-reduce per step; for a message
            # this small RCCL's direct algorithms over the 7 xGMI links beat a ring
            self.model = torch.nn.parallel.DistributedDataParallel(model, device_ids=ids, bucket_cap_mb=bucket_cap_mb)
        else:
            self.model = model
        # torch.optim.Adam with clip_grad_norm_ folded into step(): two launches on a HIP device, the stock path elsewhere
        self.optimizer = ClipAdam(self.model.parameters(), lr=lr, betas=betas, eps=eps)
        self.clip = clip
        self.sink = None
        if reducer == "flat":
            from .train_ops import GradSink
            self.sink = GradSink(list(self.model.parameters()), self.device)

    def forward_backward(self, batch):
        """loss.mean() and its gradients (averaged over ranks); no optimiser step."""
        ret, _, _ = self.model(dict(batch))
        loss = ret['loss'] if ret['loss'].dim() == 0 else ret['loss'].mean()
        if self.sink is None:
            self.optimizer.zero_grad(set_to_none=True)
            loss.backward()
            return loss
        with self.sink.collecting():
            loss.backward()
        self.sink.flush()
        if self.collective:
            # the mean over ranks, as DistributedDataParallel forms it: one all-reduce of the whole gradient
            dist.all_reduce(self.sink.flat)
            if self.world > 1:
                self.sink.flat.mul_(1.0 / self.world)
        return loss

    def step(self, batch):
        loss = self.forward_backward(batch)
        self.optimizer.step(max_norm=self.clip if self.clip else None)
        self.tracker.update_global_step()
        return loss

    def no_sync(self):
        """Steps inside this scope keep their gradients local (DistributedDataParallel.no_sync(), or the flat reducer's
        all-reduce skipped): what bench.py times to see how much of the all-reduce a step does not hide."""
        if self.ddp:
            return self.model.no_sync()
        trainer = self

        class _Scope(object):
            def __enter__(self):
                self.was, trainer.collective = trainer.collective, False

            def __exit__(self, *exc):
                trainer.collective = self.was
        return _Scope()

    def grad_bytes_allreduced(self):
        """Bytes of gradient one step hands to the bucket all-reduce (0 without DDP): every parameter that requires a gradient."""
        if not self.collective:
            return 0
        return sum(p.numel() * p.element_size() for p in self.model.parameters() if p.requires_grad)

    def ranks_seen(self):
        """All-reduce of ones: how many ranks actually take part in the collective."""
        if not self.collective:
            return 1
        one = torch.ones(1, device=self.device)
        dist.all_reduce(one)
        return int(one.item())
