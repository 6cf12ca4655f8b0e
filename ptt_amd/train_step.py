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

On a HIP device with the flat reducer the step is CAPTURED after `graph_warmup` eager steps and replayed as hipGraphs from then
on (`graph=None`, the default; `graph=False` keeps every step eager): forward + backward + the gradient-finishing launch as one
graph, clip + Adam as a second, the all-reduce issued eagerly between the two (no collective is ever captured: the first run
with more than one RCCL rank is the driver's, and a captured collective is the one thing that could not be tried here). Without
a collective the two are ONE graph. Queuing a step then costs the host two replays instead of ~415 launches through Python and
autograd (12.4 ms -> well under a millisecond), so that eight ranks no longer need eight busy cores and small per-GPU batches
become device-bound. The replayed step is bit-identical to the eager one (tests/test_train_graph_gpu.py).

The same class runs on `gloo` + CPU tensors in the tests (world_size 2) — there the HIP index ops are replaced by
the tests' oracle, because the product ops refuse CPU tensors.
"""
import os
import warnings

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


def broadcast_module_state(module, buffers_only=False, src=0):
    """Parameters (unless buffers_only) and buffers of `module` on every rank := rank `src`'s, one broadcast per dtype over a
    flattened copy. No-op without a process group of more than one rank."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    with torch.no_grad():
        tensors = ([] if buffers_only else [p.data for p in module.parameters()]) + [b.data for b in module.buffers()]
        by_type = {}
        for t in tensors:
            by_type.setdefault(t.dtype, []).append(t)
        for ts in by_type.values():
            flat = torch.cat([t.reshape(-1) for t in ts])
            dist.broadcast(flat, src=src)
            off = 0
            for t in ts:
                t.copy_(flat[off:off + t.numel()].view_as(t))
                off += t.numel()
    if not buffers_only:
        from . import train_ops
        train_ops.invalidate_packed()                     # written through .data: the version counters did not move


class _CapturedStep(object):
    """The hipGraphs of one training step for one batch shape, with the static tensors they read and write."""

    def __init__(self, batch):
        self.static = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
        self.signature = self._signature(batch)
        self.first, self.second = torch.cuda.CUDAGraph(), None
        self.loss, self.tables, self.optimizer_state, self.packs = None, None, None, None
        self.loaded = {}                  # key -> (data_ptr, version) of the caller's tensor the static copy holds

    @staticmethod
    def _signature(batch):
        return tuple(sorted((k, tuple(v.shape), str(v.dtype), str(v.device)) if torch.is_tensor(v) else (k, v) for k, v in batch.items()))

    def accepts(self, batch):
        return self._signature(batch) == self.signature

    def valid(self, trainer):
        return trainer.optimizer._graph is not None and trainer.optimizer._graph is self.optimizer_state

    def load(self, batch):
        """The batch into the static tensors the graphs read; a tensor that is the very one loaded last time, unmodified since
        (same storage, same version counter), is not copied again."""
        for k, v in batch.items():
            if torch.is_tensor(v):
                tag = (v.data_ptr(), v._version)
                if self.loaded.get(k) != tag:
                    self.static[k].copy_(v, non_blocking=True)
                    self.loaded[k] = tag


class DataParallelTrainer(object):
    """model (+ DDP when a process group with more than one rank is initialised) + Adam + gradient clipping.

        trainer = DataParallelTrainer(build_network(...).to(dev).train(), dev)
        loss = trainer.step(batch)          # forward, backward (all-reduce inside), clip, Adam
    """

    def __init__(self, model, device, lr=1e-3, betas=(0.5, 0.999), eps=1e-6, clip=10.0, bucket_cap_mb=25, sync_bn=False, force_ddp=False,
                 reducer=None, graph=None, graph_warmup=3):
        """graph: None = replay the step as hipGraphs where that is possible (HIP device, flat reducer, no SyncBatchNorm exchange;
        anything else steps eagerly), True = the same but raise where it is not possible, False = always eager. graph_warmup:
        eager steps before the capture (the plans, tables and workspaces of a step are built by them).
        force_ddp: reduce the gradients over the process group whenever one is initialised, a ONE-rank group included (the
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
        self.collective = collective          # no_sync() switches this off for a scope
        self._collective_cfg = collective     # what the trainer was built with (what a capture records)
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
            if self.world > 1:
                self.sync_replicas()
        can_graph = reducer == "flat" and not (sync_bn and self.world > 1)
        if graph and not can_graph:
            raise ValueError("graph=True needs the flat reducer on a HIP device and no SyncBatchNorm exchange inside the step")
        self.graph_mode = can_graph and graph is not False and os.environ.get("PTT_TRAIN_GRAPH", "1") != "0"
        self.graph_warmup = max(1, int(graph_warmup))
        self.eager_steps = 0
        self.captured = None              # _CapturedStep once the step has been captured
        self.graph_steps = 0
        self._capture_misses = 0

    def sync_replicas(self, buffers_only=False):
        """Every parameter and buffer (BatchNorm running statistics, num_batches_tracked) takes rank 0's value — what
        DistributedDataParallel does at construction, and for the buffers before every forward pass. The flat reducer calls it once
        at construction, so that ranks which built their models under different seeds still train ONE model; afterwards the
        parameters stay equal by construction (same averaged gradients, same update), while the running statistics are PER RANK
        (each rank's own shard of the batches, as in the reference's unsynchronised `--launcher pytorch` replicas): call
        sync_replicas(buffers_only=True) before evaluating or checkpointing from a rank other than 0 if that matters."""
        broadcast_module_state(self.tracker, buffers_only)

    def _local_forward_backward(self, batch):
        """This rank's loss and gradients, finished into the sink's flat buffer (flat reducer only)."""
        ret, _, _ = self.model(dict(batch))
        loss = ret['loss'] if ret['loss'].dim() == 0 else ret['loss'].mean()
        with self.sink.collecting():
            loss.backward()
        self.sink.flush()
        return loss

    def _reduce(self):
        # the mean over ranks, as DistributedDataParallel forms it: one all-reduce of the whole gradient
        dist.all_reduce(self.sink.flat)

    def forward_backward(self, batch):
        """loss.mean() and its gradients (averaged over ranks); no optimiser step."""
        if self.sink is None:
            ret, _, _ = self.model(dict(batch))
            loss = ret['loss'] if ret['loss'].dim() == 0 else ret['loss'].mean()
            self.optimizer.zero_grad(set_to_none=True)
            loss.backward()
            return loss
        loss = self._local_forward_backward(batch)
        if self.collective:
            self._reduce()
            if self.world > 1:
                self.sink.flat.mul_(1.0 / self.world)
        return loss

    def step(self, batch):
        """One training step; returns the loss (a 0-dim device tensor — in graph mode the SAME tensor every step, holding the
        latest step's value once the device gets there)."""
        if self.graph_mode:
            if self.captured is not None and not self.captured.valid(self):
                self.captured = None                          # e.g. optimizer.load_state_dict(): new moment tensors
            if self.captured is None and self.eager_steps >= self.graph_warmup:
                self._capture(batch)
            if self.captured is not None and self.captured.accepts(batch):
                return self._replay(batch)
        loss = self.forward_backward(batch)
        self.optimizer.step(max_norm=self.clip if self.clip else None)
        self.tracker.update_global_step()
        self.eager_steps += 1
        return loss

    # ------------------------------------------------------------------------------------------------ the captured step
    def _capture(self, batch):
        """Records the step for batches shaped like `batch` (other shapes keep stepping eagerly). Nothing is executed here: the
        caller replays. A capture that fails turns graph mode off (with a warning) and the trainer continues eagerly."""
        from . import ops, train_ops
        if not self.optimizer.prepare_graph_step():
            # e.g. right after optimizer.load_state_dict(): the next eager step rebuilds the optimizer's table, then this succeeds
            self._capture_misses += 1
            if self._capture_misses > 3:
                self.graph_mode = False
                warnings.warn("DataParallelTrainer: the optimizer is not in the one-table state a captured step needs; stepping eagerly")
            return
        self._capture_misses = 0
        clip = self.clip if self.clip else None
        cap = _CapturedStep(batch)
        try:
            # what the recorded launches read besides tensors of the step itself is built NOW, eagerly, and belongs to the capture:
            # the finishing launch's job table and the weight-packing launch's table (nothing the eager code touches afterwards)
            self.sink.plan.prepare_capture()
            cap.packs = train_ops.PackCapture(list(self.model.parameters()), self.device)
            torch.cuda.synchronize(self.device)
            # capture_error_mode "thread_local": calls other threads make while this one records (RCCL's watchdog polling its events,
            # a data loader pinning memory) neither fail nor invalidate the capture; the autograd thread's launches are recorded all the
            # same — capturing is a property of the stream
            with cap.packs, torch.cuda.graph(cap.first, capture_error_mode="thread_local"):
                cap.loss = self._local_forward_backward(cap.static)
                if not self._collective_cfg:
                    self.optimizer.record_graph_step(clip)
            if self._collective_cfg:
                cap.second = torch.cuda.CUDAGraph()
                with cap.packs, torch.cuda.graph(cap.second, pool=cap.first.pool(), capture_error_mode="thread_local"):
                    if self.world > 1:
                        self.sink.flat.mul_(1.0 / self.world)
                    self.optimizer.record_graph_step(clip)
            cap.tables = ops.finish_capture_uploads()
        except Exception as e:                                # noqa: BLE001 — whatever the runtime refuses, the eager step still works
            ops.finish_capture_uploads()
            self.graph_mode = False
            warnings.warn("DataParallelTrainer: capturing the training step failed (%s: %s); stepping eagerly" % (type(e).__name__, e))
            return
        cap.optimizer_state = self.optimizer._graph
        self.captured = cap

    def _replay(self, batch):
        cap = self.captured
        cap.load(batch)
        self.optimizer.begin_graph_step(self.clip if self.clip else None)
        cap.first.replay()
        if cap.second is not None:
            if self.collective:                               # False inside no_sync(): a timing scope — the 1 / world scaling stays
                self._reduce()
            cap.second.replay()
        self.optimizer.end_graph_step()
        self.tracker.update_global_step()
        self.graph_steps += 1
        return cap.loss

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
