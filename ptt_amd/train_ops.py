"""N3 — the TRAINING step of the tracker on the hand-written kernels of ptt_amd/csrc/gemm_ops.hip and train_ops.hip
(+ the linear kernels of mfma_ops.hip for odd shapes). What runs through here, as torch.autograd.Functions over
(rows, channels) activations ("point-major": one row per (centre, neighbour) / (point, neighbour) / point):

  * every SharedMLP + max-pool stage — the reference's SharedMLP in train mode followed by the max over the neighbour axis
    (pytorch_utils.py:12-36,94-114 + pointnet2_modules.py:84-88; similarity_modules/p2b_xcoor.py:39-41) — as ONE function
    (_SharedMlpPool): per layer the 1x1 convolution on the persistent row GEMM with the BatchNorm batch statistics out of its
    epilogue; the activation relu(z a + b) is DEFERRED (never written: the next convolution, the weight gradient, the pool and
    the backward apply it while they load z); backward: the last layer's BatchNorm / ReLU / max-pool backward from the pooled
    gradient, the other layers' backward sums out of the input-gradient GEMM's epilogue, weight gradients on the 256 x 256-block
    kernel; running statistics updated as nn.BatchNorm does (momentum, unbiased variance, batch counter);
  * the layer-0 hoist of the SA levels and of CosineSimAug (sa_level_hoisted, xcorr_hoisted);
  * every nn.Linear of the Point-Transformer block (_RowsLinear, _RowsMlp2) and its element-wise passes (_PairInput,
    _AttnAggregate, _KnnRel), the Conv1d stacks of the heads (conv1d_stack_rows), cov_final.

    y = shared_mlp_pool(grouped, mlp, pool_dim)      # grouped (B,C,M,ns) as QueryAndGroup / the fusion tensor yields it

replaces `mlp(grouped).max(dim=pool_dim)[0]` (= F.max_pool2d over that axis) whenever the module is in training mode on
a HIP device and has the plain [conv1x1 (no bias) -> BatchNorm2d -> ReLU] units every shipped config builds (`usable`).
Every reduction runs in a fixed order: a step is bit-reproducible run to run (tests/test_train_config3_gpu.py).
"""
import os
import weakref

import torch
import torch.distributed as dist
import torch.nn as nn

from . import ops


def _raw_pointer_ready(conv, bn, device):
    """The row kernels read the convolution weight and the BatchNorm's parameters / buffers through raw pointers (and update the
    running statistics in place): contiguous float32 on the activations' device, an int64 batch counter. Anything else (a
    module cast to float64, a channels-last experiment, parameters left on another device) takes the stock torch path."""
    ts = [conv.weight] + ([conv.bias] if conv.bias is not None else [])
    if bn is not None:
        ts += [bn.weight, bn.bias, bn.running_mean, bn.running_var]
        nbt = bn.num_batches_tracked
        if nbt is not None and (nbt.dtype != torch.int64 or nbt.device != device):
            return False
    return all(t is not None and t.dtype == torch.float32 and t.device == device and t.is_contiguous() for t in ts)


def usable(mlp, x):
    """Plain SharedMLP units, float32 on a HIP device, training mode."""
    if not (mlp.training and x.is_cuda and x.dtype == torch.float32 and len(mlp) > 0):
        return False
    for unit in mlp:
        conv = getattr(unit, 'conv', None)
        bn = getattr(getattr(unit, 'normlayer', None), 'bn', None)
        if not isinstance(conv, nn.Conv2d) or conv.kernel_size != (1, 1) or conv.bias is not None or conv.weight.shape[0] % 4:
            return False
        if conv.stride != (1, 1) or conv.padding != (0, 0) or conv.dilation != (1, 1) or conv.groups != 1:
            return False
        if not isinstance(bn, (nn.BatchNorm2d, nn.SyncBatchNorm)) or not bn.affine or not bn.track_running_stats or bn.momentum is None:
            return False
        # a BatchNorm frozen inside a training model (bn.eval(): fine-tuning) normalises with its RUNNING statistics and
        # must not have them updated — the row kernels use batch statistics, so such a stack takes the stock path
        if not bn.training:
            return False
        if not isinstance(getattr(unit, 'activation', None), nn.ReLU):
            return False
        if list(unit._modules.keys()) != ['conv', 'normlayer', 'activation']:
            return False
        if not _raw_pointer_ready(conv, bn, x.device):
            return False
    return True


_pack_cache = {}       # id(parameter tensor) -> (weak reference to it, {(offset, shape, stride, version, transposed): packed})


def _pack_slot(base):
    e = _pack_cache.get(id(base))
    if e is not None and e[0]() is base:
        return e[1]
    slot = {}
    try:
        ref = weakref.ref(base, lambda _r, k=id(base): _pack_cache.pop(k, None))     # the entry dies with the parameter
    except TypeError:
        return slot
    _pack_cache[id(base)] = (ref, slot)
    return slot


class _PackPlan:
    """Every (weight view, transposed?) a training step has asked `packed` for, per device: after an optimiser update ALL of
    them are stale at once, and the first miss re-packs the whole set with ONE launch (ptt_pack_weights_f32) into a fresh
    arena instead of ~90 five-microsecond launches spread over the step. The device job table is rebuilt only when the set
    or a parameter's address changes."""

    def __init__(self):
        self.entries = {}      # (id(base), byte offset, shape, stride, transposed) -> [weakref(base), elems]
        self.table = None      # (device job table, [entry keys in job order], [data_ptr per job], arena elems)

    def register(self, ekey, base, elems):
        if ekey not in self.entries:
            self.entries[ekey] = [weakref.ref(base), int(elems)]
            self.table = None

    def _build(self, device):
        jobs, keys, ptrs, off = [], [], [], 0
        for ekey, (ref, elems) in list(self.entries.items()):
            base = ref()
            if base is None or id(base) != ekey[0] or not base.is_cuda or base.device != device:
                del self.entries[ekey]
                continue
            _, boff, shape, stride, transposed = ekey
            so, sk = (stride[1], stride[0]) if transposed else (stride[0], stride[1])
            cout, k = (shape[1], shape[0]) if transposed else (shape[0], shape[1])
            jobs.append((base.data_ptr() + boff, off, so, sk, cout, k))
            keys.append(ekey)
            ptrs.append(base.data_ptr())
            off += elems
        self.table = (ops.pack_jobs_table(jobs, device), keys, ptrs, off) if jobs else None

    def repack(self, device):
        """Re-pack every registered weight at its current version; False when nothing is registered."""
        if self.table is not None:
            for ekey, ptr in zip(self.table[1], self.table[2]):
                base = self.entries[ekey][0]()
                if base is None or base.data_ptr() != ptr:
                    self.table = None
                    break
        if self.table is None:
            self._build(device)
            if self.table is None:
                return False
        table, keys, _, total = self.table
        arena = ops.pack_weights(table, len(keys), torch.empty((total,), dtype=torch.float32, device=device))
        off = 0
        for ekey in keys:
            ref, elems = self.entries[ekey]
            base = ref()
            slot = _pack_slot(base)
            version = (base._version, base.data_ptr())
            for k in [k for k in slot if k[3] != version]:   # retire the previous step's packs
                del slot[k]
            slot[(ekey[1], ekey[2], ekey[3], version, ekey[4])] = arena[off:off + elems]
            off += elems
        return True


FUSED_BN_BWD = os.environ.get("PTT_FUSED_BN_BWD", "1") != "0"     # a layer's BatchNorm + ReLU backward applied by its input-gradient GEMM
#                                                                  (dev A/B: 0 = the apply pass of round 4)
POOL_EPILOGUE = True   # the last layer's max-pool from the extrema its GEMM's epilogue takes (False: a pooling pass over z)
_pack_plans = {}       # device -> _PackPlan
PACK_PLAN = True       # False: every weight packed by its own launch (development comparisons)


_pack_capture = None    # the PackCapture of the training step being recorded into a hipGraph, else None


class PackCapture(object):
    """What `packed` answers while a training step is RECORDED into a hipGraph. The recorded step must depend on nothing the
    eager code can change afterwards, and must leave nothing in the shared caches that only a replay fills in: so the capture gets
    a job table of its own (built eagerly, before the stream starts capturing, from the views of `params` the device's plan has
    seen) and a cache of its own. The first request records ONE ptt_pack_weights_f32 launch over that table — every replay re-packs
    the weights of the moment — and later requests inside the capture are answered from it. Whoever owns the graph keeps this
    object (table and arena) alive with it.

        pc = PackCapture(model.parameters(), device)       # eager
        with pc, torch.cuda.graph(g): ...                  # packed() goes through pc"""

    def __init__(self, params, device):
        ids = set(id(p) for p in params)
        self.device = torch.device(device)
        self.plan = _PackPlan()
        shared = _pack_plans.get(self.device)
        for ekey, (ref, elems) in (list(shared.entries.items()) if shared is not None else []):
            if ekey[0] in ids and ref() is not None:
                self.plan.entries[ekey] = [ref, elems]
        self.plan._build(self.device)
        self.cache, self.arena = {}, None

    def __enter__(self):
        global _pack_capture
        if _pack_capture is not None:
            raise RuntimeError("PackCapture: another capture is active")
        _pack_capture = self
        return self

    def __exit__(self, *exc):
        global _pack_capture
        _pack_capture = None

    def get(self, W, transpose):
        base = W._base if W._base is not None else W
        ekey = (id(base), W.data_ptr() - base.data_ptr(), tuple(W.shape), tuple(W.stride()), bool(transpose))
        hit = self.cache.get(ekey)
        if hit is not None:
            return hit
        if self.arena is None and self.plan.table is not None:
            table, keys, _, total = self.plan.table
            self.arena = ops.pack_weights(table, len(keys), torch.empty((total,), dtype=torch.float32, device=self.device))
            off = 0
            for k in keys:
                elems = self.plan.entries[k][1]
                self.cache[k] = self.arena[off:off + elems]
                off += elems
            hit = self.cache.get(ekey)
            if hit is not None:
                return hit
        w = W.detach()                                           # a view the eager steps never asked for: its own launch
        cout, k = (w.shape[1], w.shape[0]) if transpose else w.shape
        so, sk = (w.stride(1), w.stride(0)) if transpose else (w.stride(0), w.stride(1))
        hit = self.cache[ekey] = ops.pack_weight_strided(w, cout, k, so, sk, 1, 0)[0]
        return hit


def packed(W, transpose=False):
    """ops.pack_weight(W) (or of W^T) cached per weight VERSION on the PARAMETER the view belongs to: within one training step
    a weight is packed for its forward GEMM and, transposed, for its input gradient, and the search / template branches share
    the backbone's weights; the optimiser's in-place update bumps the version. The cache lives and dies with the parameter
    object (weak references): a new model whose parameter lands on a freed one's address never sees its entries. W: a 2-D (out, in)
    parameter, or a view / column slice of one, passed as the caller holds it (not detached: the view's base is the key).
    A miss on a view seen before re-packs every registered weight of the device in one launch (_PackPlan).
    What the key cannot see: an update made through `.data` (p.data.copy_(ema), fastai-style master copies) changes neither the
    version nor the address — call invalidate_packed() after such an update."""
    if _pack_capture is not None:
        return _pack_capture.get(W, transpose)
    base = W._base if W._base is not None else W
    slot = _pack_slot(base)
    off = (W.data_ptr() - base.data_ptr())
    key = (off, tuple(W.shape), tuple(W.stride()), (W._version, base.data_ptr()), bool(transpose))
    hit = slot.get(key)
    if hit is not None:
        return hit
    ekey = (id(base), off, key[1], key[2], key[4])
    plan = _pack_plans.setdefault(W.device, _PackPlan()) if PACK_PLAN and W.dim() == 2 else None
    if plan is not None and ekey in plan.entries and plan.repack(W.device):
        hit = slot.get(key)
        if hit is not None:
            return hit
    for k in [k for k in slot if k[3] != key[3]]:           # retire the previous step's packs
        del slot[k]
    w = W.detach()
    cout, k = (w.shape[1], w.shape[0]) if transpose else w.shape
    so, sk = (w.stride(1), w.stride(0)) if transpose else (w.stride(0), w.stride(1))
    hit = slot[key] = ops.pack_weight_strided(w, cout, k, so, sk, 1, 0)[0]
    if plan is not None:
        plan.register(ekey, base, hit.numel())
    return hit


def invalidate_packed():
    """Forget every packed weight (and the re-pack plans): needed after parameter updates the version counter does not see,
    i.e. anything written through `.data` (EMA / master-copy schemes: `p.data.copy_(master)`)."""
    _pack_cache.clear()
    _pack_plans.clear()


def conv_rows(x, W2d, in_a=None, in_b=None, want_stats=False, transpose=False):
    """z = act_in(x) @ W2d^T over (rows, K) activations, act_in = relu(x * in_a + in_b) when given (the deferred BatchNorm +
    ReLU of the producing layer). On the persistent row GEMM (ptt_rows_gemm_f32) where the shape allows, else on the linear
    kernel. want_stats: also the float64 partial column sums of z from the GEMM's epilogue (None on the linear kernel: the
    caller then takes the statistics in a pass of their own). transpose: multiply by W2d instead of W2d^T (input gradients)."""
    cout, K = (W2d.shape[1], W2d.shape[0]) if transpose else W2d.shape
    wp = packed(W2d, transpose)
    if ops.rows_gemm_supported(x.shape[0], K, cout, x.stride(0), cout, x=x):
        if want_stats:
            return ops.rows_gemm(x, wp, cout, in_scale=in_a, in_shift=in_b, want_stats=True)
        return ops.rows_gemm(x, wp, cout, in_scale=in_a, in_shift=in_b), None
    z = ops.linear_act_in(x, in_a, in_b, wp, cout) if in_a is not None else ops.linear(x, wp, cout)
    return z, None


_row_counts = {}


def _row_count(rows, device):
    """The constant float64 (1,) device tensor `rows` (the row count a BatchNorm's statistics were taken over), cached."""
    key = (int(rows), str(device))
    if key not in _row_counts:
        _row_counts[key] = torch.full((1,), float(rows), dtype=torch.float64, device=device)
    return _row_counts[key]


class GradSink(object):
    """Where the parameter gradients of ONE backward pass go when a trainer owns the whole step (train_step.DataParallelTrainer):
    a flat float32 buffer holding every parameter's gradient (`.grad` of each parameter is a view of it, so the optimiser, a
    logger or a gradient all-reduce see ordinary gradients — the all-reduce is ONE collective over the buffer), filled by ONE
    launch after the backward pass (ops.GradFinishPlan) from what the functions of this module left behind: the row-chunk partial
    sums of the weight-gradient and bias-gradient kernels, and the small gradients (BatchNorm scale / shift, coordinate columns)
    other launches produced as a by-product. The functions then return None for those inputs: no finishing launch per weight
    gradient, no autograd add for the weights the search and template branches share, no concatenation of column-slice
    gradients (~120 launches of a 530-launch step). The sum per parameter runs in the order the contributions were issued — a
    step stays bit-reproducible. Without an active sink every function returns finished gradient tensors as before.
    A parameter that received no gradient in a step holds ZEROS afterwards (its view of the zeroed buffer), not None — what
    DistributedDataParallel leaves as well, and unlike zero_grad(set_to_none=True): an optimizer sees it (Adam's moments decay, a
    weight decay would apply). Every parameter of the shipped tracker receives a gradient every step. One backward pass at a time
    is collected per process (`active` is process-wide: the autograd engine's worker thread must see it).

        sink = GradSink(model.parameters(), device)        # sets p.grad = views of sink.flat
        with sink.collecting():                            # sink.flat was zeroed: gradients autograd forms itself add in place
            loss.backward()
        sink.flush()                                       # one launch; then clip / all-reduce / optimiser on the views
    """
    active = None

    def __init__(self, params, device):
        self.device = torch.device(device)
        self.params = [p for p in params if p.requires_grad]
        self.index, off = {}, 0
        for p in self.params:
            if not (p.is_cuda and p.device == self.device and p.dtype == torch.float32 and p.is_contiguous()):
                raise ValueError("GradSink: contiguous float32 parameters on %s expected" % self.device)
            if p.data_ptr() in self.index:
                raise ValueError("GradSink: two parameters share storage (tied weights): their gradients cannot be told apart by address")
            self.index[p.data_ptr()] = (off, p.numel())
            off += (p.numel() + 3) // 4 * 4                      # every parameter starts on a 16-byte boundary
        self.flat = torch.zeros((max(off, 4),), dtype=torch.float32, device=self.device)
        self.views = [self.flat[self.index[p.data_ptr()][0]:self.index[p.data_ptr()][0] + p.numel()].view_as(p) for p in self.params]
        self.plan = ops.GradFinishPlan(self.device)
        self.jobs, self.keep = [], []
        self.attach()

    def attach(self):
        """p.grad = this sink's views (again, after something replaced them)."""
        for p, v in zip(self.params, self.views):
            p.grad = v

    def collecting(self):
        sink = self

        class _Scope(object):
            def __enter__(self):
                if GradSink.active is not None:
                    raise RuntimeError("GradSink: another backward pass is being collected")
                if any(p.grad is not v for p, v in zip(sink.params, sink.views)):
                    sink.attach()
                sink.flat.zero_()
                sink.jobs, sink.keep = [], []
                GradSink.active = sink

            def __exit__(self, *exc):
                GradSink.active = None
                if exc[0] is not None:
                    sink.jobs, sink.keep = [], []
        return _Scope()

    def locate(self, w):
        """(first element, columns, row stride, elements) of the parameter, or the 2-D column slice of one, that `w` IS — None if
        `w` is not (a view of) one of this sink's parameters in a layout the finishing launch addresses."""
        if w is None:
            return None
        base = w._base if w._base is not None else w
        hit = self.index.get(base.data_ptr())
        if hit is None or hit[1] != base.numel():
            return None
        rel = (w.data_ptr() - base.data_ptr()) // 4
        n = w.numel()
        if n == 0 or rel < 0 or w.dtype != torch.float32:
            return None
        if w.is_contiguous():
            return (hit[0] + rel, n, n, n) if rel + n <= hit[1] else None
        if w.dim() == 2 and w.stride(1) == 1 and w.stride(0) >= w.shape[1] and rel + (w.shape[0] - 1) * w.stride(0) + w.shape[1] <= hit[1]:
            return (hit[0] + rel, w.shape[1], w.stride(0), n)
        return None

    def push(self, loc, partials, nchunks):
        """partials: a tensor holding [nchunks][loc's elements] float32 sums (kept alive until the flush)."""
        self.jobs.append((loc[0], loc[1], loc[2], loc[3], partials.data_ptr(), int(nchunks)))
        self.keep.append(partials)

    def flush(self):
        """The one finishing launch: afterwards every `.grad` view holds its parameter's gradient."""
        self.plan.run(self.jobs, self.flat)
        self.jobs, self.keep = [], []


def _sunk(w):
    sink = GradSink.active
    loc = sink.locate(w) if sink is not None else None
    return (sink, loc) if loc is not None else (None, None)


def weight_grad(w, dz, x, x_scale=None, x_shift=None):
    """dz^T x, the gradient of the (out, in) weight `w` (a parameter or a view of one, as the forward pass held it): the finished
    tensor — or None once an active GradSink has taken the kernel's partial sums."""
    sink, loc = _sunk(w)
    if sink is None:
        return ops.linear_wgrad(dz, x, x_scale=x_scale, x_shift=x_shift)
    ws, nch = ops.linear_wgrad_partials(dz, x, x_scale, x_shift)
    sink.push(loc, ws, nch)
    return None


def bias_grad(b, g2):
    """The column sums of g2, the gradient of the bias `b`: finished, or None (GradSink)."""
    sink, loc = _sunk(b)
    if sink is None:
        return ops.colsum(g2)
    ws, nch = ops.colsum_partials(g2)
    sink.push(loc, ws, nch)
    return None


def small_grad(p, g):
    """A gradient `g` some launch already finished, of the parameter (or column slice) `p`: returned, or handed to the GradSink."""
    if g is None:
        return None
    sink, loc = _sunk(p)
    if sink is None or g.numel() != loc[3] or g.dtype != torch.float32:
        return g
    g = g.contiguous()
    if g.data_ptr() % 16:
        g = g.clone()
    sink.push(loc, g, 1)
    return None


class _GatherRows(torch.autograd.Function):
    """rows (B,N,C), idx (B,E) -> (B,E,C); backward = the deterministic row scatter-add."""

    @staticmethod
    def forward(ctx, rows, idx):
        ctx.save_for_backward(idx)
        ctx.N = rows.shape[1]
        return ops.gather_rows(rows.contiguous(), idx)

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        return ops.scatter_rows_det(g.contiguous(), idx, ctx.N), None


def gather_rows(rows, idx):
    return _GatherRows.apply(rows, idx)


class _AddRelTerm(torch.autograd.Function):
    """z0 = gathered + rel @ Wx^T : the three relative-coordinate terms of a hoisted layer 0, on the linear kernel (K = 3,
    `gathered` as the residual) — stock BLAS picks badly shaped kernels for K = 3 and for the 3 x R x C0 weight
    gradient. rel (R,3), gathered (R,C0), Wx (C0,3)."""

    @staticmethod
    def forward(ctx, gathered, rel, wx):
        ctx.save_for_backward(rel, wx)
        ctx.wx = wx
        return ops.linear(rel.contiguous(), packed(wx), wx.shape[0], residual=gathered.contiguous())

    @staticmethod
    def backward(ctx, g):
        rel, wx = ctx.saved_tensors
        g = g.contiguous()
        d_rel = ops.linear(g, packed(ctx.wx, True), 3) if ctx.needs_input_grad[1] else None
        d_wx = weight_grad(ctx.wx, g, rel.contiguous()) if ctx.needs_input_grad[2] else None
        return g, d_rel, d_wx


class _SplitCols(torch.autograd.Function):
    """w (C, K) -> (w[:, :k], w[:, k:]) as views; backward = ONE concatenation instead of two zero-filled slice gradients and
    their sum (five launches per hoisted level). The views keep the parameter as their base: `packed` keys on it."""

    @staticmethod
    def forward(ctx, w, k):
        ctx.set_materialize_grads(False)
        ctx.w, ctx.k = w, int(k)
        return w[:, :k], w[:, k:]

    @staticmethod
    def backward(ctx, ga, gb):
        w, k = ctx.w, ctx.k
        # under a GradSink each half goes straight to its columns of the parameter's gradient (the halves' consumers may have
        # sent theirs already: None)
        ga, gb = small_grad(w[:, :k], ga), small_grad(w[:, k:], gb)
        if ga is None and gb is None:
            return None, None
        if ga is None:
            ga = gb.new_zeros((w.shape[0], k))
        if gb is None:
            gb = ga.new_zeros((w.shape[0], w.shape[1] - k))
        return torch.cat((ga, gb), dim=1), None


class _SharedMlpPool(torch.autograd.Function):
    """(rows (R,C0), ns, eps per layer, preact, [W, gamma, beta] per layer) -> (pooled (R/ns, C_L), [mean, var] per layer).
    preact: `rows` already IS layer 0's convolution output (the caller hoisted that layer: train_ops.sa_level_hoisted /
    xcorr_hoisted), so layer 0 is BatchNorm + ReLU only and its W entry is a placeholder that gets no gradient."""

    @staticmethod
    def forward(ctx, x, ns, eps, preact, sync, bns, z0_part, front, f0, f1, f2, f3, f4, *params):
        """Deferred activation: a layer's output relu(BatchNorm(z)) = relu(z * a + b) is never written — the next
        convolution, its weight gradient, the max-pool and the BatchNorm backward apply it while they load z.
        sync: per layer a torch.distributed process group (nn.SyncBatchNorm: the statistics are those of the rows of ALL
        ranks — one all-reduce of 2C + 1 float64 per layer and direction) or None. bns: per layer the BatchNorm module whose
        bookkeeping (running statistics, batch counter) the launch that forms the statistics does, with the activation
        constants a, b (ops.bn_stats / bn_finish_partials with `bn`), or None: the caller does it (SyncBatchNorm layers).
        front (with preact, x = None): the rows z0 are built HERE from f0..f4, so that the backward pass can END with one pass over the
        gradient of the activated z0 that applies layer 0's BatchNorm backward on the fly and feeds z0's consumers directly, instead
        of an apply pass that writes dz0 and further passes that read it:
          ('xcorr',): CosineSimAug's layer 0, z0 = f0[b,i] + f1[b,j,i] * f2 (P, cos, w_sim) — ops.xcorr_z0 / ops.xcorr_z0_bnbwd;
          ('sa', radius, normalize_xyz): a hoisted SA level's layer 0, z0 = f3[idx] + f4 ((f0[idx] - f1) / radius) (xyz, new_xyz, idx,
          per-point term | None, Wx) — ops.sa_z0_rows / ops.sa_z0_bnbwd."""
        ctx.set_materialize_grads(False)          # the statistics outputs carry no gradient: no zero tensors made for them
        L = len(params) // 3
        saved, stats, counts = [], [], []
        ctx.front = front
        front_saved = ()
        if front is not None and front[0] == 'xcorr':
            x, z0_part = ops.xcorr_z0(f0, f1, f2, want_stats=True)
            front_saved = (f0, f1, f2)
        elif front is not None:
            x, rel, z0_part = ops.sa_z0_rows(f0.contiguous(), f1.contiguous(), f2, f3.contiguous() if f3 is not None else None, f4.detach(),
                                             front[1], front[2], want_stats=True)
            ctx.sa_points, ctx.has_term, ctx.f4 = f0.shape[1], f3 is not None, f4
            front_saved = (f2, rel)
        cur, cur_a, cur_b = x.contiguous(), None, None
        for l in range(L):
            W, gamma, beta = params[3 * l], params[3 * l + 1], params[3 * l + 2]
            cout = gamma.shape[0]
            part, extrema = None, None
            if preact and l == 0:
                z, part = cur, z0_part                     # the caller's launch summed the statistics of its rows (or None)
            elif (l == L - 1 and ns > 1 and POOL_EPILOGUE and sync[l] is None and cur_a is not None
                  and ops.rows_gemm_pool_supported(cur.shape[0], cur.shape[1], cout, cur.stride(0), ns, x=cur)):
                # the last layer: its GEMM's epilogue also takes the per-group extrema of z, so that the max-pool needs no pass
                # over z once the statistics (summed by the same launch) are finished
                z, part, extrema = ops.rows_gemm_pool(cur, packed(W.reshape(cout, -1)), cout, cur_a, cur_b, ns)
            else:       # the statistics of z come out of the GEMM's epilogue where the persistent row GEMM runs
                z, part = conv_rows(cur, W.reshape(cout, -1), cur_a, cur_b, want_stats=True)
            if sync[l] is not None:
                sums = ops.bn_sums_partials(part, z.shape[0]) if part is not None else ops.bn_sums(z)
                dist.all_reduce(sums, group=sync[l])
                mean, var, invstd = ops.bn_finish(sums, eps[l])
                count = sums[-1:].clone()                      # global row count, float64, on the device
                a = (gamma.detach() * invstd).contiguous()
                b = (beta.detach() - mean * a).contiguous()
            else:
                if part is not None:
                    mean, var, invstd, a, b = ops.bn_finish_partials(part, z.shape[0], eps[l], bn=bns[l])
                else:
                    mean, var, invstd, a, b = ops.bn_stats(z, eps[l], bn=bns[l])
                count = _row_count(z.shape[0], z.device)
            saved += [cur, cur_a if cur_a is not None else mean.new_empty(0), cur_b if cur_b is not None else mean.new_empty(0),
                      z, mean, invstd, a, b, count]
            stats += [mean, var, count]
            cur, cur_a, cur_b = z, a, b
        pooled, arg = ops.pool_select(extrema, cur_a, cur_b) if extrema is not None else ops.pool_rows(cur, ns, cur_a, cur_b)
        ctx.save_for_backward(arg, *saved, *[p.detach() for p in params], *front_saved)
        ctx.weights = tuple(params[3 * l] for l in range(L))     # the parameter objects themselves: keys of the pack cache
        ctx.param_objs = tuple(params)                           # ... and of a GradSink
        ctx.L, ctx.ns, ctx.preact, ctx.sync = L, int(ns), bool(preact), tuple(sync)
        ctx.mark_non_differentiable(*stats)
        return (pooled,) + tuple(stats)

    @staticmethod
    def backward(ctx, dpooled, *unused):
        L, ns = ctx.L, ctx.ns
        t = ctx.saved_tensors
        arg, saved, params, front = t[0], t[1:1 + 9 * L], t[1 + 9 * L:1 + 12 * L], t[1 + 12 * L:]
        dpooled = dpooled.contiguous()
        g, part = None, None        # part: BatchNorm backward sums of THIS layer, taken by the GEMM that produced g
        grads = [None] * (3 * L)
        front_grads = [None] * 5
        kind = ctx.front[0] if ctx.front is not None else None
        for l in range(L - 1, -1, -1):
            x_in, in_a, in_b, z, mean, invstd, a, b, count = saved[9 * l:9 * l + 9]
            W, gamma = params[3 * l], params[3 * l + 1]
            last = l == L - 1
            # the last layer's gradient arrives POOLED: max-pool backward, BatchNorm sums and dz are formed from (dpooled,
            # arg) — the (R, C) gradient of the pooled layer is never written (ptt_bn_bwd_pooled_f32)
            fused = None
            if FUSED_BN_BWD and ctx.sync[l] is None and l > 0 and (last or part is not None):
                # the layer's BatchNorm + ReLU backward is applied by its input-gradient GEMM while that stages its rows (three
                # per-channel constants instead of a pass that reads g and z and writes dz); the GEMM writes dz out once for the
                # weight gradient, which therefore runs after it
                src = dpooled if last else g
                pooled = last and ns > 1
                w2 = ctx.weights[l].reshape(ctx.weights[l].shape[0], -1)
                if ops.rows_gemm_bnbwd_fused_supported(z.shape[0], w2.shape[0], w2.shape[1], ns if pooled else 0, src, z):
                    if last:
                        dgamma, dbeta, consts = ops.bn_bwd_pooled_consts(dpooled, arg, ns, z, mean, invstd, gamma, a, b)
                    else:
                        dgamma, dbeta, consts = ops.bn_bwd_consts(part, mean, invstd, gamma, z.shape[0])
                    zp, mp, ip, ap, bp = saved[9 * (l - 1) + 3], saved[9 * (l - 1) + 4], saved[9 * (l - 1) + 5], saved[9 * (l - 1) + 6], saved[9 * (l - 1) + 7]
                    fused = ops.rows_gemm_bnbwd_fused(src, arg if pooled else None, ns if pooled else 0, z, consts, mean, a, b,
                                                      packed(w2, True), w2.shape[1], zp, mp, ip, ap, bp)
            if fused is not None:
                g, part, dz = fused
                grads[3 * l + 1], grads[3 * l + 2] = dgamma, dbeta
                has_t = in_a.numel() > 0
                grads[3 * l] = weight_grad(w2, dz, x_in, in_a if has_t else None, in_b if has_t else None)
                continue
            if ctx.sync[l] is not None:
                # torch's SyncBatchNorm: dgamma / dbeta are the rank's LOCAL sums (DDP averages parameter gradients); dz
                # uses the sums of all ranks
                if last:
                    sums = ops.bn_bwd_pooled_sums(dpooled, arg, ns, z, mean, invstd, a, b)
                elif part is not None:
                    sums = ops.bn_bwd_sums_from_partials(part)
                else:
                    sums = ops.bn_bwd_sums(g, None, z, mean, invstd, act_scale=a, act_shift=b)
                local = sums.float()
                dbeta, dgamma = local[0].contiguous(), local[1].contiguous()
                dist.all_reduce(sums, group=ctx.sync[l])
                glob = sums.float()
                if last:
                    dz = ops.bn_bwd_pooled_apply(dpooled, arg, ns, z, mean, invstd, gamma, glob[0].contiguous(), glob[1].contiguous(),
                                                 count, a, b)
                else:
                    dz = ops.bn_bwd_apply(g, None, z, mean, invstd, gamma, glob[0].contiguous(), glob[1].contiguous(), count, out=g,
                                          act_scale=a, act_shift=b)
            elif last:
                dz, dgamma, dbeta = ops.bn_bwd_pooled(dpooled, arg, ns, z, mean, invstd, gamma, a, b)
            elif part is not None and l == 0 and ctx.preact and kind == 'xcorr' and mean.shape[0] <= 256 and g.is_contiguous():
                # CosineSimAug's layer 0: BatchNorm backward applied while its ONE consumer reads the gradient (no dz0 tensor)
                front_grads[0], front_grads[1], front_grads[2], dgamma, dbeta = ops.xcorr_z0_bnbwd(part, g, front[0], front[1], front[2], mean,
                                                                                                   invstd, gamma, a, b)
                grads[1], grads[2] = dgamma, dbeta
                g = None
                break
            elif (part is not None and l == 0 and ctx.preact and kind == 'sa' and mean.shape[0] <= 1024 and g.is_contiguous()
                  and z.is_contiguous()):
                # a hoisted SA level's layer 0: the same for the K = 3 weight gradient (d_wx); dz0 is written (over g) only for the
                # row scatter of a level with point features
                idx, rel = front
                want_term = ctx.has_term and ctx.needs_input_grad[11]
                sink, loc = _sunk(ctx.f4)
                dz, front_grads[4], dgamma, dbeta = ops.sa_z0_bnbwd(part, g, z, rel, mean, invstd, gamma, a, b, want_term, dwx_partials=sink is not None)
                if sink is not None:
                    sink.push(loc, *front_grads[4])
                    front_grads[4] = None
                if want_term:
                    B, M, ns_ = idx.shape
                    front_grads[3] = ops.scatter_rows_det(dz.view(B, M * ns_, -1), idx.view(B, M * ns_), ctx.sa_points)
                grads[1], grads[2] = dgamma, dbeta
                g = None
                break
            elif part is not None:
                dz, dgamma, dbeta = ops.bn_bwd_from_partials(part, g, z, mean, invstd, gamma, a, b, out=g)   # in place over g
            else:
                dz, dgamma, dbeta = ops.bn_bwd(g, None, z, mean, invstd, gamma, out=g, act_scale=a, act_shift=b)   # in place over g
            grads[3 * l + 1], grads[3 * l + 2] = dgamma, dbeta
            if ctx.preact and l == 0:
                g = dz                                                                  # d(loss)/d(layer-0 pre-activation)
                if kind == 'xcorr':
                    B, n1 = front[0].shape[0], front[0].shape[1]
                    front_grads[0], front_grads[1], front_grads[2] = ops.xcorr_z0_bwd(dz.contiguous(), front[1], front[2], B, front[1].shape[1], n1)
                    g = None
                elif kind == 'sa':
                    idx, rel = front
                    B, M, ns_ = idx.shape
                    dzc = dz.contiguous()
                    if ctx.has_term and ctx.needs_input_grad[11]:
                        front_grads[3] = ops.scatter_rows_det(dzc.view(B, M * ns_, -1), idx.view(B, M * ns_), ctx.sa_points)
                    front_grads[4] = weight_grad(ctx.f4, dzc, rel)
                    g = None
                break
            Wp = ctx.weights[l]
            w2 = Wp.reshape(Wp.shape[0], -1)
            has_t = in_a.numel() > 0
            grads[3 * l] = weight_grad(w2, dz, x_in, in_a if has_t else None, in_b if has_t else None)
            part = None
            if l > 0:
                # the gradient w.r.t. the activated input of layer l = the gradient BatchNorm l - 1 receives: its backward sums
                # come out of this GEMM's epilogue where the persistent row GEMM takes the shape
                zp, mp, ip, ap, bp = saved[9 * (l - 1) + 3], saved[9 * (l - 1) + 4], saved[9 * (l - 1) + 5], saved[9 * (l - 1) + 6], saved[9 * (l - 1) + 7]
                if ops.rows_gemm_supported(dz.shape[0], w2.shape[0], w2.shape[1], dz.stride(0), w2.shape[1], x=dz):
                    g, part = ops.rows_gemm_bnbwd(dz, packed(w2, True), w2.shape[1], zp, mp, ip, ap, bp)
                else:
                    g, _ = conv_rows(dz, w2, transpose=True)
            elif ctx.needs_input_grad[0]:
                g, _ = conv_rows(dz, w2, transpose=True)                                # w.r.t. the (not normalised) input rows
            else:
                g = None
        for l in range(L):
            if grads[3 * l] is not None:
                grads[3 * l] = grads[3 * l].view_as(params[3 * l])
            grads[3 * l + 1] = small_grad(ctx.param_objs[3 * l + 1], grads[3 * l + 1])
            grads[3 * l + 2] = small_grad(ctx.param_objs[3 * l + 2], grads[3 * l + 2])
        return (g, None, None, None, None, None, None, None) + tuple(front_grads) + tuple(grads)


def shared_mlp_pool(grouped, mlp, pool_dim):
    """mlp(grouped).max(dim=pool_dim)[0] for a (B,C,H,W) tensor in training mode; pool_dim is 2 or 3."""
    assert grouped.dim() == 4 and pool_dim in (2, 3)
    B, C, H, W = grouped.shape
    if pool_dim == 3:
        rows = grouped.permute(0, 2, 3, 1).reshape(B * H * W, C)             # (b, h, w) rows, max over w
        ns, keep = W, H
    else:
        rows = grouped.permute(0, 3, 2, 1).reshape(B * W * H, C)             # (b, w, h) rows, max over h
        ns, keep = H, W
    return rows_mlp_pool(rows, mlp, ns, B, keep, preact=False)


def _sync_group(bn):
    """The process group an nn.SyncBatchNorm synchronises over right now, else None (plain BatchNorm, evaluation, no
    initialised process group, or a single rank: torch's SyncBatchNorm falls back to local statistics in those cases)."""
    if not isinstance(bn, nn.SyncBatchNorm) or not (dist.is_available() and dist.is_initialized()):
        return None
    group = bn.process_group if bn.process_group is not None else dist.group.WORLD
    return group if dist.get_world_size(group) > 1 else None


def rows_mlp_pool(rows, mlp, ns, B, keep, preact, z0_part=None, front=None, as_rows=False):
    """The row form: rows (B * keep * ns, C) ordered (frame, kept position, pooled position) -> (B, C_L, keep).
    preact: rows are layer 0's convolution output already (hoisted by the caller); z0_part: its BatchNorm statistics as float64
    partial sums (chunks, 2, C), when the launch that built the rows summed them (ops.sa_z0_rows). front = (meta, tensors) with
    rows = None: layer 0's rows are built inside the stage's autograd function (_SharedMlpPool.forward, `front`)."""
    params, eps, sync = [], [], []
    for unit in mlp:
        bn = unit.normlayer.bn
        params += [unit.conv.weight, bn.weight, bn.bias]
        eps.append(float(bn.eps))
        sync.append(_sync_group(bn))
    bns = tuple(unit.normlayer.bn if g is None else None for unit, g in zip(mlp, sync))
    meta, ft = (front[0], tuple(front[1]) + (None,) * (5 - len(front[1]))) if front is not None else (None, (None,) * 5)
    out = _SharedMlpPool.apply(rows, ns, tuple(eps), bool(preact), tuple(sync), bns, z0_part, meta, *ft, *params)
    pooled, stats = out[0], out[1:]
    with torch.no_grad():                                   # nn.BatchNorm's bookkeeping in training mode: done by the statistics'
        for l, unit in enumerate(mlp):                      # own launch, except for SyncBatchNorm layers (all-reduced count)
            if bns[l] is None:
                ops.bn_update_running(unit.normlayer.bn, stats[3 * l], stats[3 * l + 1], stats[3 * l + 2])
    if as_rows:
        return pooled                                       # (B * keep, C_L) rows as the kernels hold them: no views for autograd to undo
    return pooled.view(B, keep, -1).transpose(1, 2)         # (B, C_L, keep)


def sa_level_hoisted(xyz, new_xyz, features, idx, mlp, radius, normalize_xyz):
    """One set-abstraction level in TRAINING mode with layer 0 hoisted, as the fused inference kernels do it (DESIGN.md
    lesson 9): the first 1x1 convolution is linear in [rel ; f_n], so its feature half is evaluated once per POINT
    (B*N rows) instead of once per (centre, neighbour) row (16-32x more), gathered per row (ptt_gather_rows_f32, with the
    deterministic row scatter-add as its backward), and the three relative-coordinate terms are added:
        z0[b,m,k,:] = (W0[:,3:] f)[b, idx[b,m,k]] + W0[:,:3] (xyz[b, idx[b,m,k]] - new_xyz[b,m]) / radius
    The grouped (B, 3+C, M, ns) tensor of QueryAndGroup (pointnet2_utils.py:350-361) and the K = 3+C convolution over all
    rows never exist. xyz (B,N,3), new_xyz (B,M,3), features (B,C,N), idx (B,M,ns) int32 -> (B, C_L, M)."""
    from .models.backbones_3d.pointnet2 import pointnet2_utils as pu
    B, M, ns = idx.shape
    w0 = mlp[0].conv.weight.reshape(mlp[0].conv.weight.shape[0], -1)                    # (C0, 3 + C): xyz first (:359-361)
    wx, wf = _SplitCols.apply(w0, 3) if features is not None else (w0, None)
    if not (xyz.requires_grad or new_xyz.requires_grad):
        # fixed coordinates (the backbone's levels): the whole front — relative coordinates, gather of the per-point terms,
        # the three coordinate channels — is one launch; features None: a level without point features (layer 0 = Wx . rel)
        term = _RowsLinear.apply(features.transpose(1, 2), wf, None, None) if features is not None else None
        # z0 = term[idx] + Wx . rel is built inside the stage's function (ops.sa_z0_rows): its backward ends with ops.sa_z0_bnbwd
        return rows_mlp_pool(None, mlp, ns, B, M, preact=True, front=(('sa', float(radius), bool(normalize_xyz)), (xyz, new_xyz, idx, term, wx)))
    rel = pu.grouping_operation(xyz.transpose(1, 2).contiguous(), idx) - new_xyz.transpose(1, 2).unsqueeze(-1)   # (B,3,M,ns)
    if normalize_xyz:
        rel = rel / radius
    rel_rows = rel.permute(0, 2, 3, 1).reshape(B * M * ns, 3)
    term = _RowsLinear.apply(features.transpose(1, 2), wf, None, None)                  # (B,N,C0), once per point
    z0 = _AddRelTerm.apply(gather_rows(term, idx.view(B, M * ns)).view(B * M * ns, -1), rel_rows, wx)
    return rows_mlp_pool(z0, mlp, ns, B, M, preact=True)


def xcorr_hoisted(search_feats, template_feats, template_xyz, mlp, eps):
    """CosineSimAug's fusion + SharedMLP + max over the template axis (p2b_xcoor.py:35-42) in TRAINING mode with layer 0
    split as the fused inference kernel splits it: only the similarity channel depends on the search point, so
        z0[b,j,i,:] = w_sim * cos(t_i, s_j) + (W0[:,1:] [xyz_i ; feat_i])        (j search, i template point)
    and neither the (B,260,n1,n2) fusion tensor nor the expanded operands of nn.CosineSimilarity are ever built: the
    cosine map is a (B,n1,n2) matrix product of the normalised features (x1.x2 / (max(|x1|,eps) max(|x2|,eps)), as
    torch.nn.functional.cosine_similarity defines it). -> (B, C_L, n2)."""
    B, C, n2 = search_feats.shape
    n1 = template_feats.shape[-1]
    w0 = mlp[0].conv.weight.reshape(mlp[0].conv.weight.shape[0], -1)                    # (C0, 1 + 3 + C)
    cos = _CosMap.apply(search_feats, template_feats, float(eps))                       # (B,n2,n1)
    rows_i = torch.cat((template_xyz, template_feats.transpose(1, 2)), dim=2)           # (B,n1,3+C)
    wsim, wrest = _SplitCols.apply(w0, 1)
    P = _RowsLinear.apply(rows_i, wrest, None, None)                                    # (B,n1,C0)
    # z0 (B*n2*n1, C0), rows ordered (b, j, i), is built inside the stage's function: its backward ends with one pass over the
    # gradient of the activated z0 (ops.xcorr_z0_bnbwd)
    return rows_mlp_pool(None, mlp, n1, B, n2, preact=True, front=(('xcorr',), (P.contiguous(), cos.contiguous(), wsim.reshape(-1).contiguous())))


class _CosMap(torch.autograd.Function):
    """cos[b,j,i] = s_j . t_i / (max(|s_j|, eps) max(|t_i|, eps)) for search features s (B,C,n2) and template features t (B,C,n1)
    in any layout: unit rows (one launch per side), one batched product; backward: two batched products and one launch per side
    (ops.cos_bwd_rows) — 3 + 4 launches for the ~40 of the element-wise formulation (norm, clamp, divide and their adjoints)."""

    @staticmethod
    def forward(ctx, s, t, eps):
        us, ns = ops.unit_rows_eps(s, eps)
        ut, nt = ops.unit_rows_eps(t, eps)
        cos = torch.bmm(us, ut.transpose(1, 2))
        ctx.save_for_backward(us, ns, ut, nt, cos)
        ctx.like = ((tuple(s.shape), tuple(s.stride())), (tuple(t.shape), tuple(t.stride())))    # the gradients' layouts
        return cos

    @staticmethod
    def backward(ctx, g):
        us, ns, ut, nt, cos = ctx.saved_tensors
        g = g.contiguous()
        ds = dt = None
        if ctx.needs_input_grad[0]:
            ds = ops.cos_bwd_rows(torch.bmm(g, ut), us, ns, g, cos, True, ctx.like[0])
        if ctx.needs_input_grad[1]:
            dt = ops.cos_bwd_rows(torch.bmm(g.transpose(1, 2), us), ut, nt, g, cos, False, ctx.like[1])
        return ds, dt, None


class _KnnRel(torch.autograd.Function):
    """rel[b,i,j,:] = xyz[b,i] - xyz[b, knn[b,i,j]] (variants.py:158) with the kNN kernel's own output as the value and a
    DETERMINISTIC backward: d xyz[b,i] = sum_j g[b,i,j] - (sum of g over the pairs whose neighbour is i, in a fixed order).
    The stock form (index_points + subtraction) scatters with atomics: run-to-run different low bits in every gradient
    upstream of the box head's proposal centres."""

    @staticmethod
    def forward(ctx, xyz, knn, rel):
        ctx.save_for_backward(knn)
        return rel.clone()

    @staticmethod
    def backward(ctx, g):
        (knn,) = ctx.saved_tensors
        B, N, k, _ = g.shape
        back = ops.scatter_add_det(g.permute(0, 3, 1, 2).reshape(B, 3, N * k).contiguous(), knn.view(B, N * k), N)   # (B,3,N)
        return g.sum(dim=2) - back.transpose(1, 2), None, None


class _PairInput(torch.autograd.Function):
    """t = q_i - kf[knn_ij] + pos_ij in one pass (variants.py:160's argument); backward: dq = sum_j dt, dpos = dt,
    dkf = -(deterministic row scatter-add of dt over knn)."""

    @staticmethod
    def forward(ctx, q, kf, knn, pos, order=None, start=None):
        # order / start: ops.scatter_csr(knn) when the caller shares it with _AttnAggregate (the two scatters of a block sort the
        # same indices)
        ctx.save_for_backward(knn, *((order, start) if order is not None else ()))
        return ops.pt_pair_input(q.contiguous(), kf.contiguous(), knn, pos.contiguous())

    @staticmethod
    def backward(ctx, dt):
        knn, csr = ctx.saved_tensors[0], (tuple(ctx.saved_tensors[1:]) or None)
        B, N, k, D = dt.shape
        dt = dt.contiguous()
        dq = dt.sum(dim=2) if ctx.needs_input_grad[0] else None
        dkf = ops.scatter_rows_det(dt.view(B, N * k, D), knn.view(B, N * k), N, csr).neg_() if ctx.needs_input_grad[1] else None
        return dq, dkf, None, dt, None, None


class _AttnAggregate(torch.autograd.Function):
    """attn = softmax_j(a / sqrt(d)); res = sum_j attn * (vf[knn] + pos) (variants.py:161-163) in one pass each way.
    Returns (res, attn); attn is returned for the caller's benefit only (both heads drop it) and carries no gradient."""

    @staticmethod
    def forward(ctx, a, vf, knn, pos, scale, order=None, start=None):
        ctx.set_materialize_grads(False)
        a, vf, pos = a.contiguous(), vf.contiguous(), pos.contiguous()
        attn, res = ops.pt_attn_train_fwd(a, vf, knn, pos, scale)
        ctx.save_for_backward(attn, vf, knn, pos, *((order, start) if order is not None else ()))
        ctx.scale = float(scale)
        ctx.mark_non_differentiable(attn)
        return res, attn

    @staticmethod
    def backward(ctx, dres, _dattn):
        attn, vf, knn, pos = ctx.saved_tensors[:4]
        csr = tuple(ctx.saved_tensors[4:]) or None
        B, N, k, D = attn.shape
        da, dvp = ops.pt_attn_train_bwd(attn, vf, knn, pos, dres.contiguous(), ctx.scale)
        dvf = ops.scatter_rows_det(dvp.view(B, N * k, D), knn.view(B, N * k), N, csr) if ctx.needs_input_grad[1] else None
        return da, dvf, None, dvp, None, None, None


ATTN_CORE = os.environ.get("PTT_ATTN_CORE", "1") != "0"      # dev A/B: 0 = _PairInput / rows_mlp2 / _AttnAggregate as three functions


class _AttnCore(torch.autograd.Function):
    """The attention core of a Point-Transformer block (variants.py:158-163) as ONE function:
        t = q_i - kf[knn_ij] + pos_ij;  a = fc_gamma(t);  attn = softmax_j(a / sqrt(d));  res = sum_j attn * (vf[knn_ij] + pos_ij)
    Forward: the launches of _PairInput, _RowsMlp2 and _AttnAggregate. Backward: theirs too, except that three passes over
    (B, N, k, D) tensors ride in the epilogue of the GEMM that forms dt, the gradient of t (ptt_rows_gemm_rsum16_f32):
      * pos_enc receives two gradients (through t, and through the aggregate: dvp), which autograd would add with a pass of its own
        — dvp is that GEMM's residual, the sum dt + dvp its second output;
      * dq = sum_j dt_ij, a reduction pass — the sums over a point's 16 rows come out of the same epilogue;
      * dkf = -scatter_knn(dt): the negation is the scatter's own (ptt_scatter_rows_csr_sub_f32).
    The values are those of the three-function form (dt itself is still written: forming dq / dkf from dt + dvp instead would lose
    them to cancellation, dvp being ~1000 x larger). Per block and step 11 -> 8 launches, the two largest ATen launches gone."""

    @staticmethod
    def forward(ctx, q, kf, vf, knn, pos, W1, b1, W2, b2, scale, order, start):
        ctx.set_materialize_grads(False)
        q, kf, vf, pos = q.contiguous(), kf.contiguous(), vf.contiguous(), pos.contiguous()
        t = ops.pt_pair_input(q, kf, knn, pos)
        x2 = t.view(-1, t.shape[-1])
        h = lin_rows(x2, W1, b1.detach(), relu=True)
        a = lin_rows(h, W2, b2.detach()).view(*t.shape[:-1], W2.shape[0])
        attn, res = ops.pt_attn_train_fwd(a, vf, knn, pos, scale)
        ctx.save_for_backward(x2, h, attn, vf, knn, pos, order, start, W1, W2)      # the weights too: see _RowsLinear
        ctx.Ws, ctx.bs, ctx.scale = (W1, W2), (b1, b2), float(scale)
        ctx.mark_non_differentiable(attn)
        return res, attn

    @staticmethod
    def backward(ctx, dres, _dattn):
        x2, h, attn, vf, knn, pos, order, start = ctx.saved_tensors[:8]
        W1, W2 = ctx.Ws
        B, N, k, D = attn.shape
        csr = (order, start)
        dres = dres.contiguous()
        da, dvp = ops.pt_attn_train_bwd(attn, vf, knn, pos, dres, ctx.scale)
        dvf = ops.scatter_rows_det(dvp.view(B, N * k, D), knn.view(B, N * k), N, csr)
        dy2 = da.view(-1, D)
        rows, D1 = h.shape
        dW2 = weight_grad(W2, dy2, h)
        db2 = bias_grad(ctx.bs[1], dy2)
        if ops.rows_gemm_supported(rows, W2.shape[0], D1, dy2.stride(0), D1, x=dy2):
            dz1, db1 = ops.rows_gemm_masked(dy2, packed(W2, True), D1, h, want_colsum=True)
        else:
            dz1 = lin_rows(dy2, W2, transpose=True) * (h > 0)
            db1 = dz1.sum(0)
        dW1 = weight_grad(W1, dz1, x2)
        Din = W1.shape[1]
        if ops.rows_gemm_rsum16_supported(dz1, D1, Din) and k == 16 and Din == D:
            dt, dpos, dq = ops.rows_gemm_rsum16(dz1, packed(W1, True), Din, dvp.view(-1, D))
            dkf = ops.scatter_rows_det(dt.view(B, N * k, D), knn.view(B, N * k), N, csr, negate=True)
            dq, dpos = dq.view(B, N, D), dpos.view(B, N, k, D)
        else:
            dt = lin_rows(dz1, W1, transpose=True).view(B, N, k, D)
            dq = dt.sum(dim=2)
            dkf = ops.scatter_rows_det(dt.view(B, N * k, D), knn.view(B, N * k), N, csr).neg_()
            dpos = dt + dvp
        return dq, dkf, dvf, None, dpos, dW1, small_grad(ctx.bs[0], db1), dW2, db2, None, None, None


def attn_core(fc_gamma, q, kf, vf, knn, pos, scale, order, start):
    """-> (res (B,N,D), attn (B,N,k,D)) for fc_gamma = nn.Sequential(Linear, ReLU, Linear) through _AttnCore."""
    return _AttnCore.apply(q, kf, vf, knn, pos, fc_gamma[0].weight, fc_gamma[0].bias, fc_gamma[2].weight, fc_gamma[2].bias, scale, order, start)


def pt_block_usable(block, xyz, features):
    return (block.training and xyz.is_cuda and features.dtype == torch.float32 and block.k == 16 and block.d_model % 4 == 0
            and xyz.shape[1] * block.k <= 16384)


def lin_rows(x2, W, b=None, relu=False, residual=None, transpose=False):
    """relu?(x2 @ W^T + b) (+ residual) over (rows, K) activations (transpose: x2 @ W): the persistent row GEMM where the
    shape allows, else the linear kernel (K = 3, N = 1 / 5 / 259, ...)."""
    cout, K = (W.shape[1], W.shape[0]) if transpose else W.shape
    wp = packed(W, transpose)
    if ops.rows_gemm_supported(x2.shape[0], K, cout, x2.stride(0), cout, x=x2):
        return ops.rows_gemm(x2, wp, cout, bias=b, relu=relu, residual=residual)
    return ops.linear(x2, wp, cout, None, b, relu, residual)


class _RowsLinear(torch.autograd.Function):
    """y = x W^T + b (+ residual) over (rows, K) on the hand-written kernels: forward and input gradient on the row GEMM /
    linear kernel, weight gradient on ptt_linear_wgrad(2)_f32, bias gradient = column sums. Every nn.Linear of the
    Point-Transformer block in training mode (fc1, w_qs, w_ks, w_vs, fc2: variants.py:154-156,164), cov_final
    (pointnet2_backbone.py:46) and the last Conv1d of the heads' stacks."""

    @staticmethod
    def forward(ctx, x, W, b, residual):
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        assert W.dim() == 2
        ctx.save_for_backward(x2, W)                         # W too: autograd's version check then catches an in-place update
        ctx.W = W                                            # between forward and backward (backward re-packs the weight).
        #                                                      ctx.W = the object as passed (a parameter or a view of one): key of the pack cache
        ctx.b = b
        ctx.shape = x.shape
        r2 = residual.reshape(-1, W.shape[0]).contiguous() if residual is not None else None
        y = lin_rows(x2, W, b.detach() if b is not None else None, residual=r2)
        return y.view(*x.shape[:-1], W.shape[0])

    @staticmethod
    def backward(ctx, g):
        x2, _ = ctx.saved_tensors
        g2 = g.reshape(-1, g.shape[-1]).contiguous()
        dx = lin_rows(g2, ctx.W, transpose=True).view(ctx.shape) if ctx.needs_input_grad[0] else None
        dW = weight_grad(ctx.W, g2, x2) if ctx.needs_input_grad[1] else None
        db = bias_grad(ctx.b, g2) if ctx.needs_input_grad[2] else None
        return dx, dW, db, (g if ctx.needs_input_grad[3] else None)


def rows_linear(layer, x, residual=None):
    """nn.Linear / 1x1 Conv1d `layer` applied to x (..., K) (+ residual (..., Cout)) through _RowsLinear."""
    W = layer.weight
    return _RowsLinear.apply(x, W.reshape(W.shape[0], -1) if W.dim() > 2 else W, layer.bias, residual)


class _RowsMlp2(torch.autograd.Function):
    """y = relu(x W1^T + b1) W2^T + b2 over (rows, K): nn.Sequential(Linear, ReLU, Linear) — fc_delta and fc_gamma of the
    Point-Transformer block (variants.py:139-148) over the (point, neighbour) rows, 92 % of the block's FLOPs. Forward: bias
    and ReLU in the first GEMM's epilogue. Backward: the input gradient of the second layer comes out of its GEMM already
    masked by the ReLU, together with its column sums (= db1) — ptt_rows_gemm_masked_f32."""

    @staticmethod
    def forward(ctx, x, W1, b1, W2, b2):
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        h = lin_rows(x2, W1, b1.detach(), relu=True)
        y = lin_rows(h, W2, b2.detach())
        ctx.save_for_backward(x2, h, W1, W2)                 # the weights too: see _RowsLinear
        ctx.Ws, ctx.bs = (W1, W2), (b1, b2)
        ctx.shape = x.shape
        return y.view(*x.shape[:-1], W2.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, h, _, _ = ctx.saved_tensors
        W1, W2 = ctx.Ws
        dy2 = dy.reshape(-1, dy.shape[-1]).contiguous()
        rows, D1 = h.shape
        dW2 = weight_grad(W2, dy2, h)
        db2 = bias_grad(ctx.bs[1], dy2)
        if ops.rows_gemm_supported(rows, W2.shape[0], D1, dy2.stride(0), D1, x=dy2):
            dz1, db1 = ops.rows_gemm_masked(dy2, packed(W2, True), D1, h, want_colsum=True)
        else:
            dz1 = lin_rows(dy2, W2, transpose=True) * (h > 0)
            db1 = dz1.sum(0)
        dW1 = weight_grad(W1, dz1, x2)
        dx = lin_rows(dz1, W1, transpose=True).view(ctx.shape) if ctx.needs_input_grad[0] else None
        return dx, dW1, small_grad(ctx.bs[0], db1), dW2, db2


def rows_mlp2(seq, x):
    """nn.Sequential(Linear, ReLU, Linear) `seq` applied to x (..., K) through _RowsMlp2."""
    return _RowsMlp2.apply(x, seq[0].weight, seq[0].bias, seq[2].weight, seq[2].bias)


def conv1d_stack_usable(seq, x):
    """A heads-style Conv1d stack in training mode on a HIP device: [Conv1d(k=1, no bias) -> BatchNorm1d -> ReLU] units
    followed by at most one plain Conv1d(k=1) (centroids_voting_head.py:15-21, box_voting_head.py:25, p2b_xcoor.py:20-24)."""
    if not (seq.training and x.is_cuda and x.dtype == torch.float32 and len(seq) > 0):
        return False
    units = list(seq)
    for k, unit in enumerate(units):
        conv = getattr(unit, 'conv', None)
        if not isinstance(conv, nn.Conv1d) or conv.kernel_size != (1,) or conv.stride != (1,) or conv.padding != (0,) or conv.groups != 1:
            return False
        if hasattr(unit, 'normlayer'):
            bn = getattr(unit.normlayer, 'bn', None)
            if (not isinstance(bn, (nn.BatchNorm1d, nn.SyncBatchNorm)) or not bn.affine or not bn.track_running_stats
                    or bn.momentum is None or not bn.training or conv.bias is not None or conv.weight.shape[0] % 4):
                return False
            if not isinstance(getattr(unit, 'activation', None), nn.ReLU) or list(unit._modules.keys()) != ['conv', 'normlayer', 'activation']:
                return False
            if not _raw_pointer_ready(conv, bn, x.device):
                return False
        elif k != len(units) - 1 or list(unit._modules.keys()) != ['conv'] or not _raw_pointer_ready(conv, None, x.device):
            return False
    return True


def conv1d_stack_rows(seq, rows, residual=None):
    """seq(rows^T)^T in TRAINING mode for such a stack: rows (B, N, Cin) -> (B, N, Cout) (+ residual). The BatchNorm'd units run
    as one _SharedMlpPool over B * N rows (pool width 1: convolution + batch statistics out of the GEMM epilogue, deferred
    activation, fused BatchNorm / ReLU backward), the trailing plain convolution through _RowsLinear."""
    B, N, _ = rows.shape
    units = list(seq)
    bn_units = [u for u in units if hasattr(u, 'normlayer')]
    x = rows.reshape(B * N, -1)
    if bn_units:
        # (B*N, C) rows straight from the function: the (1, C, B*N) view + [0].t() of the general form cost a zero fill and two copies
        # per stack in the backward pass (select_backward, then a .contiguous() of the transposed gradient)
        x = rows_mlp_pool(x, bn_units, 1, 1, B * N, preact=False, as_rows=True)
    if len(bn_units) < len(units):
        last = units[-1].conv
        res2 = residual.reshape(B * N, -1) if residual is not None else None
        return rows_linear(last, x, res2).view(B, N, -1)
    assert residual is None
    return x.reshape(B, N, -1)


# --------------------------------------------------------------------------- the four tracking losses, one launch each way
class LossValues(object):
    """The un-weighted loss values a training forward reports (tb_dict / disp_dict of reference ptt.py:44-60) as numbers that
    are fetched from the device when first LOOKED at: the reference calls .item() four times inside the forward pass (four
    host-device synchronisations before the backward pass can be queued); here the four values live in one device tensor and
    one copy serves all of them, at the time a logger formats or adds them — never, if nobody looks."""

    class Value(object):
        """One of the values: behaves as the float it will be (arithmetic, comparisons, formatting, numpy / tensorboard conversion,
        pickling as a plain float); registered as numbers.Real."""
        __slots__ = ("owner", "k")

        def __init__(self, owner, k):
            self.owner, self.k = owner, k

        def __float__(self):
            return self.owner.fetch()[self.k]

        def item(self):
            return float(self)

        def __repr__(self):
            return repr(float(self))

        __str__ = __repr__

        def __format__(self, spec):
            return format(float(self), spec)

        def __array__(self, dtype=None, copy=None):
            import numpy as np
            return np.array(float(self), dtype=dtype or np.float64)

        def __reduce__(self):
            return (float, (float(self),))

        def __hash__(self):
            return hash(float(self))

        def __bool__(self):
            return bool(float(self))

        def __int__(self):
            return int(float(self))

    def __init__(self, device_values):
        self.device_values, self.host = device_values, None

    def fetch(self):
        if self.host is None:
            self.host = self.device_values.tolist()
        return self.host

    def __getitem__(self, k):
        return LossValues.Value(self, k)


class _TrackLosses(torch.autograd.Function):
    @staticmethod
    def forward(ctx, seed_cls, votes, box_data, centres, cls_label, search_inds, reg_label, pw_seed, pw_box, weights):
        args = (seed_cls.contiguous(), cls_label.contiguous(), search_inds.contiguous() if search_inds is not None else None,
                votes.contiguous(), reg_label.contiguous(), box_data.contiguous(), centres.detach().contiguous(), pw_seed, pw_box)
        total, out = ops.track_losses(*args, weights)
        ctx.save_for_backward(out, *[a for a in args if a is not None])
        ctx.has_inds, ctx.weights = search_inds is not None, tuple(weights)
        ctx.mark_non_differentiable(out)
        return total, out

    @staticmethod
    def backward(ctx, g_total, _g_out):
        out, t = ctx.saved_tensors[0], list(ctx.saved_tensors[1:])
        if not ctx.has_inds:
            t.insert(2, None)
        g = g_total.contiguous().float() if g_total is not None else None
        g_cls, g_votes, g_box = ops.track_losses_bwd(out, g, *t, ctx.weights)
        return g_cls, g_votes, g_box, None, None, None, None, None, None, None


def _numeric_protocol(cls):
    import numbers
    import operator
    for name in ("add", "sub", "mul", "truediv", "floordiv", "mod", "pow"):
        op = getattr(operator, name)
        setattr(cls, "__%s__" % name, lambda self, o, op=op: op(float(self), float(o) if isinstance(o, cls) else o))
        setattr(cls, "__r%s__" % name, lambda self, o, op=op: op(float(o) if isinstance(o, cls) else o, float(self)))
    for name in ("lt", "le", "gt", "ge", "eq", "ne"):
        op = getattr(operator, name)
        setattr(cls, "__%s__" % name, lambda self, o, op=op: op(float(self), float(o) if isinstance(o, cls) else o))
    for name, fn in (("neg", operator.neg), ("pos", operator.pos), ("abs", abs), ("round", round)):
        setattr(cls, "__%s__" % name, lambda self, *a, fn=fn: fn(float(self), *a))
    numbers.Real.register(cls)


_numeric_protocol(LossValues.Value)


def track_losses_usable(*tensors, search_inds=None, seeds_shape=None):
    """The one-launch losses take float32 device tensors and, with `search_inds`, a (B, N) int64 index table on the device
    (what ops._track_loss_desc would otherwise refuse with a ValueError: the caller falls back to the heads' own losses)."""
    if not all(t is not None and t.is_cuda and t.dtype == torch.float32 for t in tensors):
        return False
    if search_inds is not None:
        return bool(search_inds.is_cuda and search_inds.dtype == torch.int64
                    and (seeds_shape is None or tuple(search_inds.shape) == tuple(seeds_shape)))
    return True


def track_losses(seed_cls, votes, box_data, centres, cls_label, search_inds, reg_label, pw_seed, pw_box, weights):
    """-> (total loss (scalar tensor with the graph), LossValues of [total, seed cls, seed reg, proposal cls, proposal reg])."""
    total, out = _TrackLosses.apply(seed_cls, votes, box_data, centres, cls_label, search_inds, reg_label, pw_seed, pw_box, tuple(weights))
    return total, LossValues(out.detach())
