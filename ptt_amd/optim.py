"""Gradient clipping + Adam of the training step (reference tools/train_utils/train_utils.py:47-51:
`clip_grad_norm_(model.parameters(), GRAD_NORM_CLIP); optimizer.step()` with the optimizer of
tools/train_utils/optimization/__init__.py:12-14) as TWO launches over a device table of all parameters
(ptt_adam_clip_step_f32, ptt_amd/csrc/step_ops.hip) instead of torch's ~25 multi-tensor launches.

ClipAdam IS a torch.optim.Adam: same constructor, same param_groups, same state keys ('step', 'exp_avg', 'exp_avg_sq'), so
state_dict() / load_state_dict() interchange with the stock optimizer and LR schedulers drive it unchanged. Only step() differs:
`step(max_norm=10.0)` clips and updates in one pass. Whatever the table does not take (several param groups with different
hyper-parameters are fine; amsgrad / maximize / non-float32 / CPU parameters are not) runs on the stock path."""
import math

import torch

from . import ops


def _bump_versions(params):
    """The launch updated the parameters through raw pointers: tell autograd (saved-tensor checks) and every cache keyed on
    `_version` (train_ops.packed: the MFMA weight packs) that they changed, as an in-place torch op would have."""
    try:
        torch._C._autograd._unsafe_set_version_counter(params, [p._version + 1 for p in params])
    except (AttributeError, TypeError):
        torch._foreach_add_(params, 0.0)             # older torch: a real (multi-tensor) in-place op


class ClipAdam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad, foreach=True)
        self._tables = {}                 # group index -> (ids of the parameters in the table, ops.AdamTable)
        self.last_norm = None             # the total gradient norm of the last clipped step: a (1,) device tensor
        self._ring = None                 # ops.AdamHyperRing of the captured form of the step
        self._graph = None                # (group, params, table, step counters) pinned by prepare_graph_step()

    def _fusable(self, group, params):
        return (not group['amsgrad'] and not group.get('maximize', False) and not group.get('capturable', False)
                and not group.get('differentiable', False) and len(params) > 0
                and all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.grad.dtype == torch.float32
                        and not p.grad.is_sparse and p.grad.is_contiguous() and p.device == params[0].device for p in params))

    @torch.no_grad()
    def step(self, closure=None, max_norm=None):
        """max_norm: clip the global gradient norm first (torch.nn.utils.clip_grad_norm_ over ALL parameters of the optimizer)."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        groups = [(gi, g, [p for p in g['params'] if p.grad is not None]) for gi, g in enumerate(self.param_groups)]
        one_table = len(groups) == 1 and self._fusable(groups[0][1], groups[0][2])
        if not one_table:
            # several groups share one norm: clip with torch, then update every group that qualifies without clipping
            if max_norm is not None:
                self.last_norm = torch.nn.utils.clip_grad_norm_([p for _, _, ps in groups for p in ps], max_norm)
            max_norm = None
        for gi, group, params in groups:
            if not params:
                continue
            if not self._fusable(group, params):
                self._stock_group(group)
                continue
            for p in params:
                st = self.state[p]
                if len(st) == 0:
                    st['step'] = torch.tensor(0.0, dtype=torch.float32)
                    st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
            steps = [self.state[p]['step'] for p in params]
            if any(s.is_cuda for s in steps):
                self._stock_group(group)
                continue
            torch._foreach_add_(steps, 1)
            t = float(steps[0])
            if float(steps[-1]) != t:
                raise RuntimeError("ClipAdam: parameters of one group at different step counts")
            key = tuple((id(p), p.data_ptr()) for p in params)      # a moved parameter (model.to(...), p.data = ...) rebuilds the table
            cached = self._tables.get(gi)
            if cached is None or cached[0] != key:
                cached = (key, ops.AdamTable(params, [self.state[p]['exp_avg'] for p in params], [self.state[p]['exp_avg_sq'] for p in params]))
                self._tables[gi] = cached
            beta1, beta2 = group['betas']
            lr = float(group['lr'])
            norm = cached[1].step([p.grad for p in params], beta1, beta2, group['eps'], lr / (1.0 - beta1 ** t), math.sqrt(1.0 - beta2 ** t),
                                  group['weight_decay'], max_norm if max_norm is not None else 0.0)
            if max_norm is not None:
                self.last_norm = norm
            _bump_versions(params)
        return loss

    # ------------------------------------------------------------------ the step in two halves, for a captured training step
    def _graph_table(self):
        """(group, params, AdamTable) when the whole optimizer is ONE table that has already stepped eagerly (state, table and
        gradient addresses exist) — the precondition of the captured form; None otherwise."""
        if len(self.param_groups) != 1:
            return None
        group = self.param_groups[0]
        params = [p for p in group['params'] if p.grad is not None]
        cached = self._tables.get(0)
        if not params or cached is None or not self._fusable(group, params):
            return None
        if cached[0] != tuple((id(p), p.data_ptr()) for p in params) or not cached[1].check_grads_in_place([p.grad for p in params]):
            return None
        return group, params, cached[1]

    def prepare_graph_step(self):
        """Before a capture: checks the one-table state and pins it (the per-step halves below do no checking of their own — they
        run once per replay). False when the optimizer cannot take the captured form (it then steps eagerly)."""
        hit = self._graph_table()
        if hit is None:
            self._graph = None
            return False
        group, params, table = hit
        if self._ring is None:
            self._ring = ops.AdamHyperRing(table.device)
        self._graph = (group, params, table, [self.state[p]['step'] for p in params])
        return True

    def begin_graph_step(self, max_norm=None):
        """Host half of a captured step, BEFORE the replay: the step counters advance and this step's hyper-parameters (the lr of
        the moment, the bias corrections) go to the device struct the recorded launches read."""
        group, params, table, steps = self._graph
        torch._foreach_add_(steps, 1)
        t = float(steps[0])
        beta1, beta2 = group['betas']
        self._ring.upload(table.hyper(beta1, beta2, group['eps'], float(group['lr']) / (1.0 - beta1 ** t), math.sqrt(1.0 - beta2 ** t),
                                      group['weight_decay'], max_norm if max_norm is not None else 0.0))

    def record_graph_step(self, max_norm=None):
        """Device half: the two launches, reading the device struct (called while the stream is capturing)."""
        table = self._graph[2]
        norm = table.step_device_hyper(self._ring.dev, clip=max_norm is not None)
        if max_norm is not None:
            self.last_norm = norm

    def end_graph_step(self):
        """Host half AFTER the replay was queued: the parameters changed under autograd's feet."""
        _bump_versions(self._graph[1])

    def _stock_group(self, group):
        saved = self.param_groups
        try:
            self.param_groups = [group]
            torch.optim.Adam.step(self)
        finally:
            self.param_groups = saved

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._graph = None
        self._tables = {}                 # the moments are new tensors
