"""Dev tool: the ATen operators one training step of the full tracker still dispatches, grouped by (operator, Python call
site inside this package) with the element counts — where the small launches come from. Uses a TorchDispatchMode, so it
sees operators, not kernels; pair it with rocprofv3 --kernel-trace for time."""
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ptt_amd.config import StubDataset, ptt_model_cfg
from ptt_amd.models import build_network
from ptt_amd.train_step import DataParallelTrainer, synthetic_train_batch

SKIP = ("aten::view", "aten::_unsafe_view", "aten::reshape", "aten::as_strided", "aten::slice", "aten::select", "aten::t",
        "aten::transpose", "aten::permute", "aten::detach", "aten::alias", "aten::expand", "aten::unsqueeze", "aten::squeeze",
        "aten::empty", "aten::empty_like", "aten::empty_strided", "aten::_local_scalar_dense", "aten::item", "aten::size",
        "aten::stride", "aten::is_contiguous", "aten::unbind", "aten::split", "aten::chunk", "aten::narrow", "aten::lift_fresh",
        "aten::new_empty", "aten::sym_size", "aten::sym_stride", "aten::sym_numel", "aten::storage_offset", "aten::sym_storage_offset")


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.seen = collections.defaultdict(lambda: [0, 0])

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func._schema.name
        if name not in SKIP:
            n = 0
            for o in (out if isinstance(out, (tuple, list)) else (out,)):
                if isinstance(o, torch.Tensor) and o.is_cuda:
                    n += o.numel()
            if n == 0:
                for a in args:
                    if isinstance(a, torch.Tensor) and a.is_cuda:
                        n = max(n, a.numel())
            if n:
                site = "(autograd engine / torch)"
                for fr in reversed(traceback.extract_stack(limit=24)):
                    if "/ptt_amd/" in fr.filename or fr.filename.endswith("bench.py"):
                        site = "%s:%d %s" % (os.path.relpath(fr.filename, ROOT), fr.lineno, fr.name)
                        break
                k = (name, site)
                self.seen[k][0] += 1
                self.seen[k][1] += n
        return out


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    model = build_network(ptt_model_cfg(), 1, StubDataset(training=True)).to(dev).train()
    trainer = DataParallelTrainer(model, dev)
    batch = synthetic_train_batch(100, 48, dev)
    for _ in range(2):
        trainer.step(batch)
    torch.cuda.synchronize()
    with Log() as log:
        trainer.step(batch)
        torch.cuda.synchronize()
    by_op = collections.defaultdict(lambda: [0, 0])
    for (name, _), (c, n) in log.seen.items():
        by_op[name][0] += c
        by_op[name][1] += n
    print("operators that touch device memory in ONE training step (calls, elements):")
    for name, (c, n) in sorted(by_op.items(), key=lambda kv: -kv[1][0])[:40]:
        print("  %-40s %5d  %14d" % (name, c, n))
    print("\nby call site (calls >= 2 or >= 1M elements):")
    for (name, site), (c, n) in sorted(log.seen.items(), key=lambda kv: (-kv[1][0], -kv[1][1])):
        if c >= 2 or n >= 1 << 20:
            print("  %-30s x%-4d %13d  %s" % (name, c, n, site))


if __name__ == "__main__":
    main()
