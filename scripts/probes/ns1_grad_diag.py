"""Diagnostic: SharedMLP + max-pool on the row kernels and on stock torch (MIOpen) against the SAME modules in float64, per-gradient
relative error, for row counts that are / are not multiples of the GEMM's 64-row tile. (Round 5: at M = 1031, ns <= 2 the row
kernels and stock torch disagreed by 1e-2 .. 1e-1 in every gradient — which side is off?)"""
import copy, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ptt_amd import train_ops
from ptt_amd.models.backbones_3d.pointnet2 import pytorch_utils as pt_utils
dev = torch.device("cuda:0")
for M, ns, spec in [(1024, 1, [128, 128, 256]), (1031, 1, [128, 128, 256]), (1031, 2, [128, 128, 256]), (1031, 32, [128, 128, 256]), (1000, 1, [256, 256, 256]),
                    (1031, 1, [256, 256, 256])]:
    torch.manual_seed(11)
    a = pt_utils.SharedMLP(list(spec), bn=True).to(dev).train()
    b = copy.deepcopy(a)
    d = copy.deepcopy(a).double()
    x0 = torch.randn(1, spec[0], M, ns, device=dev)
    up = torch.randn(1, spec[-1], M, device=dev)
    x1, x2, x3 = x0.clone().requires_grad_(True), x0.clone().requires_grad_(True), x0.double().requires_grad_(True)
    y1 = train_ops.shared_mlp_pool(x1, a, 3); y1.backward(up)
    y2 = b(x2).max(dim=3)[0]; y2.backward(up)
    y3 = d(x3).max(dim=3)[0]; y3.backward(up.double())
    for tag, y, x, m in (("row kernels", y1, x1, a), ("stock torch", y2, x2, b)):
        errs = {"y": float((y.detach().double() - y3.detach()).abs().max() / y3.detach().abs().max()),
                "input": float((x.grad.double() - x3.grad).abs().max() / x3.grad.abs().max())}
        for (n, p), (_, q) in zip(m.named_parameters(), d.named_parameters()):
            errs[n.replace("normlayer.bn.", "bn.").replace("layer", "L")] = float((p.grad.double() - q.grad).abs().max() / (q.grad.abs().max() + 1e-12))
        print("M=%d ns=%d %s  %-11s vs float64: %s" % (M, ns, spec, tag, " ".join("%s=%.1e" % kv for kv in errs.items())), flush=True)
