#!/usr/bin/env python
"""Dev: where do the slow GEMMs of the TransformerBlock training path come from? Times fwd / dX / dW formulations of a
(98304 x 512) @ (512 x 512) linear layer in fp32."""
import os, sys, torch, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ptt_amd import ops
dev = torch.device("cuda:0")
B, N, k, D = 48, 128, 16, 512
x4 = torch.randn(B, N, k, D, device=dev)
W = torch.randn(D, D, device=dev) / 22
b = torch.randn(D, device=dev)
dy4 = torch.randn(B, N, k, D, device=dev)
x2, dy2 = x4.reshape(-1, D), dy4.reshape(-1, D)
def t(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
fl = 2.0 * B * N * k * D * D
def show(name, ms): print("%-46s %8.3f ms  %6.1f TFLOP/s" % (name, ms, fl / ms / 1e9))
show("fwd  F.linear 4-D", t(lambda: torch.nn.functional.linear(x4, W, b)))
show("fwd  F.linear 2-D", t(lambda: torch.nn.functional.linear(x2, W, b)))
wp = ops.pack_weight(W)
show("fwd  ops.linear (MFMA kernel)", t(lambda: ops.linear(x2, wp, D, None, b)))
show("dX   dy2 @ W", t(lambda: dy2 @ W))
show("dX   dy4 @ W (4-D matmul)", t(lambda: dy4 @ W))
wtp = ops.pack_weight(W.t().contiguous())
show("dX   ops.linear(dy, pack(W^T))", t(lambda: ops.linear(dy2, wtp, D)))
show("dW   dy2.t() @ x2", t(lambda: dy2.t() @ x2))
show("dW   (x2.t() @ dy2).t()", t(lambda: (x2.t() @ dy2).t()))
dyt = dy2.t().contiguous()
show("dW   dy2.t().contiguous() @ x2 (incl. copy)", t(lambda: dy2.t().contiguous() @ x2))
show("dW   einsum bnkd,bnke->de", t(lambda: torch.einsum('bnkd,bnke->de', dy4, x4)))
xs, dys = x2.view(64, -1, D), dy2.view(64, -1, D)
show("dW   split-K: bmm 64 chunks + sum", t(lambda: torch.bmm(dys.transpose(1, 2), xs).sum(0)))
# autograd end to end
xa = x4.clone().requires_grad_(True); lin = torch.nn.Linear(D, D).to(dev)
def fb():
    y = lin(xa); y.backward(dy4)
show("autograd nn.Linear fwd+bwd 4-D (3 GEMMs)", t(fb) )
xa2 = x2.clone().requires_grad_(True)
def fb2():
    y = lin(xa2); y.backward(dy2)
show("autograd nn.Linear fwd+bwd 2-D (3 GEMMs)", t(fb2))
