#!/usr/bin/env python
"""Dev: accuracy of the stock ROCm layers the training path uses (1x1 Conv2d fwd / dgrad / wgrad, BatchNorm2d training
fwd / bwd) against float64, next to the same maths written as matmuls / with MIOpen switched off."""
import torch, time
dev = torch.device("cuda:0")
torch.manual_seed(0)
B, C, M, ns, Co = 48, 131, 256, 32, 128
x = torch.randn(B, C, M, ns, device=dev); W = torch.randn(Co, C, 1, 1, device=dev) / C ** 0.5; gy = torch.randn(B, Co, M, ns, device=dev)
def rel(a, b): return float((a.double() - b).norm() / b.norm())
def timed(fn, n=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
xd, Wd, gyd = x.double(), W.double(), gy.double()
yd = torch.einsum('oc,bcmn->bomn', Wd[:, :, 0, 0], xd)
dxd = torch.einsum('oc,bomn->bcmn', Wd[:, :, 0, 0], gyd)
dWd = torch.einsum('bomn,bcmn->oc', gyd, xd)
def conv_path():
    xx = x.clone().requires_grad_(True); ww = W.clone().requires_grad_(True)
    y = torch.nn.functional.conv2d(xx, ww); y.backward(gy); return y, xx.grad, ww.grad
def mm_path():
    xx = x.clone().requires_grad_(True); ww = W.clone().requires_grad_(True)
    y = torch.matmul(ww.view(Co, C), xx.view(B, C, M * ns)).view(B, Co, M, ns); y.backward(gy); return y, xx.grad, ww.grad
for name, fn in (("conv2d (MIOpen)", conv_path), ("matmul", mm_path)):
    y, dx, dW = fn()
    print("%-18s fwd %.2e  dgrad %.2e  wgrad %.2e   %.2f ms fwd+bwd" % (name, rel(y, yd), rel(dx, dxd), rel(dW[:, :, 0, 0] if dW.dim() == 4 else dW, dWd), timed(fn)))
# BatchNorm2d training
bn = torch.nn.BatchNorm2d(Co).to(dev).train(); bnd = torch.nn.BatchNorm2d(Co).to(dev).double().train()
with torch.no_grad():
    bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(); bnd.weight.copy_(bn.weight); bnd.bias.copy_(bn.bias)
z = (torch.randn(B, Co, M, ns, device=dev) * 2 + 0.5)
def bn_path(enabled):
    with torch.backends.cudnn.flags(enabled=enabled):
        zz = z.clone().requires_grad_(True); bn.zero_grad()
        o = bn(zz); o.backward(gy); return o, zz.grad, bn.weight.grad.clone(), bn.bias.grad.clone()
zz = z.double().clone().requires_grad_(True); od = bnd(zz); od.backward(gyd)
for name, en in (("BN MIOpen", True), ("BN native", False)):
    o, dz, dg, db = bn_path(en)
    print("%-18s fwd %.2e  dx %.2e  dgamma %.2e  dbeta %.2e   %.2f ms fwd+bwd" % (name, rel(o, od), rel(dz, zz.grad), rel(dg, bnd.weight.grad), rel(db, bnd.bias.grad), timed(lambda: bn_path(en))))
