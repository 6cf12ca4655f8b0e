"""N3 — the TRAINING forward and backward of the shared-MLP stages on the hand-written kernels of
ptt_amd/csrc/train_ops.hip (+ ptt_linear_f32): the reference's SharedMLP in train mode followed by the max over the
neighbour axis (pytorch_utils.py:12-36,94-114 + pointnet2_modules.py:84-88; similarity_modules/p2b_xcoor.py:39-41), as
ONE autograd function over (rows, channels) activations.

    y = shared_mlp_pool(grouped, mlp, pool_dim)      # grouped (B,C,M,ns) as QueryAndGroup / the fusion tensor yields it

replaces `mlp(grouped).max(dim=pool_dim)[0]` (= F.max_pool2d over that axis) whenever the module is in training mode on
a HIP device and has the plain [conv1x1 (no bias) -> BatchNorm2d -> ReLU] units every shipped config builds. Forward:
per layer ptt_linear_f32 -> ptt_bn_stats_f32 -> ptt_bn_apply_f32, then ptt_pool_rows_f32. Backward: ptt_pool_rows_bwd_f32,
then per layer ptt_bn_bwd_f32 -> ptt_linear_wgrad_f32 (weight gradient) -> ptt_linear_f32 on the transposed weight
(input gradient). Running statistics are updated as nn.BatchNorm2d does (momentum, unbiased variance, batch counter).
"""
import torch
import torch.nn as nn

from . import ops


def usable(mlp, x):
    """Plain SharedMLP units, float32 on a HIP device, training mode."""
    if not (mlp.training and x.is_cuda and x.dtype == torch.float32 and len(mlp) > 0):
        return False
    for unit in mlp:
        conv = getattr(unit, 'conv', None)
        bn = getattr(getattr(unit, 'normlayer', None), 'bn', None)
        if not isinstance(conv, nn.Conv2d) or conv.kernel_size != (1, 1) or conv.bias is not None:
            return False
        if not isinstance(bn, nn.BatchNorm2d) or not bn.affine or not bn.track_running_stats or bn.momentum is None:
            return False
        if not isinstance(getattr(unit, 'activation', None), nn.ReLU):
            return False
        if list(unit._modules.keys()) != ['conv', 'normlayer', 'activation']:
            return False
    return True


class _SharedMlpPool(torch.autograd.Function):
    """(rows (R,C0), ns, eps per layer, [W, gamma, beta] per layer) -> (pooled (R/ns, C_L), [mean, var] per layer)."""

    @staticmethod
    def forward(ctx, x, ns, eps, *params):
        L = len(params) // 3
        saved, stats = [], []
        cur = x.contiguous()
        for l in range(L):
            W, gamma, beta = params[3 * l], params[3 * l + 1], params[3 * l + 2]
            cout = W.shape[0]
            z = ops.linear(cur, ops.pack_weight(W), cout)
            mean, var, invstd = ops.bn_stats(z, eps[l])
            nxt = ops.bn_apply(z, mean, invstd, gamma.detach(), beta.detach(), relu=True)
            saved += [cur, z, nxt, mean, invstd]
            stats += [mean, var]
            cur = nxt
        pooled, arg = ops.pool_rows(cur, ns)
        ctx.save_for_backward(arg, *saved, *[p.detach() for p in params])
        ctx.L, ctx.ns = L, int(ns)
        ctx.mark_non_differentiable(*stats)
        return (pooled,) + tuple(stats)

    @staticmethod
    def backward(ctx, dpooled, *unused):
        L, ns = ctx.L, ctx.ns
        t = ctx.saved_tensors
        arg, saved, params = t[0], t[1:1 + 5 * L], t[1 + 5 * L:]
        g = ops.pool_rows_bwd(dpooled.contiguous(), arg, ns)
        grads = [None] * (3 * L)
        for l in range(L - 1, -1, -1):
            x_in, z, act, mean, invstd = saved[5 * l:5 * l + 5]
            W, gamma = params[3 * l], params[3 * l + 1]
            dz, dgamma, dbeta = ops.bn_bwd(g, act, z, mean, invstd, gamma, out=g)      # in place over the incoming gradient
            w2 = W.reshape(W.shape[0], -1)
            grads[3 * l] = ops.linear_wgrad(dz, x_in).view_as(W)
            grads[3 * l + 1], grads[3 * l + 2] = dgamma, dbeta
            if l > 0 or ctx.needs_input_grad[0]:
                g = ops.linear(dz, ops.pack_weight(w2.t().contiguous()), w2.shape[1])
            else:
                g = None
        return (g, None, None) + tuple(grads)


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
    params, eps = [], []
    for unit in mlp:
        bn = unit.normlayer.bn
        params += [unit.conv.weight, bn.weight, bn.bias]
        eps.append(float(bn.eps))
    out = _SharedMlpPool.apply(rows, ns, tuple(eps), *params)
    pooled, stats = out[0], out[1:]
    R = rows.shape[0]
    with torch.no_grad():                                   # nn.BatchNorm2d's bookkeeping in training mode
        for l, unit in enumerate(mlp):
            bn = unit.normlayer.bn
            m = bn.momentum
            bn.running_mean.mul_(1 - m).add_(stats[2 * l], alpha=m)
            bn.running_var.mul_(1 - m).add_(stats[2 * l + 1], alpha=m * R / max(R - 1, 1))
            bn.num_batches_tracked.add_(1)
    return pooled.view(B, keep, -1).transpose(1, 2)         # (B, C_L, keep)
