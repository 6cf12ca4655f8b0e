"""Shared helpers for the tests: numpy-seeded weights (bit-reproducible on any box)."""
import numpy as np
import torch


def mlp_layers(seed, spec, nontrivial_bn=True):
    """SharedMLP weights for channel spec [c0, c1, ...] as oracle.dense_ref.shared_mlp_eval wants them."""
    rs = np.random.RandomState(seed)
    layers = []
    for cin, cout in zip(spec[:-1], spec[1:]):
        w = (rs.standard_normal((cout, cin, 1, 1)) * np.sqrt(2.0 / cin)).astype(np.float32)
        if nontrivial_bn:
            g = rs.uniform(0.5, 1.5, cout).astype(np.float32) * np.where(rs.rand(cout) < 0.1, -1, 1).astype(np.float32)
            b = (rs.standard_normal(cout) * 0.2).astype(np.float32)
            m = (rs.standard_normal(cout) * 0.3).astype(np.float32)
            v = rs.uniform(0.3, 2.0, cout).astype(np.float32)
        else:
            g, b = np.ones(cout, np.float32), np.zeros(cout, np.float32)
            m, v = np.zeros(cout, np.float32), np.ones(cout, np.float32)
        layers.append({"conv_weight": torch.from_numpy(w), "bn_weight": torch.from_numpy(g),
                       "bn_bias": torch.from_numpy(b), "bn_mean": torch.from_numpy(m),
                       "bn_var": torch.from_numpy(v), "eps": 1e-5})
    return layers


def transformer_params(seed, d_points=256, d_model=512):
    rs = np.random.RandomState(seed)

    def lin(o, i, bias=True, name=""):
        bound = 1.0 / np.sqrt(i)
        P[name + ".weight"] = torch.from_numpy(rs.uniform(-bound, bound, (o, i)).astype(np.float32))
        if bias:
            P[name + ".bias"] = torch.from_numpy(rs.uniform(-bound, bound, o).astype(np.float32))

    P = {}
    lin(d_model, d_points, True, "fc1")
    lin(d_points, d_model, True, "fc2")
    lin(d_model, 3, True, "fc_delta.0")
    lin(d_model, d_model, True, "fc_delta.2")
    lin(d_model, d_model, True, "fc_gamma.0")
    lin(d_model, d_model, True, "fc_gamma.2")
    lin(d_model, d_model, False, "w_qs")
    lin(d_model, d_model, False, "w_ks")
    lin(d_model, d_model, False, "w_vs")
    return P


def fold_layers(layers, dev, ops, scale_in_weights=False):
    """oracle layer dicts -> the (wpacked, scale, shift, cin, cout, relu) tuples of ops.sa_fused_forward.
    scale_in_weights: the BatchNorm scale multiplied into the packed weights (scale = None), which is what the modules
    pass; the default keeps the separate-scale form of the C ABI covered."""
    out = []
    for li, L in enumerate(layers):
        w = L["conv_weight"].to(dev)
        rot = 3 if (li == 0 and w.shape[1] > 3) else 0      # the fused kernel's row layout is [features | xyz]
        scale = (L["bn_weight"] / torch.sqrt(L["bn_var"] + L["eps"]))
        shift = L["bn_bias"] - L["bn_mean"] * scale
        if scale_in_weights:
            out.append((ops.pack_weight(w * scale.to(dev).view(-1, 1, 1, 1), rot), None, shift.to(dev).contiguous(),
                        w.shape[1], w.shape[0], True))
        else:
            out.append((ops.pack_weight(w, rot), scale.to(dev).contiguous(), shift.to(dev).contiguous(),
                        w.shape[1], w.shape[0], True))
    return out


def cosine_sim_params(seed):
    """Weights of CosineSimAug (SharedMLP [260,256,256,256] + Seq conv stack) as oracle.dense_ref.cosine_sim_aug wants."""
    rs = np.random.RandomState(seed)
    mlp = mlp_layers(seed, [260, 256, 256, 256])
    t = lambda a: torch.from_numpy(a.astype(np.float32))
    conv = {"conv0_weight": t(rs.standard_normal((256, 256, 1)) / 16), "bn0_weight": t(rs.uniform(0.5, 1.5, 256)),
            "bn0_bias": t(rs.standard_normal(256) * 0.1), "bn0_mean": t(rs.standard_normal(256) * 0.2),
            "bn0_var": t(rs.uniform(0.5, 1.5, 256)), "conv1_weight": t(rs.standard_normal((256, 256, 1)) / 16),
            "conv1_bias": t(rs.standard_normal(256) * 0.1)}
    return mlp, conv


def load_cosine_sim(module, mlp, conv):
    """Copy those weights into a CosineSimAug module (reference's or ours: same attribute names)."""
    with torch.no_grad():
        for unit, L in zip(module.mlp, mlp):
            unit.conv.weight.copy_(L["conv_weight"])
            bn = unit.normlayer.bn
            bn.weight.copy_(L["bn_weight"]); bn.bias.copy_(L["bn_bias"])
            bn.running_mean.copy_(L["bn_mean"]); bn.running_var.copy_(L["bn_var"])
        c0, c1 = module.conv[0], module.conv[1]
        c0.conv.weight.copy_(conv["conv0_weight"])
        c0.normlayer.bn.weight.copy_(conv["bn0_weight"]); c0.normlayer.bn.bias.copy_(conv["bn0_bias"])
        c0.normlayer.bn.running_mean.copy_(conv["bn0_mean"]); c0.normlayer.bn.running_var.copy_(conv["bn0_var"])
        c1.conv.weight.copy_(conv["conv1_weight"]); c1.conv.bias.copy_(conv["conv1_bias"])


def fill_state_dict_(model, seed):
    """Deterministic numpy-seeded values for EVERY entry of model.state_dict(), in sorted key order, so that two
    models with the same key set (the reference's and ours) end up with identical weights on any box."""
    rs = np.random.RandomState(seed)
    sd = model.state_dict()
    new = {}
    for k in sorted(sd.keys()):
        v = sd[k]
        shape = tuple(v.shape)
        if not v.is_floating_point():
            new[k] = torch.zeros_like(v)
        elif k.endswith("running_var"):
            new[k] = torch.from_numpy(rs.uniform(0.5, 1.5, shape).astype(np.float32))
        elif k.endswith("running_mean"):
            new[k] = torch.from_numpy((rs.standard_normal(shape) * 0.2).astype(np.float32))
        elif k.endswith("bn.weight"):
            new[k] = torch.from_numpy(rs.uniform(0.5, 1.5, shape).astype(np.float32))
        elif k.endswith("pos_weight"):
            new[k] = v.clone()
        elif len(shape) >= 2:
            fan_in = int(np.prod(shape[1:]))
            new[k] = torch.from_numpy((rs.standard_normal(shape) * np.sqrt(2.0 / fan_in)).astype(np.float32))
        else:
            new[k] = torch.from_numpy((rs.standard_normal(shape) * 0.1).astype(np.float32))
    model.load_state_dict(new)
    return model


def outside(a, b, atol=1e-4, rtol=1e-4):
    """(number of elements of `a` outside atol + rtol * |b| of `b`, worst error in units of that tolerance): the tests
    that hold a float32 result against the north_star bar assert the COUNT they observed instead of loosening the bar."""
    a, b = a.detach().double(), b.detach().double()
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    return int((err > tol).sum()), float((err / tol).max())
