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


def fold_layers(layers, dev, ops):
    """oracle layer dicts -> the (wpacked, scale, shift, cin, cout, relu) tuples of ops.sa_fused_forward."""
    out = []
    for li, L in enumerate(layers):
        w = L["conv_weight"].to(dev)
        rot = 3 if (li == 0 and w.shape[1] > 3) else 0      # the fused kernel's row layout is [features | xyz]
        scale = (L["bn_weight"] / torch.sqrt(L["bn_var"] + L["eps"]))
        shift = L["bn_bias"] - L["bn_mean"] * scale
        out.append((ops.pack_weight(w, rot), scale.to(dev).contiguous(), shift.to(dev).contiguous(),
                    w.shape[1], w.shape[0], True))
    return out
