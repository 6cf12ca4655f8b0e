#!/usr/bin/env python
"""Dev tool: per-phase cycle stamps of sa_fused_kernel (PTT_DEBUG_STAMPS hook)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptt_amd import ops, synth
from tests.util import fold_layers, mlp_layers
dev = torch.device("cuda:0"); B = 48
cases = {"sa1": (512, 256, 128, [131, 128, 128, 256], 0.5, 32), "sa2": (256, 128, 256, [259, 128, 128, 256], 0.7, 32),
         "box": (128, 64, 257, [260, 256, 256, 256], 0.3, 16)}
for name in sys.argv[1:] or ["sa1", "sa2", "box"]:
    N, M, C, spec, r, ns = cases[name]
    s, _ = synth.frames(2, B, N, 64, K_s=max(64, int(N * 0.3))); xyz = torch.from_numpy(s).to(dev)
    new_xyz = xyz[:, :M].contiguous(); idx = ops.ball_query(new_xyz, xyz, r, ns)
    feats = torch.randn(B, N, C, device=dev).transpose(1, 2)
    if name == "box": feats = feats.contiguous()
    layers = fold_layers(mlp_layers(3, spec), dev, ops, scale_in_weights=True)
    fn = lambda: ops.sa_fused_forward(xyz, new_xyz, idx, feats, layers, r, True, True)
    if os.environ.get("HOIST", "1") != "0":      # what the module runs: layer 0 as a per-point term
        w0 = mlp_layers(3, spec)[0]["conv_weight"].reshape(spec[1], spec[0]).to(dev)
        term = ops.linear(feats.transpose(1, 2).contiguous(), ops.pack_weight(w0[:, 3:].contiguous()), spec[1],
                          fold_layers(mlp_layers(3, [cases[name][3][0], cases[name][3][1]]), dev, ops)[0][1], layers[0][2], relu=False)
        wx = (w0[:, 0:3] * fold_layers(mlp_layers(3, [cases[name][3][0], cases[name][3][1]]), dev, ops)[0][1][:, None]).t().contiguous()
        fn = lambda: ops.sa_fused_forward(xyz, new_xyz, idx, None, layers[1:], r, True, True, l0=(term, wx, True))
        spec = spec[1:]
    for _ in range(3): fn()
    buf = torch.zeros(4096 * 8, dtype=torch.int64, device=dev)
    os.environ["PTT_DEBUG_STAMPS"] = "%x" % buf.data_ptr()
    fn(); torch.cuda.synchronize(); os.environ.pop("PTT_DEBUG_STAMPS")
    nl = len(spec) - 1
    st = buf.cpu().numpy().reshape(4096, 8); st = st[st[:, 1 + nl] > 0]
    d = np.diff(st[:, :2 + nl], axis=1)
    mf = [2 * 4 * ((ci + 7) // 8) * max(1, co // 128) * 64 for ci, co in zip(spec[:-1], spec[1:])]
    print(name, "WGs", len(st), "total median", int(np.median(st[:, 1 + nl] - st[:, 0])), "MFMA cycles/wave per layer", mf)
    for i, n in enumerate(["gather", "layer0", "layer1", "layer2"][:1 + nl]):
        print("   %-8s %8.0f %8.0f %8.0f" % (n, np.median(d[:, i]), np.percentile(d[:, i], 10), np.percentile(d[:, i], 90)))
