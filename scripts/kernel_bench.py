#!/usr/bin/env python
"""Micro-benchmarks of the individual hot-path kernels at bench.py's shapes (B=48, KITTI-Car config).
Used for A/B work and as the small target of `rocprofv3 --pmc` passes. Prints one line per case:
name, mean ms over --iters launches (HIP events on the launch stream), algorithmic TFLOP/s."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptt_amd import ops, synth                                   # noqa: E402
from tests.util import fold_layers, mlp_layers, transformer_params   # noqa: E402


def timeit(fn, iters, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--batch", type=int, default=48)
    ap.add_argument("--pair-n", default="128,64", help="points per frame of the pair-kernel cases (stress: --batch 32 --pair-n 2048,64)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    B = a.batch
    want = lambda n: (not a.only) or any(n.startswith(o) for o in a.only.split(","))
    rs = np.random.RandomState(0)

    # ---- pair kernel ----
    for N in [int(v) for v in a.pair_n.split(",")]:
        name = "pair_N%d" % N
        if not want(name):
            continue
        P = {k: v.to(dev).contiguous() for k, v in transformer_params(1).items()}
        s, _ = synth.frames(1, B, N, 64, K_s=N)
        xyz = torch.from_numpy(s).to(dev)
        knn = ops.knn(xyz, 16)
        qkv = torch.randn(B, N, 1536, device=dev)
        packs = [ops.pack_weight(P[k]) for k in ("fc_delta.2.weight", "fc_gamma.0.weight", "fc_gamma.2.weight")]
        wd1p = ops.pack_delta0(P["fc_delta.0.weight"], P["fc_delta.0.bias"])
        # as TransformerBlock.forward launches it: clouds of >= 512 points in Morton order (PAIR_ORDER=none: sampling order)
        order = ops.spatial_order(xyz) if (N >= 512 and os.environ.get("PAIR_ORDER", "spatial") != "none") else None
        fn = lambda: ops.pt_attn_pair(xyz, knn, qkv, wd1p, packs[0],
                                      P["fc_delta.2.bias"], packs[1], P["fc_gamma.0.bias"], packs[2],
                                      P["fc_gamma.2.bias"], 512, False, order=order)
        ms = timeit(fn, a.iters)
        fl = 2.0 * B * N * 16 * (3 * 512 + 3 * 512 * 512)
        print("%-14s %8.4f ms  %7.2f TFLOP/s%s" % (name, ms, fl / ms / 1e9, "  (Morton order)" if order is not None else ""))

    # ---- SA levels ----
    sa_cases = [("sa0_s", 2048, 512, 0, [3, 64, 64, 128], 0.3, 32), ("sa1_s", 512, 256, 128, [131, 128, 128, 256], 0.5, 32),
                ("sa2_s", 256, 128, 256, [259, 128, 128, 256], 0.7, 32), ("sa_box", 128, 64, 257, [260, 256, 256, 256], 0.3, 16)]
    for name, N, M, C, spec, r, ns in sa_cases:
        if not want(name):
            continue
        s, _ = synth.frames(2, B, N, 64, K_s=max(64, int(N * 0.3)))
        xyz = torch.from_numpy(s).to(dev)
        new_xyz = xyz[:, :M].contiguous()
        idx = ops.ball_query(new_xyz, xyz, r, ns)
        feats = torch.randn(B, N, C, device=dev).transpose(1, 2) if C else None
        if name == "sa_box":
            feats = feats.contiguous()                     # channel-major, as the box head hands it over
        raw = mlp_layers(3, spec)
        layers = fold_layers(raw, dev, ops, scale_in_weights=True)
        fn = lambda: ops.sa_fused_forward(xyz, new_xyz, idx, feats, layers, r, True, True)
        ms = timeit(fn, a.iters)
        fl = 2.0 * B * M * ns * sum(ci * co for ci, co in zip(spec[:-1], spec[1:]))
        print("%-14s %8.4f ms  %7.2f TFLOP/s" % (name, ms, fl / ms / 1e9))
        if C:       # layer 0 hoisted (what the module runs): per-point linear + the two remaining layers
            w0 = raw[0]["conv_weight"].reshape(spec[1], spec[0]).to(dev)
            rows = feats.transpose(1, 2).contiguous()
            wf = ops.pack_weight(w0[:, 3:].contiguous())
            sc0 = fold_layers(raw[:1], dev, ops)[0][1]
            wx = (w0[:, 0:3] * sc0[:, None]).t().contiguous()
            lin = lambda: ops.linear(rows, wf, spec[1], sc0, layers[0][2], relu=False)
            term = lin()
            fn = lambda: ops.sa_fused_forward(xyz, new_xyz, idx, None, layers[1:], r, True, True, l0=(term, wx, True))
            ms_l, ms_h = timeit(lin, a.iters), timeit(fn, a.iters)
            fl_h = 2.0 * B * M * ns * sum(ci * co for ci, co in zip(spec[1:-1], spec[2:]))
            print("%-14s %8.4f ms  %7.2f TFLOP/s   (+ per-point linear %.4f ms)" % (name + "_hoist", ms_h, fl_h / ms_h / 1e9, ms_l))
        if os.environ.get("SWEEP_SA_STAGGER"):
            for sg in (1, 2, 3, 4, 6):
                os.environ["PTT_SA_STAGGER"] = str(sg)
                ms = timeit(fn, a.iters)
                print("%-14s %8.4f ms  %7.2f TFLOP/s" % (name + "_sg%d" % sg, ms, fl / ms / 1e9))
            os.environ.pop("PTT_SA_STAGGER")

    # ---- CosineSimAug core (N1): cosine map + fused layer chain, 128 search x 64 template seeds ----
    if want("xcorr"):
        raw = mlp_layers(5, [260, 256, 256, 256])
        layers = fold_layers(raw[1:], dev, ops, scale_in_weights=True)
        sf = torch.randn(B, 128, 256, device=dev).transpose(1, 2)
        tf = torch.randn(B, 64, 256, device=dev).transpose(1, 2)
        Pm = torch.randn(B, 64, 256, device=dev)
        wsim = torch.randn(256, device=dev)
        cos_t = ops.cosine_map(sf, tf)
        fn = lambda: ops.xcorr_fused(sf, tf, Pm, wsim, None, None, layers, cos_t=cos_t)
        ms = timeit(fn, a.iters)
        fl = 2.0 * B * 128 * 64 * 2 * 256 * 256
        print("%-14s %8.4f ms  %7.2f TFLOP/s   (cosine map %.4f ms)" % ("xcorr", ms, fl / ms / 1e9, timeit(lambda: ops.cosine_map(sf, tf), a.iters)))

    # ---- linear ----
    for name, rows, K, Cout in (("lin_hoist1", B * 512, 128, 128), ("lin_hoist2", B * 256, 256, 128), ("lin_qkvf", B * 128, 256, 1536), ("lin_qkvf64", B * 64, 256, 1536), ("lin_fc1", B * 128, 256, 512), ("lin_qkv", B * 128, 512, 1536), ("lin_fc2", B * 128, 512, 256),
                                ("lin_cov", B * 128, 256, 256), ("lin_qkv64", B * 64, 512, 1536), ("lin_fc2_64", B * 64, 512, 256)):
        if not want(name):
            continue
        x = torch.randn(rows, K, device=dev)
        w = ops.pack_weight(torch.randn(Cout, K, device=dev) / K ** 0.5)
        b = torch.randn(Cout, device=dev)
        fn = lambda: ops.linear(x, w, Cout, None, b)
        ms = timeit(fn, a.iters)
        print("%-14s %8.4f ms  %7.2f TFLOP/s" % (name, ms, 2.0 * rows * K * Cout / ms / 1e9))
        if os.environ.get("SWEEP_LINEAR"):
            for tile in ("11", "12", "21", "22"):
                os.environ["PTT_LINEAR_TILE"] = tile
                ms = timeit(fn, a.iters)
                print("   tile RT,CT=%s %8.4f ms  %7.2f TFLOP/s" % (tile, ms, 2.0 * rows * K * Cout / ms / 1e9))
            os.environ.pop("PTT_LINEAR_TILE")

    # ---- the row jobs of ONE tracklet frame (ptt_row_jobs_f32; independent of --batch) ----
    if want("rj"):
        D, N = 512, 128
        qkv = torch.randn((N, 3 * D), device=dev)
        knn1 = torch.stack([torch.randperm(N)[:16] for _ in range(N)]).to(torch.int32).to(dev)
        pos = torch.randn((N * 16, D), device=dev)
        rel = torch.randn((N * 16, 3), device=dev)
        w1 = torch.randn((D, 4), device=dev)
        wp = ops.pack_weight(torch.randn((D, D), device=dev) / 22.6)
        wq = ops.pack_weight(torch.randn((3 * D, 256), device=dev) / 16)
        b = torch.zeros(3 * D, device=dev)
        feats = torch.randn((N, 256), device=dev)
        g = torch.empty((N * 16, D), device=dev)
        res = torch.empty((N, D), device=dev)
        o = torch.empty((N, D), device=dev)
        q2 = torch.empty((N, 3 * D), device=dev)
        cases = (("rj_qkv_delta", lambda: ops.row_jobs([ops.row_job(wq, 3 * D, x=feats, shift=b, out=q2),
                                                         ops.row_job(wp, D, prologue=1, rel=rel, w1=w1, K=D, shift=b[:D], out=g)]),
                  2.0 * (N * 256 * 3 * D + N * 16 * D * D)),
                 ("rj_gamma0", lambda: ops.row_jobs([ops.row_job(wp, D, prologue=2, qkv=qkv, knn=knn1, pos=pos, k_off=D, N=N, K=D, shift=b[:D], act=1, out=g)]),
                  2.0 * N * 16 * D * D),
                 ("rj_gamma2", lambda: ops.row_jobs([ops.row_job(wp, D, x=g, epilogue=1, qkv=qkv, knn=knn1, pos=pos, v_off=2 * D, N=N, sm_scale=0.0442, out=res)]),
                  2.0 * N * 16 * D * D),
                 ("rj_plain", lambda: ops.row_jobs([ops.row_job(wp, D, x=res, shift=b[:D], out=o)]), 2.0 * N * D * D))
        for name, fn, fl in cases:
            ms = timeit(fn, a.iters)
            print("%-14s %8.4f ms  %7.2f TFLOP/s" % (name, ms, fl / ms / 1e9))

    # ---- FPS ----
    for name, N, m in (("fps_2048", 2048, 512), ("fps_1024", 1024, 256), ("fps_128", 128, 64)):
        if not want(name):
            continue
        s, _ = synth.frames(4, B, N, 64, K_s=max(64, int(N * 0.3)))
        xyz = torch.from_numpy(s).to(dev)
        ms = timeit(lambda: ops.furthest_point_sampling(xyz, m), a.iters)
        print("%-14s %8.4f ms  %7.3f us/iteration" % (name, ms, ms * 1e3 / (m - 1)))
        if N == 2048:
            for T in (128, 512, 1024):
                os.environ["PTT_FPS_T"] = str(T)
                ms = timeit(lambda: ops.furthest_point_sampling(xyz, m), a.iters)
                print("%-14s %8.4f ms  %7.3f us/iteration" % (name + "_T%d" % T, ms, ms * 1e3 / (m - 1)))
            os.environ.pop("PTT_FPS_T")


if __name__ == "__main__":
    main()
