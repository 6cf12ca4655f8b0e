"""Fixture G14 — the REFERENCE model's training step of fixture G10 evaluated in FLOAT64 (run in the build container
against /root/reference, like make_golden.py; only arrays are committed).

Why: G10 holds the reference's float32 loss and gradients. The tracker is full of max-pools and ReLUs whose gradient
routing flips under 1e-6 perturbations, so two correct float32 implementations differ by per cent in some gradients. G14
gives the yardstick: the same reference code, same weights and inputs, with every dense operation in float64 (sample /
neighbour INDICES still come from the float32 index ops, as in G10, so the discrete structure is identical). The GPU test
then measures both `ours(float32) vs float64` and `reference(float32, G10) vs float64` and requires ours to be no further
from the float64 gradient than a small multiple of the reference's own float32 evaluation is.

    python tests/golden/make_golden_f64.py        # writes tests/golden/G14_train_step_f64.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tests.golden import make_golden as MG          # noqa: E402
from oracle import index_ops as O                   # noqa: E402
from tests.util import fill_state_dict_             # noqa: E402


def main():
    EasyDict = MG._install_stubs()
    ext = sys.modules["pointnet2_ops._ext"]
    f32 = lambda x: np.ascontiguousarray(x.detach().numpy().astype(np.float32))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    # indices from the float32 index ops (exactly G10's); value ops dtype-preserving (float64 flows through)
    picks = {}

    def fps(xyz, n):
        idx = O.fps(f32(xyz), n)
        picks[(xyz.shape[1], n)] = idx
        return t(idx)
    ext.furthest_point_sampling = fps
    ext.ball_query = lambda new_xyz, xyz, r, ns: t(O.ball_query(f32(new_xyz), f32(xyz), r, ns))
    ext.gather_points = lambda f, i: torch.gather(f, 2, i.long()[:, None, :].expand(-1, f.shape[1], -1))
    ext.gather_points_grad = lambda g, i, n: torch.zeros(g.shape[0], g.shape[1], n, dtype=g.dtype).scatter_add_(
        2, i.long()[:, None, :].expand(-1, g.shape[1], -1), g)
    ext.group_points = lambda f, i: torch.gather(f, 2, i.long().reshape(i.shape[0], 1, -1).expand(-1, f.shape[1], -1)).reshape(
        f.shape[0], f.shape[1], i.shape[1], i.shape[2]).clone()
    ext.group_points_grad = lambda g, i, n: torch.zeros(g.shape[0], g.shape[1], n, dtype=g.dtype).scatter_add_(
        2, i.long().reshape(i.shape[0], 1, -1).expand(-1, g.shape[1], -1), g.reshape(g.shape[0], g.shape[1], -1))
    sys.path.insert(0, MG.REF)
    from ptt.config import cfg_from_yaml_file as ref_cfg_from_yaml
    from ptt.models import build_network as ref_build_network
    from ptt_amd.config import StubDataset
    g10 = np.load(os.path.join(HERE, "G10_train_step.npz"))
    rcfg = ref_cfg_from_yaml(os.path.join(MG.REF, "tools/cfgs/kitti_models/ptt.yaml"), EasyDict())
    model = fill_state_dict_(ref_build_network(rcfg.MODEL, 1, StubDataset(training=True)), int(g10["seed"])).double().train()
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).double()
    ret, _, _ = model({'search_points': d(g10["search"]), 'template_points': d(g10["template"]), 'batch_size': 3,
                       'cls_label': d(g10["cls_label"]), 'reg_label': d(g10["reg_label"])})
    loss = ret['loss'].mean()
    loss.backward()
    named = dict(model.named_parameters())
    keys = [str(k) for k in g10["grad_keys"]]
    assert sorted(k for k, p in named.items() if p.grad is not None) == keys
    # norms of all gradients; in full (float32 storage of the float64 values) those of G10's full_keys and every tensor of
    # at most 20 000 elements (BatchNorm parameters, biases, the backbone's convolutions): 0.9 M numbers
    full = [k for k in keys if named[k].numel() <= 20000 or k in set(str(x) for x in g10["full_keys"])]
    # the one DATA-DEPENDENT sampling of the step: FPS of the 128 predicted votes down to 64 proposals (box_voting_head.py:75-79).
    # A vote that moves by 1e-7 can flip a pick, and a different proposal set is a different (equally valid) gradient: the
    # GPU test feeds these picks to its own run so that what it compares is arithmetic, not a coin toss.
    vote_picks = picks[(128, 64)]
    # the float32 reference run (G10) on the same stubs: did it pick the same proposals?
    rcfg2 = ref_cfg_from_yaml(os.path.join(MG.REF, "tools/cfgs/kitti_models/ptt.yaml"), EasyDict())   # the constructor mutates MLPS
    m32 = fill_state_dict_(ref_build_network(rcfg2.MODEL, 1, StubDataset(training=True)), int(g10["seed"])).train()
    f = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    with torch.no_grad():
        m32({'search_points': f(g10["search"]), 'template_points': f(g10["template"]), 'batch_size': 3,
             'cls_label': f(g10["cls_label"]), 'reg_label': f(g10["reg_label"])})
    same32 = bool(np.array_equal(picks[(128, 64)], vote_picks))
    np.savez_compressed(os.path.join(HERE, "G14_train_step_f64.npz"), loss=np.float64(loss.item()), grad_keys=np.array(keys),
                        vote_picks=vote_picks, ref32_same_picks=np.array(same32),
                        grad_norms=np.array([float(named[k].grad.norm()) for k in keys]), full_keys=np.array(full),
                        **{"grad_%d" % i: named[k].grad.numpy().astype(np.float32) for i, k in enumerate(full)})
    ref32 = dict(zip(keys, g10["grad_norms"]))
    worst = max(abs(float(named[k].grad.norm()) - ref32[k]) / ref32[k] for k in keys if ref32[k] > 1e-3)
    line = ("G14 reference training step in float64 written: loss %.9f (float32 run: %.9f); the reference's own float32 gradient "
            "norms (G10) are within %.4f of the float64 ones; the float32 run picks %s 64 proposals out of the 128 votes"
            % (loss.item(), float(g10["loss"]), worst, "the same" if same32 else "DIFFERENT"))
    print(line)
    rep = os.path.join(HERE, "GOLDEN_REPORT.txt")
    lines = [l for l in open(rep).read().splitlines() if not l.startswith("G14 ")]
    open(rep, "w").write("\n".join(lines + [line]) + "\n")


if __name__ == "__main__":
    main()
