"""Fixture G15 — the REFERENCE model's training step at BASELINE.json configs[3]'s own sparsity (run in the build container
against /root/reference, like make_golden.py; only arrays are committed).

G10 / G14 are KITTI-shaped (600 / 300 unique points per cloud, B = 3). configs[3] is the nuScenes-Car DDP step:
tools/cfgs/nuscenes_models/ptt.yaml (MODEL section functionally identical to KITTI's) on clouds with K_s = 200 /
K_t = 100 unique points — heavy duplication, most balls under-filled. G15 = loss, the norm of every
parameter gradient and eight full gradients of the reference tracker built from THAT yaml, on
ptt_amd.train_step.synthetic_train_batch(1515, 4) (the generator bench.py's train workload uses), with the float32 index ops
of the oracle behind `pointnet2_ops._ext` exactly as for G10.

    python tests/golden/make_golden_g15.py        # writes tests/golden/G15_train_step_nuscenes.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tests.golden import make_golden as MG          # noqa: E402
from tests.util import fill_state_dict_             # noqa: E402

SEED, B = 1515, 4


def main():
    EasyDict = MG._install_stubs()
    sys.path.insert(0, MG.REF)
    from ptt.config import cfg_from_yaml_file as ref_cfg_from_yaml
    from ptt.models import build_network as ref_build_network
    from ptt_amd.config import StubDataset
    from ptt_amd.train_step import synthetic_train_batch
    rcfg = ref_cfg_from_yaml(os.path.join(MG.REF, "tools/cfgs/nuscenes_models/ptt.yaml"), EasyDict())
    model = fill_state_dict_(ref_build_network(rcfg.MODEL, 1, StubDataset(training=True)), SEED).train()
    batch = synthetic_train_batch(SEED, B, "cpu")                      # K_s = 200, K_t = 100, 1024 + 512 points
    ret, _, _ = model(dict(batch))
    loss = ret['loss'].mean()
    loss.backward()
    named = dict(model.named_parameters())
    gkeys = sorted(k for k, p in named.items() if p.grad is not None)
    full = ['backbone_3d.SA_modules.0.mlp_module.layer0.conv.weight', 'backbone_3d.SA_modules.2.mlp_module.layer2.conv.weight',
            'backbone_3d.cov_final.bias', 'centroid_voting_head.transformer_block.fc_delta.0.weight',
            'centroid_voting_head.transformer_block.w_ks.weight', 'box_voting_head.transformer_block.fc_gamma.2.bias',
            'similarity_module.mlp.layer0.conv.weight', 'box_voting_head.refine_layer.2.conv.weight']
    full = [k for k in full if k in named and named[k].grad is not None]
    uniq = [int(len(np.unique(batch['search_points'][b].numpy(), axis=0))) for b in range(B)]
    np.savez_compressed(os.path.join(HERE, "G15_train_step_nuscenes.npz"), seed=SEED, batch=B, loss=np.float64(loss.item()),
                        search=batch['search_points'].numpy(), template=batch['template_points'].numpy(),
                        cls_label=batch['cls_label'].numpy(), reg_label=batch['reg_label'].numpy(),
                        grad_keys=np.array(gkeys), grad_norms=np.array([float(named[k].grad.double().norm()) for k in gkeys]),
                        full_keys=np.array(full), unique_search_points=np.array(uniq),
                        **{"grad_%d" % i: named[k].grad.numpy() for i, k in enumerate(full)})
    line = ("G15 nuScenes-shaped training step written: cfg tools/cfgs/nuscenes_models/ptt.yaml, B = %d, unique search points %s, "
            "loss %.6f, %d parameter gradients (%d in full)" % (B, uniq, loss.item(), len(gkeys), len(full)))
    print(line)
    rep = os.path.join(HERE, "GOLDEN_REPORT.txt")
    old = [l for l in open(rep).read().splitlines() if not l.startswith("G15 ")] if os.path.exists(rep) else []
    with open(rep, "w") as fh:
        fh.write("\n".join(old + [line]) + "\n")


if __name__ == "__main__":
    main()
