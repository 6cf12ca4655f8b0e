"""Dev probe: run-to-run determinism of the whole tracker (fused and unfused eval paths)."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from ptt_amd import synth
from ptt_amd.config import StubDataset, ptt_model_cfg
from ptt_amd.models import build_network
from ptt_amd.hot_path import randomize_
dev = torch.device('cuda:0')
KEYS = ('search_seeds', 'search_feats', 'cosine_feats', 'pred_centroids_cls', 'pred_centroids_votes', 'pred_box_center', 'pred_box_data')
for variant in ('p2b', 'ptt'):
    for seed in range(6):
        cfg = ptt_model_cfg()
        if variant == 'p2b':
            cfg.BACKBONE_3D.SA_CONFIG.SAMPLE_METHOD = ['sequence'] * 3
            cfg.CENTROID_HEAD.TRANSFORMER_BLOCK.ENABLE = False
            cfg.BOX_HEAD.TRANSFORMER_BLOCK.ENABLE = False
        model = randomize_(build_network(cfg, 1, StubDataset()), seed=11 + seed).to(dev).eval()
        s, t = synth.frames(21 + seed, 3, 1024, 512)
        def run():
            b = {'search_points': torch.from_numpy(s).to(dev), 'template_points': torch.from_numpy(t).to(dev), 'batch_size': 3}
            with torch.no_grad():
                o = model(b)
            return {k: o[k].clone() for k in KEYS}
        junk = torch.full((64 << 20,), float('nan'), device=dev); del junk      # poison the allocator cache
        f1 = run(); 
        junk = torch.full((64 << 20,), 1e30, device=dev); del junk
        f2 = run()
        for m in model.modules():
            if hasattr(m, '_fusable'):
                m._fusable = lambda *a, **k: False
        p1 = run(); p2 = run()
        d = lambda a, b: max(float((a[k] - b[k]).abs().max()) for k in KEYS)
        worst = max(KEYS, key=lambda k: float((f1[k] - p1[k]).abs().max()))
        print(variant, seed, 'fused-fused %.3g plain-plain %.3g fused-plain %.3g (%s)' % (d(f1, f2), d(p1, p2), d(f1, p1), worst), flush=True)
