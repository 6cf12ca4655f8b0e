"""Diagnostic: the full tracker trained in lockstep twice — BatchNorm backward applied by the input-gradient GEMM (A) and by the apply
pass (B) — with the same stock optimiser: per step the parameters and gradients that differ most."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ptt_amd import train_ops
from ptt_amd.config import StubDataset, ptt_model_cfg
from ptt_amd.models import build_network
from ptt_amd.train_step import synthetic_train_batch
dev = torch.device("cuda:0")
models, opts = [], []
for k in range(2):
    torch.manual_seed(11)
    m = build_network(ptt_model_cfg(), 1, StubDataset(training=True)).to(dev).train()
    models.append(m); opts.append(torch.optim.Adam(m.parameters(), lr=1e-3, betas=(0.5, 0.999), eps=1e-6))
for step in range(3):
    models[0].load_state_dict(models[1].state_dict())          # identical weights: the forward passes are then bit-identical
    losses = []
    for k, (m, o) in enumerate(zip(models, opts)):
        train_ops.FUSED_BN_BWD = k == 0
        ret, _, _ = m(dict(synthetic_train_batch(20 + step, 4, dev)))
        o.zero_grad(set_to_none=True)
        ret['loss'].backward()
        losses.append(float(ret['loss'].detach()))
    rows = []
    for (n, p), (_, q) in zip(models[0].named_parameters(), models[1].named_parameters()):
        if p.grad is None: continue
        gd = float((p.grad - q.grad).abs().max()) / (float(q.grad.abs().max()) + 1e-30)
        pd = float((p - q).abs().max()) / (float(q.abs().max()) + 1e-30)
        rows.append((gd, pd, float(q.grad.abs().max()), n))
    rows.sort(reverse=True)
    print("step %d losses %s; largest gradient differences (rel diff, param rel diff, |grad|max, name):" % (step, ["%.6f" % v for v in losses]))
    for r in rows[:8]:
        print("    %.2e %.2e %.2e %s" % r)
    torch.nn.utils.clip_grad_norm_(models[1].parameters(), 10.0)
    opts[1].step()
