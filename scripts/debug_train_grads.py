import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptt_amd import train_ops
from ptt_amd.config import StubDataset, ptt_model_cfg
from ptt_amd.models import build_network
from tests.util import fill_state_dict_
dev = torch.device("cuda:0")
g = np.load("tests/golden/G10_train_step.npz")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
def run(flag_fn):
    train_ops.usable_orig = getattr(train_ops, "usable_orig", train_ops.usable)
    train_ops.usable = flag_fn
    import ptt_amd.models.backbones_3d.pointnet2.pointnet2_modules as pm
    model = fill_state_dict_(build_network(ptt_model_cfg(), 1, StubDataset(training=True)), int(g["seed"])).to(dev).train()
    ret, _, _ = model({'search_points': t(g["search"]), 'template_points': t(g["template"]), 'batch_size': 3,
                       'cls_label': t(g["cls_label"]), 'reg_label': t(g["reg_label"])})
    loss = ret['loss'].mean(); loss.backward()
    return float(loss), {k: p.grad.double().cpu() for k, p in model.named_parameters() if p.grad is not None}
orig = train_ops.usable
l0, g0 = run(lambda m, x: False)
gk = [str(k) for k in g["grad_keys"]]
refn = dict(zip(gk, g["grad_norms"]))
big = [k for k in gk if refn[k] > 1e-3]
def vs_ref(gr):
    return max(abs(float(gr[k].norm()) - refn[k]) / refn[k] for k in big)
print("stock path vs reference norms:", vs_ref(g0))
def only(pred):
    return lambda m, x: orig(m, x) and pred(m, x)
for name, pred in [("all", lambda m, x: True), ("C0==3", lambda m, x: x.shape[1] == 3), ("C0==131", lambda m, x: x.shape[1] == 131),
                   ("C0==259", lambda m, x: x.shape[1] == 259), ("C0==260 box", lambda m, x: x.shape[1] == 260 and x.shape[3] == 16),
                   ("C0==260 xcorr", lambda m, x: x.shape[1] == 260 and x.shape[3] != 16)]:
    l1, g1 = run(only(pred))
    errs = sorted(((float((g1[k] - g0[k]).norm() / (g0[k].norm() + 1e-30)), k) for k in big), reverse=True)
    print(name, "loss", l0, l1, "vs ref norms", round(vs_ref(g1), 4), "worst vs stock:", [(round(e, 4), k) for e, k in errs[:3]])
