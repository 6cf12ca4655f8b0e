"""Dev tool: where does the float32 training step differ from the float64 gradient (fixture G14)? Per parameter:
relative L2 error of (a) this build's float32 step on the row kernels, (b) this build's model in float64 on stock torch
ops (isolates logic from arithmetic), (c) the reference's float32 norms (G10), all against G14."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ptt_amd.config import StubDataset, ptt_model_cfg
from ptt_amd.models import build_network
from tests.util import fill_state_dict_
G = os.path.join(ROOT, "tests", "golden")
g10, g14 = np.load(os.path.join(G, "G10_train_step.npz")), np.load(os.path.join(G, "G14_train_step_f64.npz"))
dev = torch.device("cuda:0")


from ptt_amd.models.backbones_3d.pointnet2 import pointnet2_utils as PU
_fps = PU.furthest_point_sample
FORCE = os.environ.get("G14_FORCE_PICKS", "1") == "1"


def fps_forced(xyz, npoint):
    if FORCE and xyz.shape[1] == 128 and npoint == 64:
        mine = _fps(xyz, npoint)
        want = torch.from_numpy(g14["vote_picks"]).to(mine.device).to(mine.dtype)
        ms, ws = [set(r.tolist()) for r in mine.cpu()], [set(r.tolist()) for r in want.cpu()]
        print("vote FPS: %d of %d picks differ from the float64 reference run; per frame, proposals not in the reference's set: %s"
              % (int((mine != want).sum()), want.numel(), [sorted(a - b) for a, b in zip(ms, ws)]))
        return want
    return _fps(xyz, npoint)


PU.furthest_point_sample = fps_forced


def run(dtype):
    m = fill_state_dict_(build_network(ptt_model_cfg(), 1, StubDataset(training=True)), int(g10["seed"])).to(dev).train()
    if dtype == torch.float64:
        m = m.double()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(dtype)
    ret, _, _ = m({'search_points': t(g10["search"]), 'template_points': t(g10["template"]), 'batch_size': 3,
                   'cls_label': t(g10["cls_label"]), 'reg_label': t(g10["reg_label"])})
    loss = ret['loss'].mean()
    loss.backward()
    return float(loss), {k: p.grad.detach().double().cpu() for k, p in m.named_parameters() if p.grad is not None}


from ptt_amd import train_ops
mode = os.environ.get("G14_PATH", "rows")
if mode in ("stock", "stock_tb", "stock_heads", "stock_mlp"):
    if mode in ("stock", "stock_mlp"):
        train_ops.usable = lambda *a: False
    if mode in ("stock", "stock_tb"):
        train_ops.pt_block_usable = lambda *a: False
    if mode in ("stock", "stock_heads"):
        train_ops.conv1d_stack_usable = lambda *a: False
print("path:", mode)
keys = [str(k) for k in g14["grad_keys"]]
n64 = dict(zip(keys, g14["grad_norms"]))
n32ref = dict(zip([str(k) for k in g10["grad_keys"]], g10["grad_norms"]))
full = {str(k): torch.from_numpy(g14["grad_%d" % i]).double() for i, k in enumerate(g14["full_keys"])}
l32, ours32 = run(torch.float32)
try:
    l64, ours64 = run(torch.float64)
except Exception as e:
    print("float64 run failed:", type(e).__name__, e); l64, ours64 = None, None
print("loss: G14 %.9f ours32 %.9f ours64 %s ref32 %.9f" % (float(g14["loss"]), l32, l64, float(g10["loss"])))
rows = []
for k in keys:
    if n64[k] <= 1e-3:
        continue
    e_norm32 = abs(float(ours32[k].norm()) - n64[k]) / n64[k]
    e_ref = abs(n32ref[k] - n64[k]) / n64[k]
    e_norm64 = abs(float(ours64[k].norm()) - n64[k]) / n64[k] if ours64 else float('nan')
    l2 = float((ours32[k].flatten() - full[k].flatten()).norm() / full[k].norm()) if k in full else float('nan')
    l2_64 = float((ours64[k].flatten() - full[k].flatten()).norm() / full[k].norm()) if (k in full and ours64) else float('nan')
    rows.append((e_norm32, e_ref, e_norm64, l2, l2_64, k))
rows.sort(reverse=True)
print("norm err ours32 | ref32 | ours64 | L2 ours32 | L2 ours64 | key")
for r in rows[:int(os.environ.get("G14_ROWS", "6"))]:
    print("%.4f | %.4f | %.2e | %.4f | %.2e | %s" % r)
groups = {}
for r in rows:
    k = r[5]
    gk = ".".join(k.split(".")[:3]) if k.startswith(("centroid", "box")) else ".".join(k.split(".")[:2])
    groups.setdefault(gk, []).append(r[0])
for gk, v in sorted(groups.items()):
    print("   group %-55s n=%2d  norm error max %.4f median %.4f" % (gk, len(v), max(v), sorted(v)[len(v) // 2]))
print("worst ours32 %.4f, worst ref32 %.4f, worst ours64 %.2e" % (max(r[0] for r in rows), max(r[1] for r in rows), max(r[2] for r in rows)))
