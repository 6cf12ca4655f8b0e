"""Dev tool: where does the float32 training FORWARD lose accuracy? The mirrored tracker in float64 on the CPU (stock torch
ops, oracle index ops) against the float32 GPU run (row kernels, and stock torch ops on the GPU): relative L2 error of every
tensor the modules hand over (batch_dict entries) on fixture G10's inputs."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ptt_amd.config import StubDataset, ptt_model_cfg
from ptt_amd.models import build_network
from ptt_amd import ops, train_ops
from oracle import index_ops as O
from tests.util import fill_state_dict_
g10 = np.load(os.path.join(ROOT, "tests", "golden", "G10_train_step.npz"))
KEYS = ('search_feats', 'template_feats', 'cosine_feats', 'pred_centroids_cls', 'pred_centroids_votes', 'votes_feats', 'pred_box_center',
        'pred_box_data')


def run(dev, dtype):
    m = fill_state_dict_(build_network(ptt_model_cfg(), 1, StubDataset(training=True)), int(g10["seed"])).to(dev).train()
    if dtype == torch.float64:
        m = m.double()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(dtype)
    b = {'search_points': t(g10["search"]), 'template_points': t(g10["template"]), 'batch_size': 3, 'cls_label': t(g10["cls_label"]),
         'reg_label': t(g10["reg_label"])}
    cap = {}
    sm = m.similarity_module
    h = [sm.conv.register_forward_pre_hook(lambda mod, inp: cap.__setitem__('xcorr_pooled (conv input)', inp[0].detach().double().cpu())),
         sm.conv[0].conv.register_forward_hook(lambda mod, inp, o: cap.__setitem__('conv0 pre-BN', o.detach().double().cpu())),
         sm.conv[0].register_forward_hook(lambda mod, inp, o: cap.__setitem__('conv0 post-BN-ReLU', o.detach().double().cpu())),
         sm.mlp.register_forward_hook(lambda mod, inp, o: cap.__setitem__('xcorr mlp out (stock only)', o.detach().double().cpu())),
         sm.mlp[0].conv.register_forward_hook(lambda mod, inp, o: cap.__setitem__('xcorr z0 (stock only)', o.detach().double().cpu())),
         sm.cosine.register_forward_hook(lambda mod, inp, o: cap.__setitem__('cosine map (stock only)', o.detach().double().cpu()))]
    orig = train_ops.conv1d_stack_rows

    def spy(seq, rows, residual=None):
        if seq is sm.conv:
            cap['xcorr_pooled (conv input)'] = rows.transpose(1, 2).detach().double().cpu()
        return orig(seq, rows, residual)
    train_ops.conv1d_stack_rows = spy
    ret, _, _ = m(b)
    train_ops.conv1d_stack_rows = orig
    for x in h:
        x.remove()
    out = {k: b[k].detach().double().cpu() for k in KEYS if k in b}
    out.update(cap)
    out['loss'] = ret['loss'].mean().detach().double().cpu()
    return out


saved = {k: getattr(ops, k) for k in ("furthest_point_sampling", "gather_points", "gather_points_grad", "ball_query", "group_points", "group_points_grad")}
f32 = lambda x: np.ascontiguousarray(x.detach().numpy().astype(np.float32))
tt = lambda a: torch.from_numpy(np.ascontiguousarray(a))
ops.furthest_point_sampling = lambda xyz, n: tt(O.fps(f32(xyz), n))
ops.ball_query = lambda new_xyz, xyz, r, ns: tt(O.ball_query(f32(new_xyz), f32(xyz), r, ns))
ops.gather_points = lambda f, i: torch.gather(f, 2, i.long()[:, None, :].expand(-1, f.shape[1], -1))
ops.group_points = lambda f, i: torch.gather(f, 2, i.long().reshape(i.shape[0], 1, -1).expand(-1, f.shape[1], -1)).reshape(
    f.shape[0], f.shape[1], i.shape[1], i.shape[2]).clone()
ref = run("cpu", torch.float64)
cpu32 = run("cpu", torch.float32)
for k, v in saved.items():
    setattr(ops, k, v)
gpu_rows = run("cuda:0", torch.float32)
u0, p0, c0 = train_ops.usable, train_ops.pt_block_usable, train_ops.conv1d_stack_usable
train_ops.usable = lambda *a: False; train_ops.pt_block_usable = lambda *a: False; train_ops.conv1d_stack_usable = lambda *a: False
gpu_stock = run("cuda:0", torch.float32)
train_ops.usable, train_ops.pt_block_usable, train_ops.conv1d_stack_usable = u0, p0, c0
print("%-24s %12s %12s %12s" % ("tensor (rel. L2 vs f64)", "cpu f32", "gpu stock", "gpu rows"))
for k in list(KEYS) + ['loss'] + sorted(k for k in ref if k not in KEYS and k != 'loss'):
    if k not in ref:
        continue
    e = lambda o: float((o[k].flatten() - ref[k].flatten()).norm() / ref[k].flatten().norm()) if k in o else float('nan')
    print("%-30s %12.2e %12.2e %12.2e" % (k, e(cpu32), e(gpu_stock), e(gpu_rows)))
x = ref['xcorr_pooled (conv input)']          # (B,C,n2): how much of each channel varies across the search points?
print("xcorr_pooled: median over channels of std across (b, j) / mean |value| = %.3e" % float((x.std(dim=(0, 2)) / x.abs().mean(dim=(0, 2))).median()))
