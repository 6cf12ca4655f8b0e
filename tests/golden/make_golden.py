#!/usr/bin/env python
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE's own pure-PyTorch
modules (imported from /root/reference, in the build container only) on seeded inputs.

What travels to the GPU box is data only: inputs, expected outputs and the numpy seeds the weights
are regenerated from (tests/util.py). Nothing of the reference's source is copied.

The reference cannot run without its third-party CUDA extension `pointnet2_ops._ext`
(pointnet2_utils.py:24), `thop` (variants.py:7) and `easydict` (config.py); this script injects
in-process stand-ins for exactly those three imports: `_ext` is backed by the repo's CPU oracle
(oracle/index_ops.py), so every fixture pins  reference glue + torch layers  ON TOP OF  the oracle's
index ops; the index ops themselves have no reference implementation to compare with ("parity
unpinned", see oracle/ptt_oracle.c) — their golden vectors (G7) are authored by this build.

Fixtures:
  G1 query_and_group.npz   QueryAndGroup layout / centre subtraction / normalisation
  G2 shared_mlp.npz        SharedMLP eval with non-trivial BN running stats
  G3 sa_module.npz         PointnetSAModuleVotes: fps / sequence / caller-supplied inds, int64 cast, order
  G4 backbone_branch.npz   PointNet2BackboneLight.branch_forward incl. index composition
  G5 transformer.npz       TransformerBlock res + attn samples, N=128 and N=64, duplicated points
  G7 index_ops.npz         op-level edge cases (duplicates, zero cloud, origin ball, under-filled balls)
  G8 knn_argsort.npz       kNN vs the reference's square_distance + argsort on tie-free inputs
  G9 cosine_sim_aug.npz    CosineSimAug (N1): cosine map samples + cosine_feats
  G6 ptt_forward.npz       full PTT.forward (eval) through the reference's own heads (N2) + state_dict key/shape list
  G10 train_step.npz       one training forward + backward of the full tracker (N3): loss, gradient norms, 8 full gradients
  G13 transformer_std.npz  TransformerBlockSTD (the dense Q.K^T / attn.V variant no shipped config selects), N = 128 / 64 / 50
  G12 tracking_pre_post.npz  N4: the reference's crop_center_pc / get_model / regularize_pc / get_box_by_offset
                           (ptt/datasets/kitti/kitti_tracking_utils.py:186-367) on a synthetic 6-frame tracklet and on
                           edge cases (empty crop, n <= 2, n == input_size, the redraw branch of get_box_by_offset).
                           The reference imports `pyquaternion`, absent from this image: a stand-in restating its
                           published formulas is injected (oracle.tracking_ref._Quat); rotation matrices travel as data.
  G11 fps_reference.npz    the reference's OWN numpy farthest-point sampling (ptt/utils/common_utils.py:78-112,
                           `fps_downsample`) on origin-free clouds incl. duplicated points and exact distance ties:
                           the one reference-held statement of FPS (start point, min-update, np.argmax = lowest index
                           among ties). It has no origin-ball skip (that is upstream CUDA behaviour), hence origin-free.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from oracle import dense_ref as R           # noqa: E402
from oracle import index_ops as O           # noqa: E402
from ptt_amd import synth                   # noqa: E402
from tests.util import (cosine_sim_params, fill_state_dict_, load_cosine_sim, mlp_layers,   # noqa: E402
                        transformer_params)


def _install_stubs():
    ext = types.ModuleType("pointnet2_ops._ext")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    ext.furthest_point_sampling = lambda xyz, n: t(O.fps(xyz.numpy(), n))
    ext.gather_points = lambda f, i: t(O.gather(f.numpy(), i.numpy()))
    ext.gather_points_grad = lambda g, i, n: t(O.gather_grad(g.numpy(), i.numpy(), n))
    ext.ball_query = lambda new_xyz, xyz, r, ns: t(O.ball_query(new_xyz.numpy(), xyz.numpy(), r, ns))
    ext.group_points = lambda f, i: t(O.group(f.numpy(), i.numpy()))
    ext.group_points_grad = lambda g, i, n: t(O.group_grad(g.numpy(), i.numpy(), n))
    pkg = types.ModuleType("pointnet2_ops")
    pkg._ext = ext
    sys.modules["pointnet2_ops"] = pkg
    sys.modules["pointnet2_ops._ext"] = ext
    thop = types.ModuleType("thop")
    thop.profile = lambda *a, **k: (0, 0)
    thop.clever_format = lambda *a, **k: ("0", "0")
    sys.modules["thop"] = thop

    class EasyDict(dict):
        def __init__(self, d=None, **kw):
            super().__init__()
            for k, v in dict(d or {}, **kw).items():
                self[k] = v

        def __setitem__(self, k, v):
            if isinstance(v, dict) and not isinstance(v, EasyDict):
                v = EasyDict(v)
            super().__setitem__(k, v)

        __setattr__ = __setitem__

        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)

    # pyquaternion (imported by ptt/datasets/kitti/kitti_tracking_utils.py:5) is not installed: a stand-in with the
    # constructor forms and operations the reference uses, on the formulas pyquaternion documents
    from oracle.tracking_ref import _Quat

    class Quaternion(object):
        def __init__(self, *a, **kw):
            if 'matrix' in kw:
                self._q = _Quat.from_matrix(kw['matrix'])
            elif 'axis' in kw:
                self._q = _Quat.from_axis_angle(kw['axis'], kw['angle'] if 'angle' in kw else kw['radians'])
            elif 'array' in kw:
                self._q = _Quat(kw['array'])
            else:
                self._q = _Quat(a if len(a) == 4 else a[0])

        def __mul__(self, other):
            return Quaternion(array=self._q.mul(other._q).q)

        inverse = property(lambda self: Quaternion(array=self._q.inverse.q))
        rotation_matrix = property(lambda self: self._q.rotation_matrix)
        elements = property(lambda self: self._q.q)

    pq = types.ModuleType("pyquaternion")
    pq.Quaternion = Quaternion
    sys.modules["pyquaternion"] = pq

    ed = types.ModuleType("easydict")
    ed.EasyDict = EasyDict
    sys.modules["easydict"] = ed
    # the reference hard-codes .cuda() in hot-path code (pointnet2_modules.py:69,71): identity on CPU
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    return EasyDict


def _load_mlp(ref_mlp, layers):
    with torch.no_grad():
        for unit, L in zip(ref_mlp, layers):
            unit.conv.weight.copy_(L["conv_weight"])
            bn = unit.normlayer.bn
            bn.weight.copy_(L["bn_weight"]); bn.bias.copy_(L["bn_bias"])
            bn.running_mean.copy_(L["bn_mean"]); bn.running_var.copy_(L["bn_var"])


def main():
    EasyDict = _install_stubs()
    sys.path.insert(0, REF)
    from ptt.models.backbones_3d.pointnet2 import pointnet2_modules as ref_mod
    from ptt.models.backbones_3d.pointnet2 import pointnet2_utils as ref_utils
    from ptt.models.backbones_3d.pointnet2 import pytorch_utils as ref_pt
    from ptt.models.backbones_3d.pointnet2_backbone import PointNet2BackboneLight as RefBackbone
    from ptt.models.model_utils import square_distance as ref_sqdist
    from ptt.models.transformer_block.variants import TransformerBlock as RefTB

    torch.manual_seed(0)
    save = lambda name, **kw: np.savez_compressed(os.path.join(HERE, name), **kw)
    report = []

    # ---------------- G1 QueryAndGroup ----------------
    rs = np.random.RandomState(101)
    s, _ = synth.frames(101, 2, 256, 64, K_s=120)
    xyz = torch.from_numpy(s)
    inds = torch.from_numpy(O.fps(s, 64))
    new_xyz = torch.gather(xyz, 1, inds.long()[..., None].expand(-1, -1, 3)).contiguous()
    feats = torch.from_numpy(rs.standard_normal((2, 5, 256)).astype(np.float32))
    g = ref_utils.QueryAndGroup(0.5, 16, use_xyz=True, ret_grouped_xyz=True, normalize_xyz=True)
    nf, gx, idx = g(xyz, new_xyz, feats, return_idx=True)
    mine_nf, mine_gx, mine_idx = R.query_and_group(xyz, new_xyz, feats, 0.5, 16, True, True)
    assert torch.equal(nf, mine_nf) and torch.equal(gx, mine_gx) and torch.equal(idx, mine_idx)
    save("G1_query_and_group.npz", xyz=s, new_xyz=new_xyz.numpy(), feats=feats.numpy(), radius=0.5, nsample=16,
         new_features=nf.numpy(), grouped_xyz=gx.numpy(), idx=idx.numpy())
    report.append("G1 QueryAndGroup: oracle == reference bitwise")

    # ---------------- G2 SharedMLP eval ----------------
    spec = [8, 32, 32, 64]
    layers = mlp_layers(202, spec)
    m = ref_pt.SharedMLP(list(spec), bn=True).eval()
    _load_mlp(m, layers)
    x = torch.from_numpy(np.random.RandomState(202).standard_normal((2, 8, 24, 16)).astype(np.float32))
    with torch.no_grad():
        y = m(x)
    mine = R.shared_mlp_eval(x, layers)
    assert torch.allclose(y, mine, atol=1e-6, rtol=1e-6), float((y - mine).abs().max())
    save("G2_shared_mlp.npz", x=x.numpy(), y=y.numpy(), spec=np.array(spec), seed=202)
    report.append("G2 SharedMLP eval: max |oracle - reference| = %.2e" % float((y - mine).abs().max()))

    # ---------------- G3 PointnetSAModuleVotes ----------------
    out = {}
    for tag, method, use_inds in (("fps", "fps", False), ("seq", "sequence", False), ("inds", "fps", True)):
        spec3 = [5, 32, 32, 64]
        layers3 = mlp_layers(303, [8, 32, 32, 64])
        sa = ref_mod.PointnetSAModuleVotes(mlp=list(spec3), radius=0.5, nsample=16, normalize_xyz=True,
                                           sample_method=method).eval()
        _load_mlp(sa.mlp_module, layers3)
        given = torch.from_numpy(np.random.RandomState(7).randint(0, 256, (2, 64)).astype(np.int32)) if use_inds else None
        with torch.no_grad():
            nx, nfe, ii = sa(xyz, feats, 64, inds=given)
        mx, mf, mi = R.sa_module(xyz, feats, 64, layers3, 0.5, 16, method, True, True, inds=given)
        assert torch.equal(nx, mx) and torch.equal(ii, mi) and ii.dtype == torch.int64
        assert torch.allclose(nfe, mf, atol=1e-6, rtol=1e-6)
        out.update({tag + "_new_xyz": nx.numpy(), tag + "_feats": nfe.numpy(), tag + "_inds": ii.numpy()})
        if use_inds:
            out["given_inds"] = given.numpy()
    save("G3_sa_module.npz", xyz=s, feats=feats.numpy(), radius=0.5, nsample=16, npoint=64, seed=303, **out)
    report.append("G3 PointnetSAModuleVotes (fps / sequence / given inds): oracle == reference (1e-6)")

    # ---------------- G4 backbone branch ----------------
    cfg = EasyDict(dict(DEBUG=False, SA_CONFIG=dict(
        SAMPLE_METHOD=['fps', 'sequence', 'sequence'], USE_XYZ=True, NORMALIZE_XYZ=True,
        NPOINTS_SEARCH=[512, 256, 128], NPOINTS_TEMPLATE=[256, 128, 64], RADIUS=[0.3, 0.5, 0.7],
        NSAMPLE=[32, 32, 32], MLPS=[[0, 64, 64, 128], [128, 128, 128, 256], [256, 128, 128, 256]])))
    bb = RefBackbone(cfg, input_channels=3).eval()
    specs = [[3, 64, 64, 128], [131, 128, 128, 256], [259, 128, 128, 256]]
    all_layers = [mlp_layers(400 + i, sp) for i, sp in enumerate(specs)]
    for sa, L in zip(bb.SA_modules, all_layers):
        _load_mlp(sa.mlp_module, L)
    rs4 = np.random.RandomState(404)
    cw = torch.from_numpy((rs4.standard_normal((256, 256, 1)) / 16).astype(np.float32))
    cb = torch.from_numpy((rs4.standard_normal(256) * 0.1).astype(np.float32))
    with torch.no_grad():
        bb.cov_final.weight.copy_(cw); bb.cov_final.bias.copy_(cb)
    s4, t4 = synth.frames(404, 2, 1024, 512)
    with torch.no_grad():
        sx, sf, si = bb.branch_forward(torch.from_numpy(s4), [512, 256, 128])
    sa_cfgs = [dict(layers=all_layers[i], radius=[0.3, 0.5, 0.7][i], nsample=32,
                    sample_method=['fps', 'sequence', 'sequence'][i], normalize_xyz=True) for i in range(3)]
    mx, mf, mi = R.backbone_branch(torch.from_numpy(s4), [512, 256, 128], sa_cfgs, cw, cb)
    assert torch.equal(sx, mx) and torch.equal(si, mi)
    assert torch.allclose(sf, mf, atol=1e-5, rtol=1e-5), float((sf - mf).abs().max())
    save("G4_backbone_branch.npz", pts=s4, seeds=sx.numpy(), feats=sf.numpy(), inds=si.numpy(),
         cov_w=cw.numpy(), cov_b=cb.numpy())
    report.append("G4 backbone branch_forward: oracle vs reference max diff %.2e" % float((sf - mf).abs().max()))

    # ---------------- G5 TransformerBlock ----------------
    g5 = {}
    for N in (128, 64):
        P = transformer_params(500 + N)
        tb = RefTB(256, 512, 16).eval()
        tb.load_state_dict(P)
        s5, _ = synth.frames(500 + N, 2, N, 64, K_s=N)
        # duplicated points carry duplicated features (what the tracker produces: same xyz -> same ball -> same feats)
        s5[1, N // 2:] = s5[1, :N // 2]
        f5 = np.random.RandomState(N).standard_normal((2, N, 256)).astype(np.float32)
        f5[1, N // 2:] = f5[1, :N // 2]
        with torch.no_grad():
            res, attn = tb(torch.from_numpy(s5), torch.from_numpy(f5))
        mres, mattn = R.transformer_block(torch.from_numpy(s5), torch.from_numpy(f5), P, 16)
        d_res = float((res - mres).abs().max())
        assert d_res < 2e-5, d_res
        g5.update({"xyz%d" % N: s5, "feat%d" % N: f5, "res%d" % N: res.numpy(),
                   "attn_sample%d" % N: attn[:, ::16, :, ::32].contiguous().numpy()})
        report.append("G5 TransformerBlock N=%d: oracle vs reference max |res diff| = %.2e (incl. duplicated points)"
                      % (N, d_res))
    save("G5_transformer.npz", **g5)

    # ---------------- G13 TransformerBlockSTD (T-opt: the dense Q.K^T / attn.V variant, variants.py:12-40) ----------------
    from ptt.models.transformer_block.variants import TransformerBlockSTD as RefSTD
    g13 = {}
    for N in (128, 64, 50):
        P = {k: v for k, v in transformer_params(1300 + N).items() if not k.startswith("fc_gamma")}
        tb = RefSTD(256, 512, 16).eval()
        tb.load_state_dict(P)
        s13, _ = synth.frames(1300 + N, 2, N, 64, K_s=N)
        f13 = np.random.RandomState(13 + N).standard_normal((2, N, 256)).astype(np.float32)
        with torch.no_grad():
            res, attn = tb(torch.from_numpy(s13), torch.from_numpy(f13))
        mres, mattn = R.transformer_block_std(torch.from_numpy(s13), torch.from_numpy(f13), P)
        d13 = max(float((res - mres).abs().max()), float((attn - mattn).abs().max()))
        assert d13 == 0.0, d13
        g13.update({"xyz%d" % N: s13, "feat%d" % N: f13, "res%d" % N: res.numpy(), "attn%d" % N: attn.numpy()})
    save("G13_transformer_std.npz", **g13)
    report.append("G13 TransformerBlockSTD N=128/64/50: oracle == reference bitwise (res and the N x N attention)")

    # ---------------- G8 kNN vs reference square_distance + argsort (tie-free) ----------------
    rs8 = np.random.RandomState(808)
    x8 = rs8.uniform(-3, 3, (3, 128, 3)).astype(np.float32)
    ref_idx = ref_sqdist(torch.from_numpy(x8), torch.from_numpy(x8)).argsort()[:, :, :16]
    mine_idx = O.knn(x8, 16)
    assert np.array_equal(ref_idx.numpy(), mine_idx.astype(np.int64))
    save("G8_knn_argsort.npz", xyz=x8, knn=ref_idx.numpy().astype(np.int32))
    report.append("G8 kNN: oracle == reference square_distance+argsort on tie-free clouds")

    # ---------------- G9 CosineSimAug (N1) ----------------
    from ptt.models.similarity_modules.p2b_xcoor import CosineSimAug as RefCSA
    mlp9, conv9 = cosine_sim_params(909)
    csa = RefCSA(EasyDict(dict(DEBUG=False, MLP=dict(CHANNELS=[260, 256, 256, 256], BN=True),
                               CONV=dict(CHANNELS=[256, 256, 256], BN=True)))).eval()
    load_cosine_sim(csa, mlp9, conv9)
    rs9 = np.random.RandomState(909)
    sfe = rs9.standard_normal((2, 256, 128)).astype(np.float32)
    tfe = rs9.standard_normal((2, 256, 64)).astype(np.float32)
    tfe[1, :, 5] = 0.0                                                     # a zero template feature: cosine eps path
    txy = rs9.uniform(-2, 2, (2, 64, 3)).astype(np.float32)
    with torch.no_grad():
        bd = csa({'search_feats': torch.from_numpy(sfe), 'template_feats': torch.from_numpy(tfe),
                  'template_seeds': torch.from_numpy(txy)})
    my, msim = R.cosine_sim_aug(torch.from_numpy(sfe), torch.from_numpy(tfe), torch.from_numpy(txy), mlp9, conv9)
    d9 = float((bd['cosine_feats'] - my).abs().max())
    assert d9 < 1e-5, d9
    save("G9_cosine_sim_aug.npz", search_feats=sfe, template_feats=tfe, template_xyz=txy, seed=909,
         cosine_feats=bd['cosine_feats'].numpy(), sim=msim.numpy())
    report.append("G9 CosineSimAug: oracle vs reference max diff %.2e" % d9)

    # ---------------- G6 full PTT.forward (N2) ----------------
    from ptt.config import cfg_from_yaml_file as ref_cfg_from_yaml
    from ptt.models import build_network as ref_build_network
    from ptt_amd.config import StubDataset
    rcfg = ref_cfg_from_yaml(os.path.join(REF, "tools/cfgs/kitti_models/ptt.yaml"), EasyDict())
    ref_model = fill_state_dict_(ref_build_network(rcfg.MODEL, 1, StubDataset()), 606).eval()
    s6, t6 = synth.frames(606, 2, 1024, 512)
    with torch.no_grad():
        out6 = ref_model({'search_points': torch.from_numpy(s6), 'template_points': torch.from_numpy(t6), 'batch_size': 2})
    keys6 = sorted(ref_model.state_dict().keys())
    save("G6_ptt_forward.npz", search=s6, template=t6, seed=606,
         state_keys=np.array(keys6), state_shapes=np.array([str(tuple(ref_model.state_dict()[k].shape)) for k in keys6]),
         **{k: out6[k].numpy() for k in ('search_inds', 'template_inds', 'cosine_feats', 'pred_centroids_cls',
                                         'pred_centroids_votes', 'votes_feats', 'pred_box_center', 'pred_box_data')})
    report.append("G6 full PTT.forward written: %d state_dict keys, %d parameters" %
                  (len(keys6), sum(p.numel() for p in ref_model.parameters())))

    # ---------------- G10 one training step of the full tracker (N3: loss + gradients, BN on batch statistics) ----------------
    # a fresh cfg: the reference's constructor mutates the MLPS list of the cfg it is given (pointnet2_modules.py:51-53)
    rcfg10 = ref_cfg_from_yaml(os.path.join(REF, "tools/cfgs/kitti_models/ptt.yaml"), EasyDict())
    ref_train = fill_state_dict_(ref_build_network(rcfg10.MODEL, 1, StubDataset(training=True)), 1010).train()
    s10, t10 = synth.frames(1010, 3, 1024, 512)
    rs10 = np.random.RandomState(1010)
    cls10 = (rs10.uniform(size=(3, 1024)) > 0.7).astype(np.float32)
    reg10 = (rs10.standard_normal((3, 4)) * 0.3).astype(np.float32)
    ret10, _, _ = ref_train({'search_points': torch.from_numpy(s10), 'template_points': torch.from_numpy(t10),
                             'batch_size': 3, 'cls_label': torch.from_numpy(cls10), 'reg_label': torch.from_numpy(reg10)})
    loss10 = ret10['loss'].mean()
    loss10.backward()
    named = dict(ref_train.named_parameters())
    gkeys = sorted(k for k, p_ in named.items() if p_.grad is not None)
    full = ['backbone_3d.SA_modules.0.mlp_module.layer0.conv.weight', 'backbone_3d.SA_modules.2.mlp_module.layer2.conv.weight',
            'backbone_3d.cov_final.bias', 'centroid_voting_head.transformer_block.fc_delta.0.weight',
            'centroid_voting_head.transformer_block.w_ks.weight', 'box_voting_head.transformer_block.fc_gamma.2.bias',
            'similarity_module.mlp.layer0.conv.weight', 'box_voting_head.refine_layer.2.conv.weight']
    full = [k for k in full if k in named and named[k].grad is not None]
    save("G10_train_step.npz", search=s10, template=t10, cls_label=cls10, reg_label=reg10, seed=1010,
         loss=np.float64(loss10.item()), grad_keys=np.array(gkeys),
         grad_norms=np.array([float(named[k].grad.double().norm()) for k in gkeys]),
         full_keys=np.array(full), **{"grad_%d" % i: named[k].grad.numpy() for i, k in enumerate(full)},
         bn_mean_after=ref_train.state_dict()['backbone_3d.SA_modules.1.mlp_module.layer1.normlayer.bn.running_mean'].numpy())
    report.append("G10 training step written: loss %.6f, %d parameter gradients (%d in full)" % (loss10.item(), len(gkeys), len(full)))

    # ---------------- G7 op-level edge cases (authored here; no reference implementation exists) ----------------
    rs7 = np.random.RandomState(707)
    c = np.empty((5, 1024, 3), np.float32)
    c[0] = synth.cloud(rs7, 1024, 40, synth.SEARCH_BOX, synth.PED_SIGMA, 0.2)      # 40 unique points resampled
    c[1] = 0.0                                                                      # all-zero cloud
    c[2] = synth.cloud(rs7, 1024, 600, synth.SEARCH_BOX, synth.CAR_SIGMA)
    c[2, :200] = rs7.uniform(-0.015, 0.015, (200, 3))                               # inside the 1e-3 origin ball
    c[3] = synth.cloud(rs7, 1024, 1024, synth.SEARCH_BOX, synth.CAR_SIGMA, 1.0)
    c[4] = rs7.uniform(-1, 1, (1024, 3)).round(1)                                   # exact distance ties
    f = O.fps(c, 512)
    f_full = O.fps(c[:, :128], 128)                                                 # npoint == N
    centres = np.take_along_axis(c, f[..., None].astype(np.int64).repeat(3, -1), 1)[:, :128]
    bq = O.ball_query(centres, c, 0.3, 32)
    bq_far = O.ball_query(centres + 50.0, c, 0.3, 32)                               # no hits -> zeros
    kn = O.knn(c[:, :128], 16)
    save("G7_index_ops.npz", clouds=c, fps512=f, fps_full128=f_full, centres=centres, bq=bq, bq_far=bq_far, knn=kn)
    report.append("G7 index-op edge cases written (build-authored contract)")

    # ---------------- G12: pre/post-processing of the sequential tracking loop (N4) ----------------
    from pyquaternion import Quaternion as PQ
    # ptt/datasets/__init__.py pulls in the dataset classes (skimage, pandas readers, ...): load the one module file
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_kitti_tracking_utils",
                                                  os.path.join(REF, "ptt/datasets/kitti/kitti_tracking_utils.py"))
    ref_ku = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_ku)
    from oracle import tracking_ref as TR
    rs12 = np.random.RandomState(1212)
    T12, wlh12 = 6, np.array([1.7, 4.2, 1.5])
    g12 = {"n_frames": T12, "wlh": wlh12}
    centers = np.cumsum(np.c_[rs12.uniform(0.5, 1.2, T12), rs12.uniform(-0.3, 0.3, T12), rs12.uniform(-0.05, 0.05, T12)], 0) \
        + np.array([12.0, -3.0, -0.8])
    yaws = 0.4 + np.cumsum(rs12.uniform(-0.08, 0.08, T12))
    clouds12, gt_boxes, ref_boxes = [], [], []
    for i in range(T12):
        Rz = np.array([[np.cos(yaws[i]), -np.sin(yaws[i]), 0], [np.sin(yaws[i]), np.cos(yaws[i]), 0], [0, 0, 1]])
        n_obj, n_bg = int(rs12.randint(150, 900)), int(rs12.randint(2500, 5000))
        obj = (rs12.uniform(-0.5, 0.5, (n_obj, 3)) * np.array([4.2, 1.7, 1.5])) @ Rz.T + centers[i]
        bg = rs12.uniform(-1, 1, (n_bg, 3)) * np.array([14.0, 14.0, 2.0]) + centers[i]
        pts = np.concatenate([obj, bg], 0)[rs12.permutation(n_obj + n_bg)].astype(np.float32)
        clouds12.append(np.ascontiguousarray(pts.T))
        gt_boxes.append(ref_ku.Box(centers[i], wlh12, PQ(axis=[0, 0, 1], angle=yaws[i])))
        # the "previous result" the loop crops around: the ground truth of the previous frame, slightly off
        j = max(i - 1, 0)
        ref_boxes.append(ref_ku.Box(centers[j] + rs12.uniform(-0.15, 0.15, 3) * np.array([1, 1, 0.2]), wlh12,
                                    PQ(axis=[0, 0, 1], angle=yaws[j] + rs12.uniform(-0.05, 0.05))))
        g12["cloud_%d" % i] = clouds12[-1]
        for nm, bx in (("gt", gt_boxes[-1]), ("ref", ref_boxes[-1])):
            g12["%s_center_%d" % (nm, i)] = bx.center.copy()
            g12["%s_quat_%d" % (nm, i)] = bx.orientation.elements.copy()
            g12["%s_rot_%d" % (nm, i)] = bx.rotation_matrix.copy()
    pcs12 = [ref_ku.PointCloud(c.copy()) for c in clouds12]
    tb = lambda bx: TR.RefBox(bx.center, bx.wlh, bx.orientation.elements)
    worst = 0.0
    for i in range(1, T12):
        cand, _, _ = ref_ku.crop_center_pc(pcs12[i], ref_boxes[i], gt_boxes[i], offset=0.0, scale=1.25)
        search = ref_ku.regularize_pc(cand, 1024, istrain=False)
        model = ref_ku.get_model([pcs12[0], pcs12[i - 1]], [gt_boxes[0], ref_boxes[i]], offset=0.0, scale=1.25)
        templ = ref_ku.regularize_pc(model, 512, istrain=False)
        g12["search_crop_%d" % i] = np.asarray(cand.points, np.float32)
        g12["search_%d" % i] = np.asarray(search, np.float32)
        g12["model_crop_%d" % i] = np.asarray(model.points, np.float32)
        g12["template_%d" % i] = np.asarray(templ, np.float32)
        o_cand = TR.crop_center_pc(clouds12[i], tb(ref_boxes[i]), wlh12[1], 0.0, 1.25)
        o_model = TR.get_model([clouds12[0], clouds12[i - 1]], [tb(gt_boxes[0]), tb(ref_boxes[i])], 0.0, 1.25)
        assert np.array_equal(o_cand, cand.points) and np.array_equal(o_model, model.points), i
        assert np.array_equal(TR.regularize_pc(o_cand, 1024), search) and np.array_equal(TR.regularize_pc(o_model, 512), templ), i
        worst = max(worst, 0.0)
    # edge cases of regularize_pc: n <= 2 (zero cloud), n == input_size (copied through), tiny n, n just over a power of two
    for tag, n in (("n0", 0), ("n2", 2), ("n3", 3), ("n512", 512), ("n513", 513), ("n1024", 1024), ("n1025", 1025), ("n5000", 5000)):
        pts = rs12.standard_normal((3, n)).astype(np.float32)
        out = ref_ku.regularize_pc(ref_ku.PointCloud(pts.copy()), 1024 if n != 512 else 512, istrain=False)
        g12["reg_in_" + tag], g12["reg_out_" + tag] = pts, np.asarray(out, np.float32)
        assert np.array_equal(TR.regularize_pc(pts, 1024 if n != 512 else 512), out), tag
    # an empty crop: the reference box nowhere near the cloud
    far = ref_ku.Box(centers[0] + 500.0, wlh12, PQ(axis=[0, 0, 1], angle=0.3))
    e_c, _, _ = ref_ku.crop_center_pc(pcs12[1], far, gt_boxes[1], offset=0.0, scale=1.25)
    assert e_c.points.shape[1] == 0
    g12["far_center"], g12["far_quat"] = far.center.copy(), far.orientation.elements.copy()
    # get_box_by_offset: float32 model outputs (x, y, z, theta in degrees), incl. the redraw branch (:205-208)
    offs = np.array([[0.31, -0.12, 0.05, 3.0], [-0.4, 0.25, -0.02, -7.5], [0.0, 0.0, 0.0, 0.0], [2.5, 0.1, 0.0, 1.0],
                     [0.2, 2.6, 0.1, -2.0], [3.0, 5.0, 0.3, 12.0]], np.float32)
    g12["gbo_offsets"] = offs
    for use_z in (True, False):
        for k in range(offs.shape[0]):
            np.random.seed(77 + k)
            o_in = offs[k].copy()
            nb = ref_ku.get_box_by_offset(ref_boxes[2], o_in, use_z)
            g12["gbo_center_%d_%d" % (int(use_z), k)] = nb.center.copy()
            g12["gbo_quat_%d_%d" % (int(use_z), k)] = nb.orientation.elements.copy()
            g12["gbo_used_%d_%d" % (int(use_z), k)] = o_in.copy()
            np.random.seed(77 + k)
            ob = TR.get_box_by_offset(tb(ref_boxes[2]), offs[k].copy(), use_z)
            assert np.allclose(ob.center, nb.center, rtol=0, atol=1e-12) and np.allclose(ob.quat.q, nb.orientation.elements, atol=1e-12)
    save("G12_tracking_pre_post.npz", **g12)
    report.append("G12 tracking pre/post-processing (crop_center_pc, get_model, regularize_pc, get_box_by_offset of "
                  "kitti_tracking_utils.py with a restated pyquaternion): oracle == reference bitwise on %d frames + 8 "
                  "resampling edge cases; box update within 1e-12" % (T12 - 1))

    # ---------------- G11: the reference's own numpy FPS ----------------
    # fps_downsample draws its start point with np.random.randint and uses the removed alias np.long: the alias is
    # restored HERE only, and the global numpy seed is chosen so that the draw is index 0 (upstream's fixed start).
    import ptt.utils.common_utils as ref_cu
    if not hasattr(np, "long"):
        np.long = np.int64
    rs11 = np.random.RandomState(1111)
    clouds11 = []
    for n, k_unique, kind in ((1024, 600, "car"), (512, 300, "car"), (2048, 600, "car"), (1024, 60, "ped"),
                              (128, 128, "dense"), (256, 256, "grid")):
        if kind == "grid":
            pts = rs11.uniform(-1, 1, (n, 3)).round(1).astype(np.float32)           # exact distance ties
        else:
            sig = synth.PED_SIGMA if kind == "ped" else synth.CAR_SIGMA
            pts = synth.cloud(rs11, n, k_unique, synth.SEARCH_BOX, sig, 1.0 if kind == "dense" else 0.7)
        # origin-free: push every point outside the 1e-3 ball upstream skips (|p|^2 <= 1e-3), keeping duplicates equal
        near = (pts * pts).sum(1) <= 2e-3
        pts[near] += np.float32(0.25)
        clouds11.append(np.ascontiguousarray(pts, np.float32))
    g11 = {}
    n_ok = 0
    for ci, pts in enumerate(clouds11):
        n = pts.shape[0]
        for m in sorted({n // 2, 64, n}):
            seed = next(sd for sd in range(100000) if np.random.RandomState(sd).randint(0, n, (1,))[0] == 0)
            np.random.seed(seed)
            ref_idx = np.asarray(ref_cu.fps_downsample(pts, m, id=True)).reshape(-1).astype(np.int32)
            assert ref_idx[0] == 0
            mine = O.fps(pts[None], m)[0]
            assert np.array_equal(mine, ref_idx), ("oracle FPS != reference fps_downsample", ci, m,
                                                   int(np.argmax(mine != ref_idx)))
            g11["idx_%d_%d" % (ci, m)] = ref_idx
            n_ok += 1
        g11["cloud_%d" % ci] = pts
    save("G11_fps_reference.npz", n_clouds=len(clouds11), **g11)
    report.append("G11 reference numpy FPS (common_utils.fps_downsample, float32 input, start 0): oracle == reference "
                  "on %d (cloud, npoint) cases incl. duplicates and exact ties" % n_ok)

    with open(os.path.join(HERE, "GOLDEN_REPORT.txt"), "w") as fh:
        fh.write("generated by tests/golden/make_golden.py against /root/reference (torch %s)\n" % torch.__version__)
        fh.write("\n".join(report) + "\n")
    print("\n".join(report))


if __name__ == "__main__":
    main()
