"""CPU suite (-m "not gpu"): the oracle against the committed golden fixtures (generated from the imported
reference by tests/golden/make_golden.py), the C-ABI library's symbol table, checkpoint key names and the
host-side sharding logic. No kernel is launched here."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from oracle import dense_ref as R
from oracle import index_ops as O
from tests.util import cosine_sim_params, mlp_layers, transformer_params

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _g(name):
    return np.load(os.path.join(GOLD, name))


def test_G1_query_and_group():
    g = _g("G1_query_and_group.npz")
    nf, gx, idx = R.query_and_group(torch.from_numpy(g["xyz"]), torch.from_numpy(g["new_xyz"]),
                                    torch.from_numpy(g["feats"]), float(g["radius"]), int(g["nsample"]), True, True)
    np.testing.assert_array_equal(idx.numpy(), g["idx"])
    np.testing.assert_array_equal(gx.numpy(), g["grouped_xyz"])
    np.testing.assert_array_equal(nf.numpy(), g["new_features"])
    assert nf.shape[1] == 3 + g["feats"].shape[1]                  # xyz channels first, then features


def test_G2_shared_mlp_eval():
    g = _g("G2_shared_mlp.npz")
    y = R.shared_mlp_eval(torch.from_numpy(g["x"]), mlp_layers(int(g["seed"]), list(g["spec"])))
    np.testing.assert_allclose(y.numpy(), g["y"], atol=1e-6, rtol=1e-6)


@pytest.mark.parametrize("tag,method", [("fps", "fps"), ("seq", "sequence"), ("inds", "fps")])
def test_G3_sa_module(tag, method):
    g = _g("G3_sa_module.npz")
    inds = torch.from_numpy(g["given_inds"]) if tag == "inds" else None
    nx, nf, ii = R.sa_module(torch.from_numpy(g["xyz"]), torch.from_numpy(g["feats"]), int(g["npoint"]),
                             mlp_layers(int(g["seed"]), [8, 32, 32, 64]), float(g["radius"]), int(g["nsample"]),
                             method, True, True, inds=inds)
    assert ii.dtype == torch.int64
    np.testing.assert_array_equal(ii.numpy(), g[tag + "_inds"])
    np.testing.assert_array_equal(nx.numpy(), g[tag + "_new_xyz"])
    np.testing.assert_allclose(nf.numpy(), g[tag + "_feats"], atol=1e-6, rtol=1e-6)


def test_G4_backbone_branch():
    g = _g("G4_backbone_branch.npz")
    specs = [[3, 64, 64, 128], [131, 128, 128, 256], [259, 128, 128, 256]]
    sa_cfgs = [dict(layers=mlp_layers(400 + i, sp), radius=[0.3, 0.5, 0.7][i], nsample=32,
                    sample_method=['fps', 'sequence', 'sequence'][i], normalize_xyz=True) for i, sp in enumerate(specs)]
    x, f, i = R.backbone_branch(torch.from_numpy(g["pts"]), [512, 256, 128], sa_cfgs, torch.from_numpy(g["cov_w"]),
                                torch.from_numpy(g["cov_b"]))
    np.testing.assert_array_equal(i.numpy(), g["inds"])
    np.testing.assert_array_equal(x.numpy(), g["seeds"])
    np.testing.assert_allclose(f.numpy(), g["feats"], atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize("N", [128, 64])
def test_G5_transformer_block(N):
    g = _g("G5_transformer.npz")
    res, attn = R.transformer_block(torch.from_numpy(g["xyz%d" % N]), torch.from_numpy(g["feat%d" % N]),
                                    transformer_params(500 + N), 16)
    np.testing.assert_allclose(res.numpy(), g["res%d" % N], atol=2e-5, rtol=2e-5)
    np.testing.assert_allclose(attn[:, ::16, :, ::32].numpy(), g["attn_sample%d" % N], atol=1e-6, rtol=1e-5)
    np.testing.assert_allclose(attn.sum(-2).numpy(), 1.0, atol=1e-5)     # softmax over the neighbour axis


def test_G9_cosine_sim_aug():
    g = _g("G9_cosine_sim_aug.npz")
    mlp, conv = cosine_sim_params(int(g["seed"]))
    y, sim = R.cosine_sim_aug(torch.from_numpy(g["search_feats"]), torch.from_numpy(g["template_feats"]),
                              torch.from_numpy(g["template_xyz"]), mlp, conv)
    np.testing.assert_allclose(y.numpy(), g["cosine_feats"], atol=1e-5, rtol=1e-5)
    np.testing.assert_allclose(sim.numpy(), g["sim"], atol=1e-6)
    assert np.abs(g["sim"][1, 5]).max() == 0.0            # zero template feature -> cosine 0 (eps clamp), not NaN


def test_G7_index_op_edge_cases():
    g = _g("G7_index_ops.npz")
    c = g["clouds"]
    np.testing.assert_array_equal(O.fps(c, 512), g["fps512"])
    np.testing.assert_array_equal(O.fps(c[:, :128], 128), g["fps_full128"])
    np.testing.assert_array_equal(O.ball_query(g["centres"], c, 0.3, 32), g["bq"])
    np.testing.assert_array_equal(O.ball_query(g["centres"] + 50.0, c, 0.3, 32), g["bq_far"])
    np.testing.assert_array_equal(O.knn(c[:, :128], 16), g["knn"])
    assert (g["fps512"][1] == 0).all() and (g["bq_far"] == 0).all()       # zero cloud / empty balls
    assert sorted(g["fps_full128"][3].tolist()) == list(range(128))       # npoint == N on distinct points: a permutation
    # origin-ball points are never selected (except the forced start index 0)
    mag = (c[2] ** 2).sum(-1)
    assert (mag[g["fps512"][2][1:]] > 1e-3).all()


def test_G11_oracle_fps_equals_the_references_own_numpy_fps():
    """G11 = outputs of the reference's fps_downsample (ptt/utils/common_utils.py:78-112) on origin-free clouds with
    duplicates and exact ties: the one reference-held pin of FPS semantics (start 0, min-update, lowest index wins)."""
    g = np.load(os.path.join(GOLD, "G11_fps_reference.npz"))
    n = 0
    for ci in range(int(g["n_clouds"])):
        pts = g["cloud_%d" % ci]
        for key in [k for k in g.files if k.startswith("idx_%d_" % ci)]:
            m = int(key.split("_")[2])
            np.testing.assert_array_equal(O.fps(pts[None], m)[0], g[key], err_msg=key)
            n += 1
    assert n == 17


def test_G13_transformer_block_std_oracle_and_mirror():
    """Fixture G13 (the reference's TransformerBlockSTD): the oracle restatement and the mirror module's stock-torch
    path (CPU) both reproduce it."""
    from ptt_amd.models.transformer_block.variants import TransformerBlockSTD
    from tests.util import transformer_params
    g = np.load(os.path.join(GOLD, "G13_transformer_std.npz"))
    for N in (128, 64, 50):
        P = {k: v for k, v in transformer_params(1300 + N).items() if not k.startswith("fc_gamma")}
        xyz, feat = torch.from_numpy(g["xyz%d" % N]), torch.from_numpy(g["feat%d" % N])
        res, attn = R.transformer_block_std(xyz, feat, P)
        np.testing.assert_array_equal(res.numpy(), g["res%d" % N])
        np.testing.assert_array_equal(attn.numpy(), g["attn%d" % N])
        tb = TransformerBlockSTD(256, 512, 16).eval()
        tb.load_state_dict(P)
        with torch.no_grad():
            r2, a2 = tb(xyz, feat)
        np.testing.assert_allclose(r2.numpy(), g["res%d" % N], atol=1e-6, rtol=1e-6)
        np.testing.assert_allclose(a2.numpy(), g["attn%d" % N], atol=1e-7, rtol=1e-6)


def test_G8_knn_equals_reference_argsort():
    g = _g("G8_knn_argsort.npz")
    np.testing.assert_array_equal(O.knn(g["xyz"], 16), g["knn"])


def test_oracle_semantics_small():
    """Hand-checkable cases of the spec in SURVEY.md §8c."""
    pts = np.array([[[0, 0, 0], [1, 0, 0], [0, 2, 0], [3, 0, 0], [1, 0, 0]]], np.float32)
    # start 0 (skipped point: inside the origin ball), farthest from it is 3 (d=9), then 2, then 1 (tie 1/4 -> lowest)
    np.testing.assert_array_equal(O.fps(pts, 4), [[0, 3, 2, 1]])
    bq = O.ball_query(np.array([[[1, 0, 0]]], np.float32), pts, 1.01, 4)
    np.testing.assert_array_equal(bq, [[[0, 1, 4, 0]]])                    # hits 0,1,4 in index order, pad = first hit
    np.testing.assert_array_equal(O.ball_query(np.array([[[9, 9, 9]]], np.float32), pts, 0.5, 3), [[[0, 0, 0]]])
    np.testing.assert_array_equal(O.knn(pts, 3)[0, 1], [1, 4, 0])          # duplicate at distance 0, index order


def test_gather_group_grads_are_adjoint():
    rs = np.random.RandomState(0)
    f = rs.standard_normal((2, 3, 17)).astype(np.float32)
    idx = rs.randint(0, 17, (2, 5, 4)).astype(np.int32)
    go = rs.standard_normal((2, 3, 5, 4)).astype(np.float32)
    lhs = float((O.group(f, idx) * go).sum())
    rhs = float((f * O.group_grad(go, idx, 17)).sum())
    assert abs(lhs - rhs) < 1e-3


def test_library_exports_every_declared_symbol():
    """include/ptt_hip.h <-> libptt_hip.so: every declared entry point is exported (no compute call)."""
    header = open(os.path.join(ROOT, "include", "ptt_hip.h")).read()
    declared = set(re.findall(r"\b(ptt_[a-z0-9_]+)\s*\(", header))
    from ptt_amd import _lib
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    assert os.path.exists(_lib.LIB_PATH), "run `python -m ptt_amd.build` first"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    lib.ptt_version.restype = ctypes.c_int
    version = int(re.search(r"#define PTT_ABI_VERSION (\d+)", header).group(1))
    assert lib.ptt_version() == version == _lib.ABI_VERSION


def test_product_path_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "ptt_amd")):
        for fn in files:
            if fn.endswith(".py"):
                src = open(os.path.join(dirpath, fn)).read()
                path = os.path.join(dirpath, fn)
                assert not re.search(r"^\s*(from|import)\s+\.*oracle", src, re.M), path      # no import of oracle/
                assert "libptt_oracle" not in src or fn == "build.py", path               # only build.py names the checker's .so


def test_ops_reject_cpu_tensors_loudly():
    from ptt_amd import ops
    with pytest.raises(RuntimeError):
        ops.furthest_point_sampling(torch.zeros(1, 8, 3), 4)
    with pytest.raises(RuntimeError):
        ops.knn(torch.zeros(1, 8, 3), 4)


def test_state_dict_keys_match_reference_contract():
    from ptt_amd.hot_path import FrameHotPath
    sd = FrameHotPath().state_dict()
    for k in ["backbone_3d.SA_modules.0.mlp_module.layer0.conv.weight",
              "backbone_3d.SA_modules.2.mlp_module.layer2.normlayer.bn.running_var",
              "backbone_3d.SA_modules.1.mlp_module.layer1.normlayer.bn.num_batches_tracked",
              "backbone_3d.cov_final.weight", "backbone_3d.cov_final.bias",
              "vote_aggregation.mlp_module.layer0.conv.weight",
              "centroid_transformer.fc_delta.2.bias", "box_transformer.fc_gamma.0.weight",
              "box_transformer.w_qs.weight"]:
        assert k in sd, k
    assert tuple(sd["vote_aggregation.mlp_module.layer0.conv.weight"].shape) == (256, 260, 1, 1)
    assert tuple(sd["backbone_3d.SA_modules.0.mlp_module.layer0.conv.weight"].shape) == (64, 3, 1, 1)
    tb = sum(v.numel() for k, v in sd.items() if k.startswith("box_transformer."))
    assert tb == 1839360                                                    # SURVEY.md §8a T3 [probe]


def test_G6_state_dict_keys_and_shapes_equal_the_reference_model():
    """N2: our assembled PTT has exactly the reference model's state_dict (175 keys, every shape) — checkpoints load."""
    from ptt_amd.config import StubDataset, ptt_model_cfg
    from ptt_amd.models import build_network
    g = _g("G6_ptt_forward.npz")
    m = build_network(ptt_model_cfg(), 1, StubDataset())
    sd = m.state_dict()
    assert sorted(sd.keys()) == list(g["state_keys"])
    assert [str(tuple(sd[k].shape)) for k in sorted(sd.keys())] == list(g["state_shapes"])
    assert sum(p.numel() for p in m.parameters()) == 4903113


def test_config_mirror_semantics(tmp_path):
    from ptt_amd.config import EasyDict, cfg_from_list, cfg_from_yaml_file
    base = tmp_path / "base.yaml"
    base.write_text("A: {x: 1, y: [1, 2]}\nB: hello\n")
    top = tmp_path / "top.yaml"
    top.write_text("_BASE_CONFIG_: %s\nA: {x: 5}\nC: {d: {e: 0.5}}\n" % base)
    c = cfg_from_yaml_file(str(top), EasyDict())
    assert c.A.x == 5 and c.A.y == [1, 2] and c.B == "hello" and c.C.d.e == 0.5
    cfg_from_list(["A.x", "7", "A.y", "3,4", "B", "bye"], c)
    assert c.A.x == 7 and c.A.y == [3, 4] and c.B == "bye"
    with pytest.raises(AssertionError):
        cfg_from_list(["A.nope", "1"], c)


def test_G10_module_mirror_training_step_equals_reference_on_cpu(monkeypatch):
    """N3 logic parity without a GPU: the module mirror in TRAINING mode (reference op sequence, BatchNorm on batch
    statistics, autograd through gather/group) with ptt_amd.ops' six index ops swapped for the CPU oracle reproduces the
    reference model's loss and all 106 parameter gradients of fixture G10 exactly — same torch CPU kernels underneath,
    so any difference would be a difference in the mirrored logic (layer order, loss terms, routing of gradients)."""
    import ptt_amd.ops as ops
    from ptt_amd.config import StubDataset, ptt_model_cfg
    from ptt_amd.models import build_network
    from tests.util import fill_state_dict_
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    c = lambda x: x.detach().contiguous().numpy()
    monkeypatch.setattr(ops, "furthest_point_sampling", lambda xyz, n: t(O.fps(c(xyz), n)))
    monkeypatch.setattr(ops, "gather_points", lambda f, i: t(O.gather(c(f), i.numpy())))
    monkeypatch.setattr(ops, "gather_points_grad", lambda g_, i, n: t(O.gather_grad(c(g_), i.numpy(), n)))
    monkeypatch.setattr(ops, "ball_query", lambda new_xyz, xyz, r, ns: t(O.ball_query(c(new_xyz), c(xyz), r, ns)))
    monkeypatch.setattr(ops, "group_points", lambda f, i: t(O.group(c(f), i.numpy())))
    monkeypatch.setattr(ops, "group_points_grad", lambda g_, i, n: t(O.group_grad(c(g_), i.numpy(), n)))
    g = np.load(os.path.join(GOLD, "G10_train_step.npz"))
    model = fill_state_dict_(build_network(ptt_model_cfg(), 1, StubDataset(training=True)), int(g["seed"])).train()
    ret, _, _ = model({'search_points': t(g["search"]), 'template_points': t(g["template"]), 'batch_size': 3,
                       'cls_label': t(g["cls_label"]), 'reg_label': t(g["reg_label"])})
    loss = ret['loss'].mean()
    loss.backward()
    np.testing.assert_allclose(float(loss.detach()), float(g["loss"]), rtol=1e-6)
    named = dict(model.named_parameters())
    keys = [str(k) for k in g["grad_keys"]]
    assert sorted(k for k, p in named.items() if p.grad is not None) == keys
    norms = np.array([float(named[k].grad.double().norm()) for k in keys])
    np.testing.assert_allclose(norms, g["grad_norms"], rtol=1e-4, atol=1e-7)
    for i, k in enumerate(str(k) for k in g["full_keys"]):
        ref = g["grad_%d" % i]
        np.testing.assert_allclose(named[k].grad.numpy(), ref, atol=1e-5 * float(np.abs(ref).max()) + 1e-9, rtol=1e-4,
                                   err_msg=k)


def test_checkpoint_helpers_partial_and_full_load(tmp_path):
    """N2: load_params_from_file takes every entry whose key AND shape match (tracker3d_template.py:96-124) and leaves the
    rest; load_params_with_optimizer restores model + optimizer + (it, epoch) (:126-155)."""
    import logging
    from ptt_amd.config import StubDataset, ptt_model_cfg
    from ptt_amd.models import build_network
    from tests.util import fill_state_dict_
    src = fill_state_dict_(build_network(ptt_model_cfg(), 1, StubDataset()), 5)
    sd = {k: v.clone() for k, v in src.state_dict().items()}
    odd = 'backbone_3d.cov_final.bias'
    sd[odd] = torch.zeros(7)                               # wrong shape: must be skipped, not raise
    sd['not.in.the.model'] = torch.zeros(3)
    ck = tmp_path / "ckpt.pth"
    torch.save({'model_state': sd, 'version': 'test', 'epoch': 3, 'it': 17}, str(ck))
    dst = fill_state_dict_(build_network(ptt_model_cfg(), 1, StubDataset()), 6)
    before = dst.state_dict()[odd].clone()
    dst.load_params_from_file(str(ck), logging.getLogger("ckpt"), to_cpu=True)
    got = dst.state_dict()
    for k, v in src.state_dict().items():
        if k == odd:
            assert torch.equal(got[k], before)
        else:
            assert torch.equal(got[k], v), k
    # full load with optimizer state
    opt = torch.optim.Adam(src.parameters(), lr=1e-3)
    torch.save({'model_state': src.state_dict(), 'optimizer_state': opt.state_dict(), 'epoch': 3, 'it': 17}, str(ck))
    dst2 = build_network(ptt_model_cfg(), 1, StubDataset())
    opt2 = torch.optim.Adam(dst2.parameters(), lr=5e-2)
    it, epoch = dst2.load_params_with_optimizer(str(ck), to_cpu=True, optimizer=opt2, logger=logging.getLogger("ckpt"))
    assert (it, epoch) == (17, 3) and opt2.param_groups[0]['lr'] == 1e-3
    assert all(torch.equal(a, b) for a, b in zip(dst2.state_dict().values(), src.state_dict().values()))


def test_ctypes_structures_match_the_c_header(tmp_path):
    """Every structure ptt_amd/_lib.py mirrors from include/ptt_hip.h has the size and the field offsets the C compiler gives it:
    a probe compiled here with gcc prints sizeof / offsetof, ctypes must agree (a drifted field would only show up as wrong
    results on the GPU box)."""
    import subprocess
    from ptt_amd import _lib
    pairs = [("ptt_row_job", _lib.RowJob), ("ptt_point_job", _lib.PointJob), ("ptt_xcorr_desc", _lib.XcorrDesc), ("ptt_sa_desc", _lib.SaDesc),
             ("ptt_attn_desc", _lib.AttnDesc), ("ptt_sa_layer", _lib.SaLayer), ("ptt_crop_job", _lib.CropJob),
             ("ptt_regularize_job", _lib.RegularizeJob), ("ptt_pack_job", _lib.PackJob), ("ptt_bn_train_tail", _lib.BnTrainTail),
             ("ptt_track_loss_desc", _lib.TrackLossDesc), ("ptt_adam_tensor", _lib.AdamTensor), ("ptt_adam_hyper", _lib.AdamHyper),
             ("ptt_bn_bwd_input", _lib.BnBwdInput), ("ptt_grad_job", _lib.GradJob), ("ptt_grad_segment", _lib.GradSegment)]
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "ptt_hip.h"', 'int main(void) {']
    for cname, st in pairs:
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (cname, cname))
        for fname, *_ in st._fields_:
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (cname, fname, cname, fname))
    lines += ['return 0;', '}']
    src = tmp_path / "abi_probe.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "abi_probe"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = dict(l.split() for l in subprocess.check_output([str(exe)]).decode().splitlines())
    for cname, st in pairs:
        assert int(got[cname]) == ctypes.sizeof(st), (cname, got[cname], ctypes.sizeof(st))
        for fname, *_ in st._fields_:
            assert int(got["%s.%s" % (cname, fname)]) == getattr(st, fname).offset, (cname, fname)


def _g17_check_ball(idx, cnt, first, tag):
    """idx (B,M,ns) against fixture G17: slots 0 .. min(count, ns) - 1 = the reference's ascending in-ball list; the remaining slots
    hold the first hit (zeros when the ball is empty) — upstream's fill rule, the part only this build's restatement states."""
    ns = idx.shape[-1]
    c = np.minimum(cnt, ns)
    slot = np.arange(ns)[None, None, :]
    pinned = slot < c[:, :, None]
    assert np.array_equal(idx[pinned], first[pinned]), tag
    fill = np.where(cnt[:, :, None] > 0, first[:, :, :1], 0)
    assert np.array_equal(idx[~pinned], np.broadcast_to(fill, idx.shape)[~pinned]), tag + " (fill rule)"
    return int(pinned.sum())


def test_G17_oracle_ball_query_hits_are_the_references_square_distance_hits():
    """The one reference-held statement about ball query: which points lie inside the ball under the reference's own fp32
    square_distance (layer_utils.py:12-26), in ascending order — for the nine (M, N, r, nsample) calls of fixture G17."""
    g = _g("G17_ball_hits_origin_fps.npz")
    pinned = 0
    for name in [str(n) for n in g["ball_cases"]]:
        xyz, sel = g[name + "_xyz"], g[name + "_sel"]
        r, ns = float(g[name + "_rn"][0]), int(g[name + "_rn"][1])
        centres = np.take_along_axis(xyz, sel[:, :, None].astype(np.int64), 1)
        pinned += _g17_check_ball(O.ball_query(centres, xyz, r, ns), g[name + "_count"], g[name + "_first"], name)
    assert pinned > 50000


def test_G17_oracle_fps_with_origin_ball_points_is_the_references_fps_on_the_kept_points():
    g = _g("G17_ball_hits_origin_fps.npz")
    for ci, m in g["fps_cases"]:
        pts = g["fps_cloud_%d" % ci]
        np.testing.assert_array_equal(O.fps(pts[None], int(m))[0], g["fps_idx_%d_%d" % (ci, m)], err_msg="cloud %d npoint %d" % (ci, m))
    assert len(g["fps_cases"]) == 15
