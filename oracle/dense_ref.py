"""CPU ORACLE for the dense half of PTT's hot path — TEST INFRASTRUCTURE ONLY.

Functional (weights passed in as dicts) torch-CPU restatement of the reference's pure
PyTorch modules, each function citing the reference lines it follows. Pinned against the
imported reference by tests/golden/make_golden.py (fixtures under tests/golden/).
May be imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import index_ops


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def query_and_group(xyz, new_xyz, features, radius, nsample, use_xyz=True, normalize_xyz=False):
    """QueryAndGroup.forward — pointnet2_utils.py:320-380 (sample_uniformly off).
    xyz (B,N,3), new_xyz (B,M,3), features (B,C,N)|None -> (new_features (B,3+C,M,ns), grouped_xyz, idx)"""
    idx = index_ops.ball_query(new_xyz.numpy(), xyz.numpy(), radius, nsample)            # :337
    xyz_trans = xyz.transpose(1, 2).contiguous()                                            # :350
    grouped_xyz = _t(index_ops.group(xyz_trans.numpy(), idx))                               # :351
    grouped_xyz = grouped_xyz - new_xyz.transpose(1, 2).unsqueeze(-1)                       # :352
    if normalize_xyz:
        grouped_xyz = grouped_xyz / radius                                                  # :353-354
    if features is not None:
        grouped_features = _t(index_ops.group(features.contiguous().numpy(), idx))         # :357
        new_features = torch.cat([grouped_xyz, grouped_features], dim=1) if use_xyz else grouped_features  # :359-363
    else:
        new_features = grouped_xyz                                                          # :368
    return new_features, grouped_xyz, _t(idx)


def shared_mlp_eval(x, layers):
    """SharedMLP in eval mode — pytorch_utils.py:12-36: per layer Conv2d(1x1, bias=False) ->
    BatchNorm2d(running stats) -> ReLU. layers: list of dicts with conv_weight (Cout,Cin,1,1),
    bn_weight, bn_bias, bn_mean, bn_var, eps."""
    for L in layers:
        x = F.conv2d(x, L["conv_weight"])
        x = F.batch_norm(x, L["bn_mean"], L["bn_var"], L["bn_weight"], L["bn_bias"], False, 0.0, L.get("eps", 1e-5))
        x = F.relu(x)
    return x


def sa_module(xyz, features, npoint, layers, radius, nsample, sample_method="fps", use_xyz=True,
              normalize_xyz=False, inds=None):
    """PointnetSAModuleVotes.forward — pointnet2_modules.py:57-90."""
    B = xyz.shape[0]
    xyz_flipped = xyz.transpose(1, 2).contiguous()                                          # :62
    if inds is None:
        if sample_method in ("rs", "sequence"):
            inds = torch.arange(npoint).repeat(B, 1).int()                                  # :68-71
        elif sample_method == "fps":
            inds = _t(index_ops.fps(xyz.numpy(), npoint))                                   # :72-73
        else:
            raise NotImplementedError(sample_method)
    else:
        assert inds.shape[1] == npoint
    new_xyz = _t(index_ops.gather(xyz_flipped.numpy(), inds.numpy().astype(np.int32))).transpose(1, 2).contiguous()  # :79-81
    grouped, _, _ = query_and_group(xyz, new_xyz, features, radius, nsample, use_xyz, normalize_xyz)    # :83
    y = shared_mlp_eval(grouped, layers)                                                    # :84
    y = F.max_pool2d(y, kernel_size=[1, y.size(3)]).squeeze(-1)                             # :85-88
    return new_xyz, y, inds.to(torch.int64)                                                 # :90


def backbone_branch(pts, npoints, sa_cfgs, cov_w, cov_b):
    """PointNet2BackboneLight.branch_forward — pointnet2_backbone.py:41-50.
    sa_cfgs: 3 dicts(layers, radius, nsample, sample_method, normalize_xyz)."""
    xyz = pts[..., 0:3].contiguous()
    features = pts[..., 3:].transpose(1, 2).contiguous() if pts.size(-1) > 3 else None
    inds_all = []
    for cfg, npnt in zip(sa_cfgs, npoints):
        xyz, features, inds = sa_module(xyz, features, npnt, cfg["layers"], cfg["radius"], cfg["nsample"],
                                        cfg["sample_method"], True, cfg["normalize_xyz"])
        inds_all.append(inds)
    point_features = F.conv1d(features, cov_w, cov_b)                                       # :46
    inds = inds_all[0].gather(1, inds_all[1]).gather(1, inds_all[2])                        # :48
    return xyz, point_features, inds


def square_distance(src, dst):
    """model_utils/layer_utils.py:12-26"""
    return torch.sum((src[:, :, None] - dst[:, None]) ** 2, dim=-1)


def index_points(points, idx):
    """model_utils/layer_utils.py:29-40"""
    raw = idx.size()
    idx = idx.reshape(raw[0], -1)
    res = torch.gather(points, 1, idx[..., None].expand(-1, -1, points.size(-1)))
    return res.reshape(*raw, -1)


def transformer_block(xyz, features, P, k, knn_idx=None):
    """TransformerBlock.forward — transformer_block/variants.py:149-165.
    P: dict of fc1.weight/bias, fc2.*, fc_delta.0.*, fc_delta.2.*, fc_gamma.0.*, fc_gamma.2.*,
    w_qs.weight, w_ks.weight, w_vs.weight. kNN order: (distance, index) ascending (oracle_knn)."""
    if knn_idx is None:
        knn_idx = _t(index_ops.knn(xyz.numpy(), k)).long()                                  # :150-151
    knn_xyz = index_points(xyz, knn_idx)                                                    # :152
    pre = features
    x = F.linear(features, P["fc1.weight"], P["fc1.bias"])                                  # :155
    q = F.linear(x, P["w_qs.weight"])
    kk = index_points(F.linear(x, P["w_ks.weight"]), knn_idx)
    v = index_points(F.linear(x, P["w_vs.weight"]), knn_idx)                                # :156
    d = xyz[:, :, None] - knn_xyz
    pos_enc = F.linear(F.relu(F.linear(d, P["fc_delta.0.weight"], P["fc_delta.0.bias"])),
                       P["fc_delta.2.weight"], P["fc_delta.2.bias"])                        # :158
    a = q[:, :, None] - kk + pos_enc
    a = F.linear(F.relu(F.linear(a, P["fc_gamma.0.weight"], P["fc_gamma.0.bias"])),
                 P["fc_gamma.2.weight"], P["fc_gamma.2.bias"])                              # :160
    attn = F.softmax(a / np.sqrt(kk.size(-1)), dim=-2)                                      # :161
    res = torch.einsum("bmnf,bmnf->bmf", attn, v + pos_enc)                                 # :163
    res = F.linear(res, P["fc2.weight"], P["fc2.bias"]) + pre                               # :164
    return res, attn


def transformer_block_std(xyz, features, P):
    """TransformerBlockSTD.forward — transformer_block/variants.py:29-40 (the dense Q.K^T / attn.V variant).
    P: fc1.*, fc2.*, fc_delta.0.*, fc_delta.2.*, w_qs.weight, w_ks.weight, w_vs.weight."""
    pre = features
    x = F.linear(features, P["fc1.weight"], P["fc1.bias"])                                  # :31
    q, k, v = F.linear(x, P["w_qs.weight"]), F.linear(x, P["w_ks.weight"]), F.linear(x, P["w_vs.weight"])   # :32
    attn = q @ k.transpose(1, 2)                                                            # :34
    attn = F.softmax(attn / np.sqrt(k.size(-1)), dim=-1)                                    # :35
    pos_enc = F.linear(F.relu(F.linear(xyz, P["fc_delta.0.weight"], P["fc_delta.0.bias"])),
                       P["fc_delta.2.weight"], P["fc_delta.2.bias"])                        # :37
    res = attn @ (v + pos_enc)                                                              # :38
    res = F.linear(res, P["fc2.weight"], P["fc2.bias"]) + pre                               # :39
    return res, attn


def cosine_sim_aug(search_feats, template_feats, template_xyz, mlp_layers, conv):
    """CosineSimAug.forward — similarity_modules/p2b_xcoor.py:25-46.
    search_feats (B,f,n2), template_feats (B,f,n1), template_xyz (B,n1,3); mlp_layers as shared_mlp_eval;
    conv: dict with conv0_weight (256,256,1), bn0_*, conv1_weight, conv1_bias (the Seq stack, :19-23)."""
    b, f, n2 = search_feats.shape
    n1 = template_feats.shape[-1]
    sim = F.cosine_similarity(template_feats.unsqueeze(-1).expand(b, f, n1, n2),
                              search_feats.unsqueeze(2).expand(b, f, n1, n2), dim=1)              # :35-36
    txyz = template_xyz.transpose(1, 2).contiguous().unsqueeze(-1).expand(b, 3, n1, n2)           # :37
    fusion = torch.cat((sim.unsqueeze(1), txyz), dim=1)                                           # :38
    fusion = torch.cat((fusion, template_feats.unsqueeze(-1).expand(b, f, n1, n2)), dim=1)        # :39
    fusion = shared_mlp_eval(fusion, mlp_layers)                                                  # :40
    fusion = F.max_pool2d(fusion, kernel_size=[fusion.size(2), 1]).squeeze(2)                     # :41-42
    y = F.conv1d(fusion, conv["conv0_weight"])
    y = F.relu(F.batch_norm(y, conv["bn0_mean"], conv["bn0_var"], conv["bn0_weight"], conv["bn0_bias"], False, 0.0, 1e-5))
    y = F.conv1d(y, conv["conv1_weight"], conv["conv1_bias"])                                     # :43
    return y, sim
