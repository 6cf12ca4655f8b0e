"""CPU ORACLE of the whole per-frame hot path — TEST INFRASTRUCTURE ONLY.

Takes the state_dict of ptt_amd.hot_path.FrameHotPath (reference key names) and evaluates the
same frame on the CPU with oracle.index_ops + oracle.dense_ref, i.e. the restated reference
path: PointNet2BackboneLight.forward (pointnet2_backbone.py:52-67) -> TransformerBlock
(centroids_voting_head.py:71-76) -> vote_aggregation (box_voting_head.py:75-79) -> TransformerBlock
(box_voting_head.py:81-86). Used by tests, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import torch

from . import dense_ref as R


def _mlp_layers(sd, prefix):
    layers, i = [], 0
    while prefix + "layer%d.conv.weight" % i in sd:
        p = prefix + "layer%d." % i
        layers.append({"conv_weight": sd[p + "conv.weight"], "bn_weight": sd[p + "normlayer.bn.weight"],
                       "bn_bias": sd[p + "normlayer.bn.bias"], "bn_mean": sd[p + "normlayer.bn.running_mean"],
                       "bn_var": sd[p + "normlayer.bn.running_var"], "eps": 1e-5})
        i += 1
    return layers


def _sub(sd, prefix):
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}


def frame(state_dict, cfg, search_points, template_points):
    sd = {k: v.detach().cpu().float() for k, v in state_dict.items()}
    sa = cfg.BACKBONE_3D.SA_CONFIG
    sa_cfgs = [dict(layers=_mlp_layers(sd, "backbone_3d.SA_modules.%d.mlp_module." % k), radius=sa.RADIUS[k],
                    nsample=sa.NSAMPLE[k], sample_method=sa.SAMPLE_METHOD[k], normalize_xyz=sa.NORMALIZE_XYZ)
               for k in range(3)]
    cw, cb = sd["backbone_3d.cov_final.weight"], sd["backbone_3d.cov_final.bias"]
    out = {}
    out["search_seeds"], out["search_feats"], out["search_inds"] = R.backbone_branch(
        search_points, sa.NPOINTS_SEARCH, sa_cfgs, cw, cb)
    out["template_seeds"], out["template_feats"], out["template_inds"] = R.backbone_branch(
        template_points, sa.NPOINTS_TEMPLATE, sa_cfgs, cw, cb)
    k = cfg.CENTROID_HEAD.TRANSFORMER_BLOCK.KNN
    fused, _ = R.transformer_block(out["search_seeds"], out["search_feats"].transpose(1, 2).contiguous(),
                                   _sub(sd, "centroid_transformer."), k)
    score = torch.sigmoid(fused[:, :, :1])
    votes_feats = torch.cat((score, fused), dim=2).transpose(1, 2).contiguous()
    bs = cfg.BOX_HEAD.SA_CONFIG
    centres, prop, _ = R.sa_module(out["search_seeds"], votes_feats, bs.NPOINTS,
                                   _mlp_layers(sd, "vote_aggregation.mlp_module."), bs.RADIUS, bs.NSAMPLE,
                                   bs.SAMPLE_METHOD, True, bs.NORMALIZE_XYZ)
    box, _ = R.transformer_block(centres, prop.transpose(1, 2).contiguous(), _sub(sd, "box_transformer."),
                                 cfg.BOX_HEAD.TRANSFORMER_BLOCK.KNN)
    out["centroid_feats"], out["pred_box_center"], out["box_feats"] = fused, centres, box
    return out
