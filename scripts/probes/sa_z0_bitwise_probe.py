"""Dev probe: is ptt_sa_z0_rows_f32 bit-identical to the launches it replaces (group / subtract / divide / gather_rows / K = 3
linear with residual)?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ptt_amd import ops, synth, train_ops
from ptt_amd.models.backbones_3d.pointnet2 import pointnet2_utils as pu
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (B, N, M, ns, C0, C, radius) in ((4, 1024, 512, 32, 64, 0, 0.3), (4, 512, 256, 32, 128, 128, 0.5)):
    s, _ = synth.frames(3, B, N, 64, K_s=N // 3)
    xyz = torch.from_numpy(s).to(dev)
    inds = pu.furthest_point_sample(xyz, M)
    new_xyz, _, idx = ops.centres_ball_query(xyz, inds, M, radius, ns)
    new_xyz2 = pu.gather_operation(xyz.transpose(1, 2).contiguous(), inds).transpose(1, 2).contiguous()
    idx2 = pu.ball_query(radius, ns, xyz, new_xyz2)
    print("centres equal", torch.equal(new_xyz, new_xyz2), "idx equal", torch.equal(idx, idx2))
    w0 = torch.randn(C0, 3 + C, device=dev) * 0.3
    rel = pu.grouping_operation(xyz.transpose(1, 2).contiguous(), idx) - new_xyz.transpose(1, 2).unsqueeze(-1)
    rel = rel / radius
    rel_rows = rel.permute(0, 2, 3, 1).reshape(B * M * ns, 3).contiguous()
    if C:
        f = torch.randn(B, N, C, device=dev)
        term = ops.linear(f, ops.pack_weight(w0[:, 3:].contiguous()), C0)
        gathered = ops.gather_rows(term, idx.view(B, M * ns)).view(B * M * ns, C0)
        old = ops.linear(rel_rows, ops.pack_weight(w0[:, 0:3].contiguous()), C0, residual=gathered)
    else:
        term = None
        old = ops.linear(rel_rows, ops.pack_weight(w0[:, 0:3].contiguous()), C0)
    new, rel_new = ops.sa_z0_rows(xyz, new_xyz, idx, term, w0[:, 0:3], radius, True)
    d = (old - new).abs()
    print("C=%d: rel rows equal %s; z0 equal %s (differing %d of %d, worst %.3e, worst rel %.3e)" % (
        C, torch.equal(rel_rows, rel_new), torch.equal(old, new), int((d > 0).sum()), d.numel(), float(d.max()),
        float((d / old.abs().clamp_min(1e-6)).max())))
