"""The attention core of a Point-Transformer block in training mode as one autograd function (train_ops._AttnCore:
ptt_rows_gemm_rsum16_f32, ptt_scatter_rows_csr_sub_f32) against the three-function form it replaces and against the reference's
own op sequence (variants.py:149-165) in stock torch."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_rows_gemm_with_residual_and_group_sums(dev):
    from ptt_amd import ops
    torch.manual_seed(2)
    for rows, K, N in [(98304, 512, 512), (6144, 512, 512), (4096, 128, 256), (1600, 256, 128)]:
        x, W = torch.randn(rows, K, device=dev), torch.randn(N, K, device=dev) / K ** 0.5
        res = torch.randn(rows, N, device=dev) * 100.0
        assert ops.rows_gemm_rsum16_supported(x, K, N)
        plain, out, gsum = ops.rows_gemm_rsum16(x, ops.pack_weight(W), N, res)
        want = x.double() @ W.double().t()
        assert float((plain.double() - want).abs().max()) < 2e-6 * float(want.abs().max()), (rows, K, N)
        assert torch.equal(out, plain + res)
        wsum = want.view(rows // 16, 16, N).sum(1)
        assert float((gsum.double() - wsum).abs().max()) < 2e-6 * float(wsum.abs().max()), (rows, K, N)
        plain2, out2, gsum2 = ops.rows_gemm_rsum16(x, ops.pack_weight(W), N, res)
        assert torch.equal(out, out2) and torch.equal(gsum, gsum2) and torch.equal(plain, plain2)
    assert not ops.rows_gemm_rsum16_supported(torch.randn(1608, 256, device=dev), 256, 128)      # not whole groups of 16 rows


def test_row_scatter_subtracted_from_a_minuend(dev):
    from ptt_amd import ops
    torch.manual_seed(4)
    B, N, E, C = 3, 50, 800, 64
    g = torch.randn(B, E, C, device=dev)
    idx = torch.randint(0, N, (B, E), device=dev, dtype=torch.int32)
    m = torch.randn(B, N, C, device=dev)
    plain = ops.scatter_rows_det(g, idx, N)
    sub = ops.scatter_rows_det(g, idx, N, minuend=m)
    assert torch.equal(sub, m - plain)
    assert torch.equal(ops.scatter_rows_det(g, idx, N, negate=True), 0.0 - plain)


@pytest.mark.parametrize("B,N", [(48, 128), (3, 64)])
def test_the_fused_attention_core_leaves_the_gradients_of_the_three_function_form(dev, B, N):
    """TransformerBlock (d_points 256, d_model 512, k 16) in training mode, forward + backward with train_ops.ATTN_CORE on and
    off from the same weights and inputs: identical outputs (the forward launches are the same), every gradient within 2e-6 of
    its largest element (the sums over the 16 neighbours are taken in another order)."""
    from ptt_amd import train_ops
    from ptt_amd.models.transformer_block.variants import TransformerBlock
    torch.manual_seed(7)
    block = TransformerBlock(256, 512, 16).to(dev).train()
    xyz = torch.randn(B, N, 3, device=dev)
    feats = torch.randn(B, N, 256, device=dev)
    gout = torch.randn(B, N, 256, device=dev)
    res = {}
    for mode in (True, False, True):
        train_ops.ATTN_CORE = mode
        try:
            for p in block.parameters():
                p.grad = None
            x, f = xyz.clone().requires_grad_(True), feats.clone().requires_grad_(True)
            assert train_ops.pt_block_usable(block, x, f)
            out, attn = block(x, f)
            (out * gout).sum().backward()
            grads = {k: p.grad.detach().clone() for k, p in block.named_parameters()}
            grads["xyz"], grads["features"] = x.grad.detach().clone(), f.grad.detach().clone()
            res.setdefault(mode, []).append((out.detach().clone(), attn.detach().clone(), grads))
        finally:
            train_ops.ATTN_CORE = True
    (o1, a1, g1), (o3, a3, g3) = res[True]
    o2, a2, g2 = res[False][0]
    assert torch.equal(o1, o2) and torch.equal(a1, a2)
    assert torch.equal(o1, o3) and all(torch.equal(g1[k], g3[k]) for k in g1)           # bit-reproducible
    worst = 0.0
    for k in g2:
        err = float((g1[k] - g2[k]).abs().max()) / max(float(g2[k].abs().max()), 1e-30)
        worst = max(worst, err)
        assert err < 2e-6, (k, err)
    print("fused attention core vs three functions at B = %d, N = %d: worst relative gradient difference %.2e" % (B, N, worst))
