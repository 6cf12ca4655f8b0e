"""The parameter gradients of a training step finished by ONE launch (train_ops.GradSink, ops.GradFinishPlan,
ptt_grad_finish_f32 / ptt_linear_wgrad*_partials_f32 / ptt_colsum_partials_f32) against the per-weight finished form the
functions return without a sink — what loss.backward() leaves in every .grad in the reference
(tools/train_utils/train_utils.py:47-49)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
SEED = 41


def test_grad_finish_sums_every_job_of_a_destination_in_a_fixed_order(dev):
    """Destinations of every kind the training step has — whole parameters, column slices of a 2-D weight (scalar path),
    float4-addressable and odd sizes — with 1 ... 1500 chunks spread over one to three jobs; against float64 sums, twice."""
    from ptt_amd import ops
    g = torch.Generator(device="cpu").manual_seed(5)
    total = 0
    dests = []                               # (dst, cols, ld, n)
    for n, cols, ld in [(512 * 512, 512 * 512, 512 * 512), (64 * 3, 3, 67), (64 * 64, 64, 67), (128, 128, 128), (1, 1, 1), (259 * 4, 259 * 4, 259 * 4),
                        (256 * 12, 12, 16), (5, 5, 5)]:
        span = (n // cols - 1) * ld + cols
        dests.append((total, cols, ld, n))
        total += (span + 3) // 4 * 4
    flat0 = torch.randn(total, generator=g).to(dev)
    chunk_sets = [[64, 64], [300], [1, 1, 1], [7], [1], [1500, 37], [2, 9], [3]]
    jobs, keep, want = [], [], flat0.double().clone()
    for (dst, cols, ld, n), chunks in zip(dests, chunk_sets):
        for nch in chunks:
            part = torch.randn(nch, n, generator=g).to(dev)
            keep.append(part)
            jobs.append((dst, cols, ld, n, part.data_ptr(), nch))
            idx = dst + (torch.arange(n, device=dev) // cols) * ld + torch.arange(n, device=dev) % cols
            want[idx] += part.double().sum(0)
    order = list(range(len(jobs)))
    order = order[1::2] + order[0::2]        # jobs of one destination need not be adjacent in issue order
    jobs = [jobs[k] for k in order]
    plan = ops.GradFinishPlan(dev)
    outs = []
    for _ in range(2):
        flat = flat0.clone()
        plan.run(jobs, flat)
        outs.append(flat)
    err = float((outs[0].double() - want).abs().max() / want.abs().max())
    assert err < 2e-6, err
    assert torch.equal(outs[0], outs[1])
    # untouched: the padding between destinations and the gaps of strided ones
    touched = torch.zeros(total, dtype=torch.bool, device=dev)
    for dst, cols, ld, n in dests:
        touched[dst + (torch.arange(n, device=dev) // cols) * ld + torch.arange(n, device=dev) % cols] = True
    assert torch.equal(outs[0][~touched], flat0[~touched])
    with pytest.raises(ValueError):
        plan.run([(total - 2, 4, 4, 4, keep[0].data_ptr(), 1)], flat0.clone())
    # a weight handed over whole AND as a column slice: two workgroups would add into the same elements without atomics
    with pytest.raises(ValueError, match="overlap"):
        plan.run([(0, 64 * 67, 64 * 67, 64 * 67, keep[0].data_ptr(), 1), (0, 3, 67, 64 * 3, keep[2].data_ptr(), 1)], flat0.clone())
    # column slices of one weight that tile its rows side by side are fine ([:, :3] and [:, 3:67] of a 64 x 67 weight)
    a, b = torch.randn(2, 64 * 3, generator=g).to(dev), torch.randn(1, 64 * 64, generator=g).to(dev)
    flat = torch.zeros(64 * 67 + 4, device=dev)
    plan.run([(0, 3, 67, 64 * 3, a.data_ptr(), 2), (3, 64, 67, 64 * 64, b.data_ptr(), 1)], flat)
    w = flat[:64 * 67].view(64, 67)
    assert torch.allclose(w[:, :3], (a[0] + a[1]).view(64, 3)) and torch.equal(w[:, 3:], b.view(64, 64))


def test_partial_sums_entries_match_the_finished_kernels(dev):
    from ptt_amd import ops
    torch.manual_seed(3)
    for R, Cout, Cin in [(98304, 512, 512), (4096, 128, 256), (50000, 64, 64), (70001, 128, 3), (1031, 259, 256), (300, 5, 256)]:
        dz, x = torch.randn(R, Cout, device=dev), torch.randn(R, Cin, device=dev)
        ws, nch = ops.linear_wgrad_partials(dz, x)
        part = ws.view(torch.float32)[:nch * Cout * Cin].view(nch, Cout, Cin)
        ref = ops.linear_wgrad(dz, x)
        got = part.double().sum(0)
        assert float((got - ref.double()).abs().max() / ref.abs().max()) < 1e-6, (R, Cout, Cin)
        ws, nch = ops.colsum_partials(dz)
        part = ws.view(torch.float32)[:nch * Cout].view(nch, Cout)
        ref = ops.colsum(dz)
        assert float((part.double().sum(0) - ref.double()).abs().max() / ref.abs().max()) < 1e-6, (R, Cout)


def _build(dev):
    from ptt_amd.config import StubDataset, ptt_model_cfg
    from ptt_amd.models import build_network
    from tests.util import fill_state_dict_
    return fill_state_dict_(build_network(ptt_model_cfg(), 1, StubDataset(training=True)), SEED).to(dev).train()


@pytest.mark.parametrize("B", [2, 12])
def test_the_sink_leaves_the_gradients_autograd_leaves(dev, B):
    """The same model and batch through the trainer with the flat gradient buffer and without it (finished gradients returned by
    every function, shared weights added by autograd): the forward passes are bit-identical, so the gradients differ only in the
    order of their sums. Every .grad is a view of the one buffer, the step is bit-reproducible, and clip + Adam on the views
    leaves the parameters the per-tensor form leaves."""
    from ptt_amd import train_ops
    from ptt_amd.train_step import DataParallelTrainer, synthetic_train_batch
    batch = synthetic_train_batch(100, B, dev)
    res = {}
    for reducer in ("flat", "ddp", "flat"):
        trainer = DataParallelTrainer(_build(dev), dev, reducer=reducer)
        assert (trainer.sink is not None) == (reducer == "flat") and train_ops.GradSink.active is None
        loss = trainer.forward_backward(batch)
        grads = {k: p.grad.detach().clone() for k, p in trainer.tracker.named_parameters()}
        if reducer == "flat":
            lo, hi = trainer.sink.flat.data_ptr(), trainer.sink.flat.data_ptr() + trainer.sink.flat.numel() * 4
            assert all(lo <= p.grad.data_ptr() < hi for p in trainer.tracker.parameters())
            assert not trainer.sink.jobs and not trainer.sink.keep
        trainer.step(batch)
        params = {k: p.detach().clone() for k, p in trainer.tracker.named_parameters()}
        res.setdefault(reducer, []).append((float(loss.detach()), grads, params))
    (l0, g0, p0), (l2, g2, p2) = res["flat"]
    l1, g1, p1 = res["ddp"][0]
    assert l0 == l1 == l2
    assert all(torch.equal(g0[k], g2[k]) for k in g0) and all(torch.equal(p0[k], p2[k]) for k in p0)        # bit-reproducible
    gmax = max(float(v.abs().max()) for v in g1.values())
    worst = 0.0
    for k in g1:
        scale = max(float(g1[k].abs().max()), 1e-3 * gmax)
        err = float((g0[k] - g1[k]).abs().max()) / scale
        worst = max(worst, err)
        assert err < 1e-5, (k, err)
    # clip + Adam over the views of the flat buffer = torch's clip_grad_norm_ + Adam on those gradients
    ref = _build(dev)
    for k, p in ref.named_parameters():
        p.grad = g0[k].clone()
    torch.nn.utils.clip_grad_norm_(ref.parameters(), 10.0)
    torch.optim.Adam(ref.parameters(), lr=1e-3, betas=(0.5, 0.999), eps=1e-6).step()
    pw = max(float((p0[k] - p.detach()).abs().max()) for k, p in ref.named_parameters())
    assert pw < 2e-6, pw
    print("flat gradient buffer vs per-weight gradients at B = %d: worst relative gradient difference %.2e; parameters after clip + "
          "Adam vs torch on the same gradients %.2e" % (B, worst, pw))


def test_a_failed_backward_leaves_no_sink_behind(dev):
    from ptt_amd import train_ops
    model = torch.nn.Linear(8, 8).to(dev)
    sink = train_ops.GradSink(list(model.parameters()), dev)
    with pytest.raises(RuntimeError):
        with sink.collecting():
            raise RuntimeError("backward failed")
    assert train_ops.GradSink.active is None and not sink.jobs
    with sink.collecting():                                 # plain autograd accumulates into the views in place
        model(torch.ones(3, 8, device=dev)).sum().backward()
    sink.flush()
    assert float(model.bias.grad.min()) == 3.0 and model.bias.grad.data_ptr() == sink.views[1].data_ptr()
