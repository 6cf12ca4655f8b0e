"""The captured training step (train_step.DataParallelTrainer, graph mode: forward + backward + gradient finish + clip + Adam
replayed as a hipGraph) against the eager step it records — reference tools/train_utils/train_utils.py:44-51. The bar is
bit-identity: parameters, BatchNorm buffers, Adam moments, the flat gradient buffer, loss and clipped norm after every step."""
import time

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    return torch.device("cuda:0")


def _trainer(dev, graph, **kw):
    from ptt_amd.config import StubDataset, ptt_model_cfg
    from ptt_amd.models import build_network
    from ptt_amd.train_step import DataParallelTrainer
    torch.manual_seed(1)
    model = build_network(ptt_model_cfg(), 1, StubDataset(training=True)).to(dev).train()
    return DataParallelTrainer(model, dev, graph=graph, **kw)


def _same_state(a, b):
    bad = [k for (k, p), q in zip(a.tracker.state_dict().items(), b.tracker.state_dict().values()) if not torch.equal(p, q)]
    for p, q in zip(a.optimizer.param_groups[0]['params'], b.optimizer.param_groups[0]['params']):
        sa, sb = a.optimizer.state[p], b.optimizer.state[q]
        if not (torch.equal(sa['exp_avg'], sb['exp_avg']) and torch.equal(sa['exp_avg_sq'], sb['exp_avg_sq']) and float(sa['step']) == float(sb['step'])):
            bad.append("adam state")
            break
    if not torch.equal(a.sink.flat, b.sink.flat):
        bad.append("flat gradient buffer")
    if not torch.equal(a.optimizer.last_norm, b.optimizer.last_norm):
        bad.append("clipped norm")
    return bad


def test_replayed_step_is_bit_identical_to_the_eager_step(dev):
    """Two trainers from one seed in ONE process (they share the device's weight-pack plan: the capture must neither read what the
    eager twin rewrites nor leave entries behind that only a replay fills in), batches rotating, the learning rate changed mid-way
    (a scheduler's doing: the recorded launches read the step's hyper-parameters from device memory)."""
    from ptt_amd.train_step import synthetic_train_batch
    eager, graphed = _trainer(dev, False), _trainer(dev, True)
    batches = [synthetic_train_batch(100 + k, 8, dev) for k in range(3)]
    for k in range(9):
        if k == 6:
            for t in (eager, graphed):
                t.optimizer.param_groups[0]['lr'] *= 0.5
        le = eager.step(batches[k % 3]).detach().clone()
        lg = graphed.step(batches[k % 3]).detach().clone()
        torch.cuda.synchronize()
        assert torch.equal(le, lg) and bool(torch.isfinite(lg)), (k, float(le), float(lg))
        assert not _same_state(eager, graphed), (k, _same_state(eager, graphed))
        assert (graphed.captured is not None) == (k >= 3) and eager.captured is None
    assert graphed.graph_steps == 6 and graphed.eager_steps == 3 and graphed.captured.second is None
    assert int(graphed.tracker.global_step) == int(eager.tracker.global_step) if hasattr(eager.tracker, "global_step") else True


def test_queuing_a_replayed_step_costs_the_host_under_a_millisecond(dev):
    from ptt_amd.train_step import synthetic_train_batch
    tr = _trainer(dev, True)
    b = synthetic_train_batch(100, 8, dev)
    for _ in range(6):
        tr.step(b)
    torch.cuda.synchronize()
    assert tr.captured is not None
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(20):
            tr.step(b)
        best = min(best, (time.perf_counter() - t0) / 20)
        torch.cuda.synchronize()
    assert best < 1.0e-3, "host time to queue one replayed step: %.3f ms" % (best * 1e3)


def test_other_batch_shapes_step_eagerly_and_a_reloaded_optimizer_drops_the_capture(dev):
    from ptt_amd.train_step import synthetic_train_batch
    eager, graphed = _trainer(dev, False), _trainer(dev, True)
    b8, b4 = synthetic_train_batch(100, 8, dev), synthetic_train_batch(101, 4, dev)
    for k in range(5):
        eager.step(b8), graphed.step(b8)
    assert graphed.captured is not None and graphed.graph_steps == 2
    eager.step(b4), graphed.step(b4)                                    # another shape: eager, same numbers
    assert graphed.graph_steps == 2 and not _same_state(eager, graphed)
    eager.step(b8), graphed.step(b8)                                    # back on the captured shape
    assert graphed.graph_steps == 3 and not _same_state(eager, graphed)
    # optimizer.load_state_dict() makes new moment tensors: the recorded launches address the old ones -> the capture is dropped,
    # the next step runs eagerly (it rebuilds the optimizer's table), the one after is captured again
    for t in (eager, graphed):
        t.optimizer.load_state_dict(t.optimizer.state_dict())
    old = graphed.captured
    eager.step(b8), graphed.step(b8)
    assert graphed.captured is None or graphed.captured is not old
    eager.step(b8), graphed.step(b8)
    torch.cuda.synchronize()
    assert graphed.captured is not None and graphed.captured is not old and not _same_state(eager, graphed)


def test_graph_mode_refuses_what_it_cannot_capture(dev):
    with pytest.raises(ValueError):
        _trainer(dev, True, reducer="ddp")
    assert _trainer(dev, None, reducer="ddp").graph_mode is False


def test_replayed_step_without_clipping_equals_the_eager_step(dev):
    """clip = 0 (GRAD_NORM_CLIP unset): no norm pass is recorded, the update launch reads max_norm = 0 from the device struct."""
    from ptt_amd.train_step import synthetic_train_batch
    eager, graphed = _trainer(dev, False, clip=0), _trainer(dev, True, clip=0)
    b = synthetic_train_batch(100, 4, dev)
    for k in range(6):
        le, lg = eager.step(b).detach().clone(), graphed.step(b).detach().clone()
    torch.cuda.synchronize()
    assert graphed.graph_steps == 3 and torch.equal(le, lg)
    bad = [k for (k, p), q in zip(eager.tracker.state_dict().items(), graphed.tracker.state_dict().values()) if not torch.equal(p, q)]
    assert not bad and torch.equal(eager.sink.flat, graphed.sink.flat), bad[:4]
