"""bench.py on the GPU box: the ONE JSON line the driver parses, for the headline workload and for the training step — the contract
keys of the task (metric / value / unit / n_gpus / steps / warmup / ms_per_step / higher_is_better / scaling / vs_baseline / dtype /
data / config.workload), the `roofline` object (bound, achieved, peak, frac = achieved / peak, traffic) and, when asked for, the
`cpu_baseline` object; short runs (the numbers themselves are not asserted beyond sanity)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BASE = json.load(open(os.path.join(ROOT, "BASELINE.json")))


def _run(*flags):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(flags), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines                                     # ONE line on stdout
    return json.loads(lines[0])


def _contract(d, steps, warmup):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == steps and d["warmup"] == warmup and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and "synthetic" in d["data"]
    assert isinstance(d["config"].get("workload"), str) and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and 0.0 < r["frac"] < 1.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 2e-3 and "traffic" in r
    assert d["value"] > 0 and d["ms_per_step"] > 0


def test_headline_line_and_its_cpu_baseline():
    d = _run("--steps", "3", "--warmup", "1", "--sustain", "0", "--no-workloads", "--no-full-model", "--no-latency", "--cpu-frames", "2")
    _contract(d, 3, 1)
    assert "tracklet frames/sec" in d["metric"] and BASE["metric"].startswith("tracklet frames/sec")
    assert d["unit"] == "frames/s" and "2048 search + 1024 template" in d["config"]["workload"] and d["roofline"]["bound"] == "mfma"
    assert d["roofline"]["peak"] == 157.3 and d["roofline"]["kernel"].startswith("pt_attn_pair_kernel")
    batch = 48
    assert abs(d["value"] - batch / d["ms_per_step"] * 1e3) <= 1e-3 * d["value"]          # whole-job frames per second
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["value"] > 0 and c["cores"] >= 1 and isinstance(c["sample"], str) and c["unit"] == "frames/s"


def test_training_line():
    d = _run("--workload", "train", "--steps", "2", "--warmup", "1", "--sustain", "0", "--no-cpu-baseline")
    _contract(d, 2, 5)                        # the warm-up is raised to reach the first replay of the captured step
    assert "forward + backward" in d["metric"] and d["ms_per_step"] < 100.0
    assert d["config"]["graphs_per_step"] == 1 and d["host_issue_ms_per_step"] < 2.0          # the step is a hipGraph replay
    assert d["small_batch"]["frames_per_step"] == 8 and d["small_batch"]["graphs_per_step"] == 1 and d["small_batch"]["ms_per_step"] < 12.0
    r = d["roofline"]
    assert r["alg_bytes"] > 3e10 and (r["traffic"] is None or r["traffic"] > r["alg_bytes"])


def _torchrun_one_rank(script, *flags):
    """`python -m torch.distributed.run --nproc-per-node 1 ...`: the launcher a multi-GPU driver uses, with one rank."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, script)] + list(flags), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_the_rccl_branches_of_the_training_line_run_at_world_size_one():
    """What an 8-GPU launch runs, on this one-GPU box: init_process_group("nccl", device_id=...), the ranks-seen all-reduce,
    the tracker on the row kernels with its flat gradient buffer all-reduced once per step on the device (19.6 MB), and the
    no_sync() exposure measurement (tools/train_tracking.py:158-159, ptt/utils/common_utils.py:275-289)."""
    d = _torchrun_one_rank("bench.py", "--gpus", "1", "--workload", "train", "--steps", "2", "--warmup", "1", "--sustain", "0",
                           "--no-cpu-baseline", "--force-collective")
    _contract(d, 2, 5)
    assert d["config"]["graphs_per_step"] == 2              # forward + backward + finish | the eager all-reduce | clip + Adam
    assert d["rccl_ranks_seen"] == 1 and d["grad_bytes_allreduced_per_step"] == 4903113 * 4
    a = d["allreduce"]
    assert a["ms_per_step_without_allreduce"] > 0 and abs(a["exposed_ms_per_step"]) < d["ms_per_step"]
    assert "RCCL" in d["config"]["sharding"]


def test_a_launcher_started_car_line_is_followed_by_the_training_launch(monkeypatch):
    """`python -m torch.distributed.run ... bench.py --gpus N` (how a driver starts N > 1): after the car workload rank 0 launches the
    ranks again on the training step and carries its line as workloads.train — here forced at one rank (PTT_BENCH_TRAIN_AFTER)."""
    monkeypatch.setenv("PTT_BENCH_TRAIN_AFTER", "1")
    d = _torchrun_one_rank("bench.py", "--gpus", "1", "--steps", "3", "--warmup", "1", "--sustain", "0", "--no-cpu-baseline", "--no-full-model",
                           "--no-latency", "--workloads", "train")
    _contract(d, 3, 1)
    t = d["workloads"]["train"]
    assert "error" not in t and t["rccl_ranks_seen"] == 1 and t["config"]["graphs_per_step"] == 2 and t["value"] > 0
    assert t["grad_bytes_allreduced_per_step"] == 4903113 * 4


def test_ddp_on_one_rccl_rank_is_bit_identical_to_the_unwrapped_trainer():
    """Both reducers — the flat gradient buffer's one all-reduce (default) and DistributedDataParallel — on one RCCL rank: bit-identical
    to the same trainer without a collective; the two reducers' gradients agree to fp32 rounding (different summation trees)."""
    d = _torchrun_one_rank("scripts/rccl_one_rank_check.py")
    assert d["backend"] == "nccl" and d["world"] == 1 and d["ranks_seen"] == 1 and d["flat"] and d["ddp"] and d["unwrapped_is_plain"]
    assert d["grad_keys_equal"] and d["grads_bit_equal"] and d["params_bit_equal"] and d["loss_equal"], d
    assert d["graph_captured"] and d["graph_params_bit_equal"], d            # the captured step: two graphs around the eager all-reduce
    assert d["flat_vs_ddp_max_rel"] < 1e-5, d["flat_vs_ddp_max_rel"]
    assert d["grad_bytes_allreduced_per_step"] == d["expected_grad_bytes"] == 4903113 * 4


def test_the_default_line_fits_the_drivers_tail_and_carries_every_workload():
    """`python bench.py` exactly as the driver runs it (N = 1): ONE stdout line of at most 6144 characters — the driver keeps a tail
    of the output — that still holds the B = 1 tracklet latency and the ped / stress / train workloads."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       timeout=1500, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and len(lines[0]) < 6144, (len(lines), len(lines[0]))
    d = json.loads(lines[0])
    _contract(d, 20, 5)
    assert d["latency_b1"]["tracklet_loop"]["b1"]["ms_per_step"] > 0 and d["latency_b1"]["tracklet_loop"]["b48"]["frames_per_s"] > 0
    for name in ("ped", "stress", "train"):
        w = d["workloads"][name]
        assert "error" not in w and w["value"] > 0 and 0 < w["roofline"]["frac"] < 1, (name, w)
    assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["cores"] >= 1
    ws = d["whole_step"]                                                                 # what the hardware did, beside the effective rate
    assert 0 < ws["frac_of_mfma_peak_executed"] < ws["frac_of_mfma_peak"] < 1 and ws["executed_gflop_per_frame"] < ws["alg_gflop_per_frame"]
    t = d["workloads"]["train"]
    assert t["roofline"]["alg_bytes"] > 3e10 and t["host_issue_ms_per_step"] < 2.0 and t["small_batch"]["ms_per_step"] > 0
    assert any(l.startswith("[bench detail] {") for l in p.stderr.splitlines())          # the prose went to stderr
