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
    _contract(d, 2, 1)
    assert "forward + backward" in d["metric"] and d["ms_per_step"] < 100.0
