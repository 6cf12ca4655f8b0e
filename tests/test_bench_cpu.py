"""bench.py's launch logic on a GPU-less host: workload table vs BASELINE.json, and `--gpus 2` with WORLD_SIZE unset
really re-executes the script as 2 ranks under torch.distributed.run (each rank then refuses to run without a HIP
device — there is no CPU fallback — which is the observable proof that two ranks were started)."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_workloads_cover_the_baseline_configs():
    sys.path.insert(0, ROOT)
    import bench
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert len(base["configs"]) == 5
    refs = sorted(w["ref"] for w in bench.WORKLOADS.values())
    assert refs == ["BASELINE.json configs[%d]" % i for i in (1, 2, 3, 4)]      # configs[0] is the CPU plumbing case
    car = bench.WORKLOADS["car"]
    assert (car["batch"], car["ns"], car["nt"]) == (48, 2048, 1024)
    st = bench.WORKLOADS["stress"]
    assert (st["ns"], st["nt"], st["npoints_s"], st["npoints_t"]) == (16384, 4096, [8192, 4096, 2048], [2048, 1024, 512])
    assert bench.pair_kernel_flops(48, 128) + bench.pair_kernel_flops(48, 64) == 2 * 116190609408.0


@pytest.mark.skipif(torch.cuda.is_available(), reason="only meaningful where no HIP device is visible")
def test_gpus_2_spawns_two_ranks_itself():
    """`python bench.py --gpus 2` (what a driver runs at N > 1) launches the 2 ranks TWICE: the car headline, then the DDP
    training step of configs[3] whose line rides as workloads.train — on a GPU-less host every rank of both launches stops
    at "needs a HIP device", which is the observable proof that 2 x 2 ranks were started."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode != 0
    spawns = [l for l in p.stderr.splitlines() if "spawn: 2 ranks" in l]
    assert len(spawns) == 2 and "--workload train" in spawns[1] and "--workload train" not in spawns[0], p.stderr[-2000:]
    # one rank's exit makes torchrun SIGTERM its sibling, which may not have printed yet: at least one per launch
    assert p.stderr.count("bench.py needs a HIP device") >= 2, p.stderr[-2000:]


def test_launched_ranks_get_an_environment_without_the_outer_launchers_variables(monkeypatch):
    """bench.py started BY a launcher (python -m torch.distributed.run ... bench.py --gpus N) follows the car line with a launch of
    its own for the training step: the inner torch.distributed.run must not inherit the outer one's rendezvous."""
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    def run(cmd, env, timeout):
        seen["cmd"], seen["env"], seen["timeout"] = cmd, env, timeout
        return 0, b'{"value": 1.0, "config": {"workload": "w", "sharding": "s", "graphs_per_step": 2}, "roofline": {"frac": 0.5}}\n'
    for k, v in (("WORLD_SIZE", "8"), ("RANK", "0"), ("LOCAL_RANK", "0"), ("MASTER_PORT", "29400"), ("TORCHELASTIC_RUN_ID", "x"), ("OMP_NUM_THREADS", "1")):
        monkeypatch.setenv(k, v)
    monkeypatch.setattr(bench, "run_captured", run)
    rec = bench.train_launch(8)
    assert rec["value"] == 1.0 and rec["config"]["graphs_per_step"] == 2 and seen["timeout"] == bench.TRAIN_LAUNCH_TIMEOUT_S
    assert not any(k in seen["env"] for k in bench.LAUNCHER_ENV) and "TORCHELASTIC_RUN_ID" not in seen["env"]
    assert seen["cmd"][1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in seen["cmd"] and "train" in seen["cmd"]
    assert "--force-collective" not in seen["cmd"]
    bench.train_launch(1)
    assert "--force-collective" in seen["cmd"]                      # one rank: the collective branches are taken anyway


def test_gpus_2_with_another_workload_is_one_launch():
    """Only the default car line carries the train launch: `--workload train --gpus 2` itself is a single launch."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    if torch.cuda.is_available():
        pytest.skip("only meaningful where no HIP device is visible")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "train", "--steps", "1",
                        "--warmup", "0"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode != 0 and sum("spawn: 2 ranks" in l for l in p.stderr.splitlines()) == 1


def test_committed_traffic_reads_the_newest_pmc_summary_and_names_its_source():
    """roofline.traffic of the bench line: bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes
    (a process cannot PMC itself), with the file and the build it was taken from."""
    sys.path.insert(0, ROOT)
    import bench
    tr, src = bench.committed_traffic("pt_attn_pair_kernel<512>")
    assert tr is not None and src["file"].startswith("profiles/") and src["file"].endswith("pmc_summary.json")
    assert len(src["per_launch_shape"]) == 2                                  # the N = 128 and the N = 64 launch of a step
    per = [v["read_bytes"] + v["write_bytes"] for v in src["per_launch_shape"].values()]
    assert abs(tr - sum(per) / 2) < 1.0
    alg = 0.5 * (bench.pair_alg_bytes(48, 128) + bench.pair_alg_bytes(48, 64))
    assert alg < tr < 3 * alg                                                 # measured traffic is 1.8x the compulsory bytes
    assert bench.committed_traffic("no_such_kernel") == (None, None)


def test_side_workload_reports_a_failing_process_instead_of_raising(monkeypatch):
    """The default line's `workloads` entries come from child processes: a child that dies yields an `error` entry, the
    headline line is still printed."""
    sys.path.insert(0, ROOT)
    import bench

    class P(object):
        returncode, stdout, stderr = 139, b"", b"Segmentation fault"
    monkeypatch.setattr(bench.subprocess, "run", lambda *a, **k: P())
    rec = bench.side_workload("stress", 5, 2)
    assert "error" in rec and "139" in rec["error"]

    class Q(object):
        returncode, stderr = 0, b""
        stdout = json.dumps({"metric": "m", "value": 1.0, "unit": "frames/s", "steps": 5, "warmup": 2, "ms_per_step": 2.0, "dtype": "f32",
                             "roofline": {"frac": 0.5}, "config": {"workload": "w"}, "cpu_baseline": None}).encode()
    monkeypatch.setattr(bench.subprocess, "run", lambda *a, **k: Q())
    rec = bench.side_workload("ped", 10, 3)
    assert rec["value"] == 1.0 and rec["config"]["ref"] == "BASELINE.json configs[2]" and "cpu_baseline" not in rec


def test_the_stdout_line_is_compact_ordered_and_bounded():
    """compact_line: prose keys dropped, contract keys first, latency_b1 / workloads in front of the per-kernel tables, and at most
    LINE_LIMIT characters (the tables are dropped first when a line would not fit)."""
    sys.path.insert(0, ROOT)
    import bench
    out = {"kernel_ms_per_step": {"k%d" % i: 0.123456789 for i in range(8)}, "latency_b1": {"tracklet_loop": {"b1": {"ms_per_step": 0.61, "split": "x" * 900}}},
           "workloads": {"ped": {"value": 1.0, "process": "own process", "roofline": {"frac": 0.9, "traffic": None, "traffic_note": "y" * 500}}},
           "roofline": {"frac": 0.5, "traffic": None, "timing": "z" * 300, "traffic_source": {"file": "profiles/x", "per_launch_shape": {"a": 1}}},
           "cpu_baseline": None, "vs_baseline": None, "value": 12345.678912, "metric": "m", "config": {"workload": "w", "launch": "v" * 400},
           "index_ops": {"fps": {"note": "n" * 300, "alg_GBps": 5.8}}}
    line = bench.compact_line(out)
    keys = list(line)
    assert keys[:2] == ["metric", "value"] and keys.index("latency_b1") < keys.index("workloads") < keys.index("kernel_ms_per_step")
    text = json.dumps(line)
    assert len(text) < 900 and "xxxx" not in text and "yyyy" not in text and "zzzz" not in text and "vvvv" not in text and "nnnn" not in text
    assert line["roofline"]["traffic"] is None and line["cpu_baseline"] is None and line["vs_baseline"] is None      # contract keys stay
    assert line["roofline"]["traffic_source"] == {"file": "profiles/x"} and line["value"] == 12345.7
    out["kernel_ms_per_step"] = {"k%d" % i: 1.0 for i in range(2000)}
    small = bench.compact_line(out)
    assert len(json.dumps(small)) <= bench.LINE_LIMIT and "kernel_ms_per_step" not in small and "latency_b1" in small


def test_a_captured_launch_that_never_returns_is_killed_with_its_whole_process_group():
    """run_captured: the command runs in a process group of its own; past the timeout the group (a launcher AND what it started) is
    killed and the caller gets exit code -9 with whatever was printed."""
    sys.path.insert(0, ROOT)
    import time
    import bench
    code = ("import subprocess, sys, time; print('{\"started\": 1}', flush=True); "
            "subprocess.Popen([sys.executable, '-c', 'import time; time.sleep(600)']); time.sleep(600)")
    t0 = time.time()
    rc, out = bench.run_captured([sys.executable, "-c", code], dict(os.environ), 3)
    assert rc == -9 and b"started" in out and time.time() - t0 < 30
    rc, out = bench.run_captured([sys.executable, "-c", "print('{}')"], dict(os.environ), 30)
    assert rc == 0 and out.strip() == b"{}"
