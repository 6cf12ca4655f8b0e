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
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode != 0
    assert p.stderr.count("bench.py needs a HIP device") >= 2, p.stderr[-2000:]
