set -x
mkdir -p gpurun_out
timeout 120 python scripts/probes/graph_dot_probe.py > gpurun_out/r04b_dot.log 2>&1
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 > gpurun_out/r04b_pytest.log
timeout 300 python scripts/tracklet_b1_profile.py > gpurun_out/r04b_b1.log 2>&1
cat gpurun_out/r04b_dot.log | head -60; cat gpurun_out/r04b_pytest.log gpurun_out/r04b_b1.log
