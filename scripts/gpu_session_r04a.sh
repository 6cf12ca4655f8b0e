set -x
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_rowjobs_gpu.py -x -q 2>&1 | tail -15 > gpurun_out/r04a_rowjobs_test.log
timeout 600 python scripts/probes/launch_floor_probe.py > gpurun_out/r04a_launch_floor.log 2>&1
timeout 300 python scripts/tracklet_b1_profile.py > gpurun_out/r04a_b1_baseline.log 2>&1
cat gpurun_out/r04a_rowjobs_test.log gpurun_out/r04a_launch_floor.log gpurun_out/r04a_b1_baseline.log
