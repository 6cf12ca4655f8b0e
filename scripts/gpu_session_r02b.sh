#!/bin/bash
set -u
O=gpurun_out/r02b
mkdir -p $O
timeout 900 python -m pytest tests/test_tracking_gpu.py -x -q > $O/pytest_tracking.log 2>&1; echo "rc=$?" >> $O/pytest_tracking.log
tail -30 $O/pytest_tracking.log
