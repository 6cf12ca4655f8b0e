#!/bin/bash
mkdir -p gpurun_out/r02p
timeout 900 python -m pytest tests/test_golden_gpu.py tests/test_dense_gpu.py tests/test_point_ops_gpu.py -x -q 2>&1 | tail -12
