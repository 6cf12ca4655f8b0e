#!/bin/bash
set -u
O=gpurun_out/r02g
mkdir -p $O
timeout 900 python -m pytest tests/test_train_gpu.py -x -q > $O/pytest_train.log 2>&1; echo "rc=$?" >> $O/pytest_train.log
tail -30 $O/pytest_train.log
timeout 600 python bench.py --workload train --steps 10 --warmup 3 > $O/bench_train.json 2> $O/bench_train.err; tail -c 900 $O/bench_train.json; tail -3 $O/bench_train.err
