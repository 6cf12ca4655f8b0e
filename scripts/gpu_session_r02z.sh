#!/bin/bash
# Round-2 closing evidence: parity tests, the four bench workloads, a serial kernel trace, PMC passes, the training profile.
set -u
O=gpurun_out/r02z
mkdir -p $O
REPO=$(pwd)
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
timeout 600 python bench.py > $O/bench_car.json 2> $O/bench_car.err; tail -c 300 $O/bench_car.json
timeout 600 python bench.py --workload ped > $O/bench_ped.json 2> $O/bench_ped.err
timeout 900 python bench.py --workload stress --steps 5 --warmup 2 > $O/bench_stress.json 2> $O/bench_stress.err
timeout 600 python bench.py --workload train --steps 10 --warmup 3 > $O/bench_train.json 2> $O/bench_train.err
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- \
    python $REPO/bench.py --serial --steps 10 --warmup 3 --no-cpu-baseline --sustain 0 > $REPO/$O/serial_bench.json 2> $REPO/$O/serial_bench.err; \
    f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $REPO/$O/serial_kernel_stats.csv)
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ktt && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktt -- \
    python $REPO/bench.py --workload train --steps 5 --warmup 2 --sustain 0 > $REPO/$O/train_prof.json 2> $REPO/$O/train_prof.err; \
    f=$(find /tmp/ktt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $REPO/$O/train_kernel_stats.csv)
bash scripts/pmc_passes.sh $O/pmc "pair,sa0_s,sa1_s,sa2_s,sa_box" > $O/pmc.log 2>&1
tail -12 $O/pmc.log | cut -c1-330
