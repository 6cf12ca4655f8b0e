#!/bin/bash
set -u
O=gpurun_out/r02i
mkdir -p $O
REPO=$(pwd)
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ktt && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktt -- \
    python $REPO/bench.py --workload train --steps 5 --warmup 2 --sustain 0 > $REPO/$O/train_prof.json 2> $REPO/$O/train_prof.err; \
    f=$(find /tmp/ktt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $REPO/$O/train_kernel_stats.csv)
head -32 $O/train_kernel_stats.csv | cut -c1-150
