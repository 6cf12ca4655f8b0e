#!/bin/bash
O=gpurun_out/r03f; mkdir -p $O; REPO=$(pwd)
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tl && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- \
   python $REPO/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-latency --no-full-model --sustain 0 > $REPO/$O/bench.json 2> $REPO/$O/bench.err; \
   f=$(find /tmp/tl -name "*kernel_trace.csv" | head -1); echo $f; python $REPO/scripts/step_timeline.py $f; head -2 $f | cut -c1-400)
