#!/bin/bash
# Round 5, session c: the pure-torch reproduction for the record (1 hardware queue: crash; 4: pass), and the mitigation (8 queues) over
# seven rounds of car -> ped -> stress in one process
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05c; mkdir -p $O; : > $O/repro4.log
r() { echo "== $@" >> $O/repro4.log; env "$@" >> $O/repro4.log 2>&1; echo "rc=$?" >> $O/repro4.log; }
for q in 1 1 2 4 4; do for b in 2 3; do r GPU_MAX_HW_QUEUES=$q timeout 120 python scripts/probes/graph_queue_repro.py $b 0 drop null; done; done
r GPU_MAX_HW_QUEUES=8 PROBE_ROUNDS=7 timeout 1200 python scripts/probes/graph_sequence_probe.py bench car,ped,stress
r GPU_MAX_HW_QUEUES=8 PROBE_ROUNDS=3 timeout 1200 python scripts/probes/graph_sequence_probe.py bench car,ped,stress,train
grep -E "^==|rc=|PASSED|Fatal" $O/repro4.log | cut -c1-170
