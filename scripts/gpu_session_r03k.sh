#!/bin/bash
for w in 2 3 2 3 2 3; do
  echo "== --ways $w"
  timeout 300 python bench.py --ways $w --steps 20 --warmup 5 --no-cpu-baseline --no-latency --no-full-model --sustain 2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('   bench', d['value'], d['ms_per_step'], d['sustained']['ms_per_step'])"
done
