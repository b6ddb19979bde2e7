#!/bin/bash
# diagnostic: per-launch device time of the FNN L3 step with roles of launches 2/3 switched off
# (FNN_ROLE_OFF bits: 1 sort, 2 dense, 4 sparse).  Results of such runs are wrong by construction.
for m in 0 1 2 4 3 5 6; do
  FNN_ROLE_OFF=$m python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
k = d['kernel_ms']
print('role_off=$m', 'ms/step %.4f' % d['ms_per_step'], ' '.join('%s %.4f' % (n, v) for n, v in k.items() if v > 0))
"
done
