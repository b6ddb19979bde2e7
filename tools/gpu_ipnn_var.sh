#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
for m in 5 8 11 14; do
  IPNN_DUO_MIN=$m timeout -k 10 200 python bench.py --workload ipnn --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/ipnn_v.json 2> gpurun_out/ipnn_v.err
  python - <<PY
import json
d = json.loads(open('gpurun_out/ipnn_v.json').read().strip().splitlines()[-1])
print('min_blocks $m', 'ms/step %.4f' % d['ms_per_step'], {k: round(v * 1e3, 1) for k, v in d['kernel_ms'].items() if k in ('fwd', 'bwd')})
PY
done
