#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
for w in 144 72 288 576; do
  IPNN_WGRAD_WANT=$w timeout -k 10 200 python bench.py --workload ipnn --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/ipnn_v.json 2> gpurun_out/ipnn_v.err
  python - <<PY
import json
d = json.loads(open('gpurun_out/ipnn_v.json').read().strip().splitlines()[-1])
print('want $w', 'ms/step %.4f' % d['ms_per_step'], {k: round(v * 1e3, 1) for k, v in d['kernel_ms'].items() if k in ('fwd', 'bwd', 'wgrad', 'update')})
PY
done
timeout -k 10 600 python -m pytest tests/test_gpu_ipnn.py -m gpu -q --timeout 600 2>&1 | tail -2
