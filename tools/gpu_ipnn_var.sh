#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_ipnn.py tests/test_gpu_fullsize.py -m gpu -q --timeout 600 2>&1 | tail -3
for rep in 1 2; do
for cfg in "IPNN_FUSE_MASK=1" "IPNN_FUSE_MASK=0"; do
  env $cfg timeout -k 10 200 python bench.py --workload ipnn --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/ipnn_v.json 2> gpurun_out/ipnn_v.err
  python - <<PY
import json
d = json.loads(open('gpurun_out/ipnn_v.json').read().strip().splitlines()[-1])
print('$cfg', 'ms/step %.4f' % d['ms_per_step'], {k: round(v * 1e3, 1) for k, v in d['kernel_ms'].items() if k in ('ip_fwd', 'fwd', 'bwd', 'wgrad', 'update')}, 'loss', d['train_logloss_last_step'])
PY
done
done
