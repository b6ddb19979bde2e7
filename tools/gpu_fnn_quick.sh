#!/bin/bash
# FNN step: parity tests of the step, then the headline bench without the extra legs
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dp.py tests/test_gpu_golden.py -m gpu -q --timeout 600 2>&1 | tail -3
timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline --steps 400 > gpurun_out/fnn_quick.json 2> gpurun_out/fnn_quick.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/fnn_quick.json').read().strip().splitlines()[-1])
print('ms/step %.4f' % d['ms_per_step'], '%.2f M ex/s' % (d['value'] / 1e6), {k: round(v * 1e3, 1) for k, v in d['kernel_ms'].items() if v})
PY
