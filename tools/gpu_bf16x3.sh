#!/bin/bash
# the bf16-pair precision: its tests, then the three precisions of the FNN step side by side
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_bf16x3.py -m gpu -q -s --timeout 600 > gpurun_out/bf16x3_tests.log 2>&1; rc=$?
grep -E "demo deltas|max \|p|passed|failed|Error|error" gpurun_out/bf16x3_tests.log | tail -12
if [ $rc -ge 124 ]; then exit $rc; fi
for prec in bf16 bf16x3 f32; do
  timeout -k 10 200 python bench.py --precision $prec --no-extras --no-cpu-baseline --steps 300 > gpurun_out/fnn_$prec.json 2> gpurun_out/fnn_$prec.err; rc=$?
  if [ $rc -ge 124 ]; then exit $rc; fi
done
timeout -k 10 200 python bench.py --workload snn --precision bf16x3 --no-cpu-baseline --steps 300 > gpurun_out/snn_bf16x3.json 2> gpurun_out/snn_bf16x3.err
python - <<'PY'
import json
for f in ('fnn_bf16', 'fnn_bf16x3', 'fnn_f32', 'snn_bf16x3'):
    try:
        d = json.loads(open('gpurun_out/%s.json' % f).read().strip().splitlines()[-1])
        print(f, 'ms/step %.4f' % d['ms_per_step'], '%.2f M ex/s' % (d['value'] / 1e6), {k: round(v * 1e3, 1) for k, v in d['kernel_ms'].items() if v}, d['roofline']['frac'])
    except Exception as e:
        print(f, 'ERR', e, open('gpurun_out/%s.err' % f).read()[-600:])
PY
