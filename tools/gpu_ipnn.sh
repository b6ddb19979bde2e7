#!/bin/bash
# inner-product family: parity tests, then the FNN_IP_L7 step with two workgroups per strip and with one
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_ipnn.py tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -q --timeout 600 -k "ipnn or fullsize or full_table or two_features or same_row or shadowed" > gpurun_out/ipnn_tests.log 2>&1; rc=$?
tail -8 gpurun_out/ipnn_tests.log
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 200 python bench.py --workload ipnn --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/ipnn_duo.json 2> gpurun_out/ipnn_duo.err; rc=$?; echo "duo rc=$rc"
if [ $rc -ge 124 ]; then exit $rc; fi
IPNN_STRIP_DUO=0 timeout -k 10 200 python bench.py --workload ipnn --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/ipnn_solo.json 2> gpurun_out/ipnn_solo.err; echo "solo rc=$?"
python - <<'PY'
import json
for f in ('ipnn_duo', 'ipnn_solo'):
    try:
        d = json.loads(open('gpurun_out/%s.json' % f).read().strip().splitlines()[-1])
        print(f, 'ms/step %.4f' % d['ms_per_step'], '%.2f M ex/s' % (d['value'] / 1e6), {k: round(v * 1e3, 1) for k, v in d['kernel_ms'].items()}, 'loss', d['train_logloss_last_step'])
    except Exception as e:
        print(f, 'ERR', e, open('gpurun_out/%s.err' % f).read()[-600:])
PY
