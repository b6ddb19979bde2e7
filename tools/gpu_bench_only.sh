#!/bin/bash
# the default bench line (with every leg) under the clock, then the end-to-end workload at its full size
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout -k 10 500 python bench.py > gpurun_out/bench_default.log 2> gpurun_out/bench_default.err ) 2> gpurun_out/bench_default.time || { echo "default bench failed"; tail -5 gpurun_out/bench_default.err; exit 1; }
tail -4 gpurun_out/bench_default.time
timeout -k 10 400 python bench.py --workload e2e > gpurun_out/bench_e2e.log 2> gpurun_out/bench_e2e.err || { echo "e2e failed"; tail -5 gpurun_out/bench_e2e.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open('gpurun_out/bench_default.log').read().strip().splitlines()[-1])
print('default: %.2f M ex/s' % (d['value'] / 1e6), {k: (v.get('value') if isinstance(v, dict) else v) for k, v in d['extra_workloads'].items()})
e = json.loads(open('gpurun_out/bench_e2e.log').read().strip().splitlines()[-1])
print('e2e:', e['value'], e.get('phases_s') or {k: v for k, v in e.items() if 'parse' in k or 'phase' in k})
PY
