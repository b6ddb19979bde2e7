#!/bin/bash
# A/B/... of environment knobs on ONE box, alternating: AB_ENVS = "NAME=v NAME2=v;NAME=w;..." (';' between variants, "-" = nothing set),
# AB_ARGS to bench.py; AB_TESTS (optional): a pytest selection run under every variant first
export TMPDIR=/tmp
mkdir -p gpurun_out
IFS=';' read -ra VARS <<< "${AB_ENVS:--}"
if [ -n "$AB_TESTS" ]; then
 for e in "${VARS[@]}"; do
  [ "$e" = "-" ] && e=""
  env $e timeout -k 10 500 python -m pytest $AB_TESTS -m gpu -q -x > gpurun_out/ab_tests.log 2>&1 || { echo "tests under '$e' failed"; tail -30 gpurun_out/ab_tests.log; exit 1; }
  echo "tests under '$e': $(tail -1 gpurun_out/ab_tests.log)"
 done
fi
for rep in 1 2 3; do
 i=0
 for e in "${VARS[@]}"; do
  [ "$e" = "-" ] && e=""
  i=$((i+1))
  env $e timeout -k 10 300 python bench.py ${AB_ARGS:---no-extras --no-cpu-baseline --steps 400} > gpurun_out/ab_$i.json 2> gpurun_out/ab_$i.err || { echo "'$e' failed"; tail -3 gpurun_out/ab_$i.err; exit 1; }
  python - <<PY
import json
d = json.loads(open('gpurun_out/ab_$i.json').read().strip().splitlines()[-1])
print('[$e] rep $rep', 'ms/step %.4f' % d['ms_per_step'], {k: round(v * 1e3, 1) for k, v in d['kernel_ms'].items() if v and k != 'sort_now'})
PY
 done
done
