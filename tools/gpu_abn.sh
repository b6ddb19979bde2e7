#!/bin/bash
# A/B/... of several builds of the library on ONE box: tools/ab/libfnn_<V>.so for V in $AB_VARIANTS, alternating, AB_ARGS to bench.py
export TMPDIR=/tmp
mkdir -p gpurun_out
for rep in 1 2 3; do
 for v in ${AB_VARIANTS:-A B}; do
  FNN_HIP_LIB=$PWD/tools/ab/libfnn_$v.so timeout -k 10 300 python bench.py ${AB_ARGS:---no-extras --no-cpu-baseline --steps 400} > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err || { echo "$v failed"; tail -3 gpurun_out/ab_$v.err; exit 1; }
  python - <<PY
import json
d = json.loads(open('gpurun_out/ab_$v.json').read().strip().splitlines()[-1])
print('$v rep $rep', 'ms/step %.4f' % d['ms_per_step'], {k: round(v * 1e3, 1) for k, v in d['kernel_ms'].items() if v and k != 'sort_now'})
PY
 done
done
