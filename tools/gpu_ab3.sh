#!/bin/bash
# A (tools/ab/libfnn_A.so) against B (libfnn_B.so) and B under AB_ENV_B2, alternating on one box
export TMPDIR=/tmp
mkdir -p gpurun_out
for rep in 1 2 3; do
 for v in A B B2; do
  lib=$v; e=""
  if [ $v = B2 ]; then lib=B; e="$AB_ENV_B2"; fi
  env $e FNN_HIP_LIB=$PWD/tools/ab/libfnn_$lib.so timeout -k 10 300 python bench.py ${AB_ARGS} > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err || { echo "$v failed"; tail -3 gpurun_out/ab_$v.err; exit 1; }
  python - <<PY
import json
d = json.loads(open('gpurun_out/ab_$v.json').read().strip().splitlines()[-1])
print('$v [$e] rep $rep', 'ms/step %.4f' % d['ms_per_step'], {k: round(v * 1e3, 1) for k, v in d['kernel_ms'].items() if v and k != 'sort_now'})
PY
 done
done
