#!/bin/bash
# rocprofv3 kernel stats of the FNN step in the three precisions (headline leg only), and the default bench line
export TMPDIR=/tmp
mkdir -p gpurun_out
step() { name=$1; secs=$2; shift 2; echo "== $name"; timeout -k 10 $secs "$@" > gpurun_out/$name.log 2> gpurun_out/$name.err; rc=$?; echo "   rc=$rc"; if [ $rc -ge 124 ]; then echo killed; exit $rc; fi; }
for prec in bf16 bf16x3 f32; do
  rm -rf gpurun_out/prof_$prec
  step prof_$prec 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$prec -o fnn -- python3 bench.py --precision $prec --steps 400 --warmup 20 --no-cpu-baseline --no-extras
  find gpurun_out/prof_$prec -name "*kernel_stats.csv" -exec cp {} gpurun_out/fnn_${prec}_kernel_stats.csv \;
  head -4 gpurun_out/fnn_${prec}_kernel_stats.csv | cut -c1-150
done
( time timeout -k 10 500 python bench.py > gpurun_out/bench_default.log 2> gpurun_out/bench_default.err ) 2> gpurun_out/bench_default.time
tail -3 gpurun_out/bench_default.time
python - <<'PY'
import json
d = json.loads(open('gpurun_out/bench_default.log').read().strip().splitlines()[-1])
print('default: %.2f M ex/s' % (d['value'] / 1e6), 'f32 %.2f M' % (d['precision_f32']['value'] / 1e6), 'bf16x3 %.2f M' % (d['precision_bf16x3']['value'] / 1e6),
      {k: (v.get('value') if isinstance(v, dict) else v) for k, v in d['extra_workloads'].items()})
PY
