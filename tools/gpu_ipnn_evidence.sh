#!/bin/bash
# the inner-product step's evidence on the final code: bench line, rocprofv3 kernel stats, stamps of the wide launches and of the tail
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 300 python bench.py --workload ipnn --steps 300 > gpurun_out/ipnn_bench.json 2> gpurun_out/ipnn_bench.err || { tail -3 gpurun_out/ipnn_bench.err; exit 1; }
rm -rf gpurun_out/prof_ipnn
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ipnn -o ipnn -- python3 bench.py --workload ipnn --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/prof_ipnn.log 2> gpurun_out/prof_ipnn.err || { tail -3 gpurun_out/prof_ipnn.err; exit 1; }
find gpurun_out/prof_ipnn -name "*kernel_stats.csv" -exec cp {} gpurun_out/ipnn_kernel_stats.csv \;
cut -c1-150 gpurun_out/ipnn_kernel_stats.csv | head -14
STAMP_SELS="0" bash tools/gpu_ipnn_stamps.sh > /dev/null
cat gpurun_out/ipnn_stamps.txt | cut -c1-400
python - <<'PY'
import json
d = json.loads(open('gpurun_out/ipnn_bench.json').read().strip().splitlines()[-1])
print('ms/step %.4f' % d['ms_per_step'], 'value %.3e' % d['value'], json.dumps(d['roofline'])[:600])
PY
