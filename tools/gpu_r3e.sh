#!/bin/bash
# round 3, session e: the stack split into wide (pairs) and narrow (16-row strips) launches: tests, A/B, kernel trace
export TMPDIR=/tmp
mkdir -p gpurun_out
step() {   # name, seconds, command...
  name=$1; secs=$2; shift 2
  echo "== $name"; timeout -k 10 $secs "$@" > gpurun_out/$name.log 2> gpurun_out/$name.err; rc=$?
  echo "   rc=$rc"; tail -c 400 gpurun_out/$name.log | tail -3
  if [ $rc -ge 124 ]; then echo "killed: stopping"; exit $rc; fi
}
step tests_ip 900 python -m pytest tests/test_gpu_ipnn.py tests/test_gpu_fullsize.py -q --timeout 600
B="--workload ipnn --steps 100 --warmup 10 --no-cpu-baseline"
step ip_split 300 python bench.py $B
step ip_nosplit 300 env IPNN_TAIL_SPLIT=0 python bench.py $B
step ip_split2 300 python bench.py $B
rm -rf gpurun_out/prof_ipnn
step prof_ipnn 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ipnn -o ip -- python3 bench.py $B
find gpurun_out/prof_ipnn -name "*kernel_stats.csv" -exec cp {} gpurun_out/ipnn_kernel_stats.csv \;
head -10 gpurun_out/ipnn_kernel_stats.csv | cut -c1-150
for f in ip_split ip_nosplit ip_split2; do grep -o '"ms_per_step": [0-9.]*' gpurun_out/$f.log | head -1; grep -o '"kernel_ms": {[^}]*}' gpurun_out/$f.log | head -1; done
