#!/bin/bash
# round 3, session h: DAE + IP tests, mask prefetch A/B, rbm kernel trace
export TMPDIR=/tmp
mkdir -p gpurun_out
step() {   # name, seconds, command...
  name=$1; secs=$2; shift 2
  echo "== $name"; timeout -k 10 $secs "$@" > gpurun_out/$name.log 2> gpurun_out/$name.err; rc=$?
  echo "   rc=$rc"; tail -c 400 gpurun_out/$name.log | tail -3
  if [ $rc -ge 124 ]; then echo "killed: stopping"; exit $rc; fi
}
step tests_h 900 python -m pytest tests/test_gpu_dae.py tests/test_gpu_ipnn.py -q --timeout 600
B="--workload ipnn --steps 100 --warmup 10 --no-cpu-baseline"
step ip_pf 300 python bench.py $B
step ip_nopf 300 env IPNN_BENCH_NOPREFETCH=1 python bench.py $B
for f in ip_pf ip_nopf; do grep -o '"ms_per_step": [0-9.]*' gpurun_out/$f.log | head -1; done
rm -rf gpurun_out/prof_rbm
step prof_rbm 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_rbm -o rbm -- python3 bench.py --workload rbm --no-cpu-baseline
find gpurun_out/prof_rbm -name "*kernel_stats.csv" -exec cp {} gpurun_out/rbm_kernel_stats.csv \;
grep "k_rbm\|k_rs_\|k_group" gpurun_out/rbm_kernel_stats.csv | cut -d, -f1-4 | cut -c1-60,100-200
