#!/bin/bash
# round 3, session d: the whole GPU suite again (new full-shape anchors, RBM DP), kernel traces of the rbm and ipnn workloads
export TMPDIR=/tmp
mkdir -p gpurun_out
step() {   # name, seconds, command...
  name=$1; secs=$2; shift 2
  echo "== $name"; timeout -k 10 $secs "$@" > gpurun_out/$name.log 2> gpurun_out/$name.err; rc=$?
  echo "   rc=$rc"; tail -c 500 gpurun_out/$name.log | tail -4
  if [ $rc -ge 124 ]; then echo "killed: stopping"; exit $rc; fi
}
step gpu_tests 1100 python -m pytest tests -m gpu -q --timeout 600
rm -rf gpurun_out/prof_rbm gpurun_out/prof_ipnn
step prof_rbm 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_rbm -o rbm -- python3 bench.py --workload rbm --no-cpu-baseline
find gpurun_out/prof_rbm -name "*kernel_stats.csv" -exec cp {} gpurun_out/rbm_kernel_stats.csv \;
head -14 gpurun_out/rbm_kernel_stats.csv | cut -c1-150
step prof_ipnn 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ipnn -o ip -- python3 bench.py --workload ipnn --steps 100 --warmup 10 --no-cpu-baseline
find gpurun_out/prof_ipnn -name "*kernel_stats.csv" -exec cp {} gpurun_out/ipnn_kernel_stats.csv \;
head -14 gpurun_out/ipnn_kernel_stats.csv | cut -c1-150
step ipnn_stamps 300 env IPNN_STAMPS=1 python bench.py --workload ipnn --steps 20 --warmup 5 --no-cpu-baseline
tail -12 gpurun_out/ipnn_stamps.err | cut -c1-400
