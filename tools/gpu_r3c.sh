#!/bin/bash
# round 3, session c: own radix sort (group_global, RBM mini-batches), the DP tests again, RBM bench (sorted vs atomics)
export TMPDIR=/tmp
mkdir -p gpurun_out
step() {   # name, seconds, command...
  name=$1; secs=$2; shift 2
  echo "== $name"; timeout -k 10 $secs "$@" > gpurun_out/$name.log 2> gpurun_out/$name.err; rc=$?
  echo "   rc=$rc"; tail -c 700 gpurun_out/$name.log | tail -5
  if [ $rc -ge 124 ]; then echo "killed: stopping"; exit $rc; fi
}
step tests_c 900 python -m pytest tests/test_gpu_rbm.py tests/test_gpu_dp.py -q --timeout 600
step rbm_sorted 400 python bench.py --workload rbm --no-cpu-baseline
step rbm_atomics 400 env RBM_BATCH_ATOMICS=1 python bench.py --workload rbm --no-cpu-baseline
