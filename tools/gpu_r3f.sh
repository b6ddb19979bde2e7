#!/bin/bash
# round 3, session f: RBM mini-batch with the scratch arena; kernel trace of it
export TMPDIR=/tmp
mkdir -p gpurun_out
step() {   # name, seconds, command...
  name=$1; secs=$2; shift 2
  echo "== $name"; timeout -k 10 $secs "$@" > gpurun_out/$name.log 2> gpurun_out/$name.err; rc=$?
  echo "   rc=$rc"; tail -c 500 gpurun_out/$name.log | tail -3
  if [ $rc -ge 124 ]; then echo "killed: stopping"; exit $rc; fi
}
step tests_rbm 600 python -m pytest tests/test_gpu_rbm.py tests/test_gpu_fullsize.py -q --timeout 600
step rbm_sorted 400 python bench.py --workload rbm --no-cpu-baseline
grep -o '"sparse_minibatch_4096": {[^}]*}' gpurun_out/rbm_sorted.log
