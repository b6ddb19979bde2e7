#!/bin/bash
# round 3, session g: the split f64 DAE trainer, the two-level RBM tail; pretrain + rbm benches
export TMPDIR=/tmp
mkdir -p gpurun_out
step() {   # name, seconds, command...
  name=$1; secs=$2; shift 2
  echo "== $name"; timeout -k 10 $secs "$@" > gpurun_out/$name.log 2> gpurun_out/$name.err; rc=$?
  echo "   rc=$rc"; tail -c 600 gpurun_out/$name.log | tail -4
  if [ $rc -ge 124 ]; then echo "killed: stopping"; exit $rc; fi
}
step tests_g 900 python -m pytest tests/test_gpu_dae.py tests/test_gpu_rbm.py -q --timeout 600 -x
step rbm_sorted 400 python bench.py --workload rbm --no-cpu-baseline
grep -o '"sparse_minibatch_4096": {[^}]*}' gpurun_out/rbm_sorted.log
step pretrain 400 python bench.py --workload pretrain --no-cpu-baseline
grep -o '"dae_online": {[^}]*}' gpurun_out/pretrain.log
step pretrain_nosplit 400 env DAE_SPLIT=0 python bench.py --workload pretrain --no-cpu-baseline
grep -o '"dae_online": {[^}]*}' gpurun_out/pretrain_nosplit.log
