#!/bin/bash
# round 3, session k: k_rbm_batch32, RBM edge shapes, a partial global batch through the own radix sort; rbm kernel trace
export TMPDIR=/tmp
mkdir -p gpurun_out
step() {   # name, seconds, command...
  name=$1; secs=$2; shift 2
  echo "== $name"; timeout -k 10 $secs "$@" > gpurun_out/$name.log 2> gpurun_out/$name.err; rc=$?
  echo "   rc=$rc"; tail -c 300 gpurun_out/$name.log | tail -2
  if [ $rc -ge 124 ]; then echo "killed: stopping"; exit $rc; fi
}
step tests_k 900 python -m pytest tests/test_gpu_rbm.py tests/test_gpu_dp.py -q --timeout 600 -k "rbm or sparse or scatter_global"
step rbm32 300 python bench.py --workload rbm --no-cpu-baseline
step rbm_generic 300 env RBM_BATCH_GENERIC=1 python bench.py --workload rbm --no-cpu-baseline
for f in rbm32 rbm_generic; do grep -o '"sparse_minibatch_4096": {[^}]*}' gpurun_out/$f.log | cut -c1-130; done
rm -rf gpurun_out/prof_rbm
step prof_rbm 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_rbm -o rbm -- python3 bench.py --workload rbm --no-cpu-baseline
find gpurun_out/prof_rbm -name "*kernel_stats.csv" -exec cp {} gpurun_out/rbm_kernel_stats.csv \;
