#!/bin/bash
# round 3, session l: small products' weights in LDS for the 16-example-strip launches (A/B), radix sort layout, RBM
export TMPDIR=/tmp
mkdir -p gpurun_out
step() {   # name, seconds, command...
  name=$1; secs=$2; shift 2
  echo "== $name"; timeout -k 10 $secs "$@" > gpurun_out/$name.log 2> gpurun_out/$name.err; rc=$?
  echo "   rc=$rc"; tail -c 300 gpurun_out/$name.log | tail -2
  if [ $rc -ge 124 ]; then echo "killed: stopping"; exit $rc; fi
}
step tests_l 900 python -m pytest tests/test_gpu_ipnn.py tests/test_gpu_fullsize.py tests/test_gpu_rbm.py tests/test_gpu_dp.py -q --timeout 600 -k "ipnn or rbm or sparse or scatter_global or fullsize or strip"
B="--workload ipnn --steps 100 --warmup 10 --no-cpu-baseline"
step ip_lds 300 python bench.py $B
step ip_nolds 300 env IPNN_TAIL_LDS=0 python bench.py $B
step ip_lds2 300 python bench.py $B
step ip_nolds2 300 env IPNN_TAIL_LDS=0 python bench.py $B
for f in ip_lds ip_nolds ip_lds2 ip_nolds2; do grep -o '"ms_per_step": [0-9.]*' gpurun_out/$f.log | head -1; grep -o '"fwd": [0-9.]*, "bwd": [0-9.]*' gpurun_out/$f.log | head -1; done
step rbm 300 python bench.py --workload rbm --no-cpu-baseline
grep -o '"sparse_minibatch_4096": {[^}]*}' gpurun_out/rbm.log | cut -c1-130
