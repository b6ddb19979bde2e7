#!/bin/bash
# round 3, session i: fixes of session h (RBM tail error sum, DAE test, mask prefetch at the end of the side chain)
export TMPDIR=/tmp
mkdir -p gpurun_out
step() {   # name, seconds, command...
  name=$1; secs=$2; shift 2
  echo "== $name"; timeout -k 10 $secs "$@" > gpurun_out/$name.log 2> gpurun_out/$name.err; rc=$?
  echo "   rc=$rc"; tail -c 300 gpurun_out/$name.log | tail -2
  if [ $rc -ge 124 ]; then echo "killed: stopping"; exit $rc; fi
}
step tests_i 900 python -m pytest tests/test_gpu_dae.py tests/test_gpu_ipnn.py tests/test_gpu_rbm.py -q --timeout 600
B="--workload ipnn --steps 100 --warmup 10 --no-cpu-baseline"
step ip_pf 300 python bench.py $B
step ip_nopf 300 env IPNN_BENCH_NOPREFETCH=1 python bench.py $B
step ip_pf2 300 python bench.py $B
step ip_nopf2 300 env IPNN_BENCH_NOPREFETCH=1 python bench.py $B
for f in ip_pf ip_nopf ip_pf2 ip_nopf2; do grep -o '"ms_per_step": [0-9.]*' gpurun_out/$f.log | head -1; done
step rbm_sorted 400 python bench.py --workload rbm --no-cpu-baseline
grep -o '"sparse_minibatch_4096": {[^}]*}' gpurun_out/rbm_sorted.log | cut -c1-220
step rbm_wgs2048 400 env RBM_BATCH_WGS=2048 python bench.py --workload rbm --no-cpu-baseline
grep -o '"sparse_minibatch_4096": {[^}]*}' gpurun_out/rbm_wgs2048.log | cut -c1-120
step rbm_wgs512 400 env RBM_BATCH_WGS=512 python bench.py --workload rbm --no-cpu-baseline
grep -o '"sparse_minibatch_4096": {[^}]*}' gpurun_out/rbm_wgs512.log | cut -c1-120
