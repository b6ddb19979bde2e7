#!/bin/bash
# round 3, session b: the whole GPU suite (hidden-visibility build, new DP / pin / duo tests), the DP forms again, gather counters
export TMPDIR=/tmp
mkdir -p gpurun_out
step() {   # name, seconds, command...
  name=$1; secs=$2; shift 2
  echo "== $name"; timeout -k 10 $secs "$@" > gpurun_out/$name.log 2> gpurun_out/$name.err; rc=$?
  echo "   rc=$rc"; tail -c 600 gpurun_out/$name.log | tail -4
  if [ $rc -ge 124 ]; then echo "killed: stopping"; exit $rc; fi
}
step gpu_tests 1100 python -m pytest tests -m gpu -q --timeout 600
B="--steps 200 --warmup 20 --no-extras --no-cpu-baseline"
step w1_slabs 300 env FNN_BENCH_FORCE_DP=1 python bench.py $B
step w1_bucket 300 env FNN_BENCH_FORCE_DP=1 python bench.py $B --dp-payload bucket
step w1_p2p 300 env FNN_BENCH_FORCE_DP=1 python bench.py $B --dp-collective p2p
step rh2_p2p 300 env FNN_BENCH_REHEARSE=1 python bench.py --gpus 2 $B --dp-collective p2p
step pmc_gather 900 bash tools/pmc_gather.sh
