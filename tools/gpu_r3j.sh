#!/bin/bash
# round 3, session j: the clock-bounded p2p wait: DP tests, two-process rehearsals (LOCAL and EXCHANGE)
export TMPDIR=/tmp
mkdir -p gpurun_out
step() {   # name, seconds, command...
  name=$1; secs=$2; shift 2
  echo "== $name"; timeout -k 10 $secs "$@" > gpurun_out/$name.log 2> gpurun_out/$name.err; rc=$?
  echo "   rc=$rc"; tail -c 300 gpurun_out/$name.log | tail -2
  if [ $rc -ge 124 ]; then echo "killed: stopping"; exit $rc; fi
}
step tests_dp 900 python -m pytest tests/test_gpu_dp.py -q --timeout 600
B="--steps 200 --warmup 20 --no-extras --no-cpu-baseline"
step rh2_p2p 300 env FNN_BENCH_REHEARSE=1 python bench.py --gpus 2 $B --dp-collective p2p
step rh2_p2p_exchange 400 env FNN_BENCH_REHEARSE=1 python bench.py --gpus 2 $B --dp-collective p2p --dp-sparse exchange
grep -o '"p2p_max_flag_wait_us": [0-9.e+]*' gpurun_out/rh2_p2p.log gpurun_out/rh2_p2p_exchange.log
grep -o '"exact_mode_check": {[^}]*}' gpurun_out/rh2_p2p_exchange.log
