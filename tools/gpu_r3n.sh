#!/bin/bash
# round 3, session n: the next item's epilogue inputs one item ahead in the 16-example-strip launches: tests, A/B against the previous build
export TMPDIR=/tmp
mkdir -p gpurun_out
step() {   # name, seconds, command...
  name=$1; secs=$2; shift 2
  echo "== $name"; timeout -k 10 $secs "$@" > gpurun_out/$name.log 2> gpurun_out/$name.err; rc=$?
  echo "   rc=$rc"; tail -c 200 gpurun_out/$name.log | tail -1
  if [ $rc -ge 124 ]; then echo "killed: stopping"; exit $rc; fi
}
step tests_n 600 python -m pytest tests/test_gpu_ipnn.py tests/test_gpu_fullsize.py -q --timeout 600 -k "ipnn or strip"
B="--workload ipnn --steps 100 --warmup 10 --no-cpu-baseline"
step ip_new 300 python bench.py $B
step ip_prev 300 env FNN_HIP_LIB=$PWD/tools/exp/libfnn_prev.so python bench.py $B
step ip_new2 300 python bench.py $B
step ip_prev2 300 env FNN_HIP_LIB=$PWD/tools/exp/libfnn_prev.so python bench.py $B
for f in ip_new ip_prev ip_new2 ip_prev2; do grep -o '"ms_per_step": [0-9.]*' gpurun_out/$f.log | head -1; grep -o '"fwd": [0-9.]*, "bwd": [0-9.]*' gpurun_out/$f.log | head -1; done
step ip_stamps_tail 300 env IPNN_STAMPS=2 python bench.py --workload ipnn --steps 20 --warmup 5 --no-cpu-baseline
grep "ipnn stamps" gpurun_out/ip_stamps_tail.err | cut -c1-330
