#!/bin/bash
# round 3, session m: stamps of the 16-example-strip (tail) launches
export TMPDIR=/tmp
mkdir -p gpurun_out
step() {   # name, seconds, command...
  name=$1; secs=$2; shift 2
  echo "== $name"; timeout -k 10 $secs "$@" > gpurun_out/$name.log 2> gpurun_out/$name.err; rc=$?
  echo "   rc=$rc"; tail -c 200 gpurun_out/$name.log | tail -1
  if [ $rc -ge 124 ]; then echo "killed: stopping"; exit $rc; fi
}
step tests_m 600 python -m pytest tests/test_gpu_ipnn.py -q --timeout 600
step ip_stamps_tail 300 env IPNN_STAMPS=2 python bench.py --workload ipnn --steps 20 --warmup 5 --no-cpu-baseline
grep "ipnn stamps" gpurun_out/ip_stamps_tail.err | cut -c1-330
step ip_stamps_tail_sel 300 env IPNN_STAMPS=2 IPNN_STAMP_SEL=1 python bench.py --workload ipnn --steps 20 --warmup 5 --no-cpu-baseline
grep "ipnn stamps fwd" gpurun_out/ip_stamps_tail_sel.err | cut -c1-400
