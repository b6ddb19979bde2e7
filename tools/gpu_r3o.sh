#!/bin/bash
# round 3, closing session: the whole GPU suite and the default bench line on the final code
export TMPDIR=/tmp
mkdir -p gpurun_out
step() {   # name, seconds, command...
  name=$1; secs=$2; shift 2
  echo "== $name"; timeout -k 10 $secs "$@" > gpurun_out/$name.log 2> gpurun_out/$name.err; rc=$?
  echo "   rc=$rc"; tail -c 300 gpurun_out/$name.log | tail -2
  if [ $rc -ge 124 ]; then echo "killed: stopping"; exit $rc; fi
}
step gpu_tests 1100 python -m pytest tests -m gpu -q --timeout 600
step smoke 200 python -c "import __graft_entry__ as g; g.smoke()"
step bench_default 900 python bench.py
step bench_ipnn 300 python bench.py --workload ipnn --steps 100 --warmup 10 --no-cpu-baseline
