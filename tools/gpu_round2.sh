#!/bin/bash
# one GPU session of round 2: tests, the default bench line, the data-parallel path at world size 1 (RCCL) and as a
# two-rank gloo rehearsal on one GPU, the strip kernel's phase stamps.  A step that is KILLED ends the session.
export TMPDIR=/tmp
mkdir -p gpurun_out
step() {   # name, seconds, command...
  name=$1; secs=$2; shift 2
  echo "== $name"; timeout -k 10 $secs "$@" > gpurun_out/$name.log 2> gpurun_out/$name.err; rc=$?
  echo "   rc=$rc"; tail -c 400 gpurun_out/$name.log | tail -3
  if [ $rc -ge 124 ]; then echo "killed: stopping"; exit $rc; fi
}
step gpu_tests 1000 python -m pytest tests -m gpu -q --timeout 600
step bench_default 400 bash -c "time python bench.py"
step bench_dp_world1 200 env FNN_BENCH_FORCE_DP=1 python bench.py --no-extras --no-cpu-baseline
step bench_rehearse2 300 env FNN_BENCH_REHEARSE=1 python bench.py --gpus 2 --no-extras --no-cpu-baseline --steps 100
step bench_rehearse2x 300 env FNN_BENCH_REHEARSE=1 python bench.py --gpus 2 --no-extras --no-cpu-baseline --steps 50 --dp-sparse exchange
step step1_phases 200 bash tools/step1_phases.sh
step smoke 200 python -c "import __graft_entry__ as g; g.smoke()"
