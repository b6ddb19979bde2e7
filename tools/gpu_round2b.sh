#!/bin/bash
# second GPU session of round 2: all GPU tests, the driver's own launch form at N = 1 (torchrun, WORLD_SIZE = 1), a strong-scaling
# rehearsal of two ranks on one GPU (gloo), the e2e workload with the id cache, smoke.  A step that is KILLED ends the session.
export TMPDIR=/tmp
mkdir -p gpurun_out
step() {   # name, seconds, command...
  name=$1; secs=$2; shift 2
  echo "== $name"; timeout -k 10 $secs "$@" > gpurun_out/$name.log 2> gpurun_out/$name.err; rc=$?
  echo "   rc=$rc"; tail -c 400 gpurun_out/$name.log | tail -3
  if [ $rc -ge 124 ]; then echo "killed: stopping"; exit $rc; fi
}
step gpu_tests 1000 python -m pytest tests -m gpu -q --timeout 600
step torchrun_n1 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline
step rehearse_strong2 300 env FNN_BENCH_REHEARSE=1 python bench.py --gpus 2 --scaling strong --no-extras --no-cpu-baseline --steps 100
step bench_e2e 400 python bench.py --workload e2e
step smoke 200 python -c "import __graft_entry__ as g; g.smoke()"
