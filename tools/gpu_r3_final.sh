#!/bin/bash
# round 3, evidence session: whole GPU suite, smoke, default bench line, rocprofv3 kernel stats of the headline command, PMC passes
# (fnn, ipnn), the data-parallel forms at world 1 and as two-process rehearsals.  A step that is KILLED ends the session.
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
rm -rf gpurun_out/prof_fnn
step prof_fnn 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_fnn -o fnn -- python3 bench.py --steps 400 --warmup 20 --no-cpu-baseline --no-extras
find gpurun_out/prof_fnn -name "*kernel_stats.csv" -exec cp {} gpurun_out/fnn_kernel_stats.csv \;
head -5 gpurun_out/fnn_kernel_stats.csv | cut -c1-140
B="--steps 200 --warmup 20 --no-extras --no-cpu-baseline"
step w1_slabs 300 env FNN_BENCH_FORCE_DP=1 python bench.py $B
step w1_bucket 300 env FNN_BENCH_FORCE_DP=1 python bench.py $B --dp-payload bucket
step w1_p2p 300 env FNN_BENCH_FORCE_DP=1 python bench.py $B --dp-collective p2p
step rh2_bucket 300 env FNN_BENCH_REHEARSE=1 python bench.py --gpus 2 $B --dp-payload bucket
step rh2_p2p 300 env FNN_BENCH_REHEARSE=1 python bench.py --gpus 2 $B --dp-collective p2p
step rh2_p2p_exchange 300 env FNN_BENCH_REHEARSE=1 python bench.py --gpus 2 $B --dp-collective p2p --dp-sparse exchange
step pmc_fnn 400 bash tools/pmc_traffic.sh fnn
step pmc_ipnn 400 bash tools/pmc_traffic.sh ipnn
step bench_pretrain 300 python bench.py --workload pretrain --no-cpu-baseline
step bench_rbm 300 python bench.py --workload rbm --no-cpu-baseline
ls gpurun_out/pmc_traffic_*.json
