#!/bin/bash
# evidence run: bf16 demo deltas, rocprofv3 kernel stats of the headline bench, PMC traffic passes (separate runs)
export TMPDIR=/tmp
mkdir -p gpurun_out
step() { name=$1; secs=$2; shift 2; echo "== $name"; timeout -k 10 $secs "$@" > gpurun_out/$name.log 2> gpurun_out/$name.err; rc=$?; echo "   rc=$rc"; if [ $rc -ge 124 ]; then echo killed; exit $rc; fi; }
step demo_deltas 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "demo_epochs" 
grep "deltas" gpurun_out/demo_deltas.log
rm -rf gpurun_out/prof_fnn gpurun_out/prof_ipnn
step prof_fnn 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_fnn -o fnn -- python3 bench.py --steps 400 --warmup 20 --no-cpu-baseline --no-extras
find gpurun_out/prof_fnn -name "*kernel_stats.csv" -exec cp {} gpurun_out/fnn_kernel_stats.csv \;
head -5 gpurun_out/fnn_kernel_stats.csv | cut -c1-160
step prof_ipnn 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ipnn -o ip -- python3 bench.py --workload ipnn --steps 100 --warmup 10 --no-cpu-baseline
find gpurun_out/prof_ipnn -name "*kernel_stats.csv" -exec cp {} gpurun_out/ipnn_kernel_stats.csv \;
head -12 gpurun_out/ipnn_kernel_stats.csv | cut -c1-160
step pmc_fnn 400 bash tools/pmc_traffic.sh fnn
step pmc_ipnn 400 bash tools/pmc_traffic.sh ipnn
step pmc_snn 400 bash tools/pmc_traffic.sh snn
ls gpurun_out/pmc_traffic_*.json
