#!/bin/bash
# HBM traffic per launch of the step kernels: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 passes
# (MI355X_MICROARCH.md, rocprofv3 PMC slots), same bench command; summary -> gpurun_out/pmc_traffic.json
export TMPDIR=/tmp
W=${1:-fnn}
rm -rf gpurun_out/pmc_f gpurun_out/pmc_w
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_f -o f -- python3 bench.py --workload $W --steps 40 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_w -o w -- python3 bench.py --workload $W --steps 40 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/pmc_w.log 2>&1
python3 tools/pmc_summarise.py gpurun_out/pmc_f gpurun_out/pmc_w > gpurun_out/pmc_traffic_$W.json
cat gpurun_out/pmc_traffic_$W.json
