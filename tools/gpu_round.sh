#!/bin/bash
# one GPU session: tests, the three bench workloads, rocprof kernel stats of the default bench
export TMPDIR=/tmp
python -m pytest tests -m gpu -q > gpurun_out/gpu_tests.log 2>&1; tail -3 gpurun_out/gpu_tests.log
python bench.py > gpurun_out/bench_fnn.json 2> gpurun_out/bench_fnn.err; tail -c 1500 gpurun_out/bench_fnn.json
python bench.py --workload snn > gpurun_out/bench_snn.json 2>/dev/null; tail -c 600 gpurun_out/bench_snn.json
python bench.py --workload ipnn --steps 50 --warmup 5 > gpurun_out/bench_ipnn.json 2>/dev/null; tail -c 900 gpurun_out/bench_ipnn.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_fnn -o fnn -- python3 bench.py --steps 300 --warmup 20 --no-cpu-baseline > gpurun_out/prof_fnn.log 2>&1
ls gpurun_out/prof_fnn | head
