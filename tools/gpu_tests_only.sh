#!/bin/bash
# GPU tests only (optionally a -k expression as $1)
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q --timeout 600 ${1:+-k "$1"} > gpurun_out/gpu_tests.log 2>&1; rc=$?
tail -15 gpurun_out/gpu_tests.log; exit $rc
