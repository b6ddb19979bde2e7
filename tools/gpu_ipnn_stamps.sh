#!/bin/bash
# per-product stamps of the wide strip kernels (IPNN_STAMPS=1) with the detail of one product's last block of wave 0 (IPNN_STAMP_SEL),
# then the tails (IPNN_STAMPS=2)
export TMPDIR=/tmp
mkdir -p gpurun_out
: > gpurun_out/ipnn_stamps.txt
for sel in ${STAMP_SELS:-0 1 2 3}; do
  IPNN_STAMPS=1 IPNN_STAMP_SEL=$sel timeout -k 10 200 python bench.py --workload ipnn --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ipnn_st.json 2> gpurun_out/ipnn_st.err || exit 1
  echo "== sel=$sel" >> gpurun_out/ipnn_stamps.txt; grep "ipnn stamps" gpurun_out/ipnn_st.err | tail -4 >> gpurun_out/ipnn_stamps.txt
done
IPNN_STAMPS=2 timeout -k 10 200 python bench.py --workload ipnn --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ipnn_st.json 2> gpurun_out/ipnn_st.err || exit 1
echo "== tails" >> gpurun_out/ipnn_stamps.txt; grep "ipnn stamps" gpurun_out/ipnn_st.err | tail -2 >> gpurun_out/ipnn_stamps.txt
cat gpurun_out/ipnn_stamps.txt
