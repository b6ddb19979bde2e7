#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
for sel in 0 1 4 6; do
  IPNN_STRIP_DUO=0 IPNN_STAMPS=1 IPNN_STAMP_SEL=$sel timeout -k 10 200 python bench.py --workload ipnn --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ipnn_st.json 2> gpurun_out/ipnn_st.err
  echo "== sel=$sel"; grep "ipnn stamps fwd" gpurun_out/ipnn_st.err | tail -1 | sed 's/.*| product/product/'
done
