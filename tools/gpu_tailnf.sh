#!/bin/bash
# parity of the fragment-granular tail items (IPNN_TAIL_NF / IPNN_TAIL_NW), then an alternating A/B of the step time on one box
# TAIL_VARIANTS: "nf:nw ..." (default "4:8 2:8 1:8 1:16")
export TMPDIR=/tmp
mkdir -p gpurun_out
V=${TAIL_VARIANTS:-4:8 2:8 1:8 1:16}
for v in $V; do
  nf=${v%%:*}; nw=${v##*:}
  IPNN_TAIL_NF=$nf IPNN_TAIL_NW=$nw timeout -k 10 400 python -m pytest tests/test_gpu_ipnn.py tests/test_gpu_fullsize.py -m gpu -q -x -k "ip" > gpurun_out/tailnf_tests_${nf}_$nw.log 2>&1 || { echo "tests NF=$nf NW=$nw failed"; tail -30 gpurun_out/tailnf_tests_${nf}_$nw.log; exit 1; }
  echo "NF=$nf NW=$nw: $(tail -1 gpurun_out/tailnf_tests_${nf}_$nw.log)"
done
for rep in 1 2 3; do
 for v in $V; do
  nf=${v%%:*}; nw=${v##*:}
  IPNN_TAIL_NF=$nf IPNN_TAIL_NW=$nw timeout -k 10 300 python bench.py --workload ipnn --no-cpu-baseline --steps 300 > gpurun_out/ab_nf${nf}_$nw.json 2> gpurun_out/ab_nf${nf}_$nw.err || { echo "nf $nf failed"; tail -3 gpurun_out/ab_nf${nf}_$nw.err; exit 1; }
  python - <<PY
import json
d = json.loads(open('gpurun_out/ab_nf${nf}_$nw.json').read().strip().splitlines()[-1])
print('NF=$nf NW=$nw rep $rep', 'ms/step %.4f' % d['ms_per_step'], {k: round(v * 1e3, 1) for k, v in d['kernel_ms'].items() if v and k != 'sort_now'})
PY
 done
done
