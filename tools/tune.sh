#!/bin/bash
# step time against the two launch-shape knobs (results stay correct: both only re-partition work)
for sk in ${SKS:-2 4 8}; do for s2 in ${S2S:-256}; do
  FNN_SPLITK=$sk FNN_SCAT2_WGS=$s2 python bench.py --steps 300 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); k = d['kernel_ms']
print('splitk=$sk scat2_wgs=$s2', 'ms/step %.4f' % d['ms_per_step'], 'step1 %.4f step2 %.4f step3 %.4f' % (k['step1'], k['step2'], k['step3']))
"
done; done
