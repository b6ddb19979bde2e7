#!/bin/bash
# per-dispatch device time of the inner-product family's kernels, grouped by (kernel, grid) = layer
export TMPDIR=/tmp
rm -rf gpurun_out/prof_ipl
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_ipl -o ip -- python3 bench.py --workload ipnn --steps 20 --warmup 3 > gpurun_out/prof_ipl.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/prof_ipl/**/*kernel_trace.csv', recursive=True)[0]
acc = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    if 'fnn' not in n and 'k_ip' not in n and 'GLOBAL' not in n and 'k_mask' not in n: continue
    key = (n[:70], r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size'), r.get('Grid_Size_Y'), r.get('Grid_Size_Z'))
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    acc.setdefault(key, []).append(d)
for k, v in acc.items():
    print('%-72s grid %s %s %s  n=%d  avg %.1f us' % (k[0], k[1], k[2], k[3], len(v), sum(v) / len(v) / 1e3))
PY
