#!/bin/bash
# MFMA busy cycles of the FNN step's launches in the three precisions (PMC pass of its own: --pmc with --kernel-trace only)
export TMPDIR=/tmp
mkdir -p gpurun_out
for prec in bf16 bf16x3 f32; do
  rm -rf gpurun_out/pmc_m_$prec
  timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv -d gpurun_out/pmc_m_$prec -o m -- python3 bench.py --precision $prec --steps 40 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/pmc_m_$prec.log 2>&1 || { echo "pmc pass $prec failed"; tail -3 gpurun_out/pmc_m_$prec.log; exit 1; }
done
python3 - <<'PY' > gpurun_out/pmc_mfma_prec.json
import csv, glob, json, collections
res = {}
for prec in ('bf16', 'bf16x3', 'f32'):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob('gpurun_out/pmc_m_%s/**/*counter_collection.csv' % prec, recursive=True):
        for r in csv.DictReader(open(path)):
            if 'k_step' in r['Kernel_Name']:
                acc[r['Kernel_Name'].split('<')[0].split('::')[-1][:12]][r['Counter_Name']].append(float(r['Counter_Value']))
    out = {}
    for name, c in sorted(acc.items()):
        n = len(c['GRBM_GUI_ACTIVE'])
        busy, act = sum(c['SQ_VALU_MFMA_BUSY_CYCLES']) / n, sum(c['GRBM_GUI_ACTIVE']) / n
        out[name] = {'launches': n, 'mfma_busy_cycles': busy, 'gui_active_cycles': act, 'mfma_util_pct': 100.0 * busy / (act * 1024.0),
                     'mfma_mops_bf16': sum(c.get('SQ_INSTS_VALU_MFMA_MOPS_BF16', [0])) / n, 'mfma_mops_f32': sum(c.get('SQ_INSTS_VALU_MFMA_MOPS_F32', [0])) / n}
    res[prec] = out
print(json.dumps(res, indent=1))
PY
cat gpurun_out/pmc_mfma_prec.json
