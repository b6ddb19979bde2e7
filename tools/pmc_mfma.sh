#!/bin/bash
# MFMA utilisation per kernel from PMC counters: SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 1024 SIMDs),
# the MfmaUtil expression of rocprofv3 (gfx94x formula); its own pass (kernel trace only beside it)
export TMPDIR=/tmp
W=${1:-ipnn}
rm -rf gpurun_out/pmc_m
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 --kernel-trace --output-format csv -d gpurun_out/pmc_m -o m -- python3 bench.py --workload $W --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/pmc_m.log 2>&1
python3 - <<'PY' > gpurun_out/pmc_mfma_$W.json
import csv, glob, json, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob('gpurun_out/pmc_m/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(path)):
        k = (r['Kernel_Name'][:70], r.get('Grid_Size', ''))
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
out = {}
for (name, grid), c in acc.items():
    if 'SQ_VALU_MFMA_BUSY_CYCLES' not in c or not any(c['SQ_VALU_MFMA_BUSY_CYCLES']):
        continue
    busy = sum(c['SQ_VALU_MFMA_BUSY_CYCLES']) / len(c['SQ_VALU_MFMA_BUSY_CYCLES'])
    act = sum(c['GRBM_GUI_ACTIVE']) / len(c['GRBM_GUI_ACTIVE'])
    mops = sum(c.get('SQ_INSTS_VALU_MFMA_MOPS_BF16', [0])) / max(1, len(c.get('SQ_INSTS_VALU_MFMA_MOPS_BF16', [0])))
    out['%s grid %s' % (name, grid)] = {'launches': len(c['GRBM_GUI_ACTIVE']), 'mfma_busy_cycles': busy, 'gui_active_cycles': act,
                                       'mfma_util_pct': 100.0 * busy / (act * 1024.0),
                                       # GRBM_GUI_ACTIVE is reported summed over the 8 XCDs: the fraction of SIMD cycles with an MFMA in flight is 8 x the above
                                       'mfma_util_pct_xcd_normalised': 100.0 * busy / (act / 8.0 * 1024.0), 'bf16_mfma_flops': mops * 512}
print(json.dumps(out, indent=1))
PY
cat gpurun_out/pmc_mfma_$W.json
