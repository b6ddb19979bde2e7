#!/bin/bash
# HBM traffic per launch of the standalone gather kernels (k_gather_ref: A3, k_bag_ref: A8), one VARIANT per process (the variants
# share kernel names): FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes -> gpurun_out/pmc_traffic_gather.json
export TMPDIR=/tmp
echo "{" > gpurun_out/pmc_traffic_gather.json
first=1
for v in ${@:-fm fm_uniform bag bag_zipf}; do
  rm -rf gpurun_out/pmc_gf gpurun_out/pmc_gw
  export FNN_GATHER_ONLY=$v
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_gf -o f -- python3 bench.py --workload gather --steps 30 --warmup 5 > gpurun_out/pmc_g_$v.f.log 2>&1 || exit 1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_gw -o w -- python3 bench.py --workload gather --steps 30 --warmup 5 > gpurun_out/pmc_g_$v.w.log 2>&1 || exit 1
  [ $first = 1 ] || echo "," >> gpurun_out/pmc_traffic_gather.json
  first=0
  python3 - $v >> gpurun_out/pmc_traffic_gather.json <<'PY'
import json, subprocess, sys
d = json.loads(subprocess.run([sys.executable, 'tools/pmc_summarise.py', 'gpurun_out/pmc_gf', 'gpurun_out/pmc_gw'], capture_output=True, text=True, check=True).stdout)
k = [v for n, v in d.items() if 'k_gather_ref' in n or 'k_bag_ref' in n]
assert len(k) == 1, d.keys()
print('"%s": %s' % (sys.argv[1], json.dumps(k[0])))
PY
  echo "== $v done"
done
echo "}" >> gpurun_out/pmc_traffic_gather.json
cat gpurun_out/pmc_traffic_gather.json
