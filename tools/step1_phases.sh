#!/bin/bash
# per-phase s_memtime stamps of the strip kernel (k_step1) on the bench's ids -> gpurun_out/step1_phases.json
# (copy to profiles/rNN_step1_phases.json: bench.py quotes the in-step gather figure from there).  Diagnostic build.
set -e
mkdir -p gpurun_out
python3 - <<'PY'
import sys; sys.path.insert(0, '.')
import deep_ctr_amd  # noqa
from deep_ctr_amd import synth
ids = synth.zipf_ids(32 * 4096, synth.field_sizes_ipinyou(), 1.1, 1234)[:4096]
ids.astype('int32').tofile('gpurun_out/zipf_ids.bin')
PY
hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/mlp_stamps tools/exp/mlp_stamps.hip
/tmp/mlp_stamps | tee gpurun_out/step1_phases.txt
cat gpurun_out/step1_phases.json
