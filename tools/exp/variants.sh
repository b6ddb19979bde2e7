#!/bin/bash
# build-and-run the stamped strip kernel (diagnostics only); optional sed script patches a copy of the kernel header
set -e
SRC=deep-ctr_amd/csrc
run_variant() {   # name, sed script
  d=/tmp/fnnv_$1; rm -rf $d; mkdir -p $d/deep-ctr_amd/csrc $d/tools/exp
  cp $SRC/fnn_kernels.hip.h $SRC/fnn_step_kernels.hip.h $d/deep-ctr_amd/csrc/
  cp tools/exp/mlp_stamps.hip $d/tools/exp/
  if [ -n "$2" ]; then sed -i -E "$2" $d/deep-ctr_amd/csrc/fnn_kernels.hip.h; fi
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -o $d/ms $d/tools/exp/mlp_stamps.hip 2>&1 | grep -E " error" || true
  echo "== $1"; $d/ms | head -13
}
run_variant current ""
run_variant nostores 's/store4\(a\.(xpT|d1T|d2T|dl1T|dl2T)/if (a.B < 0) store4(a.\1/g'
