#!/bin/bash
# Run ON the GPU box: macro-tile variants (SF_MT_VARIANT) alone on the chip for the shapes given as "B L C N taps" strings
# the SF_* hooks exist only in the tuning build of the library (make -C syncfusion_amd/csrc tuning)
export SF_LIB_PATH=${SF_LIB_PATH:-${GRAFT_REPO_ROOT:-$(pwd)}/syncfusion_amd/lib/libsyncfusion_amd_tuning.so}
cd $GRAFT_REPO_ROOT
for shape in "$@"; do
  for v in ${MT_VARIANTS:-1 3 5 7 4}; do
    echo "shape [$shape] v$v: $(SF_MT_VARIANT=$v python tools/gemm_one.py $shape 6 -1 -1 50 2>/dev/null | tail -1)"
  done
done
