#!/bin/bash
# Run ON the GPU box: phase offset between the two independent branch pipelines (SF_BRANCH_STAGGER_US) at batch 32 without guidance.
# the SF_* hooks exist only in the tuning build of the library (make -C syncfusion_amd/csrc tuning)
export SF_LIB_PATH=${SF_LIB_PATH:-${GRAFT_REPO_ROOT:-$(pwd)}/syncfusion_amd/lib/libsyncfusion_amd_tuning.so}
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do
for st in 0 400 800 1200 1600 2000 2400; do
  echo -n "stagger $st us: "
  SF_BRANCH_STAGGER_US=$st python3 $R/tools/sample_one.py 32 1.0 50 bf16 2>/dev/null
done
done
