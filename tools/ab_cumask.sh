#!/bin/bash
# Run ON the GPU box: A/B of CU-masked clip-parallel branch streams (tuning build, SF_BRANCH_CUMASK=d: branch b gets the mask bits k with
# (k / d) % branches == b), alternating, bf16 engine: configs[1] (two branches of 4 clips), configs[2] (two branches of 32 evaluations),
# and four branches at configs[1].
R=${GRAFT_REPO_ROOT:-$(pwd)}
export SF_LIB_PATH=$R/syncfusion_amd/lib/libsyncfusion_amd_tuning.so
for rep in $(seq 1 ${AB_REPS:-3}); do
  for spec in "-" "SF_BRANCH_CUMASK=128" "SF_BRANCH_CUMASK=64" "SF_UNET_BRANCHES=4" "SF_UNET_BRANCHES=4 SF_BRANCH_CUMASK=64"; do
    if [ "$spec" = "-" ]; then e=""; else e="$spec"; fi
    a=$(env $e python3 $R/tools/sample_one.py 8 1.0 50 bf16 2>/dev/null | sed 's/.*= //')
    b=$(env $e python3 $R/tools/sample_one.py 32 2.0 20 bf16 2>/dev/null | sed 's/.*= //')
    echo "[$spec]  configs[1] ${a:-FAILED}   configs[2] ${b:-FAILED}"
  done
done
