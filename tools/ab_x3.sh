#!/bin/bash
# Run ON the GPU box: A/B of tuning-hook settings on the fp32x engine (tuning build), alternating: configs[2] (B=32, guidance) and configs[1].
#   bash tools/ab_x3.sh "NAME=VALUE ..." "-" ...    ("-" = no extra environment)
R=${GRAFT_REPO_ROOT:-$(pwd)}
export SF_LIB_PATH=$R/syncfusion_amd/lib/libsyncfusion_amd_tuning.so
for rep in $(seq 1 ${AB_REPS:-2}); do
  for spec in "$@"; do
    if [ "$spec" = "-" ]; then e=""; else e="$spec"; fi
    a=$(env $e python3 $R/tools/sample_one.py 32 2.0 ${AB_STEPS:-12} fp32x 2>/dev/null | sed 's/.*= //')
    b=""
    if [ "${AB_CFG1:-1}" = "1" ]; then b=$(env $e python3 $R/tools/sample_one.py 8 1.0 30 fp32x 2>/dev/null | sed 's/.*= //'); fi
    echo "[$spec]  configs[2] ${a:-FAILED}   configs[1] ${b:-}"
  done
done
