#!/bin/bash
# Run ON the GPU box: alternating A/B of HIP-library builds (SF_LIB_PATH) on the batch-32 legs + the headline.
#   bash tools/ab_libs_b32.sh name1=path1 name2=path2 ...   (paths relative to the repo root)
# the SF_* hooks exist only in the tuning build of the library (make -C syncfusion_amd/csrc tuning)
export SF_LIB_PATH=${SF_LIB_PATH:-${GRAFT_REPO_ROOT:-$(pwd)}/syncfusion_amd/lib/libsyncfusion_amd_tuning.so}
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in $(seq 1 ${AB_REPS:-2}); do
  for spec in "$@"; do
    v=${spec%%=*}; L=$R/${spec#*=}
    a=$(SF_LIB_PATH=$L python3 $R/tools/sample_one.py 8 1.0 50 bf16 2>/dev/null | sed 's/.*= //')
    b=$(SF_LIB_PATH=$L python3 $R/tools/sample_one.py 32 2.0 50 bf16 2>/dev/null | sed 's/.*= //')
    c=$(SF_LIB_PATH=$L python3 $R/tools/sample_one.py 32 1.0 50 bf16 2>/dev/null | sed 's/.*= //')
    d=$(SF_LIB_PATH=$L python3 $R/tools/sample_one.py 10 2.0 20 bf16 262144 2>/dev/null | sed 's/.*= //')
    echo "$v  cfg1 $a  cfg2 $b  cfg3 $c  ref $d"
  done
done
