#!/bin/bash
# Run ON the GPU box: A/B of HIP-library builds (SF_LIB_PATH), alternating, two rounds, on three workloads:
# configs[1] (bench.py), configs[2] (batch 32, guidance 2.0) and one GPU's share of configs[3] (batch 32, no guidance).
#   bash tools/ab_libs.sh name1=path1 name2=path2 ...   (paths relative to the repo root)
# the SF_* hooks exist only in the tuning build of the library (make -C syncfusion_amd/csrc tuning)
export SF_LIB_PATH=${SF_LIB_PATH:-${GRAFT_REPO_ROOT:-$(pwd)}/syncfusion_amd/lib/libsyncfusion_amd_tuning.so}
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in $(seq 1 ${AB_REPS:-2}); do
  for spec in "$@"; do
    v=${spec%%=*}; L=$R/${spec#*=}
    a=$(SF_LIB_PATH=$L python3 $R/bench.py --steps 50 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])")
    if [ "${AB_SECONDARY:-1}" = "1" ]; then
      b=$(SF_LIB_PATH=$L python3 $R/tools/sample_one.py 32 2.0 20 bf16 2>/dev/null | sed 's/.*= //')
      c=$(SF_LIB_PATH=$L python3 $R/tools/sample_one.py 32 1.0 20 bf16 2>/dev/null | sed 's/.*= //')
    fi
    echo "$v  configs[1] $a steps/s   configs[2] ${b:-}   b32 ${c:-}"
  done
done
