#!/bin/bash
# Run ON the GPU box: alternating A/B of environment settings on the batch-32 legs (cfg2 = scale 2.0, cfg3 = scale 1.0; 50 steps, bf16).
#   bash tools/ab_env_b32.sh "name1:VAR=val VAR2=val" "name2:" ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in $(seq 1 ${AB_REPS:-2}); do
  for spec in "$@"; do
    v=${spec%%:*}; e=${spec#*:}
    b=$(env $e python3 $R/tools/sample_one.py 32 2.0 50 bf16 2>/dev/null | sed 's/.*= //')
    c=$(env $e python3 $R/tools/sample_one.py 32 1.0 50 bf16 2>/dev/null | sed 's/.*= //')
    echo "$v  cfg2 $b   cfg3 $c"
  done
done
