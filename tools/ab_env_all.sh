#!/bin/bash
# Run ON the GPU box: alternating A/B of environment settings on four workloads (bf16): cfg1 = batch 8 (headline), cfg2 = batch 32 with
# guidance, cfg3 = batch 32 without, ref = batch 10 x 2^18 samples with guidance.   bash tools/ab_env_all.sh "name1:VAR=val ..." "name2:" ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in $(seq 1 ${AB_REPS:-2}); do
  for spec in "$@"; do
    v=${spec%%:*}; e=${spec#*:}
    a=$(env $e python3 $R/tools/sample_one.py 8 1.0 50 bf16 2>/dev/null | sed 's/.*= //')
    b=$(env $e python3 $R/tools/sample_one.py 32 2.0 50 bf16 2>/dev/null | sed 's/.*= //')
    c=$(env $e python3 $R/tools/sample_one.py 32 1.0 50 bf16 2>/dev/null | sed 's/.*= //')
    d=$(env $e python3 $R/tools/sample_one.py 10 2.0 20 bf16 262144 2>/dev/null | sed 's/.*= //')
    echo "$v  cfg1 $a  cfg2 $b  cfg3 $c  ref $d"
  done
done
