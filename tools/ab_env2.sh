#!/bin/bash
# Run ON the GPU box: A/B of environment knobs on the batch-32 workloads (tools/sample_one.py), alternating.
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in $(seq 1 ${AB_REPS:-3}); do
  for spec in "$@"; do
    if [ "$spec" = "-" ]; then e=""; else e="$spec"; fi
    b=$(env $e python3 $R/tools/sample_one.py 32 1.0 20 bf16 2>/dev/null | sed 's/.*= //')
    c=$(env $e python3 $R/tools/sample_one.py 32 2.0 20 bf16 2>/dev/null | sed 's/.*= //')
    d=$(env $e python3 $R/tools/sample_one.py 16 1.0 20 bf16 2>/dev/null | sed 's/.*= //')
    echo "[$spec]  b32 $b   b32+cfg $c   b16 $d"
  done
done
