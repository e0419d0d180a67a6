#!/bin/bash
# Run ON the GPU box: A/B of environment knobs on ONE sample_one.py workload ($AB_ARGS, default "32 1.0 20 bf16"), alternating.
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in $(seq 1 ${AB_REPS:-3}); do
  for spec in "$@"; do
    if [ "$spec" = "-" ]; then e=""; else e="$spec"; fi
    b=$(env $e python3 $R/tools/sample_one.py ${AB_ARGS:-32 1.0 20 bf16} 2>/dev/null | sed 's/.*= //')
    echo "[$spec]  $b"
  done
done
