#!/bin/bash
# Run ON the GPU box: A/B of environment knobs on any command printing one result line ($AB_CMD), alternating.
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in $(seq 1 ${AB_REPS:-2}); do
  for spec in "$@"; do
    if [ "$spec" = "-" ]; then e=""; else e="$spec"; fi
    echo "[$spec]  $(cd $R && env $e $AB_CMD 2>/dev/null | tail -1)"
  done
done
