#!/bin/bash
# Run ON the GPU box: same-box alternating A/B of whole source trees (each with its own built library) on the batch-32 legs.
#   bash tools/ab_trees.sh name1=path1 name2=path2 ...     (paths relative to the repo root; AB_REPS rounds, AB_LEGS legs)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in $(seq 1 ${AB_REPS:-2}); do
  for spec in "$@"; do
    v=${spec%%=*}; T=$R/${spec#*=}
    echo "== round $rep  $v ($T)"
    python3 $R/tools/ab_tree_legs.py $T ${AB_INNER:-3} ${AB_LEGS:-cfg1,cfg2,cfg3,ref,e2e} 2>/dev/null | sed "s/^/$v  /"
  done
done
