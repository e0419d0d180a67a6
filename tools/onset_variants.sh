#!/bin/bash
# Run ON the GPU box: VideoOnsetNet forward time under macro-tile variant choices for the video geometry (alternating, 2 reps).
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do
  for spec in "$@"; do
    if [ "$spec" = "-" ]; then e=""; else e="$spec"; fi
    echo "[$spec]  $(env $e python3 $R/tools/onset_time.py 32 bf16 2>/dev/null | tail -1)"
  done
done
