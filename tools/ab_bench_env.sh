#!/bin/bash
# Run ON the GPU box: A/B of environment knobs on the headline bench (configs[1], 50 steps), alternating.
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in $(seq 1 ${AB_REPS:-2}); do
  for spec in "$@"; do
    if [ "$spec" = "-" ]; then e=""; else e="$spec"; fi
    a=$(env $e python3 $R/bench.py --steps 50 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])")
    echo "[$spec]  configs[1] $a steps/s"
  done
done
