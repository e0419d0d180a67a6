#!/bin/bash
# Run ON the GPU box: A/B of runtime environment knobs on configs[1] (bench.py), alternating.
#   bash tools/ab_env.sh "NAME=VALUE ..." "NAME2=VALUE2" ...   ("-" = no extra environment)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in $(seq 1 ${AB_REPS:-3}); do
  for spec in "$@"; do
    if [ "$spec" = "-" ]; then e=""; else e="$spec"; fi
    a=$(env $e python3 $R/bench.py --steps 50 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])" 2>/dev/null)
    echo "[$spec]  configs[1] ${a:-FAILED} steps/s"
  done
done
