#!/bin/bash
# Run ON the GPU box: A/B of environment knobs on bench.py with extra arguments (BENCH_ARGS, e.g. "--batch 32 --scale 2.0 --steps 30"),
# alternating AB_REPS times.  Specs: "-" = no variable, or one or more NAME=value words.
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in $(seq 1 ${AB_REPS:-2}); do
  for spec in "$@"; do
    if [ "$spec" = "-" ]; then e=""; else e="$spec"; fi
    a=$(env $e python3 $R/bench.py ${BENCH_ARGS:---steps 50 --warmup 5} --no-extra --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['launches_per_eval_total'])")
    echo "[$spec] ${BENCH_ARGS:-configs[1]}: $a (steps/s, launches per evaluation)"
  done
done
