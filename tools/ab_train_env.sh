#!/bin/bash
# Run ON the GPU box: same-box alternating A/B of environment settings on the training step (graph replay = GPU-bound, and eager).
#   bash tools/ab_train_env.sh "" "SF_TRAIN_SELF_PACK=1" ...        (AB_REPS rounds, default 3)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in $(seq 1 ${AB_REPS:-3}); do
  for v in "$@"; do
    g=$(env $v python3 $R/tools/train_step_bench.py --steps 6 --graph 2>/dev/null | grep '^{' | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.readline()); print("graph %.2f ms (fwd+bwd %.2f, opt %.2f)" % (d["step_ms"], d["fwd_bwd_ms"], d["optimizer_ms"]))')
    e=$(env $v python3 $R/tools/train_step_bench.py --steps 6 2>/dev/null | grep '^{' | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.readline()); print("eager %.2f ms (fwd %.2f, bwd %.2f, opt %.2f)" % (d["step_ms"], d["forward_ms"], d["backward_ms"], d["optimizer_ms"]))')
    echo "[${v:--}]  $g   $e"
  done
done
