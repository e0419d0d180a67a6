#!/bin/bash
# Run ON the GPU box: sweep of the stand-alone 8-channel-level timing (tools/d0_bench.hip variants built into build/)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for v in "$@"; do
  echo "== $v"
  for cfg in "64 11264 2048" "64 11264 992" "64 11264 512" "32 11264 1408" "32 11264 992" "2 11264 96"; do
    for w in 4 8 16; do
      SF_D0_WAVES=$w timeout 60 $R/build/d0_bench_$v $cfg
    done
  done
done
