#!/bin/bash
# Run ON the GPU box: macro-tile variants (SF_MT_VARIANT) alone on the chip for the shapes given as "B L C N taps" strings
cd $GRAFT_REPO_ROOT
for shape in "$@"; do
  for v in ${MT_VARIANTS:-1 3 5 7 4}; do
    echo "shape [$shape] v$v: $(SF_MT_VARIANT=$v python tools/gemm_one.py $shape 6 -1 -1 50 2>/dev/null | tail -1)"
  done
done
