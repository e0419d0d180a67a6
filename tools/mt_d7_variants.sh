cd $GRAFT_REPO_ROOT
for shape in "64 44 1024 1024 3" "32 44 1024 1024 3" "32 88 1024 1024 3" "64 88 1024 1024 3"; do
  for v in 1 3 5 7 4; do
    echo "shape [$shape] v$v: $(SF_MT_VARIANT=$v python tools/gemm_one.py $shape 6 -1 -1 50 2>/dev/null | tail -1)"
  done
done
