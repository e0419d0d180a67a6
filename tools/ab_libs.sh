for rep in 1 2; do
for v in old v1 v2 new; do
  if [ $v = new ]; then L=$GRAFT_REPO_ROOT/syncfusion_amd/lib/libsyncfusion_amd.so; else L=$GRAFT_REPO_ROOT/syncfusion_amd/lib/libsf_$v.so; fi
  SF_LIB_PATH=$L python bench.py --steps 50 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'])"
done
done
