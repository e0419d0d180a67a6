#!/bin/bash
# Run ON the GPU box: rocprofv3 kernel-time totals of the training-step bench under two environment settings (same box, alternating):
#   bash tools/prof_train_ab.sh "SF_TRAIN_SELF_PACK=1"        -> per setting: total kernel ms per step and the pack / fill families
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/trainprof_ab; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
  for v in "" "$@"; do
    tag=$(echo "${v:-base}" | tr -c 'A-Za-z0-9_\n' '_')
    rm -rf $O/stats
    ( [ -n "$v" ] && export $v; [ -n "$AB_LIB" ] && export SF_LIB_PATH=$AB_LIB; rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/train_step_bench.py --steps 3 > /dev/null 2> $O/stats.log )
    f=$(ls $O/stats/*/*kernel_stats.csv | head -1)
    cp $f $O/${tag}_$rep.csv
    python3 - "$f" "${v:-base}" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = 4.0
tot = sum(int(r["TotalDurationNs"]) for r in rows) / 1e6 / steps
def fam(k): return sum(int(r["TotalDurationNs"]) for r in rows if k in r["Name"]) / 1e6 / steps, sum(int(r["Calls"]) for r in rows if k in r["Name"]) / steps
print(f"[{sys.argv[2]}] kernel time {tot:.2f} ms/step; pack kernels {fam('pack_')[0]:.2f} ms in {fam('pack_')[1]:.0f} launches; gn_* kernels {fam('gn_')[0]:.2f} ms; wgrad* {fam('wgrad')[0]:.2f} ms; launches/step {sum(int(r['Calls']) for r in rows) / steps:.0f}")
PY
  done
done
rm -rf $O/stats
