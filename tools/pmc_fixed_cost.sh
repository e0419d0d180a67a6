#!/bin/bash
# Run ON the GPU box: where does the FIXED time of the short kernels go?  Three separate --pmc passes over the eager bench command
# (wave cycles / waits, instruction fetch, issue activity), per-kernel averages into gpurun_out/$1/*.txt
set -u
TAG=${1:-fixed}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() {  # name counters...
  local N=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$N -- python3 $R/bench.py --no-cpu-baseline --no-extra --steps 3 --warmup 1 --no-graph > /dev/null 2> $O/$N.log
  python3 $R/tools/pmc_by_kernel.py $O/$N $O/$N.csv > $O/$N.txt 2>&1
  python3 - "$O/$N" >> $O/$N.txt <<'PY'
import csv, glob, os, sys, collections
d = sys.argv[1]
kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
agg = collections.defaultdict(lambda: [0, 0])
for r in csv.DictReader(open(kt)):
    a = agg[r["Kernel_Name"].split("(")[0][:100]]
    a[0] += 1
    a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print("# durations under this pass")
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(k, n, round(t / n))
PY
  rm -rf $O/$N
}
run waves SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY
run ifetch SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_MISSES
run issue SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM
ls -la $O
