#!/bin/bash
# Run ON the GPU box: rocprofv3 kernel stats of the bench command under two environments ($1 and $2, "-" = none) -> gpurun_out/kstats/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/kstats; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
i=0
for e in "$@"; do
  i=$((i+1))
  if [ "$e" != "-" ]; then export $e; fi   # one or more NAME=value words
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/s$i -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $O/bench_$i.json 2> $O/log_$i.txt
  cp $(ls $O/s$i/*/*kernel_stats.csv | head -1) $O/kernel_stats_$i.csv; rm -rf $O/s$i
  if [ "$e" != "-" ]; then for kv in $e; do unset ${kv%%=*}; done; fi
done
