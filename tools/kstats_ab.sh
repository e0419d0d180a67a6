#!/bin/bash
# Run ON the GPU box: rocprofv3 kernel stats of the headline bench command under each environment spec ("-" = none; several NAME=value
# words per spec allowed) -> gpurun_out/$TAG/kernel_stats_<i>.csv + bench_<i>.json.  Usage: TAG=r4c bash tools/kstats_ab.sh - SF_NO_CB=1
# the SF_* hooks exist only in the tuning build of the library (make -C syncfusion_amd/csrc tuning)
export SF_LIB_PATH=${SF_LIB_PATH:-${GRAFT_REPO_ROOT:-$(pwd)}/syncfusion_amd/lib/libsyncfusion_amd_tuning.so}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${TAG:-kstats}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
i=0
for e in "$@"; do
  i=$((i+1))
  if [ "$e" != "-" ]; then export $e; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/s$i -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $O/bench_$i.json 2> $O/log_$i.txt
  cp $(ls $O/s$i/*/*kernel_stats.csv | head -1) $O/kernel_stats_$i.csv; rm -rf $O/s$i
  echo "[$e] -> kernel_stats_$i.csv" >> $O/index.txt
  if [ "$e" != "-" ]; then for kv in $e; do unset ${kv%%=*}; done; fi
done
