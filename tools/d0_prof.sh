#!/bin/bash
# Run ON the GPU box: per-kernel durations of the 8-channel level under different knobs (single branch: clean durations).
# the SF_* hooks exist only in the tuning build of the library (make -C syncfusion_amd/csrc tuning)
export SF_LIB_PATH=${SF_LIB_PATH:-${GRAFT_REPO_ROOT:-$(pwd)}/syncfusion_amd/lib/libsyncfusion_amd_tuning.so}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/d0prof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export SF_TWO_BRANCH_MAX=0
i=0
for spec in "$@"; do
  i=$((i+1))
  if [ "$spec" != "-" ]; then export $spec; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/s$i -- python3 $R/tools/sample_one.py ${D0_ARGS:-32 2.0 4 bf16} > $O/run$i.txt 2> $O/log$i.txt
  echo "== [$spec] $(cat $O/run$i.txt)"
  f=$(ls $O/s$i/*/*kernel_stats.csv | head -1)
  grep -E "d0_|Li8ELi3ELi0ELi1ELi8E|thin_tail_kernelIDF16bLi8E|Li32ELi3ELi0ELi0ELi8E|gn_stats" $f | awk -F, '{printf "   %-90s calls %5d avg %8.1f us\n", substr($1,1,90), $2, $4/1000}'
  if [ "$spec" != "-" ]; then for kv in $spec; do unset ${kv%%=*}; done; fi
  rm -rf $O/s$i
done
