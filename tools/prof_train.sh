#!/bin/bash
# Run ON the GPU box: rocprofv3 kernel stats of the training-step bench
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/trainprof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/train_step_bench.py > $O/run.txt 2> $O/stats.log
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
rm -rf $O/stats
cat $O/run.txt
