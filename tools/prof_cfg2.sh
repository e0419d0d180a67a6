R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/cfg2prof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/sample_one.py 32 2.0 6 bf16 45056 > $O/run.txt 2> $O/stats.log
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats3 -- python3 $R/tools/sample_one.py 32 1.0 6 bf16 45056 > $O/run3.txt 2> $O/stats3.log
cp $(ls $O/stats3/*/*kernel_stats.csv | head -1) $O/kernel_stats_cfg3.csv
rm -rf $O/stats $O/stats3
cat $O/run.txt $O/run3.txt
