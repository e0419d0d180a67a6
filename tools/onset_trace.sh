R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/otrace; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/t -- python3 $R/tools/onset_one.py 32 bf16 3 > /dev/null 2> $O/log.txt
cp $(ls $O/t/*/*kernel_trace.csv | head -1) $O/kernel_trace.csv; rm -rf $O/t
