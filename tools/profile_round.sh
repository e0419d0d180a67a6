#!/bin/bash
# Run ON the GPU box (gpurun): rocprofv3 kernel stats + separate PMC passes of the bench command, summaries under gpurun_out/$1
# (every pass runs on the PRODUCT library except the two traffic passes, which need the SF_NO_PREFETCH hook of the tuning build)
set -u
TAG=${1:-prof}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
if [ "${SF_PROFILE_PRIMARY:-1}" = "1" ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $O/bench_under_profiler.json 2> $O/stats.log
# traffic passes WITHOUT the hosted weight prefetch: a prefetch workgroup's reads are charged to the launch that hosts it (the NEXT GEMM's
# weights), which would triple the apparent traffic of the small-batch GEMMs; the timed runs keep the prefetch on
export SF_NO_PREFETCH=1 SF_LIB_PATH=$R/syncfusion_amd/lib/libsyncfusion_amd_tuning.so
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 $R/bench.py --no-cpu-baseline --no-extra --steps 3 --warmup 1 --no-graph > /dev/null 2> $O/fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 $R/bench.py --no-cpu-baseline --no-extra --steps 3 --warmup 1 --no-graph > /dev/null 2> $O/write.log
unset SF_NO_PREFETCH SF_LIB_PATH
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $O/mfma -- python3 $R/bench.py --no-cpu-baseline --no-extra --steps 3 --warmup 1 --no-graph > /dev/null 2> $O/mfma.log
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $O/mfma_onset -- python3 $R/tools/onset_one.py 32 bf16 3 > /dev/null 2> $O/mfma_onset.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/onset_stats -- python3 $R/tools/onset_one.py 32 bf16 5 > /dev/null 2> $O/onset_stats.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_stats -- python3 $R/tools/train_step_bench.py > $O/train_step_under_profiler.json 2> $O/train_stats.log
cd $R
cp $(ls $O/onset_stats/*/*kernel_stats.csv | head -1) $O/onset_kernel_stats.csv
cp $(ls $O/train_stats/*/*kernel_stats.csv | head -1) $O/train_step_kernel_stats.csv
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
python3 tools/pmc_traffic.py $(ls $O/fetch/*/*counter_collection.csv | head -1) $(ls $O/write/*/*counter_collection.csv | head -1) $O/pmc_traffic.json > $O/pmc_traffic.txt
python3 tools/mfma_busy.py $O/mfma $O/pmc_mfma_by_kernel.csv > /dev/null
python3 tools/mfma_busy.py $O/mfma_onset $O/pmc_mfma_onset_by_kernel.csv > /dev/null
python3 bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
fi
cd $R
# secondary workloads: kernel stats + matrix-core busy summaries of configs[2] (batch 32, guidance 2.0), one GPU's
# share of configs[3] (batch 32, no guidance) and the reference's own evaluation shape (batch 10, 2^18 samples, guidance 2.0).
# The program itself follows `--` (no env / shell hop under rocprofv3); SF_NO_GRAPH is exported for the eager counter passes.
sec() {  # tag B scale steps L0
  local T=$1; shift
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_stats -- python3 $R/tools/sample_one.py "$@" > $O/${T}.txt 2> $O/${T}_stats.log )
  cp $(ls $O/${T}_stats/*/*kernel_stats.csv | head -1) $O/${T}_kernel_stats.csv
  ( cd /tmp && export SF_NO_GRAPH=1 && rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $O/${T}_mfma -- python3 $R/tools/sample_one.py "$@" > /dev/null 2> $O/${T}_mfma.log )
  python3 tools/mfma_busy.py $O/${T}_mfma $O/${T}_pmc_mfma_by_kernel.csv > /dev/null
  rm -rf $O/${T}_stats $O/${T}_mfma
}
if [ "${SF_PROFILE_SECONDARY:-1}" = "1" ]; then
  sec cfg2_b32_cfg 32 2.0 6 bf16 45056
  sec cfg3share_b32 32 1.0 6 bf16 45056
  sec refshape_b10_2p18 10 2.0 4 bf16 262144
  # the parity-grade fast path (fp32 activations, products from split 16-bit operands)
  sec x3_cfg1_b8 8 1.0 20 fp32x 45056
  sec x3_cfg2_b32_cfg 32 2.0 6 fp32x 45056
fi
rm -rf $O/stats $O/fetch $O/write $O/mfma $O/mfma_onset $O/onset_stats $O/train_stats
ls -la $O
