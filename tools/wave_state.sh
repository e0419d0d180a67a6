#!/bin/bash
# Run ON the GPU box: SQ wave-state counters per kernel for one sample_one.py workload (eager launches), e.g.
#   bash tools/wave_state.sh cfg2 32 2.0 3 bf16 45056
# the SF_* hooks exist only in the tuning build of the library (make -C syncfusion_amd/csrc tuning)
export SF_LIB_PATH=${SF_LIB_PATH:-${GRAFT_REPO_ROOT:-$(pwd)}/syncfusion_amd/lib/libsyncfusion_amd_tuning.so}
T=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/wstate; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && export SF_NO_GRAPH=1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/${T}_a -- python3 $R/tools/sample_one.py "$@" > /dev/null 2> $O/${T}_a.log
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/${T}_b -- python3 $R/tools/sample_one.py "$@" > /dev/null 2> $O/${T}_b.log
cd $R
python3 tools/pmc_by_kernel.py $O/${T}_a $O/${T}_wave_state.csv > /dev/null
python3 tools/pmc_by_kernel.py $O/${T}_b $O/${T}_wave_insts.csv > /dev/null
rm -rf $O/${T}_a $O/${T}_b
