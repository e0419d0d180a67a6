#!/bin/bash
# Run ON the GPU box: LDS-array counters per kernel for one sample_one.py workload (eager launches):  bash tools/lds_state.sh tag B scale steps dtype [L0]
# the SF_* hooks exist only in the tuning build of the library (make -C syncfusion_amd/csrc tuning)
export SF_LIB_PATH=${SF_LIB_PATH:-${GRAFT_REPO_ROOT:-$(pwd)}/syncfusion_amd/lib/libsyncfusion_amd_tuning.so}
T=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/wstate; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && export SF_NO_GRAPH=1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/${T}_l -- python3 $R/tools/sample_one.py "$@" > /dev/null 2> $O/${T}_l.log
cd $R
python3 tools/pmc_by_kernel.py $O/${T}_l $O/${T}_lds.csv > /dev/null
rm -rf $O/${T}_l
