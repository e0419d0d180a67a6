#!/bin/bash
# Run ON the GPU box: phase removal (SF_SKIP_LABELS) on one workload:  bash tools/phase_removal.sh B scale steps [L0]
# the SF_* hooks exist only in the tuning build of the library (make -C syncfusion_amd/csrc tuning)
export SF_LIB_PATH=${SF_LIB_PATH:-${GRAFT_REPO_ROOT:-$(pwd)}/syncfusion_amd/lib/libsyncfusion_amd_tuning.so}
R=${GRAFT_REPO_ROOT:-$(pwd)}
B=$1; SC=$2; ST=$3; L0=${4:-45056}
for fam in "" gn_silu ln_modulate "conv_thin,thin_tail,d0_,gn_stats,conv_direct" attention conv_gemm_mt conv_gemm_v2 "conv_gemm_rs,conv_gemm_wp,conv_gemm_fast,conv_gemm_sk" "conv_cb,cb_reduce" ""; do
  echo -n "without [$fam]: "
  SF_SKIP_LABELS="$fam" python3 $R/tools/sample_one.py $B $SC $ST bf16 $L0 2>/dev/null | sed 's/.*: //'
done
