#!/bin/bash
# Run ON the GPU box: sweep of the stand-alone 8-channel-level timing.  Build first (here or on the box):
#   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I syncfusion_amd/csrc -mllvm -amdgpu-kernarg-preload-count=16 tools/d0_bench.hip -o build/d0_bench
# the SF_* hooks exist only in the tuning build of the library (make -C syncfusion_amd/csrc tuning)
export SF_LIB_PATH=${SF_LIB_PATH:-${GRAFT_REPO_ROOT:-$(pwd)}/syncfusion_amd/lib/libsyncfusion_amd_tuning.so}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export SF_D0_MIN_ROWS=0   # time the vector kernels at every size (the engine switches to them from 256 K positions per launch)
for cfg in "64 11264 2048" "64 11264 992" "64 11264 512" "32 11264 1408" "32 11264 992" "2 11264 96" "64 45056 992" "10 262144 992"; do
  for w in 4 8 16; do
    SF_D0_WAVES=$w timeout 60 $R/build/d0_bench $cfg
  done
done
