#!/bin/bash
# Round 6, first GPU call: hardware probes, fp32-engine kernel breakdown at configs[1] / configs[2] shapes, full bench line with the new legs.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6_a
mkdir -p $O
$R/build/tools/r6_probes > $O/probes.txt 2>&1
python3 $R/bench.py --dtype fp32 --steps 20 --no-extra --no-cpu-baseline > $O/fp32_cfg1.json 2> $O/fp32_cfg1.err
python3 $R/bench.py --dtype fp32 --batch 32 --scale 2.0 --steps 10 --warmup 2 --no-extra --no-cpu-baseline > $O/fp32_cfg2.json 2> $O/fp32_cfg2.err
python3 $R/bench.py > $O/bench.json 2> $O/bench.err
tail -c 600 $O/bench.err
cat $O/probes.txt
