#!/bin/bash
# round 2, call 18: where the time goes in the sender-kept-packets kernel: per-phase cycles and one-part-left-out launches
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/c18
mkdir -p $OUT
cd $ROOT
timeout 300 python tools/tick_timing.py > $OUT/tick_timing.txt 2>&1; echo "timing rc=$?"; cat $OUT/tick_timing.txt
timeout 600 python tools/ablate.py > $OUT/ablation.txt 2>&1; echo "ablate rc=$?"; cat $OUT/ablation.txt
