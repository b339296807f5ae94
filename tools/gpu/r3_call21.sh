#!/bin/bash
# round 3, call 21: where the tick goes now — per-phase cycles (-DTICK_TIMING build) and the leave-one-out ablation
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/c21
mkdir -p $OUT
cd $ROOT
timeout 200 python tools/tick_timing.py > $OUT/tick_timing.txt 2>&1; echo "timing rc=$?"
cat $OUT/tick_timing.txt
timeout 300 python tools/ablate.py > $OUT/ablation.txt 2>&1; echo "ablate rc=$?"
cat $OUT/ablation.txt
