#!/bin/bash
# round 3, call 22: per-phase cycles with the handler-loop counters
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/c22
mkdir -p $OUT
cd $ROOT
timeout 200 python tools/tick_timing.py > $OUT/tick_timing.txt 2>&1; echo "timing rc=$?"
cat $OUT/tick_timing.txt
