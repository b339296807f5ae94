#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/c6
mkdir -p $OUT
cd $ROOT
timeout 900 python tools/ab.py --ticks 120 --rounds 3 serf_amd/csrc/libserf_sim.so serf_amd/csrc/variants/early.so serf_amd/csrc/variants/pref4.so serf_amd/csrc/variants/both.so > $OUT/ab.log 2>&1; echo "ab rc=$?"; tail -1 $OUT/ab.log
timeout 300 python tools/tick_timing.py > $OUT/tick_timing.txt 2>&1; tail -14 $OUT/tick_timing.txt
