#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/c12
mkdir -p $OUT
cd $ROOT
timeout 900 python tools/ab.py --ticks 120 --rounds 3 serf_amd/csrc/variants/prev.so serf_amd/csrc/libserf_sim.so > $OUT/ab.log 2>&1; echo "ab rc=$?"; tail -1 $OUT/ab.log
