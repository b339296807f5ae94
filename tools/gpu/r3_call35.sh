#!/bin/bash
# round 3, call 35: the tick kernel of the round's final state against the kernel of the session's start (same call, alternating)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/c35
mkdir -p $OUT
cd $ROOT
V=serf_amd/csrc/variants
timeout 600 python tools/ab.py --ticks 120 --rounds 3 $V/prerf.so serf_amd/csrc/libserf_sim.so > $OUT/ab.log 2>&1; echo "ab rc=$?"
grep -v amdgpu.ids $OUT/ab.log | tail -8
