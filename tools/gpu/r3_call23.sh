#!/bin/bash
# round 3, call 23: A/B of the tail second look (ring duplicates whose key sits in the tail of a full bucket are retired in
# the fast path) against the previous kernel, then the timing counters of the new build
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/c23
mkdir -p $OUT
cd $ROOT
V=serf_amd/csrc/variants
timeout 600 python tools/ab.py --ticks 120 --rounds 2 $V/base.so serf_amd/csrc/libserf_sim.so $(ls $V/*.so | grep -v base.so) > $OUT/ab.log 2>&1; echo "ab rc=$?"
grep -v amdgpu.ids $OUT/ab.log | tail -12
timeout 200 python tools/tick_timing.py > $OUT/tick_timing.txt 2>&1; echo "timing rc=$?"
grep -v amdgpu.ids $OUT/tick_timing.txt
