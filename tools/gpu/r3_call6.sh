#!/bin/bash
# round 3, call 6: A/B of the 8-key drain round against the same source without it
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3c6
mkdir -p $OUT
cd $ROOT
timeout 600 python tools/ab.py --ticks 120 --rounds 3 serf_amd/csrc/variants/base.so serf_amd/csrc/variants/round8.so > $OUT/ab.log 2>&1; echo "ab rc=$?"; grep -v amdgpu.ids $OUT/ab.log | tail -8
