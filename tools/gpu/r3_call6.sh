#!/bin/bash
# round 3: A/B of kernel variants (tools/ab.py: bench workload, alternating rounds)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3c6
mkdir -p $OUT
cd $ROOT
timeout 600 python tools/ab.py --ticks 120 --rounds 2 serf_amd/csrc/variants/lean.so serf_amd/csrc/variants/current.so > $OUT/ab3.log 2>&1; echo "ab rc=$?"; grep -v amdgpu.ids $OUT/ab3.log | tail -6
