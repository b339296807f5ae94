#!/bin/bash
# round 2, call 28: the multi-process check with query status, at 2 and 4 ranks
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
for cfg in "2 2 4 2048" "4 1 0 2048" "4 4 4 4096"; do
  timeout 75 python tools/shard_procs_check.py $cfg > $OUT/procs_check.log 2>&1; echo "cfg [$cfg] rc=$?"
  grep -v "amdgpu.ids\|socket.cpp\|Gloo\|resource_tracker\|warnings.warn" $OUT/procs_check.log | tail -12
done
