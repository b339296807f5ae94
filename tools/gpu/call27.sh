#!/bin/bash
# round 2, call 27: one process per shard over the HIP library on one GPU (gloo for RCCL), each rank against the oracle
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 75 python tools/shard_procs_check.py 2 2 4 2048 > $OUT/procs_check.log 2>&1; echo "rc=$?"
for r in 0 1; do echo "== rank $r progress"; tail -3 $OUT/progress_rank$r.txt; done
grep -v "amdgpu.ids\|socket.cpp\|Gloo\|resource_tracker\|warnings.warn" $OUT/procs_check.log | tail -8
