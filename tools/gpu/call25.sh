#!/bin/bash
# round 2, call 25: one process per shard over the HIP library on one GPU (gloo for RCCL), each rank against the oracle
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/c25
mkdir -p $OUT
cd $ROOT
timeout 200 python tools/shard_procs_check.py 2 2 4 2048 > $OUT/procs_2_2_4.log 2>&1; echo "rc=$?"; grep -v "amdgpu.ids\|socket.cpp\|Gloo" $OUT/procs_2_2_4.log | tail -40
