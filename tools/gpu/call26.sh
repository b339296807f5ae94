#!/bin/bash
# round 2, call 26: where does the multi-process check hang?  (progress marks + python stacks of both ranks after 25 s)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 70 python tools/shard_procs_check.py 2 2 4 2048 > $OUT/procs_check.log 2>&1; echo "rc=$?"
for r in 0 1; do echo "== rank $r progress"; tail -6 $OUT/progress_rank$r.txt; echo "== rank $r stack"; grep -v "^$" $OUT/stack_rank$r.txt | head -30; done
grep -v "amdgpu.ids\|socket.cpp\|Gloo\|resource_tracker\|warnings.warn" $OUT/procs_check.log | tail -8
