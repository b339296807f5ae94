#!/bin/bash
# round 3, call 37: dependent loads out of the handlers — the query handler's three independent table reads issued together, the
# suspicion-timer list read as the two uint4 it is (handlers and the timer walk) — parity, then A/B against the commit before
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/c37
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "not 1m and not 64k" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -3 $OUT/pytest.log
V=serf_amd/csrc/variants
timeout 600 python tools/ab.py --ticks 120 --rounds 3 $V/head.so serf_amd/csrc/libserf_sim.so > $OUT/ab.log 2>&1; echo "ab rc=$?"
grep -v amdgpu.ids $OUT/ab.log | tail -8
