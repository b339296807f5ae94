#!/bin/bash
# round 3, call 29: event-ring buckets with four keys in the head plane (Lamport time in the tail) — GPU parity, then the A/B
# against the two-key head (same source, -DTICK_EV_HEAD2) and the timing counters
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/c29
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "not 1m and not 64k" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -4 $OUT/pytest.log
V=serf_amd/csrc/variants
timeout 600 python tools/ab.py --ticks 120 --rounds 3 $V/evhead2.so serf_amd/csrc/libserf_sim.so > $OUT/ab.log 2>&1; echo "ab rc=$?"
grep -v amdgpu.ids $OUT/ab.log | tail -8
