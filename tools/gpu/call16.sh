#!/bin/bash
# round 2, call 16: packets kept at the sender (one cell per DISTINCT packet + a map word; the receiver fetches through
# the inverse of the fan-out map): quick parity subset, A/B against the push build of HEAD, then the whole GPU suite
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/c16
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests -m gpu -x -q -k "sender or small or backend" > $OUT/pytest_quick.log 2>&1; rc=$?; echo "quick rc=$rc"; tail -15 $OUT/pytest_quick.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout 600 python tools/ab.py --ticks 120 --rounds 2 serf_amd/csrc/variants/base.so serf_amd/csrc/libserf_sim.so > $OUT/ab.log 2>&1; echo "ab rc=$?"; tail -6 $OUT/ab.log
timeout 1200 python -m pytest tests -m gpu -x -q --durations=6 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -14 $OUT/pytest.log
