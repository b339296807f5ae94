#!/bin/bash
# round 2, call 29: the next packet requested behind the slot-map loads (vmcnt counts in issue order: issued first it
# made the slot-map wait an HBM round trip and the head wait a second one): quick parity, A/B, then the whole suite
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/c29
mkdir -p $OUT
cd $ROOT
timeout 120 python -m pytest tests -m gpu -x -q -k "sender or small or random_conf or four_shards" > $OUT/pytest_quick.log 2>&1; rc=$?; echo "quick rc=$rc"; tail -3 $OUT/pytest_quick.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout 150 python tools/ab.py --ticks 120 --rounds 2 serf_amd/csrc/variants/head.so serf_amd/csrc/libserf_sim.so > $OUT/ab.log 2>&1; echo "ab rc=$?"; tail -5 $OUT/ab.log | cut -c1-400
