#!/bin/bash
# round 2, call 17: work lists for the whole tick (-DTICK_BATCH: the handlers run once per tick over per-lane lists
# instead of once per packet): parity of that build, then A/B against the shipped build
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/c17
mkdir -p $OUT
cd $ROOT
SERF_SIM_LIB=$ROOT/serf_amd/csrc/variants/batch.so timeout 900 python -m pytest tests -m gpu -x -q -k "not four_shards and not host_cpp" > $OUT/pytest_batch.log 2>&1; rc=$?; echo "batch pytest rc=$rc"; tail -15 $OUT/pytest_batch.log
timeout 600 python tools/ab.py --ticks 120 --rounds 2 serf_amd/csrc/libserf_sim.so serf_amd/csrc/variants/batch.so > $OUT/ab.log 2>&1; echo "ab rc=$?"; tail -6 $OUT/ab.log
