#!/bin/bash
# round 2, call 13: query filters / tag classes — GPU parity, then A/B of the tick kernel with the filter check
# inlined, out of line, and left out (the benchmark workload carries no filters: all three compute the same thing)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/c13
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "filters or backend or abi or small or swim_crash or four_shards" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
timeout 900 python tools/ab.py --ticks 120 --rounds 3 serf_amd/csrc/variants/qf_off.so serf_amd/csrc/libserf_sim.so serf_amd/csrc/variants/qf_noinline.so > $OUT/ab.log 2>&1; echo "ab rc=$?"; tail -1 $OUT/ab.log
