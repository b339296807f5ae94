#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/c5
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest.log
timeout 600 python tools/ab.py --ticks 120 --rounds 3 serf_amd/csrc/variants/base.so serf_amd/csrc/variants/mapv2.so serf_amd/csrc/libserf_sim.so > $OUT/ab.log 2>&1; echo "ab rc=$?"; tail -1 $OUT/ab.log
bash tools/gpu/pmc_ab.sh split serf_amd/csrc/libserf_sim.so 2>&1 | tail -1
