#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/c7
mkdir -p $OUT
cd $ROOT
for lib in serf_amd/csrc/variants/linstore.so serf_amd/csrc/variants/prefslot.so; do
  SERF_SIM_LIB=$ROOT/$lib timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -x -q > $OUT/pytest_$(basename $lib).log 2>&1; echo "$lib pytest rc=$?"; tail -2 $OUT/pytest_$(basename $lib).log
done
timeout 900 python tools/ab.py --ticks 120 --rounds 3 serf_amd/csrc/libserf_sim.so serf_amd/csrc/variants/linstore.so serf_amd/csrc/variants/prefslot.so serf_amd/csrc/variants/lin_prefslot.so serf_amd/csrc/variants/occ3.so > $OUT/ab.log 2>&1; echo "ab rc=$?"; tail -1 $OUT/ab.log
