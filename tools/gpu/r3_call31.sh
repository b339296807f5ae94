#!/bin/bash
# round 3, call 31: the random fan-out mode (memberlist's literal kRandomNodes, explicit per-tick CSR) on the GPU: parity sweep
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/c31
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "random_fanout or backend_is_hip" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -25 $OUT/pytest.log
