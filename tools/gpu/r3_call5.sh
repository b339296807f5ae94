#!/bin/bash
# round 3, call 5: the whole GPU suite
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3c5
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -q --durations=12 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -22 $OUT/pytest.log | cut -c1-220
