#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/c8
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $OUT/pytest.log
timeout 600 python tools/overlap_rehearsal.py --out $OUT/overlap_rehearsal.json > $OUT/overlap.log 2>&1; echo "overlap rc=$?"; tail -6 $OUT/overlap.log
