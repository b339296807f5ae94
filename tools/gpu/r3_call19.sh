#!/bin/bash
# round 3, call 19: A/B — non-temporal cell loads (nt1), plus non-temporal row / key loads and stores and cell stores (nt2)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/c19
mkdir -p $OUT
cd $ROOT
V=serf_amd/csrc/variants
timeout 600 python tools/ab.py --ticks 120 --rounds 2 $V/base.so $V/nt1.so $V/nt2.so > $OUT/ab.log 2>&1; echo "ab rc=$?"
tail -20 $OUT/ab.log
