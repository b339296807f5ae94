#!/bin/bash
# round 2, call 21: the four map-word loads at kernel entry issued back to back (each sat in its own branch and got its
# own s_waitcnt: four serial round trips before the row was asked for); recycle scan on the sender-kept cells; variant
# with the first two parked broadcasts prefetched next to the sort keys.  Quick parity, then A/B.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/c21
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests -m gpu -x -q -k "sender or small or recycling or golden" > $OUT/pytest_quick.log 2>&1; rc=$?; echo "quick rc=$rc"; tail -5 $OUT/pytest_quick.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout 900 python tools/ab.py --ticks 120 --rounds 2 serf_amd/csrc/variants/prev.so serf_amd/csrc/libserf_sim.so serf_amd/csrc/variants/pendpf.so > $OUT/ab.log 2>&1; echo "ab rc=$?"; tail -8 $OUT/ab.log
SERF_SIM_LIB=$ROOT/serf_amd/csrc/variants/pendpf.so timeout 600 python -m pytest tests -m gpu -x -q -k "small or overload or soak" > $OUT/pytest_pendpf.log 2>&1; echo "pendpf rc=$?"; tail -3 $OUT/pytest_pendpf.log
