#!/bin/bash
# level-1 bucket size / entry width of the graph build (no rebuild: environment switches)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4lb
mkdir -p $OUT
B="--fanout-model krandomnodes --steps 20 --warmup 5 --no-second-load --no-cpu-baseline --no-convergence --no-long-window"
cd /tmp && export TMPDIR=/tmp
for V in "10 " "11 1" "10 1" "9 " "8 "; do
  set -- $V
  LB=$1; W=${2:-}
  if [ -n "$W" ]; then export SERF_RF_WIDE=1; else unset SERF_RF_WIDE; fi
  SERF_RF_SYNC=1 SERF_RF_LB=$LB timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t_$LB$W -o t -- python $ROOT/bench.py $B > $OUT/t_$LB$W.log 2>&1
  echo "== LB $LB wide=$W"; grep '"metric"' $OUT/t_$LB$W.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('value', d['value'], 'ms/step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'])"
  grep "rf_" $OUT/t_$LB$W/t_kernel_stats.csv | cut -d, -f1-4 | cut -c1-60,150-
done
