#!/bin/bash
# round 4, call 3: level-1 bucket size of the graph build, 64-byte pushed cells, ablation of the RF tick kernel
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4c3
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_parity_gpu.py -k "random_fanout" -x -q > $OUT/rf_tests.log 2>&1
echo "rf tests rc=$?" | tee -a $OUT/summary.txt
tail -3 $OUT/rf_tests.log
B="--random-fanout --steps 100 --warmup 20 --no-second-load --no-cpu-baseline --no-convergence"
cd /tmp && export TMPDIR=/tmp
for LB in 10 11 12; do
  SERF_RF_SYNC=1 SERF_RF_LB=$LB timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_lb$LB -o t -- python $ROOT/bench.py $B > $OUT/trace_lb$LB.log 2>&1
  echo "== LB $LB (sync build)"; grep '"metric"' $OUT/trace_lb$LB.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('value', d['value'], 'ms/step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'])"
  head -7 $OUT/trace_lb$LB/t_kernel_stats.csv | cut -d, -f1-4 | cut -c1-120
done
cd $ROOT
for LB in 10 11 12; do
  SERF_RF_LB=$LB timeout 300 python bench.py $B > $OUT/bench_lb$LB.log 2>&1
  echo "== LB $LB (overlapped build)"; grep '"metric"' $OUT/bench_lb$LB.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('value', d['value'], 'ms/step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'])"
done
SERF_RF_SYNC=1 timeout 600 python tools/ablate.py 1048576 --random-fanout masks=0,1,2,4,32,64 > $OUT/ablate_rf.txt 2>&1
cat $OUT/ablate_rf.txt
