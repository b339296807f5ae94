#!/bin/bash
# round 4, call 4: random fan-out "pull v3" (64-byte sender cells with the map word inside, entry -> cell)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4c4
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_parity_gpu.py -k "random_fanout" -x -q > $OUT/rf_tests.log 2>&1
echo "rf tests rc=$?" | tee -a $OUT/summary.txt
tail -15 $OUT/rf_tests.log
B="--random-fanout --steps 100 --warmup 20 --no-second-load --no-cpu-baseline --no-convergence"
cd /tmp && export TMPDIR=/tmp
SERF_RF_SYNC=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $ROOT/bench.py $B > $OUT/trace.log 2>&1
echo "== sync build"; grep '"metric"' $OUT/trace.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('value', d['value'], 'ms/step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'])"
python - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/trace/t_kernel_stats.csv")))
for r in rows[:9]:
    print(f"{r['Name'][:60]:60s} {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:9.1f} us  {float(r['Percentage']):5.1f}%")
PY
cd $ROOT
for LB in 10 11; do
SERF_RF_LB=$LB timeout 300 python bench.py $B > $OUT/bench_lb$LB.log 2>&1
echo "== overlapped build LB $LB"; grep '"metric"' $OUT/bench_lb$LB.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('value', d['value'], 'ms/step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'])"
done
SERF_RF_SYNC=1 timeout 600 python tools/ablate.py 1048576 --random-fanout masks=0,1,2,4,32,64 > $OUT/ablate_rf.txt 2>&1
cat $OUT/ablate_rf.txt
