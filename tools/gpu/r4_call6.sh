#!/bin/bash
# round 4, call 6: balanced classification for the random fan-out
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4c6
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_parity_gpu.py -k "random_fanout" -x -q > $OUT/rf_tests.log 2>&1
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
grep -m1 "tick_kernel" $OUT/trace/t_kernel_trace.csv | cut -c1-400
head -1 $OUT/trace/t_kernel_trace.csv
cd $ROOT
timeout 300 python bench.py $B > $OUT/bench.log 2>&1
echo "== overlapped build"; grep '"metric"' $OUT/bench.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('value', d['value'], 'ms/step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'])"
