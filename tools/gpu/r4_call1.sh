#!/bin/bash
# round 4, call 1: random fan-out v2 (pushed packets + own bucket sort): parity, then timing + kernel trace
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4c1
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_parity_gpu.py -k "random_fanout" -x -q > $OUT/rf_tests.log 2>&1
echo "rf tests rc=$?" | tee -a $OUT/summary.txt
tail -5 $OUT/rf_tests.log
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1
echo "gpu tests rc=$?" | tee -a $OUT/summary.txt
tail -5 $OUT/gpu_tests.log
timeout 300 python bench.py --random-fanout --steps 100 --warmup 20 --no-second-load --no-cpu-baseline --no-convergence > $OUT/bench_rf.log 2>&1
echo "bench rf rc=$?" | tee -a $OUT/summary.txt
grep '"metric"' $OUT/bench_rf.log > $OUT/bench_rf.json
python - <<PY
import json
d=json.load(open("$OUT/bench_rf.json"))
print("RF value", d["value"], "ms/step", d["ms_per_step"], "kernel_ms", d["roofline"]["kernel_ms"])
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $ROOT/bench.py --random-fanout --steps 100 --warmup 20 --no-second-load --no-cpu-baseline --no-convergence > $OUT/trace.log 2>&1
python $ROOT/tools/rocpd_summary.py $OUT/trace 2>/dev/null | head -30
find $OUT/trace -name '*kernel_stats.csv' | head -2 | while read f; do head -25 "$f"; done
