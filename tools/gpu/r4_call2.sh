#!/bin/bash
# round 4, call 2: graph build on its own stream; per-phase timing of the RF tick kernel
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4c2
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_parity_gpu.py -k "random_fanout" -x -q > $OUT/rf_tests.log 2>&1
echo "rf tests rc=$?" | tee -a $OUT/summary.txt
tail -3 $OUT/rf_tests.log
timeout 300 python bench.py --random-fanout --steps 100 --warmup 20 --no-second-load --no-cpu-baseline --no-convergence > $OUT/bench_rf.log 2>&1
echo "bench rf rc=$?" | tee -a $OUT/summary.txt
grep '"metric"' $OUT/bench_rf.log > $OUT/bench_rf.json
python - <<PY
import json
d=json.load(open("$OUT/bench_rf.json"))
print("RF value", d["value"], "ms/step", d["ms_per_step"], "kernel_ms", d["roofline"]["kernel_ms"])
PY
timeout 300 python tools/tick_timing.py 1048576 --random-fanout > $OUT/timing_rf.txt 2>&1
cat $OUT/timing_rf.txt
timeout 300 python tools/tick_timing.py 1048576 > $OUT/timing_bij.txt 2>&1
cat $OUT/timing_bij.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $ROOT/bench.py --random-fanout --steps 100 --warmup 20 --no-second-load --no-cpu-baseline --no-convergence > $OUT/trace.log 2>&1
head -8 $OUT/trace/t_kernel_stats.csv | cut -c1-160
