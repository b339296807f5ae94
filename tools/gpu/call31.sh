#!/bin/bash
# round 2, call 31: rocprofv3 kernel-trace stats of the driver's bench command on the final state of the round
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/c31
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $ROOT/bench.py --no-cpu-baseline --no-convergence --steps 20 --warmup 5 > $OUT/trace.log 2>&1; echo "rc=$?"
grep '"metric"' $OUT/trace.log > $OUT/bench_traced.json
head -4 $OUT/trace/t_kernel_stats.csv | cut -c1-220
python - <<PY
import csv, json
rows = [r for r in csv.DictReader(open("$OUT/trace/t_kernel_trace.csv")) if "tick_kernel" in r["Kernel_Name"]]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000 for r in rows][-20:]
b = json.load(open("$OUT/bench_traced.json"))
print("last 20 launches: mean %.1f us; bench kernel_ms %.4f" % (sum(d) / len(d), b["roofline"]["kernel_ms"]))
PY
