#!/bin/bash
# round 6: kernel timeline of a short headline run (per-dispatch start / end: where the step's time outside the tick kernel goes)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r6t
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--fanout-model krandomnodes --no-cpu-baseline --no-convergence --no-second-load --no-long-window --steps 20 --warmup 5 ${EXTRA_ARGS:-}"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $ROOT/bench.py $ARGS > $OUT/trace.log 2>&1; echo "trace rc=$?"
grep '"metric"' $OUT/trace.log | cut -c1-600
python - <<PY
import csv, glob, collections
f = glob.glob('$OUT/trace/**/t_kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last 6 tick kernels and everything between them
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('void tick_kernel') or 'tick_kernel' in r['Kernel_Name'].split('(')[0]]
lo = idx[-7]
t0 = int(rows[lo]['Start_Timestamp'])
with open('$OUT/timeline.txt', 'w') as o:
    for r in rows[lo:]:
        name = r['Kernel_Name'].split('(')[0][-40:]
        line = f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} {(int(r['End_Timestamp']) - t0) / 1e3:9.1f} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f}  q{r.get('Queue_Id','?')} {name}"
        print(line); o.write(line + '\n')
PY
find $OUT -name '*kernel_trace.csv' -delete
