#!/bin/bash
# round 6: kernel timeline around the first recycling pass of the long window (what the pass and the ticks behind it hold)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r6tp
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--fanout-model krandomnodes --no-cpu-baseline --no-convergence --no-second-load --no-long-window --steps 80 --warmup 5"
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/trace -o t -- python $ROOT/bench.py $ARGS > $OUT/trace.log 2>&1; echo "trace rc=$?"
python - <<PY
import csv, glob
f = glob.glob('$OUT/trace/**/t_kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
i0 = next(i for i, r in enumerate(rows) if 'recycle_refd' in r['Kernel_Name'])
ticks = [i for i, r in enumerate(rows) if 'tick_kernel' in r['Kernel_Name'].split('(')[0]]
lo = max(t for t in ticks if t < i0)
lo = ticks[ticks.index(lo) - 2]
hi = ticks[min(len(ticks) - 1, ticks.index(max(t for t in ticks if t < i0)) + 5)]
t0 = int(rows[lo]['Start_Timestamp'])
with open('$OUT/timeline_pass.txt', 'w') as o:
    for r in rows[lo:hi + 1]:
        name = r['Kernel_Name'].split('(')[0][-40:]
        line = f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} {(int(r['End_Timestamp']) - t0) / 1e3:9.1f} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f}  q{r.get('Queue_Id','?')} {name}"
        print(line); o.write(line + '\n')
PY
find $OUT -name '*_trace.csv' -delete
