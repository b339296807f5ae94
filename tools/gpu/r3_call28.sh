#!/bin/bash
# round 3, call 28: what sits between two tick kernels — kernel trace of the bench with and without the SWIM layer (without it
# no launch carries an event in the pre-roll: the bare kernel boundary of this kernel)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/c28
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-convergence --no-second-load --steps 20 --warmup 5"
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/swim -o t -- python $ROOT/bench.py $ARGS > $OUT/swim.log 2>&1; echo "swim rc=$?"
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/noswim -o t -- python $ROOT/bench.py $ARGS --probe-interval 0 > $OUT/noswim.log 2>&1; echo "noswim rc=$?"
cd $ROOT
python - <<PY
import csv, glob, statistics as st
for w in ('swim', 'noswim'):
    f = glob.glob('$OUT/%s/**/t_kernel_trace.csv' % w, recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
    pure = [(int(b['Start_Timestamp']) - int(a['End_Timestamp'])) / 1e3 for a, b in zip(rows, rows[1:]) if 'tick_kernel' in a['Kernel_Name'] and 'tick_kernel' in b['Kernel_Name']]
    print(w, 'tick->tick gaps us: n', len(pure), 'median', st.median(pure), 'first 200 median', st.median(pure[:200]), 'last 12', [round(x, 1) for x in pure[-12:]])
PY
