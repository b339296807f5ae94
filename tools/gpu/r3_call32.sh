#!/bin/bash
# round 3, call 32: the bench with gossip targets by memberlist's literal kRandomNodes (--random-fanout) at 1 Mi nodes on the GPU:
# throughput, kernel times (tick kernel, the draw, the sort, the CSR) and rounds-to-99 %
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/c32
mkdir -p $OUT
cd $ROOT
timeout 600 python bench.py --random-fanout --steps 20 --warmup 5 --no-second-load > $OUT/bench_rf.json 2> $OUT/bench_rf.err; echo "bench rf rc=$?"; tail -3 $OUT/bench_rf.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $ROOT/bench.py --random-fanout --no-cpu-baseline --no-convergence --no-second-load --steps 20 --warmup 5 > $OUT/trace.log 2>&1; echo "trace rc=$?"
cd $ROOT
python - <<PY
import json, glob, csv
d = json.loads(open('$OUT/bench_rf.json').read().strip().splitlines()[-1]); r = d['roofline']
print('value %.3e' % d['value'], 'ms/step %.4f' % d['ms_per_step'], 'kernel_ms %.4f' % r['kernel_ms'], 'drops', d['config']['model_bound_drops'], 'parity', d.get('parity', {}).get('digest_match'), 'rounds', d['rounds_to_99'] and {k: d['rounds_to_99'][k] for k in ('median', 'p90', 'max', 'n', 'histogram')})
f = glob.glob('$OUT/trace/**/t_kernel_stats.csv', recursive=True)[0]
for row in list(csv.DictReader(open(f)))[:8]:
    print(row['Name'][:70], row['Calls'], row['AverageNs'])
PY
