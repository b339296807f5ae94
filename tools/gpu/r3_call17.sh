#!/bin/bash
# round 3, call 17: the GPU suite on the request list without a memset / copy between two ticks and the export / import
# hand-over; one PROCESS per shard on the one GPU with slot-less suspicions crossing the shards; the bench line
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/c17
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
timeout 120 python tools/shard_procs_check.py 2 2 2 1024 0.12 > $OUT/procs_2x2.log 2>&1; echo "procs 2x2 loss .12 rc=$?"; tail -3 $OUT/procs_2x2.log
timeout 120 python tools/shard_procs_check.py 2 1 4 2048 > $OUT/procs_2x1.log 2>&1; echo "procs 2x1 rc=$?"; tail -3 $OUT/procs_2x1.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_20_5.json 2> $OUT/bench_20_5.err; echo "bench rc=$?"
timeout 300 python bench.py --no-cpu-baseline --no-second-load > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench default rc=$?"
python - <<PY
import json
for f in ('bench_20_5','bench_default'):
    try:
        d=json.loads(open('$OUT/%s.json'%f).read().strip().splitlines()[-1]); r=d['roofline']
        print(f, 'value %.3e'%d['value'], 'ms/step %.4f'%d['ms_per_step'], 'kernel_ms %.4f'%r['kernel_ms'], 'gap us %.1f'%((d['ms_per_step']-r['kernel_ms'])*1e3), 'drops', d['config']['model_bound_drops'], 'parity', d.get('parity', {}).get('digest_match'))
    except Exception as e:
        print(f, 'unreadable', e)
PY
