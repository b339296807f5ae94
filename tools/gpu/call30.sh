#!/bin/bash
# round 2, call 30: the state the round ends in: whole GPU suite, smoke, the driver's bench line
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/c30
mkdir -p $OUT
cd $ROOT
timeout 170 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 60 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_20_5.json 2> $OUT/bench_20_5.err; echo "bench rc=$?"
timeout 60 python bench.py --no-cpu-baseline > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench default rc=$?"
python -c "
import json
for f in ('bench_20_5','bench_default'):
    d=json.load(open('$OUT/%s.json'%f)); r=d['roofline']
    print(f, 'value %.3e'%d['value'], 'ms/step %.4f'%d['ms_per_step'], 'kernel_ms %.4f'%r['kernel_ms'], 'frac %.3f'%r['frac'], 'drops', d['config']['model_bound_drops'], 'rounds', d['rounds_to_99']['median'])
"
