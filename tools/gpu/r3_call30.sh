#!/bin/bash
# round 3, call 30: the driver's bench line on the two event-ring layouts, alternating
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/c30
mkdir -p $OUT
cd $ROOT
ARGS="--no-cpu-baseline --no-convergence --no-second-load --steps 20 --warmup 5"
for r in 1 2 3; do
  for v in evhead2 new; do
    LIB=$ROOT/serf_amd/csrc/libserf_sim.so; [ $v = evhead2 ] && LIB=$ROOT/serf_amd/csrc/variants/evhead2.so
    SERF_SIM_LIB=$LIB timeout 200 python bench.py $ARGS > $OUT/bench_${v}_$r.json 2> $OUT/bench_${v}_$r.err; echo "bench $v $r rc=$?"
  done
done
python - <<PY
import json, glob
for f in sorted(glob.glob('$OUT/bench*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); r = d['roofline']
        print(f.split('/')[-1], 'value %.4e' % d['value'], 'ms/step %.4f' % d['ms_per_step'], 'kernel_ms %.4f' % r['kernel_ms'], 'max %.4f' % r['kernel_ms_max'], 'drops', d['config']['model_bound_drops'])
    except Exception as e:
        print(f, 'unreadable', e)
PY
