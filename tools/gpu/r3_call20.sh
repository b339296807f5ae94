#!/bin/bash
# round 3, call 20: host path A/B — the request-list event riding on the tick's dispatch and the one-launch slot allocation
# (new) against the previous build (variants/base.so): bench.py with the driver's arguments, alternating, then the GPU tests
# that exercise slot allocation, recycling and the request list.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/c20
mkdir -p $OUT
cd $ROOT
ARGS="--no-cpu-baseline --no-convergence --no-second-load --steps 20 --warmup 5"
for r in 1 2; do
  for v in base new; do
    LIB=$ROOT/serf_amd/csrc/libserf_sim.so; [ $v = base ] && LIB=$ROOT/serf_amd/csrc/variants/base.so
    SERF_SIM_LIB=$LIB timeout 200 python bench.py $ARGS > $OUT/bench_${v}_$r.json 2> $OUT/bench_${v}_$r.err; echo "bench $v $r rc=$?"
  done
done
ARGS2="--no-cpu-baseline --no-convergence --no-second-load --steps 300 --warmup 20"
for v in base new; do
  LIB=$ROOT/serf_amd/csrc/libserf_sim.so; [ $v = base ] && LIB=$ROOT/serf_amd/csrc/variants/base.so
  SERF_SIM_LIB=$LIB timeout 200 python bench.py $ARGS2 > $OUT/bench300_${v}.json 2> $OUT/bench300_${v}.err; echo "bench300 $v rc=$?"
done
python - <<PY
import json, glob
for f in sorted(glob.glob('$OUT/bench*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); r = d['roofline']
        print(f.split('/')[-1], 'value %.4e' % d['value'], 'ms/step %.4f' % d['ms_per_step'], 'stream %.4f' % r['stream_ms_per_step'], 'kernel_ms %.4f' % r['kernel_ms'], 'drops', d['config']['model_bound_drops'])
    except Exception as e:
        print(f, 'unreadable', e)
PY
timeout 900 python -m pytest tests -m gpu -x -q -k "(random_configurations and not paged) or recycl or swim or timer_on or bench_configuration_64k or memberlist_flags or long_soak or host_cpp" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -5 $OUT/pytest.log
