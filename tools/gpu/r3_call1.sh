#!/bin/bash
# round 3, call 1: the whole GPU suite with the new tests (sharded B=64 at 4 x 64 Ki, host paths over HIP events), smoke, the
# driver's bench line with the parity block, the self-launched 2-rank rehearsal on one GPU, the gloo-without-simulator repro
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3c1
mkdir -p $OUT
cd $ROOT
timeout 420 python -m pytest tests -m gpu -x -q --durations=8 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -14 $OUT/pytest.log
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 240 python bench.py --steps 20 --warmup 5 > $OUT/bench_20_5.json 2> $OUT/bench_20_5.err; echo "bench rc=$?"; tail -2 $OUT/bench_20_5.err
timeout 120 python bench.py --no-cpu-baseline > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench default rc=$?"
python - <<PY
import json
for f in ('bench_20_5','bench_default'):
    try:
        d=json.load(open('$OUT/%s.json'%f)); r=d['roofline']
        print(f, 'value %.3e'%d['value'], 'ms/step %.4f'%d['ms_per_step'], 'kernel_ms %.4f'%r['kernel_ms'], 'frac %.3f'%r['frac'], 'alg %.3f'%r['algorithmic']['frac'],
              'drops', d['config']['model_bound_drops'], 'rounds', {k: d['rounds_to_99'][k] for k in ('median','p90','max','n','histogram')}, 'parity', d.get('parity', {}).get('digest_match'))
    except Exception as e:
        print(f, 'unreadable', e)
PY
timeout 200 python bench.py --gpus 2 --backend gloo --single-device --steps 20 --warmup 5 --nodes-per-gpu 262144 --no-convergence > $OUT/bench_2ranks.json 2> $OUT/bench_2ranks.err; echo "2-rank self-launch rc=$?"; cut -c1-400 $OUT/bench_2ranks.json; tail -3 $OUT/bench_2ranks.err
for cfg in "2 4 200" "4 2 200" "4 4 200"; do
  REPRO_LIMIT=45 timeout 70 python tools/gloo_cuda_async_repro.py $cfg > $OUT/repro_$(echo $cfg | tr ' ' _).log 2>&1; echo "repro [$cfg] rc=$?"
  grep -v "amdgpu.ids\|socket.cpp\|Gloo\|resource_tracker\|warnings.warn" $OUT/repro_$(echo $cfg | tr ' ' _).log | tail -6
done
