#!/bin/bash
# round 6: events riding on the pack / heads dispatches (one rank through RCCL), then an A/B of tick-kernel variants
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r6h
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests -q -m gpu -x -k "rccl or packed" 2>&1 | tail -2
ARGS="--gpus 1 --force-sharded --exchange rccl --steps 20 --warmup 5 --no-cpu-baseline --no-second-load --no-convergence"
for V in krandomnodes:1 bijection:2; do
  set -- ${V//:/ }
  T=$1.c$2.ev
  timeout 300 python bench.py $ARGS --fanout-model $1 --chunks $2 > $OUT/$T.json 2> $OUT/$T.err || echo "$T failed"
  python - <<PY
import json
try:
    d = json.load(open('$OUT/$T.json'))
    print('$T', d['value'], d['ms_per_step'], d.get('value_long_window'), d['roofline']['kernel_ms'], d.get('exchange'), d.get('parity'))
except Exception as e:
    print('$T', 'no line', e)
PY
done
# (the variant builds are made by hand into gpurun_tmp/ — e.g. hipcc ... -DHSH_LOW -o gpurun_tmp/libserf_sim_hsh.so — and are not kept)
LIBS=$(ls gpurun_tmp/libserf_sim_*.so 2>/dev/null)
[ -n "$LIBS" ] && timeout 900 python tools/ab.py --fanout-model krandomnodes --ticks 120 --rounds 3 serf_amd/csrc/libserf_sim.so $LIBS 2>&1 | tail -30
