#!/bin/bash
# round 6: the suspicion lists' heads travel with the library's exchange (no side stream, no second collective), rfx_meta 16 bytes a thread —
# the shard tests, then one rank of the sharded kRandomNodes path through RCCL (one and two chunks), the bijection's, and the timeline
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r6h
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -q -m gpu -x -k "rccl or shard or packed or vshard or handles" 2>&1 | tail -3
ARGS="--gpus 1 --force-sharded --exchange rccl --steps 20 --warmup 5 --no-cpu-baseline --no-second-load --no-convergence"
for V in krandomnodes:1 krandomnodes:2 bijection:2; do
  set -- ${V//:/ }
  T=$1.c$2
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
EXTRA_ARGS="--force-sharded --exchange rccl --chunks 1" bash tools/gpu/r6_trace.sh 2>&1 | tail -45
