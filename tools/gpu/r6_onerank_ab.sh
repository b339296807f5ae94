#!/bin/bash
# round 6: one rank of the sharded kRandomNodes path through RCCL — the priority of the build stream
# (a priority level = hardware queues of its own), one or two sender chunks
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r6ab
mkdir -p $OUT
cd $ROOT
ARGS="--gpus 1 --force-sharded --fanout-model krandomnodes --exchange rccl --steps 20 --warmup 5 --no-cpu-baseline --no-second-load --no-convergence"
# a variant = chunks:build-stream priority (high | low | n)
for V in ${VARIANTS:-1:high 1:n 2:high 2:n}; do
  set -- ${V//:/ }
  export SERF_RF_PRIO=$2
  T=c$1.rf$2
  timeout 300 python bench.py $ARGS --chunks $1 > $OUT/$T.json 2> $OUT/$T.err || echo "$T failed"
  python - <<PY
import json
try:
    d = json.load(open('$OUT/$T.json'))
    print('$T', d['value'], d['ms_per_step'], d.get('value_long_window'), d['roofline']['kernel_ms'], d.get('exchange'), d.get('parity'))
except Exception as e:
    print('$T', 'no line', e)
PY
done
