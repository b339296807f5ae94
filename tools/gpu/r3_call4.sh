#!/bin/bash
# round 3, call 4: whole GPU suite again + where the zero-drop boundary of 16-record packets lies
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3c4
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -x -q --durations=10 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -16 $OUT/pytest.log | cut -c1-220
for cfg in "16 0.4" "16 0.5" "16 0.6" "16 0.75" "12 0.5" "8 0.4"; do
  set -- $cfg
  f=$OUT/bench_p$1_r$2.json
  timeout 150 python bench.py --no-cpu-baseline --steps 100 --warmup 20 --pkt-records $1 --rate $2 --allow-drops > $f 2> $OUT/bench.err; echo "bench P=$1 rate=$2 rc=$?"
  python - <<PY
import json
try:
    d=json.load(open('$f')); r=d['roofline']; l=d['config']['load']
    print('  value %.3e'%d['value'], 'kernel_ms %.4f'%r['kernel_ms'], 'drops', d['config']['model_bound_drops'], 'rec/pkt', l['records_per_packet_end'], 'queued', l['queued_per_node_end'], 'deepest', l['deepest_queue'],
          'rounds', {k: d['rounds_to_99'][k] for k in ('median','p90','max','n')})
except Exception as e:
    print('  unreadable', e); print(open('$OUT/bench.err').read()[-600:])
PY
done
