#!/bin/bash
# round 3, call 16: GPU suite on the final kernel source; which second load stays inside every bound
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3c16
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log | cut -c1-200
for cfg in "16 0.35" "16 0.3" "8 0.4" "8 0.35"; do
  set -- $cfg
  f=$OUT/bench_p$1_r$2.json
  timeout 150 python bench.py --no-cpu-baseline --no-second-load --no-convergence --steps 100 --warmup 20 --pkt-records $1 --rate $2 --allow-drops > $f 2> $OUT/bench.err; echo "bench P=$1 rate=$2 rc=$?"
  python - <<PY
import json
try:
    d=json.load(open('$f')); r=d['roofline']; l=d['config']['load']
    print('  value %.3e'%d['value'], 'kernel_ms %.4f'%r['kernel_ms'], 'drops', d['config']['model_bound_drops'], 'rec/pkt', l['records_per_packet_end'], 'queued', l['queued_per_node_end'], 'deepest', l['deepest_queue'])
except Exception as e:
    print('  unreadable', e); print(open('$OUT/bench.err').read()[-600:])
PY
done
