#!/bin/bash
# round 4: rf_scatter with its pairs staged in LDS (coalesced stores): parity of the build, then timing of workgroup shapes
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4sc2
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "random_fanout or rf or rccl" 2>&1 | tail -3
B="--fanout-model krandomnodes --steps 20 --warmup 5 --no-second-load --no-cpu-baseline --no-convergence --no-long-window"
cd /tmp && export TMPDIR=/tmp
one() {  # name, env...
  local name=$1; shift
  env "$@" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -o t -- python $ROOT/bench.py $B > $OUT/$name.log 2>&1
  echo "== $name"; grep '"metric"' $OUT/$name.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('value %.3e' % d['value'], 'ms/step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'parity', d.get('parity', {}).get('digest_match'))"
  python - <<PY
import csv
for r in csv.DictReader(open("$OUT/$name/t_kernel_stats.csv")):
    if "rf_" in r["Name"] or "tick_kernel" in r["Name"]:
        print("   ", r["Name"].split("(")[0][:40], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1))
PY
}
one base SERF_RF_SYNC=1
one base_async A=1
one base_async_again A=1
