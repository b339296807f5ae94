#!/bin/bash
# round 2, call 24: the 48-configuration sweep; the bench line at other cluster sizes on one GPU (64 Ki ... 4 Mi nodes)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/c24
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "random_configurations" > $OUT/pytest_fuzz.log 2>&1; echo "fuzz rc=$?"; tail -4 $OUT/pytest_fuzz.log
for N in 65536 262144 1048576 4194304; do
  timeout 600 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --nodes-per-gpu $N --view-slots 256 --ring 128 > $OUT/bench_$N.json 2> $OUT/bench_$N.err; echo "bench $N rc=$?"
done
python - <<PY
import json
for n in (65536, 262144, 1048576, 4194304):
    try:
        d = json.load(open("$OUT/bench_%d.json" % n)); r = d["roofline"]
        print(n, "value %.3e" % d["value"], "ms/step %.4f" % d["ms_per_step"], "kernel_ms %.4f" % r["kernel_ms"], "frac %.3f" % r["frac"], "drops", d["config"]["model_bound_drops"], "rounds", d["rounds_to_99"]["median"], "rpp", d["config"]["load"]["records_per_packet_end"])
    except Exception as e:
        print(n, "failed", e)
PY
