#!/bin/bash
# round 3, call 38: the bench line of the final kernel at other cluster sizes on one GPU (64 Ki ... 4 Mi nodes; 256 view slots, rings 128)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/c38
mkdir -p $OUT
cd $ROOT
for N in 65536 262144 1048576 4194304; do
  timeout 600 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-second-load --nodes-per-gpu $N --view-slots 256 --ring 128 > $OUT/bench_$N.json 2> $OUT/bench_$N.err; echo "bench $N rc=$?"
done
python - <<PY
import json
out = {}
for n in (65536, 262144, 1048576, 4194304):
    try:
        d = json.loads(open("$OUT/bench_%d.json" % n).read().strip().splitlines()[-1]); r = d["roofline"]
        out[str(n)] = {"value": d["value"], "ms_per_step": d["ms_per_step"], "kernel_ms": r["kernel_ms"], "frac": r["frac"], "layout_frac": r["layout"]["frac"],
                       "model_bound_drops": d["config"]["model_bound_drops"], "rounds_to_99_median": d["rounds_to_99"]["median"]}
        print(n, "value %.3e" % d["value"], "ms/step %.4f" % d["ms_per_step"], "kernel_ms %.4f" % r["kernel_ms"], "frac %.3f" % r["frac"], "drops", d["config"]["model_bound_drops"], "rounds", d["rounds_to_99"]["median"])
    except Exception as e:
        print(n, "failed", e)
json.dump(out, open("$OUT/size_sweep.json", "w"), indent=1)
PY
