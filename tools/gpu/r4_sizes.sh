#!/bin/bash
# round 4: the bench line (both fan-out models) at other cluster sizes on one GPU (64 Ki ... 4 Mi nodes; 256 view slots, rings 128).
# 4 Mi nodes: the graph build's 64-bit entries (a pair id and a target's offset no longer fit 32 bits) at scale.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4sz
mkdir -p $OUT
cd $ROOT
for N in 65536 262144 4194304; do
  timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-second-load --no-long-window --nodes-per-gpu $N --view-slots 256 --ring 128 > $OUT/bench_$N.json 2> $OUT/bench_$N.err; echo "bench $N rc=$?"
done
python - <<PY
import json
out = {}
for n in (65536, 262144, 4194304):
    try:
        d = json.loads([l for l in open("$OUT/bench_%d.json" % n) if l.startswith("{")][-1])
        out[str(n)] = {m: {"value": v["value"], "ms_per_step": v["ms_per_step"], "kernel_ms": v["kernel_ms"], "frac": v["roofline"]["frac"],
                           "model_bound_drops": v["model_bound_drops"], "rounds_to_99_median": (v.get("rounds_to_99") or {}).get("median")}
                       for m, v in d["fanout_models"].items()}
        print(n, {m: ("%.3e" % v["value"], "%.4f" % v["kernel_ms"], v["rounds_to_99_median"], v["model_bound_drops"]) for m, v in out[str(n)].items()})
    except Exception as e:
        print(n, "failed", e)
json.dump(out, open("$OUT/size_sweep.json", "w"), indent=1)
PY
