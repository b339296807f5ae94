#!/bin/bash
# round 5: the bench line (both fan-out models) at other cluster sizes on one GPU, AT THE BENCH'S OWN view / ring sizes (1 024 view
# slots, rings of 512 — the reference's event_buffer_size): with whole arrays that is 67 GiB per Mi nodes and 4 Mi nodes do not fit;
# with planes on demand (LazyPlanes) 8 Mi nodes do.  8 Mi nodes: memberlist's kRandomNodes over 2^23 nodes in ONE handle (round 4
# refused above 2^23 on shards and ran 4 Mi at 256 view slots).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5sz
mkdir -p $OUT
cd $ROOT
for N in 262144 4194304 8388608; do
  timeout 600 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-second-load --no-long-window --nodes-per-gpu $N > $OUT/bench_$N.json 2> $OUT/bench_$N.err; echo "bench $N rc=$?"
done
python - <<PY
import json
out = {"what": "bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-second-load --no-long-window --nodes-per-gpu N (view_slots 1024, rings 512: the bench's own), one MI355X"}
for n in (262144, 4194304, 8388608):
    try:
        d = json.loads([l for l in open("$OUT/bench_%d.json" % n) if l.startswith("{")][-1])
        out[str(n)] = {m: {"value": v["value"], "ms_per_step": v["ms_per_step"], "kernel_ms": v["kernel_ms"], "frac": v["roofline"]["frac"],
                           "model_bound_drops": v["model_bound_drops"], "rounds_to_99_median": (v.get("rounds_to_99") or {}).get("median"),
                           "footprint": v.get("footprint")}
                       for m, v in d["fanout_models"].items()}
        print(n, {m: ("%.3e" % v["value"], "%.4f" % v["kernel_ms"], v["rounds_to_99_median"], v["model_bound_drops"], (v["footprint"] or {}).get("device_GiB_in_use_per_Mi_nodes")) for m, v in out[str(n)].items()})
    except Exception as e:
        print(n, "failed", e)
        import subprocess; print(open("$OUT/bench_%d.err" % n).read()[-800:])
json.dump(out, open("$OUT/size_sweep.json", "w"), indent=1)
PY
