#!/bin/bash
# round 6, last call: the GPU suite and smoke(), the bench line at the driver's arguments, the one-rank RCCL lines of both fan-out models
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r6z
mkdir -p $OUT
cd $ROOT
if [ "${SKIP_TESTS:-0}" != 1 ]; then
  timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -3
  python -c 'import __graft_entry__ as g; g.smoke()'
fi
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_20_5.json 2> $OUT/bench_20_5.err; echo "bench (driver args) rc=$?"; cp bench_detail.json $OUT/bench_20_5_detail.json; wc -c $OUT/bench_20_5.json
timeout 400 python bench.py --gpus 1 --force-sharded --fanout-model krandomnodes --exchange rccl --chunks 1 --steps 20 --warmup 5 --no-cpu-baseline --no-second-load > $OUT/bench_one_rank_rccl_krandomnodes.json 2> $OUT/bench_one_rank_rccl_krandomnodes.err; echo "one rank kRandomNodes rc=$?"; cp bench_detail.json $OUT/bench_one_rank_rccl_krandomnodes_detail.json
timeout 400 python bench.py --gpus 1 --force-sharded --fanout-model krandomnodes --exchange rccl --chunks 2 --steps 20 --warmup 5 --no-cpu-baseline --no-second-load --no-convergence > $OUT/bench_one_rank_rccl_krandomnodes_c2.json 2> $OUT/bench_one_rank_rccl_krandomnodes_c2.err; echo "one rank kRandomNodes, two chunks rc=$?"
timeout 400 python bench.py --gpus 1 --force-sharded --fanout-model bijection --exchange rccl --chunks 2 --steps 20 --warmup 5 --no-cpu-baseline --no-second-load > $OUT/bench_one_rank_rccl.json 2> $OUT/bench_one_rank_rccl.err; echo "one rank bijection rc=$?"
python - <<PY
import json
for f in ("bench_20_5", "bench_one_rank_rccl_krandomnodes", "bench_one_rank_rccl_krandomnodes_c2", "bench_one_rank_rccl"):
    try:
        d = json.load(open("$OUT/" + f + ".json"))
        print(f, d["value"], d["ms_per_step"], d.get("value_long_window"), d["roofline"].get("frac"), d["roofline"].get("frac_measured"), d.get("exchange"), d.get("parity"))
    except Exception as e:
        print(f, "no line:", e)
PY
