#!/bin/bash
# round 4, call 10: full GPU suite (RCCL world-1, third model), the driver's bench line (both fan-out models), the one-rank RCCL bench
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4c10
mkdir -p $OUT
cd $ROOT
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1
echo "gpu tests rc=$?" | tee -a $OUT/summary.txt
tail -12 $OUT/gpu_tests.log
( time timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver.log 2>$OUT/bench_driver.err ) 2>&1 | grep real
echo "bench rc=$?" | tee -a $OUT/summary.txt
grep '"metric"' $OUT/bench_driver.log > $OUT/bench_driver.json
tail -3 $OUT/bench_driver.err
python - <<PY
import json
d=json.load(open("$OUT/bench_driver.json"))
print("HEADLINE", d["config"]["fanout_model"], d["value"], "ms/step", d["ms_per_step"], "kernel_ms", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"])
for k,v in d["fanout_models"].items():
    print(k, v["value"], v["ms_per_step"], v["kernel_ms"], "long:", v.get("long_window",{}).get("value"), "r99", v["rounds_to_99"]["histogram"] if v["rounds_to_99"] else None)
print("parity", {k:(v.get("digest_match"), v.get("digest_match_per_tick")) if isinstance(v,dict) else v for k,v in d["parity"].items()})
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["single_thread_value"])
print("second", d.get("second_load",{}).get("value"))
PY
timeout 600 python bench.py --gpus 1 --force-sharded --exchange rccl --chunks 2 --steps 20 --warmup 5 --no-cpu-baseline --no-second-load > $OUT/bench_rccl1.log 2>$OUT/bench_rccl1.err
echo "bench force-sharded rc=$?" | tee -a $OUT/summary.txt
grep '"metric"' $OUT/bench_rccl1.log > $OUT/bench_rccl1.json
tail -3 $OUT/bench_rccl1.err
python - <<PY
import json
d=json.load(open("$OUT/bench_rccl1.json"))
print("force-sharded", d["value"], d["ms_per_step"], d["distributed"], d["exchange"]["exchange_ms"], d["exchange"]["kernel_ms"])
PY
