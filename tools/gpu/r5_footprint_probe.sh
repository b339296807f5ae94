#!/bin/bash
# does the tick kernel's time depend on how far apart a node's view entries / ring buckets lie?  (the footprint)
mkdir -p gpurun_out/r5n
for cfg in "1024 512" "128 512" "1024 64" "128 64" "64 32"; do
  set -- $cfg
  timeout 400 python bench.py --steps 20 --warmup 5 --fanout-model krandomnodes --view-slots $1 --ring $2 --no-cpu-baseline --no-convergence --no-second-load \
    > gpurun_out/r5n/bench_v$1_r$2.json 2> gpurun_out/r5n/bench_v$1_r$2.err
  python - <<P
import json
d=json.loads(open("gpurun_out/r5n/bench_v$1_r$2.json").read().strip().splitlines()[-1])
print("view $1 ring $2:", "%.4g"%d["value"], "kernel %.4f"%d["roofline"]["kernel_ms"], "long %.4g"%d.get("value_long_window",0), "kernel_long %.4f"%d["long_window"]["kernel_ms"])
P
done
