#!/bin/bash
# planes on demand (LazyPlanes) against whole arrays at the bench configuration: kernel time (the mappings are 128 MiB chunks
# of the virtual-memory API instead of one hipMalloc: TLB reach may differ), create time, device memory in use
mkdir -p gpurun_out/r5q
for rep in 1 2; do
for eager in 0 1; do
  SERF_SIM_EAGER=$eager timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-convergence --no-second-load \
    > gpurun_out/r5q/bench_eager${eager}_$rep.json 2> gpurun_out/r5q/bench_eager${eager}_$rep.err
  python - <<P
import json
d=json.loads(open("gpurun_out/r5q/bench_eager${eager}_$rep.json").read().strip().splitlines()[-1])
b=d["fanout_models"]["bijection"]
print("eager=$eager rep $rep: krn %.4g kernel %.4f long %.4g kernel_long %.4f | bij %.4g kernel %.4f long kernel %.4f | footprint %s" % (
  d["value"], d["roofline"]["kernel_ms"], d.get("value_long_window",0), d["long_window"]["kernel_ms"], b["value"], b["kernel_ms"], b["long_window"]["kernel_ms"],
  json.dumps(d["config"]["footprint"])))
P
done
done
