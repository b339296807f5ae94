#!/usr/bin/env python
"""Copy the summaries of tools/gpu/r3_measure.sh (gpurun_out/r3m) into profiles/ under their judged names and derive
profiles/r03_pmc_traffic.json — HBM bytes per tick-kernel launch with the calibration factors of
profiles/r02_hbm_counter_calibration.json — stamped with the commit and the hash of the kernel source it was measured on
(bench.py reports the figure as roofline.traffic only while that hash is the current one).

usage: python tools/collect_r3.py [call_dir]"""
import hashlib
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
call = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r3m")
out = os.path.join(ROOT, "profiles")


def last_json(path):
    text = open(path).read().strip()
    try:
        return json.loads(text)                    # a JSON document ...
    except json.JSONDecodeError:
        return json.loads(text.splitlines()[-1])   # ... or a log whose last line is one


for src, dst in (("bench_default.json", "r03_bench.json"), ("bench_20_5.json", "r03_bench_driver_args.json"),
                 ("bench_traced.json", "r03_bench_under_rocprof.json"), ("tick_kernel_pmc.json", "r03_tick_kernel_pmc.json")):
    json.dump(last_json(os.path.join(call, src)), open(os.path.join(out, dst), "w"), indent=1)
shutil.copy(os.path.join(call, "trace", [d for d in os.listdir(os.path.join(call, "trace"))][0], "t_kernel_stats.csv")
            if not os.path.exists(os.path.join(call, "trace", "t_kernel_stats.csv")) else os.path.join(call, "trace", "t_kernel_stats.csv"),
            os.path.join(out, "r03_kernel_stats.csv"))
pmc = json.load(open(os.path.join(call, "tick_kernel_pmc.json")))
c = pmc["counters"]
n = 1 << 20
read, write = 2.0 * c["FETCH_SIZE"] * 1024, c["WRITE_SIZE"] * 1024   # both counters are in KiB
sha = hashlib.sha256(open(os.path.join(ROOT, "serf_amd", "csrc", "serf_sim.hip"), "rb").read()).hexdigest()[:16]
commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
traffic = {
    "kernel": "tick_kernel", "launches_averaged": pmc["launches"], "kernel_us_mean_profiled": pmc["kernel_us_mean"],
    "hbm_read_bytes": read, "hbm_write_bytes": write, "hbm_bytes_per_launch": read + write,
    "fetch_size_raw_kib": c["FETCH_SIZE"], "write_size_raw_kib": c["WRITE_SIZE"],
    "commit": commit, "kernel_source_sha16": sha, "steps": 20, "warmup": 5,
    "workload": "bench.py defaults: 1 Mi nodes, fan-out 4, 0.25 API ops/tick, 4 records per packet, the 20 timed launches of --steps 20 --warmup 5",
    "calibration": "reads = 2 x FETCH_SIZE, writes = WRITE_SIZE: tools/calib on this kernel's access shapes "
                   "(profiles/r02_hbm_counter_calibration.json): TCC_EA0_RDREQ counts 128-byte requests, FETCH_SIZE prices "
                   "them at 64 B (factor 0.500 for every read pattern); WRITE_SIZE exact (factor 1.000)",
    "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes of `python bench.py --no-cpu-baseline "
              "--no-convergence --no-second-load --steps 20 --warmup 5` (tools/gpu/r3_measure.sh), mean over the 20 timed launches",
    "algorithmic_bytes_per_launch_v0": 1176 * n, "traffic_over_algorithmic": (read + write) / (1176 * n),
}
json.dump(traffic, open(os.path.join(out, "r03_pmc_traffic.json"), "w"), indent=1)
print(json.dumps({k: traffic[k] for k in ("kernel_us_mean_profiled", "hbm_bytes_per_launch", "traffic_over_algorithmic", "commit", "kernel_source_sha16")}))
