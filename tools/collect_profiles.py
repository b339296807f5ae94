#!/usr/bin/env python
"""Copy the summaries of a measurement call (tools/gpu/call15.sh -> gpurun_out/c15, gpurun_out/prof_r02b) into profiles/
under their judged names and derive r02_pmc_traffic.json from the PMC summary with the calibration factors.

usage: python tools/collect_profiles.py [call_dir] [prof_dir]"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
call = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "c15")
prof = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "prof_r02b")
out = os.path.join(ROOT, "profiles")

for src, dst in (("bench_default.json", "r02_bench.json"), ("bench_20_5.json", "r02_bench_driver_args.json"),
                 ("r02_tick_kernel_pmc.json", "r02_tick_kernel_pmc.json"),
                 ("r02_tick_kernel_pmc_memory_path.json", "r02_tick_kernel_pmc_memory_path.json"),
                 ("ablation.txt", "r02_ablation.txt"), ("tick_timing.txt", "r02_tick_timing.txt")):
    shutil.copy(os.path.join(call, src), os.path.join(out, dst))
shutil.copy(os.path.join(prof, "trace", "t_kernel_stats.csv"), os.path.join(out, "r02_kernel_stats.csv"))
shutil.copy(os.path.join(prof, "bench_traced.json"), os.path.join(out, "r02_bench_under_rocprof.json"))

pmc = json.load(open(os.path.join(call, "r02_tick_kernel_pmc.json")))
c = pmc["counters"]
n = 1 << 20
read, write = 2.0 * c["FETCH_SIZE"] * 1024, c["WRITE_SIZE"] * 1024   # both counters are in KiB
algo = 1176 * n
traffic = {
    "kernel": "tick_kernel", "launches_averaged": pmc["launches"], "kernel_us_mean_profiled": pmc["kernel_us_mean"],
    "hbm_read_bytes": read, "hbm_write_bytes": write, "hbm_bytes_per_launch": read + write,
    "fetch_size_raw_kib": c["FETCH_SIZE"], "write_size_raw_kib": c["WRITE_SIZE"],
    "calibration": "reads = 2 x FETCH_SIZE, writes = WRITE_SIZE: tools/calib on this kernel's access shapes "
                   "(profiles/r02_hbm_counter_calibration.json): TCC_EA0_RDREQ counts 128-byte requests, FETCH_SIZE prices "
                   "them at 64 B (factor 0.500 for every read pattern); WRITE_SIZE exact (factor 1.000)",
    "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of `python bench.py --no-cpu-baseline "
              "--no-convergence --steps 20 --warmup 5` (tools/profile_gpu.sh r02d, tools/gpu/call22.sh: the driver's command "
              "line), mean over the 20 timed launches",
    "algorithmic_bytes_per_launch_v0": algo, "traffic_over_algorithmic": (read + write) / algo,
}
json.dump(traffic, open(os.path.join(out, "r02_pmc_traffic.json"), "w"), indent=1)
print(json.dumps({k: traffic[k] for k in ("kernel_us_mean_profiled", "hbm_bytes_per_launch", "traffic_over_algorithmic")}))
