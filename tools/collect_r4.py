#!/usr/bin/env python
"""Copy the summaries of tools/gpu/r4_measure.sh (gpurun_out/r4m) into profiles/ under their judged names and derive, for
EACH fan-out model, profiles/r04_pmc_traffic_<model>.json — HBM bytes per tick-kernel launch with the calibration factors
of profiles/r02_hbm_counter_calibration.json — stamped with the commit and the hash of the kernel source it was measured on
(bench.py reports the figure as roofline.traffic / roofline.frac_measured only while that hash is the current one).

usage: python tools/collect_r4.py [call_dir]"""
import hashlib
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
call = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r4m")
out = os.path.join(ROOT, "profiles")


def last_json(path):
    text = open(path).read().strip()
    try:
        return json.loads(text)                    # a JSON document ...
    except json.JSONDecodeError:
        return json.loads([ln for ln in text.splitlines() if ln.startswith("{")][-1])   # ... or a log with one JSON line in it


for src, dst in (("bench_default.json", "r04_bench.json"), ("bench_20_5.json", "r04_bench_driver_args.json"),
                 ("bench_one_rank_rccl.json", "r04_bench_one_rank_rccl.json"),
                 ("bench_one_rank_rccl_krandomnodes.json", "r04_bench_one_rank_rccl_krandomnodes.json")):
    if os.path.exists(os.path.join(call, src)):
        json.dump(last_json(os.path.join(call, src)), open(os.path.join(out, dst), "w"), indent=1)
sha = hashlib.sha256(open(os.path.join(ROOT, "serf_amd", "csrc", "serf_sim.hip"), "rb").read()).hexdigest()[:16]
commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
n = 1 << 20
for model in ("krandomnodes", "bijection"):
    d = os.path.join(call, model)
    json.dump(last_json(os.path.join(d, "bench_traced.json")), open(os.path.join(out, f"r04_bench_under_rocprof_{model}.json"), "w"), indent=1)
    shutil.copy(os.path.join(d, "trace", "t_kernel_stats.csv"), os.path.join(out, f"r04_kernel_stats_{model}.csv"))
    pmc = json.load(open(os.path.join(d, "tick_kernel_pmc.json")))
    json.dump(pmc, open(os.path.join(out, f"r04_tick_kernel_pmc_{model}.json"), "w"), indent=1)
    c = pmc["counters"]
    read, write = 2.0 * c["FETCH_SIZE"] * 1024, c["WRITE_SIZE"] * 1024   # both counters are in KiB
    traffic = {
        "kernel": "tick_kernel", "fanout_model": model, "launches_averaged": pmc["launches"], "kernel_us_mean_profiled": pmc["kernel_us_mean"],
        "hbm_read_bytes": read, "hbm_write_bytes": write, "hbm_bytes_per_launch": read + write,
        "hbm_gbps_over_the_profiled_launches": (read + write) / pmc["kernel_us_mean"] / 1e3,
        "frac_of_8_tbps": (read + write) / pmc["kernel_us_mean"] / 1e3 / 8000.0,
        "fetch_size_raw_kib": c["FETCH_SIZE"], "write_size_raw_kib": c["WRITE_SIZE"],
        "commit": commit, "kernel_source_sha16": sha, "steps": 20, "warmup": 5,
        "workload": f"bench.py defaults, fan-out model {model}: 1 Mi nodes, fan-out 4, 0.25 API ops/tick, 4 records per packet, the 20 timed "
                    "launches of --steps 20 --warmup 5",
        "calibration": "reads = 2 x FETCH_SIZE, writes = WRITE_SIZE: tools/calib on this kernel's access shapes "
                       "(profiles/r02_hbm_counter_calibration.json): TCC_EA0_RDREQ counts 128-byte requests, FETCH_SIZE prices "
                       "them at 64 B (factor 0.500 for every read pattern); WRITE_SIZE exact (factor 1.000)",
        "source": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes of `python bench.py --fanout-model {model} "
                  "--no-cpu-baseline --no-convergence --no-second-load --no-long-window --steps 20 --warmup 5` (tools/gpu/r4_measure.sh), mean over "
                  "the 20 timed launches",
        "algorithmic_bytes_per_launch_v0": 1176 * n, "traffic_over_algorithmic": (read + write) / (1176 * n),
    }
    json.dump(traffic, open(os.path.join(out, f"r04_pmc_traffic_{model}.json"), "w"), indent=1)
    print(model, json.dumps({k: traffic[k] for k in ("kernel_us_mean_profiled", "hbm_bytes_per_launch", "frac_of_8_tbps", "traffic_over_algorithmic", "commit", "kernel_source_sha16")}))
