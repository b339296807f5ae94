#!/usr/bin/env python
"""Copy the summaries of tools/gpu/r6_measure.sh (gpurun_out/r6m) into profiles/ under their judged names and derive, for EACH
fan-out model and EACH window (the driver's 20 timed launches; the 300 launches of the long window, ticks 345 .. 644),
profiles/r06_pmc_traffic_<model>[_long].json — HBM bytes per tick-kernel launch with the calibration factors of
profiles/r02_hbm_counter_calibration.json — stamped with the commit and the hash of the kernel source it was measured on
(bench.py reports the figure as roofline.traffic / roofline.frac_measured only while that hash is the current one).

usage: python tools/collect_r6.py [call_dir [git-rev-that-was-measured]]"""
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
call = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r6m")
out = os.path.join(ROOT, "profiles")


def last_json(path):
    text = open(path).read().strip()
    try:
        return json.loads(text)
    except json.JSONDecodeError:
        return json.loads([ln for ln in text.splitlines() if ln.startswith("{")][-1])


for src, dst in (("bench_20_5.json", "r06_bench_driver_args.json"), ("bench_20_5_detail.json", "r06_bench_driver_args_detail.json"), ("bench_one_rank_rccl.json", "r06_bench_one_rank_rccl.json"),
                 ("bench_one_rank_rccl_krandomnodes.json", "r06_bench_one_rank_rccl_krandomnodes.json")):
    if os.path.exists(os.path.join(call, src)) and os.path.getsize(os.path.join(call, src)):
        json.dump(last_json(os.path.join(call, src)), open(os.path.join(out, dst), "w"), indent=1)
sys.path.insert(0, ROOT)
import bench  # noqa: E402
sha = bench.kernel_source_sha16()   # the device side of the tick kernel: serf_sim_state.inc + serf_sim_handlers.inc + serf_sim_tick.inc
if len(sys.argv) > 2:               # ... as it stood at the commit the call measured (the working tree has moved on since)
    import hashlib
    rev = sys.argv[2]
    sha = hashlib.sha256(b"".join(subprocess.run(["git", "-C", ROOT, "show", f"{rev}:{f}"], capture_output=True, check=True).stdout for f in bench.KERNEL_SOURCES)).hexdigest()[:16]
commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", sys.argv[2] if len(sys.argv) > 2 else "HEAD"], capture_output=True, text=True).stdout.strip()
n = 1 << 20
for model in ("krandomnodes", "bijection"):
    for window, steps, warmup, sfx in (("short", 20, 5, ""), ("long", 300, 25, "_long")):
        d = os.path.join(call, f"{model}_{window}")
        if not os.path.exists(os.path.join(d, "tick_kernel_pmc.json")):
            continue
        if os.path.getsize(os.path.join(d, "bench_traced.json")):
            json.dump(last_json(os.path.join(d, "bench_traced.json")), open(os.path.join(out, f"r06_bench_under_rocprof_{model}{sfx}.json"), "w"), indent=1)
        for f in (os.path.join(d, "trace", "t_kernel_stats.csv"),):
            if os.path.exists(f):
                shutil.copy(f, os.path.join(out, f"r06_kernel_stats_{model}{sfx}.csv"))
        pmc = json.load(open(os.path.join(d, "tick_kernel_pmc.json")))
        json.dump(pmc, open(os.path.join(out, f"r06_tick_kernel_pmc_{model}{sfx}.json"), "w"), indent=1)
        c = pmc["counters"]
        if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
            continue
        read, write = 2.0 * c["FETCH_SIZE"] * 1024, c["WRITE_SIZE"] * 1024   # both counters are in KiB
        traffic = {
            "kernel": "tick_kernel", "fanout_model": model, "window": window, "launches_averaged": pmc["launches"], "kernel_us_mean_profiled": pmc["kernel_us_mean"],
            "hbm_read_bytes": read, "hbm_write_bytes": write, "hbm_bytes_per_launch": read + write,
            "hbm_gbps_over_the_profiled_launches": (read + write) / pmc["kernel_us_mean"] / 1e3,
            "frac_of_8_tbps": (read + write) / pmc["kernel_us_mean"] / 1e3 / 8000.0,
            "fetch_size_raw_kib": c["FETCH_SIZE"], "write_size_raw_kib": c["WRITE_SIZE"],
            "commit": commit, "kernel_source_sha16": sha, "steps": steps, "warmup": warmup,
            "workload": f"bench.py defaults, fan-out model {model}: 1 Mi nodes, fan-out 4, 0.25 API ops/tick, 4 records per packet, the {steps} timed "
                        f"launches of --steps {steps} --warmup {warmup} (ticks {320 + warmup} .. {320 + warmup + steps - 1})",
            "calibration": "reads = 2 x FETCH_SIZE, writes = WRITE_SIZE: tools/calib on this kernel's access shapes "
                           "(profiles/r02_hbm_counter_calibration.json): TCC_EA0_RDREQ counts 128-byte requests, FETCH_SIZE prices "
                           "them at 64 B (factor 0.500 for every read pattern); WRITE_SIZE exact (factor 1.000)",
            "source": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes of `python bench.py --fanout-model {model} "
                      f"--no-cpu-baseline --no-convergence --no-second-load --no-long-window --steps {steps} --warmup {warmup}` (tools/gpu/r6_measure.sh), mean over "
                      f"the {steps} timed launches",
            "algorithmic_bytes_per_launch_v0": 1176 * n, "traffic_over_algorithmic": (read + write) / (1176 * n),
        }
        json.dump(traffic, open(os.path.join(out, f"r06_pmc_traffic_{model}{sfx}.json"), "w"), indent=1)
        print(model, window, json.dumps({k: traffic[k] for k in ("kernel_us_mean_profiled", "hbm_bytes_per_launch", "frac_of_8_tbps", "traffic_over_algorithmic", "commit", "kernel_source_sha16")}))

# the second load: its own traffic figure (bench.py's PMC_TRAFFIC_SECOND reads it)
d = os.path.join(call, "second_load")
if os.path.exists(os.path.join(d, "tick_kernel_pmc.json")):
    pmc = json.load(open(os.path.join(d, "tick_kernel_pmc.json")))
    json.dump(pmc, open(os.path.join(out, "r06_tick_kernel_pmc_second_load.json"), "w"), indent=1)
    if os.path.exists(os.path.join(d, "deep_kernel_pmc.json")):
        shutil.copy(os.path.join(d, "deep_kernel_pmc.json"), os.path.join(out, "r06_deep_kernel_pmc_second_load.json"))
    for f in (os.path.join(d, "trace", "t_kernel_stats.csv"),):
        if os.path.exists(f):
            shutil.copy(f, os.path.join(out, "r06_kernel_stats_second_load.csv"))
    c = pmc["counters"]
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        read, write = 2.0 * c["FETCH_SIZE"] * 1024, c["WRITE_SIZE"] * 1024
        bt = 2 * 64 + 2 * 16 * 16 + 2 * 4 * 16 * 16 + 4 * 6
        t = {"kernel": "tick_kernel (multi-page instantiation)", "fanout_model": "krandomnodes", "rate": 0.8, "pkt_records": 16, "ring_overflow": 8, "steps": 60, "warmup": 20, "preroll": 160,
             "launches_averaged": pmc["launches"], "kernel_us_mean_profiled": pmc["kernel_us_mean"], "hbm_read_bytes": read, "hbm_write_bytes": write,
             "hbm_bytes_per_launch": read + write, "frac_of_8_tbps": (read + write) / pmc["kernel_us_mean"] / 1e3 / 8000.0,
             "algorithmic_bytes_per_launch_P16": bt * n, "traffic_over_algorithmic": (read + write) / (bt * n), "commit": commit, "kernel_source_sha16": sha,
             "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes of `python bench.py --fanout-model krandomnodes --rate 0.8 --pkt-records 16 "
                       "--ring-overflow 8 --preroll 160 --warmup 20 --steps 60 --no-cpu-baseline --no-convergence --no-second-load --no-long-window` (tools/gpu/r6_measure.sh): "
                       "the cluster, schedule and ticks of bench.py's second_load; mean over the 60 timed launches"}
        json.dump(t, open(os.path.join(out, "r06_pmc_traffic_second_load.json"), "w"), indent=1)
        print("second_load", json.dumps({k: t[k] for k in ("kernel_us_mean_profiled", "hbm_bytes_per_launch", "frac_of_8_tbps", "traffic_over_algorithmic")}))
