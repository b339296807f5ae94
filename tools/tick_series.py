"""Tick-by-tick series of the headline kernel over the benchmark's two timed windows: for every tick of
[first, first + count) the tick kernel's duration (HIP events on the dispatch, product library) and — from the -DTICK_TIMING
build rolled through the same schedule — the wave-averaged handler iterations, classification rounds, records left for the
handlers and records fetched again, next to what the schedule did in that tick.  Writes one JSON document.

usage: python tools/tick_series.py OUT.json [--first 325] [--count 320] [bench.py flags, e.g. --fanout-model krandomnodes]"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402
import bench  # noqa: E402
import serf_amd  # noqa: E402
from serf_amd import _ffi  # noqa: E402

out_path = sys.argv[1]
argv = sys.argv[2:]
first, count = 325, 320
for flag in ("--first", "--count"):
    if flag in argv:
        i = argv.index(flag)
        v = int(argv[i + 1])
        del argv[i:i + 2]
        if flag == "--first":
            first = v
        else:
            count = v
if "--fanout-model" not in argv and "--random-fanout" not in argv:
    argv += ["--fanout-model", "krandomnodes"]
args = bench.parse_args(argv)
n = args.nodes_per_gpu
kw, ops = bench.workload(args, n)
by_tick = {}
for t, op, node, a, b in ops:
    by_tick.setdefault(t, []).append(int(op))


def cluster(lib):
    sim = _ffi.Sim(lib, _ffi.make_config(n, **kw))
    for o in ops:
        sim.inject(*o)
    sim.step(first)
    sim.sync()
    return sim


doc = {"nodes": n, "fanout_model": args.fanout_model, "first_tick": first, "ticks": count, "kernel_source_sha16": bench.kernel_source_sha16(),
       "what": "per tick: kernel_ms = the tick kernel's own duration (start/stop events on the dispatch); iters / rounds / slow / refetched = "
               "wave averages from the -DTICK_TIMING build on the same schedule (handler-loop iterations, classification rounds that looked "
               "anything up, records left for the handlers, of those fetched again); ops = operation codes the schedule applies in that tick"}
# pass 1: the product library, one timed launch per tick
sim = cluster(serf_amd.load() if not os.environ.get("SERIES_LIB") else _ffi.SimLib(os.environ["SERIES_LIB"]))
sim.profile(1)
kms = []
load = []
for i in range(count):
    sim.step(1)
    ms, cnt = sim.profile_read()
    kms.append(ms / max(1, cnt))
    if i % 20 == 0:
        cs = sim.cluster_stats()
        load.append({"tick": first + i, "records_per_packet": round(cs["inbox_records"] / (args.fanout * n), 4),
                     "queued_per_node": round(sum(cs["queued"]) / n, 4), "deepest_queue": int(cs["max_queue"]), "failed": int(cs["failed"]), "left": int(cs["left"]),
                     "slots_in_use": int(cs["slots_in_use"]), "drops": int(cs["overflow"])})
sim.profile(0)
sim.close()
# pass 2: the instrumented build, counters read and reset every tick
tlib_path = os.path.join(os.path.dirname(serf_amd.LIB_PATH), "libserf_sim_timing.so")
series = []
if os.path.exists(tlib_path):
    tlib = _ffi.SimLib(tlib_path)
    sim = cluster(tlib)
    buf = (C.c_ulonglong * 32)()
    tlib.dll.sim_debug_timing(buf, 1)
    waves = n // 64
    for i in range(count):
        sim.step(1)
        sim.sync()
        tlib.dll.sim_debug_timing(buf, 1)
        series.append({"iters": buf[12] / waves, "rounds": buf[13] / waves, "slow": buf[14] / waves, "refetched": buf[15] / waves,
                       "slow_by_kind": [round(buf[17 + i] / waves, 2) for i in range(7)],  # JOIN LEAVE EVENT QUERY ALIVE SUSPECT DEAD
                       "cyc_handlers": buf[5] / waves, "cyc_total": sum(buf[:12]) / waves})
    sim.close()
rows = []
for i in range(count):
    r = {"tick": first + i, "kernel_ms": round(kms[i], 5), "ops": by_tick.get(first + i, [])}
    if series:
        r.update({k: (round(v, 3) if not isinstance(v, list) else v) for k, v in series[i].items()})
    rows.append(r)
doc["series"] = rows
doc["load_every_20_ticks"] = load
k = sorted(kms)
doc["summary"] = {"kernel_ms_mean": sum(kms) / len(kms), "kernel_ms_median": k[len(k) // 2], "kernel_ms_min": k[0], "kernel_ms_max": k[-1],
                  "mean_first_20": sum(kms[:20]) / 20, "mean_rest": sum(kms[20:]) / max(1, len(kms) - 20)}
os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
json.dump(doc, open(out_path, "w"), indent=1)
print(json.dumps(doc["summary"]))
