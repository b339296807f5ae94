#!/usr/bin/env python
"""Tick-by-tick STREAM time of the headline configuration (run on the GPU box): one event behind every tick's launches, the
differences between consecutive events, the slow ticks next to what the schedule / the library did in them.
usage: python tools/step_series.py [--ticks 300] [--null-stream] [bench.py flags]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
import serf_amd  # noqa: E402
from serf_amd import _ffi  # noqa: E402

argv = sys.argv[1:]
ticks, null_stream = 300, False
if "--ticks" in argv:
    i = argv.index("--ticks"); ticks = int(argv[i + 1]); del argv[i:i + 2]
if "--null-stream" in argv:
    argv.remove("--null-stream"); null_stream = True
if "--fanout-model" not in argv:
    argv += ["--fanout-model", "krandomnodes"]
args = bench.parse_args(argv)
n = args.nodes_per_gpu
kw, ops = bench.workload(args, n)
by_tick = {}
for t, op, node, a, b in ops:
    by_tick.setdefault(t, []).append(int(op))
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
sim = _ffi.Sim(serf_amd.load(), _ffi.make_config(n, **kw))
stream = torch.cuda.current_stream(dev)
if not null_stream:
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    sim.set_stream(stream.cuda_stream)
for o in ops:
    sim.inject(*o)
sim.step(args.preroll)
sim.sync()
evs = [torch.cuda.Event(enable_timing=True) for _ in range(ticks + 1)]
evs[0].record(stream)
first = args.preroll
for i in range(ticks):
    sim.step(1)
    evs[i + 1].record(stream)
sim.sync()
torch.cuda.synchronize()
dt = [evs[i].elapsed_time(evs[i + 1]) * 1e3 for i in range(ticks)]
srt = sorted(dt)
print(json.dumps({"null_stream": null_stream, "ticks": [first, first + ticks - 1], "mean_us": sum(dt) / ticks, "median_us": srt[ticks // 2], "p90_us": srt[int(ticks * 0.9)], "max_us": srt[-1],
                  "per_100": [round(sum(dt[j:j + 100]) / len(dt[j:j + 100]), 1) for j in range(0, ticks, 100)]}))
slow = [(first + i, round(dt[i], 1), by_tick.get(first + i, [])) for i in range(ticks) if dt[i] > 1.5 * srt[ticks // 2]]
print("slow ticks (> 1.5 x median):", slow[:60])
