#!/usr/bin/env python
"""A/B of builds of the product library on the benchmark workload (run on the GPU box).

usage: python tools/ab.py [--ticks 120] [--rounds 3] [--nodes N] lib1.so lib2.so ...

Every library runs bench.py's configuration and schedule: untimed pre-roll into the stationary load, then `ticks`
ticks with HIP events around every tick-kernel launch; the libraries alternate `rounds` times (box-to-box and
minute-to-minute drift is larger than most effects).  All builds implement the same SIMSPEC, so their state digests
after the run must agree — checked, a variant that computes something else is flagged.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401  (one HIP runtime per process: torch's first)

import bench  # noqa: E402
from serf_amd import _ffi  # noqa: E402


def run_one(path, args, ticks):
    lib = _ffi.SimLib(path, optional=("bind_exchange3", "resident_planes"))   # (a build of the previous ABI may be one of the candidates)
    n = args.nodes_per_gpu
    kw, ops = bench.workload(args, n)
    sim = _ffi.Sim(lib, _ffi.make_config(n, **kw))
    for o in ops:
        sim.inject(*o)
    sim.step(args.preroll)
    sim.sync()
    sim.profile(1)
    import time
    t0 = time.perf_counter()
    sim.step(ticks)
    sim.sync()
    step_us = (time.perf_counter() - t0) / ticks * 1e6   # the whole step: graph build, deep-queue kernel, launch gaps
    (tot, mn, mx), cnt = sim.profile_read_stats()
    sim.profile(0)
    dig = sim.digest()
    drops = sim.cluster_stats()["overflow"]
    sim.close()
    return tot / cnt * 1e3, mn * 1e3, mx * 1e3, dig, drops, step_us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ticks", type=int, default=120)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--nodes", type=int, default=1 << 20)
    ap.add_argument("--fanout-model", default="bijection", choices=["bijection", "krandomnodes"])
    ap.add_argument("--rate", type=float, default=None, help="API operations per tick (default: the bench's)")
    ap.add_argument("--pkt-records", type=int, default=None)
    ap.add_argument("--ring-overflow", type=int, default=None)
    ap.add_argument("libs", nargs="+")
    a = ap.parse_args()
    extra = []
    if a.rate is not None:
        extra += ["--rate", str(a.rate)]
    if a.pkt_records is not None:
        extra += ["--pkt-records", str(a.pkt_records)]
    if a.ring_overflow is not None:
        extra += ["--ring-overflow", str(a.ring_overflow)]
    args = bench.parse_args(["--nodes-per-gpu", str(a.nodes), "--fanout-model", a.fanout_model] + extra)
    res = {p: [] for p in a.libs}
    steps = {p: [] for p in a.libs}
    digs = {}
    for r in range(a.rounds):
        for p in a.libs:
            mean, mn, mx, dig, drops, step_us = run_one(os.path.join(ROOT, p) if not os.path.isabs(p) else p, args, a.ticks)
            res[p].append(mean)
            steps[p].append(step_us)
            digs[p] = dig
            print(f"round {r} {p}: kernel {mean:.1f} us/tick (min {mn:.1f} max {mx:.1f}), step {step_us:.1f} us, drops {drops}", flush=True)
    ref = digs[a.libs[0]]
    out = {"ticks": a.ticks, "nodes": a.nodes, "results": {}}
    for p in a.libs:
        v = sorted(res[p])
        sv = sorted(steps[p])
        out["results"][p] = {"us_per_tick_median": v[len(v) // 2], "us_per_tick_all": res[p], "step_us_median": sv[len(sv) // 2], "step_us_all": steps[p],
                             "same_digest_as_first": digs[p] == ref}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
