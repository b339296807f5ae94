#!/usr/bin/env python
"""Where the step's time outside the tick kernel goes (run on the GPU box): wall-clock step time of the headline configuration over
`--ticks` ticks behind the pre-roll, under the library's environment switches, each variant in a process of its own.
usage: python tools/gap_probe.py [--ticks 200] [--variants name=ENV1=v,ENV2=v ...]"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(ticks, profile, lib):
    sys.path.insert(0, ROOT)
    import torch  # noqa: F401
    import bench
    from serf_amd import _ffi
    args = bench.parse_args(["--fanout-model", "krandomnodes"])
    L = _ffi.SimLib(lib)
    kw, ops = bench.workload(args, args.nodes_per_gpu, "krandomnodes")
    sim = _ffi.Sim(L, _ffi.make_config(args.nodes_per_gpu, **kw))
    for o in ops:
        sim.inject(*o)
    sim.step(args.preroll)
    sim.sync()
    out = {}
    for rep in range(3):
        sim.profile(profile)
        t0 = time.perf_counter()
        sim.step(ticks)
        sim.sync()
        dt = time.perf_counter() - t0
        (ms, mn, mx), cnt = sim.profile_read_stats()
        sim.profile(0)
        out.setdefault("step_us", []).append(round(dt / ticks * 1e6, 1))
        out.setdefault("kernel_us", []).append(round(ms / max(1, cnt) * 1e3, 1))
    print("RESULT " + json.dumps(out), flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--ticks", type=int, default=100)
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--profile", type=int, default=0)
    ap.add_argument("--lib", default=os.path.join(ROOT, "serf_amd", "csrc", "libserf_sim.so"))
    ap.add_argument("--variants", nargs="*", default=["base=", "prof1=PROFILE=1", "sync=SERF_RF_SYNC=1", "nolean=SERF_RF_LEAN=0"])
    a = ap.parse_args()
    if a.child:
        child(a.ticks, a.profile, a.lib)
        sys.exit(0)
    for v in a.variants:
        name, _, envs = v.partition("=")
        env = dict(os.environ)
        prof, lib = 0, a.lib
        for kv in [x for x in envs.split(",") if x]:
            k, _, val = kv.partition("=")
            if k == "PROFILE":
                prof = int(val)
            elif k == "LIB":
                lib = os.path.join(ROOT, val)
            else:
                env[k] = val
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--ticks", str(a.ticks), "--profile", str(prof), "--lib", lib],
                           env=env, capture_output=True, text=True)
        res = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")]
        print(name, res[-1][7:] if res else ("FAILED " + p.stderr[-300:]), flush=True)
