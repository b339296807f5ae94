"""Marginal cost of the parts of the tick kernel (measurement build: libserf_sim_ablate.so = -DTICK_ABLATE).

Runs the bench workload normally up to a tick, then times ONE launch with parts left out (the state
after that launch is garbage, so every measurement starts from a fresh run of the same schedule).
"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (one HIP runtime per process: torch's first)
import serf_amd
from serf_amd import _ffi
import bench

lib = _ffi.SimLib(os.path.join(os.path.dirname(serf_amd.LIB_PATH), "libserf_sim_ablate.so"))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
MASKS = [(0, "full tick"), (1, "phase 1 only (no queue phase)"), (2, "phase 2 only (no deliver/timers)"),
         (4, "no scatter stores"), (8, "no payload gathers"), (16, "no row/key stores"), (4 | 8 | 16, "no scatter/gather/row stores"),
         (32, "no handler loop"), (64, "records loaded, no lookups, no handlers"), (1 | 64, "row + record loads only"), (1 | 2, "row load/store only")]
ticks = (330, 333, 351)  # inside the benchmark's timed region (stationary load); probe phases differ per tick
args = bench.parse_args(["--nodes-per-gpu", str(n)] + [a for a in sys.argv[2:] if not a.startswith(("masks=", "ticks="))])  # e.g. --random-fanout
for a in sys.argv[2:]:
    if a.startswith("masks="):
        want = {int(x, 0) for x in a[6:].split(",")}
        MASKS = [m for m in MASKS if m[0] in want] + [(m, f"mask {m}") for m in sorted(want - {m[0] for m in MASKS})]
    if a.startswith("ticks="):   # (under rocprofv3 --pmc: one mask, one tick per process — the ablated launch is the last tick-kernel dispatch)
        ticks = tuple(int(x) for x in a[6:].split(","))
kw, ops = bench.workload(args, n)
res = {}
for mask, name in MASKS:
    tot = 0.0
    for t in ticks:
        sim = _ffi.Sim(lib, _ffi.make_config(n, **kw))
        for o in ops:
            sim.inject(*o)
        sim.step(t); sim.sync()
        lib.dll.sim_debug_ablate(C.c_uint(mask))
        sim.profile(True); sim.profile_read()
        sim.step(1); sim.sync()
        ms, k = sim.profile_read()
        lib.dll.sim_debug_ablate(C.c_uint(0))
        tot += ms / max(k, 1)
        del sim
    res[name] = tot / len(ticks) * 1e3
    print(f"{name:45s} {res[name]:8.1f} us", flush=True)
