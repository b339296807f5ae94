"""Per-phase cycle breakdown of the tick kernel (needs libserf_sim_timing.so = -DTICK_TIMING build) on the benchmark
workload (bench.workload): pre-roll into the stationary load, then 50 instrumented ticks."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
import bench
import serf_amd
from serf_amd import _ffi
lib = _ffi.SimLib(os.path.join(os.path.dirname(serf_amd.LIB_PATH), "libserf_sim_timing.so"))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
args = bench.parse_args(["--nodes-per-gpu", str(n)] + sys.argv[2:])  # e.g. --random-fanout
kw, ops = bench.workload(args, n)
sim = _ffi.Sim(lib, _ffi.make_config(n, **kw))
for o in ops:
    sim.inject(*o)
sim.step(args.preroll); sim.sync()
buf = (C.c_ulonglong * 32)()
lib.dll.sim_debug_timing(buf, 1)
sim.step(50); sim.sync()
lib.dll.sim_debug_timing(buf, 1)
names = ["row load", "cell r1-3 (x4)", "slot_of (x4)", "entry ptrs+issue (x4)", "entry heads wait (x4)", "handler loop (x4)", "timers+probe+reaper",
         "keys+pend inserts", "q_round x4", "payload gather (all)", "perm+store x4", "row/keys store"]
waves = n // 64 * 50
tot = sum(buf[:12])
for i, nm in enumerate(names):
    print(f"{nm:44s} {buf[i]/waves:10.0f} cyc/wave  {100*buf[i]/tot:5.1f}%")
print(f"{'total':44s} {tot/waves:10.0f} cyc/wave (clock ticks of s_memtime / readcyclecounter)")
rf = "--random-fanout" in sys.argv
if rf:
    names[1:6] = ["RF: owner table + node state to LDS", "RF: balanced rounds (cell, slot map, heads, classify, stash)", "RF: every node over its packets' notes", "-", "RF: handlers, one record per lane and iteration"]
    print("(random fan-out: rows 2 - 6 are the balanced classification and the handler loop)")
    for i, nm in enumerate(names[:6]):
        print(f"{nm:60s} {buf[i]/waves:10.0f} cyc/wave  {100*buf[i]/tot:5.1f}%")
    for i, nm in ((13, "balanced rounds with records, per wave and tick"), (12, "handler iterations per wave and tick"),
                  (14, "records left for the handlers, per wave and tick"), (15, "of those, fetched again (not stashed)")):
        print(f"{nm:60s} {buf[i]/waves:8.2f}")
    sys.exit(0)
for i, nm in ((13, "pages with records, per wave and tick"), (15, "... whose handler loop ran"), (12, "handler-loop iterations per wave and tick"),
              (14, "lanes with a slow record, per wave and tick (sum over pages)")):
    print(f"{nm:60s} {buf[i]/waves:8.2f}")
