"""Per-phase cycle breakdown of the tick kernel (needs libserf_sim_timing.so = -DTICK_TIMING build)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import serf_amd
from serf_amd import _ffi
from tests import _scenario as sc
lib = _ffi.SimLib(os.path.join(os.path.dirname(serf_amd.LIB_PATH), "libserf_sim_timing.so"))
swim = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
sim = _ffi.Sim(lib, _ffi.make_config(n, fanout=4, view_slots=1024, event_ring=512, query_ring=512, probe_interval=swim,
                                     push_pull_interval=150 if swim else 0, reap_interval=75 if swim else 0))
import bench
for t, op, node, a, b in sc.schedule(n, 200, rate=0.4, seed=3, mix=bench.MIX, max_member_subjects=512, even=True):
    sim.inject(t, op, node, a, b)
sim.step(100); sim.sync()
buf = (C.c_ulonglong * 16)()
lib.dll.sim_debug_timing(buf, 1)
sim.step(50); sim.sync()
lib.dll.sim_debug_timing(buf, 1)
names = ["row load", "cell r1-3 (x4)", "slot_of (x4)", "entry ptrs+issue (x4)", "entry heads wait (x4)", "handler loop (x4)", "timers+probe", "keys+pend inserts", "q_round x4", "payload gather (all)", "perm+store x4", "row/keys store"]
waves = n // 64 * 50
tot = sum(buf[:12])
for i, nm in enumerate(names):
    print(f"{nm:24s} {buf[i]/waves:10.0f} cyc/wave  {100*buf[i]/tot:5.1f}%")
print(f"{'total':24s} {tot/waves:10.0f} cyc/wave (clock ticks of s_memtime / readcyclecounter)")
