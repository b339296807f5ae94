import os, sys, time
sys.path.insert(0, '.')
print("nproc", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
try: print("cpu.max", open('/sys/fs/cgroup/cpu.max').read().strip())
except Exception as e: print("cpu.max n/a", e)
from serf_amd import _ffi
import serf_amd
from tests._oracle import load_oracle
from tests import _scenario as sc
def T(label, f, n=1):
    t=time.time()
    for _ in range(n): r=f()
    print(f"{label}: {(time.time()-t)/n*1e3:.2f} ms"); return r
o = load_oracle(); g = serf_amd.load()
kw = dict(fanout=3, view_slots=0, event_ring=16, query_ring=8)
so = T("oracle create", lambda: _ffi.Sim(o, _ffi.make_config(128, **kw)))
sg = T("gpu create", lambda: _ffi.Sim(g, _ffi.make_config(128, **kw)))
T("oracle step", lambda: so.step(1), 20)
T("gpu step", lambda: (sg.step(1), sg.sync()), 20)
T("oracle digest", lambda: so.digest(), 20)
T("gpu digest", lambda: sg.digest(), 20)
T("oracle dump view", lambda: so.dump(_ffi.ARR_VIEW), 5)
T("gpu dump view", lambda: sg.dump(_ffi.ARR_VIEW), 5)
T("gpu dump rows", lambda: sg.dump(_ffi.ARR_ROWS), 5)
T("gpu dump queue", lambda: sg.dump(_ffi.ARR_QUEUE), 5)
T("inject", lambda: (sg.user_event(1, 5, 32)), 5)
T("gpu step w/ ops", lambda: (sg.step(1), sg.sync()), 1)
