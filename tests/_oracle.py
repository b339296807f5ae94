"""Test-side loader for the CPU oracle (oracle/liboracle.so) and its handler-level hooks.

The oracle is test infrastructure: it is built by `make -C oracle` (or __graft_entry__.build())
and bound here with the same ctypes class the product uses, plus the `osim_t_*` hooks that let
the reference's known-answer tests be replayed handler by handler.
"""
import ctypes as C
import os
import subprocess

from serf_amd import _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liboracle.so")


def load_oracle():
    if not os.path.exists(ORACLE_SO):
        subprocess.check_call(["make", "-C", ORACLE_DIR])
    lib = _ffi.SimLib(ORACLE_SO, prefix="osim_")
    d, H, u32, u64 = lib.dll, C.c_void_p, C.c_uint32, C.c_uint64
    P32, P64 = C.POINTER(u32), C.POINTER(u64)
    hooks = {
        "clock_get": (C.c_int, [H, u32, u32, P64]),
        "clock_set": (C.c_int, [H, u32, u32, u64]),
        "clock_witness": (C.c_int, [H, u32, u32, u64]),
        "clock_increment": (C.c_int, [H, u32, u32, P64]),
        "set_member": (C.c_int, [H, u32, u32, u32, u64, u32]),
        "set_tick": (C.c_int, [H, u64]),
        "set_serf_state": (C.c_int, [H, u32, u32]),
        "set_min_time": (C.c_int, [H, u32, u32, u64]),
        "recent_intent": (C.c_int, [H, u32, u32, u32, P64]),
        "upsert_intent": (C.c_int, [H, u32, u32, u32, u64]),
        "join_intent": (C.c_int, [H, u32, u32, u64]),
        "leave_intent": (C.c_int, [H, u32, u32, u64, C.c_int]),
        "user_event": (C.c_int, [H, u32, u32, u64]),
        "query": (C.c_int, [H, u32, u32, u64, u32]),
        "notify_join": (C.c_int, [H, u32, u32]),
        "notify_leave": (C.c_int, [H, u32, u32]),
        "reap": (C.c_int, [H, u32, u64, u64, u64, u64]),
        "queue_max": (u64, [u64, u64, u64]),
        "merge_remote_state": (C.c_int, [H, u32, u64, u64, u64, P32, P64, u32, P32, u32, P64, P32, u32, C.c_int, C.c_int]),
        "targets": (C.c_int, [H, u64, u32, P32]),
        "swim_alive": (C.c_int, [H, u32, u32, u32]),
        "swim_suspect": (C.c_int, [H, u32, u32, u32, u32]),
        "swim_dead": (C.c_int, [H, u32, u32, u32, u32]),
        "swim_timers": (C.c_int, [H, u32]),
        "swim_params": (C.c_int, [H, P32, P32]),
        "view_get": (C.c_int, [H, u32, u32, C.c_void_p]),
    }
    lib.t = {}
    for name, (res, args) in hooks.items():
        fn = getattr(d, "osim_t_" + name)
        fn.restype, fn.argtypes = res, args
        lib.t[name] = fn
    return lib


class Node:
    """Handler-level view of one simulated node of an oracle Sim (mirrors a `Serf` instance
    in the reference's white-box tests)."""

    CLOCK, EVENT, QUERY = 0, 1, 2
    JOIN, LEAVE = 1, 2

    def __init__(self, lib, sim, node=0):
        self.lib, self.sim, self.node, self.t = lib, sim, node, lib.t

    def clock(self, which=0):
        v = C.c_uint64()
        assert self.t["clock_get"](self.sim.h, self.node, which, C.byref(v)) == 0
        return v.value

    def witness(self, which, t):
        assert self.t["clock_witness"](self.sim.h, self.node, which, t) == 0

    def increment(self, which=0):
        v = C.c_uint64()
        assert self.t["clock_increment"](self.sim.h, self.node, which, C.byref(v)) == 0
        return v.value

    def set_clock(self, which, t):
        assert self.t["clock_set"](self.sim.h, self.node, which, t) == 0

    def set_member(self, subject, status, ltime, stamp=0):
        assert self.t["set_member"](self.sim.h, self.node, subject, status, ltime, stamp) == 0

    def recent_intent(self, subject, ty):
        v = C.c_uint64()
        found = self.t["recent_intent"](self.sim.h, self.node, subject, ty, C.byref(v))
        return v.value if found else None

    def upsert_intent(self, subject, ty, ltime):
        return bool(self.t["upsert_intent"](self.sim.h, self.node, subject, ty, ltime))

    def join_intent(self, subject, ltime):
        return bool(self.t["join_intent"](self.sim.h, self.node, subject, ltime))

    def leave_intent(self, subject, ltime, prune=False):
        return bool(self.t["leave_intent"](self.sim.h, self.node, subject, ltime, int(prune)))

    def user_event(self, key, ltime):
        return bool(self.t["user_event"](self.sim.h, self.node, key, ltime))

    def query(self, qid, ltime, flags=0):
        return bool(self.t["query"](self.sim.h, self.node, qid, ltime, flags))

    def notify_join(self, subject):
        assert self.t["notify_join"](self.sim.h, self.node, subject) == 0

    def notify_leave(self, subject):
        assert self.t["notify_leave"](self.sim.h, self.node, subject) == 0

    def reap(self, now, reconnect_timeout, tombstone_timeout, intent_timeout):
        assert self.t["reap"](self.sim.h, self.node, now, reconnect_timeout, tombstone_timeout, intent_timeout) == 0

    # ---- memberlist layer ----
    def alive(self, subject, inc):
        assert self.t["swim_alive"](self.sim.h, self.node, subject, inc) == 0

    def suspect(self, subject, inc, frm):
        assert self.t["swim_suspect"](self.sim.h, self.node, subject, inc, frm) == 0

    def dead(self, subject, inc, frm):
        assert self.t["swim_dead"](self.sim.h, self.node, subject, inc, frm) == 0

    def run_timers(self):
        assert self.t["swim_timers"](self.sim.h, self.node) == 0

    def view(self, subject):
        """(swim state, incarnation, nconf, MemberStatus, known) of `subject` as seen by this node."""
        import numpy as np
        buf = np.zeros(1, _ffi.VIEW_DTYPE)
        assert self.t["view_get"](self.sim.h, self.node, subject, buf.ctypes.data) == 0
        b = int(buf["bits"][0])
        return {"swim": (b >> 4) & 3, "inc": int(buf["inc"][0]), "nconf": (b >> 8) & 7,
                "status": (b >> 1) & 7, "known": b & 1, "stamp": b >> 11,
                "ltime": int(buf["ltime"][0]), "conf": [int(x) for x in buf["conf"][0]]}

    def queue_kinds(self):
        q = self.sim.dump(_ffi.ARR_QUEUE).reshape(-1, _ffi.Q)[self.node]
        return [int((m >> 4) & 15) for m in q["meta"] if m != 0xFFFFFFFF]

    def member(self, subject):
        st, lt = self.sim.members(self.node)
        return int(st[subject]), int(lt[subject])
