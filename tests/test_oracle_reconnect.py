"""Reconnector (serf-core/src/serf/base.rs:612-681; SIMSPEC §2.9): a node with failed members attempts — with probability
failed / alive per reconnect interval — a memberlist.join, i.e. a push-pull, with one of them.  CPU oracle."""
import numpy as np
import pytest

from serf_amd import _ffi
from tests._oracle import load_oracle


@pytest.fixture(scope="module")
def oracle():
    return load_oracle()


KW = dict(fanout=3, view_slots=32, event_ring=16, query_ring=8, probe_interval=2, suspicion_mult=4, suspicion_max_mult=2)


def _failed_then_back(oracle, n, reconnect_interval, ticks, **extra):
    """Node 10 goes down at tick 3, is declared failed, and its process resumes at tick 32 with its old state while nobody
    gossips to it any more (gossip_to_the_dead = 1): test_oracle_swim.py::test_gossip_to_the_dead_time's second half."""
    sim = _ffi.Sim(oracle, _ffi.make_config(n, gossip_to_the_dead=1, reconnect_interval=reconnect_interval, **dict(KW, **extra)))
    sim.inject(3, _ffi.OP_CRASH, 10)
    sim.inject(32, _ffi.OP_REVIVE, 10)
    sim.step(ticks)
    return sim


def test_reconnector_brings_a_resumed_node_back(oracle):
    """Without the Reconnector the resumed node stays failed for ever (nobody tells it; it has nothing to refute).  With it
    one of the 511 others attempts a join within an interval or two (every node draws with probability 1 / 511 per
    interval): the push-pull shows node 10 its own obituary, it refutes with the next incarnation, and the refutation
    travels like any alive message: everybody has it back."""
    n = 512
    sim = _failed_then_back(oracle, n, 0, 200)
    st, _ = sim.members(200)
    assert int(st[10]) == _ffi.STATUS_FAILED
    sim = _failed_then_back(oracle, n, 8, 200)
    rows = sim.dump(_ffi.ARR_ROWS)
    assert int(rows["inc"][10]) >= 1, "node 10 refuted"
    for obs in (0, 200, 511):
        st, _ = sim.members(obs)
        assert int(st[10]) == _ffi.STATUS_ALIVE, obs
    assert int(rows["n_failed"].max()) == 0
    assert sim.cluster_stats()["overflow"] == 0


def test_attempt_rate_is_one_per_failed_member_and_interval(oracle):
    """base.rs:643-648: "we probabilistically expect the cluster to attempt to connect to each failed member once per
    reconnect interval".  Eight nodes stay down; the attempts (the tagged entries of the request list) are counted over
    40 intervals: 8 per interval expected, every one from a running node to one of the eight."""
    n, RI = 1024, 4
    down = list(range(100, 900, 100))
    sim = _ffi.Sim(oracle, _ffi.make_config(n, reconnect_interval=RI, **KW))
    for x in down:
        sim.inject(2, _ffi.OP_CRASH, x)
    sim.step(60)                                   # everybody has declared the eight failed
    assert int(sim.dump(_ffi.ARR_ROWS)["n_failed"][0]) == 8
    import ctypes as C
    attempts, buf = [], (C.c_uint32 * 512)()
    oracle.dll.osim_t_scheduled.restype = C.c_uint32
    for _ in range(40 * RI):
        sim.step(1)   # the attempts drawn two ticks ago are on the schedule of the tick that runs next
        k = oracle.dll.osim_t_scheduled(sim.h, C.c_uint32(_ffi.OP_RECONNECT), C.c_uint64(sim.tick), buf, C.c_uint32(256))
        attempts += [(buf[2 * i], buf[2 * i + 1]) for i in range(k)]
    assert all(b in down and a not in down for a, b in attempts)
    assert 0.7 * 8 * 40 < len(attempts) < 1.3 * 8 * 40, len(attempts)
    hit = np.bincount([down.index(b) for _, b in attempts], minlength=8)
    assert hit.min() > 15, hit                      # the target is drawn uniformly among the failed members


def test_attempts_of_one_tick_are_disjoint_pairs_and_the_rest_waits(oracle):
    """Two attempts that name the same target in one tick: the first runs, the second is put back on the schedule for the
    next tick (the pairs of a tick run side by side); an attempt on a process that is down is forgotten."""
    n = 512
    sim = _failed_then_back(oracle, n, 0, 60)       # node 10 resumed at 32, failed in every view, nobody talks to it
    t = sim.tick
    sim.inject(t, _ffi.OP_RECONNECT, 5, 10)
    sim.inject(t, _ffi.OP_RECONNECT, 6, 10)         # shares node 10 with the first: next tick
    sim.inject(t, _ffi.OP_RECONNECT, 7, 7)          # not a pair
    sim.step(1)
    rows = sim.dump(_ffi.ARR_ROWS)
    assert int(rows["inc"][10]) == 1                # the push-pull with node 5 made node 10 refute
    view5, _ = sim.members(5)
    view6, _ = sim.members(6)
    assert int(view5[10]) == _ffi.STATUS_FAILED     # the initiator merged first — node 10's state from BEFORE its refutation (both
    assert int(view6[10]) == _ffi.STATUS_FAILED     # sides of a join ship what they had); node 6's attempt has not run yet
    sim.step(1)
    view6, _ = sim.members(6)
    assert int(view6[10]) == _ffi.STATUS_ALIVE      # node 6's own push-pull, one tick later, reads the refuted incarnation
    sim.step(40)
    for obs in (5, 300):
        st, _ = sim.members(obs)
        assert int(st[10]) == _ffi.STATUS_ALIVE     # and the refutation reaches everybody by gossip
