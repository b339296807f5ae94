"""Reference-format per-node snapshot (serf_amd/snapshot.py) against the reference's own snapshotter
tests, restated (serf-core/src/serf/base/tests/serf/snapshot.rs), and over a simulated node's event log."""
import struct

import pytest

from serf_amd import _ffi, snapshot as snap


def kat_stream(rejoin):
    # snapshot.rs tests :6-212 / :280-388 / :389-493: user event @42, query @50, clock witnessed to 100, then
    # Join(foo), Failed(foo), Join(foo)
    s = snap.Snapshotter(0, rejoin_after_leave=rejoin)
    s.user_event(42)
    s.query(50)
    clock_time = 101            # LamportClock::witness(100) => time() == 101
    s.member_event(snap.EV_JOIN, 7, clock_time)
    s.member_event(snap.EV_FAILED, 7, clock_time)
    s.member_event(snap.EV_JOIN, 7, clock_time)
    return s


def test_snapshoter_kat():
    s = kat_stream(False)
    r = snap.replay(s.bytes())
    assert (r.last_clock, r.last_event_clock, r.last_query_clock) == (100, 42, 50)
    assert r.alive_nodes == {7}
    # record framing: type byte, u32-LE node length / u64-LE clock
    b = s.bytes()
    assert b[0] == snap.EVENT_CLOCK and struct.unpack_from("<Q", b, 1)[0] == 42
    assert b[9] == snap.QUERY_CLOCK and struct.unpack_from("<Q", b, 10)[0] == 50
    assert b[18] == snap.ALIVE and struct.unpack_from("<I", b, 19)[0] == len(b[23:23 + struct.unpack_from("<I", b, 19)[0]])


def test_snapshoter_force_compact():
    # :213-279: 1024 user events and 1024 queries with times 0..1023 => the last ones survive (time 0 is never newer
    # than the initial 0 and leaves no record)
    s = snap.Snapshotter(0)
    for i in range(1024):
        s.user_event(i)
    for i in range(1024):
        s.query(i)
    full = s.bytes()
    assert len(full) == 2 * 1023 * 9
    r = snap.replay(s.compact())
    assert (r.last_event_clock, r.last_query_clock) == (1023, 1023)
    assert len(s.bytes()) == 3 * 9 and snap.replay(full).last_event_clock == 1023


def test_snapshoter_leave():
    # :280-388: after leave() the replay is empty and the clocks are zero
    s = kat_stream(False)
    s.leave()
    s.user_event(99)  # "stop recording events after a leave is issued": leaves no record
    assert s.bytes()[-1] == snap.LEAVE and s.alive == set()
    assert s.compact() == snap._clock_record(snap.CLOCK, s.last_clock) + snap._clock_record(snap.EVENT_CLOCK, s.last_event_clock) + \
        snap._clock_record(snap.QUERY_CLOCK, s.last_query_clock)  # compaction after a leave re-emits no live node
    s = kat_stream(False)
    s.leave()
    r = snap.replay(s.bytes())
    assert r.alive_nodes == set() and (r.last_clock, r.last_event_clock, r.last_query_clock) == (0, 0, 0)


def test_snapshoter_leave_rejoin():
    # :389-493: with rejoin_after_leave the Leave record is still written (handle_leave, snapshot.rs:562-580, appends it
    # unconditionally) but the state is kept, and a replay that plans to rejoin skips the record
    s = kat_stream(True)
    s.leave()
    assert s.bytes()[-1] == snap.LEAVE and s.alive == {7}
    # the same file replayed by a process that does NOT plan to come back: the record clears everything
    r0 = snap.replay(s.bytes(), rejoin_after_leave=False)
    assert r0.alive_nodes == set() and (r0.last_clock, r0.last_event_clock, r0.last_query_clock) == (0, 0, 0)
    r = snap.replay(s.bytes(), rejoin_after_leave=True)
    assert (r.last_clock, r.last_event_clock, r.last_query_clock) == (100, 42, 50) and r.alive_nodes == {7}
    # and a Leave record written by a process that did NOT plan to come back is ignored by one that does
    t = kat_stream(False)
    t.leave()
    assert snap.replay(t.bytes(), rejoin_after_leave=True).alive_nodes == {7}


def test_unknown_record_type_is_an_error():
    with pytest.raises(ValueError):
        snap.replay(bytes([9]))
    with pytest.raises(ValueError):
        snap.replay(bytes([snap.CLOCK, 1, 2]))


def test_snapshot_of_a_simulated_node(oracle):
    # a watched node sees a crash (Failed), a graceful leave (Leave), a rejoin (Join), user events and queries; the
    # file its snapshotter would have written replays to what the node itself reports
    n = 128
    sim = _ffi.Sim(oracle, _ffi.make_config(n, fanout=3, view_slots=0, probe_interval=2, suspicion_mult=3, suspicion_max_mult=2, leave_delay=4,
                                            flags=_ffi.CF_BASELINE_JOINED))
    obs = 5
    sim.watch(obs)
    sim.inject(2, _ffi.OP_CRASH, 40)
    sim.inject(3, _ffi.OP_USER_EVENT, 9, 0xAB, 40)
    sim.inject(5, _ffi.OP_QUERY, 11, 77, _ffi.F_ACK)
    sim.inject(6, _ffi.OP_LEAVE, 60)
    sim.inject(11, _ffi.OP_LEAVE_FINISH, 60)
    sim.inject(16, _ffi.OP_CRASH, 60)
    sim.inject(60, _ffi.OP_JOIN, 60)
    sim.step(160)
    events = [e for e in sim.drain_events() if e[1] == obs]
    kinds = [e[2] for e in events]
    assert snap.EV_FAILED in kinds and snap.EV_LEAVE in kinds and snap.EV_JOIN in kinds and snap.EV_USER in kinds and snap.EV_QUERY in kinds
    s = snap.snapshot_of(sim, obs, events)
    r = snap.replay(s.bytes())
    st = sim.stats(obs)
    assert r.last_clock == st.member_time - 1
    assert r.last_event_clock == max(e[4] for e in events if e[2] == snap.EV_USER)
    assert r.last_query_clock == max(e[4] for e in events if e[2] == snap.EV_QUERY)
    # alive set: everybody the node saw join and not leave / fail afterwards — 60 came back, 40 did not
    assert 60 in r.alive_nodes and 40 not in r.alive_nodes
    status, _ = sim.members(obs)
    assert status[60] == _ffi.STATUS_ALIVE and status[40] == _ffi.STATUS_FAILED
    # compaction keeps the replay result
    c = snap.replay(s.compact())
    assert (c.alive_nodes, c.last_clock, c.last_event_clock, c.last_query_clock) == (r.alive_nodes, r.last_clock, r.last_event_clock, r.last_query_clock)
    # a restarted snapshotter continues where the file ends (Snapshot::from_replay_result)
    t = snap.Snapshotter(obs, replay=c)
    t.user_event(r.last_event_clock)      # not newer: no record
    assert t.bytes() == b""
