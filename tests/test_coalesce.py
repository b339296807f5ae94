"""The reference's coalescer tests (serf-core/src/coalesce/member.rs:147-330, coalesce/user.rs:118-255)
restated on the tick-based host-side coalescers, plus one run on a simulated cluster's event stream."""
from serf_amd import _ffi
from serf_amd.coalesce import (FAILED, JOIN, LEAVE, QUERY, REAP, UPDATE, USER, MemberEventCoalescer,
                               UserEventCoalescer, coalesce_loop)

FOO, BAR, ZIP, DEAD = 1, 2, 3, 4


def test_member_event_coalesce_basic():
    # member.rs:147-269: Join(foo), Leave(foo), Leave(bar), Update(zip) x2, Reap(dead) in one window
    send = [(0, 9, JOIN, FOO, 0), (0, 9, LEAVE, FOO, 0), (1, 9, LEAVE, BAR, 0), (1, 9, UPDATE, ZIP, 0),
            (2, 9, UPDATE, ZIP, 0), (2, 9, REAP, DEAD, 0)]
    out = coalesce_loop(send, MemberEventCoalescer(), coalesce_period=20, quiescent_period=20)
    by_type = {ty: sorted(members) for _, _, ty, members in out}
    assert len(out) == 3
    assert by_type == {LEAVE: [FOO, BAR], UPDATE: [ZIP], REAP: [DEAD]}
    assert all(t == 20 and obs == 9 for t, obs, _, _ in out)          # flushed when the quantum ran out


def test_member_event_coalesce_repeats_and_updates():
    # member.rs:271-330: the same type twice in consecutive windows is reported once, except Update
    c = MemberEventCoalescer()
    first = coalesce_loop([(0, 1, UPDATE, FOO, 0)], c, 5, 5)
    again = coalesce_loop([(30, 1, UPDATE, FOO, 0)], c, 5, 5)
    assert [e[2:] for e in first] == [(UPDATE, [FOO])] and [e[2:] for e in again] == [(UPDATE, [FOO])]
    c = MemberEventCoalescer()
    assert len(coalesce_loop([(0, 1, FAILED, FOO, 0)], c, 5, 5)) == 1
    assert coalesce_loop([(30, 1, FAILED, FOO, 0)], c, 5, 5) == []      # nothing new to say
    assert [e[2] for e in coalesce_loop([(60, 1, JOIN, FOO, 0)], c, 5, 5)] == [JOIN]


def test_member_event_coalesce_pass_through():
    # member.rs "pass through": only member events are handled; everything else is forwarded at once
    c = MemberEventCoalescer()
    assert [c.handle((0, 0, ty, 1, 0)) for ty in (USER, QUERY, JOIN, LEAVE, FAILED, UPDATE, REAP)] == [False, False, True, True, True, True, True]
    out = coalesce_loop([(0, 1, USER, 77, 5), (0, 1, JOIN, FOO, 0)], c, 10, 10)
    assert out[0] == (0, 1, USER, 77, 5) and out[1][2:] == (JOIN, [FOO])


def test_user_event_coalesce_basic():
    # user.rs:118-196: foo@1, foo@2, bar@2 "test1", bar@2 "test2" => foo@2 and both bars
    name = {0x100: "foo", 0x101: "foo", 0x200: "bar", 0x201: "bar"}      # key = name << 8 | payload variant
    send = [(0, 4, USER, 0x100, 1), (0, 4, USER, 0x101, 2), (1, 4, USER, 0x200, 2), (1, 4, USER, 0x201, 2)]
    out = coalesce_loop(send, UserEventCoalescer(), 20, 20)
    assert sorted((name[e[3]], e[3], e[4]) for e in out) == [("bar", 0x200, 2), ("bar", 0x201, 2), ("foo", 0x101, 2)]


def test_user_event_coalesce_pass_through():
    # user.rs:198-255: only user events that asked for coalescing are handled
    c = UserEventCoalescer(is_cc=lambda ev: ev[3] & 1 == 1)
    cases = [((0, 0, USER, 2, 1), False), ((0, 0, USER, 3, 1), True), ((0, 0, JOIN, 1, 0), False),
             ((0, 0, LEAVE, 1, 0), False), ((0, 0, FAILED, 1, 0), False)]
    assert [c.handle(ev) for ev, _ in cases] == [want for _, want in cases]


def test_quiescence_flushes_before_the_quantum():
    c = MemberEventCoalescer()
    out = coalesce_loop([(0, 1, JOIN, FOO, 0), (2, 1, JOIN, BAR, 0), (50, 1, LEAVE, FOO, 0)], c, coalesce_period=20, quiescent_period=5)
    assert [(t, ty, m) for t, _, ty, m in out] == [(7, JOIN, [FOO, BAR]), (55, LEAVE, [FOO])]


def test_coalescing_a_simulated_clusters_member_events(oracle):
    # a crash and a graceful leave seen by one observer: raw stream Failed / Leave, coalesced per window
    sim = _ffi.Sim(oracle, _ffi.make_config(64, fanout=3, probe_interval=5, leave_delay=8))
    sim.watch(3)
    sim.inject(1, _ffi.OP_CRASH, 20)
    sim.step(1)
    sim.leave(30)
    sim.step(400)
    raw = [e for e in sim.drain_events() if e[1] == 3]
    assert sorted((e[2], e[3]) for e in raw) == [(LEAVE, 30), (FAILED, 20)]
    out = coalesce_loop(raw, MemberEventCoalescer(), coalesce_period=1000, quiescent_period=1000, end_tick=2000)
    assert {ty: m for _, _, ty, m in out} == {LEAVE: [30], FAILED: [20]}
