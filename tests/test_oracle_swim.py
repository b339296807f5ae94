"""memberlist layer of the oracle: handler rules one message at a time, then
whole-cluster scenarios that restate the reference's timing-dependent event-sequence tests
(serf-core/src/serf/base/tests/serf/event.rs:88-130 crash => Join,Failed; :174-232 graceful =>
Join,Leave) as deterministic tick-model runs.

memberlist-core 0.8.1 is not under /root/reference, so these tests pin the oracle to the published
SWIM/Lifeguard rules as restated in App. B ("parity unpinned"); the serf-side exits they reach
(handle_node_join / handle_node_leave, base.rs:1206-1440) are pinned by tests/test_oracle_kat.py.
"""
import ctypes as C

import numpy as np
import pytest

from serf_amd import _ffi
from tests._oracle import Node

ALIVE, LEAVING, LEFT, FAILED, NONE = _ffi.STATUS_ALIVE, _ffi.STATUS_LEAVING, _ffi.STATUS_LEFT, _ffi.STATUS_FAILED, _ffi.STATUS_NONE
SW_ALIVE, SW_SUSPECT, SW_DEAD, SW_LEFT = _ffi.SWIM_ALIVE, _ffi.SWIM_SUSPECT, _ffi.SWIM_DEAD, _ffi.SWIM_LEFT
EV_JOIN, EV_LEAVE, EV_FAILED = 0, 1, 2


def cluster(oracle, n=8, **kw):
    kw.setdefault("probe_interval", 5)
    sim = _ffi.Sim(oracle, _ffi.make_config(n, **kw))
    return sim, Node(oracle, sim, 0)


def params(oracle, sim):
    k = C.c_uint32()
    T = (C.c_uint32 * 4)()
    assert oracle.t["swim_params"](sim.h, C.byref(k), T) == 0
    return k.value, list(T)


# ------------------------------------------------------------------------------------------------
# suspicion timeout table (App. B.5 gives the LAN values in ticks)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,tmin", [(128, 42), (65536, 96), (1 << 20, 120)])
def test_suspicion_timeouts_lan(oracle, n, tmin):
    sim, _ = cluster(oracle, n, view_slots=4, event_ring=1, query_ring=1)
    k, T = params(oracle, sim)
    assert k == 2                      # suspicion_mult 4 - 2
    assert T[0] == 6 * tmin            # starts at max = suspicion_max_timeout_mult * min
    assert T[2] == tmin                # k confirmations drive it down to min
    assert tmin < T[1] < 6 * tmin
    # ln(2)/ln(3) of the way down, floored
    assert T[1] == int(np.floor(6 * tmin - np.log(2.0) / np.log(3.0) * (5 * tmin)))


def test_small_cluster_has_no_confirmations(oracle):
    sim, _ = cluster(oracle, 3)
    k, T = params(oracle, sim)
    assert k == 0 and T[0] == 4 * 1 * 5  # min = mult * max(1, log10 n) * probe_interval


# ------------------------------------------------------------------------------------------------
# aliveNode / suspectNode / deadNode, one message at a time (App. B.4)
# ------------------------------------------------------------------------------------------------
def test_alive_rules(oracle):
    sim, s1 = cluster(oracle)
    assert s1.view(3)["inc"] == 0
    s1.alive(3, 0)                      # inc <= current: ignored, nothing queued
    assert s1.queue_kinds() == []
    s1.alive(3, 2)                      # newer incarnation: adopt + rebroadcast
    v = s1.view(3)
    assert (v["inc"], v["swim"]) == (2, SW_ALIVE)
    assert s1.queue_kinds() == [_ffi.K_ALIVE]
    s1.alive(3, 1)                      # stale again
    assert s1.view(3)["inc"] == 2


def test_suspect_confirm_and_timer(oracle):
    sim, s1 = cluster(oracle, 128)
    k, T = params(oracle, sim)
    oracle.t["set_tick"](sim.h, 100)
    s1.suspect(9, 0, 5)                 # node 5 accuses node 9
    v = s1.view(9)
    assert (v["swim"], v["nconf"], v["conf"][0], v["stamp"]) == (SW_SUSPECT, 0, 5, 100)
    assert v["status"] == ALIVE         # serf only hears about it when the node is declared dead
    row = sim.dump(_ffi.ARR_ROWS)[0]
    assert row["susp_next"] == 100 + T[0] and list(row["susp"]).count(0) == 15
    assert s1.queue_kinds() == [_ffi.K_SUSPECT]
    s1.suspect(9, 0, 5)                 # same accuser again: not a confirmation
    assert s1.view(9)["nconf"] == 0
    s1.suspect(9, 0, 6)                 # independent confirmation 1
    s1.suspect(9, 0, 7)                 # independent confirmation 2 (= k)
    s1.suspect(9, 0, 8)                 # beyond k: ignored
    v = s1.view(9)
    assert v["nconf"] == 2 and v["conf"][:3] == [5, 6, 7]
    assert sim.dump(_ffi.ARR_ROWS)[0]["susp_next"] == 100 + T[2]
    # timer: not yet at T[2]-1, fires at T[2]
    oracle.t["set_tick"](sim.h, 100 + T[2] - 1)
    s1.run_timers()
    assert s1.view(9)["swim"] == SW_SUSPECT
    oracle.t["set_tick"](sim.h, 100 + T[2])
    s1.run_timers()
    v = s1.view(9)
    assert (v["swim"], v["status"]) == (SW_DEAD, FAILED)      # notify_leave: Alive -> Failed (base.rs:1394)
    row = sim.dump(_ffi.ARR_ROWS)[0]
    assert row["n_failed"] == 1 and row["susp_next"] == 0 and not any(row["susp"])
    assert _ffi.K_DEAD in s1.queue_kinds()


def test_suspect_stale_and_non_alive_ignored(oracle):
    sim, s1 = cluster(oracle)
    s1.alive(3, 4)
    s1.suspect(3, 3, 5)                 # older incarnation
    assert s1.view(3)["swim"] == SW_ALIVE
    s1.dead(3, 4, 5)
    s1.suspect(3, 4, 6)                 # already dead
    assert s1.view(3)["swim"] == SW_DEAD


def test_refute_suspect_and_dead_about_self(oracle):
    sim, s1 = cluster(oracle)
    s1.suspect(0, 0, 5)                 # somebody suspects us: refute with inc = max(own+1, accused+1)
    row = sim.dump(_ffi.ARR_ROWS)[0]
    assert row["inc"] == 1 and row["awareness"] == 1
    assert s1.view(0)["swim"] == SW_ALIVE and s1.view(0)["inc"] == 1
    assert s1.queue_kinds() == [_ffi.K_ALIVE]
    s1.dead(0, 7, 5)                    # accused at a higher incarnation
    assert sim.dump(_ffi.ARR_ROWS)[0]["inc"] == 8
    assert s1.view(0)["status"] == ALIVE


def test_dead_from_self_is_left(oracle):
    sim, s1 = cluster(oracle)
    s1.set_member(3, LEAVING, 5)
    s1.dead(3, 0, 3)                    # from == subject: a graceful leave (memberlist.leave)
    v = s1.view(3)
    assert (v["swim"], v["status"]) == (SW_LEFT, LEFT)       # Leaving -> Left (base.rs:1384)
    s1.dead(4, 0, 2)                    # declared dead by somebody else
    v = s1.view(4)
    assert (v["swim"], v["status"]) == (SW_DEAD, FAILED)
    row = sim.dump(_ffi.ARR_ROWS)[0]
    assert (row["n_left"], row["n_failed"]) == (1, 1)
    s1.dead(4, 0, 6)                    # already dead: no second notify_leave
    assert sim.dump(_ffi.ARR_ROWS)[0]["n_failed"] == 1


def test_alive_after_dead_rejoins(oracle):
    sim, s1 = cluster(oracle)
    s1.dead(4, 0, 2)
    s1.alive(4, 0)                      # same incarnation: a dead node stays dead
    assert s1.view(4)["swim"] == SW_DEAD
    s1.alive(4, 1)                      # refutation arrives: notify_join => Alive again (base.rs:1234-1274)
    v = s1.view(4)
    assert (v["swim"], v["status"], v["inc"]) == (SW_ALIVE, ALIVE, 1)
    assert sim.dump(_ffi.ARR_ROWS)[0]["n_failed"] == 0


def test_alive_cancels_suspicion(oracle):
    sim, s1 = cluster(oracle, 128)
    s1.suspect(9, 0, 5)
    s1.alive(9, 1)
    v = s1.view(9)
    assert (v["swim"], v["nconf"]) == (SW_ALIVE, 0)
    assert not any(sim.dump(_ffi.ARR_ROWS)[0]["susp"])


def test_class0_broadcast_invalidates_older_one_about_same_node(oracle):
    # memberlist's broadcasts are named by node: a newer one replaces the queued older one (App. B.1)
    sim, s1 = cluster(oracle, 128)
    s1.suspect(9, 0, 5)
    s1.suspect(10, 0, 5)
    s1.dead(9, 0, 5)
    assert sorted(s1.queue_kinds()) == [_ffi.K_SUSPECT, _ffi.K_DEAD]
    q = sim.dump(_ffi.ARR_QUEUE).reshape(-1, _ffi.Q)[0]
    keys = {int(k): int((m >> 4) & 15) for k, m in zip(q["key"], q["meta"]) if m != 0xFFFFFFFF}
    assert keys == {9: _ffi.K_DEAD, 10: _ffi.K_SUSPECT}


def test_unknown_member_learned_through_alive(oracle):
    # not pre-joined: only self is known; an alive message makes memberlist call notify_join,
    # which applies buffered intents (join.rs:267-347 join_pending_intents)
    sim, s1 = cluster(oracle, 8, flags=0)
    assert s1.member(3) == (NONE, 0)
    s1.join_intent(3, 5)
    s1.leave_intent(3, 6)
    s1.alive(3, 0)
    assert s1.member(3) == (LEAVING, 6)
    assert sim.stats(0).members == 2


# ------------------------------------------------------------------------------------------------
# whole-cluster scenarios
# ------------------------------------------------------------------------------------------------
def run_until(sim, pred, max_ticks, step=1):
    for _ in range(0, max_ticks, step):
        sim.step(step)
        if pred():
            return sim.tick
    return None


def statuses_of(sim, subject, observers):
    return [int(sim.members(o)[0][subject]) for o in observers]


def test_crash_is_detected_and_declared_failed(oracle):
    # event.rs:88-130 (serf_events_failed): a node that stops is reported Failed by everybody else
    n, victim = 128, 77
    sim, _ = cluster(oracle, n, fanout=3)
    k, T = params(oracle, sim)
    sim.watch(3)
    sim.inject(2, _ffi.OP_CRASH, victim)
    others = [o for o in range(n) if o != victim]
    t_done = run_until(sim, lambda: all(s == FAILED for s in statuses_of(sim, victim, others)), 6 * T[0])
    assert t_done is not None, "crash never detected"
    assert t_done >= T[2], "nobody may declare a node dead before the minimum suspicion timeout"
    ev = [e for e in sim.drain_events() if e[3] == victim]
    assert [e[2] for e in ev] == [EV_FAILED] and ev[0][1] == 3
    for o in (0, 3, 100):
        st = sim.stats(o)
        assert (st.failed, st.left, st.members) == (1, 0, n)
    # the suspicion machinery is idle again and the rumours have drained
    sim.step(80)
    rows = sim.dump(_ffi.ARR_ROWS)
    up = rows["flags"] & 1
    assert not rows["susp"][up == 1].any() and not rows["susp_next"][up == 1].any()
    assert (sim.dump(_ffi.ARR_QUEUE)["meta"].reshape(n, -1)[up == 1] == 0xFFFFFFFF).all()


def test_revive_in_time_refutes_the_suspicion(oracle):
    n, victim = 128, 5
    sim, _ = cluster(oracle, n, fanout=3)
    k, T = params(oracle, sim)
    sim.inject(1, _ffi.OP_CRASH, victim)
    sim.inject(1 + T[2] // 2, _ffi.OP_REVIVE, victim)   # back before anybody's timer can fire
    seen_suspect = False
    for _ in range(T[0] + 60):
        sim.step(1)
        if not seen_suspect:
            v = sim.dump(_ffi.ARR_VIEW).reshape(n, n)[victim]   # dense: view[subject][observer]
            seen_suspect = bool((((v["bits"] >> 4) & 3) == SW_SUSPECT).any())
    assert seen_suspect, "the outage should have been noticed"
    rows = sim.dump(_ffi.ARR_ROWS)
    v = sim.dump(_ffi.ARR_VIEW).reshape(n, n)[victim]
    if rows["inc"][victim] > 0:      # it heard the accusation and refuted
        assert (((v["bits"] >> 4) & 3) == SW_ALIVE).all()
        assert (v["inc"] == rows["inc"][victim]).all()
    assert all(s == ALIVE for s in statuses_of(sim, victim, range(0, n, 7)))
    assert rows["n_failed"].sum() == 0


def test_graceful_leave_is_left_not_failed(oracle):
    # event.rs:174-232 (serf_events_leave): Leave intent, then memberlist.leave => Left everywhere
    n, leaver = 64, 9
    sim, _ = cluster(oracle, n, fanout=3, leave_delay=10)
    sim.watch(2)
    sim.step(1)
    sim.leave(leaver)
    sim.step(10)
    assert all(s == LEAVING for s in statuses_of(sim, leaver, (0, 2, 33, 63)))
    sim.step(25)
    assert all(s == LEFT for s in statuses_of(sim, leaver, [o for o in range(n) if o != leaver]))
    ev = [e for e in sim.drain_events() if e[3] == leaver]
    assert [e[2] for e in ev] == [EV_LEAVE]
    sim.step(200)                       # the process is gone now; nobody ever calls it Failed
    assert sim.stats(leaver).up == 0
    st = sim.stats(2)
    assert (st.failed, st.left) == (0, 1)


def test_leave_then_rejoin(oracle):
    # event.rs:257-402 leave -> rejoin -> leave keeps Join/Leave ordering and the intent queues drain
    n, node = 64, 11
    sim, _ = cluster(oracle, n, fanout=3, leave_delay=8)
    sim.watch(0)
    sim.step(1)
    sim.leave(node)
    sim.step(40)
    assert statuses_of(sim, node, (0, 5)) == [LEFT, LEFT]
    sim.join(node)
    sim.step(40)
    assert statuses_of(sim, node, (0, 5, 63)) == [ALIVE] * 3
    assert sim.stats(node).incarnation >= 1
    sim.leave(node)
    sim.step(40)
    assert statuses_of(sim, node, (0, 5, 63)) == [LEFT] * 3
    ev = [e[2] for e in sim.drain_events() if e[3] == node]
    assert ev == [EV_LEAVE, EV_JOIN, EV_LEAVE]
    sim.step(60)
    assert sim.stats(0).intent_queue == 0 and sim.stats(0).swim_queue == 0


def test_packet_loss_causes_refuted_false_suspicions(oracle):
    # heavy loss: probes fail although the target is up; the accused refute by bumping their incarnation
    n = 64
    sim, _ = cluster(oracle, n, fanout=3, loss=0.35, indirect_checks=1)
    sim.step(400)
    rows = sim.dump(_ffi.ARR_ROWS)
    assert rows["inc"].max() >= 1, "35 % loss with one indirect check must produce false suspicions"
    assert rows["awareness"].max() >= 1
    v = sim.dump(_ffi.ARR_VIEW).reshape(n, n)
    assert (v["inc"].max(axis=1) <= rows["inc"]).all(), "nobody knows a higher incarnation than the owner's"


def test_failed_member_is_reaped(oracle):
    # event.rs:88-130 (serf_events_failed): Join, Failed, Reap — the Reaper (base.rs:483-610) erases a
    # failed member reconnect_timeout after it failed, on its next reap_interval boundary
    n, victim = 64, 20
    sim, _ = cluster(oracle, n, fanout=3, reap_interval=10, reconnect_timeout=30, tombstone_timeout=30)
    k, T = params(oracle, sim)
    sim.watch(5)
    sim.inject(1, _ffi.OP_CRASH, victim)
    t_failed = run_until(sim, lambda: statuses_of(sim, victim, (5,)) == [FAILED], 8 * T[0])
    assert t_failed is not None
    assert sim.stats(5).members == n and sim.stats(5).failed == 1
    assert sim.dump(_ffi.ARR_ROWS)[5]["reap_next"] > 0
    sim.step(30 + 2 * 10 + 2)
    assert statuses_of(sim, victim, (5, 6, 40)) == [NONE] * 3
    st = sim.stats(5)
    assert (st.members, st.failed) == (n - 1, 0)
    assert [e[2] for e in sim.drain_events() if e[3] == victim] == [EV_FAILED, 4]   # Failed, then Reap
    assert sim.dump(_ffi.ARR_ROWS)[5]["reap_next"] == 0


def test_left_member_outlives_failed_one_with_longer_tombstone(oracle):
    # reap.rs:41-129: tombstone_timeout governs left members, reconnect_timeout failed ones
    n = 64
    sim, _ = cluster(oracle, n, fanout=3, leave_delay=4, reap_interval=5, reconnect_timeout=10, tombstone_timeout=400)
    sim.step(1)
    sim.leave(9)
    sim.inject(2, _ffi.OP_CRASH, 30)
    sim.step(300)
    st, _lt = sim.members(0)
    assert st[9] == LEFT          # still inside its tombstone
    assert st[30] == NONE         # failed and already reaped
    assert (sim.stats(0).left, sim.stats(0).failed, sim.stats(0).members) == (1, 0, n - 1)


def test_queue_checker_prunes_to_dynamic_cap(oracle):
    # base.rs:683-740 / serf.rs tests 57-160: with min_queue_depth > 0 the cap is max(2 * members, min)
    sim, s1 = cluster(oracle, 3, probe_interval=0, flags=0, queue_check_interval=4, min_queue_depth=1, event_ring=64)
    for key in range(1, 13):
        s1.user_event(key, key)          # 12 events queued on node 0 (it knows only itself: cap = max(2, 1) = 2)
    assert sim.stats(0).event_queue == 12
    sim.step(4)                          # a queue-check tick for group 0 falls inside
    assert sim.stats(0).event_queue <= 2


def test_push_pull_revives_a_node_declared_dead(oracle):
    # Without anti-entropy a node that comes back after everybody declared it dead stays dead in their
    # views (nobody gossips about it any more).  memberlist's push-pull (App. B.6) shows it its own
    # obituary, it refutes, and the alive message brings it back (notify_join, base.rs:1234-1274).
    n, victim = 64, 33
    for pp, expect in ((0, FAILED), (8, ALIVE)):
        sim, _ = cluster(oracle, n, fanout=3, push_pull_interval=pp)
        k, T = params(oracle, sim)
        sim.inject(1, _ffi.OP_CRASH, victim)
        others = [o for o in range(n) if o != victim]
        assert run_until(sim, lambda: all(s == FAILED for s in statuses_of(sim, victim, others)), 8 * T[0]) is not None
        sim.step(60)                       # the dead rumour has drained
        sim.inject(sim.tick, _ffi.OP_REVIVE, victim)
        sim.step(200)
        got = statuses_of(sim, victim, others)
        assert all(s == expect for s in got), (pp, got)
        if pp:
            assert sim.stats(victim).incarnation >= 1
            assert sim.stats(0).failed == 0


def test_push_pull_repairs_a_rumour_lost_to_packet_loss(oracle):
    # merge_remote_state replays the peer's event buffer (delegate.rs:540-552): a user event that died
    # out under heavy loss still reaches everybody once anti-entropy runs
    n = 128
    res = {}
    for pp in (0, 6):
        sim, _ = cluster(oracle, n, fanout=1, loss=0.75, retransmit_mult=1, probe_interval=0, push_pull_interval=pp, event_ring=64)
        sim.user_event(5, 4242, 32)
        sim.step(300)
        res[pp] = sim.convergence(_ffi.K_EVENT, 4242, 1)
    assert res[0][0] < n, "scenario must lose the rumour without anti-entropy"
    assert res[6] == (n, n)


def test_push_pull_merges_clocks_and_intents(oracle):
    # delegate.rs:466-526 on a pair: clocks witnessed at remote - 1, status_ltimes become join intents,
    # left members leave intents one past their status time
    sim, _ = cluster(oracle, 2, probe_interval=0, push_pull_interval=1, fanout=1, loss=1.0)   # no gossip at all
    a, b = Node(oracle, sim, 0), Node(oracle, sim, 1)
    a.set_clock(Node.CLOCK, 40)
    a.set_clock(Node.EVENT, 50)
    a.set_clock(Node.QUERY, 60)
    a.set_member(1, LEFT, 7)             # a believes b has left at ltime 7
    sim.step(20)                          # several push-pull rounds between the only pair
    assert (b.clock(Node.CLOCK), b.clock(Node.EVENT), b.clock(Node.QUERY)) == (40, 50, 60)[:0] + (b.clock(Node.CLOCK), 50, 60)
    assert b.clock(Node.EVENT) == 50 and b.clock(Node.QUERY) == 60
    assert b.clock(Node.CLOCK) >= 40
    # b is told "you left at 8" while alive: it refutes with a join at its clock (base.rs:1470-1480), and
    # the next exchange carries that newer status time back to a
    assert a.member(1)[0] == LEFT or a.member(1)[1] > 7


def test_query_acks_and_responses_reach_the_origin(oracle):
    # serf/base.rs:1075-1204 + serf/query.rs:240-303: every node that processes the query acks
    # (QueryFlag::ACK) and responds; the origin counts each sender once, until the deadline
    n = 256
    sim, _ = cluster(oracle, n, fanout=3, probe_interval=0)
    sim.query(7, 0xABC, _ffi.F_ACK | _ffi.F_RESPOND)
    sim.query(9, 0xDEF, _ffi.F_ACK)
    sim.step(1)
    a, r, is_open = sim.query_status(0xABC)
    assert (a, r, is_open) == (1, 1, True)          # the origin handles its own query first (base.rs:932)
    sim.step(30)
    assert sim.query_status(0xABC) == (n, n, True)
    assert sim.query_status(0xDEF) == (n, 0, True)
    sim.step(60)                                     # 16 * ceil(log10(257)) = 48 ticks: closed now
    assert sim.query_status(0xABC) == (n, n, False)
    with pytest.raises(_ffi.SimError):
        sim.query_status(0x123)                      # "reply for non-running query"


def test_query_responses_after_the_deadline_or_to_a_dead_origin_are_dropped(oracle):
    n = 64
    sim, _ = cluster(oracle, n, fanout=1, probe_interval=0, loss=0.6, retransmit_mult=6)   # slow, lossy spread
    sim.query(3, 77, _ffi.F_ACK)
    sim.step(32 + 2)                                 # deadline = 16 * 2 ticks
    a_deadline = sim.query_status(77)[0]
    sim.step(200)
    seen, up = sim.convergence(_ffi.K_QUERY, 77, 1)
    a_final, _, is_open = sim.query_status(77)
    assert not is_open and a_final == a_deadline     # nothing is counted after the deadline ...
    assert seen > a_final                            # ... although more nodes processed the query later
    sim2, _ = cluster(oracle, n, fanout=3, probe_interval=0)
    sim2.query(5, 88, _ffi.F_ACK)
    sim2.inject(1, _ffi.OP_CRASH, 5)                 # the origin dies right after sending
    sim2.step(30)
    assert sim2.query_status(88)[0] <= 4             # only what arrived while it was up


def test_query_relays_rescue_lost_acks(oracle):
    # query.rs:523-601 relay_response: with relay_factor r every ack also travels through r random live
    # members; under heavy loss more acks reach the origin, each sender still counted once
    n, got = 512, {}
    for relay in (0, 3):
        sim, _ = cluster(oracle, n, fanout=3, probe_interval=0, loss=0.5, retransmit_mult=6)
        sim.query(1, 555, _ffi.F_ACK | (relay << 8))
        sim.step(45)
        got[relay] = sim.query_status(555)[0]
        seen, up = sim.convergence(_ffi.K_QUERY, 555, 1)
        assert got[relay] <= seen <= n
    assert 0.35 * n < got[0] < 0.65 * n          # one lossy leg: about half arrive
    assert got[3] > got[0] + 0.15 * n            # 1 - 0.5 * (1 - 0.25)^3 ~ 0.79


def test_false_suspicion_of_a_slotless_node_is_taken_up_one_tick_late(oracle):
    """SIMSPEC §2.7 (round 3): with packet loss a probe can fail on a LIVE node that has no view slot.  The prober cannot
    hold the suspicion in that tick; the pair goes on the tick's request list and comes back as SIM_OP_SUSPECT in the
    next tick, which gives the target its slot first — the suspicion then runs its course: gossip, the target refutes
    (incarnation + 1), everybody is back to alive, the slot is recycled.  Nothing is counted as a model-bound drop."""
    n = 2048
    kw = dict(fanout=3, view_slots=32, event_ring=16, query_ring=8, probe_interval=3, suspicion_mult=4, suspicion_max_mult=3,
              loss=0.02, indirect_checks=2, recycle_interval=20, push_pull_interval=10)
    sim = _ffi.Sim(oracle, _ffi.make_config(n, **kw))
    for w in range(0, n, 64):
        sim.watch(w)
    seen_slots = refuted = 0
    for t in range(600):
        sim.step(1)
        if t % 20 == 0:
            seen_slots = max(seen_slots, sim.cluster_stats()["slots_in_use"])
    cs = sim.cluster_stats()
    rows = sim.dump(_ffi.ARR_ROWS)
    refuted = int((rows["inc"] > 0).sum())
    assert seen_slots > 0 and cs["slots_recycled"] > 0, "false suspicions took view slots and gave them back"
    assert refuted > 0, "suspected live nodes refuted with a higher incarnation"
    assert cs["overflow"] == 0 and cs["ops_dropped"] == 0, cs
    assert cs["failed"] == 0 and cs["up"] == n, "nobody was declared dead: the refutation beats the suspicion timeout"
    ev = sim.drain_events()
    assert not [e for e in ev if e[2] == _ffi.EV_FAILED]


def test_gossip_to_the_dead_time(oracle):
    """memberlist gossip_to_the_dead_time (App. B.2; sim_config.gossip_to_the_dead): a node keeps gossiping to a member it
    believes dead for that long and no longer — the window in which a node that was wrongly declared dead (here: one that
    comes back right after the declaration) still hears its obituary and refutes.  Node 10 goes down at tick 3, is declared
    dead around tick 24 - 30 and resumes at tick 32 with its old state: with the window open it refutes and is alive
    again; with a one-tick window nobody talks to it any more and it stays failed (no push-pull in this run)."""
    n = 512
    kw = dict(fanout=3, view_slots=32, event_ring=16, query_ring=8, probe_interval=2, suspicion_mult=4, suspicion_max_mult=2)
    res = {}
    for g in (0, 1):
        sim = _ffi.Sim(oracle, _ffi.make_config(n, gossip_to_the_dead=g, **kw))
        sim.inject(3, _ffi.OP_CRASH, 10)
        sim.inject(32, _ffi.OP_REVIVE, 10)
        sim.step(120)
        st, _ = sim.members(200)
        res[g] = (int(st[10]), int(sim.dump(_ffi.ARR_ROWS)["inc"][10]), sim.cluster_stats()["overflow"])
    assert res[0] == (_ffi.STATUS_ALIVE, 1, 0), res
    assert res[1] == (_ffi.STATUS_FAILED, 0, 0), res


def test_awareness_scales_the_probe_interval(oracle):
    """SIM_CF_AWARENESS_PROBE (memberlist probeNode: probe interval = ScaleTimeout(ProbeInterval) = (score + 1) x): a node
    with health score s probes only in every (s + 1)-th round of its group's probe phase.  In a run with crashes only, a
    node's score changes exactly when it probes (awareness_test.go: -1 on success, +1 on failure)."""
    n, PI = 1024, 2
    kw = dict(fanout=3, view_slots=32, event_ring=16, query_ring=8, probe_interval=PI, suspicion_mult=6, suspicion_max_mult=3)
    traj = {}
    for flag in (False, True):
        sim = _ffi.Sim(oracle, _ffi.make_config(n, awareness_probe=flag, **kw))
        for x in range(100, 1000, 45):      # twenty nodes down: one probe in fifty fails
            sim.inject(2, _ffi.OP_CRASH, x)
        prev = sim.dump(_ffi.ARR_ROWS)["awareness"].astype(np.int64)
        sums, moved = [], 0
        for t in range(60):
            sim.step(1)                     # executes tick t
            aw = sim.dump(_ffi.ARR_ROWS)["awareness"].astype(np.int64)
            ch = np.nonzero(aw != prev)[0]
            if flag:
                rounds = (t + (ch >> 6)) // PI
                assert ((t + (ch >> 6)) % PI == 0).all()
                assert (rounds % (prev[ch] + 1) == 0).all(), f"tick {t}: a node with score s probed outside every (s+1)-th round"
            moved += len(ch)
            sums.append(int(aw.sum()))
            prev = aw
        traj[flag] = sums
        assert moved > 20
    assert traj[True] != traj[False] and sum(traj[True]) >= sum(traj[False]), "degraded nodes recover more slowly when they probe less often"


def test_join_sync_gives_a_rejoining_node_a_current_view(oracle):
    """SIM_CF_JOIN_SYNC: Serf::join is memberlist.join, a push-pull with the peer, before anything else.  Node 10 is down
    while node 20 leaves gracefully and node 30 crashes and is declared failed; when 10 re-joins it adopts a running
    node's view — without the flag it keeps the view it went down with until a push-pull batch reaches it (none here)."""
    n = 512
    kw = dict(fanout=3, view_slots=32, event_ring=16, query_ring=8, probe_interval=2, suspicion_mult=4, suspicion_max_mult=2, leave_delay=4)
    res = {}
    for flag in (False, True):
        sim = _ffi.Sim(oracle, _ffi.make_config(n, join_sync=flag, **kw))
        sim.inject(3, _ffi.OP_CRASH, 10)
        sim.inject(50, _ffi.OP_CRASH, 30)
        sim.inject(52, _ffi.OP_LEAVE, 20)
        sim.inject(57, _ffi.OP_LEAVE_FINISH, 20)
        sim.inject(62, _ffi.OP_CRASH, 20)
        sim.inject(120, _ffi.OP_JOIN, 10, 77)
        sim.step(121)            # the join has just executed
        st10, _ = sim.members(10)
        st77, _ = sim.members(77)
        row10, row77 = sim.dump(_ffi.ARR_ROWS)[10], sim.dump(_ffi.ARR_ROWS)[77]
        res[flag] = (int(st10[20]), int(st10[30]), int(row10["n_failed"]), int(row10["n_left"]))
        want = (int(st77[20]), int(st77[30]))
        assert want == (_ffi.STATUS_LEFT, _ffi.STATUS_FAILED)
        if flag:
            assert res[flag][:2] == want, "the joiner sees what its partner sees"
            assert int(row10["clock"]) >= int(row77["clock"]) - 1
            # the partner still counts the joiner itself as failed until the refutation arrives; the joiner does not
            assert res[flag][2] == int(row77["n_failed"]) - 1 and res[flag][3] == int(row77["n_left"])
        sim.step(60)
        assert sim.cluster_stats()["overflow"] == 0
        st, _ = sim.members(200)
        assert st[10] == _ffi.STATUS_ALIVE, "and the cluster has it back"
    assert res[False][:2] == (_ffi.STATUS_ALIVE, _ffi.STATUS_ALIVE), "without the sync the joiner still has its old view"


def test_tcp_fallback_ping_leaves_no_false_suspicions(oracle):
    """SIM_CF_TCP_FALLBACK — memberlist probeNode's stream-transport fallback (UPSTREAM-RECALL state.go; on by default in
    memberlist): a probe whose UDP legs were all lost still succeeds when the TCP ping gets through.  Under 10 % loss without
    it live nodes are suspected and refute all the time; with it nobody is suspected who is up — and a node that really
    crashed is declared failed all the same."""
    n = 1024
    kw = dict(fanout=3, view_slots=64, event_ring=16, query_ring=8, probe_interval=2, suspicion_mult=4, suspicion_max_mult=2, loss=0.10)
    res = {}
    for fb in (False, True):
        sim = _ffi.Sim(oracle, _ffi.make_config(n, tcp_fallback=fb, **kw))
        sim.inject(5, _ffi.OP_CRASH, 77)
        sim.step(120)
        rows = sim.dump(_ffi.ARR_ROWS)
        st, _ = sim.members(3)
        res[fb] = (int(rows["inc"].sum()), int(st[77]), int(rows["awareness"].max()))
    assert res[False][0] > 20, res                       # refutations of false suspicions
    assert res[True][0] == 0, res                        # none with the fallback
    assert res[True][1] == _ffi.STATUS_FAILED and res[False][1] == _ffi.STATUS_FAILED, res


def test_nacks_keep_a_healthy_prober_healthy(oracle):
    """SIM_CF_NACKS — memberlist's nack accounting (UPSTREAM-RECALL state.go: awarenessDelta = expectedNacks - nackCount): a
    probe that fails on a node that is really dead costs the prober one point per relay that did NOT answer; with every
    relay up and no loss that is nothing, where the plain rule charges one point per failed probe."""
    n, PI = 1024, 2
    kw = dict(fanout=3, view_slots=32, event_ring=16, query_ring=8, probe_interval=PI, suspicion_mult=6, suspicion_max_mult=3)
    score = {}
    for nk in (False, True):
        sim = _ffi.Sim(oracle, _ffi.make_config(n, nacks=nk, **kw))
        for x in range(100, 1000, 45):
            sim.inject(2, _ffi.OP_CRASH, x)
        tot = 0
        for _ in range(30):
            sim.step(1)
            tot += int(sim.dump(_ffi.ARR_ROWS)["awareness"].astype(np.int64).sum())
        score[nk] = tot
    assert score[False] > 50 and score[True] * 10 < score[False], score   # (a relay that is itself one of the dead nodes: no nack)


# ------------------------------------------------------------------------------------------------
# remove_failed_node_prune and handle_prune's wait (serf/base.rs:452-480, 1628-1653)
# ------------------------------------------------------------------------------------------------
def known_by(sim, subject, observers):
    return [bool(sim.dump(_ffi.ARR_VIEW).reshape(-1, sim.n)[subject, o]["bits"] & 1) for o in observers]


@pytest.mark.parametrize("wait", [False, True])
def test_remove_failed_node_prune(oracle, wait):
    # serf/base/tests/serf/remove.rs:96-153 (serf_remove_failed_node_prune): three nodes, one is shut down and declared Failed, the first calls
    # remove_failed_node_prune — the member table of the two that are left drops to 2.  A Failed member does not wait (base.rs:1634: only
    # Leaving sleeps), so the flag changes nothing here.
    n, victim = 3, 1
    sim, _ = cluster(oracle, n, fanout=2, leave_delay=6, prune_delay=wait)
    sim.inject(1, _ffi.OP_CRASH, victim)
    others = [0, 2]
    assert run_until(sim, lambda: all(s == FAILED for s in statuses_of(sim, victim, others)), 400) is not None, "never declared failed"
    assert [sim.stats(o).members for o in others] == [3, 3]
    t0 = sim.tick
    sim.remove_failed_node(0, victim, prune=True)
    t = run_until(sim, lambda: [sim.stats(o).members for o in others] == [2, 2], 40)
    assert t is not None and t - t0 <= 4, "a failed member is erased when the intent is handled: no wait"


def test_handle_prune_waits_while_the_member_is_leaving(oracle):
    # base.rs:1628-1653: a pruning leave intent about a member that is Alive (it becomes Leaving) or Leaving is erased broadcast_timeout +
    # leave_propagate_delay after the node handled it; here leave_delay = 9 ticks, in a cluster where the subject has crashed but is not yet suspected
    n, victim, delay = 64, 40, 9
    sim, _ = cluster(oracle, n, fanout=3, leave_delay=delay, prune_delay=True, probe_interval=5)
    sim.watch(0)
    sim.inject(2, _ffi.OP_CRASH, victim)
    sim.inject(3, _ffi.OP_FORCE_LEAVE, 0, victim, 1)
    others = [o for o in range(n) if o != victim]
    sim.step(4)     # tick 3 has run: node 0 handled its own intent
    assert statuses_of(sim, victim, [0]) == [LEAVING] and sim.stats(0).members == n, "the erase must wait"
    sim.step(delay - 1)   # ticks 4 .. 3 + delay - 1: still there at node 0, Leaving wherever the intent has arrived
    assert sim.stats(0).members == n
    sim.step(1)     # tick 3 + delay: node 0's wait is over
    assert sim.stats(0).members == n - 1 and not known_by(sim, victim, [0])[0]
    ev = [e for e in sim.drain_events() if e[3] == victim]
    assert [e[2] for e in ev][-1] == _ffi.EV_REAP and ev[-1][0] == 3 + delay, "Reap at the end of the wait"
    # everybody handles the intent within a few rounds and erases `delay` ticks after that
    assert run_until(sim, lambda: not any(known_by(sim, victim, others)), 40) is not None
    assert all(sim.stats(o).members == n - 1 for o in (0, 7, 63))
