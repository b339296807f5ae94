"""The oracle (CPU) and the HIP library (GPU) against the memberlist half of the THIRD model (tests/third_model_swim.py: pure Python,
dictionary state, unbounded, written from SURVEY.md Appendix B and the SIMSPEC — VERDICT r4 item 6).  Closed loop: the model makes its
own packets out of its own TransmitLimitedQueue, delivers them by memberlist's kRandomNodes, loses them by the specified draw, runs its
own suspicion timers and probes; after EVERY tick the implementation under test must show the same packets in flight, the same queues
in drain order (class, transmits, length, kind, flags, key, value), clocks, SerfState, member tables, memberlist states, incarnations,
confirmers, awareness and de-dup rings.  SWIM on, packet loss on, crashes, graceful leaves, re-joins, refutations.  Model bounds never
bite in these runs (overflow == 0 is asserted).  memberlist-core's source is absent, so rows a13 / a16 stay parity-unpinned; what this
removes is the common mode of two restatements by one author."""
import numpy as np
import pytest

from serf_amd import _ffi
from tests import third_model as tm
from tests import third_model_swim as tms
from tests._oracle import load_oracle

RING_EV, RING_Q, PG = 64, 32, 4


def _packets(sim, n, fanout, PG=PG):
    """the packets in flight, sender-indexed (the canonical form with memberlist's kRandomNodes): [sender][slot] -> records"""
    inbox = sim.dump(_ffi.ARR_INBOX).reshape(fanout * PG, n)
    out = []
    for snd in range(n):
        per = []
        for k in range(fanout):
            recs = []
            for pg in range(PG):
                pk = inbox[k * PG + pg, snd]
                for r in range(4):
                    hm = int(pk["hi_meta"][r])
                    kind = (hm >> 4) & 15
                    if kind == 0:
                        continue
                    v48 = int(pk["val_lo"][r]) | ((hm >> 16) << 32)
                    val = (v48 & 0xFFFFFF) | ((v48 >> 24) << 32) if kind in (tms.K_SUSPECT, tms.K_DEAD) else v48
                    recs.append((kind, hm & 15, 63 - ((hm >> 8) & 63), int(pk["key"][r]), val))
            per.append(recs)
        out.append(per)
    return out


def run(sim, n, ops, ticks, joined, **kw):
    fanout = kw["fanout"]
    PG = kw.get("pkt_records", 4 * globals()["PG"]) // 4        # pages of a packet: 4 records each
    par = tms.Params(n, fanout, kw["probe_interval"], kw.get("suspicion_mult", 4), kw.get("suspicion_max_mult", 6), kw.get("indirect_checks", 3),
                     kw.get("retransmit_mult", 4), kw.get("loss", 0.0), kw.get("pkt_records", 16), kw.get("leave_delay", 30), push_pull_interval=kw.get("push_pull_interval", 0),
                     reap_interval=kw.get("reap_interval", 0), reconnect_timeout=kw.get("reconnect_timeout", 432000),
                     tombstone_timeout=kw.get("tombstone_timeout", 432000), intent_timeout=kw.get("intent_timeout", 0),
                     queue_check_interval=kw.get("queue_check_interval", 0), max_queue_depth=kw.get("max_queue_depth", 4096),
                     reconnect_interval=kw.get("reconnect_interval", 0), awareness_probe=kw.get("awareness_probe", False),
                     tcp_fallback=kw.get("tcp_fallback", False), nacks=kw.get("nacks", False), gossip_to_the_dead=kw.get("gossip_to_the_dead", 0),
                     join_sync=kw.get("join_sync", False), prune_delay=kw.get("prune_delay", False))
    model = tms.Cluster(par, RING_EV, RING_Q, joined)
    by_tick = {}
    for o in ops:
        by_tick.setdefault(o[0], []).append(o[1:])
        sim.inject(*o)
    seen_kinds = set()
    for t in range(ticks):
        model.step(by_tick.get(t, ()))
        sim.step(1)
        rows = sim.dump(_ffi.ARR_ROWS)
        view = sim.dump(_ffi.ARR_VIEW).reshape(n, n)        # dense view: [subject][observer]
        er = sim.dump(_ffi.ARR_ERING).reshape(RING_EV, n)
        qr = sim.dump(_ffi.ARR_QRING).reshape(RING_Q, n)
        queue = sim.dump(_ffi.ARR_QUEUE).reshape(n, _ffi.Q)
        assert int(rows["overflow"].sum()) == 0, f"tick {t}: a model bound was hit"
        # ---- the packets sent this tick
        got = _packets(sim, n, fanout, PG)
        for i in range(n):
            for k in range(fanout):
                want = [(kd, fl, ln, key, val) for kd, fl, ln, key, val in (model.flight[i][k] or ())]
                assert got[i][k] == want, f"tick {t} sender {i} slot {k}: packet {got[i][k]} != {want}"
                seen_kinds.update(r[0] for r in want)
        for i, x in enumerate(model.nodes):
            w = f"tick {t} node {i}"
            assert (int(rows["clock"][i]), int(rows["event_clock"][i]), int(rows["query_clock"][i])) == \
                   (x.clock.time(), x.event_clock.time(), x.query_clock.time()), w
            assert bool(rows["flags"][i] & 1) == x.up and ((int(rows["flags"][i]) >> 1) & 3) == x.state, w
            assert (int(rows["inc"][i]), int(rows["awareness"][i])) == (x.inc, x.awareness), w
            st = [m[0] for m in x.members.values()]
            assert (int(rows["n_known"][i]), int(rows["n_failed"][i]), int(rows["n_left"][i])) == (len(st), st.count(tm.FAILED), st.count(tm.LEFT)), w
            # ---- the queue in drain order
            live = [(int(r["meta"]) >> 30, (int(r["meta"]) >> 24) & 63, 63 - ((int(r["meta"]) >> 18) & 63), (int(r["meta"]) >> 4) & 15,
                     int(r["meta"]) & 15, int(r["key"]), int(r["val"])) for r in queue[i] if r["meta"] != 0xFFFFFFFF]
            want = [(e[0], e[1], e[2], e[4], e[5], e[6], e[7]) for e in x.drain_order()]
            assert live == want, f"{w}: queue {live} != {want}"
            # ---- member table, memberlist state, suspicions
            for s in range(n):
                e = view[s, i]
                bits = int(e["bits"])
                if s in x.members:
                    assert bits & 1 and ((bits >> 1) & 7, int(e["ltime"])) == tuple(x.members[s]), f"{w} subject {s}"
                    ml = x.ml[s]
                    assert ((bits >> 4) & 3, int(e["inc"])) == (ml[0], ml[1]), f"{w} subject {s}: memberlist state"
                    if ml[0] == tms.ML_SUSPECT:
                        conf = x.susp[s][1]
                        assert (bits >> 8) & 7 == len(conf) - 1 and [int(c) for c in e["conf"][:len(conf)]] == conf, f"{w} subject {s}: confirmers"
                        assert (bits >> 11) == x.susp[s][0], f"{w} subject {s}: suspicion start"
                else:
                    it = x.intents.get(s)
                    assert not (bits & 1) and s not in x.ml, f"{w} subject {s}"
                    assert ((bits >> 6) & 3, int(e["ltime"])) == ((it[0], it[1]) if it else (0, 0)), f"{w} subject {s} intent"
            for ring, buf, name in ((er, x.event_buf, "event"), (qr, x.query_buf, "query")):
                for j, want in enumerate(buf):
                    b = ring[j, i]
                    got_k = [int(v) for v in b["keys"] if v]
                    if want is None:
                        assert not got_k, f"{w} {name} bucket {j}"
                    else:
                        assert (int(b["ltime"]), got_k) == (want[0], want[1]), f"{w} {name} bucket {j}"
    return seen_kinds, model


def _schedule(n, ticks, seed):
    """user events, queries, graceful leaves (leave intent, memberlist.leave after the leave delay, shutdown), crashes, re-joins"""
    rng = np.random.default_rng(seed)
    ops, key = [], 100
    busy = {}
    for t in range(2, ticks - 30):
        if rng.random() < 0.45:
            node = int(rng.integers(0, n))
            if busy.get(node, 0) > t:
                continue
            r = rng.random()
            key += 1
            if r < 0.4:
                ops.append((t, _ffi.OP_USER_EVENT, node, key, 40))
            elif r < 0.6:
                ops.append((t, _ffi.OP_QUERY, node, key, 0))
            elif r < 0.8:   # crash, suspected / declared dead by the others, back with a refuting incarnation
                ops.append((t, _ffi.OP_CRASH, node, 0, 0))
                back = t + int(rng.integers(12, 40))
                ops.append((back, _ffi.OP_JOIN, node, 0, 0))
                busy[node] = back + 10
            else:           # Serf::leave
                ops.append((t, _ffi.OP_LEAVE, node, 0, 0))
                ops.append((t + 4, _ffi.OP_LEAVE_FINISH, node, 0, 0))
                ops.append((t + 8, _ffi.OP_CRASH, node, 0, 0))
                busy[node] = ticks
    ops.sort(key=lambda o: o[0])
    return ops


def _burst_schedule(n, ticks, seed):
    """the schedule above plus bursts of user events and queries from different nodes in one tick: queues that hold four and more of serf's messages"""
    ops = _schedule(n, ticks, seed)
    rng = np.random.default_rng(seed + 1000)
    key = 5000
    for t in range(6, ticks - 30, 9):
        for _ in range(5):
            key += 1
            ops.append((t, _ffi.OP_USER_EVENT if key % 3 else _ffi.OP_QUERY, int(rng.integers(0, n)), key, 40 if key % 3 else 0))
    ops.sort(key=lambda o: o[0])
    return ops


CASES = [(11, 48, 3, 0.03, 2), (12, 64, 4, 0.0, 3), (13, 33, 3, 0.08, 1), (14, 24, 2, 0.02, 2)]
KW = dict(view_slots=0, event_ring=RING_EV, query_ring=RING_Q, leave_delay=4, pkt_records=4 * PG, suspicion_mult=3, suspicion_max_mult=2,
          flags=_ffi.CF_BASELINE_JOINED | _ffi.CF_RANDOM_FANOUT)


@pytest.mark.parametrize("seed,n,fanout,loss,pi", CASES)
def test_oracle_matches_the_third_model_with_the_memberlist_layer_on(seed, n, fanout, loss, pi):
    kw = dict(KW, fanout=fanout, loss=loss, probe_interval=pi)
    sim = _ffi.Sim(load_oracle(), _ffi.make_config(n, **kw))
    kinds, model = run(sim, n, _schedule(n, 110, seed), 110, True, **kw)
    assert {tms.K_ALIVE, tms.K_SUSPECT, tms.K_DEAD} <= kinds, kinds   # the scenario did exercise refutations, suspicions and declarations


# with memberlist's push-pull (B.6) and the serf delegate's merge_remote_state (delegate.rs:427-554) in the third model: a batch every few
# ticks (the interval is scaled by log2 n and cut into eight classes), under loss — left members, suspects and dead nodes cross in the merges
PP_CASES = [(21, 48, 3, 0.05, 2, 12), (22, 64, 4, 0.02, 3, 8), (23, 33, 3, 0.1, 1, 16)]


@pytest.mark.parametrize("seed,n,fanout,loss,pi,ppi", PP_CASES)
def test_oracle_matches_the_third_model_with_push_pull(seed, n, fanout, loss, pi, ppi):
    kw = dict(KW, fanout=fanout, loss=loss, probe_interval=pi, push_pull_interval=ppi)
    sim = _ffi.Sim(load_oracle(), _ffi.make_config(n, **kw))
    kinds, model = run(sim, n, _schedule(n, 110, seed), 110, True, **kw)
    assert {tms.K_ALIVE, tms.K_SUSPECT, tms.K_DEAD} <= kinds, kinds


# with the Reaper (base.rs:483-610): failed members erased after the reconnect timeout, left ones after the tombstone timeout, buffered intents
# after the intent timeout — short ones, so that members are reaped and come back inside the run
REAP_CASES = [(31, 48, 3, 0.03, 2, 0), (32, 64, 4, 0.0, 3, 12), (33, 40, 3, 0.06, 1, 0)]
REAP_KW = dict(reap_interval=4, reconnect_timeout=14, tombstone_timeout=22, intent_timeout=10)


@pytest.mark.parametrize("seed,n,fanout,loss,pi,ppi", REAP_CASES)
def test_oracle_matches_the_third_model_with_the_reaper(seed, n, fanout, loss, pi, ppi):
    kw = dict(KW, fanout=fanout, loss=loss, probe_interval=pi, push_pull_interval=ppi, **REAP_KW)
    sim = _ffi.Sim(load_oracle(), _ffi.make_config(n, **kw))
    ev0 = len(sim.drain_events())
    for w in range(0, n, 7):
        sim.watch(w)
    kinds, model = run(sim, n, _schedule(n, 110, seed), 110, True, **kw)
    assert any(e[2] == _ffi.EV_REAP for e in sim.drain_events()), "the run must reap somebody"


# handle_prune's wait (base.rs:1628-1653, SIM_CF_PRUNE_DELAY): a pruning leave intent about a member that is Alive or Leaving erases it leave_delay ticks
# after the node handled it; a Failed or Left member goes at once.  Forced removals of running members (they refute: the join intent races the erase), of
# members that are leaving by themselves, of crashed members before and after they were declared failed
def _prune_schedule(n, ticks, seed):
    rng = np.random.default_rng(seed)
    ops, key = [], 300
    who = rng.choice(n, 10, replace=False).tolist()
    t = 3
    for i, x in enumerate(who):
        by = int(rng.integers(0, n))
        by = by if by != x else (x + 1) % n
        if i % 4 == 0:      # a running member
            ops.append((t, _ffi.OP_FORCE_LEAVE, by, x, 1))
        elif i % 4 == 1:    # a member on its way out (its process dies before the removal reaches it: a leaving node that erased ITSELF would
            ops.append((t, _ffi.OP_LEAVE, x, 0, 0))         # keep its memberlist state in the third model and lose it in the simulator — §2.7's merged entry)
            ops.append((t + 1, _ffi.OP_CRASH, x, 0, 0))
            ops.append((t + 3, _ffi.OP_FORCE_LEAVE, by, x, 1))
        elif i % 4 == 2:    # a crashed member, at once
            ops.append((t, _ffi.OP_CRASH, x, 0, 0))
            ops.append((t + 1, _ffi.OP_FORCE_LEAVE, by, x, 1))
        else:               # a crashed member, once it has been declared failed
            ops.append((t, _ffi.OP_CRASH, x, 0, 0))
            ops.append((t + 22, _ffi.OP_FORCE_LEAVE, by, x, 1))
            ops.append((t + 23, _ffi.OP_FORCE_LEAVE, (by + 2) % n, x, 0))
        t += 7
    for tt in range(2, ticks - 25, 5):
        key += 1
        ops.append((tt, _ffi.OP_USER_EVENT, int(rng.integers(0, n)), key, 40))
    ops = [o for o in ops if o[0] < ticks - 12]
    ops.sort(key=lambda o: o[0])
    return ops


PRUNE_CASES = [(61, 48, 3, 0.02, 2, 5), (62, 64, 4, 0.0, 3, 2), (63, 40, 3, 0.05, 2, 9)]


def _prune_run(lib, seed, n, fanout, loss, pi, delay):
    kw = dict(KW, fanout=fanout, loss=loss, probe_interval=pi, leave_delay=delay, prune_delay=True, reap_interval=6, reconnect_timeout=60, tombstone_timeout=60)
    sim = _ffi.Sim(lib, _ffi.make_config(n, **kw))
    sim.drain_events()
    for w in range(0, n, 3):
        sim.watch(w)
    ticks = 120
    ops = _prune_schedule(n, ticks, seed)
    run(sim, n, ops, ticks, True, **kw)
    reaps = [e for e in sim.drain_events() if e[2] == _ffi.EV_REAP]
    assert reaps, "somebody must have been erased"
    # the same schedule without the wait erases earlier: the two runs must differ (the flag does something)
    return sim.digest()


@pytest.mark.parametrize("seed,n,fanout,loss,pi,delay", PRUNE_CASES)
def test_oracle_matches_the_third_model_with_handle_prunes_wait(seed, n, fanout, loss, pi, delay):
    _prune_run(load_oracle(), seed, n, fanout, loss, pi, delay)


def test_create_refuses_the_prune_wait_without_the_memberlist_layer():
    with pytest.raises(Exception):
        _ffi.Sim(load_oracle(), _ffi.make_config(16, fanout=3, prune_delay=True))


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n,fanout,loss,pi,delay", PRUNE_CASES[:2])
def test_hip_matches_the_third_model_with_handle_prunes_wait(hiplib, seed, n, fanout, loss, pi, delay):
    assert _prune_run(hiplib, seed, n, fanout, loss, pi, delay) == _prune_run(load_oracle(), seed, n, fanout, loss, pi, delay)


# with the Reconnector (base.rs:612-681): nodes that crash and silently resume stay failed in the others' tables until somebody's reconnect attempt
# (a push-pull pair of its own, two ticks after the draw) or a push-pull batch reaches them
def _resume_schedule(n, ticks, seed):
    rng = np.random.default_rng(seed)
    ops, key = [], 100
    for i, x in enumerate(rng.choice(n, 6, replace=False).tolist()):
        ops.append((1 + i, _ffi.OP_CRASH, x, 0, 0))
        if i % 3 != 2:
            ops.append((40 + 5 * i, _ffi.OP_REVIVE, x, 0, 0))
    for t in range(3, ticks - 25, 4):
        key += 1
        ops.append((t, _ffi.OP_USER_EVENT, int(rng.integers(0, n)), key, 40))
    ops.sort(key=lambda o: o[0])
    return ops


RC_KW = dict(reconnect_interval=3, suspicion_mult=3, suspicion_max_mult=2)


@pytest.mark.parametrize("seed,n,fanout,loss,pi,ppi", [(51, 48, 3, 0.02, 2, 0), (52, 64, 4, 0.0, 2, 16)])
def test_oracle_matches_the_third_model_with_the_reconnector(seed, n, fanout, loss, pi, ppi):
    kw = dict(KW, fanout=fanout, loss=loss, probe_interval=pi, push_pull_interval=ppi, **RC_KW)
    sim = _ffi.Sim(load_oracle(), _ffi.make_config(n, **kw))
    made = [0]
    orig = tms.SwimNode.reconnect

    def counting(self):
        tgt = orig(self)
        made[0] += tgt is not None
        return tgt

    tms.SwimNode.reconnect = counting
    try:
        run(sim, n, _resume_schedule(n, 120, seed), 120, True, **kw)
    finally:
        tms.SwimNode.reconnect = orig
    assert made[0] > 3, "the run must contain reconnect attempts"


@pytest.mark.gpu
def test_hip_matches_the_third_model_with_the_reconnector(hiplib):
    seed, n, fanout, loss, pi, ppi = 51, 48, 3, 0.02, 2, 0
    kw = dict(KW, fanout=fanout, loss=loss, probe_interval=pi, push_pull_interval=ppi, **RC_KW)
    sim = _ffi.Sim(hiplib, _ffi.make_config(n, **kw))
    run(sim, n, _resume_schedule(n, 120, seed), 120, True, **kw)


def _tags_schedule(n, ticks, seed):
    """Serf::set_tags calls next to the usual churn: a node's tags change while it is alive, while it is suspected, right before it crashes"""
    ops = _schedule(n, ticks, seed)
    rng = np.random.default_rng(seed + 7)
    for t in range(4, ticks - 30, 6):
        ops.append((t, _ffi.OP_SET_TAGS, int(rng.integers(0, n)), int(rng.integers(0, 4)), 0))
    ops.sort(key=lambda o: o[0])
    return ops


@pytest.mark.parametrize("seed,n,fanout,loss,pi", [(81, 48, 3, 0.03, 2), (82, 64, 4, 0.0, 3)])
def test_oracle_matches_the_third_model_with_set_tags(seed, n, fanout, loss, pi):
    kw = dict(KW, fanout=fanout, loss=loss, probe_interval=pi)
    sim = _ffi.Sim(load_oracle(), _ffi.make_config(n, **kw))
    for w in range(0, n, 5):
        sim.watch(w)
    run(sim, n, _tags_schedule(n, 110, seed), 110, True, **kw)
    assert any(e[2] == _ffi.EV_UPDATE for e in sim.drain_events()), "somebody must have seen a member update"


@pytest.mark.gpu
def test_hip_matches_the_third_model_with_set_tags(hiplib):
    seed, n, fanout, loss, pi = 81, 48, 3, 0.03, 2
    kw = dict(KW, fanout=fanout, loss=loss, probe_interval=pi)
    sim = _ffi.Sim(hiplib, _ffi.make_config(n, **kw))
    run(sim, n, _tags_schedule(n, 110, seed), 110, True, **kw)


# THE BENCHMARK'S configuration tuple (bench.workload: fan-out 4, memberlist's kRandomNodes, probe interval 5, push-pull 150 x the log2 scaling,
# Reaper every 75, QueueChecker every 150, packets of 4 records, every other knob at its default) and its own schedule generator (the same mix, evenly
# spaced), on clusters the third model can follow: 64 and 100 nodes, 220 ticks.  (Dense views and rings of 64 / 32: the model is unbounded.)
@pytest.mark.parametrize("n,rate", [(64, 0.25), (100, 0.3)])
def test_oracle_matches_the_third_model_on_the_benchmarks_configuration(n, rate):
    from serf_amd import workload as wl
    kw = dict(fanout=4, view_slots=0, event_ring=RING_EV, query_ring=RING_Q, probe_interval=5, push_pull_interval=150, reap_interval=75, queue_check_interval=150,
              flags=_ffi.CF_BASELINE_JOINED | _ffi.CF_RANDOM_FANOUT)
    ops = wl.schedule(n, 190, rate=rate, seed=3, mix=wl.BENCH_MIX, max_member_subjects=n // 2, even=True)
    sim = _ffi.Sim(load_oracle(), _ffi.make_config(n, **kw))
    kinds, model = run(sim, n, ops, 220, True, **dict(kw, pkt_records=4))
    assert any(tm.push_pull_pairs(_ffi.DEFAULT_SEED, t, n, 150) for t in range(1, 220))


@pytest.mark.gpu
def test_hip_matches_the_third_model_on_the_benchmarks_configuration(hiplib):
    from serf_amd import workload as wl
    n = 64
    kw = dict(fanout=4, view_slots=0, event_ring=RING_EV, query_ring=RING_Q, probe_interval=5, push_pull_interval=150, reap_interval=75, queue_check_interval=150,
              flags=_ffi.CF_BASELINE_JOINED | _ffi.CF_RANDOM_FANOUT)
    ops = wl.schedule(n, 190, rate=0.25, seed=3, mix=wl.BENCH_MIX, max_member_subjects=n // 2, even=True)
    sim = _ffi.Sim(hiplib, _ffi.make_config(n, **kw))
    run(sim, n, ops, 220, True, **dict(kw, pkt_records=4))


def _rejoin_schedule(n, ticks, seed):
    """crash / re-join cycles with a named peer (what the configs[4] runs do): the joiner adopts the peer's view when SIM_CF_JOIN_SYNC is set"""
    rng = np.random.default_rng(seed)
    ops, key = [], 100
    for i, x in enumerate(rng.choice(n, 8, replace=False).tolist()):
        t0 = 2 + 5 * i
        ops.append((t0, _ffi.OP_CRASH, x, 0, 0))
        ops.append((t0 + int(rng.integers(18, 45)), _ffi.OP_JOIN, x, int(rng.integers(0, n)), 0))
    for t in range(3, ticks - 25, 4):
        key += 1
        ops.append((t, _ffi.OP_USER_EVENT, int(rng.integers(0, n)), key, 40))
    ops.sort(key=lambda o: o[0])
    return ops


@pytest.mark.parametrize("seed,n,fanout,loss,pi", [(91, 48, 3, 0.03, 2), (92, 64, 4, 0.05, 1)])
def test_oracle_matches_the_third_model_with_join_sync(seed, n, fanout, loss, pi):
    kw = dict(KW, fanout=fanout, loss=loss, probe_interval=pi, join_sync=True, tcp_fallback=True)
    sim = _ffi.Sim(load_oracle(), _ffi.make_config(n, **kw))
    run(sim, n, _rejoin_schedule(n, 110, seed), 110, True, **kw)


@pytest.mark.gpu
def test_hip_matches_the_third_model_with_join_sync(hiplib):
    seed, n, fanout, loss, pi = 91, 48, 3, 0.03, 2
    kw = dict(KW, fanout=fanout, loss=loss, probe_interval=pi, join_sync=True, tcp_fallback=True)
    sim = _ffi.Sim(hiplib, _ffi.make_config(n, **kw))
    run(sim, n, _rejoin_schedule(n, 110, seed), 110, True, **kw)


# a sweep over COMBINATIONS: every knob drawn at random per case — cluster size, fan-out, loss, probe interval, packet size, push-pull, Reaper with
# short timeouts, QueueChecker, Reconnector, the four memberlist switches, join sync — over a schedule with crashes, silent resumes, re-joins with a
# peer, graceful leaves, tag changes, events and queries.  (A case whose load hits the 16-slot queue is skipped: the model has no bounds.)
def _sweep_case(i):
    rng = np.random.default_rng(9000 + i)
    n = int(rng.choice([24, 33, 40, 48, 64, 80]))
    kw = dict(KW, fanout=int(rng.integers(2, 5)), loss=float(rng.choice([0.0, 0.02, 0.05])), probe_interval=int(rng.integers(1, 4)),
              pkt_records=int(rng.choice([4, 8, 16, 16])), push_pull_interval=int(rng.choice([0, 8, 16])),
              awareness_probe=bool(rng.integers(0, 2)), tcp_fallback=bool(rng.integers(0, 2)), nacks=bool(rng.integers(0, 2)),
              gossip_to_the_dead=int(rng.choice([0, 0, 8])), join_sync=bool(rng.integers(0, 2)), reconnect_interval=int(rng.choice([0, 3])))
    if rng.integers(0, 2):
        kw.update(REAP_KW)
    if rng.integers(0, 2):
        kw.update(queue_check_interval=5, max_queue_depth=3, min_queue_depth=0)
    ops, key = [], 300
    busy = {}
    for t in range(2, 80):
        if rng.random() < (0.25 if kw["pkt_records"] == 4 else 0.4):
            node = int(rng.integers(0, n))
            if busy.get(node, 0) > t:
                continue
            r, key = rng.random(), key + 1
            if r < 0.35:
                ops.append((t, _ffi.OP_USER_EVENT, node, key, 40))
            elif r < 0.5:
                ops.append((t, _ffi.OP_QUERY, node, key, int(rng.choice([0, _ffi.F_ACK, _ffi.F_ACK | _ffi.F_RESPOND | (2 << 8)]))))
            elif r < 0.6:
                ops.append((t, _ffi.OP_SET_TAGS, node, int(rng.integers(0, 4)), 0))
            elif r < 0.8:
                back = t + int(rng.integers(10, 35))
                ops.append((t, _ffi.OP_CRASH, node, 0, 0))
                ops.append((back, _ffi.OP_JOIN, node, int(rng.integers(0, n)), 0) if rng.random() < 0.6 else (back, _ffi.OP_REVIVE, node, 0, 0))
                busy[node] = back + 8
            else:
                ops += [(t, _ffi.OP_LEAVE, node, 0, 0), (t + 4, _ffi.OP_LEAVE_FINISH, node, 0, 0), (t + 8, _ffi.OP_CRASH, node, 0, 0)]
                busy[node] = 10 ** 6
    ops.sort(key=lambda o: o[0])
    return n, kw, ops


def _run_sweep_case(lib, i):
    n, kw, ops = _sweep_case(i)
    sim = _ffi.Sim(lib, _ffi.make_config(n, **kw))
    try:
        run(sim, n, ops, 110, True, **kw)
    except AssertionError as e:
        if "a model bound was hit" in str(e):
            pytest.skip(f"case {i}: the load hit the bounded queue ({e})")
        raise


@pytest.mark.parametrize("i", range(16))
def test_oracle_matches_the_third_model_over_random_combinations(i):
    _run_sweep_case(load_oracle(), i)


@pytest.mark.gpu
@pytest.mark.parametrize("i", [1, 4, 7, 10])
def test_hip_matches_the_third_model_over_random_combinations(hiplib, i):
    _run_sweep_case(hiplib, i)


def _light_schedule(n, ticks, seed):
    """a load the 16-slot queue carries with packets of 4 records: a rumour every few ticks, two crashes (one re-joins), one graceful leave"""
    rng = np.random.default_rng(seed)
    ops, key = [], 100
    for t in range(2, ticks - 30, 5):
        key += 1
        ops.append((t, _ffi.OP_USER_EVENT if key % 3 else _ffi.OP_QUERY, int(rng.integers(0, n)), key, 40 if key % 3 else 0))
    a, b, c = (int(x) for x in rng.choice(n, 3, replace=False))
    ops += [(8, _ffi.OP_CRASH, a, 0, 0), (45, _ffi.OP_JOIN, a, 0, 0), (22, _ffi.OP_CRASH, b, 0, 0),
            (30, _ffi.OP_LEAVE, c, 0, 0), (34, _ffi.OP_LEAVE_FINISH, c, 0, 0), (38, _ffi.OP_CRASH, c, 0, 0)]
    ops.sort(key=lambda o: o[0])
    return ops


# packets of 4 and 8 records (the benchmark's are 4): a queue that holds more than a packet carries — get_broadcasts' walk takes what drains first and
# what still fits, the rest waits its turn with fewer transmits than the records that went out
@pytest.mark.parametrize("seed,n,fanout,loss,pi,pk", [(71, 48, 3, 0.03, 2, 4), (72, 64, 4, 0.0, 3, 8), (73, 40, 2, 0.05, 2, 4)])
def test_oracle_matches_the_third_model_with_small_packets(seed, n, fanout, loss, pi, pk):
    kw = dict(KW, fanout=fanout, loss=loss, probe_interval=pi, pkt_records=pk)
    sim = _ffi.Sim(load_oracle(), _ffi.make_config(n, **kw))
    waited = [0]
    orig = tms.SwimNode.get_broadcasts

    def counting(self):
        out = orig(self)
        waited[0] += len(out) == self.par.P and len(self.queue) > 0
        return out

    tms.SwimNode.get_broadcasts = counting
    try:
        run(sim, n, _light_schedule(n, 110, seed), 110, True, **kw)
    finally:
        tms.SwimNode.get_broadcasts = orig
    assert pk > 4 or waited[0] > 10, "packets of 4 records must have gone out full with records left waiting"


@pytest.mark.gpu
def test_hip_matches_the_third_model_with_small_packets(hiplib):
    seed, n, fanout, loss, pi, pk = 71, 48, 3, 0.03, 2, 4
    kw = dict(KW, fanout=fanout, loss=loss, probe_interval=pi, pkt_records=pk)
    sim = _ffi.Sim(hiplib, _ffi.make_config(n, **kw))
    run(sim, n, _light_schedule(n, 110, seed), 110, True, **kw)


# memberlist's behaviours behind switches: awareness-scaled probing, the stream-transport fallback ping, nacks, gossip_to_the_dead_time
SWITCH_CASES = [(61, 48, 3, 0.05, 2, dict(awareness_probe=True)), (62, 48, 3, 0.06, 2, dict(nacks=True)), (63, 64, 4, 0.1, 2, dict(tcp_fallback=True, nacks=True)),
                (64, 48, 3, 0.03, 2, dict(gossip_to_the_dead=6, **RC_KW)), (65, 40, 2, 0.05, 2, dict(awareness_probe=True, tcp_fallback=True, nacks=True, gossip_to_the_dead=10))]


@pytest.mark.parametrize("seed,n,fanout,loss,pi,sw", SWITCH_CASES)
def test_oracle_matches_the_third_model_with_the_memberlist_switches(seed, n, fanout, loss, pi, sw):
    kw = dict(KW, fanout=fanout, loss=loss, probe_interval=pi, **sw)
    sim = _ffi.Sim(load_oracle(), _ffi.make_config(n, **kw))
    run(sim, n, _resume_schedule(n, 110, seed) if "gossip_to_the_dead" in sw else _schedule(n, 110, seed), 110, True, **kw)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n,fanout,loss,pi,sw", [SWITCH_CASES[2], SWITCH_CASES[4]])
def test_hip_matches_the_third_model_with_the_memberlist_switches(hiplib, seed, n, fanout, loss, pi, sw):
    kw = dict(KW, fanout=fanout, loss=loss, probe_interval=pi, **sw)
    sim = _ffi.Sim(hiplib, _ffi.make_config(n, **kw))
    run(sim, n, _resume_schedule(n, 110, seed) if "gossip_to_the_dead" in sw else _schedule(n, 110, seed), 110, True, **kw)


# with the QueueChecker (base.rs:683-740) at a depth that bites: a queue of serf's that holds more than two messages is pruned to the two that drain first
QC_KW = dict(queue_check_interval=3, max_queue_depth=2, min_queue_depth=0)


@pytest.mark.parametrize("seed,n,fanout,loss,pi,ppi", [(41, 48, 3, 0.02, 2, 0), (42, 64, 2, 0.0, 3, 0)])
def test_oracle_matches_the_third_model_with_the_queue_checker(seed, n, fanout, loss, pi, ppi):
    kw = dict(KW, fanout=fanout, loss=loss, probe_interval=pi, push_pull_interval=ppi, **QC_KW, **REAP_KW)
    sim = _ffi.Sim(load_oracle(), _ffi.make_config(n, **kw))
    pruned = [0]
    orig = tms.SwimNode.queue_check

    def counting(self):
        before = len(self.queue)
        orig(self)
        pruned[0] += before - len(self.queue)

    tms.SwimNode.queue_check = counting
    try:
        run(sim, n, _burst_schedule(n, 110, seed), 110, True, **kw)
    finally:
        tms.SwimNode.queue_check = orig
    assert pruned[0] > 0, "the checker must have pruned something"


@pytest.mark.gpu
def test_hip_matches_the_third_model_with_the_queue_checker(hiplib):
    seed, n, fanout, loss, pi = 41, 48, 3, 0.02, 2
    kw = dict(KW, fanout=fanout, loss=loss, probe_interval=pi, **QC_KW, **REAP_KW)
    sim = _ffi.Sim(hiplib, _ffi.make_config(n, **kw))
    run(sim, n, _burst_schedule(n, 110, seed), 110, True, **kw)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n,fanout,loss,pi,ppi", REAP_CASES[:2])
def test_hip_matches_the_third_model_with_the_reaper(hiplib, seed, n, fanout, loss, pi, ppi):
    kw = dict(KW, fanout=fanout, loss=loss, probe_interval=pi, push_pull_interval=ppi, **REAP_KW)
    sim = _ffi.Sim(hiplib, _ffi.make_config(n, **kw))
    run(sim, n, _schedule(n, 110, seed), 110, True, **kw)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n,fanout,loss,pi,ppi", PP_CASES[:2])
def test_hip_matches_the_third_model_with_push_pull(hiplib, seed, n, fanout, loss, pi, ppi):
    kw = dict(KW, fanout=fanout, loss=loss, probe_interval=pi, push_pull_interval=ppi)
    sim = _ffi.Sim(hiplib, _ffi.make_config(n, **kw))
    run(sim, n, _schedule(n, 110, seed), 110, True, **kw)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n,fanout,loss,pi", CASES[:3])
def test_hip_matches_the_third_model_with_the_memberlist_layer_on(hiplib, seed, n, fanout, loss, pi):
    kw = dict(KW, fanout=fanout, loss=loss, probe_interval=pi)
    sim = _ffi.Sim(hiplib, _ffi.make_config(n, **kw))
    run(sim, n, _schedule(n, 110, seed), 110, True, **kw)
