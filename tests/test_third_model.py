"""The oracle (CPU) and the HIP library (GPU) against a THIRD model of the serf handlers (tests/third_model.py: pure Python,
dictionary state, unbounded, written from the Rust sources alone).  Oracle and kernel have one author and one reading of
base.rs; their 4-node known-answer tests pin single handlers.  Here whole runs are compared: the third model is fed the very
packets the implementation under test delivers — its canonical inbox, dumped before every tick — and the API operations of
the schedule, and after every tick every node's clocks, member table, buffered intents, both de-dup rings and every
rebroadcast decision must be the same ("two restatements by one author").  Model bounds never bite in these runs
(overflow == 0 is asserted): a bounded run without drops is the unbounded protocol (tests/test_oracle_unbounded.py)."""
import numpy as np
import pytest

from serf_amd import _ffi
from tests import third_model as tm
from tests._oracle import load_oracle
from tests import _scenario as sc

RING_EV, RING_Q, PG = 64, 32, 4   # packets of 4 pages = 16 records: they carry a node's whole queue, nothing waits for a turn


def run_against_third_model(sim, n, fanout, ops, ticks, joined, rf=False, pp_interval=0, loss=0.0, X=0):
    nodes = [tm.Node(i, n, RING_EV, RING_Q, joined) for i in range(n)]
    # the query path end to end: the origin's trackers and the responders' acks / responses, relays and loss included
    trackers = tm.QueryTrackers(_ffi.DEFAULT_SEED, n, loss)
    now = [0]
    for x in nodes:
        x.on_query = lambda me, qid, flags: trackers.respond(me, qid, flags, now[0], [y.up for y in nodes])
    counted = 0
    by_tick = {}
    for o in ops:
        by_tick.setdefault(o[0], []).append(o)
        sim.inject(*o)
    approved = [set() for _ in range(n)]   # every (kind, key, ltime) a node's delegate has asked to (re)broadcast so far
    for t in range(ticks):
        inbox = sim.dump(_ffi.ARR_INBOX).reshape(fanout * PG, n)   # the packets about to be delivered: [slot * PG + page][receiver]
        if rf and t > 0:
            # memberlist's kRandomNodes: the dump is SENDER-indexed in this mode; who receives what is drawn here, from the
            # specification (third_model.k_random_nodes), and handed over in (sender, slot) order
            rows = [[] for _ in range(n)]
            for snd in range(n):
                for k, tgt in enumerate(tm.k_random_nodes(_ffi.DEFAULT_SEED, t - 1, snd, n, fanout)):
                    rows[tgt].append((snd, k))
        # (0) the tick's operations, in call order (SIMSPEC §2.1)
        now[0] = t
        for _, op, node, a, b in by_tick.get(t, ()):
            x = nodes[node]
            if op == _ffi.OP_QUERY:                         # base.rs:905-930: the QueryResponse is registered before the query is handled / sent
                trackers.register(a, node, b, t)
            if op == _ffi.OP_CRASH:
                x.up = False
            elif op == _ffi.OP_REVIVE:
                x.up = True
            elif op == _ffi.OP_JOIN:
                x.join()
            elif op == _ffi.OP_LEAVE_FINISH:
                if x.up and x.state == tm.S_LEAVING:
                    x.state = tm.S_LEFT
            elif not x.up:
                pass
            elif op == _ffi.OP_USER_EVENT:
                x.user_event(a)
            elif op == _ffi.OP_QUERY:
                x.query(a, b & 15)
            elif op == _ffi.OP_LEAVE:
                x.leave()
            elif op == _ffi.OP_FORCE_LEAVE:
                x.force_leave(a, bool(b))
        # (0b) the tick's push-pull batch (SIMSPEC §2.10): both processes must be running; `a` merges first, then `b` merges a's updated state
        for a, b in tm.push_pull_pairs(_ffi.DEFAULT_SEED, t, n, pp_interval):
            if nodes[a].up and nodes[b].up:
                nodes[a].merge_remote_state(nodes[b].local_state())
                nodes[b].merge_remote_state(nodes[a].local_state())
        # (1) deliveries: slot 0 first, records in packet order
        for i, x in enumerate(nodes):
            if not x.up:
                continue
            cells = [inbox[k * PG + pg, snd] for snd, k in rows[i] for pg in range(PG)] if rf and t > 0 else \
                    [] if rf else [inbox[k, i] for k in range(fanout * PG)]
            for pk in cells:
                for r in range(4):
                    hm = int(pk["hi_meta"][r])
                    kind = (hm >> 4) & 15
                    if kind == 0:
                        continue
                    assert kind <= 4, "the SWIM layer is off in these runs"
                    x.notify(kind, int(pk["key"][r]), int(pk["val_lo"][r]) | ((hm >> 16) << 32), hm & 15)
        sim.step(1)
        # ---- compare
        rows = sim.dump(_ffi.ARR_ROWS)
        view = sim.dump(_ffi.ARR_VIEW).reshape(n, n)        # dense view: [subject][observer]
        er = sim.dump(_ffi.ARR_ERING).reshape(X + RING_EV, n)   # (X = sim_config.ring_overflow: the rings' overflow rows come first)
        qr = sim.dump(_ffi.ARR_QRING).reshape(X + RING_Q, n)
        queue = sim.dump(_ffi.ARR_QUEUE).reshape(n, _ffi.Q)
        assert int(rows["overflow"].sum()) == 0, f"tick {t}: a model bound was hit"
        for qid in trackers.running:                        # what every running query's origin has counted so far, and whether it still listens
            got = sim.query_status(qid)
            assert (got[0], got[1], bool(got[2])) == trackers.status(qid, t + 1), f"tick {t} query {qid}: {got} != {trackers.status(qid, t + 1)}"
            counted = max(counted, got[0] + got[1])

        for i, x in enumerate(nodes):
            w = f"tick {t} node {i}"
            assert (int(rows["clock"][i]), int(rows["event_clock"][i]), int(rows["query_clock"][i])) == \
                   (x.clock.time(), x.event_clock.time(), x.query_clock.time()), w
            assert bool(rows["flags"][i] & 1) == x.up and ((int(rows["flags"][i]) >> 1) & 3) == x.state, w
            assert int(rows["n_known"][i]) == len(x.members), w
            for s in range(n):
                e = view[s, i]
                bits = int(e["bits"])
                if s in x.members:
                    assert bits & 1 and ((bits >> 1) & 7, int(e["ltime"])) == tuple(x.members[s]), f"{w} subject {s}"
                else:
                    it = x.intents.get(s)
                    assert not (bits & 1), f"{w} subject {s}"
                    assert ((bits >> 6) & 3, int(e["ltime"])) == ((it[0], it[1]) if it else (0, 0)), f"{w} subject {s} intent"
            for ring, buf, name in ((er, x.event_buf, "event"), (qr, x.query_buf, "query")):
                for j, want in enumerate(buf):
                    b = ring[X + j, i]
                    got = [int(v) for v in b["keys"] if v]
                    if len(got) == len(b["keys"]):      # a full bucket continues in its overflow rows, in row order (include/serf_sim.h sim_bucket)
                        for o in ring[:X, i]:
                            if o["ltime"] == 0:
                                break
                            if o["ltime"] == j + 1:
                                got += [int(v) for v in o["keys"] if v]
                    if want is None:
                        assert not got, f"{w} {name} bucket {j}"
                    else:
                        assert (int(b["ltime"]), got) == (want[0], want[1]), f"{w} {name} bucket {j}"
            # every rebroadcast the delegate asked for this tick is in the node's queue (nothing expires within a tick:
            # retransmit limit 8 > fanout); everything in the queue was asked for at some point
            approved[i].update(x.rebroadcast)
            live = {(int((r["meta"] >> 4) & 15), int(r["key"]), int(r["val"])) for r in queue[i] if r["meta"] != 0xFFFFFFFF}
            if x.up:
                assert set(x.rebroadcast) <= live, f"{w}: asked for {set(x.rebroadcast) - live} and it is not queued"
            assert live <= approved[i], f"{w}: queued without the delegate's say: {live - approved[i]}"
            x.rebroadcast = []
    return counted


def _schedule(n, ticks, seed, joined):
    ops = sc.schedule(n, ticks - 20, rate=0.6, seed=seed, mix=(0.45, 0.2, 0.15, 0.1, 0.1), max_member_subjects=n // 3)
    if not joined:  # nobody knows anybody (and with the SWIM layer off nobody ever will): every intent about another node is buffered
        ops = [o for o in ops if o[1] != _ffi.OP_FORCE_LEAVE or True]
    return ops


@pytest.mark.parametrize("seed,n,fanout,joined,rf", [(1, 48, 3, True, False), (2, 64, 4, True, False), (3, 33, 2, True, False), (4, 48, 3, False, False), (5, 20, 3, False, False),
                                                     (6, 48, 3, True, True), (7, 64, 4, True, True), (8, 24, 3, False, True)])   # (not joined: every node knows one member, the retransmit limit is 4 — fan-out 3 keeps a record queued past its tick)
def test_oracle_matches_the_third_model(seed, n, fanout, joined, rf):
    # rf: memberlist's kRandomNodes — who receives which packet is drawn by the third model itself, from the specification
    kw = dict(fanout=fanout, view_slots=0, event_ring=RING_EV, query_ring=RING_Q, leave_delay=4, pkt_records=4 * PG,
              flags=(_ffi.CF_BASELINE_JOINED if joined else 0) | (_ffi.CF_RANDOM_FANOUT if rf else 0))
    sim = _ffi.Sim(load_oracle(), _ffi.make_config(n, **kw))
    run_against_third_model(sim, n, fanout, _schedule(n, 70, seed, joined), 70, joined, rf)


# push-pull anti-entropy (delegate.rs:386-554) in the third model: the batches' pairs drawn from the specification, local_state /
# merge_remote_state written from the Rust source; packet loss on, so that there is something for a push-pull to repair
PP_CASES = [(11, 48, 3, True, False, 12), (12, 64, 4, True, True, 8), (13, 33, 2, True, False, 16), (14, 40, 3, False, False, 8)]


def _pp_kw(fanout, joined, rf, ppi):
    return dict(fanout=fanout, view_slots=0, event_ring=RING_EV, query_ring=RING_Q, leave_delay=4, pkt_records=4 * PG, push_pull_interval=ppi, loss=0.15,
                flags=(_ffi.CF_BASELINE_JOINED if joined else 0) | (_ffi.CF_RANDOM_FANOUT if rf else 0))


@pytest.mark.parametrize("seed,n,fanout,joined,rf,ppi", PP_CASES)
def test_oracle_matches_the_third_model_with_push_pull(seed, n, fanout, joined, rf, ppi):
    sim = _ffi.Sim(load_oracle(), _ffi.make_config(n, **_pp_kw(fanout, joined, rf, ppi)))
    assert any(tm.push_pull_pairs(_ffi.DEFAULT_SEED, t, n, ppi) for t in range(1, 70)), "the run must contain push-pull batches"
    counted = run_against_third_model(sim, n, fanout, _schedule(n, 70, seed, joined), 70, joined, rf, pp_interval=ppi, loss=0.15)
    assert counted > n // 2, "the runs must contain queries whose acks / responses reach the origin"


def _sweep(i):
    """a random combination: size, fan-out, pre-joined or not, fan-out model, push-pull interval, loss (a cluster nobody has joined keeps its
    retransmit limit at 4: a fan-out below that keeps a record queued past its first tick, which the rebroadcast check relies on)"""
    rng = np.random.default_rng(5000 + i)
    n = int(rng.choice([20, 33, 48, 64, 90]))
    joined = bool(rng.random() < 0.8)
    fanout = int(rng.integers(2, 5)) if joined else int(rng.integers(2, 4))
    rf, ppi, loss = bool(rng.integers(0, 2)), int(rng.choice([0, 8, 16])), float(rng.choice([0.0, 0.1, 0.2]))
    return n, fanout, joined, rf, ppi, loss


@pytest.mark.parametrize("i", range(14))
def test_oracle_matches_the_third_model_over_random_combinations(i):
    n, fanout, joined, rf, ppi, loss = _sweep(i)
    sim = _ffi.Sim(load_oracle(), _ffi.make_config(n, **dict(_pp_kw(fanout, joined, rf, ppi), loss=loss)))
    run_against_third_model(sim, n, fanout, _schedule(n, 70, 100 + i, joined), 70, joined, rf, pp_interval=ppi, loss=loss)


def _origin_goes_down_schedule(n):
    """queries whose origin crashes while the query is still spreading (what comes back after that is dropped: the origin is not
    running), one of them back up before the spread is over, and one query issued by a node that is down (registered, never sent)"""
    ops = []
    for i, (origin, t0) in enumerate(((5, 2), (9, 6), (17, 11), (21, 15))):
        ops.append((t0, _ffi.OP_QUERY, origin, 700 + i, _ffi.F_ACK | _ffi.F_RESPOND | ((i % 3) << 8)))
        ops.append((t0 + 1 + i % 2, _ffi.OP_CRASH, origin, 0, 0))
        if i == 1:
            ops.append((t0 + 3, _ffi.OP_REVIVE, origin, 0, 0))
    ops.append((20, _ffi.OP_QUERY, 5, 750, _ffi.F_ACK))          # node 5 is down
    ops.append((22, _ffi.OP_USER_EVENT, 30, 760, 40))
    ops.sort(key=lambda o: o[0])
    return ops


def test_oracle_matches_the_third_model_when_a_query_origin_goes_down():
    n, fanout = 64, 2
    kw = dict(_pp_kw(fanout, True, False, 0), loss=0.1)
    sim = _ffi.Sim(load_oracle(), _ffi.make_config(n, **kw))
    counted = run_against_third_model(sim, n, fanout, _origin_goes_down_schedule(n), 45, True, False, loss=0.1)
    assert counted > 0


@pytest.mark.gpu
def test_hip_matches_the_third_model_when_a_query_origin_goes_down(hiplib):
    n, fanout = 64, 2
    sim = _ffi.Sim(hiplib, _ffi.make_config(n, **dict(_pp_kw(fanout, True, False, 0), loss=0.1)))
    run_against_third_model(sim, n, fanout, _origin_goes_down_schedule(n), 45, True, False, loss=0.1)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n,fanout,joined,rf,ppi", PP_CASES[:3])
def test_hip_matches_the_third_model_with_push_pull(hiplib, seed, n, fanout, joined, rf, ppi):
    sim = _ffi.Sim(hiplib, _ffi.make_config(n, **_pp_kw(fanout, joined, rf, ppi)))
    run_against_third_model(sim, n, fanout, _schedule(n, 70, seed, joined), 70, joined, rf, pp_interval=ppi, loss=0.15)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n,fanout,joined,rf", [(1, 48, 3, True, False), (2, 64, 4, True, False), (4, 48, 3, False, False), (6, 48, 3, True, True), (7, 64, 4, True, True)])
def test_hip_matches_the_third_model(hiplib, seed, n, fanout, joined, rf):
    kw = dict(fanout=fanout, view_slots=0, event_ring=RING_EV, query_ring=RING_Q, leave_delay=4, pkt_records=4 * PG,
              flags=(_ffi.CF_BASELINE_JOINED if joined else 0) | (_ffi.CF_RANDOM_FANOUT if rf else 0))
    sim = _ffi.Sim(hiplib, _ffi.make_config(n, **kw))
    run_against_third_model(sim, n, fanout, _schedule(n, 70, seed, joined), 70, joined, rf)


@pytest.mark.parametrize("seed,n,fanout,rf", [(31, 48, 3, False), (32, 64, 4, True)])
def test_beyond_the_old_bounds_against_the_third_model(seed, n, fanout, rf):
    # (r6) the third model's Vec-per-bucket and unbounded queues against the oracle's overflow rows and 64-slot queue: a burst of user
    # events and queries of ONE Lamport time (20 + 12 of them in one tick: one bucket each, 6 keys in place, the rest in overflow rows)
    # on top of a load that takes queues past the old bound of 16
    X = 4
    kw = dict(fanout=fanout, view_slots=0, event_ring=RING_EV, query_ring=RING_Q, leave_delay=4, pkt_records=4 * PG, ring_overflow=X,
              flags=_ffi.CF_BASELINE_JOINED | (_ffi.CF_RANDOM_FANOUT if rf else 0))
    sim = _ffi.Sim(load_oracle(), _ffi.make_config(n, **kw))
    ops = sc.schedule(n, 40, rate=1.2, seed=seed, mix=(0.6, 0.25, 0.15, 0.0, 0.0), max_member_subjects=n // 3)
    ops += [(5, _ffi.OP_USER_EVENT, (7 * i + 1) % n, 9000 + i, 32 + i) for i in range(20)]
    ops += [(9, _ffi.OP_QUERY, (5 * i + 2) % n, 10340 + i, _ffi.F_ACK) for i in range(12)]
    ops.sort(key=lambda o: o[0])
    run_against_third_model(sim, n, fanout, ops, 60, True, rf, X=X)
    er = sim.dump(_ffi.ARR_ERING).reshape(X + RING_EV, n)
    assert (er[:X]["ltime"] != 0).any(), "the burst was meant to spill into the overflow rows"
