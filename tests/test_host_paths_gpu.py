"""SURVEY.md §8f.3 / §8f.4 over the PRODUCT: the reference-format per-node snapshot (serf_amd/snapshot.py <- snapshot.rs),
the event coalescers (serf_amd/coalesce.py <- coalesce/*.rs) and the codec's message lengths (serf_amd/wire.py <-
types/*.rs) fed from what the HIP library itself drains and queues — the same scenarios the CPU tests run over the oracle
(tests/test_snapshot_format.py, tests/test_coalesce.py, tests/test_wire.py, tests/test_byte_budget.py), with the oracle
run beside the GPU as the checker."""
import pytest

from serf_amd import _ffi, snapshot as snap, wire
from serf_amd.coalesce import MemberEventCoalescer, UserEventCoalescer, coalesce_loop
from tests import _scenario as sc

pytestmark = pytest.mark.gpu


def both(oracle, hiplib, n, **kw):
    return _ffi.Sim(hiplib, _ffi.make_config(n, **kw)), _ffi.Sim(oracle, _ffi.make_config(n, **kw))


def test_snapshot_of_a_node_simulated_on_the_gpu(oracle, hiplib):
    # tests/test_snapshot_format.py::test_snapshot_of_a_simulated_node, events drained from the HIP library
    n, obs = 128, 5
    g, o = both(oracle, hiplib, n, fanout=3, view_slots=0, probe_interval=2, suspicion_mult=3, suspicion_max_mult=2, leave_delay=4)
    for s in (g, o):
        s.watch(obs)
        s.inject(2, _ffi.OP_CRASH, 40)
        s.inject(3, _ffi.OP_USER_EVENT, 9, 0xAB, 40)
        s.inject(5, _ffi.OP_QUERY, 11, 77, _ffi.F_ACK)
        s.inject(6, _ffi.OP_LEAVE, 60)
        s.inject(11, _ffi.OP_LEAVE_FINISH, 60)
        s.inject(16, _ffi.OP_CRASH, 60)
        s.inject(60, _ffi.OP_JOIN, 60)
        s.step(160)
    eg, eo = g.drain_events(), o.drain_events()
    assert eg == eo and len(eg) >= 5
    events = [e for e in eg if e[1] == obs]
    kinds = [e[2] for e in events]
    assert snap.EV_FAILED in kinds and snap.EV_LEAVE in kinds and snap.EV_JOIN in kinds and snap.EV_USER in kinds and snap.EV_QUERY in kinds
    sg, so = snap.snapshot_of(g, obs, events), snap.snapshot_of(o, obs, events)
    assert sg.bytes() == so.bytes(), "the file a node simulated on the GPU would write is the oracle's, byte for byte"
    r = snap.replay(sg.bytes())
    st = g.stats(obs)
    assert r.last_clock == st.member_time - 1
    assert r.last_event_clock == max(e[4] for e in events if e[2] == snap.EV_USER)
    assert r.last_query_clock == max(e[4] for e in events if e[2] == snap.EV_QUERY)
    assert 60 in r.alive_nodes and 40 not in r.alive_nodes
    status, _ = g.members(obs)
    assert status[60] == _ffi.STATUS_ALIVE and status[40] == _ffi.STATUS_FAILED
    c = snap.replay(sg.compact())
    assert (c.alive_nodes, c.last_clock, c.last_event_clock, c.last_query_clock) == (r.alive_nodes, r.last_clock, r.last_event_clock, r.last_query_clock)
    # leave (snapshot.rs:562-580): the record is written, the state forgotten, and a replay starts from nothing
    sg.leave()
    assert snap.replay(sg.bytes()).alive_nodes == set()


def test_coalescers_over_the_gpus_event_stream(oracle, hiplib):
    # tests/test_coalesce.py::test_coalescing_a_simulated_clusters_member_events + user events with the cc flag
    g, o = both(oracle, hiplib, 64, fanout=3, probe_interval=5, leave_delay=8)
    for s in (g, o):
        s.watch(3)
        s.inject(1, _ffi.OP_CRASH, 20)
        s.step(1)
        s.leave(30)
        for i, (key, cc) in enumerate([(0x10, True), (0x11, True), (0x10, True), (0x12, False)]):
            s.inject(40 + 30 * i, _ffi.OP_USER_EVENT, 7 + i, key, wire.user_event_len(1, b"deploy", b"v%d" % i, cc=cc) | (0x80000000 if cc else 0))
        s.step(400)
    raw = [e for e in g.drain_events() if e[1] == 3]
    assert raw == [e for e in o.drain_events() if e[1] == 3]
    members = [e for e in raw if e[2] in (_ffi.EV_LEAVE, _ffi.EV_FAILED)]
    assert sorted((e[2], e[3]) for e in members) == [(_ffi.EV_LEAVE, 30), (_ffi.EV_FAILED, 20)]
    out = coalesce_loop(members, MemberEventCoalescer(), coalesce_period=1000, quiescent_period=1000, end_tick=2000)
    assert {ty: m for _, _, ty, m in out} == {_ffi.EV_LEAVE: [30], _ffi.EV_FAILED: [20]}
    users = [e for e in raw if e[2] == _ffi.EV_USER]
    assert len(users) == 4
    # coalesce/user.rs: per event NAME only the newest Lamport time survives a window; an event that did not ask to be
    # coalesced passes straight through.  The simulator's event key stands for (name, payload): key = name here.
    uc = UserEventCoalescer(name_of=lambda key: key, is_cc=lambda ev: ev[3] != 0x12)
    out = coalesce_loop(users, uc, coalesce_period=1000, quiescent_period=1000, end_tick=2000)
    assert [e[3] for e in out if e[3] == 0x12] == [0x12] and out[0][3] == 0x12, "not coalesced: delivered at once"
    kept = [e for e in out if e[3] != 0x12]
    assert sorted(e[3] for e in kept) == [0x10, 0x11]
    assert [e[4] for e in kept if e[3] == 0x10] == [max(e[4] for e in users if e[3] == 0x10)]


def test_codec_lengths_ride_the_gpus_queue_and_byte_budget(oracle, hiplib):
    # tests/test_wire.py::test_user_event_through_the_simulator and tests/test_byte_budget.py on the HIP library: the
    # host prices the message with the codec, the record carries the length, the packet fills up by BYTES
    def transmits(sim, node):
        q = sim.dump(_ffi.ARR_QUEUE).reshape(sim.n, _ffi.Q)[node]
        return {int(r["key"]): (int(r["meta"]) >> 24) & 63 for r in q if r["meta"] != 0xFFFFFFFF}

    g, o = both(oracle, hiplib, 64, fanout=3, view_slots=8)
    name, payload = b"deploy", b"v1.2.3" * 20
    nbytes = wire.user_event_len(1, name, payload, cc=True)
    big = wire.user_event_len(1, b"deploy", b"x" * 506)       # the largest event the default limit allows: 528 bytes framed
    for s in (g, o):
        s.user_event(3, 0xBEEF, nbytes, coalesce=True)
        for key in (101, 102, 103):
            s.user_event(0, key, big)
        s.user_event(0, 104, 16)
        s.query(5, 77, _ffi.F_ACK)
        s.step(1)
    qg = g.dump(_ffi.ARR_QUEUE).reshape(64, _ffi.Q)
    assert (qg == o.dump(_ffi.ARR_QUEUE).reshape(64, _ffi.Q)).all()
    meta = int(qg[3]["meta"][0])
    assert 63 - ((meta >> 18) & 63) == (nbytes + 15) // 16 and (meta >> 4) & 15 == _ffi.K_EVENT and meta & 1
    qm = int(qg[5]["meta"][0])
    ref_q = wire.Query(5000, 77, 999999, flags=_ffi.F_ACK, relay_factor=0, timeout_ms=16 * 7 * 200)
    assert 63 - ((qm >> 18) & 63) == (wire.encoded_len(ref_q) + 15) // 16 == 3
    # three 528-byte events do not share a 1 400-byte packet: 103 + 102 go, 101 is skipped, 104 still fits ...
    assert transmits(g, 0) == transmits(o, 0) == {101: 2, 102: 2, 103: 2, 104: 3}
    g.step(30)
    o.step(30)
    assert g.digest() == o.digest()


def test_byte_boundary_of_the_delegate_on_the_gpu(oracle, hiplib):
    # tests/test_bridge.py on the HIP library: deliver(encode(x)) == inject_record(x); the packet bytes the C++ codec
    # (serf_amd/host/wire.hpp, inside the library) writes are the oracle's C codec's, byte for byte; a packet peeked
    # from one cluster and delivered to a node of another reproduces its rumours
    from tests.test_bridge import KW, deliver_all, event_key, inject_all

    n = 256
    ga, gb = _ffi.Sim(hiplib, _ffi.make_config(n, **KW)), _ffi.Sim(hiplib, _ffi.make_config(n, **KW))
    oa = _ffi.Sim(oracle, _ffi.make_config(n, **KW))
    deliver_all(ga)
    deliver_all(oa)
    inject_all(gb)
    for t in range(30):
        for s in (ga, gb, oa):
            s.step(1)
        assert ga.digest() == gb.digest() == oa.digest(), f"tick {t}"
    c, co = both(oracle, hiplib, n, **KW)
    for s in (c, co):
        s.user_event_bytes(5, b"restart", b"now", True)
        s.user_event(5, 0xABCDEF01, 40)
        s.query(5, 91, _ffi.F_ACK | (2 << 8))
        s.leave(5)
        s.step(2)
    for node in (5, 17, 200):
        for k in range(3):
            assert c.peek_packet(node, k) == co.peek_packet(node, k)
    raw = c.peek_packet(5, 0)
    assert len(raw) > 40
    d, do = both(oracle, hiplib, n, **KW)
    for s in (d, do):
        off = 0
        while off < len(raw):
            off += s.deliver_message(100, raw[off:])
        s.step(25)
    assert d.digest() == do.digest()
    assert d.convergence(_ffi.K_EVENT, event_key(b"restart", b"now"), 1) == (n, n)
    assert d.convergence(_ffi.K_QUERY, 91, 1) == (n, n)
    # paged packets: sixteen records of one packet come out as sixteen messages
    e, eo = both(oracle, hiplib, n, pkt_records=16, **KW)
    for s in (e, eo):
        for i in range(14):
            s.user_event_bytes(9, b"ev%d" % i, b"", False)
        s.step(1)
    r1, r2 = e.peek_packet(9, 1), eo.peek_packet(9, 1)
    assert r1 == r2
    names, off = [], 0
    while off < len(r1):
        m, used = wire.decode_message(r1[off:])
        off += used
        names.append(m.name)
    assert sorted(names) == sorted(b"ev%d" % i for i in range(14))


@pytest.mark.parametrize("view_slots", [0, 16])
def test_query_response_relay_and_push_pull_on_the_gpu(oracle, hiplib, view_slots):
    # tests/test_bridge.py's round-4 scenarios (QueryResponse, Relay, PushPull at sim_deliver_message) on the HIP library, the
    # oracle beside it: the C++ decoder (serf_amd/host/wire.hpp) against the oracle's C one, SIM_OP_QRESP / SIM_OP_WITNESS /
    # the muted deliveries in ops_kernel against apply_op
    from tests.test_bridge import KW, deliver_query_traffic, push_pull_message
    from tests.test_bridge import event_key as _ffi_event_key

    n = 64
    kw = dict(KW, view_slots=view_slots)
    g, o = both(oracle, hiplib, n, **kw)
    for s in (g, o):
        deliver_query_traffic(s, n)
    assert g.digest() == o.digest()
    for which in (0, 1):
        assert g.query_responders(77, which) == o.query_responders(77, which)
    assert 9 in g.query_responders(77, 0) and 30 in g.query_responders(77, 1) and 31 not in g.query_responders(77, 1)
    assert g.query_status(77) == o.query_status(77)
    data = wire.encode_message(push_pull_message())
    for s in (g, o):
        s.watch(2)
        assert s.deliver_message(2, data + b"\x00") == len(data)
        s.deliver_message(6, wire.encode_message(wire.PushPull(10, {6: 5}, [6])))   # names its receiver as left: refuted
        s.step(1)
    assert g.digest() == o.digest()
    assert g.drain_events() == o.drain_events()
    for node in (2, 6):
        assert g.peek_packet(node, 0) == o.peek_packet(node, 0)
    m, _ = wire.decode_message(g.peek_packet(6, 0))
    assert isinstance(m, wire.Join) and m.id == 6
    for t in range(20):
        g.step(1)
        o.step(1)
        assert g.digest() == o.digest(), f"tick {t}"
    # a relay forwards whatever it wraps; a relay that is down forwards nothing; a ConflictResponse is ignored
    ev = wire.encode_message(wire.Relay(12, wire.UserEvent(5, b"deploy", b"v3", False)))
    conflict = bytes([wire.merge(wire.WIRE_LEN, wire.CONFLICT_RESPONSE), 2, 0x08, 0x01])
    for s in (g, o):
        assert s.deliver_message(7, ev + b"\x09") == len(ev)
        assert s.deliver_message(30, ev) == len(ev)          # node 30 is down
        assert s.deliver_message(3, conflict) == len(conflict)
    for t in range(12):
        g.step(1)
        o.step(1)
        assert g.digest() == o.digest(), f"relayed event, tick {t}"
    seen = g.convergence(_ffi.K_EVENT, _ffi_event_key(b"deploy", b"v3"), 5)
    assert seen == o.convergence(_ffi.K_EVENT, _ffi_event_key(b"deploy", b"v3"), 5) == (n - 2, n - 2)   # (nodes 9 and 30 are down)
    # the same refusals
    qr = wire.QueryResponse(3, 9, 5, 1)
    for bad in (wire.encode_message(wire.PushPull(4, {9999: 3})), wire.encode_message(wire.QueryResponse(3, 9, 64, 1)),
                wire.encode_message(wire.Relay(7, wire.PushPull(4))), wire.encode_message(wire.Relay(7, wire.Relay(8, qr))),
                wire.encode_message(wire.Relay(7, qr))[:-2], wire.encode_message(wire.Relay(64, qr))):
        for s in (g, o):
            with pytest.raises(_ffi.SimError) as e:
                s.deliver_message(1, bad)
            assert e.value.code == _ffi.EINVAL
    for s in (g, o):
        for op in (_ffi.OP_DELIVER, _ffi.OP_QRESP, _ffi.OP_WITNESS):
            with pytest.raises(_ffi.SimError):
                s.inject(s.tick, op, 1, 1, 0)


def test_two_decoders_agree_on_mutated_frames(oracle, hiplib):
    # tests/test_bridge.py::test_byte_boundary_survives_mutated_frames on BOTH libraries: the C++ decoder of the HIP library
    # (serf_amd/host/wire.hpp) and the oracle's C one take and refuse the same frames, with the same error and the same number
    # of bytes used, and what they schedule leaves the two simulations digest-identical
    from tests.test_bridge import KW, mutated_frames

    n = 64
    g, o = both(oracle, hiplib, n, **KW)
    for s in (g, o):
        s.query(4, 77, _ffi.F_ACK)
        s.step(2)
    taken = 0
    frames = list(mutated_frames(n, 2500, seed=11)) + list(mutated_frames(n, 2500, seed=12))
    for it, (node, buf) in enumerate(frames):
        res = []
        for s in (g, o):
            try:
                res.append(("ok", s.deliver_message(node, buf)))
            except _ffi.SimError as e:
                res.append(("err", e.code))
        assert res[0] == res[1], f"frame {it} {buf.hex()}: HIP {res[0]} oracle {res[1]}"
        taken += res[0][0] == "ok"
        if it % 100 == 99:
            g.step(1)
            o.step(1)
            assert g.digest() == o.digest(), f"after frame {it}"
    assert taken > 500


def test_push_pull_decoder_rules_on_the_gpu(oracle, hiplib):
    # a UserEvents bucket without its Lamport time is refused (types/user_event/user_events.rs:102); a status id that
    # comes twice keeps its place and takes the last value (the reference's IndexMap) — the library's C++ decoder and the oracle's
    # C one take and refuse the same frames and leave identical simulations
    from tests.test_bridge import _ld, _vi
    n = 8
    clocks = _vi(1, 40) + _vi(4, 30) + _vi(6, 20)
    ev = _ld(2, _ld(1, b"deploy") + _ld(2, b"x"))
    st = lambda nid, lt: _ld(2, _ld(1, str(nid).encode()) + _vi(2, lt))   # noqa: E731
    no_ltime = _ld(wire.PUSH_PULL, clocks + _ld(5, ev))
    good = _ld(wire.PUSH_PULL, clocks + st(3, 5) + st(4, 6) + st(3, 9) + _ld(5, _vi(1, 7) + ev))
    sims = [_ffi.Sim(lib, _ffi.make_config(n, flags=0, view_slots=0, event_ring=512)) for lib in (hiplib, oracle)]
    for sim in sims:
        with pytest.raises(_ffi.SimError) as ei:
            sim.deliver_message(0, no_ltime)
        assert ei.value.code == _ffi.EINVAL
        assert sim.deliver_message(0, good) == len(good)
        sim.step(2)
    assert sims[0].digest() == sims[1].digest()
    view = sims[0].dump(_ffi.ARR_VIEW).reshape(n, n)
    assert int(view[3, 0]["ltime"]) == 9 and int(view[4, 0]["ltime"]) == 6


def test_reference_merge_remote_state_kat_on_the_gpu(hiplib):
    # delegate_merge_remote_state (serf/base/tests/serf/delegate.rs:117-180), the message in bytes, the HIP library alone
    from tests.test_bridge import check_merge_kat, merge_kat_message

    n = 8
    sim = _ffi.Sim(hiplib, _ffi.make_config(n, flags=0, view_slots=0, event_ring=512))
    data = wire.encode_message(merge_kat_message())
    assert sim.deliver_message(0, data) == len(data)
    sim.step(1)
    check_merge_kat(sim, n)


def test_memberlist_flags_on_the_gpu(oracle, hiplib):
    # tests/test_oracle_swim.py::test_gossip_to_the_dead_time and ::test_awareness_scales_the_probe_interval on the HIP
    # library, the oracle beside it tick by tick
    n = 512
    kw = dict(fanout=3, view_slots=32, event_ring=16, query_ring=8, probe_interval=2, suspicion_mult=4, suspicion_max_mult=2)
    for gttd, want in ((0, _ffi.STATUS_ALIVE), (1, _ffi.STATUS_FAILED)):
        g, o = both(oracle, hiplib, n, gossip_to_the_dead=gttd, **kw)
        for s in (g, o):
            s.inject(3, _ffi.OP_CRASH, 10)
            s.inject(32, _ffi.OP_REVIVE, 10)
        for t in range(120):
            g.step(1)
            o.step(1)
            assert g.digest() == o.digest(), f"gossip_to_the_dead {gttd}: tick {t}"
        assert int(g.members(200)[0][10]) == want
    g, o = both(oracle, hiplib, 1024, awareness_probe=True, **dict(kw, suspicion_mult=6, suspicion_max_mult=3))
    for s in (g, o):
        for x in range(100, 1000, 45):
            s.inject(2, _ffi.OP_CRASH, x)
    for t in range(80):
        g.step(1)
        o.step(1)
        assert g.digest() == o.digest(), f"awareness probe: tick {t}"


def test_handle_prunes_wait_on_the_gpu(oracle, hiplib):
    # tests/test_oracle_swim.py::test_handle_prune_waits_while_the_member_is_leaving and ::test_remove_failed_node_prune (serf/base.rs:1628-1653,
    # remove.rs:96-153) on the HIP library, the oracle beside it tick by tick: the member table of node 0 keeps the pruned member for leave_delay
    # ticks and loses it then, with a Reap event of that tick; a Failed member goes at once
    n, victim, delay = 4096, 40, 9
    kw = dict(fanout=3, view_slots=32, event_ring=16, query_ring=8, probe_interval=5, leave_delay=delay, prune_delay=True)
    g, o = both(oracle, hiplib, n, **kw)
    for s in (g, o):
        s.watch(0)
        s.inject(2, _ffi.OP_CRASH, victim)
        s.inject(3, _ffi.OP_FORCE_LEAVE, 0, victim, 1)
    for t in range(3 + delay + 30):
        g.step(1)
        o.step(1)
        assert g.digest() == o.digest(), f"tick {t}"
        if 3 <= t < 3 + delay:
            assert g.stats(0).members == n and int(g.members(0)[0][victim]) == _ffi.STATUS_LEAVING, f"tick {t}: the erase must wait"
        if t == 3 + delay:
            assert g.stats(0).members == n - 1
    ev_g, ev_o = g.drain_events(), o.drain_events()
    assert [tuple(e) for e in ev_g] == [tuple(e) for e in ev_o]
    reap = [e for e in ev_g if e[2] == _ffi.EV_REAP and e[3] == victim]
    assert reap and reap[0][0] == 3 + delay
    assert all(g.stats(x).members == n - 1 for x in (0, 7, 4000)), "everybody has erased the member by now"
    g.close(); o.close()
    g, o = both(oracle, hiplib, 3, fanout=2, probe_interval=5, leave_delay=6, prune_delay=True)   # remove.rs:96-153
    for s in (g, o):
        s.inject(1, _ffi.OP_CRASH, 1)
    for t in range(400):
        g.step(1)
        o.step(1)
        if int(g.members(0)[0][1]) == _ffi.STATUS_FAILED and int(g.members(2)[0][1]) == _ffi.STATUS_FAILED:
            break
    assert g.digest() == o.digest()
    for s in (g, o):
        s.remove_failed_node(0, 1, prune=True)
    for t in range(6):
        g.step(1)
        o.step(1)
        assert g.digest() == o.digest()
    assert [g.stats(x).members for x in (0, 2)] == [2, 2]


def test_reconnector_on_the_gpu(oracle, hiplib):
    # tests/test_oracle_reconnect.py on the HIP library, the oracle beside it tick by tick: a node that resumes after it was
    # declared failed, with nobody gossiping to it any more, is reached by a peer's Reconnector (base.rs:612-681), refutes,
    # and is alive again everywhere; eight nodes that stay down keep drawing attempts (request list -> SIM_OP_RECONNECT ->
    # nothing happens); hand-placed attempts that share a node run one per tick
    n = 512
    kw = dict(fanout=3, view_slots=32, event_ring=16, query_ring=8, probe_interval=2, suspicion_mult=4, suspicion_max_mult=2)
    g, o = both(oracle, hiplib, n, gossip_to_the_dead=1, reconnect_interval=8, push_pull_interval=40, **kw)
    for s in (g, o):
        s.inject(3, _ffi.OP_CRASH, 10)
        s.inject(32, _ffi.OP_REVIVE, 10)
        for x in range(100, 500, 50):
            s.inject(2, _ffi.OP_CRASH, x)
    for t in range(200):
        g.step(1)
        o.step(1)
        assert g.digest() == o.digest(), f"reconnector: tick {t}"
    assert int(g.members(201)[0][10]) == _ffi.STATUS_ALIVE and int(g.dump(_ffi.ARR_ROWS)["inc"][10]) >= 1
    assert int(g.members(201)[0][100]) == _ffi.STATUS_FAILED
    t0 = g.tick
    for s in (g, o):
        s.inject(t0, _ffi.OP_RECONNECT, 5, 100)     # the target is down: forgotten
        s.inject(t0, _ffi.OP_RECONNECT, 6, 10)
        s.inject(t0, _ffi.OP_RECONNECT, 7, 10)      # shares node 10 with the attempt before: next tick
        s.inject(t0, _ffi.OP_RECONNECT, 10, 8)      # and so does this one
        s.inject(t0, _ffi.OP_RECONNECT, 20, 21)
    for t in range(6):
        g.step(1)
        o.step(1)
        assert g.digest() == o.digest(), f"hand-placed attempts: tick {t}"
    sc.assert_same_state(g, o, "reconnector final")


def test_suspect_import_without_an_exchange_of_the_librarys(oracle, hiplib):
    # (r6) heads == NULL means "the lists the library's own exchange carried" (sim_exchange_chunk): a handle that never called
    # sim_exchange_init has none — SIM_EINVAL on the product, as on the oracle (which has no collective library at all)
    for lib in (hiplib, oracle):
        s = _ffi.Sim(lib, _ffi.make_config(256, fanout=3, view_slots=16, probe_interval=2))
        s.step(3)
        with pytest.raises(_ffi.SimError) as ei:
            s.suspect_import(1, 0, 1)
        assert ei.value.code == _ffi.EINVAL
        s.close()
