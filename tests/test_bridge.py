"""The byte boundary of the delegate: SerfDelegate::notify_message(buf)
(delegate.rs:157-163) and SerfDelegate::broadcast_messages -> Bytes (delegate.rs:317-384) as C-ABI entry points —
sim_deliver_message / sim_peek_packet (+ sim_inject_record, sim_user_event_bytes).  CPU side: the oracle's C restatement of
the codec against serf_amd/wire.py (the Python restatement the reference's round-trip tests pin, tests/test_wire.py);
the HIP library (C++ codec, serf_amd/host/wire.hpp) is compared byte for byte in tests/test_host_paths_gpu.py."""
import numpy as np
import pytest

from serf_amd import _ffi, wire

KW = dict(fanout=3, view_slots=16, event_ring=32, query_ring=16)


def event_key(name: bytes, payload: bytes) -> int:
    h = 2166136261
    for x in name + b"\xff" + payload:
        h = ((h ^ x) * 16777619) & 0xFFFFFFFF
    return h or 1


def wmeta(kind, flags, nbytes):
    return ((63 - min(63, (nbytes + 15) // 16)) << 18) | (kind << 4) | flags


def scenario_messages():
    return [(10, wire.UserEvent(7, b"deploy", b"v1", True)), (11, wire.Join(9, 33)), (12, wire.Leave(11, 44, True)),
            (13, wire.Query(3, 77, 5, flags=1, relay_factor=2, timeout_ms=1000, name=b"q", payload=b"x"))]


def deliver_all(sim):
    for node, m in scenario_messages():
        data = wire.encode_message(m)
        assert sim.deliver_message(node, data + b"\xAA\xBB") == len(data)   # trailing bytes belong to the next message


def inject_all(sim):
    msgs = [m for _, m in scenario_messages()]
    sim.inject_record(0, 10, event_key(b"deploy", b"v1"), wmeta(_ffi.K_EVENT, 1, len(wire.encode_message(msgs[0]))), 7)
    sim.inject_record(0, 11, 33, wmeta(_ffi.K_JOIN, 0, len(wire.encode_message(msgs[1]))), 9)
    sim.inject_record(0, 12, 44, wmeta(_ffi.K_LEAVE, 1, len(wire.encode_message(msgs[2]))), 11)
    sim.inject_record(0, 13, 77, wmeta(_ffi.K_QUERY, _ffi.F_ACK, 48), 3)


def test_deliver_of_encoded_message_is_inject_of_the_record(oracle):
    n = 256
    a, b = _ffi.Sim(oracle, _ffi.make_config(n, **KW)), _ffi.Sim(oracle, _ffi.make_config(n, **KW))
    deliver_all(a)
    inject_all(b)
    for t in range(30):
        a.step(1)
        b.step(1)
        assert a.digest() == b.digest(), f"tick {t}"
    assert a.convergence(_ffi.K_EVENT, event_key(b"deploy", b"v1"), 7) == (n, n)
    st, lt = a.members(200)
    assert st[44] == _ffi.STATUS_NONE and st[33] == _ffi.STATUS_ALIVE and lt[33] == 9   # 44 was pruned, 33's join intent applied


def test_peek_decode_deliver_reproduces_the_rumour_in_a_second_cluster(oracle):
    n = 256
    c, d = _ffi.Sim(oracle, _ffi.make_config(n, **KW)), _ffi.Sim(oracle, _ffi.make_config(n, **KW))
    c.user_event_bytes(5, b"restart", b"now", True)
    c.user_event(5, 0xABCDEF01, 40)        # a bare key: the library was never told its content
    c.query(5, 91, _ffi.F_ACK | (2 << 8))
    c.leave(5)
    c.step(2)
    got = []
    for k in range(3):
        raw, off = c.peek_packet(5, k), 0
        while off < len(raw):
            m, used = wire.decode_message(raw[off:])
            off += used
            got.append(m)
    ev = [m for m in got if isinstance(m, wire.UserEvent)]
    assert any(m.name == b"restart" and m.payload == b"now" and m.cc for m in ev)
    assert any(m.name == b"#abcdef01" and m.payload == b"" for m in ev)
    q = next(m for m in got if isinstance(m, wire.Query))
    assert (q.id, q.flags, q.relay_factor, q.from_node) == (91, 1, 2, 5)
    lv = next(m for m in got if isinstance(m, wire.Leave))
    assert lv.id == 5 and not lv.prune
    # a packet is 4 records: intents first, then queries, then events (delegate.rs:328-383)
    first = []
    raw, off = c.peek_packet(5, 0), 0
    while off < len(raw):
        m, used = wire.decode_message(raw[off:])
        off += used
        first.append(type(m).__name__)
    assert first[:2] == ["Leave", "Query"] and set(first[2:]) == {"UserEvent"}
    # hand the packet to node 100 of a second cluster, message by message
    raw, off = c.peek_packet(5, 0), 0
    while off < len(raw):
        off += d.deliver_message(100, raw[off:])
    d.step(25)
    assert d.convergence(_ffi.K_EVENT, event_key(b"restart", b"now"), 1) == (n, n)
    assert d.convergence(_ffi.K_QUERY, 91, 1) == (n, n)
    assert d.convergence(_ffi.K_LEAVE, 5, 1)[0] == n
    assert d.peek_packet(100, 0) != b"" or d.tick > 20   # node 100 rebroadcast what it was handed


def test_malformed_and_unsupported_messages_are_refused(oracle):
    sim = _ffi.Sim(oracle, _ffi.make_config(64, **KW))
    good = wire.encode_message(wire.Join(3, 5))
    for bad in (b"", good[:-1], bytes([good[0] ^ 1]) + good[1:], wire.encode_message(wire.Join(3, 9999)),
                wire.encode_message(wire.PushPull(4))):
        if not bad:
            continue
        with pytest.raises(_ffi.SimError):
            sim.deliver_message(1, bad)
    with pytest.raises(_ffi.SimError) as e:
        sim.user_event_bytes(1, b"n" * 300, b"p" * 300)
    assert e.value.code == _ffi.ETOOBIG
    # Filter::Id lists are installed, a Filter::Tag is refused (the host evaluates tag expressions)
    ids = b"".join(bytes([wire.merge(wire.WIRE_LEN, 1)]) + wire.ld(wire.node_id(g)) for g in (7, 8))
    q = wire.Query(1, 55, 2, flags=1, relay_factor=0, timeout_ms=100, name=b"", payload=b"", filters=[ids])
    sim.deliver_message(2, wire.encode_message(q))
    tagf = bytes([wire.merge(wire.WIRE_LEN, 2)]) + wire.ld(b"\x0a\x04role\x12\x03web")
    with pytest.raises(_ffi.SimError):
        sim.deliver_message(2, wire.encode_message(wire.Query(1, 56, 2, flags=1, relay_factor=0, timeout_ms=100, name=b"", payload=b"", filters=[tagf])))
    sim.watch(7)
    sim.watch(9)
    sim.step(20)
    seen = {(e[1], e[3]) for e in sim.drain_events() if e[2] == _ffi.EV_QUERY}
    assert (7, 55) in seen and (9, 55) not in seen, "the Id filter of the delivered query is in force"


def test_swim_records_can_be_handed_in_too(oracle):
    # memberlist's own messages are not serf messages (no byte form here: memberlist-proto is not vendored), but the
    # record boundary takes them: a dead{from = X} about node 9 handed to node 3
    n = 128
    sim = _ffi.Sim(oracle, _ffi.make_config(n, probe_interval=50, **KW))
    sim.inject(0, _ffi.OP_CRASH, 9)
    sim.inject_record(1, 3, 9, wmeta(_ffi.K_DEAD, 0, 32), 0 | (4 << 32))
    sim.step(30)
    st, _ = sim.members(100)
    assert st[9] == _ffi.STATUS_FAILED
