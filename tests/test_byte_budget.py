"""The packet's byte budget: `broadcast_messages(limit)` (delegate.rs:317-384) and memberlist's
`get_broadcasts` (App. B.1) fill a packet by BYTES — walk the queue in drain order, take what still fits, skip what
does not (a smaller message further on may).  The simulator keeps its 4-record cell and adds that rule on top: a packet
takes at most SIM_P records and at most SIM_PKT_BYTES = 1 400 bytes, lengths in 16-byte units as the codec prices them."""
from serf_amd import _ffi, wire


def transmits_by_key(sim, node):
    q = sim.dump(_ffi.ARR_QUEUE).reshape(sim.n, _ffi.Q)[node]
    return {int(r["key"]): (int(r["meta"]) >> 24) & 63 for r in q if r["meta"] != 0xFFFFFFFF}


def test_large_events_do_not_all_fit_one_packet(oracle):
    sim = _ffi.Sim(oracle, _ffi.make_config(64, fanout=3, view_slots=8))
    big = wire.user_event_len(1, b"deploy", b"x" * 506)        # the largest event the default limit allows: 528 bytes framed
    assert (big + 15) // 16 == 33 and 3 * 33 > 1400 // 16 >= 2 * 33 + 1
    for key in (101, 102, 103):
        sim.user_event(0, key, big)
    sim.user_event(0, 104, 16)
    sim.step(1)
    # three packets went out.  Packet 1: 103, 102 (66 units), 101 does not fit, 104 does; packet 2: 101 first (fewest
    # transmits), then 103, then 104; packet 3: 102, 101, 104.  With a pure record budget all four would have gone 3 times.
    assert transmits_by_key(sim, 0) == {101: 2, 102: 2, 103: 2, 104: 3}


def test_small_records_are_unaffected(oracle):
    sim = _ffi.Sim(oracle, _ffi.make_config(64, fanout=3, view_slots=8))
    for key in (1, 2, 3, 4, 5):
        sim.user_event(0, key, 40)
    sim.step(1)
    t = transmits_by_key(sim, 0)
    # five 3-unit events, four per packet, fewest transmits first: 12 transmits spread 3,3,2,2,2
    assert sorted(t.values()) == [2, 2, 2, 3, 3] and sum(t.values()) == 12


def test_paged_packet_fills_up_by_bytes_not_by_records(oracle):
    # pkt_records = 16: the cell could hold the whole queue, the 1 400-byte budget is what binds — sixteen 96-byte events
    # (6 units each: 16 x 6 = 96 > 87 units) go out fourteen at a time ("so the byte budget is the
    # one that binds")
    sim = _ffi.Sim(oracle, _ffi.make_config(64, fanout=3, view_slots=8, pkt_records=16))
    for key in range(1, 17):
        sim.user_event(0, key, 96)
    sim.step(1)
    t = transmits_by_key(sim, 0)
    assert len(t) == 16 and sum(t.values()) == 3 * 14 and set(t.values()) == {2, 3}
    pk = sim.dump(_ffi.ARR_INBOX).reshape(3, 4, 64)     # [slot][page][node]: what went out, receiver-indexed
    per_packet = ((pk["hi_meta"] >> 4) & 15 != 0).sum(axis=(1, 3))   # records per (slot, receiver)
    assert sorted(per_packet[per_packet > 0].tolist()) == [14, 14, 14]
    # small messages: the whole queue in every packet (16 x 1 unit)
    sim = _ffi.Sim(oracle, _ffi.make_config(64, fanout=3, view_slots=8, pkt_records=16))
    for key in range(1, 17):
        sim.user_event(0, key, 16)
    sim.step(1)
    assert set(transmits_by_key(sim, 0).values()) == {3}


def test_record_budget_of_a_one_page_packet_still_binds_first(oracle):
    sim = _ffi.Sim(oracle, _ffi.make_config(64, fanout=3, view_slots=8, pkt_records=4))
    for key in range(1, 17):
        sim.user_event(0, key, 16)
    sim.step(1)
    assert sum(transmits_by_key(sim, 0).values()) == 3 * 4
