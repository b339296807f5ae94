"""QueryParam.filters and Serf::set_tags.

Reference: should_process_query query.rs:439-521, its call site base.rs:1062-1073 ("even if we don't process it
further, we should rebroadcast"), Serf::set_tags api.rs:219-235, handle_node_update base.rs:1576-1624.  The cases
follow the reference's own tests: `should_process` (base/tests/serf/event.rs:563-635), `serf_query_filter`
(event.rs:905-985) and `serf_set_tags` (base/tests/serf.rs:348-420).  CPU side: the oracle; HIP parity for the same
scenarios is in tests/test_parity_gpu.py."""
import numpy as np
import pytest

from serf_amd import _ffi
from serf_amd.filters import Filter, TagTable

ACK, RESPOND = _ffi.F_ACK, _ffi.F_RESPOND
KW = dict(fanout=3, view_slots=16, event_ring=64, query_ring=64)


def run_query(lib, n, origin, qid, ids, mask, tags=(), flags=ACK | RESPOND, ticks=40, watch=(), **kw):
    sim = _ffi.Sim(lib, _ffi.make_config(n, **{**KW, **kw}))
    for node, cls in tags:
        sim.set_tags(node, cls)
    for w in watch:
        sim.watch(w)
    sim.step(1)
    sim.query(origin, qid, flags, ids, mask)
    sim.step(ticks)
    return sim


def test_should_process_reference_cases(oracle):
    """event.rs:563-635 with the reference's tags and expressions, on node 5 of a 64-node cluster."""
    n, me = 64, 5
    tt = TagTable()
    cls = tt.class_of({"role": "webserver", "datacenter": "east-aws"})
    assert cls == 1 and tt.class_of({"role": "webserver", "datacenter": "east-aws"}) == 1

    def processed_by_me(filters):
        ids, mask = tt.compile(filters)
        sim = run_query(oracle, n, origin=0, qid=77, ids=ids, mask=mask, tags=[(me, cls)], watch=[me])
        evs = [e for e in sim.drain_events() if e[1] == me and e[2] == _ffi.EV_QUERY and e[3] == 77]
        seen, up = sim.convergence(_ffi.K_QUERY, 77, 1)
        assert seen == up == n, "filtered or not, every node saw (and rebroadcast) the query"
        acks, resp, _ = sim.query_status(77)
        sim.close()
        return len(evs) == 1, acks, resp

    # ids "foo", "bar" of the reference: two other nodes that carry no tags; the tag filters exclude them
    ok, acks, resp = processed_by_me([Filter.id([1, 2, me]), Filter.tag("role", "^web"), Filter.tag("datacenter", "aws$")])
    assert ok and acks == 1 and resp == 1
    # "Omit node"
    ok, acks, resp = processed_by_me([Filter.id([1, 2])])
    assert not ok and acks == 2 and resp == 2
    # "Filter on missing tag"
    ok, acks, resp = processed_by_me([Filter.tag("other", "cool")])
    assert not ok and acks == 0 and resp == 0
    # "Bad tag"
    ok, acks, resp = processed_by_me([Filter.tag("role", "db")])
    assert not ok and acks == 0 and resp == 0
    # no filters at all: everybody
    ok, acks, resp = processed_by_me([])
    assert ok and acks == n and resp == n


def test_serf_query_filter(oracle):
    """event.rs:905-985: three nodes, the query from s2 is filtered "to only s1", acks requested, relay_factor 1:
    exactly one ack and one response arrive."""
    sim = _ffi.Sim(oracle, _ffi.make_config(3, fanout=2, view_slots=3, event_ring=16, query_ring=16))
    for w in range(3):
        sim.watch(w)
    sim.query(1, 9, ACK | RESPOND | (1 << 8), ids=[0])
    sim.step(12)
    acks, resp, _ = sim.query_status(9)
    assert (acks, resp) == (1, 1)
    evs = [(e[1], e[3]) for e in sim.drain_events() if e[2] == _ffi.EV_QUERY]
    assert evs == [(0, 9)], "only s1's user code sees the query (base.rs:1126-1151 comes after the filter)"
    seen, up = sim.convergence(_ffi.K_QUERY, 9, 1)
    assert seen == up == 3


def test_tag_classes_partition_a_cluster(oracle):
    """A role filter over a big cluster: the responders are exactly the nodes of the matching classes."""
    n = 2048
    tt = TagTable()
    roles = [tt.class_of({"role": r, "dc": dc}) for r in ("web", "db", "cache") for dc in ("east", "west")]
    rng = np.random.default_rng(4)
    cls_of = rng.integers(0, len(roles) + 1, n)          # 0 = no tags
    sim = _ffi.Sim(oracle, _ffi.make_config(n, **KW))
    sim.init_tags([roles[c - 1] if c else 0 for c in cls_of])   # Options::with_tags
    ids, mask = tt.compile([Filter.tag("role", "^(web|cache)$"), Filter.tag("dc", "st$")])   # both dcs end in "st"
    want = {roles[i] for i, (r, dc) in enumerate((r, dc) for r in ("web", "db", "cache") for dc in ("east", "west")) if r != "db"}
    assert ids is None and mask == sum(1 << c for c in want)
    sim.query(17, 5, ACK, ids, mask)
    sim.step(40)
    acks, resp, _ = sim.query_status(5)
    expect = int(sum(1 for node in range(n) if cls_of[node] and roles[cls_of[node] - 1] in want))
    assert acks == expect and resp == 0
    seen, up = sim.convergence(_ffi.K_QUERY, 5, 1)
    assert seen == up == n


def test_filter_bounds(oracle):
    sim = _ffi.Sim(oracle, _ffi.make_config(64, **KW))
    with pytest.raises(_ffi.SimError):          # a 13th id: refused at the API ...
        sim.query(0, 3, ACK, ids=list(range(1, 14)))
    for i in range(13):                         # ... dropped and counted when scheduled one by one
        sim.inject(0, _ffi.OP_QUERY_FILTER_ID, 0, 3, 1 + i)
    sim.inject(0, _ffi.OP_QUERY, 0, 3, ACK)
    sim.step(30)
    assert sim.cluster_stats()["ops_dropped"] == 1
    assert sim.query_status(3)[0] == 12
    assert sim.query_responders(3, 0) == list(range(1, 13)) and sim.query_responders(3, 1) == []   # the twelve ids of the filter acked
    with pytest.raises(_ffi.SimError):
        sim.query_responders(4, 0)              # no such running query
    with pytest.raises(_ffi.SimError):
        sim.set_tags(0, 32)
    with pytest.raises(_ffi.SimError):
        sim.inject(0, _ffi.OP_QUERY_FILTER_ID, 0, 3, 64)   # not a node
    # a later query with the same residue takes the filter entry over (SIM_QT = 256 direct-mapped, like the tracker)
    sim.query(0, 3 + 256, ACK)
    sim.step(30)
    assert sim.query_status(3 + 256)[0] == 64
    tt = TagTable()
    with pytest.raises(ValueError):
        tt.compile([Filter.id(range(13))])
    assert tt.compile([Filter.id([1, 2]), Filter.id([3])]) == (None, 0)       # empty intersection: nobody
    assert tt.compile([Filter.id([1, 2, 3]), Filter.id([3, 2])]) == ([2, 3], _ffi.NO_TAG_FILTER)


def test_set_tags_updates_reach_the_cluster(oracle):
    """serf.rs:348-420 (two nodes set tags, each learns the other's) at cluster size: update_node bumps the
    incarnation and gossips an alive message; every member that knew the node alive gets MemberEventType::Update."""
    n = 256
    kw = dict(KW, probe_interval=2, suspicion_mult=3, suspicion_max_mult=2)
    sim = _ffi.Sim(oracle, _ffi.make_config(n, **kw))
    watchers = [0, 1, 100, 255]
    for w in watchers:
        sim.watch(w)
    sim.set_tags(0, 1)   # {"port": "8080"}
    sim.set_tags(1, 2)   # {"datacenter": "east-aws"}
    sim.step(40)
    evs = [e for e in sim.drain_events() if e[2] == _ffi.EV_UPDATE]
    got = {(e[1], e[3]) for e in evs}
    want = {(w, s) for w in watchers for s in (0, 1) if w != s}
    assert got == want, "one update per (observer, subject), none about oneself"
    assert len(evs) == len(want)
    assert all(e[4] == 1 for e in evs), "the new incarnation (0 -> 1)"
    assert sim.stats(0).incarnation == 1
    assert sim.cluster_stats()["failed"] == 0
    # a second change is a second update
    sim.set_tags(0, 3)
    sim.step(40)
    evs = [e for e in sim.drain_events() if e[2] == _ffi.EV_UPDATE]
    assert {(e[1], e[3], e[4]) for e in evs} == {(w, 0, 2) for w in watchers if w != 0}


def test_filters_and_tags_survive_a_checkpoint(oracle):
    n = 256
    cfg = dict(KW, probe_interval=2)

    def drive(sim):
        sim.step(25)
        return sim.digest(), sim.query_status(11), sim.query_status(12)

    a = _ffi.Sim(oracle, _ffi.make_config(n, **cfg))
    a.init_tags([(1 + node % 5) if node % 3 == 0 else 0 for node in range(n)])
    a.query(4, 11, ACK, ids=[9, 10, 11], tag_mask=0b110)
    a.inject(6, _ffi.OP_QUERY_FILTER_TAGS, 8, 12, 0b1010)   # pending at the time of the image
    a.inject(6, _ffi.OP_QUERY, 8, 12, ACK | RESPOND)
    a.inject(9, _ffi.OP_SET_TAGS, 10, 2, 0)
    a.step(5)
    img = a.snapshot()
    b = _ffi.Sim(oracle, _ffi.make_config(n, **cfg))
    b.restore(img)
    assert b.digest() == a.digest()
    assert drive(a) == drive(b)
    assert a.query_status(12)[0] > 0


def test_query_id_reused_does_not_inherit_filters(oracle):
    """A filter entry is sealed by its SIM_OP_QUERY: issuing a query again under an id that was used before starts from
    no filters (`sim_query(id)` after `sim_query_filtered(id)` was silently filtered)."""
    n = 64
    sim = _ffi.Sim(oracle, _ffi.make_config(n, **KW))
    sim.step(1)
    sim.query(0, 77, ACK, ids=[1, 2], tag_mask=_ffi.NO_TAG_FILTER)
    sim.step(30)
    assert sim.query_status(77)[0] == 2
    sim.query(3, 77, ACK)  # same id, no filters this time
    sim.step(30)
    assert sim.query_status(77)[0] == n
    sim.query(4, 77, ACK, ids=[9])  # and a third time with another list: not appended to the first one
    sim.step(30)
    assert sim.query_status(77)[0] == 1
    sim.close()
