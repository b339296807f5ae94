"""(r6) The model bounds lifted: a queue of up to SIM_Q = 64 entries (the tick kernel keeps 16 in registers, deeper nodes are
finished by deep_queue_kernel) and ring buckets that continue in overflow rows (sim_config.ring_overflow) — the reference's
queues hold 4 096 (options.rs:513, base.rs:728-739), its buckets are Vecs (base.rs:801-813, 1027-1042).  The HIP path against
the oracle, bit for bit, under loads that overflowed the old bounds (16 queue slots, 6 keys): every fan-out model and packet
size, one handle and shard handles, with the memberlist layer, push-pull, the QueueChecker, checkpoints — and past the NEW
bounds, where both count the same drops."""
import numpy as np
import pytest

from serf_amd import _ffi
from tests import _scenario as sc

pytestmark = pytest.mark.gpu


def pair(oracle, hiplib, n, **kw):
    return _ffi.Sim(hiplib, _ffi.make_config(n, **kw)), _ffi.Sim(oracle, _ffi.make_config(n, **kw))


def run_pair(g, o, ops, ticks, what, every=1, full_every=10):
    sc.apply_schedule(g, ops)
    sc.apply_schedule(o, ops)
    deepest = 0
    for t in range(0, ticks, every):
        g.step(every)
        o.step(every)
        deepest = max(deepest, o.cluster_stats()["max_queue"])
        if g.digest() != o.digest():
            sc.assert_same_state(g, o, f"{what} tick {t + every}")
            raise AssertionError(f"{what}: digest differs after tick {t + every} but the arrays agree")
        if (t // every) % full_every == 0:
            sc.assert_same_state(g, o, f"{what} tick {t + every}")
    sc.assert_same_state(g, o, f"{what} final")
    assert g.cluster_stats() == o.cluster_stats() or {k: v for k, v in g.cluster_stats().items() if k != "events_lost"} == \
        {k: v for k, v in o.cluster_stats().items() if k != "events_lost"}
    return deepest


CASES = [
    # n, fanout, P, swim, rf, vshards, chunks, rate, X
    (256, 3, 4, 0, False, 1, 0, 1.5, 4),
    (1024, 3, 4, 5, False, 1, 0, 2.0, 8),
    (1024, 4, 4, 5, True, 1, 0, 2.0, 8),
    (1024, 4, 16, 4, True, 1, 0, 2.5, 8),
    (2048, 4, 8, 4, False, 4, 2, 2.0, 8),
    (1000, 3, 4, 3, True, 1, 0, 1.5, 6),     # ragged: the last wave is not whole
    (4096, 4, 4, 5, False, 1, 0, 3.0, 8),
]


@pytest.mark.parametrize("n,fanout,P,swim,rf,vshards,chunks,rate,X", CASES)
def test_deep_queues_and_overflow_rows_match_the_oracle(oracle, hiplib, n, fanout, P, swim, rf, vshards, chunks, rate, X):
    kw = dict(fanout=fanout, view_slots=96, event_ring=32, query_ring=16, ring_overflow=X, pkt_records=P, probe_interval=swim,
              loss=0.01 if swim else 0.0, push_pull_interval=6 if swim else 0, leave_delay=6, vshards=vshards, chunks=chunks,
              queue_check_interval=9, reap_interval=7 if swim else 0, reconnect_timeout=60, tombstone_timeout=80, intent_timeout=30)
    if rf:
        kw["flags"] = _ffi.CF_BASELINE_JOINED | _ffi.CF_RANDOM_FANOUT
    g, o = pair(oracle, hiplib, n, **kw)
    ops = sc.schedule(n, 40, rate=rate, seed=n + fanout + P, max_member_subjects=40)
    deepest = run_pair(g, o, ops, 80, f"n={n} f={fanout} P={P} rf={rf}")
    assert deepest > _ffi.Q_HOT, f"the load was meant to take queues beyond the {_ffi.Q_HOT} hot keys (deepest {deepest})"
    g.close()
    o.close()


def test_same_lamport_time_fills_a_bucket_and_its_overflow_rows(oracle, hiplib):
    # 40 user events and 30 queries issued in the SAME tick by distinct nodes: every one gets Lamport time 1 (2 for the pre-joined
    # clocks) — one event bucket and one query bucket take 40 / 30 keys: 6 in place, the rest in overflow rows
    n, X = 2048, 8
    for rf in (False, True):
        kw = dict(fanout=4, view_slots=64, event_ring=32, query_ring=16, ring_overflow=X, pkt_records=8)
        if rf:
            kw["flags"] = _ffi.CF_BASELINE_JOINED | _ffi.CF_RANDOM_FANOUT
        g, o = pair(oracle, hiplib, n, **kw)
        ops = [(2, _ffi.OP_USER_EVENT, 17 * i + 3, 1000 + i, 40 + i) for i in range(40)] + \
              [(2, _ffi.OP_QUERY, 29 * i + 5, 5000 + i, _ffi.F_ACK) for i in range(30)]
        run_pair(g, o, ops, 50, f"same-ltime rf={rf}", full_every=5)
        er = o.dump(_ffi.ARR_ERING).reshape(X + 32, n)
        qr = o.dump(_ffi.ARR_QRING).reshape(X + 16, n)
        assert (er[:X]["ltime"] != 0).sum(axis=0).max() >= 6 and (qr[:X]["ltime"] != 0).sum(axis=0).max() >= 4, "the overflow rows were meant to be used"
        assert o.cluster_stats()["overflow"] == 0
        # every node has applied every event: convergence counts go through the overflow rows
        lt = int(o.dump(_ffi.ARR_ROWS)["event_clock"].max()) - 1
        for key in (1000, 1017, 1039):
            sg, ug = g.convergence(_ffi.K_EVENT, key, lt)
            so, uo = o.convergence(_ffi.K_EVENT, key, lt)
            assert (sg, ug) == (so, uo) and so == uo == n
        seen, up = g.convergence_many([(_ffi.K_EVENT, 1000 + i, lt) for i in range(0, 40, 3)])
        assert list(seen) == [n] * len(seen) and up == n
        g.close()
        o.close()


def test_past_the_new_bounds_both_count_the_same_drops(oracle, hiplib):
    # a load no 64-slot queue and no 2 overflow rows hold: the bounds bite — identically
    n = 1024
    kw = dict(fanout=4, view_slots=96, event_ring=32, query_ring=16, ring_overflow=2, pkt_records=16, probe_interval=5, loss=0.01,
              push_pull_interval=6, leave_delay=6, flags=_ffi.CF_BASELINE_JOINED | _ffi.CF_RANDOM_FANOUT)
    g, o = pair(oracle, hiplib, n, **kw)
    ops = sc.schedule(n, 40, rate=4.0, seed=5, max_member_subjects=40)
    deepest = run_pair(g, o, ops, 70, "past the bounds")
    assert deepest == _ffi.Q and o.cluster_stats()["overflow"] > 0
    g.close()
    o.close()


def test_checkpoint_with_deep_queues(oracle, hiplib):
    # an image taken while queues are deep and overflow rows are in use restores into both implementations and carries on
    n = 1024
    kw = dict(fanout=3, view_slots=96, event_ring=32, query_ring=16, ring_overflow=8, pkt_records=4, probe_interval=5, loss=0.01,
              push_pull_interval=6, leave_delay=6)
    g, o = pair(oracle, hiplib, n, **kw)
    ops = sc.schedule(n, 40, rate=2.0, seed=77, max_member_subjects=40)
    sc.apply_schedule(g, ops)
    sc.apply_schedule(o, ops)
    g.step(30)
    o.step(30)
    assert o.cluster_stats()["max_queue"] > _ffi.Q_HOT
    img_g, img_o = g.snapshot(), o.snapshot()
    assert bytes(img_g) == bytes(img_o)
    g2, o2 = pair(oracle, hiplib, n, **kw)
    g2.restore(img_o)
    o2.restore(img_g)
    for s in (g, o, g2, o2):
        s.step(40)
    assert g.digest() == o.digest() == g2.digest() == o2.digest()
    sc.assert_same_state(g2, o, "restored")
    for s in (g, o, g2, o2):
        s.close()


@pytest.mark.parametrize("V,C,n", [(4, 1, 2048), (4, 2, 4096), (8, 1, 4096)])
def test_a_shard_packs_its_own_slab_in_place(oracle, hiplib, V, C, n):
    # (r6) When the library issues the round's exchange itself (sim_exchange_init), the slab a shard addresses to ITSELF is packed straight
    # into its place in the receive buffer and left out of the group of sends / receives.  RCCL refuses several ranks on one device, so the
    # V > 1 form is driven here through the test hook that switches the same code on without a communicator: V shard handles on one
    # GPU, device copies for every slab but the diagonal, against the oracle's slices.
    import ctypes as C_
    import torch

    from tests.test_parity_gpu import _push_pull_on_one_gpu, _suspicions_on_one_gpu

    m = n // V
    kw = dict(fanout=4, view_slots=96, event_ring=16, query_ring=8, leave_delay=6, probe_interval=4, loss=0.02, push_pull_interval=3,
              pkt_records=8, ring_overflow=4, flags=_ffi.CF_BASELINE_JOINED | _ffi.CF_RANDOM_FANOUT)
    ref = _ffi.Sim(oracle, _ffi.make_config(n, vshards=V, **kw))
    shards, send, recv = [], [], []
    for g in range(V):
        s = _ffi.Sim(hiplib, _ffi.make_config(n, vshards=V, shard_rank=g, shard_count=V, chunks=C if C > 1 else 0, **kw))
        kind, planes, pb, rb = s.exchange_layout()
        send.append(torch.zeros(pb, dtype=torch.uint8, device="cuda"))
        recv.append([torch.zeros(rb, dtype=torch.uint8, device="cuda") for _ in range(2 if C > 1 else 1)])
        s.bind_exchange3(send[-1].data_ptr(), pb, recv[-1][0].data_ptr(), recv[-1][-1].data_ptr(), rb)
        assert hiplib.dll.sim_t_self_direct(s.h, C_.c_int(1)) == 0
        shards.append(s)
    ops = sc.schedule(n, 25, rate=2.5, seed=9, max_member_subjects=40)
    for s in shards + [ref]:
        sc.apply_schedule(s, ops)
    reg = send[0].numel() // C
    slab = reg // V
    for t in range(50):
        for s in shards:
            s.step_begin()
        if shards[0].pp_due():
            _push_pull_on_one_gpu(shards)
        into = [r[shards[0].tick & 1] if C > 1 else r[0] for r in recv]
        for c in range(C):
            for s in shards:
                s.step_chunk(c)
                s.sync()
            for g in range(V):
                for src in range(V):
                    if src != g:   # the diagonal never travels: shard g packed it into `into[g]` itself
                        into[g][c * reg + src * slab:c * reg + (src + 1) * slab].copy_(send[src][c * reg + g * slab:c * reg + (g + 1) * slab])
        for s in shards:
            s.step_end()
            s.sync()
        _suspicions_on_one_gpu(shards)
        torch.cuda.synchronize()
        ref.step(1)
        if t % 7 == 0 or t == 49:
            for g, s in enumerate(shards):
                for which in (_ffi.ARR_ROWS, _ffi.ARR_QUEUE):
                    a, b = s.dump(which), ref.dump(which)
                    per = len(b) // n
                    i = sc.first_diff(a, b[g * m * per:(g + 1) * m * per])
                    assert i is None, f"shard {g} array {which} element {i} differs at tick {t}"
    for s in shards + [ref]:
        s.close()
