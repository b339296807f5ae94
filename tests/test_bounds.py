"""(r6) The lifted model bounds on the CPU: the oracle with its product bounds — SIM_Q = 64 queue slots, ring buckets of 6 keys that
continue in `ring_overflow` overflow rows — against the oracle built without bounds (liboracle_unbounded.so: 256 queue slots, 62
keys per bucket, no overflow rows needed) under loads that overflowed the OLD bounds (16 slots, 6 keys): as long as no bound is
hit, every row, every queue in drain order, every packet in flight, every bucket's keys IN PUSH ORDER and the event log agree.
Reference: queues of up to 4 096 entries (options.rs:513, base.rs:728-739), a Vec per bucket (base.rs:801-813, 1027-1042)."""
import os

import numpy as np
import pytest

from serf_amd import _ffi
from tests import _scenario as sc
from tests._oracle import ORACLE_DIR
from tests.test_oracle_unbounded import UC, UQ, US, bucket_dtype, raw, row_dtype

UNB_SO = os.path.join(ORACLE_DIR, "liboracle_unbounded.so")


@pytest.fixture(scope="module")
def unbounded():
    if not os.path.exists(UNB_SO):
        import subprocess
        subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle_unbounded.so"])
    return _ffi.SimLib(UNB_SO, prefix="osim_")


def bucket_keys(ring, X, B, node):
    """every bucket's keys in push order: its own, then its overflow rows in ascending row order (include/serf_sim.h sim_bucket)"""
    col = ring.reshape(X + B, -1)[:, node]
    out = []
    for i in range(B):
        keys = [int(k) for k in col[X + i]["keys"] if k]
        if len(keys) == len(col[X + i]["keys"]):
            for j in range(X):
                if col[j]["ltime"] == 0:
                    break
                if col[j]["ltime"] == i + 1:
                    keys += [int(k) for k in col[j]["keys"] if k]
        out.append((int(col[X + i]["ltime"]) if keys else 0, keys))
    return out


@pytest.mark.parametrize("n,fanout,P,swim,rate,X,rf", [(256, 3, 4, 0, 1.5, 4, False), (512, 3, 4, 5, 1.5, 8, False), (512, 4, 8, 4, 2.0, 8, True)])
def test_loads_beyond_the_old_bounds_equal_the_unbounded_run(oracle, unbounded, n, fanout, P, swim, rate, X, rf):
    A, Bev, Bq = 96, 32, 16
    kw = dict(fanout=fanout, event_ring=Bev, query_ring=Bq, leave_delay=6, probe_interval=swim, loss=0.01 if swim else 0.0, pkt_records=P,
              reap_interval=7 if swim else 0, reconnect_timeout=60, tombstone_timeout=80, intent_timeout=30, queue_check_interval=9,
              push_pull_interval=6 if swim else 0)
    if rf:
        kw["flags"] = _ffi.CF_BASELINE_JOINED | _ffi.CF_RANDOM_FANOUT
    b = _ffi.Sim(oracle, _ffi.make_config(n, view_slots=A, ring_overflow=X, **kw))
    u = _ffi.Sim(unbounded, _ffi.make_config(n, view_slots=0, **kw))
    ops = sc.schedule(n, 40, rate=rate, seed=n + fanout + P, max_member_subjects=40)
    for s in (b, u):
        sc.apply_schedule(s, ops)
        s.watch(3)
        s.watch(n - 2)
    deepest, most_keys = 0, 0
    for t in range(0, 72, 4):
        b.step(4)
        u.step(4)
        rb, ru = raw(b, _ffi.ARR_ROWS, row_dtype(16)), raw(u, _ffi.ARR_ROWS, row_dtype(US))
        assert rb["overflow"].sum() == 0 and ru["overflow"].sum() == 0, "the scenario is meant to stay inside the NEW bounds"
        for f in rb.dtype.names:
            if f != "susp":
                assert (rb[f] == ru[f]).all(), f"tick {t + 4}: rows.{f} differs at node {np.nonzero(rb[f] != ru[f])[0][0]}"
        qb, qu = b.dump(_ffi.ARR_QUEUE).reshape(n, _ffi.Q), u.dump(_ffi.ARR_QUEUE).reshape(n, UQ)
        assert qb.tobytes() == np.ascontiguousarray(qu[:, :_ffi.Q]).tobytes() and (qu[:, _ffi.Q:]["meta"] == 0xFFFFFFFF).all(), f"tick {t + 4}: queues differ"
        deepest = max(deepest, int((qb["meta"] != 0xFFFFFFFF).sum(axis=1).max()))
        assert b.dump(_ffi.ARR_INBOX).tobytes() == u.dump(_ffi.ARR_INBOX).tobytes(), f"tick {t + 4}: packets in flight differ"
        for which, B in ((_ffi.ARR_ERING, Bev), (_ffi.ARR_QRING, Bq)):
            kb, ku = raw(b, which, bucket_dtype(6)), raw(u, which, bucket_dtype(UC))
            for node in (0, 3, n // 2, n - 2):
                got, want = bucket_keys(kb, X, B, node), bucket_keys(ku, 0, B, node)
                assert got == want, f"tick {t + 4}: ring {which} of node {node} differs"
                most_keys = max(most_keys, max(len(k) for _, k in got))
    assert b.drain_events() == u.drain_events()
    assert deepest > _ffi.Q_HOT, f"queues were meant to go beyond the old bound of {_ffi.Q_HOT} (deepest {deepest})"
    if rate >= 1.5 and swim:
        assert most_keys > 6, "a bucket was meant to go beyond the old bound of 6 keys"


def test_a_full_set_of_overflow_rows_is_counted_not_silent(oracle):
    # 20 user events of one Lamport time with ONE overflow row: 6 + 6 keys fit, the rest are treated as seen — and counted
    n = 64
    s = _ffi.Sim(oracle, _ffi.make_config(n, fanout=3, view_slots=0, event_ring=16, query_ring=8, ring_overflow=1))
    for i in range(20):
        s.inject(1, _ffi.OP_USER_EVENT, 3 * i + 1, 500 + i, 32)
    s.step(30)
    er = s.dump(_ffi.ARR_ERING).reshape(1 + 16, n)
    assert (er[0]["ltime"] != 0).all() and s.cluster_stats()["overflow"] > 0
    keys = bucket_keys(er, 1, 16, 5)
    assert max(len(k) for _, k in keys) == 12
