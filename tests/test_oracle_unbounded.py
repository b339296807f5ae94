"""Model bounds vs protocol (SURVEY.md §7 hard part 1).

The simulator bounds what the reference leaves unbounded: SIM_Q = 16 pooled queue slots (reference: three queues
of up to 4096 entries, options.rs:513, serf.rs:142-144), SIM_C = 6 keys per de-dup ring bucket (reference: a Vec,
base.rs:783-813), SIM_S = 16 suspicion timers per node, and `view_slots` active subjects instead of a member map
per node.  Every time one of those bounds bites, `sim_row.overflow` is incremented.  The claim the benchmark and
the parity tests rest on is:  a bounded run with overflow == 0 IS the unbounded run.

Checked here on the CPU oracle built twice from the same source: `liboracle.so` (product bounds) and
`liboracle_unbounded.so` (-DSIM_Q=256 -DSIM_C=62 -DSIM_S=64, run with a DENSE view: one entry per (observer,
subject)).  After every tick every piece of state is compared field by field: clocks, flags, counters, timers, the
queue in drain order, every view entry (slotted subjects against their dense entries, all other dense entries
against the baseline), every ring bucket, and the event log of watched nodes.
"""
import os

import numpy as np
import pytest

from serf_amd import _ffi
from tests import _scenario as sc
from tests._oracle import ORACLE_DIR, load_oracle

UQ, UC, US = 256, 62, 64
UNB_SO = os.path.join(ORACLE_DIR, "liboracle_unbounded.so")


def row_dtype(s):
    return np.dtype([("clock", "<u8"), ("event_clock", "<u8"), ("query_clock", "<u8"), ("event_min", "<u8"), ("query_min", "<u8"),
                     ("flags", "<u4"), ("inc", "<u4"), ("n_known", "<u4"), ("n_failed", "<u4"), ("n_left", "<u4"), ("next_seq", "<u4"),
                     ("overflow", "<u4"), ("susp_next", "<u4"), ("awareness", "<u4"), ("reap_next", "<u4"), ("susp", "<u2", (s,))])


def bucket_dtype(c):
    return np.dtype([("ltime", "<u8"), ("keys", "<u4", (c,))])


@pytest.fixture(scope="module")
def unbounded():
    if not os.path.exists(UNB_SO):
        import subprocess
        subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle_unbounded.so"])
    return _ffi.SimLib(UNB_SO, prefix="osim_")


def raw(sim, which, dtype):
    return sim.dump(which, dtype)


def compare(b, u, n, A, Bev, Bq, what):
    """b: bounded sim (view_slots = A), u: unbounded sim (dense)."""
    rb, ru = raw(b, _ffi.ARR_ROWS, row_dtype(16)), raw(u, _ffi.ARR_ROWS, row_dtype(US))
    assert rb["overflow"].sum() == 0, f"{what}: the bounded run hit a bound — pick a lighter scenario"
    assert ru["overflow"].sum() == 0
    for f in rb.dtype.names:
        if f != "susp":
            assert (rb[f] == ru[f]).all(), f"{what}: rows.{f} differs at node {np.nonzero(rb[f] != ru[f])[0][0]}"
    slot_of = b.dump(_ffi.ARR_SLOTMAP)
    # suspicion timers name view slots + 1: translate the bounded run's slots to subjects (dense: slot == subject)
    subj_of = np.full(A + 1, 0xFFFFFFFF, np.int64)
    for s_, a in enumerate(slot_of):
        if a != 0xFFFFFFFF:
            subj_of[a] = s_
    sb = rb["susp"].astype(np.int64)
    tb = np.where(sb > 0, subj_of[np.maximum(sb - 1, 0)] + 1, 0)
    assert (tb == ru["susp"][:, :16]).all() and (ru["susp"][:, 16:] == 0).all(), f"{what}: suspicion timer lists differ"
    qb = b.dump(_ffi.ARR_QUEUE).reshape(n, _ffi.Q)
    qu = u.dump(_ffi.ARR_QUEUE).reshape(n, UQ)
    assert qb.tobytes() == np.ascontiguousarray(qu[:, :_ffi.Q]).tobytes(), f"{what}: queues differ"
    assert (qu[:, _ffi.Q:]["meta"] == 0xFFFFFFFF).all(), f"{what}: the unbounded queue holds more than {_ffi.Q} entries somewhere"
    assert b.dump(_ffi.ARR_INBOX).tobytes() == u.dump(_ffi.ARR_INBOX).tobytes(), f"{what}: packets in flight differ"
    vb = b.dump(_ffi.ARR_VIEW).reshape(A, n)
    vu = u.dump(_ffi.ARR_VIEW).reshape(n, n)   # [subject][observer]
    base = np.zeros(1, _ffi.VIEW_DTYPE)
    base["ltime"], base["bits"] = 1, 1 | (_ffi.STATUS_ALIVE << 1)
    for s_ in range(n):
        a = slot_of[s_]
        want = vb[a] if a != 0xFFFFFFFF else np.broadcast_to(base, (n,))
        if vu[s_].tobytes() != np.ascontiguousarray(want).tobytes():
            i = sc.first_diff(vu[s_], np.ascontiguousarray(want))
            raise AssertionError(f"{what}: view of subject {s_} (slot {a}) differs at observer {i}: {vu[s_][i]} vs {want[i]}")
    for which, B in ((_ffi.ARR_ERING, Bev), (_ffi.ARR_QRING, Bq)):
        kb, ku = raw(b, which, bucket_dtype(6)), raw(u, which, bucket_dtype(UC))
        assert (kb["ltime"] == ku["ltime"]).all() and (kb["keys"] == ku["keys"][:, :6]).all() and (ku["keys"][:, 6:] == 0).all(), \
            f"{what}: ring {which} differs"


@pytest.mark.parametrize("n,fanout,swim,loss", [(128, 3, 0, 0.0), (128, 3, 5, 0.02), (1024, 4, 4, 0.01), (4096, 4, 5, 0.0)])
def test_zero_overflow_run_equals_unbounded_run(oracle, unbounded, n, fanout, swim, loss):
    A, Bev, Bq = 96, 32, 16
    kw = dict(fanout=fanout, event_ring=Bev, query_ring=Bq, leave_delay=6, probe_interval=swim, loss=loss,
              reap_interval=7 if swim else 0, reconnect_timeout=60, tombstone_timeout=80, intent_timeout=30,
              queue_check_interval=9, push_pull_interval=6 if swim else 0)
    b = _ffi.Sim(oracle, _ffi.make_config(n, view_slots=A, **kw))
    u = _ffi.Sim(unbounded, _ffi.make_config(n, view_slots=0, **kw))
    ticks = 160 if n <= 1024 else 70
    ops = sc.schedule(n, ticks * 2 // 3, rate=0.22 if swim else 0.3, seed=n + swim, max_member_subjects=A // 2)
    for s in (b, u):
        sc.apply_schedule(s, ops)
        s.watch(3)
        s.watch(n - 2)
    every = 1 if n <= 128 else (8 if n <= 1024 else 35)
    for t in range(0, ticks, every):
        b.step(every)
        u.step(every)
        compare(b, u, n, A, Bev, Bq, f"n={n} swim={swim} tick {t + every}")
    assert b.drain_events() == u.drain_events()
    rows = raw(b, _ffi.ARR_ROWS, row_dtype(16))
    if swim:
        assert rows["n_failed"].sum() + rows["n_left"].sum() > 0, "scenario should exercise the failure detector"


def test_overflow_is_what_separates_them(oracle, unbounded):
    # the converse: push the bounded model over its queue bound and the two runs part ways — and `overflow` says so
    n, A = 256, 64
    kw = dict(fanout=3, event_ring=16, query_ring=8)
    b = _ffi.Sim(oracle, _ffi.make_config(n, view_slots=A, **kw))
    u = _ffi.Sim(unbounded, _ffi.make_config(n, view_slots=0, **kw))
    ops = sc.schedule(n, 30, rate=4.0, seed=9, max_member_subjects=20)
    for s in (b, u):
        sc.apply_schedule(s, ops)
    b.step(60)
    u.step(60)
    rb, ru = raw(b, _ffi.ARR_ROWS, row_dtype(16)), raw(u, _ffi.ARR_ROWS, row_dtype(US))
    assert rb["overflow"].sum() > 0 and ru["overflow"].sum() == 0
    assert u.cluster_stats()["max_queue"] > _ffi.Q or (rb["event_clock"] != ru["event_clock"]).any() or \
        b.dump(_ffi.ARR_INBOX).tobytes() != u.dump(_ffi.ARR_INBOX).tobytes()
