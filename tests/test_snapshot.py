"""Checkpoint / resume (sim_snapshot / sim_restore): the image is canonical, so a run can be stopped in one
implementation of the ABI and resumed in another — oracle -> oracle on CPU, oracle <-> HIP on the GPU box.
(Reference analogue: Snapshotter, serf-core/src/snapshot.rs:117-126,228-347, per node; here per simulation.)"""
import numpy as np
import pytest

from serf_amd import _ffi
from tests import _scenario as sc

KW = dict(fanout=3, view_slots=48, event_ring=16, query_ring=8, leave_delay=6, probe_interval=4, loss=0.03,
          push_pull_interval=5, reap_interval=9, reconnect_timeout=40, tombstone_timeout=60, intent_timeout=30)


def started(lib, n=384, ticks=45, **extra):
    kw = dict(KW, **extra)
    sim = _ffi.Sim(lib, _ffi.make_config(n, **kw))
    sc.apply_schedule(sim, sc.schedule(n, 80, rate=0.7, seed=21, max_member_subjects=40))   # part of it still pending at `ticks`
    sim.step(ticks)
    return sim, kw


def resume(lib, n, kw, image):
    sim = _ffi.Sim(lib, _ffi.make_config(n, **kw))
    sim.restore(image)
    return sim


def test_oracle_snapshot_resumes_identically(oracle):
    a, kw = started(oracle)
    img = a.snapshot()
    b = resume(oracle, 384, kw, img)
    assert b.tick == a.tick and b.digest() == a.digest()
    a.step(60)
    b.step(60)
    assert a.digest() == b.digest()
    sc.assert_same_state(a, b, "resumed oracle")
    assert a.stats(7).members == b.stats(7).members


def test_restore_rejects_wrong_config_and_used_handles(oracle):
    a, kw = started(oracle)
    img = a.snapshot()
    other = _ffi.Sim(oracle, _ffi.make_config(384, **dict(kw, fanout=4)))
    with pytest.raises(_ffi.SimError):
        other.restore(img)
    with pytest.raises(_ffi.SimError):
        a.restore(img)                       # a handle that has been stepped
    with pytest.raises(_ffi.SimError):
        resume(oracle, 384, kw, img[:100])   # truncated image


@pytest.mark.gpu
@pytest.mark.parametrize("vshards", [1, 4])
def test_oracle_image_resumes_on_hip_and_back(oracle, hiplib, vshards):
    a, kw = started(oracle, n=512, vshards=vshards)
    img = a.snapshot()
    g = resume(hiplib, 512, kw, img)
    assert g.tick == a.tick and g.digest() == a.digest()
    for t in range(6):
        a.step(10)
        g.step(10)
        assert a.digest() == g.digest(), f"diverged {10 * (t + 1)} ticks after the resume"
    sc.assert_same_state(g, a, "oracle image resumed on HIP")
    img2 = g.snapshot()                      # and back: HIP image into a fresh oracle
    b = resume(oracle, 512, kw, img2)
    assert b.digest() == g.digest()
    b.step(20)
    g.step(20)
    assert b.digest() == g.digest()
    assert np.array_equal(img2[:64], a.snapshot()[:64])   # same header for the same state


@pytest.mark.gpu
def test_huge_lamport_times_and_odd_ring_sizes(oracle, hiplib):
    # state no API call can reach quickly — Lamport clocks beyond 2^32, rings whose size is not a power
    # of two (64-bit modulo on the device), the query ring past quirk Q1's 2*B horizon — is built on the
    # oracle through its test hooks, carried over as a snapshot image, and must evolve identically on HIP
    from tests._oracle import Node
    n = 200
    kw = dict(KW, event_ring=12, query_ring=7, view_slots=40)
    a = _ffi.Sim(oracle, _ffi.make_config(n, **kw))
    for node, (c, e, q) in {3: (2 ** 33 + 5, 2 ** 34 + 1, 9), 77: (5, 2 ** 32 - 2, 2 ** 40), 150: (2 ** 63, 11, 13)}.items():
        nd = Node(oracle, a, node)
        nd.set_clock(Node.CLOCK, c)
        nd.set_clock(Node.EVENT, e)
        nd.set_clock(Node.QUERY, q)
    # EventCore / QueryCore min_time (snapshot restore, event_join_ignore: base.rs:146-147, delegate.rs:531-537)
    assert oracle.t["set_min_time"](a.h, 10, 1, 3) == 0
    assert oracle.t["set_min_time"](a.h, 11, 2, 2) == 0
    sc.apply_schedule(a, sc.schedule(n, 50, rate=1.0, seed=8, max_member_subjects=30))
    for t, node in ((1, 3), (2, 77), (3, 150), (4, 3)):
        a.inject(t, _ffi.OP_USER_EVENT, node, 9000 + t, 40)
        a.inject(t, _ffi.OP_QUERY, node, 9100 + t, _ffi.F_ACK)
    img = a.snapshot()
    g = resume(hiplib, n, kw, img)
    for t in range(8):
        a.step(10)
        g.step(10)
        assert a.digest() == g.digest(), f"diverged after {10 * (t + 1)} ticks"
    sc.assert_same_state(g, a, "huge clocks")
    assert a.dump(_ffi.ARR_ROWS)["event_clock"].max() > 2 ** 34
