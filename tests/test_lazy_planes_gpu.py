"""View entries and ring buckets that get their memory on demand (include/serf_sim.h sim_resident_planes;
serf_sim_host.inc LazyPlanes): the HIP library with planes mapped as slots are handed out / Lamport times admitted must be the
library with whole arrays — and the oracle, which keeps whole arrays — bit for bit: digests (an unmapped plane enters as the
zeros it stands for), dumps, images; a record that comes in over the byte boundary with a Lamport time far ahead, a question
about a time nobody has used, a push-pull over a ring that is mostly unmapped, a restore into a fresh handle.
128 Ki nodes: a plane (128 Ki x 16 bytes) is one 2 MiB mapping granule — the smallest cluster the mechanism is on for."""
import os

import pytest

from serf_amd import _ffi
from tests import _scenario as sc

pytestmark = pytest.mark.gpu

N = 1 << 17
KW = dict(fanout=4, view_slots=64, event_ring=64, query_ring=32, probe_interval=5, loss=0.01, push_pull_interval=12, leave_delay=6,
          reap_interval=15, queue_check_interval=30, recycle_interval=25)


def _create(hiplib, eager, **kw):
    old = os.environ.get("SERF_SIM_EAGER")
    os.environ["SERF_SIM_EAGER"] = "1" if eager else "0"
    try:
        return _ffi.Sim(hiplib, _ffi.make_config(N, **kw))
    finally:
        if old is None:
            del os.environ["SERF_SIM_EAGER"]
        else:
            os.environ["SERF_SIM_EAGER"] = old


@pytest.mark.parametrize("model", ["bijection", "krandomnodes"])
def test_lazy_planes_equal_whole_arrays_and_the_oracle(oracle, hiplib, model):
    kw = dict(KW)
    if model == "krandomnodes":
        kw["flags"] = _ffi.CF_BASELINE_JOINED | _ffi.CF_RANDOM_FANOUT
    lazy, whole, o = _create(hiplib, False, **kw), _create(hiplib, True, **kw), _ffi.Sim(oracle, _ffi.make_config(N, **kw))
    r0 = lazy.resident_planes()
    if r0["view"][0] == r0["view"][1]:
        pytest.skip("the mapping granularity of this device does not divide a plane of 128 Ki nodes: nothing is lazy here")
    assert r0["view"][0] == 0 and r0["event_ring"][0] < r0["event_ring"][1] and r0["bytes_per_plane"] == N * 32
    rw = whole.resident_planes()
    assert rw["view"][0] == rw["view"][1] and rw["event_ring"][0] == rw["event_ring"][1]
    assert lazy.digest() == whole.digest() == o.digest()          # nothing mapped: every view plane enters as zeros
    ops = sc.schedule(N, 40, rate=0.6, seed=5, max_member_subjects=20)
    for x in (lazy, whole, o):
        for op in ops:
            x.inject(*op)
    for t in range(0, 60, 6):
        if t == 12:   # a record from outside, Lamport time far ahead of every clock: its plane gets memory now
            for x in (lazy, whole, o):
                x.inject_record(x.tick, 77, 0x5000, ((63 - 3) << 18) | (_ffi.K_EVENT << 4), 57)
        for x in (lazy, whole, o):
            x.step(6)
        dl, dw, do = lazy.digest(), whole.digest(), o.digest()
        assert dl == dw == do, f"after tick {t + 5}: lazy {dl} whole {dw} oracle {do}"
    r1 = lazy.resident_planes()
    assert 0 < r1["view"][0] < r1["view"][1], r1                  # some slots were handed out, most never
    assert r1["event_ring"][0] >= 58                              # the time-57 record's plane
    sc.assert_same_state(lazy, o, "lazy planes vs oracle")        # dumps: the unmapped planes come back as zeros
    # a question about a Lamport time nobody has used (its plane may have no memory yet), next to one that was used
    assert lazy.convergence(_ffi.K_QUERY, 999, 31) == o.convergence(_ffi.K_QUERY, 999, 31)
    assert lazy.convergence_many([(_ffi.K_EVENT, 0x5000, 57), (_ffi.K_EVENT, 5, 63)]) == o.convergence_many([(_ffi.K_EVENT, 0x5000, 57), (_ffi.K_EVENT, 5, 63)])
    # the image of the lazy handle restores into a fresh lazy handle (memory for the planes the image has anything in), into a
    # whole one and into the oracle; all continue alike
    img = lazy.snapshot()
    assert bytes(img) == bytes(o.snapshot()), "the images differ"
    fresh, fresh_whole, fo = _create(hiplib, False, **kw), _create(hiplib, True, **kw), _ffi.Sim(oracle, _ffi.make_config(N, **kw))
    for x in (fresh, fresh_whole, fo):
        x.restore(img)
    rf = fresh.resident_planes()
    assert rf["view"][0] < rf["view"][1] and rf["view"][0] >= 1
    more = [(op[0] + 60, *op[1:]) for op in sc.schedule(N, 10, rate=0.8, seed=6, max_member_subjects=4)]
    for x in (lazy, fresh, fresh_whole, fo):
        for op in more:
            x.inject(*op)
        x.step(20)
    assert lazy.digest() == fresh.digest() == fresh_whole.digest() == fo.digest()
    for x in (lazy, whole, o, fresh, fresh_whole, fo):
        x.close()


def test_restore_gives_memory_to_every_slot_that_was_handed_out(oracle, hiplib):
    # a slot that was handed out may hold nothing but zeros (a cluster nobody has joined yet: the baseline entry is empty) — the image's
    # view section is all zeros then, and the restored handle must give the slot's plane memory all the same: the tick kernel walks it
    kw = dict(KW, flags=0)
    a, o = _create(hiplib, False, **kw), _ffi.Sim(oracle, _ffi.make_config(N, **kw))
    if a.resident_planes()["view"][1] == a.resident_planes()["view"][0]:
        pytest.skip("nothing is lazy on this device")
    for x in (a, o):
        x.leave(9)              # executes in tick 0: subject 9 gets slot 0 now
        x.inject(3, _ffi.OP_CRASH, 11)
    img = a.snapshot()
    assert bytes(img) == bytes(o.snapshot())
    b = _create(hiplib, False, **kw)
    b.restore(img)
    assert b.resident_planes()["view"][0] >= 1
    for x in (a, b, o):
        x.step(12)
    assert a.digest() == b.digest() == o.digest()
    for x in (a, b, o):
        x.close()
