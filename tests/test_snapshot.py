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
