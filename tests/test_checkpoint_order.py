"""A checkpoint must not change what a run computes.  sim_snapshot consumes the pending lists of slot-less
suspicions (SIMSPEC §2.7) and puts them into the schedule as SIM_OP_SUSPECT operations of the ticks they are due in — ahead of
the replay at sim_step_begin, which puts them BEHIND everything the caller has scheduled for that tick.  An operation the caller
schedules for the same tick after the checkpoint used to run behind the suspicions instead of before them (a different view
slot for its subject, a different queue order at a node both touch).  The order within a tick is now: the caller's operations,
then the replayed suspicions — whenever the lists reached the schedule."""
import pytest

from serf_amd import _ffi
from tests._oracle import load_oracle

N = 2048
KW = dict(fanout=3, view_slots=512, event_ring=16, query_ring=8, probe_interval=3, suspicion_mult=4, suspicion_max_mult=3,
          loss=0.02, indirect_checks=2, recycle_interval=0, push_pull_interval=10)


def _run(lib, checkpoint_every):
    sim = _ffi.Sim(lib, _ffi.make_config(N, **KW))
    digests, suspects = [], 0
    for t in range(70):
        sim.step(1)                                    # executes tick t
        if checkpoint_every and t % checkpoint_every == 0:
            sim.snapshot()                             # taken and thrown away
        # the caller schedules an operation that needs a view slot for the tick AFTER the next one — the tick the suspicions of tick t
        # are due in (an operation of a later tick gets its slot when it executes: the order of a tick's operations decides who gets which)
        if t % 3 == 0:
            sim.inject(t + 2, _ffi.OP_SET_TAGS, 100 + 7 * t, 1 + t % 3)
        digests.append(sim.digest())
    cs = sim.cluster_stats()
    suspects = cs["slots_in_use"] - 24
    sim.close()
    return digests, suspects, cs


def _check(lib):
    plain, extra, cs = _run(lib, 0)
    assert extra > 3, f"the scenario must produce slot-less suspicions (view slots beyond the caller's 24: {extra})"
    assert cs["ops_dropped"] == 0
    for every in (1, 3):
        with_cp, _, _ = _run(lib, every)
        bad = [t for t, (a, b) in enumerate(zip(plain, with_cp)) if a != b]
        assert not bad, f"a checkpoint every {every} ticks changed the run from tick {bad[0]} on (arrays {[i for i in range(8) if plain[bad[0]][i] != with_cp[bad[0]][i]]})"
    return plain


def test_a_checkpoint_does_not_change_the_run_oracle():
    _check(load_oracle())


@pytest.mark.gpu
def test_a_checkpoint_does_not_change_the_run_hip(hiplib):
    assert _check(hiplib) == _check(load_oracle())
