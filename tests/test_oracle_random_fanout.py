"""SIM_CF_RANDOM_FANOUT on the CPU oracle: memberlist's literal kRandomNodes instead of the per-tick
bijection.  (The HIP library's twin of the mode is compared with this one in tests/test_parity_gpu.py.)"""
import numpy as np
import pytest

from serf_amd import _ffi
from tests._oracle import load_oracle

RF = _ffi.CF_BASELINE_JOINED | _ffi.CF_RANDOM_FANOUT


@pytest.fixture(scope="module")
def oracle():
    return load_oracle()


@pytest.mark.parametrize("n,fanout", [(2, 3), (3, 4), (5, 4), (5, 2)])
def test_clusters_smaller_than_the_fanout(oracle, n, fanout):
    """Fewer other nodes than `fanout`: the slots that draw no target send nothing (and, a regression: the slots a small
    cluster does not even use must not be read as targets) — a user event still reaches everybody."""
    sim = _ffi.Sim(oracle, _ffi.make_config(n, fanout=fanout, view_slots=0, event_ring=8, query_ring=8, probe_interval=2, flags=RF))
    sim.inject(1, _ffi.OP_USER_EVENT, 0, 0xABC, 40)
    sim.inject(3, _ffi.OP_CRASH, n - 1)
    sim.step(40)
    seen, up = sim.convergence(_ffi.K_EVENT, 0xABC, 1)
    assert seen == up == n - 1
    assert sim.cluster_stats()["overflow"] == 0


def test_random_fanout_spreads_one_round_slower_than_the_bijection(oracle):
    """Uniform targets give a Poisson-like in-degree: some nodes get no packet in a round.  Same cluster, same 40 rumours,
    the bijection next to it (profiles/r02_fanout_model_*.json has the 1 000-rumour histograms)."""
    n = 8192
    rounds = {}
    for flags in (_ffi.CF_BASELINE_JOINED, RF):
        sim = _ffi.Sim(oracle, _ffi.make_config(n, fanout=3, view_slots=16, event_ring=64, query_ring=8, flags=flags))
        got = []
        for i in range(40):
            key = 100 + i
            lt = int(sim.stats(7 * i).event_time)
            sim.user_event(7 * i, key, 40)
            for r in range(1, 30):
                sim.step(1)
                seen, up = sim.convergence(_ffi.K_EVENT, key, lt)
                if seen >= 0.99 * up:
                    got.append(r)
                    break
        rounds[flags] = float(np.mean(got))
        assert len(got) == 40
    assert 0.3 < rounds[RF] - rounds[_ffi.CF_BASELINE_JOINED] < 2.0, rounds


def test_no_checkpoints_in_this_mode(oracle):
    sim = _ffi.Sim(oracle, _ffi.make_config(64, fanout=3, flags=RF))
    sim.step(3)
    with pytest.raises(_ffi.SimError):
        sim.snapshot()
