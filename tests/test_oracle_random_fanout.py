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


@pytest.mark.parametrize("pkt", [4, 12])
def test_checkpoint_and_resume_in_this_mode(oracle, pkt):
    """The image holds the packets in flight in their senders' cells; where each one goes is a function of (seed, tick,
    sender) and is drawn again on restore: the resumed run is the uninterrupted one."""
    kw = dict(fanout=3, view_slots=16, event_ring=16, query_ring=8, probe_interval=3, loss=0.03, pkt_records=pkt, flags=RF)
    a = _ffi.Sim(oracle, _ffi.make_config(700, **kw))
    for i in range(30):
        a.inject(1 + i % 7, _ffi.OP_USER_EVENT, 13 * i, 900 + i, 40)
    a.inject(3, _ffi.OP_CRASH, 5)
    a.step(9)
    img = a.snapshot()
    b = _ffi.Sim(oracle, _ffi.make_config(700, **kw))
    b.restore(img)
    assert a.digest() == b.digest()
    for _ in range(6):
        a.step(4)
        b.step(4)
        assert a.digest() == b.digest()


def test_the_request_bound_of_one_tick_is_counted(oracle):
    """With a random in-degree there is no maximum to the broadcasts the packets of one tick can ask for: requests beyond
    f * P + SIM_S + 1 + SIM_RF_PEND_EXTRA are dropped and counted (the HIP library parks them in an array of that many rows).
    Far out of reach of anything but a constructed case — here: nothing is dropped under a heavy load."""
    sim = _ffi.Sim(oracle, _ffi.make_config(512, fanout=4, view_slots=64, event_ring=64, query_ring=8, pkt_records=16, flags=RF))
    for i in range(200):
        sim.inject(1 + i // 20, _ffi.OP_USER_EVENT, i % 512, 5000 + i, 30)
    sim.step(30)
    rows = sim.dump(_ffi.ARR_ROWS)
    assert int(rows["overflow"].sum()) == sim.cluster_stats()["overflow"]
