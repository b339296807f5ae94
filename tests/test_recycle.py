"""View-slot recycling (SIMSPEC §2.6): churn over more subjects than there are view slots.

A subject whose entry has settled — the same at every running node, nothing in flight about it, no timer on it — gives
its slot back (reference analogue: a forgotten member, erase_node! base.rs:499-518 / Reaper base.rs:521-553); scheduled
operations take their slot when they execute.  CPU side here: the oracle alone, and the sharded path over gloo against
the single-process run; the HIP parity is in tests/test_parity_gpu.py."""
import os
import sys

import numpy as np
import pytest

from serf_amd import _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def churn_ops(n, n_churn, every, down, seed=3, start=5):
    """crash + revive of `n_churn` distinct nodes, one every `every` ticks, each down for `down` ticks, plus a user
    event per churn step (background gossip)."""
    rng = np.random.default_rng(seed)
    nodes = rng.choice(n, n_churn, replace=False)
    ops = []
    for i, node in enumerate(nodes.tolist()):
        t = start + i * every
        ops.append((t, _ffi.OP_CRASH, node, 0, 0))
        ops.append((t + down, _ffi.OP_REVIVE, node, 0, 0))
        ops.append((t + 1, _ffi.OP_USER_EVENT, int(rng.integers(0, n)), 1000 + i, 48))
    ops.sort(key=lambda o: o[0])
    return ops


KW = dict(fanout=3, view_slots=16, event_ring=64, query_ring=16, probe_interval=2, suspicion_mult=3, suspicion_max_mult=2,
          push_pull_interval=8, recycle_interval=16)


def test_churn_over_more_subjects_than_slots(oracle):
    n = 512
    sim = _ffi.Sim(oracle, _ffi.make_config(n, **KW))
    ops = churn_ops(n, 80, every=6, down=4)   # 80 subjects through 16 slots
    for o in ops:
        sim.inject(*o)
    sim.step(80 * 6 + 200)
    cs = sim.cluster_stats()
    assert cs["ops_dropped"] == 0, "every churned node found a slot"
    assert cs["slots_recycled"] >= 64 and cs["slots_in_use"] <= 16
    assert cs["overflow"] == 0
    assert cs["up"] == n
    st, _ = sim.members(7)
    assert (st == _ffi.STATUS_ALIVE).all(), "everybody is back and known alive"
    # the incarnations the churned nodes refuted with live on in the baseline entries of the recycled subjects
    rows = sim.dump(_ffi.ARR_ROWS)
    assert (rows["inc"] > 0).sum() > 0


def test_without_recycling_the_same_churn_runs_out_of_slots(oracle):
    n = 512
    kw = dict(KW, recycle_interval=0)
    sim = _ffi.Sim(oracle, _ffi.make_config(n, **kw))
    for o in churn_ops(n, 80, every=6, down=4):
        sim.inject(*o)
    sim.step(80 * 6 + 50)
    cs = sim.cluster_stats()
    assert cs["ops_dropped"] > 0 and cs["slots_in_use"] == 16 and cs["slots_recycled"] == 0


def test_immediate_operation_still_reports_enoslot(oracle):
    sim = _ffi.Sim(oracle, _ffi.make_config(64, view_slots=2, recycle_interval=4))
    sim.leave(1)
    sim.leave(2)
    with pytest.raises(_ffi.SimError) as ei:
        sim.leave(3)
    assert ei.value.code == _ffi.ENOSLOT


def test_snapshot_carries_the_slot_bookkeeping(oracle):
    n = 256
    a = _ffi.Sim(oracle, _ffi.make_config(n, **KW))
    b = _ffi.Sim(oracle, _ffi.make_config(n, **KW))
    ops = churn_ops(n, 40, every=5, down=3)
    for o in ops:
        a.inject(*o)
    a.step(130)
    b.restore(a.snapshot())
    for _ in range(6):
        a.step(25)
        b.step(25)
        assert a.digest() == b.digest()
    assert a.cluster_stats() == b.cluster_stats()


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    from serf_amd.shard import ShardedSim
    from tests._oracle import load_oracle

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lib = load_oracle()
        n = 1024
        kw = dict(KW, view_slots=24)
        sh = ShardedSim(lib, n, torch.device("cpu"), chunks=2, **kw)
        ref = _ffi.Sim(lib, _ffi.make_config(n, vshards=world, chunks=2, **kw))
        for o in churn_ops(n, 70, every=5, down=4):
            sh.inject(*o)
            ref.inject(*o)
        m = n // world
        lo = rank * m
        for t in range(0, 450, 10):
            sh.step(10)
            ref.step(10)
            for which in (_ffi.ARR_ROWS, _ffi.ARR_QUEUE):
                a, b = sh.sim.dump(which), ref.dump(which)
                per = len(b) // n
                assert a.tobytes() == b[lo * per:(lo + m) * per].tobytes(), f"rank {rank} array {which} differs at tick {t + 10}"
            a = sh.sim.dump(_ffi.ARR_VIEW).reshape(24, m)
            b = ref.dump(_ffi.ARR_VIEW).reshape(24, n)[:, lo:lo + m]
            assert a.tobytes() == np.ascontiguousarray(b).tobytes(), f"rank {rank} view differs at tick {t + 10}"
            assert (sh.sim.dump(_ffi.ARR_SLOTMAP) == ref.dump(_ffi.ARR_SLOTMAP)).all()
        cs, cr = sh.sim.cluster_stats(), ref.cluster_stats()
        assert cs["slots_recycled"] == cr["slots_recycled"] >= 46 and cs["ops_dropped"] == cr["ops_dropped"] == 0
        q.put((rank, "ok"))
    except BaseException as e:  # noqa: BLE001
        q.put((rank, repr(e)))
        raise
    finally:
        dist.destroy_process_group()


def test_sharded_recycling_matches_single_process():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    from tests._scenario import free_port
    port = free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
    res = sorted(q.get(timeout=5) for _ in procs)
    assert res == [(0, "ok"), (1, "ok")], res


def stale_timer_ops(x=100, y=200):
    """Y goes down and is suspected by everybody; X goes down holding a suspicion timer on Y; Y comes back, refutes, its
    entry settles and its view slot is recycled; X comes back BEFORE the slot is handed out again: its timer names a slot
    that has no subject any more."""
    return [(5, _ffi.OP_CRASH, y, 0, 0), (22, _ffi.OP_CRASH, x, 0, 0), (26, _ffi.OP_REVIVE, y, 0, 0), (90, _ffi.OP_REVIVE, x, 0, 0)]


def test_timer_on_a_slot_recycled_while_its_holder_was_down(oracle):
    # round 3: found by the 4 x 64 Ki sharded GPU test — the HIP path used subject_of[slot] = NOSLOT as a subject
    # (slot_of[0xFFFFFFFF]); the rule now, in both implementations: such a timer is dropped
    n, x, y = 512, 100, 200
    sim = _ffi.Sim(oracle, _ffi.make_config(n, **dict(KW, view_slots=8)))
    for o in stale_timer_ops(x, y):
        sim.inject(*o)
    sim.step(22)
    rows = sim.dump(_ffi.ARR_ROWS)
    slot_y = int(sim.dump(_ffi.ARR_SLOTMAP)[y])
    assert slot_y + 1 in rows["susp"][x], "X holds a timer on Y when it goes down"
    sim.step(68)
    assert int(sim.dump(_ffi.ARR_SLOTMAP)[y]) == 0xFFFFFFFF, "Y's slot was recycled while X was down"
    assert slot_y + 1 in sim.dump(_ffi.ARR_ROWS)["susp"][x], "X is down: nobody touched its timers"
    sim.step(12)   # X is back (tick 90) and has looked at its timers
    assert slot_y + 1 not in sim.dump(_ffi.ARR_ROWS)["susp"][x]
    st, _ = sim.members(x)
    assert st[y] == _ffi.STATUS_ALIVE, "X sees Y through the baseline entry the recycling pass left"
    assert sim.cluster_stats()["up"] == n
