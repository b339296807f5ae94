"""Parity tests proper: the HIP path (through the C ABI) against the CPU oracle on the same seeded
inputs — bit-exact, every state array, every tick."""
import os

import numpy as np
import pytest

from serf_amd import _ffi
from tests import _scenario as sc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def pair(oracle, hiplib, n, **kw):
    cfg = _ffi.make_config(n, **kw)
    return _ffi.Sim(hiplib, cfg), _ffi.Sim(oracle, _ffi.make_config(n, **kw))


def test_backend_is_hip(hiplib):
    assert hiplib.backend_name() == "hip-gfx950"
    assert hiplib.abi_version() == 15


@pytest.mark.parametrize("swim", [0, 5, 2])
@pytest.mark.parametrize("n,fanout,dense", [(128, 3, True), (100, 3, True), (257, 4, False), (1024, 4, False), (2, 3, True), (1, 3, True), (5, 4, True)])
def test_full_state_every_tick_small(oracle, hiplib, n, fanout, dense, swim):
    # config 1 shape (128 nodes, fan-out 3) and ragged sizes; every array compared after every tick;
    # swim = probe interval in ticks (0: serf layer only)
    kw = dict(fanout=fanout, view_slots=0 if dense else 64, event_ring=16, query_ring=8, leave_delay=6,
              probe_interval=swim, suspicion_mult=3 if swim == 2 else 4, suspicion_max_mult=2 if swim == 2 else 6,
              reap_interval=7 if swim else 0, reconnect_timeout=25, tombstone_timeout=40, intent_timeout=20,
              queue_check_interval=9 if swim == 2 else 0, min_queue_depth=3 if swim == 2 else 0,
              push_pull_interval=6 if swim else 0)
    g, o = pair(oracle, hiplib, n, **kw)
    ops = sc.schedule(n, 60, rate=0.6, seed=n * 7 + fanout, max_member_subjects=min(n // 2, 40))
    sc.apply_schedule(g, ops)
    sc.apply_schedule(o, ops)
    for t in range(90 if not swim else 160):
        g.step(1)
        o.step(1)
        if g.digest() != o.digest():
            sc.assert_same_state(g, o, f"n={n} tick {t}")
            raise AssertionError(f"digest differs after tick {t} but the arrays agree")
        if t % 10 == 0 or t < 5:
            sc.assert_same_state(g, o, f"n={n} tick {t}")
    sc.assert_same_state(g, o, f"n={n} final")


@pytest.mark.parametrize("rf", [False, True])
@pytest.mark.parametrize("n,slots,delay", [(1024, 64, 5), (300, 0, 2), (4096, 96, 11)])
def test_handle_prunes_wait_parity(oracle, hiplib, n, slots, delay, rf):
    # SIM_CF_PRUNE_DELAY (serf/base.rs:1628-1653): pruning leave intents about Alive / Leaving members erase leave_delay ticks later — notes on the
    # tick's request list, SIM_OP_PRUNE operations on the schedule; a load with many forced removals (a third of them pruning), sparse and dense views,
    # recycling, push-pull, loss; checkpoints in the middle carry the lists in flight and the pending erases (HIP image -> both, oracle image -> both)
    kw = dict(fanout=4, view_slots=slots, event_ring=16, query_ring=8, leave_delay=delay, probe_interval=3, loss=0.02, push_pull_interval=8,
              reap_interval=7, reconnect_timeout=30, tombstone_timeout=45, intent_timeout=20, recycle_interval=10 if slots else 0, prune_delay=True,
              flags=_ffi.CF_BASELINE_JOINED | (_ffi.CF_RANDOM_FANOUT if rf else 0))
    g, o = pair(oracle, hiplib, n, **kw)
    ops = sc.schedule(n, 70, rate=0.9, seed=n + delay, mix=(0.3, 0.1, 0.15, 0.35, 0.1), max_member_subjects=min(n // 3, 40))
    sc.apply_schedule(g, ops)
    sc.apply_schedule(o, ops)
    for t in range(120):
        g.step(1)
        o.step(1)
        assert g.digest() == o.digest(), f"digest differs after tick {t}"
        if t % 13 == 0:
            sc.assert_same_state(g, o, f"tick {t}")
        if t in (31, 58):  # resume both from one image: the HIP library's, then the oracle's
            img = (g if t == 31 else o).snapshot()
            g.close(); o.close()
            g = _ffi.Sim(hiplib, _ffi.make_config(n, **kw)); o = _ffi.Sim(oracle, _ffi.make_config(n, **kw))
            g.restore(img); o.restore(img)
            assert g.digest() == o.digest(), f"restored images differ at tick {t}"
    sc.assert_same_state(g, o, "final")
    assert g.cluster_stats()["ops_dropped"] == o.cluster_stats()["ops_dropped"]


@pytest.mark.parametrize("n,vshards,chunks,fanout", [(4096, 4, 2, 4), (2048, 4, 0, 4), (4096, 1, 0, 3), (640, 2, 0, 2), (8192, 2, 2, 4)])
def test_packets_kept_at_the_sender_virtual_shards_and_chunks(oracle, hiplib, n, vshards, chunks, fanout):
    # One handle, every shape of the fan-out map (virtual shards, sender chunks, 64-node blocks and the B = 1
    # fallback): the product keeps each DISTINCT packet once at its sender and the receiver fetches it through the
    # map's inverse; the canonical inbox (dump, digest) is turned inside out from that on demand.  Packet loss and a
    # load that makes the four packets of a node differ exercise the map word; a checkpoint taken in the middle goes
    # through the image's receiver-indexed inbox and back to the senders.
    kw = dict(fanout=fanout, vshards=vshards, chunks=chunks, view_slots=64, event_ring=16, query_ring=8, loss=0.03,
              probe_interval=4, push_pull_interval=10, leave_delay=6)
    g, o = pair(oracle, hiplib, n, **kw)
    ops = sc.schedule(n, 50, rate=2.0, seed=n + vshards, max_member_subjects=30)
    sc.apply_schedule(g, ops)
    sc.apply_schedule(o, ops)
    for t in range(70):
        g.step(1)
        o.step(1)
        assert g.digest() == o.digest(), f"digest differs after tick {t}"
        if t % 9 == 0:
            sc.assert_same_state(g, o, f"tick {t}")
        if t == 33:  # resume both from the HIP image
            img = g.snapshot()
            g.close()
            g = _ffi.Sim(hiplib, _ffi.make_config(n, **kw))
            g.restore(img)
            assert g.digest() == o.digest(), "restored image differs"
    sc.assert_same_state(g, o, "final")


@pytest.mark.parametrize("seed", list(range(48)))
def test_random_configurations(oracle, hiplib, seed):
    # a seeded sweep over the configuration space (size — ragged sizes included —, fan-out, virtual shards and chunks,
    # view slots or dense views, ring sizes that are and are not powers of two, loss, load from idle to overload,
    # SWIM / push-pull / reaper / queue checker / recycling on or off): digests after every tick, every array at the end
    rng = np.random.default_rng(1000 + seed)
    v = int(rng.choice([1, 1, 2, 4]))
    c = int(rng.choice([0, 0, 2, 4])) if v > 1 or rng.random() < 0.3 else 0
    unit = v * v * max(c, 1)
    n = int(rng.choice([96, 200, 512, 1000, 2048, 4096, 8192]))
    n = max(unit, n // unit * unit) if (v > 1 or c) else n  # shards and chunks need equal slabs
    dense = n <= 600 and rng.random() < 0.4
    swim = int(rng.choice([0, 2, 3, 5]))
    kw = dict(fanout=int(rng.integers(1, 5)), vshards=v, chunks=c, view_slots=0 if dense else int(rng.choice([16, 48, 64])),
              event_ring=int(rng.choice([8, 12, 16, 64])), query_ring=int(rng.choice([8, 10, 32])), leave_delay=int(rng.integers(3, 9)),
              loss=float(rng.choice([0.0, 0.0, 0.02, 0.1])), probe_interval=swim, reap_interval=int(rng.choice([0, 5, 11])) if swim else 0,
              reconnect_timeout=20, tombstone_timeout=30, intent_timeout=15,
              queue_check_interval=int(rng.choice([0, 7])), min_queue_depth=int(rng.choice([0, 2])),
              push_pull_interval=int(rng.choice([0, 4, 9])), recycle_interval=int(rng.choice([0, 6])) if not dense else 0)
    # memberlist options behind flags (round 3): awareness-scaled probe interval, gossip_to_the_dead_time (drawn last: the
    # configurations above are the ones the sweep has always run)
    kw.update(awareness_probe=bool(swim and rng.random() < 0.5), gossip_to_the_dead=int(rng.choice([0, 0, 2, 8])) if swim else 0,
              join_sync=bool(rng.random() < 0.5))
    kw.update(reconnect_interval=int(rng.choice([0, 0, 3, 7])) if swim else 0)  # Reconnector (drawn after everything else)
    kw.update(tcp_fallback=bool(swim and rng.random() < 0.4), nacks=bool(swim and rng.random() < 0.4))  # memberlist's fallback ping / nack accounting
    try:
        g, o = pair(oracle, hiplib, n, **kw)
    except _ffi.SimError:
        pytest.skip(f"configuration rejected by both sides: n={n} {kw}")
    rate = float(rng.choice([0.2, 0.8, 2.5]))
    ops = sc.schedule(n, 40, rate=rate, seed=seed, max_member_subjects=12 if not dense else min(n // 2, 30))
    sc.apply_schedule(g, ops)
    sc.apply_schedule(o, ops)
    for t in range(64):
        g.step(1)
        o.step(1)
        if g.digest() != o.digest():
            sc.assert_same_state(g, o, f"seed {seed} n={n} {kw} tick {t}")
            raise AssertionError(f"digest differs after tick {t} but the arrays agree")
    sc.assert_same_state(g, o, f"seed {seed} final")
    assert g.cluster_stats()["ops_dropped"] == o.cluster_stats()["ops_dropped"]


@pytest.mark.parametrize("seed", list(range(32)))
def test_random_fanout_kRandomNodes(oracle, hiplib, seed):
    # SIM_CF_RANDOM_FANOUT: memberlist's literal kRandomNodes (App. B.2) — uniform targets, no replacement, variable in-degree —
    # in the PRODUCT: the tick's fan-out graph is built two ticks ahead by the library's own two-level bucket sort (rf_*
    # kernels), the packets stay in their senders' cells and every receiver pulls the ones its CSR row names.  Seeded sweep: tiny clusters (fewer other nodes than the fan-out: slots without a target), ragged and block-sized
    # ones, dense and slotted views, SWIM / loss / gossip_to_the_dead / push-pull / reaper / recycling / Reconnector on or off,
    # packets of 1 - 4 pages (seeds 16 ..); digests after every tick, every array at the end, a checkpoint in the middle that
    # goes through the canonical (sender-indexed) inbox and back, both ways.
    rng = np.random.default_rng(9000 + seed)
    n = int([2, 3, 5, 96, 200, 1000, 2048, 4096][seed % 8])
    dense = n <= 600 and rng.random() < 0.5
    swim = int(rng.choice([0, 2, 3]))
    kw = dict(fanout=int(rng.integers(1, 5)), view_slots=0 if dense else int(rng.choice([16, 48])),
              event_ring=int(rng.choice([8, 12, 16])), query_ring=int(rng.choice([8, 10])), leave_delay=int(rng.integers(3, 9)),
              loss=float(rng.choice([0.0, 0.02, 0.1])), probe_interval=swim, reap_interval=int(rng.choice([0, 5])) if swim else 0,
              reconnect_timeout=20, tombstone_timeout=30, intent_timeout=15, queue_check_interval=int(rng.choice([0, 7])),
              push_pull_interval=int(rng.choice([0, 4])), recycle_interval=int(rng.choice([0, 6])) if not dense else 0,
              gossip_to_the_dead=int(rng.choice([0, 2, 8])) if swim else 0, reconnect_interval=int(rng.choice([0, 3])) if swim else 0,
              join_sync=bool(rng.random() < 0.5), flags=_ffi.CF_BASELINE_JOINED | _ffi.CF_RANDOM_FANOUT)
    kw.update(tcp_fallback=bool(swim and rng.random() < 0.4), nacks=bool(swim and rng.random() < 0.4))
    rate = float(rng.choice([0.3, 1.0, 2.5]))
    if seed >= 16:
        kw.update(pkt_records=int(rng.choice([8, 12, 16])))
        rate = float(rng.choice([1.0, 2.5, 5.0]))
    if n >= 2048 and seed % 3 == 0:
        kw.update(vshards=4)   # (r4) ONE handle that holds a cluster of 4 virtual shards: what a 4-shard run is compared with
    g, o = pair(oracle, hiplib, n, **kw)
    ops = sc.schedule(n, 40, rate=rate, seed=seed, max_member_subjects=max(1, min(n // 2, 12 if not dense else 30)))
    sc.apply_schedule(g, ops)
    sc.apply_schedule(o, ops)
    g2 = o2 = None
    for t in range(64):
        g.step(1)
        o.step(1)
        if g.digest() != o.digest():
            sc.assert_same_state(g, o, f"seed {seed} n={n} {kw} tick {t}")
            raise AssertionError(f"digest differs after tick {t} but the arrays agree")
        if t == 30 and seed % 2 == 0:
            gi, oi = g.snapshot(), o.snapshot()
            assert bytes(gi) == bytes(oi)
            g2, o2 = pair(oracle, hiplib, n, **kw)
            g2.restore(oi)
            o2.restore(gi)
            assert g2.digest() == o2.digest() == g.digest()
    sc.assert_same_state(g, o, f"seed {seed} final")
    for node in (0, n // 2, n - 1):   # the byte boundary in this mode: the packets a node SENT (the canonical inbox is sender-indexed)
        for k in range(kw["fanout"]):
            assert g.peek_packet(node, k) == o.peek_packet(node, k)
    if g2 is not None:
        g2.step(33)
        o2.step(33)
        assert g2.digest() == o2.digest() == g.digest()


@pytest.mark.parametrize("n,cap", [(1 << 16, None), (300_000, None), (150_016, None), (2048, 48), (5000, 300), (20_000, -7000), (3000, -1), (70_000, 0)])
def test_random_fanout_graph_build_at_size(oracle, hiplib, n, cap, monkeypatch):
    # the graph build over many level-1 buckets and more than one workgroup of senders (64 Ki nodes: 256 buckets, 16 workgroups;
    # 300 000: a ragged last bucket and a ragged last workgroup), and — SERF_RF_CAP — buckets that do NOT fit rf_rows' LDS
    # tables, which are then ranked straight from global memory; SWIM, loss and (150 016 nodes) paged packets on; digests every
    # few ticks.  At 300 000 nodes x 40 ticks a dozen nodes receive more than sixteen packets in a tick (Poisson tail): the
    # balanced classification's cursor path for the packets beyond a node's sixteenth
    if cap is not None and cap > 0:
        monkeypatch.setenv("SERF_RF_CAP", str(cap))
    if cap == 0:   # 64-bit entries (what a shard of more than 4 Mi nodes uses: a pair id and a target's offset do not fit 32 bits)
        monkeypatch.setenv("SERF_RF_WIDE", "1")
    if cap is not None and cap < 0:   # regions too small for their buckets (8 192 pairs each): the rest goes through the overflow list
        monkeypatch.setenv("SERF_RF_BCAP", str(-cap))
    kw = dict(fanout=4, view_slots=32, event_ring=32, query_ring=16, probe_interval=5, loss=0.01, push_pull_interval=20,
              pkt_records=8 if n == 150_016 else 4, flags=_ffi.CF_BASELINE_JOINED | _ffi.CF_RANDOM_FANOUT)
    g, o = pair(oracle, hiplib, n, **kw)
    ops = sc.schedule(n, 30, rate=1.0, seed=n, max_member_subjects=12)
    sc.apply_schedule(g, ops)
    sc.apply_schedule(o, ops)
    for t in range(8):
        g.step(5)
        o.step(5)
        assert g.digest() == o.digest(), f"n={n}: digest differs after tick {5 * t + 4}"
    if n <= 5000:
        sc.assert_same_state(g, o, f"n={n} final")
    # the in-degree really is random: some node received nothing, some node more than the fan-out
    pk = g.dump(_ffi.ARR_INBOX)
    assert pk.shape[0] == 4 * (kw["pkt_records"] // 4) * n


@pytest.mark.parametrize("seed", list(range(24)))
def test_random_configurations_paged_packets(oracle, hiplib, seed):
    # the same sweep with packets of 8, 12 and 16 records (sim_config.pkt_records: pages of 4 records)
    # at loads that fill the extra pages — the multi-page instantiation of the tick kernel (page walk on the receiving
    # side, the general drain + page-by-page cell stores on the sending side, sender-kept pages and the map word's
    # first-page / page-count encoding), virtual shards and chunks included; a checkpoint in the middle goes through the
    # canonical paged inbox and back
    rng = np.random.default_rng(5000 + seed)
    v = int(rng.choice([1, 1, 2, 4]))
    c = int(rng.choice([0, 0, 2, 4])) if v > 1 or rng.random() < 0.3 else 0
    unit = v * v * max(c, 1)
    n = int(rng.choice([96, 200, 512, 1000, 2048, 4096, 8192]))
    n = max(unit, n // unit * unit) if (v > 1 or c) else n
    dense = n <= 600 and rng.random() < 0.4
    swim = int(rng.choice([0, 2, 3, 5]))
    kw = dict(fanout=int(rng.integers(1, 5)), vshards=v, chunks=c, view_slots=0 if dense else int(rng.choice([16, 48, 64])),
              event_ring=int(rng.choice([8, 12, 16, 64])), query_ring=int(rng.choice([8, 10, 32])), leave_delay=int(rng.integers(3, 9)),
              loss=float(rng.choice([0.0, 0.0, 0.02, 0.1])), probe_interval=swim, reap_interval=int(rng.choice([0, 5, 11])) if swim else 0,
              reconnect_timeout=20, tombstone_timeout=30, intent_timeout=15,
              queue_check_interval=int(rng.choice([0, 7])), min_queue_depth=int(rng.choice([0, 2])),
              push_pull_interval=int(rng.choice([0, 4, 9])), recycle_interval=int(rng.choice([0, 6])) if not dense else 0,
              pkt_records=int(rng.choice([8, 12, 16])))
    kw.update(awareness_probe=bool(swim and rng.random() < 0.5), gossip_to_the_dead=int(rng.choice([0, 0, 2, 8])) if swim else 0,
              join_sync=bool(rng.random() < 0.5))
    kw.update(reconnect_interval=int(rng.choice([0, 0, 3, 7])) if swim else 0)  # Reconnector (drawn after everything else)
    kw.update(tcp_fallback=bool(swim and rng.random() < 0.4), nacks=bool(swim and rng.random() < 0.4))  # memberlist's fallback ping / nack accounting
    try:
        g, o = pair(oracle, hiplib, n, **kw)
    except _ffi.SimError:
        pytest.skip(f"configuration rejected by both sides: n={n} {kw}")
    rate = float(rng.choice([0.8, 2.5, 5.0]))
    ops = sc.schedule(n, 40, rate=rate, seed=seed, max_member_subjects=12 if not dense else min(n // 2, 30))
    sc.apply_schedule(g, ops)
    sc.apply_schedule(o, ops)
    g2 = o2 = None
    for t in range(64):
        g.step(1)
        o.step(1)
        if g.digest() != o.digest():
            sc.assert_same_state(g, o, f"seed {seed} n={n} {kw} tick {t}")
            raise AssertionError(f"digest differs after tick {t} but the arrays agree")
        if t == 30 and seed % 3 == 0:
            gi, oi = g.snapshot(), o.snapshot()
            assert bytes(gi) == bytes(oi)
            g2, o2 = pair(oracle, hiplib, n, **kw)
            g2.restore(oi)
            o2.restore(gi)
    sc.assert_same_state(g, o, f"seed {seed} final")
    if g2 is not None:
        g2.step(33)
        o2.step(33)
        assert g2.digest() == o2.digest() == g.digest()
    pk = o.dump(_ffi.ARR_INBOX)
    assert pk.shape[0] == kw["fanout"] * (kw["pkt_records"] // 4) * n


def test_paged_packets_carry_the_whole_queue(oracle, hiplib):
    # what the pages are for: with pkt_records = 16 = SIM_Q a packet carries every queued record (delegate.rs:317-384
    # fills by bytes: tens of small messages), so a burst of rumours reaches everybody without waiting for a turn in a
    # 4-record packet — and pages beyond the first really are used
    n = 4096
    kw = dict(fanout=4, view_slots=32, event_ring=64, query_ring=16, probe_interval=5)
    res = {}
    for P in (4, 16):
        g, o = pair(oracle, hiplib, n, pkt_records=P, **kw)
        for s in (g, o):
            for i in range(12):   # twelve user events from one node in one tick (Lamport times 1 .. 12)
                s.inject(2, _ffi.OP_USER_EVENT, 100, 500 + i, 40)
        used = 0
        for t in range(40):
            g.step(1)
            o.step(1)
            assert g.digest() == o.digest(), f"P={P}: digest differs after tick {t}"
            if P == 16 and t == 12:
                pk = g.dump(_ffi.ARR_INBOX).reshape(4, 4, n)   # [slot][page][node]
                used = int(((pk["hi_meta"][:, 1:, :, :] >> 4) & 15 != 0).sum())
        seen = [g.convergence(_ffi.K_EVENT, 500 + i, 1 + i)[0] for i in range(12)]
        res[P] = (sum(x >= n * 99 // 100 for x in seen), g.cluster_stats()["overflow"], used)
        sc.assert_same_state(g, o, f"P={P} final")
    assert res[16][0] == 12 and res[16][1] == 0 and res[16][2] > 0, res
    assert res[4][0] <= 12 and res[4][2] == 0


def test_packet_loss_and_overload(oracle, hiplib):
    # 5 % packet loss and an injection rate above the protocol's capacity => queue overflow paths
    g, o = pair(oracle, hiplib, 512, fanout=3, view_slots=128, event_ring=8, query_ring=8, loss=0.05)
    ops = sc.schedule(512, 40, rate=3.0, seed=99, max_member_subjects=100)
    sc.apply_schedule(g, ops)
    sc.apply_schedule(o, ops)
    for t in range(80):
        g.step(1)
        o.step(1)
        assert g.digest() == o.digest(), f"digest differs after tick {t}"
    sc.assert_same_state(g, o, "loss+overload final")
    assert o.dump(_ffi.ARR_ROWS)["overflow"].sum() > 0, "scenario should exercise the overflow path"


def test_swim_crash_refute_leave_events(oracle, hiplib):
    # memberlist layer end to end: crashes detected through probes and suspicion timers, a revived node
    # refuting, graceful leaves, packet loss causing false suspicions; watched observers' event logs
    n = 256
    kw = dict(fanout=3, view_slots=0, event_ring=16, query_ring=8, leave_delay=5, probe_interval=3,
              suspicion_mult=4, suspicion_max_mult=3, indirect_checks=1, loss=0.2,
              reap_interval=10, reconnect_timeout=60, tombstone_timeout=80, push_pull_interval=12)
    g, o = pair(oracle, hiplib, n, **kw)
    for s in (g, o):
        for w in (0, 7, 200):
            s.watch(w)
        s.inject(2, _ffi.OP_CRASH, 50)
        s.inject(3, _ffi.OP_CRASH, 51)
        s.inject(40, _ffi.OP_REVIVE, 51)
        s.inject(5, _ffi.OP_USER_EVENT, 9, 77, 40)
        s.step(1)
        s.leave(60)
        s.inject(90, _ffi.OP_JOIN, 60)
    for t in range(260):
        g.step(1)
        o.step(1)
        if g.digest() != o.digest():
            sc.assert_same_state(g, o, f"swim tick {t}")
            raise AssertionError(f"digest differs after tick {t}")
    sc.assert_same_state(g, o, "swim final")
    eg, eo = g.drain_events(), o.drain_events()
    assert eg == eo and len(eo) > 0
    assert o.dump(_ffi.ARR_ROWS)["inc"].max() >= 1, "scenario should exercise refutation"
    assert o.dump(_ffi.ARR_ROWS)["n_failed"].max() >= 1, "scenario should declare the crashed node failed"


def test_seq_renormalisation(oracle, hiplib):
    # > 1023 queue ids on one node forces the id renumbering path
    g, o = pair(oracle, hiplib, 64, fanout=3, event_ring=2048)
    for t in range(1100):
        for s in (g, o):
            s.inject(t, _ffi.OP_USER_EVENT, 3, t + 1, 32)
    for t in range(0, 1120, 16):
        g.step(16)
        o.step(16)
        assert g.digest() == o.digest(), f"digest differs after tick {t + 16}"
    sc.assert_same_state(g, o, "renorm final")


def test_config2_64k_bit_exact(oracle, hiplib):
    # BASELINE config 2: 64 Ki nodes, fan-out 3, bit-exact vs CPU replay, >= 256 ticks
    n = 65536
    g, o = pair(oracle, hiplib, n, fanout=3, view_slots=128, event_ring=64, query_ring=64, probe_interval=5)
    ops = sc.schedule(n, 200, rate=0.5, seed=2, max_member_subjects=100)
    sc.apply_schedule(g, ops)
    sc.apply_schedule(o, ops)
    for t in range(0, 256, 8):
        g.step(8)
        o.step(8)
        assert g.digest() == o.digest(), f"digest differs after tick {t + 8}"
    for node in (0, 1, 4097, n - 1):
        a, b = g.stats(node), o.stats(node)
        for f, _ in _ffi.Stats._fields_:
            assert getattr(a, f) == getattr(b, f), (node, f)
        sa, la = g.members(node)
        sb, lb = o.members(node)
        assert (sa == sb).all() and (la == lb).all()
    k0 = next(op for op in ops if op[1] == _ffi.OP_USER_EVENT)
    assert g.convergence(_ffi.K_EVENT, k0[3], 1) == o.convergence(_ffi.K_EVENT, k0[3], 1)


def test_long_soak_everything_on(oracle, hiplib):
    # 16 Ki nodes for 1 600 ticks with every subsystem on and short periods, so that the slow machinery
    # runs many times: suspicion timers and refutations, push-pull batches, Reaper timeouts, QueueChecker,
    # query deadlines with acks / responses / relays, packet loss, renumbering of the queue ids
    n = 1 << 14
    kw = dict(fanout=3, view_slots=96, event_ring=48, query_ring=48, probe_interval=3, loss=0.02,
              push_pull_interval=16, reap_interval=25, reconnect_timeout=300, tombstone_timeout=200,
              queue_check_interval=40, max_queue_depth=12, intent_timeout=120)
    g, o = pair(oracle, hiplib, n, **kw)
    for w in (5, 4099, n - 1):
        g.watch(w)
        o.watch(w)
    ops = sc.schedule(n, 1500, rate=0.35, seed=77, max_member_subjects=90)
    sc.apply_schedule(g, ops)
    sc.apply_schedule(o, ops)
    for t in range(0, 1600, 100):
        g.step(100)
        o.step(100)
        assert g.digest() == o.digest(), f"digest differs after tick {t + 100}"
    ev_g, ev_o = g.drain_events(), o.drain_events()
    assert len(ev_g) > 100 and ev_g == ev_o  # the watched observers' event streams, in order


def test_long_soak_everything_on_random_fanout(oracle, hiplib):
    # the soak above on memberlist's literal kRandomNodes (the benchmark's headline model): the graph built two ticks ahead for
    # 1 600 ticks, balanced classification, paged packets, gossip_to_the_dead, the Reconnector, a checkpoint in the middle
    n = 1 << 14
    kw = dict(fanout=4, view_slots=96, event_ring=48, query_ring=48, probe_interval=3, loss=0.02, pkt_records=8,
              push_pull_interval=16, reap_interval=25, reconnect_timeout=300, tombstone_timeout=200, reconnect_interval=40,
              queue_check_interval=40, max_queue_depth=12, intent_timeout=120, gossip_to_the_dead=30,
              flags=_ffi.CF_BASELINE_JOINED | _ffi.CF_RANDOM_FANOUT)
    g, o = pair(oracle, hiplib, n, **kw)
    for w in (5, 4099, n - 1):
        g.watch(w)
        o.watch(w)
    ops = sc.schedule(n, 1500, rate=0.35, seed=78, max_member_subjects=90)
    sc.apply_schedule(g, ops)
    sc.apply_schedule(o, ops)
    for t in range(0, 1600, 100):
        g.step(100)
        o.step(100)
        assert g.digest() == o.digest(), f"digest differs after tick {t + 100}"
        if t == 700:   # the oracle's image into the HIP library: the run goes on from it (the graphs in flight are drawn again)
            ev_g, ev_o = g.drain_events(), o.drain_events()
            assert len(ev_g) > 50 and ev_g == ev_o
            g.close()
            g = _ffi.Sim(hiplib, _ffi.make_config(n, **kw))   # (an image goes into a fresh handle)
            g.restore(o.snapshot())
            assert g.digest() == o.digest()


def test_config3_1m_bit_exact_digests(oracle, hiplib):
    # BASELINE config 3 size (1 Mi nodes, fan-out 4, SWIM layer on): digests of every array against the
    # CPU oracle over the first ticks of a busy schedule (what the oracle finishes in seconds)
    n = 1 << 20
    g, o = pair(oracle, hiplib, n, fanout=4, view_slots=32, event_ring=32, query_ring=32, probe_interval=5, reap_interval=8)
    ops = sc.schedule(n, 40, rate=2.0, seed=5, max_member_subjects=24)
    sc.apply_schedule(g, ops)
    sc.apply_schedule(o, ops)
    for t in range(0, 48, 8):
        g.step(8)
        o.step(8)
        assert g.digest() == o.digest(), f"digest differs after tick {t + 8}"


def test_config3_1m_properties(hiplib):
    # BASELINE config 3 size (1 Mi nodes, fan-out 4): size-independent properties
    n = 1 << 20
    g = _ffi.Sim(hiplib, _ffi.make_config(n, fanout=4, view_slots=64, event_ring=64, query_ring=64))
    g.user_event(12345, 777, 64)
    g.leave(4242)
    seen_prev = 0
    for t in range(24):
        g.step(1)
        seen, up = g.convergence(_ffi.K_EVENT, 777, 1)
        assert seen >= seen_prev, "a rumor never un-spreads"
        seen_prev = seen
    assert up == n
    assert seen == n, "lossless epidemic reaches every node"
    seen, up = g.convergence(_ffi.K_LEAVE, 4242, 2)
    assert seen == up
    st, lt = g.members(0)
    assert st[4242] == _ffi.STATUS_LEAVING and lt[4242] == 2
    assert (np.delete(st, 4242) == _ffi.STATUS_ALIVE).all()
    rows = g.dump(_ffi.ARR_ROWS)
    assert rows["clock"].min() == 3 and rows["event_clock"].min() == 2  # everyone witnessed both
    # idempotence: the queues drain and the state stops changing
    g.step(40)
    d1 = g.digest()
    g.step(5)
    d2 = g.digest()
    assert d1[:2] == d2[:2] and d1[3:] == d2[3:]
    assert g.dump(_ffi.ARR_QUEUE)["meta"].min() == 0xFFFFFFFF


def _suspicions_on_one_gpu(shards):
    """serf_amd/shard.py ShardedSim._suspicions_out / _in for hand-driven shards, called after every sim_step_end: every
    shard exports the head of its list of slot-less failed probes of the tick just ended (sim_suspect_export, device
    memory), the "all-gather" is a copy to the host, and the gathered heads of tick t - 1 are imported on every shard
    (sim_suspect_import) before tick t + 1 begins."""
    import torch

    V, t = len(shards), shards[0].tick - 1
    send = torch.zeros(V, _ffi.SREQ_HEAD_WORDS, dtype=torch.int32, device="cuda")
    for v, s in enumerate(shards):
        s.suspect_export(send[v].data_ptr())
        s.sync()
    q = shards[0].__dict__.setdefault("_sq_pending", [])
    q.append((t, send.cpu().contiguous()))
    while q and q[0][0] + 2 <= shards[0].tick:
        of_tick, heads = q.pop(0)
        for s in shards:
            s.suspect_import(of_tick, heads.data_ptr(), V)


def _push_pull_on_one_gpu(shards):
    """The cross-shard push-pull batch of serf_amd/shard.py with device-to-device copies standing in for the
    all-to-all-v: two rounds of sim_pp_export -> move every (source, destination) slice -> sim_pp_merge."""
    import torch

    V = len(shards)
    plans = [s.pp_plan(V) for s in shards]           # (send1[V], recv1[V], record_bytes)
    rb = plans[0][2]
    for rnd in (1, 2):
        outc = [p[0] if rnd == 1 else p[1] for p in plans]
        inc = [p[1] if rnd == 1 else p[0] for p in plans]
        send = [torch.zeros(max(1, sum(c) * rb), dtype=torch.uint8, device="cuda") for c in outc]
        recv = [torch.zeros(max(1, sum(c) * rb), dtype=torch.uint8, device="cuda") for c in inc]
        for g, s in enumerate(shards):
            s.pp_export(rnd, send[g].data_ptr())
        for s in shards:
            s.sync()
        for src in range(V):
            so = 0
            for dst in range(V):
                n = outc[src][dst] * rb
                assert outc[src][dst] == inc[dst][src]
                ro = sum(inc[dst][:src]) * rb
                if n:
                    recv[dst][ro:ro + n].copy_(send[src][so:so + n])
                so += n
        torch.cuda.synchronize()
        for g, s in enumerate(shards):
            s.pp_merge(rnd, recv[g].data_ptr())
        for s in shards:
            s.sync()


@pytest.mark.parametrize("swim,chunks,pkt,rc", [(0, 1, 0, 0), (4, 1, 0, 0), (4, 2, 0, 0), (0, 4, 0, 0), (4, 2, 8, 0), (0, 1, 16, 0), (2, 2, 0, 3)])
def test_sharded_kernel_four_shards_on_one_gpu(oracle, hiplib, swim, chunks, pkt, rc):
    # BASELINE configs[3] shape (G shards by node-id range, the round's all-to-all), scaled down and run as 4 handles
    # on ONE GPU: the exchange of serf_amd/shard.py is done with device-to-device copies (for every sender chunk c:
    # slab g of shard s's send region c -> slab s of shard g's receive region c), which is what the per-chunk
    # all_to_all_single does over RCCL.  chunks > 1 drives the tick as sim_step_begin / sim_step_chunk / sim_step_end
    # with the double-buffered receive side.  Every shard is compared with the oracle's matching slice.
    import torch

    # rc: the Reconnector — its attempts cross the shards on the request list and run as push-pull pairs of their tick
    # (sim_pp_due / _plan / _export / _merge on a tick without a batch), most of them between two shards
    n, V, ticks = 2048, 4, 50 if not rc else 90
    m = n // V
    kw = dict(fanout=4, view_slots=96, event_ring=16, query_ring=8, leave_delay=6, probe_interval=swim, loss=0.02,
              push_pull_interval=3 if swim else 0, chunks=chunks if chunks > 1 else 0, pkt_records=pkt,   # pkt: paged packets in the exchange buffers
              prune_delay=bool(swim),   # handle_prune's wait: its notes go over the shards' request-list hand-over
              reconnect_interval=rc, **(dict(suspicion_mult=3, suspicion_max_mult=2, gossip_to_the_dead=1) if rc else {}))
    ref = _ffi.Sim(oracle, _ffi.make_config(n, vshards=V, **kw))
    shards, send, recv = [], [], []
    for g in range(V):
        s = _ffi.Sim(hiplib, _ffi.make_config(n, vshards=V, shard_rank=g, shard_count=V, **kw))
        nb = s.exchange_bytes()
        send.append(torch.zeros(nb, dtype=torch.uint8, device="cuda"))
        recv.append([torch.zeros(nb, dtype=torch.uint8, device="cuda") for _ in range(2)])
        s.bind_exchange2(send[-1].data_ptr(), recv[-1][0].data_ptr(), recv[-1][1].data_ptr())
        assert s.exchange_chunks() == (chunks, nb // chunks)
        shards.append(s)
    ops = sc.schedule(n, ticks // 2, rate=0.8 if not pkt else 3.0, seed=5, max_member_subjects=60)
    for s in shards + [ref]:
        sc.apply_schedule(s, ops)
        for x in ((7, 300, 777, 1200, 1500, 2000) if rc else ()):   # six nodes go down and are declared failed; four of them resume
            s.inject(1, _ffi.OP_CRASH, x)                            # when nobody gossips to them any more (gossip_to_the_dead):
            if x > 500:                                              # only a peer's Reconnector can bring those back
                s.inject(50, _ffi.OP_REVIVE, x)
    region = send[0].numel() // chunks
    slab = region // V
    rc_ticks = 0
    pp_step = max(1, kw["push_pull_interval"] * (int(np.ceil(np.log2(n) - 5)) + 1) // 8)   # oracle pp_params
    for t in range(ticks):
        for s in shards:
            s.step_begin()
        if shards[0].pp_due():
            rc_ticks += 0 if (t > 0 and t % pp_step == 0) else 1   # not a batch tick: the exchange runs for reconnect attempts alone
            _push_pull_on_one_gpu(shards)
        for c in range(chunks):
            for s in shards:
                s.step_chunk(c)
            for s in shards:
                s.sync()
            for g in range(V):          # the all-to-all of chunk c, into the receive buffer of tick t
                for src in range(V):
                    recv[g][t & 1][c * region + src * slab:c * region + (src + 1) * slab].copy_(
                        send[src][c * region + g * slab:c * region + (g + 1) * slab])
        for s in shards:
            s.step_end()
        _suspicions_on_one_gpu(shards)
        torch.cuda.synchronize()
        ref.step(1)
        if t % 7 == 0 or t == ticks - 1:
            for g, s in enumerate(shards):
                lo = g * m
                for which in (_ffi.ARR_ROWS, _ffi.ARR_QUEUE):
                    a, b = s.dump(which), ref.dump(which)
                    per = len(b) // n
                    i = sc.first_diff(a, b[lo * per:(lo + m) * per])
                    assert i is None, f"shard {g} array {which} element {i} differs at tick {t}"
                for which, rows in ((_ffi.ARR_VIEW, 96), (_ffi.ARR_ERING, 16), (_ffi.ARR_QRING, 8)):
                    a = s.dump(which).reshape(rows, m)
                    b = np.ascontiguousarray(ref.dump(which).reshape(rows, n)[:, lo:lo + m])
                    assert a.tobytes() == b.tobytes(), f"shard {g} array {which} differs at tick {t}"
    assert not rc or rc_ticks >= 2, "no reconnect attempt ran as a push-pull of its own"
    if rc:
        st = shards[0].members(5)[0]
        assert [int(st[x]) for x in (777, 1200, 1500, 2000)] == [_ffi.STATUS_ALIVE] * 4 and int(st[7]) == _ffi.STATUS_FAILED
    for qop in [op for op in ops if op[1] == _ffi.OP_QUERY and op[4] & _ffi.F_ACK][:3]:
        parts = [s.query_status(qop[3]) for s in shards]
        assert (sum(p[0] for p in parts), sum(p[1] for p in parts), parts[0][2]) == ref.query_status(qop[3])
        # every shard lists the responders among ITS nodes: together, the single-process list
        assert sorted(x for s in shards for x in s.query_responders(qop[3], 0)) == ref.query_responders(qop[3], 0)
    tot = [sum(x) for x in zip(*(s.convergence(_ffi.K_LEAVE, next(o for o in ops if o[1] == _ffi.OP_LEAVE)[2], 1) for s in shards))]
    assert tuple(tot) == ref.convergence(_ffi.K_LEAVE, next(o for o in ops if o[1] == _ffi.OP_LEAVE)[2], 1)


def _packed_exchange_on_one_gpu(send, recv):
    """SIM_XCHG_PACKED / SIM_XCHG_ALL_TO_ALL with device-to-device copies: slab g of shard src's send buffer -> slab src of shard g's
    receive buffer (equal split: what all_to_all_single, or the library's grouped ncclSend / ncclRecv, does)"""
    V = len(send)
    slab = send[0].numel() // V
    for g in range(V):
        for src in range(V):
            recv[g][src * slab:(src + 1) * slab].copy_(send[src][g * slab:(g + 1) * slab])


def _packed_exchange_chunk_on_one_gpu(send, recv, c, C):
    """chunk c of a chunk-wise exchange: region c of every send buffer, equal split, into region c of the receive buffers"""
    V = len(send)
    reg = send[0].numel() // C
    slab = reg // V
    for g in range(V):
        for src in range(V):
            recv[g][c * reg + src * slab:c * reg + (src + 1) * slab].copy_(send[src][c * reg + g * slab:c * reg + (g + 1) * slab])


@pytest.mark.parametrize("V,swim,pkt,rc,n,C", [(4, 0, 0, 0, 2048, 1), (4, 4, 0, 0, 2048, 1), (4, 4, 8, 0, 2048, 1), (4, 2, 0, 3, 2048, 1), (4, 4, 0, 0, 1 << 16, 1),
                                               (8, 4, 16, 0, 4096, 1), (2, 4, 0, 0, 1 << 17, 1), (8, 0, 0, 0, 64, 1),
                                               # sender chunks (r5): chunk c's slabs are packed behind chunk c's launch; a row is V * C runs
                                               (4, 4, 0, 0, 2048, 2), (4, 4, 8, 0, 1 << 16, 4), (2, 2, 0, 3, 2048, 2)])
def test_random_fanout_four_shards_on_one_gpu(oracle, hiplib, V, swim, pkt, rc, n, C):
    # memberlist's kRandomNodes on shards (r5: the scalable form): V handles on ONE GPU, each sorting the
    # (target, sender, slot) triples of its OWN senders and packing the packets bound for shard h into slab h (sim_exchange_layout:
    # SIM_XCHG_PACKED); the round's exchange is an equal-split all-to-all of the slabs, done here with device-to-device copies.
    # Every shard against the oracle's matching slice of ONE handle that holds every node; the cross-shard push-pull / suspicion
    # hand-over as in the bijection's test.  (2 x 64 Ki: 64-bit sort entries; 8 x 8 nodes: slabs that hold everything.)
    import torch

    ticks = 50 if not rc else 90
    m = n // V
    kw = dict(fanout=4, view_slots=96 if n > 64 else 0, event_ring=16, query_ring=8, leave_delay=6, probe_interval=swim, loss=0.02,
              push_pull_interval=3 if swim else 0, pkt_records=pkt, reconnect_interval=rc, flags=_ffi.CF_BASELINE_JOINED | _ffi.CF_RANDOM_FANOUT,
              **(dict(suspicion_mult=3, suspicion_max_mult=2, gossip_to_the_dead=1) if rc else {}))
    A = 96 if n > 64 else n
    ref = _ffi.Sim(oracle, _ffi.make_config(n, vshards=V, **kw))
    kw = dict(kw, chunks=C if C > 1 else 0)   # (sender chunks are the shards' exchange schedule: the handle that holds every node has none)
    shards, send, recv = [], [], []
    for g in range(V):
        s = _ffi.Sim(hiplib, _ffi.make_config(n, vshards=V, shard_rank=g, shard_count=V, **kw))
        kind, planes, pb, rb = s.exchange_layout()
        assert kind == _ffi.XCHG_PACKED and planes == 1 and pb == rb == s.exchange_bytes() and pb % (64 * V * C) == 0
        assert s.exchange_chunks() == (C, pb // C)
        # what leaves a GPU per tick: f packets of 64 bytes per node, (V - 1) / V of them, 2 % of room, a count byte per target
        if n >= 1 << 16 and C == 1:   # (12 sigma of sqrt(m) = 16 Ki packets per slab is 9 %; at 1 Mi nodes per shard it is 2 %)
            assert pb <= 1.13 * 4 * 64 * m, (pb, 4 * 64 * m)
        send.append(torch.zeros(pb, dtype=torch.uint8, device="cuda"))
        # (packets sent during tick t land in recv[t & 1]: with sender chunks the late chunks of tick t + 1 still read tick t's)
        recv.append([torch.zeros(rb, dtype=torch.uint8, device="cuda") for _ in range(2 if C > 1 else 1)])
        with pytest.raises(_ffi.SimError):   # the sized bind refuses buffers that are too small
            s.bind_exchange3(send[-1].data_ptr(), pb - 64, recv[-1][0].data_ptr(), recv[-1][-1].data_ptr(), rb)
        s.bind_exchange3(send[-1].data_ptr(), pb, recv[-1][0].data_ptr(), recv[-1][-1].data_ptr(), rb)
        shards.append(s)
    ops = sc.schedule(n, ticks // 2, rate=0.8 if not pkt else 3.0, seed=5, max_member_subjects=min(60, n // 2))
    for s in shards + [ref]:
        sc.apply_schedule(s, ops)
        for x in ((7, 300, 777, 1200, 1500, 2000) if rc else ()):
            s.inject(1, _ffi.OP_CRASH, x)
            if x > 500:
                s.inject(50, _ffi.OP_REVIVE, x)
    fp = 4 * max(1, pkt // 4)

    def tick_all(hs):
        for s in hs:
            s.step_begin()
        if hs[0].pp_due():
            _push_pull_on_one_gpu(hs)
        into = [r[hs[0].tick & 1] if C > 1 else r[0] for r in recv]
        for c in range(C):
            for s in hs:
                s.step_chunk(c)
                s.sync()
            _packed_exchange_chunk_on_one_gpu(send, into, c, C)
        for s in hs:
            s.step_end()
            s.sync()
        _suspicions_on_one_gpu(hs)
        torch.cuda.synchronize()
        ref.step(1)

    for t in range(ticks):
        tick_all(shards)
        if t % 7 == 0 or t == ticks - 1:
            for g, s in enumerate(shards):
                lo = g * m
                for which in (_ffi.ARR_ROWS, _ffi.ARR_QUEUE):
                    a, b = s.dump(which), ref.dump(which)
                    per = len(b) // n
                    i = sc.first_diff(a, b[lo * per:(lo + m) * per])
                    assert i is None, f"shard {g} array {which} element {i} differs at tick {t}"
                for which, rows in ((_ffi.ARR_VIEW, A), (_ffi.ARR_ERING, 16), (_ffi.ARR_QRING, 8), (_ffi.ARR_INBOX, fp)):
                    a = s.dump(which).reshape(rows, m)
                    b = np.ascontiguousarray(ref.dump(which).reshape(rows, n)[:, lo:lo + m])
                    assert a.tobytes() == b.tobytes(), f"shard {g} array {which} differs at tick {t}"
    tot = [sum(x) for x in zip(*(s.convergence(_ffi.K_LEAVE, next(o for o in ops if o[1] == _ffi.OP_LEAVE)[2], 1) for s in shards))]
    assert tuple(tot) == ref.convergence(_ffi.K_LEAVE, next(o for o in ops if o[1] == _ffi.OP_LEAVE)[2], 1)
    # a checkpoint of EVERY shard into fresh handles: their own cells come back, the library packs them again, the exchange is
    # run once more, and the restored run goes on in step with the oracle
    pend = shards[0].__dict__.get("_sq_pending", [])
    while pend:  # (the lists of slot-less suspicions still travelling live in the host's queue: imported first, like ShardedSim.snapshot)
        of_tick, heads = pend.pop(0)
        for s in shards:
            s.suspect_import(of_tick, heads.data_ptr(), V)
    imgs = [s.snapshot() for s in shards]
    fresh = []
    for g in range(V):
        f2 = _ffi.Sim(hiplib, _ffi.make_config(n, vshards=V, shard_rank=g, shard_count=V, **kw))
        f2.bind_exchange3(send[g].data_ptr(), send[g].numel(), recv[g][0].data_ptr(), recv[g][-1].data_ptr(), recv[g][0].numel())
        f2.restore(imgs[g])
        assert f2.digest() == shards[g].digest()
        fresh.append(f2)
    torch.cuda.synchronize()
    into = [r[(fresh[0].tick - 1) & 1] if C > 1 else r[0] for r in recv]   # the packets in flight were sent during tick - 1
    for c in range(C):
        _packed_exchange_chunk_on_one_gpu(send, into, c, C)
    for t in range(6):
        tick_all(fresh)
        for g, s in enumerate(fresh):
            a, b = s.dump(_ffi.ARR_ROWS), ref.dump(_ffi.ARR_ROWS)
            per = len(b) // n
            assert sc.first_diff(a, b[g * m * per:(g + 1) * m * per]) is None, f"restored shard {g} differs {t + 1} ticks on"


def test_sharded_kernel_b64_four_shards_of_64k_on_one_gpu(oracle, hiplib):
    """the SHARDED instantiation of the tick kernel with 64-node blocks, at a shard size near the
    bench's — 4 handles x 64 Ki nodes on one GPU, 2 sender chunks (2 x 512 blocks per shard and tick: tp.B == 64, the
    block permutation on the scalar unit, chunk launches, double-buffered receive side), failure detector on, cross-shard
    push-pull batches, view-slot recycling with the all-shard verdict, packet loss — against the single-process oracle's
    slices: rows and queues every 25 ticks, views and rings at three checkpoints, for 160 ticks."""
    import torch
    from tests.test_recycle import churn_ops

    V, m, chunks, ticks = 4, 1 << 16, 2, 200
    n = V * m
    kw = dict(fanout=4, view_slots=48, event_ring=32, query_ring=16, leave_delay=6, probe_interval=5, loss=0.01,
              suspicion_mult=3, suspicion_max_mult=2, push_pull_interval=2, recycle_interval=40, chunks=chunks)
    ref = _ffi.Sim(oracle, _ffi.make_config(n, vshards=V, **kw))
    shards, send, recv = [], [], []
    for g in range(V):
        s = _ffi.Sim(hiplib, _ffi.make_config(n, vshards=V, shard_rank=g, shard_count=V, **kw))
        nb = s.exchange_bytes()
        send.append(torch.zeros(nb, dtype=torch.uint8, device="cuda"))
        recv.append([torch.zeros(nb, dtype=torch.uint8, device="cuda") for _ in range(2)])
        s.bind_exchange2(send[-1].data_ptr(), recv[-1][0].data_ptr(), recv[-1][1].data_ptr())
        assert s.exchange_chunks() == (chunks, nb // chunks)
        shards.append(s)
    # user events, queries and short crash + revive pairs, plus five nodes that stay down for 100 ticks: long enough to be
    # suspected, confirmed and declared failed (timeout 81 .. 162 ticks here), to refute when they come back, and for their
    # view slots to be recycled afterwards
    ops = sc.schedule(n, 160, rate=0.2, seed=11, mix=(0.6, 0.25, 0.0, 0.0, 0.15), max_member_subjects=12) + \
        churn_ops(n, 5, every=9, down=100, seed=8, start=3)
    ops.sort(key=lambda o: o[0])
    for s in shards + [ref]:
        sc.apply_schedule(s, ops)
    region = send[0].numel() // chunks
    slab = region // V
    n_pp = n_rec = max_failed = 0
    for t in range(ticks):
        if shards[0].recycle_due():
            n_rec += 1
            scans = np.stack([s.recycle_scan() for s in shards])
            keep = []
            for i in range(scans.shape[1]):
                flags = scans[:, i, 2]
                if (flags & 1).any() or not (flags & 2).any():
                    continue
                refs = scans[(flags & 2) != 0, i, 4:8]
                if (refs == refs[0]).all():
                    keep.append(scans[np.nonzero(flags & 2)[0][0], i])
            for s in shards:
                s.recycle_apply(np.array(keep, dtype=np.uint32).reshape(-1, 12))
        for s in shards:
            s.step_begin()
        if shards[0].pp_due():
            n_pp += 1
            _push_pull_on_one_gpu(shards)
        for c in range(chunks):
            for s in shards:
                s.step_chunk(c)
            for s in shards:
                s.sync()
            for g in range(V):
                for src in range(V):
                    recv[g][t & 1][c * region + src * slab:c * region + (src + 1) * slab].copy_(
                        send[src][c * region + g * slab:c * region + (g + 1) * slab])
        for s in shards:
            s.step_end()
        _suspicions_on_one_gpu(shards)
        torch.cuda.synchronize()
        ref.step(1)
        if t % 25 == 0 or t == ticks - 1:
            deep = t in (50, 125, ticks - 1)
            max_failed = max(max_failed, ref.cluster_stats()["failed"])
            rrows, rq = ref.dump(_ffi.ARR_ROWS), ref.dump(_ffi.ARR_QUEUE)
            big = {w: ref.dump(w) for w in (_ffi.ARR_VIEW, _ffi.ARR_ERING, _ffi.ARR_QRING)} if deep else {}
            for g, s in enumerate(shards):
                lo = g * m
                i = sc.first_diff(s.dump(_ffi.ARR_ROWS), rrows[lo:lo + m])
                assert i is None, f"shard {g} row {i} differs at tick {t}"
                i = sc.first_diff(s.dump(_ffi.ARR_QUEUE), rq[lo * _ffi.Q:(lo + m) * _ffi.Q])
                assert i is None, f"shard {g} queue entry {i} differs at tick {t}"
                for which, rows in ((_ffi.ARR_VIEW, 48), (_ffi.ARR_ERING, 32), (_ffi.ARR_QRING, 16)):
                    if deep:
                        a = s.dump(which).reshape(rows, m)
                        b = np.ascontiguousarray(big[which].reshape(rows, n)[:, lo:lo + m])
                        assert a.tobytes() == b.tobytes(), f"shard {g} array {which} differs at tick {t}"
                assert (s.dump(_ffi.ARR_SLOTMAP) == ref.dump(_ffi.ARR_SLOTMAP)).all()
    cr = ref.cluster_stats()
    assert n_pp >= 3 and n_rec >= 3 and cr["slots_recycled"] >= 3, (n_pp, n_rec, cr)
    assert max_failed > n, "suspicion timers fired: the nodes that stayed down were declared failed inside the window"
    assert sum(s.cluster_stats()["overflow"] for s in shards) == cr["overflow"]
    assert all(s.cluster_stats()["slots_recycled"] == cr["slots_recycled"] for s in shards)


def test_bench_configuration_64k_digests(oracle, hiplib):
    # The benchmark's OWN configuration tuple and schedule (bench.workload: fan-out 4, view_slots 1024, rings 512,
    # probe interval 5, push-pull 150, reaper 75, queue checker 150, evenly spaced operations at the bench rate),
    # at 64 Ki nodes so that the oracle keeps up: 440 ticks — the pre-roll into the stationary load and beyond, a
    # push-pull batch (every 225 ticks at this size), Reaper and QueueChecker rounds, suspicion timers firing
    # (min 96 ticks) — with a digest of every array every 40 ticks, and no model bound hit on either side.
    import bench

    n = 1 << 16
    args = bench.parse_args(["--nodes-per-gpu", str(n)])
    kw, ops = bench.workload(args, n)
    g, o = pair(oracle, hiplib, n, **kw)
    sc.apply_schedule(g, ops)
    sc.apply_schedule(o, ops)
    for t in range(0, 440, 40):
        g.step(40)
        o.step(40)
        assert g.digest() == o.digest(), f"digest differs after tick {t + 40}"
    cg, co = g.cluster_stats(), o.cluster_stats()
    assert cg == co
    assert cg["overflow"] == 0, "the benchmark workload must stay inside the model bounds"
    assert cg["failed"] > 0 and cg["left"] > 0, "crashes and leaves were declared (timers fired) within the window"


def test_view_slot_recycling_parity(oracle, hiplib):
    # churn over 4x more subjects than view slots, failure detector on: slots are taken when the operations execute
    # and given back by the recycling pass (SIMSPEC §2.6) — HIP and oracle agree on every array, on the slot map and
    # on the bookkeeping, and nothing is dropped
    from tests.test_recycle import KW, churn_ops

    n = 2048
    kw = dict(KW, view_slots=24, fanout=4)
    g, o = pair(oracle, hiplib, n, **kw)
    ops = churn_ops(n, 100, every=5, down=4)
    sc.apply_schedule(g, ops)
    sc.apply_schedule(o, ops)
    for t in range(0, 700, 10):
        g.step(10)
        o.step(10)
        if g.digest() != o.digest():
            sc.assert_same_state(g, o, f"tick {t + 10}")
            raise AssertionError(f"digest differs after tick {t + 10} but the arrays agree")
    sc.assert_same_state(g, o, "final")
    cg, co = g.cluster_stats(), o.cluster_stats()
    assert cg == co
    assert cg["ops_dropped"] == 0 and cg["slots_recycled"] >= 76 and cg["overflow"] == 0 and cg["up"] == n


def test_timer_on_a_slot_recycled_while_its_holder_was_down(oracle, hiplib):
    # tests/test_recycle.py, same name: a node comes back holding a suspicion timer on a view slot that lost its subject
    # while the node was down — the kernel once used subject_of[slot] = NOSLOT as a subject (a memory fault at scale)
    from tests.test_recycle import KW, stale_timer_ops

    n = 512
    g, o = pair(oracle, hiplib, n, **dict(KW, view_slots=8))
    for s in (g, o):
        sc.apply_schedule(s, stale_timer_ops())
    for t in range(130):
        g.step(1)
        o.step(1)
        assert g.digest() == o.digest(), f"digest differs after tick {t}"
    sc.assert_same_state(g, o, "final")
    assert g.cluster_stats() == o.cluster_stats()


def test_view_slot_recycling_four_shards_on_one_gpu(oracle, hiplib):
    # the sharded form of the pass: every shard scans its own nodes (sim_recycle_scan), the host keeps the candidates
    # all shards agree on, every shard applies the same list (sim_recycle_apply) — against the single-process oracle
    import torch
    from tests.test_recycle import KW, churn_ops

    n, V = 2048, 4
    m = n // V
    kw = dict(KW, view_slots=24, fanout=4)
    ref = _ffi.Sim(oracle, _ffi.make_config(n, vshards=V, **kw))
    shards, send, recv = [], [], []
    for r in range(V):
        s = _ffi.Sim(hiplib, _ffi.make_config(n, vshards=V, shard_rank=r, shard_count=V, **kw))
        nb = s.exchange_bytes()
        send.append(torch.zeros(nb, dtype=torch.uint8, device="cuda"))
        recv.append(torch.zeros(nb, dtype=torch.uint8, device="cuda"))
        s.bind_exchange(send[-1].data_ptr(), recv[-1].data_ptr())
        shards.append(s)
    ops = churn_ops(n, 90, every=5, down=4)
    for s in shards + [ref]:
        sc.apply_schedule(s, ops)
    slab = send[0].numel() // V
    for t in range(600):
        if shards[0].recycle_due():
            scans = np.stack([s.recycle_scan() for s in shards])   # [V, n_cand, 12]
            keep = []
            for i in range(scans.shape[1]):
                flags = scans[:, i, 2]
                if (flags & 1).any() or not (flags & 2).any():
                    continue
                refs = scans[(flags & 2) != 0, i, 4:8]
                if (refs == refs[0]).all():
                    keep.append(scans[np.nonzero(flags & 2)[0][0], i])
            for s in shards:
                s.recycle_apply(np.array(keep, dtype=np.uint32).reshape(-1, 12))
        for s in shards:
            s.step_begin()
        if shards[0].pp_due():
            _push_pull_on_one_gpu(shards)
        for s in shards:
            s.step_chunk(0)
            s.step_end()
        _suspicions_on_one_gpu(shards)
        for s in shards:
            s.sync()
        for r in range(V):
            for src in range(V):
                recv[r][src * slab:(src + 1) * slab].copy_(send[src][r * slab:(r + 1) * slab])
        torch.cuda.synchronize()
        ref.step(1)
        if t % 25 == 0 or t == 599:
            for r, s in enumerate(shards):
                lo = r * m
                for which in (_ffi.ARR_ROWS, _ffi.ARR_QUEUE):
                    a, b = s.dump(which), ref.dump(which)
                    per = len(b) // n
                    i = sc.first_diff(a, b[lo * per:(lo + m) * per])
                    assert i is None, f"shard {r} array {which} element {i} differs at tick {t}"
                a = s.dump(_ffi.ARR_VIEW).reshape(24, m)
                b = np.ascontiguousarray(ref.dump(_ffi.ARR_VIEW).reshape(24, n)[:, lo:lo + m])
                assert a.tobytes() == b.tobytes(), f"shard {r} view differs at tick {t}"
                assert (s.dump(_ffi.ARR_SLOTMAP) == ref.dump(_ffi.ARR_SLOTMAP)).all()
    cr = ref.cluster_stats()
    assert cr["slots_recycled"] >= 66 and cr["ops_dropped"] == 0
    assert all(s.cluster_stats()["slots_recycled"] == cr["slots_recycled"] for s in shards)


def test_packet_byte_budget_mixed_sizes(oracle, hiplib):
    # user events of 16 .. 528 framed bytes at a rate that keeps several large ones queued: packets fill up by BYTES
    # (1 400) before they fill up by records, the general selection walk and the full re-sort run — every array, every tick
    n = 1024
    g, o = pair(oracle, hiplib, n, fanout=4, view_slots=32, event_ring=64, query_ring=16, probe_interval=3, loss=0.01)
    rng = np.random.default_rng(77)
    key = 1
    for t in range(60):
        for _ in range(int(rng.integers(1, 4))):
            node = int(rng.integers(0, n))
            size = int(rng.choice([16, 40, 300, 480, 528]))
            for s in (g, o):
                s.inject(t, _ffi.OP_USER_EVENT, node, key, size)
            key += 1
        if t % 9 == 0:
            node = int(rng.integers(0, n))
            for s in (g, o):
                s.inject(t, _ffi.OP_QUERY, node, key, _ffi.F_ACK)
            key += 1
    hit = False
    for t in range(110):
        g.step(1)
        o.step(1)
        if g.digest() != o.digest():
            sc.assert_same_state(g, o, f"tick {t}")
            raise AssertionError(f"digest differs after tick {t} but the arrays agree")
        if t % 10 == 0:
            sc.assert_same_state(g, o, f"tick {t}")
            q = o.dump(_ffi.ARR_QUEUE).reshape(n, _ffi.Q)
            lens = np.where(q["meta"] != 0xFFFFFFFF, 63 - ((q["meta"] >> 18) & 63), 0)
            hit |= bool((np.sort(lens, axis=1)[:, -4:].sum(axis=1) > 87).any())
    assert hit, "scenario should put more than 1 400 bytes of events at the head of some queue"
    sc.assert_same_state(g, o, "final")


def test_bench_configuration_1m_digests(oracle, hiplib):
    # the benchmark's configuration and schedule at its FULL size (1 Mi nodes, view_slots 1024, rings 512: 66 GB on
    # the device, ~64 GiB of mostly untouched address space on the host) for the first 120 ticks — through the first
    # recycling pass, Reaper round and a dozen operations — digests of every array every 40 ticks
    import bench

    n = 1 << 20
    args = bench.parse_args(["--nodes-per-gpu", str(n)])
    kw, ops = bench.workload(args, n)
    g, o = pair(oracle, hiplib, n, **kw)
    sc.apply_schedule(g, ops)
    sc.apply_schedule(o, ops)
    for t in range(0, 120, 40):
        g.step(40)
        o.step(40)
        assert g.digest() == o.digest(), f"digest differs after tick {t + 40}"
    cg, co = g.cluster_stats(), o.cluster_stats()
    assert cg == co and cg["overflow"] == 0 and cg["ops_dropped"] == 0


@pytest.mark.parametrize("n,swim", [(1024, 0), (1000, 3)])
def test_query_filters_and_tags_parity(oracle, hiplib, n, swim):
    """QueryParam.filters (should_process_query, query.rs:439-521; base.rs:1062-1073) and Serf::set_tags (api.rs:219):
    filtered queries with id lists and tag-class masks, tag changes gossiping as alive messages with the update
    flag; digests (they cover the filter table and the tag classes), watched nodes' event logs, every query's
    ack / response count, and a checkpoint taken mid-run restored into the OTHER implementation."""
    kw = dict(fanout=3, view_slots=96, event_ring=32, query_ring=32, leave_delay=6, probe_interval=swim,
              suspicion_mult=3, suspicion_max_mult=2, push_pull_interval=8 if swim else 0, loss=0.03)
    g, o = pair(oracle, hiplib, n, **kw)
    ops = sc.schedule(n, 80, rate=1.5, seed=31, mix=(0.3, 0.5, 0.1, 0.05, 0.05), max_member_subjects=30)
    ops, classes = sc.with_filters(ops, n, tag_changes=20 if swim else 0, n_ticks=80)
    assert sum(1 for op in ops if op[1] == _ffi.OP_QUERY_FILTER_ID) > 50
    for s in (g, o):
        s.init_tags(classes)
        for w in (0, 3, n // 2, n - 1):
            s.watch(w)
        sc.apply_schedule(s, ops)
    for t in range(140):
        g.step(1)
        o.step(1)
        assert g.digest() == o.digest(), f"digest differs after tick {t}"
        if t == 60:   # canonical image: HIP -> oracle and oracle -> HIP
            gi, oi = g.snapshot(), o.snapshot()
            assert bytes(gi) == bytes(oi)
            g2, o2 = pair(oracle, hiplib, n, **kw)
            g2.restore(oi)
            o2.restore(gi)
    sc.assert_same_state(g, o, "filters final")
    assert g.drain_events() == o.drain_events()
    filtered = {op[3] for op in ops if op[1] in (_ffi.OP_QUERY_FILTER_ID, _ffi.OP_QUERY_FILTER_TAGS)}
    n_checked = n_filtered_small = 0
    for op in ops:
        if op[1] == _ffi.OP_QUERY and op[3] > 0 and op[4] & _ffi.F_ACK:
            try:
                so = o.query_status(op[3])
            except _ffi.SimError:   # its tracker was taken over by a later query with the same residue
                continue
            assert g.query_status(op[3]) == so
            for which in (0, 1):   # and WHO they are (QueryResponse::ack_rx / response_rx): the same nodes, not only as many
                who = g.query_responders(op[3], which)
                assert who == o.query_responders(op[3], which) and len(who) == so[which] and who == sorted(set(who))
            n_checked += 1
            n_filtered_small += op[3] in filtered and so[0] < n // 2
    assert n_checked > 10 and n_filtered_small > 3, "the scenario must exercise filtered queries"
    assert g.cluster_stats() == o.cluster_stats()
    g2.step(79)
    o2.step(79)
    assert g2.digest() == o2.digest() == g.digest()


def test_event_log_overflow_is_counted(oracle, hiplib):
    """The device event log holds 2^20 events between two drains (include/serf_sim.h): what does not fit is counted in
    cluster_stats.events_lost, never silently clipped.  2048 watched nodes x 640 user events = 1.3 M events."""
    n = 2048
    kw = dict(fanout=4, view_slots=0, event_ring=1024, query_ring=8, retransmit_mult=2)
    g, o = pair(oracle, hiplib, n, **kw)
    for s in (g, o):
        for w in range(n):
            s.watch(w)
        for i in range(640):     # one origin, one event per tick: within the queue's capacity, distinct Lamport times
            s.inject(i, _ffi.OP_USER_EVENT, 3, 1 + i, 32)
        s.step(680)
    assert g.digest() == o.digest()
    assert o.cluster_stats()["overflow"] == 0

    def drain_all(s):
        tot = 0
        while True:
            ev = s.drain_events(1 << 18)
            tot += len(ev)
            if not ev:
                return tot
    lost = g.cluster_stats()["events_lost"]      # readable before the drain
    total_o, total_g = drain_all(o), drain_all(g)
    assert n * 640 - 100 < total_o <= n * 640 and total_o > 1 << 20   # a handful of (node, event) pairs are missed by gossip
    assert total_g == 1 << 20 and lost == total_o - total_g
    assert g.cluster_stats()["events_lost"] == lost and o.cluster_stats()["events_lost"] == 0
    g.step(5)       # the log works again after the overflow
    o.step(5)
    g.inject(0, _ffi.OP_USER_EVENT, 3, 5000, 32)
    o.inject(0, _ffi.OP_USER_EVENT, 3, 5000, 32)
    g.step(20)
    o.step(20)
    assert g.drain_events(1 << 18) == o.drain_events(1 << 18)


def _one_rank_rccl_worker(port, q):
    """ShardedSim with a world of one rank, the round's all-to-all issued by the library over RCCL (sim_exchange_*), against the
    plain single-handle run and the oracle."""
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    try:
        import torch
        import torch.distributed as dist

        import serf_amd
        from serf_amd.shard import ShardedSim
        from tests._oracle import load_oracle

        dist.init_process_group("gloo", rank=0, world_size=1)   # (the small collectives of the host path; the packets go over RCCL)
        lib, dev = serf_amd.load(), torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        n = 8192
        kw = dict(fanout=4, view_slots=64, event_ring=32, query_ring=16, probe_interval=4, loss=0.02, push_pull_interval=4,
                  recycle_interval=10, leave_delay=6)
        out = {}
        for chunks in (1, 2, 4):
            sh = ShardedSim(lib, n, dev, chunks=chunks, exchange="rccl", **kw)
            assert sh.use_lib and sh.collective_library().startswith("RCCL "), sh.collective_library()
            assert sh._lib_heads  # the lists of slot-less suspicions travel with the library's exchange (sim_suspect_import, heads == NULL)
            plain = _ffi.Sim(lib, _ffi.make_config(n, chunks=chunks if chunks > 1 else 0, **kw))
            orc = _ffi.Sim(load_oracle(), _ffi.make_config(n, chunks=chunks if chunks > 1 else 0, **kw))
            ops = sc.schedule(n, 40, rate=0.8, seed=23, max_member_subjects=20)
            for x in (sh, plain, orc):
                sc.apply_schedule(x, ops)
            for t in range(12):
                sh.step(5)
                plain.step(5)
                orc.step(5)
                sh.sync()
                # (word 2 of the digest is the packets in flight: a shard keeps them in its receive buffer, [chunk][peer][slot][sub] —
                # the plain handle's [slot][node] only when there is one chunk)
                keep = (lambda d: d) if chunks == 1 else (lambda d: d[:2] + d[3:])
                assert keep(sh.sim.digest()) == keep(plain.digest()) == keep(orc.digest()), f"chunks {chunks}: digests differ after tick {5 * t + 4}"
            out[chunks] = sh.collective_library()
            sh.close()
        # the random fan-out as one rank of the N > 1 path: the packed slabs (SIM_XCHG_PACKED) over the same grouped ncclSend / ncclRecv
        kw_rf = dict(kw, flags=_ffi.CF_BASELINE_JOINED | _ffi.CF_RANDOM_FANOUT, recycle_interval=0)
        for chunks in (1, 2, 4):   # (sender chunks: chunk c's slabs are packed and travel while chunk c + 1 computes)
            sh = ShardedSim(lib, n, dev, chunks=chunks, exchange="rccl", **kw_rf)
            assert sh.use_lib and sh._lib_heads and sh.kind == _ffi.XCHG_PACKED and sh.chunks == chunks
            plain = _ffi.Sim(lib, _ffi.make_config(n, **kw_rf))
            orc = _ffi.Sim(load_oracle(), _ffi.make_config(n, **kw_rf))
            for x in (sh, plain, orc):
                sc.apply_schedule(x, ops)
            for t in range(12):
                sh.step(5)
                plain.step(5)
                orc.step(5)
                sh.sync()
                assert sh.sim.digest() == plain.digest() == orc.digest(), f"random fan-out, packed slabs over RCCL, {chunks} chunk(s): digests differ after tick {5 * t + 4}"
            sh.close()
        dist.destroy_process_group()
        q.put(("ok", out[1]))
    except BaseException as e:  # noqa: BLE001
        q.put(("ERR", repr(e)))
        raise


def test_one_rank_through_rccl():
    # an RCCL call path that has executed.  One GPU, a world of one rank: the sharded instantiation of the
    # tick kernel, the exchange buffers, the host-driven push-pull / recycling / suspicion hand-over — and the round's
    # all-to-all as grouped ncclSend / ncclRecv issued by libserf_sim itself on its exchange stream (1, 2 and 4 chunks).
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_one_rank_rccl_worker, args=(29800 + os.getpid() % 150, q))
    p.start()
    p.join(300)
    status, what = q.get(timeout=5)
    assert status == "ok", what
    assert what.startswith("RCCL "), what
