"""Parity at the sizes BASELINE.json names: the HIP library against the CPU oracle at 4 Mi nodes / 4 virtual
shards on BOTH fan-out models (configs[3]'s cluster in one handle), at one rank's share of configs[4] (2 Mi nodes, 8 virtual
shards, 2 sender chunks, churn + 1 % loss, packets of 16 records, 200 ticks), and on memberlist's kRandomNodes above 4 Mi nodes
(64-bit entries in the graph build's sort).  `sim_state_digest` — all eight arrays: rows, queues, packets in flight, views, both
rings, slot map + liveness, query tables — every 10th tick and at the end; views and rings are small (32 / 128 view slots) so that
the oracle's arrays fit the host (the protocol does not depend on how many slots are spare: tests/test_oracle_unbounded.py).
The oracle is the slow side: ~0.2 - 0.5 s per tick at these sizes on the box's host cores; runtimes are printed."""
import time

import numpy as np
import pytest

from serf_amd import _ffi
from tests import _scenario as sc

pytestmark = pytest.mark.gpu


def _run(oracle, hiplib, n, ops, ticks, every, what, **kw):
    t0 = time.perf_counter()
    g = _ffi.Sim(hiplib, _ffi.make_config(n, **kw))
    o = _ffi.Sim(oracle, _ffi.make_config(n, **kw))
    for x in (g, o):
        for op in ops:
            x.inject(*op)
    done = 0
    while done < ticks:
        k = min(every, ticks - done)
        g.step(k)
        o.step(k)
        done += k
        dg, do = g.digest(), o.digest()
        assert dg == do, f"{what}: digests differ after tick {done - 1}: arrays {[i for i in range(8) if dg[i] != do[i]]}"
    cg, co = g.cluster_stats(), o.cluster_stats()
    assert cg == co, f"{what}: load figures differ {cg} {co}"
    print(f"{what}: {n} nodes x {ticks} ticks bit-exact ({ticks // every} digests), {time.perf_counter() - t0:.0f} s; "
          f"drops {cg['overflow']}, slots in use {cg['slots_in_use']}, failed {cg['failed']}, left {cg['left']}")
    g.close()
    o.close()
    return cg


@pytest.mark.parametrize("model", ["krandomnodes", "bijection"])
def test_4mi_nodes_4_vshards_both_models(oracle, hiplib, model):
    # BASELINE configs[3]'s cluster (4 Mi nodes sharded 4-way) as ONE handle: 4 virtual shards — the bijection's map has the
    # sharded shape (vblocks, rotations), kRandomNodes draws over all 4 Mi nodes (64-bit entries in the graph build: 24 + 10 bits)
    n = 1 << 22
    kw = dict(fanout=4, vshards=4, view_slots=32, event_ring=32, query_ring=16, probe_interval=5, loss=0.01, push_pull_interval=20,
              leave_delay=6, reap_interval=15, queue_check_interval=30, recycle_interval=25)
    if model == "krandomnodes":
        kw["flags"] = _ffi.CF_BASELINE_JOINED | _ffi.CF_RANDOM_FANOUT
    ops = sc.schedule(n, 45, rate=0.5, seed=41, max_member_subjects=14)
    _run(oracle, hiplib, n, ops, 60, 10, f"4 Mi nodes, 4 vshards, {model}", **kw)


def test_one_ranks_share_of_config4_2mi_8_vshards_2_chunks_churn_loss_16_records(oracle, hiplib):
    # BASELINE configs[4] (16 Mi nodes sharded 8-way, 5 % churn + 1 % loss): one rank's share — 2 Mi nodes in the shape of an
    # 8-way sharded cluster (8 virtual shards, 2 sender chunks), packets of 16 records, the failure detector with memberlist's
    # stream-transport fallback ping and the join sync (what the configs[4] runs use), crash + re-join
    # churn at the pace the view slots allow, 1 % loss on every packet and probe leg, view-slot recycling — 200 ticks
    n = 1 << 21
    kw = dict(fanout=4, vshards=8, chunks=2, view_slots=128, event_ring=32, query_ring=16, probe_interval=5, loss=0.01,
              push_pull_interval=30, leave_delay=6, reap_interval=15, queue_check_interval=30, recycle_interval=25, pkt_records=16,
              suspicion_mult=3, suspicion_max_mult=2, tcp_fallback=True, join_sync=True)
    rng = np.random.default_rng(9)
    churned = rng.choice(n, 60, replace=False).tolist()
    ops = []
    for i, node in enumerate(churned):       # one crash every 3 ticks, down 40 ticks (suspected, confirmed, declared failed: the
        ops.append((5 + 3 * i, _ffi.OP_CRASH, node, 0, 0))   # suspicion timeout is 3 x 6 x 5 = 94 ticks at most here), then Serf::join
        ops.append((5 + 3 * i + 40, _ffi.OP_JOIN, node, int(rng.integers(0, n)), 0))
    for i in range(60):                      # rumours on top: user events and queries
        ops.append((3 * i + 1, _ffi.OP_USER_EVENT, int(rng.integers(0, n)), 5000 + i, 40))
        if i % 3 == 0:
            ops.append((3 * i + 2, _ffi.OP_QUERY, int(rng.integers(0, n)), 300 + i, 0))
    ops.sort(key=lambda o: o[0])
    cs = _run(oracle, hiplib, n, ops, 200, 10, "2 Mi nodes, 8 vshards, 2 chunks, churn + 1 % loss, 16-record packets", **kw)
    assert cs["failed"] > 0 or cs["left"] > 0 or cs["slots_in_use"] > 0   # the churn happened


def test_prune_wait_at_1mi_nodes_the_request_list_bound_bites_the_same_way(oracle, hiplib):
    # SIM_CF_PRUNE_DELAY at BASELINE configs[2]'s size: three pruning removals of running and of crashed members.  Every node notes its wait on
    # the tick's request list as the intent's wavefront passes — far more than SIM_SUSPECT_REQ_MAX a tick at this size: those ticks' lists are
    # dropped and counted, the ticks at the wavefront's head and tail are replayed — and the library and the oracle must drop and replay the
    # very same ones (ops_dropped is part of the load figures compared below)
    n = 1 << 20
    kw = dict(fanout=4, view_slots=32, event_ring=32, query_ring=16, probe_interval=5, loss=0.01, push_pull_interval=20, leave_delay=7,
              reap_interval=15, recycle_interval=25, prune_delay=True, flags=_ffi.CF_BASELINE_JOINED | _ffi.CF_RANDOM_FANOUT)
    ops = [(2, _ffi.OP_FORCE_LEAVE, 11, 500000, 1), (4, _ffi.OP_CRASH, 777, 0, 0), (5, _ffi.OP_FORCE_LEAVE, 90000, 777, 1),
           (9, _ffi.OP_USER_EVENT, 3, 4242, 40), (12, _ffi.OP_FORCE_LEAVE, 1000000, 31, 1), (14, _ffi.OP_QUERY, 5, 77, 0)]
    cs = _run(oracle, hiplib, n, ops, 50, 5, "1 Mi nodes, handle_prune's wait, kRandomNodes", **kw)
    assert cs["ops_dropped"] > 0, "the request list's bound was meant to bite"


@pytest.mark.parametrize("mi,ticks", [(6, 30), (8, 20)])
def test_krandomnodes_above_4mi_nodes(oracle, hiplib, mi, ticks):
    # memberlist's kRandomNodes at 6 Mi nodes: pair ids of 25 bits, 64-bit entries in rf_scatter / rf_rows, 2 048 senders per
    # workgroup — the path profiles/r04_size_sweep.json timed and nothing checked; and at 8 Mi = 2^23 nodes, the
    # largest cluster ONE handle takes (profiles/r05_size_sweep.json: 8 Mi nodes at the bench's own view / ring sizes fit one GPU
    # since the planes get their memory on demand).  View slots and rings small, so that the oracle's whole arrays fit the host.
    n = mi << 20
    kw = dict(fanout=4, view_slots=16, event_ring=16, query_ring=8, probe_interval=5, loss=0.01, push_pull_interval=20, leave_delay=6,
              flags=_ffi.CF_BASELINE_JOINED | _ffi.CF_RANDOM_FANOUT)
    ops = sc.schedule(n, ticks - 5, rate=0.5, seed=77, max_member_subjects=7)
    _run(oracle, hiplib, n, ops, ticks, 10, f"{mi} Mi nodes, kRandomNodes (64-bit sort entries)", **kw)


def test_config3_as_four_shard_handles_through_the_packed_exchange(oracle, hiplib):
    # BASELINE configs[3] in its SHARDED form on the headline model: 4 Mi nodes as the four shard handles of four ranks (1 Mi nodes each,
    # memberlist's kRandomNodes, 2 sender chunks), on one GPU — every shard sorts its own senders' (target, sender, slot) triples, packs
    # the packets per destination behind the chunk's launch, the round's equal-split all-to-all of the packed slabs (SIM_XCHG_PACKED) is
    # done with device copies, the cross-shard push-pull batch and the suspicions' hand-over as a sharded host does them — against ONE
    # oracle handle that holds all 4 Mi nodes: every shard's rows and queues at tick 19, every array at tick 39, slice by slice.
    import torch

    from tests.test_parity_gpu import _packed_exchange_chunk_on_one_gpu, _push_pull_on_one_gpu, _suspicions_on_one_gpu

    n, V, C, A, BE, BQ, ticks = 1 << 22, 4, 2, 16, 16, 8, 40
    m = n // V
    kw = dict(fanout=4, view_slots=A, event_ring=BE, query_ring=BQ, leave_delay=6, probe_interval=5, loss=0.01, push_pull_interval=20,
              flags=_ffi.CF_BASELINE_JOINED | _ffi.CF_RANDOM_FANOUT)
    t0 = time.perf_counter()
    ref = _ffi.Sim(oracle, _ffi.make_config(n, vshards=V, **kw))
    shards, send, recv = [], [], []
    for g in range(V):
        s = _ffi.Sim(hiplib, _ffi.make_config(n, vshards=V, shard_rank=g, shard_count=V, chunks=C, **kw))
        kind, planes, pb, rb = s.exchange_layout()
        assert kind == _ffi.XCHG_PACKED and pb <= 1.06 * 4 * 64 * m       # what a shard sends per tick: f x 64 B per node + 5 % (room of 12 sigma per slab and chunk, a count byte per target)
        send.append(torch.zeros(pb, dtype=torch.uint8, device="cuda"))
        recv.append([torch.zeros(rb, dtype=torch.uint8, device="cuda") for _ in range(2)])
        s.bind_exchange3(send[-1].data_ptr(), pb, recv[-1][0].data_ptr(), recv[-1][1].data_ptr(), rb)
        shards.append(s)
    ops = sc.schedule(n, 30, rate=0.6, seed=23, max_member_subjects=7)
    for s in shards + [ref]:
        sc.apply_schedule(s, ops)
    for t in range(ticks):
        for s in shards:
            s.step_begin()
        if shards[0].pp_due():
            _push_pull_on_one_gpu(shards)
        into = [r[shards[0].tick & 1] for r in recv]
        for c in range(C):
            for s in shards:
                s.step_chunk(c)
                s.sync()
            _packed_exchange_chunk_on_one_gpu(send, into, c, C)
        for s in shards:
            s.step_end()
            s.sync()
        _suspicions_on_one_gpu(shards)
        torch.cuda.synchronize()
        ref.step(1)
        if t in (19, ticks - 1):
            arrays = [(_ffi.ARR_ROWS, None), (_ffi.ARR_QUEUE, None)]
            if t == ticks - 1:
                arrays += [(_ffi.ARR_VIEW, A), (_ffi.ARR_ERING, BE), (_ffi.ARR_QRING, BQ), (_ffi.ARR_INBOX, 4)]
            for which, rows in arrays:
                b = ref.dump(which)
                for g, s in enumerate(shards):
                    a, lo = s.dump(which), g * m
                    if rows is None:
                        per = len(b) // n
                        i = sc.first_diff(a, b[lo * per:(lo + m) * per])
                        assert i is None, f"shard {g} array {which} element {i} differs at tick {t}"
                    else:
                        assert a.reshape(rows, m).tobytes() == np.ascontiguousarray(b.reshape(rows, n)[:, lo:lo + m]).tobytes(), f"shard {g} array {which} differs at tick {t}"
                del b
    cs = [s.cluster_stats() for s in shards]
    print(f"4 Mi nodes as 4 shard handles of 1 Mi, kRandomNodes, packed exchange, 2 chunks: {ticks} ticks bit-exact against one oracle handle, "
          f"{time.perf_counter() - t0:.0f} s; slab bytes per shard and tick {send[0].numel()}; drops {sum(c['overflow'] for c in cs)}")
    for s in shards + [ref]:
        s.close()
