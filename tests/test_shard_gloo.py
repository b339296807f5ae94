"""N > 1 path on CPU: two processes (gloo), one shard each, ONE all_to_all_single per tick
(serf_amd/shard.py).  The compute behind the exchange is the CPU oracle here (this container has no
GPU); on the GPU box the same ShardedSim drives the HIP library over RCCL.  Each rank checks its
shard against the matching slice of a single-process run of the same cluster."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n, ticks, swim, chunks, q, jitter=0.0, pkt=0, loss=0.02, rc=0, rf=0):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    from serf_amd import _ffi
    from serf_amd.shard import ShardedSim
    from tests import _scenario as sc
    from tests._oracle import load_oracle

    dist.init_process_group("gloo", rank=rank, world_size=world)
    if jitter:
        # every collective is issued, and every asynchronous one is waited for, after a random per-rank delay: the ranks
        # drift apart by whole chunks, which is what four processes sharing one GPU did in the round-2 hang
        # (profiles/r02_shard_processes_check.txt) — if ShardedSim's bookkeeping (_drain, the double-buffered receive
        # side, the push-pull and recycling collectives between the chunk exchanges) depended on timing, it shows here
        import random
        import time
        rnd = random.Random(1000 + rank)
        real = dist.all_to_all_single

        class _Late:
            def __init__(self, w):
                self.w = w

            def wait(self):
                time.sleep(rnd.random() * jitter)
                return self.w.wait()

        def late(*a, **k):
            time.sleep(rnd.random() * jitter)
            w = real(*a, **k)
            return _Late(w) if k.get("async_op") else w
        dist.all_to_all_single = late
    try:
        lib = load_oracle()
        kw = dict(fanout=3, view_slots=64 if loss < 0.05 else 256, event_ring=16, query_ring=8, leave_delay=6, probe_interval=swim, loss=loss,
                  push_pull_interval=4 if swim else 0, pkt_records=pkt, reconnect_interval=rc,
                  prune_delay=bool(swim),   # handle_prune's wait: the notes cross the shards on the request list, like the slot-less suspicions
                  **(dict(suspicion_mult=3, suspicion_max_mult=2, gossip_to_the_dead=1) if rc else {}),
                  **(dict(flags=_ffi.CF_BASELINE_JOINED | _ffi.CF_RANDOM_FANOUT) if rf else {}))
        sh = ShardedSim(lib, n, torch.device("cpu"), chunks=chunks, **kw)
        # all shards in one process (memberlist's kRandomNodes: sender chunks are a shard's exchange schedule, nothing a single handle has)
        ref = _ffi.Sim(lib, _ffi.make_config(n, vshards=world, chunks=chunks if chunks > 1 and not rf else 0, **kw))
        ops = sc.schedule(n, ticks // 2, rate=0.7 if not pkt else 3.0, seed=17, max_member_subjects=40)
        # query filters and tag classes are replicated tables: every shard applies the same operations
        ops, classes = sc.with_filters(ops, n, tag_changes=6 if swim else 0)
        sh.init_tags(classes)
        ref.init_tags(classes)
        for t, op, node, a, b in ops:
            sh.inject(t, op, node, a, b)
            ref.inject(t, op, node, a, b)
        for x in ((7, 300, 777, 1000) if rc else ()):   # nodes that go down and are declared failed; two of them resume when nobody
            for s_ in (sh, ref):                         # gossips to them any more: only a peer's Reconnector brings those back
                s_.inject(1, _ffi.OP_CRASH, x)
                if x > 500:
                    s_.inject(50, _ffi.OP_REVIVE, x)
        m = n // world
        lo = rank * m
        vs = kw["view_slots"]
        handed = [0, 0]
        if swim:
            real_import = sh.sim.suspect_import

            def counting(of_tick, ptr, w):   # how many slot-less suspicions crossed the shards
                hs = sh._sq_host[of_tick % len(sh._sq_host)].view(w, -1)
                handed[0] += int(hs[:, 0].sum())
                for row in hs:   # entries with bit 31 of the second word: the Reconnector's attempts
                    handed[1] += int((row[2:1 + 2 * int(row[0]):2] < 0).sum())
                return real_import(of_tick, ptr, w)
            sh.sim.suspect_import = counting
        for t in range(0, ticks, 5):
            sh.step(5)
            ref.step(5)
            for which in (_ffi.ARR_ROWS, _ffi.ARR_QUEUE):
                a = sh.sim.dump(which)
                b = ref.dump(which)
                per = len(b) // n
                assert (a.tobytes() == b[lo * per:(lo + m) * per].tobytes()), f"rank {rank} array {which} differs at tick {t + 5}"
            for which, rows in ((_ffi.ARR_VIEW, vs), (_ffi.ARR_ERING, 16), (_ffi.ARR_QRING, 8)):
                a = sh.sim.dump(which).reshape(rows, m)
                b = ref.dump(which).reshape(rows, n)[:, lo:lo + m]
                assert a.tobytes() == np.ascontiguousarray(b).tobytes(), f"rank {rank} array {which} differs at tick {t + 5}"
            if rf:   # the packets in flight, canonical form of the mode: in their senders' cells — the shard holds its own senders'
                fp = 3 * max(1, pkt // 4)
                a = sh.sim.dump(_ffi.ARR_INBOX).reshape(fp, m)
                b = ref.dump(_ffi.ARR_INBOX).reshape(fp, n)[:, lo:lo + m]
                assert a.tobytes() == np.ascontiguousarray(b).tobytes(), f"rank {rank} packets in flight differ at tick {t + 5}"
        ev = next(op for op in ops if op[1] == _ffi.OP_USER_EVENT)
        assert sh.convergence(_ffi.K_EVENT, ev[3], 1) == ref.convergence(_ffi.K_EVENT, ev[3], 1)
        for qop in [op for op in ops if op[1] == _ffi.OP_QUERY and op[4] & _ffi.F_ACK][:3]:
            assert sh.query_status(qop[3]) == ref.query_status(qop[3])   # acks summed over the shards
        if loss >= 0.05:
            # a probe that fails on a member without a view slot: every shard's list of tick t, gathered behind the tick
            # and replayed at t + 2 on every shard (sim_suspect_export / _import), must reproduce the single-process run
            assert handed[0] > 20, handed
        if rc and not rf:   # (scenario properties tuned on the bijection; with the random fan-out the parity checks above are the test)
            # Reconnector attempts cross the shards like the suspicions (request list -> SIM_OP_RECONNECT on every shard) and
            # run as push-pull pairs of their tick through sim_pp_plan / _export / _merge, most of them between two shards
            assert handed[1] > 20, handed
            st = ref.members(5)[0]
            assert [int(st[x]) for x in (777, 1000)] == [_ffi.STATUS_ALIVE] * 2 and int(st[7]) == _ffi.STATUS_FAILED, "the Reconnector brought the resumed nodes back"
        q.put((rank, "ok"))
    except BaseException as e:  # noqa: BLE001 — report to the parent, then re-raise
        q.put((rank, repr(e)))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,swim,n,pkt,loss,rc,chunks", [(2, 0, 1024, 0, .02, 0, 1), (2, 4, 1024, 0, .02, 0, 1), (4, 2, 2048, 8, .12, 2, 1), (4, 1, 1024, 16, .0, 3, 1),
                                                             (1, 4, 1024, 0, .02, 0, 1),
                                                             # sender chunks (r5): chunk c's slabs are packed and travel while chunk c + 1 computes
                                                             (2, 4, 1024, 0, .02, 0, 2), (4, 2, 2048, 8, .12, 2, 4), (1, 4, 1024, 0, .02, 0, 2)])
def test_shards_gloo_random_fanout_match_single_process(world, swim, n, pkt, loss, rc, chunks):
    # memberlist's kRandomNodes on shards (r5: the scalable form): a packet goes to ANY node of the cluster, so the packets stay
    # in their senders' cells; every shard sorts the (target, sender, slot) triples of its OWN senders, packs the packets bound
    # for shard h into slab h and the round's exchange is one equal-split all-to-all of those slabs (sim_exchange_layout:
    # SIM_XCHG_PACKED); nobody draws anybody else's targets.  Against ONE handle that holds every node.
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    from tests._scenario import free_port
    port = free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, 60 if not rc else 100, swim, chunks, q, 0.0, pkt, loss, rc, 1)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
    res = sorted(q.get(timeout=5) for _ in procs)
    assert res == [(r, "ok") for r in range(world)], res


@pytest.mark.parametrize("world,chunks,swim,n,jitter,pkt,loss,rc", [(2, 1, 0, 1024, 0, 0, .02, 0), (2, 1, 4, 1024, 0, 0, .02, 0), (2, 2, 4, 1024, 0, 0, .02, 0),
                                                                    (4, 2, 4, 1024, 0, 0, .02, 0), (4, 4, 0, 1024, 0, 0, .02, 0), (4, 4, 4, 4096, 0.004, 0, .02, 0),
                                                                    (2, 2, 4, 1024, 0, 16, .02, 0), (2, 1, 2, 1024, 0, 0, .12, 0), (4, 2, 2, 2048, 0, 16, .12, 0),
                                                                    (2, 2, 1, 1024, 0, 0, .02, 2), (4, 1, 1, 1024, 0, 0, .0, 3),
                                                                    # a world of ONE rank: the same path (SIM_CF_FORCE_SHARDED) against the plain handle
                                                                    (1, 1, 4, 1024, 0, 0, .02, 0), (1, 2, 2, 1024, 0, 8, .12, 2)])
def test_shards_gloo_match_single_process(world, chunks, swim, n, jitter, pkt, loss, rc):
    # chunks > 1: the tick runs as `chunks` launches, each followed by the asynchronous all-to-all of its slabs
    # (double-buffered receive side) — the overlapped path of serf_amd/shard.py
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    from tests._scenario import free_port
    port = free_port()
    # (the last case is the configuration that stalled on one GPU in round 2 — world 4, 4 chunks, SWIM and push-pull
    # batches, 4 096 nodes — here with CPU tensors and every collective randomly delayed)
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, 60 if not rc else 100, swim, chunks, q, jitter, pkt, loss, rc)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
    res = sorted(q.get(timeout=5) for _ in procs)
    assert res == [(r, "ok") for r in range(world)], res
