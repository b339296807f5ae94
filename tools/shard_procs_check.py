#!/usr/bin/env python
"""One PROCESS per shard over the HIP library on ONE GPU (every rank on cuda:0, gloo standing in for RCCL, which refuses
two ranks on one device): each rank compares every array of its shard, every 5 ticks, with the matching slice of a
single-process run of the CPU oracle (test infrastructure).  The multi-process twin of
tests/test_parity_gpu.py::test_sharded_kernel_four_shards_on_one_gpu; run it on the GPU box under `timeout`.

usage: python tools/shard_procs_check.py [world] [chunks] [swim] [nodes] [loss] [rf]
(rf = 1: memberlist's kRandomNodes — the round's exchange is the all-to-all of the packed slabs, SIM_XCHG_PACKED; chunks must be 1)
(loss >= 0.05: 256 view slots, and the run must have carried slot-less suspicions across the shards)"""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(rank, world, port, n, ticks, swim, chunks, q, loss=0.02, rf=0):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import numpy as np
    import torch
    import torch.distributed as dist

    import serf_amd
    from serf_amd import _ffi
    from serf_amd.shard import ShardedSim
    from tests import _scenario as sc
    from tests._oracle import load_oracle

    import faulthandler
    global _stack_file  # (kept referenced: faulthandler writes to its descriptor)
    _stack_file = open(os.path.join(ROOT, "gpurun_out", f"stack_rank{rank}.txt"), "w")
    faulthandler.dump_traceback_later(25, exit=False, file=_stack_file)
    prog = open(os.path.join(ROOT, "gpurun_out", f"progress_rank{rank}.txt"), "w")

    def mark(what):
        prog.write(what + "\n")
        prog.flush()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    where = "start"
    try:
        vs = 64 if loss < 0.05 else 256
        kw = dict(fanout=3, view_slots=vs, event_ring=16, query_ring=8, leave_delay=6, probe_interval=swim, loss=loss,
                  push_pull_interval=4 if swim else 0)
        if rf:
            kw["flags"] = _ffi.CF_BASELINE_JOINED | _ffi.CF_RANDOM_FANOUT
        mark("creating")
        sh = ShardedSim(serf_amd.load(), n, dev, chunks=chunks, **kw)
        mark("created")
        ref = _ffi.Sim(load_oracle(), _ffi.make_config(n, vshards=world, chunks=chunks if chunks > 1 else 0, **kw))
        ops = sc.schedule(n, ticks // 2, rate=0.7, seed=17, max_member_subjects=40)
        ops, classes = sc.with_filters(ops, n, tag_changes=6 if swim else 0)
        sh.init_tags(classes)
        ref.init_tags(classes)
        for t, op, node, a, b in ops:
            sh.inject(t, op, node, a, b)
            ref.inject(t, op, node, a, b)
        m = n // world
        lo = rank * m
        handed = [0]
        if swim:
            real_import = sh.sim.suspect_import

            def counting(of_tick, ptr, w):
                handed[0] += int(sh._sq_host[of_tick % len(sh._sq_host)].view(w, -1)[:, 0].sum())
                return real_import(of_tick, ptr, w)
            sh.sim.suspect_import = counting
        for t in range(0, ticks, 5):
            where = f"step to tick {t + 5}"
            mark(where)
            for _ in range(5):
                sh.step(1)
                mark(f"  hip tick {sh.sim.tick}")
            mark("  hip stepped")
            ref.step(5)
            mark("  oracle stepped")
            sh.sync()
            mark("  synced")
            for which in (_ffi.ARR_ROWS, _ffi.ARR_QUEUE):
                where = f"tick {t + 5} array {which}"
                a = sh.sim.dump(which)
                b = ref.dump(which)
                per = len(b) // n
                if a.tobytes() != b[lo * per:(lo + m) * per].tobytes():
                    raise AssertionError(f"rank {rank} array {which} differs at tick {t + 5}")
            for which, rows in ((_ffi.ARR_VIEW, vs), (_ffi.ARR_ERING, 16), (_ffi.ARR_QRING, 8)):
                where = f"tick {t + 5} array {which}"
                a = sh.sim.dump(which).reshape(rows, m)
                b = ref.dump(which).reshape(rows, n)[:, lo:lo + m]
                if a.tobytes() != np.ascontiguousarray(b).tobytes():
                    raise AssertionError(f"rank {rank} array {which} differs at tick {t + 5}")
            if rf:  # the packets in flight: the shard's own senders' cells
                a = sh.sim.dump(_ffi.ARR_INBOX).reshape(3, m)
                b = ref.dump(_ffi.ARR_INBOX).reshape(3, n)[:, lo:lo + m]
                if a.tobytes() != np.ascontiguousarray(b).tobytes():
                    raise AssertionError(f"rank {rank} packets in flight differ at tick {t + 5}")
        where = "convergence"
        ev = next(op for op in ops if op[1] == _ffi.OP_USER_EVENT)
        assert sh.convergence(_ffi.K_EVENT, ev[3], 1) == ref.convergence(_ffi.K_EVENT, ev[3], 1)
        mark("convergence agrees")
        where = "query_status"
        for qop in [op for op in ops if op[1] == _ffi.OP_QUERY and op[4] & _ffi.F_ACK][:3]:
            assert sh.query_status(qop[3]) == ref.query_status(qop[3])  # acks summed over the shards
        mark("query status agrees")
        if loss >= 0.05 and handed[0] <= 20:
            raise AssertionError(f"only {handed[0]} slot-less suspicions crossed the shards")
        q.put((rank, f"ok: {ticks} ticks, every array of the shard equal to the oracle's slice at every 5th tick; {handed[0]} slot-less suspicions handed over"))
    except BaseException as e:  # noqa: BLE001
        q.put((rank, f"FAILED at {where}: {e!r}\n{traceback.format_exc()}"))
    q.close()
    q.join_thread()  # the report is on its way before the process goes
    mark("reported")
    os._exit(0)  # no collective teardown: a rank that failed must not hang the others


def main():
    import torch.multiprocessing as mp

    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    swim = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    n = int(sys.argv[4]) if len(sys.argv) > 4 else 2048
    loss = float(sys.argv[5]) if len(sys.argv) > 5 else 0.02
    rf = int(sys.argv[6]) if len(sys.argv) > 6 else 0
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29300 + (os.getpid() % 300)
    procs = [ctx.Process(target=worker, args=(r, world, port, n, 60, swim, chunks, q, loss, rf)) for r in range(world)]
    for p in procs:
        p.start()
    res = []
    for _ in procs:
        try:
            res.append(q.get(timeout=35))
        except Exception:  # noqa: BLE001
            res.append((-1, "no report within 35 s"))
    for p in procs:
        p.join(5)
        if p.is_alive():
            p.kill()
    for r in sorted(res, key=lambda x: x[0]):
        print(r[0], r[1])
    sys.exit(0 if all(str(r[1]).startswith("ok") for r in res) else 1)


if __name__ == "__main__":
    main()
