#!/usr/bin/env python
"""Rehearsal of the chunk-wise overlapped exchange on ONE GPU (gpurun exposes a single MI355X; the 8-GPU run is the
driver's): V shards = V handles of the product library on one device, each on its own compute stream, the
all-to-all stood in for by device-to-device copies on a separate copy stream.

Three schedules of the same run (same digests):
  kernel   : the tick kernels only, no exchange (what the compute costs when 4 shards share one GPU)
  serial   : every tick = all chunk launches, then the whole exchange, then the next tick (round 1's schedule)
  overlap  : the exchange of chunk c is enqueued on the copy stream as soon as chunk c's launch is, and runs while
             chunk c + 1 computes; only the last chunk's copies are exposed (serf_amd/shard.py over RCCL)
It shows the mechanism and its bookkeeping (events, double-buffered receive side), not xGMI bandwidth: on one GPU
the "exchange" is an HBM-to-HBM copy that competes with the kernels for the same memory system.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import serf_amd  # noqa: E402
from serf_amd import _ffi  # noqa: E402


def run(mode, V, n, chunks, ticks, preroll):
    lib = serf_amd.load()
    args = bench.parse_args(["--nodes-per-gpu", str(n // V), "--view-slots", "256", "--ring", "128"])
    kw, ops = bench.workload(args, n)
    comp = [torch.cuda.Stream() for _ in range(V)]
    copy = torch.cuda.Stream()
    shards, send, recv = [], [], []
    for g in range(V):
        s = _ffi.Sim(lib, _ffi.make_config(n, vshards=V, shard_rank=g, shard_count=V, chunks=chunks if chunks > 1 else 0, **kw))
        nb = s.exchange_bytes()
        send.append(torch.zeros(nb, dtype=torch.uint8, device="cuda"))
        recv.append([torch.zeros(nb, dtype=torch.uint8, device="cuda") for _ in range(2)])
        s.set_stream(comp[g].cuda_stream)
        s.bind_exchange2(send[-1].data_ptr(), recv[-1][0].data_ptr(), recv[-1][1].data_ptr())
        for o in ops:
            s.inject(*o)
        shards.append(s)
    region = send[0].numel() // chunks
    slab = region // V
    torch.cuda.synchronize()

    def exchange_chunk(t, c):  # on the current stream
        for g in range(V):
            for src in range(V):
                recv[g][t & 1][c * region + src * slab:c * region + (src + 1) * slab].copy_(
                    send[src][c * region + g * slab:c * region + (g + 1) * slab], non_blocking=True)

    def tick(t, timed_mode):
        for s in shards:
            s.step_begin()
        evs = []
        for c in range(chunks):
            for g, s in enumerate(shards):
                s.step_chunk(c)
            if timed_mode == "kernel":
                continue
            done = []
            for g in range(V):
                e = torch.cuda.Event()
                e.record(comp[g])
                done.append(e)
            if timed_mode == "overlap":
                with torch.cuda.stream(copy):
                    for e in done:
                        copy.wait_event(e)
                    exchange_chunk(t, c)
            else:
                evs.append(done)
        if timed_mode == "serial":
            with torch.cuda.stream(copy):
                for done in evs:
                    for e in done:
                        copy.wait_event(e)
                for c in range(chunks):
                    exchange_chunk(t, c)
        for s in shards:
            s.step_end()
        if timed_mode != "kernel":  # the next tick's kernels wait for the whole exchange
            fin = torch.cuda.Event()
            fin.record(copy)
            for st in comp:
                st.wait_event(fin)

    for t in range(preroll):
        tick(t, "overlap" if mode != "kernel" else "overlap")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(preroll, preroll + ticks):
        tick(t, mode)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / ticks * 1e3
    digs = [s.digest()[:2] for s in shards] if mode != "kernel" else None
    for s in shards:
        s.close()
    return dt, digs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shards", type=int, default=4)
    ap.add_argument("--nodes", type=int, default=1 << 20)
    ap.add_argument("--chunks", type=int, default=4)
    ap.add_argument("--ticks", type=int, default=100)
    ap.add_argument("--preroll", type=int, default=120)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    res = {}
    for mode, ch in (("kernel", a.chunks), ("serial", 1), ("serial", a.chunks), ("overlap", a.chunks)):
        dt, digs = run(mode, a.shards, a.nodes, ch, a.ticks, a.preroll)
        res[f"{mode}_c{ch}"] = {"ms_per_tick": dt, "digests": [[f"{x:016x}" for x in d] for d in digs] if digs else None}
        print(mode, ch, f"{dt:.3f} ms/tick", flush=True)
    same = res[f"serial_c{a.chunks}"]["digests"] == res[f"overlap_c{a.chunks}"]["digests"]
    out = {"config": vars(a), "what": __doc__.split("\n\n")[1], "results": {k: v["ms_per_tick"] for k, v in res.items()},
           "serial_and_overlap_same_state": same,
           "exchange_bytes_per_shard_per_tick": 4 * (a.nodes // a.shards) * 64}
    print(json.dumps(out))
    if a.out:
        json.dump(out, open(a.out, "w"), indent=1)
    assert same


if __name__ == "__main__":
    main()
