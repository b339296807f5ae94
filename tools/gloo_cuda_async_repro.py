#!/usr/bin/env python
"""Does the round-2 stall (profiles/r02_shard_processes_check.txt: world 4 x 4 chunks on ONE GPU, every rank parked in
work.wait() of its asynchronous gloo all-to-alls) need the simulator at all?  This is ShardedSim's collective pattern
with the simulator taken out: W processes on cuda:0, gloo, per "tick" C asynchronous all_to_all_single calls on slices of
CUDA byte tensors (each behind a small kernel on the current stream, double-buffered receive side), all waited for at the
start of the next tick — nothing of serf_amd is imported.

usage: python tools/gloo_cuda_async_repro.py [world] [chunks] [ticks] [bytes_per_chunk] [device: cuda|cpu]
Prints one line per rank: "ok" or where it stopped making progress (run it under `timeout`)."""
import os
import sys
import time


def worker(rank, world, port, chunks, ticks, nbytes, device, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0) if device == "cuda" else torch.device("cpu")
    if device == "cuda":
        torch.cuda.set_device(0)
    send = torch.zeros(chunks * nbytes, dtype=torch.uint8, device=dev)
    recv = [torch.zeros(chunks * nbytes, dtype=torch.uint8, device=dev) for _ in range(2)]
    pending, last = [], time.monotonic()
    where = "start"
    try:
        for t in range(ticks):
            where = f"tick {t}: waiting for the {len(pending)} exchanges of tick {t - 1}"
            t0 = time.monotonic()
            for w in pending:
                w.wait()
            pending = []
            if time.monotonic() - t0 > 5:
                print(f"rank {rank}: {where} took {time.monotonic() - t0:.1f} s", flush=True)
            for c in range(chunks):
                lo = c * nbytes
                send[lo:lo + nbytes].add_(1)  # the "chunk launch": work on the current stream the collective has to wait for
                where = f"tick {t}: issuing chunk {c}"
                pending.append(dist.all_to_all_single(recv[t & 1][lo:lo + nbytes], send[lo:lo + nbytes], async_op=True))
            if t % 10 == 0 and time.monotonic() - last > 0:
                last = time.monotonic()
        for w in pending:
            w.wait()
        if device == "cuda":
            torch.cuda.synchronize()
        ok = int(recv[(ticks - 1) & 1][0]) == ticks % 256
        q.put((rank, "ok" if ok else f"finished but the data is wrong ({int(recv[(ticks - 1) & 1][0])} != {ticks % 256})"))
    except BaseException as e:  # noqa: BLE001
        q.put((rank, f"FAILED at {where}: {e!r}"))
    q.close()
    q.join_thread()
    os._exit(0)


def main():
    import torch.multiprocessing as mp

    world = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    ticks = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    nbytes = int(sys.argv[4]) if len(sys.argv) > 4 else 48 * 3 * 256 * world  # [V][f][sub] cells of 48 bytes
    device = sys.argv[5] if len(sys.argv) > 5 else "cuda"
    limit = float(os.environ.get("REPRO_LIMIT", "60"))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29400 + (os.getpid() % 300)
    procs = [ctx.Process(target=worker, args=(r, world, port, chunks, ticks, nbytes, device, q)) for r in range(world)]
    t0 = time.monotonic()
    for p in procs:
        p.start()
    res = []
    for _ in procs:
        try:
            res.append(q.get(timeout=max(1.0, limit - (time.monotonic() - t0))))
        except Exception:  # noqa: BLE001
            res.append((-1, f"no report within {limit:.0f} s: stalled"))
    for p in procs:
        p.join(3)
        if p.is_alive():
            p.kill()
    print(f"world {world} chunks {chunks} ticks {ticks} bytes/chunk {nbytes} device {device}: {time.monotonic() - t0:.1f} s")
    for r in sorted(res, key=lambda x: x[0]):
        print(" ", r[0], r[1])
    sys.exit(0 if all(str(r[1]) == "ok" for r in res) else 1)


if __name__ == "__main__":
    main()
