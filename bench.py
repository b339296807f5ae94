#!/usr/bin/env python
"""bench.py — member-ticks/sec of the bulk SWIM/Serf gossip hot path on MI355X.

A "step" is one gossip tick of every simulated node.  N=1: BASELINE.json configs[2]
(1 Mi nodes, fan-out 4, HBM-roofline report).  N>1: one shard of 1 Mi nodes per GPU (weak
scaling), one RCCL all_to_all_single per tick.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# SURVEY.md §8d algorithmic bytes per member-tick (v0 layout): 2R + 2QE + 2fPE + 4(f+2)
def b_tick_v0(f):
    return 2 * 64 + 2 * 16 * 16 + 2 * f * 4 * 16 + 4 * (f + 2)


# the same accounting for the frozen layout: rows + (sort keys r/w + payload gathers)
# + packets written once and read once + one slot-map word and one 16-byte head per received record
def b_tick_layout(f):
    return 2 * 64 + (2 * 16 * 4 + f * 4 * 16) + 2 * f * 4 * 16 + f * 4 * (4 + 16)


# the benchmark workload: evenly spaced API operations, this mix of
# (user event, query, graceful leave [+ rejoin], crash + remove_failed_node, crash + revive)
MIX = (0.55, 0.2, 0.15, 0.05, 0.05)


PROFILE_EVERY = 4  # an event pair costs ~10 us of stream time: time a sample of the launches, not all of them


def cpu_baseline(fanout, probe_interval, push_pull_interval, rate, seconds_budget=20.0):
    """The CPU oracle ("port") on a bounded sample of the same workload, rank 0 only."""
    from serf_amd import _ffi
    from tests import _scenario as sc
    from tests._oracle import load_oracle

    lib = load_oracle()
    n, ticks = 1 << 18, 24
    sim = _ffi.Sim(lib, _ffi.make_config(n, fanout=fanout, view_slots=64, event_ring=64, query_ring=64,
                                         probe_interval=probe_interval, push_pull_interval=push_pull_interval,
                                         reap_interval=75, queue_check_interval=150))
    sc.apply_schedule(sim, sc.schedule(n, ticks, rate=rate, seed=11, mix=MIX, max_member_subjects=32, even=True))
    sim.step(4)  # warm-up (page faults, rumors in flight)
    t0 = time.perf_counter()
    done = 0
    while done < ticks and time.perf_counter() - t0 < seconds_budget:
        sim.step(4)
        done += 4
    dt = time.perf_counter() - t0
    cores = lib.dll.osim_t_threads()
    # the same run continued on one thread (SURVEY.md §8d asks for both legs), a few ticks only
    lib.dll.osim_t_set_threads(1)
    t1 = time.perf_counter()
    done1 = 0
    while done1 < 4 and time.perf_counter() - t1 < seconds_budget / 2:
        sim.step(1)
        done1 += 1
    dt1 = time.perf_counter() - t1
    lib.dll.osim_t_set_threads(0)
    sim.close()
    return {"value": n * done / dt, "unit": "member-ticks/s", "cores": int(cores), "kind": "port",
            "single_thread_value": n * done1 / dt1,
            "sample": f"{n} nodes x {done} ticks (all cores) + {done1} ticks (one thread), fan-out {fanout}, probe interval {probe_interval}, "
                      f"same operation mix and rate, view_slots=64 rings=64 (CPU oracle, OpenMP over nodes)"}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=60)
    ap.add_argument("--nodes-per-gpu", type=int, default=1 << 20)
    ap.add_argument("--fanout", type=int, default=4)
    ap.add_argument("--view-slots", type=int, default=1024)
    ap.add_argument("--ring", type=int, default=512)
    ap.add_argument("--rate", type=float, default=0.4, help="API operations injected per tick (cluster-wide)")
    ap.add_argument("--probe-interval", type=int, default=5, help="memberlist probe interval in ticks (0 = SWIM layer off)")
    ap.add_argument("--push-pull-interval", type=int, default=150, help="memberlist push_pull_interval in ticks before log2(N) scaling (0 = off)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-convergence", action="store_true", help="skip the rounds-to-99 %% measurement (profiling runs)")
    return ap.parse_args(argv)


def run(args, lib=None, dev=None, backend="nccl"):
    """The benchmark proper.  `lib`/`dev`/`backend` exist so that tests/test_bench_plumbing.py can drive the
    SAME control flow (sharded stepping, all-to-all, convergence, JSON) on CPU with gloo; main() always
    passes the HIP library, a CUDA device and RCCL."""
    import torch
    import torch.distributed as dist

    import serf_amd
    from serf_amd import _ffi
    from serf_amd.shard import ShardedSim
    from tests import _scenario as sc

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    on_gpu = dev is None
    if on_gpu:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if on_gpu:
            dist.init_process_group(backend, device_id=dev)
        else:
            dist.init_process_group(backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    n_total = args.nodes_per_gpu * world
    if lib is None:
        lib = serf_amd.load()
    kw = dict(fanout=args.fanout, view_slots=args.view_slots, event_ring=args.ring, query_ring=args.ring,
              probe_interval=args.probe_interval, push_pull_interval=args.push_pull_interval,
              reap_interval=75, queue_check_interval=150)  # options.rs defaults: reap 15 s, queue check 30 s, timeouts 24 h
    total_ticks = args.steps + args.warmup
    ops = sc.schedule(n_total, total_ticks, rate=args.rate, seed=3, mix=MIX, max_member_subjects=args.view_slots // 2, even=True)
    if world > 1:
        sim = ShardedSim(lib, n_total, dev, **kw)
        step, inject = sim.step, sim.inject
    else:
        sim = _ffi.Sim(lib, _ffi.make_config(n_total, **kw))
        if on_gpu:
            sim.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        step, inject = sim.step, sim.inject
    for t, op, node, a, b in ops:
        inject(t, op, node, a, b)

    def barrier():
        if world > 1:
            dist.barrier()
        if on_gpu:
            torch.cuda.synchronize()
        else:
            (sim.sim if world > 1 else sim).sync()

    class _HostEvent:  # CPU stand-in for torch.cuda.Event in the plumbing test
        def __init__(self, enable_timing=True):
            self.t = 0.0

        def record(self):
            self.t = time.perf_counter()

        def elapsed_time(self, other):
            return (other.t - self.t) * 1e3

    Event = torch.cuda.Event if on_gpu else _HostEvent

    step(args.warmup)
    barrier()
    raw = sim.sim if world > 1 else sim
    raw.profile(PROFILE_EVERY)  # HIP events around every 4th tick-kernel launch, on the stream it is launched on
    # The launches go to torch's current stream (sim_set_stream above), so one pair of torch events
    # brackets the K steps: at N=1 nothing but tick/ops kernels lies in between, at N>1 the all-to-alls
    # do as well (the kernel's own duration comes from sim_profile either way).
    t0 = time.perf_counter()
    ev0, ev1 = Event(enable_timing=True), Event(enable_timing=True)
    ev0.record()
    step(args.steps)
    ev1.record()
    barrier()
    dt = time.perf_counter() - t0
    ev_ms = ev0.elapsed_time(ev1)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0])

    prof_ms, prof_n = raw.profile_read()
    raw.profile(False)
    # second half of the metric: rounds to 99 % convergence, measured after the timed region on 8
    # fresh user events, one at a time, under the same background load (every rank issues the same
    # calls; the originator's rank reads the Lamport time the event is going to get)
    import numpy as _np
    rng = _np.random.default_rng(99)
    rounds = []
    for i in range(0 if args.no_convergence else 8):
        node, key = int(rng.integers(0, n_total)), 0x7F000000 + i
        owner = node // args.nodes_per_gpu
        lt = (sim.sim if world > 1 else sim).stats(node).event_time if owner == rank else 0
        if world > 1:
            t = torch.tensor([lt], dtype=torch.int64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            lt = int(t[0])
        sim.user_event(node, key, 64)
        got = None
        for r in range(1, 61):
            step(1)
            seen, up = sim.convergence(_ffi.K_EVENT, key, lt)
            if seen * 100 >= up * 99:
                got = r
                break
        rounds.append(got if got is not None else 61)
    if rank == 0:
        value = n_total * args.steps / dt
        bt, bt2 = b_tick_v0(args.fanout), b_tick_layout(args.fanout)
        # dominant kernel = tick_kernel: one launch per tick; HIP events around each launch of the timed
        # region (sim_profile); ev_ms (everything on the stream, ops + push-pull included) for reference
        kern_s = prof_ms / 1e3 / max(1, prof_n) if prof_ms > 0 else ev_ms / 1e3 / args.steps  # (the oracle behind a CPU test has no kernel)
        achieved = args.nodes_per_gpu * bt / kern_s / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "member-ticks/sec", "value": value, "unit": "member-ticks/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": f"{n_total} nodes ({args.nodes_per_gpu}/GPU), fan-out {args.fanout}, "
                                   f"{args.rate} API ops/tick evenly spaced, mix {MIX} of (user event, query, leave, crash+remove, crash+revive), "
                                   f"view_slots {args.view_slots}, rings {args.ring}, probe interval {args.probe_interval} ticks, push-pull interval "
                                   f"{args.push_pull_interval} ticks (x log2 scaling), reaper and queue checker on — BASELINE configs[2]",
                       "parallelism": f"node-id range shards x{world}, 1 all_to_all_single/tick" if world > 1 else "single GPU"},
            "rounds_to_99": ({"median": float(_np.median(rounds)), "max": int(max(rounds)), "min": int(min(rounds)), "n": len(rounds),
                              "what": "gossip rounds until >= 99 % of running nodes have applied a fresh user event, under the bench load"}
                             if rounds else None),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": traffic,
                         "kernel": "tick_kernel", "kernel_ms": kern_s * 1e3, "kernel_launches": int(prof_n),
                         "kernel_timing": f"HIP events around every {PROFILE_EVERY}th tick_kernel launch of the timed region, on its stream",
                         "stream_ms_per_step": ev_ms / args.steps, "b_tick_bytes": bt,
                         "achieved_revised": args.nodes_per_gpu * bt2 / kern_s / 1e9, "b_tick_layout_bytes": bt2,
                         "traffic_source": "profiles/r01_pmc_traffic.json (rocprofv3 FETCH_SIZE + WRITE_SIZE per launch)"},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.fanout, args.probe_interval, args.push_pull_interval, args.rate)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return out if rank == 0 else None


def main():
    run(parse_args())


if __name__ == "__main__":
    main()
