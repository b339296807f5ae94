#!/usr/bin/env python
"""bench.py — member-ticks/sec of the bulk SWIM/Serf gossip hot path on MI355X.

A "step" is one gossip tick of every simulated node.  N=1: BASELINE.json configs[2]
(1 Mi nodes, fan-out 4, HBM-roofline report).  N>1: one shard of 1 Mi nodes per GPU (weak
scaling), the round's RCCL all-to-all issued chunk-wise and overlapped with compute.  Prints ONE JSON line on rank 0.

The timed region is steady state by construction: run() first rolls the cluster forward
`--preroll` untimed ticks under the same constant load (rumours live ~20 ticks, suspicion timers
120+ ticks at this size, so a cold cluster is nearly idle), then does the `--warmup` and `--steps`
the contract asks for.  The figure therefore does not depend on --steps / --warmup.
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


# the same accounting for the frozen layout
def b_tick_layout(f):
    # rows r/w; sort keys r/w + the payload gathers of ONE packet (a node's f packets of a tick are the same packet);
    # that packet written once (48-byte cell + the map word) and fetched by its f receivers; one slot-map word and one
    # 16-byte head per received record
    return 2 * 64 + (2 * 16 * 4 + 4 * 16) + (48 + 4) + f * (48 + 4) + f * 4 * (4 + 16)


PMC_TRAFFIC = ("profiles/r02_pmc_traffic.json", "profiles/r01_pmc_traffic.json")  # newest first
CONV_RUMOURS, CONV_MAX_ROUNDS = 8, 60


def workload(args, n_total):
    """(config kwargs, operation schedule) of the benchmark — the same for the GPU run and the CPU baseline."""
    from serf_amd import workload as wl

    kw = dict(fanout=args.fanout, view_slots=args.view_slots, event_ring=args.ring, query_ring=args.ring,
              probe_interval=args.probe_interval, push_pull_interval=args.push_pull_interval,
              reap_interval=75, queue_check_interval=150,  # options.rs defaults: reap 15 s, queue check 30 s, timeouts 24 h
              recycle_interval=args.recycle_interval)
    horizon = args.preroll + args.warmup + args.steps + CONV_RUMOURS * (CONV_MAX_ROUNDS + 1)
    ops = wl.schedule(n_total, horizon, rate=args.rate, seed=3, mix=wl.BENCH_MIX,
                      max_member_subjects=args.view_slots // 2, even=True)
    return kw, ops


def cpu_baseline(args, seconds_budget=60.0):
    """The CPU oracle ("port") on the SAME configuration and schedule as the N=1 GPU run (rank 0 only): the same
    pre-roll, then a bounded number of timed ticks on all cores and a few on one thread.  When the host cannot hold
    the configuration (it needs ~64 KiB of address space per node, a fraction of it resident) it falls back to a
    smaller cluster and says so."""
    from serf_amd import _ffi

    lib = _ffi.SimLib(os.path.join(ROOT, "oracle", "liboracle.so"), prefix="osim_")  # test infrastructure: the checker, timed
    n = args.nodes_per_gpu
    note = ""
    while True:
        kw, ops = workload(args, n)
        try:
            sim = _ffi.Sim(lib, _ffi.make_config(n, **kw))
            break
        except _ffi.SimError:
            if n <= 1 << 16:
                raise
            n //= 4
            note = f" (host could not allocate {args.nodes_per_gpu} nodes: sample shrunk to {n})"
    for t, op, node, a, b in ops:
        sim.inject(t, op, node, a, b)
    cores = int(lib.dll.osim_t_threads())
    t0 = time.perf_counter()
    rolled = 0
    target = args.preroll + args.warmup
    while rolled < target and time.perf_counter() - t0 < seconds_budget:  # untimed, like the GPU run's pre-roll
        sim.step(5)
        rolled += 5
    t_roll = time.perf_counter() - t0
    ticks_all, ticks_one = 32, 8
    t1 = time.perf_counter()
    done = 0
    while done < ticks_all and time.perf_counter() - t1 < 20.0:
        sim.step(4)
        done += 4
    dt = time.perf_counter() - t1
    lib.dll.osim_t_set_threads(1)  # SURVEY.md §8d asks for both legs
    t2 = time.perf_counter()
    done1 = 0
    while done1 < ticks_one and time.perf_counter() - t2 < 15.0:
        sim.step(1)
        done1 += 1
    dt1 = time.perf_counter() - t2
    lib.dll.osim_t_set_threads(0)
    drops = sim.cluster_stats()["overflow"]
    sim.close()
    return {"value": n * done / dt, "unit": "member-ticks/s", "cores": cores, "kind": "port",
            "single_thread_value": n * done1 / dt1,
            "sample": f"same configuration and schedule as the GPU run{note}: {n} nodes, view_slots {args.view_slots}, rings {args.ring}, "
                      f"fan-out {args.fanout}; {rolled} untimed pre-roll ticks ({t_roll:.1f} s), then ticks {rolled}..{rolled + done - 1} timed on "
                      f"{cores} threads (OpenMP over nodes) and {done1} more on one thread; model_bound_drops {drops}"}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=60)
    ap.add_argument("--preroll", type=int, default=320,
                    help="untimed ticks under the same load before --warmup, so that the timed region is steady state")
    ap.add_argument("--nodes-per-gpu", type=int, default=1 << 20)
    ap.add_argument("--fanout", type=int, default=4)
    ap.add_argument("--view-slots", type=int, default=1024)
    ap.add_argument("--ring", type=int, default=512)
    ap.add_argument("--rate", type=float, default=0.25, help="API operations injected per tick (cluster-wide)")
    ap.add_argument("--probe-interval", type=int, default=5, help="memberlist probe interval in ticks (0 = SWIM layer off)")
    ap.add_argument("--push-pull-interval", type=int, default=150, help="memberlist push_pull_interval in ticks before log2(N) scaling (0 = off)")
    ap.add_argument("--recycle-interval", type=int, default=75, help="view-slot recycling pass every this many ticks (0 = never)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-convergence", action="store_true", help="skip the rounds-to-99 %% measurement (profiling runs)")
    ap.add_argument("--allow-drops", action="store_true", help="do not fail when the run hit a model bound (overflow > 0)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend of the N > 1 run (nccl = RCCL; gloo only for rehearsals)")
    ap.add_argument("--single-device", action="store_true",
                    help="rehearsal on a one-GPU box: every rank uses cuda:0 (with --backend gloo; RCCL refuses two ranks on one device)")
    ap.add_argument("--chunks", type=int, default=2,
                    help="N > 1: sender chunks per tick; the all-to-all of chunk c travels while chunk c + 1 computes (1 = one exchange after the kernel)")
    return ap.parse_args(argv)


def run(args, lib=None, dev=None, backend="nccl"):
    """The benchmark proper.  `lib`/`dev`/`backend` exist so that tests/test_bench_plumbing.py can drive the
    SAME control flow (sharded stepping, all-to-all, convergence, JSON) on CPU with gloo; main() always
    passes the HIP library, a CUDA device and RCCL."""
    import numpy as np
    import torch
    import torch.distributed as dist

    import serf_amd
    from serf_amd import _ffi
    from serf_amd.shard import ShardedSim

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    on_gpu = dev is None
    if on_gpu:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
        if getattr(args, "single_device", False):
            local_rank = 0
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
    kw, ops = workload(args, n_total)
    if world > 1:
        sim = ShardedSim(lib, n_total, dev, chunks=args.chunks, **kw)
    else:
        sim = _ffi.Sim(lib, _ffi.make_config(n_total, **kw))
        if on_gpu:
            sim.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    raw = sim.sim if world > 1 else sim
    step = sim.step
    for t, op, node, a, b in ops:
        sim.inject(t, op, node, a, b)

    def barrier():
        if world > 1:
            dist.barrier()
        if on_gpu:
            torch.cuda.synchronize()
        else:
            raw.sync()

    def allsum(vals):
        if world == 1:
            return [int(v) for v in vals]
        t = torch.tensor([int(v) for v in vals], dtype=torch.int64, device=dev)
        dist.all_reduce(t)
        return [int(x) for x in t]

    def load_now():  # cluster-wide load: records per packet in flight, queue entries per node, model-bound drops
        if world > 1:
            sim.sync()  # the round's exchanges have landed
        cs = raw.cluster_stats()
        inbox, queued, drops, up = allsum([cs["inbox_records"], sum(cs["queued"]), cs["overflow"], cs["up"]])
        # operations skipped for want of a view slot are counted on the (replicated) schedule, the same on every rank
        return {"records_per_packet": inbox / (args.fanout * n_total), "queued_per_node": queued / n_total,
                "drops": drops + int(cs["ops_dropped"]), "up": up, "slots_in_use": int(cs["slots_in_use"]),
                "slots_recycled": int(cs["slots_recycled"])}

    class _HostEvent:  # CPU stand-in for torch.cuda.Event in the plumbing test
        def __init__(self, enable_timing=True):
            self.t = 0.0

        def record(self):
            self.t = time.perf_counter()

        def elapsed_time(self, other):
            return (other.t - self.t) * 1e3

    Event = torch.cuda.Event if on_gpu else _HostEvent

    # ---- untimed: pre-roll to the stationary load, then the contract's warm-up ----
    trace = []
    done = 0
    while done < args.preroll:
        k = min(40, args.preroll - done)
        step(k)
        done += k
        trace.append(round(load_now()["records_per_packet"], 3))
    step(args.warmup)
    barrier()
    load0 = load_now()
    # an event pair costs ~10 us of stream time: time a sample of the launches on long runs, all of them on short ones
    profile_every = 4 if args.steps >= 100 else 1
    raw.profile(profile_every)
    barrier()
    # ---- timed: exactly K steps between two barriers.  The launches go to torch's current stream
    # (sim_set_stream above), so one pair of torch events brackets them as well.
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
    (prof_ms, prof_min, prof_max), prof_n = raw.profile_read_stats()
    raw.profile(False)
    load1 = load_now()
    # N > 1 diagnosis, outside the timed region: the same ticks with every all-to-all bracketed by events and run
    # synchronously (no overlap) — what one round's exchange costs on its own, next to the kernel
    exchange_ms, diag_ticks = None, 20
    if world > 1:
        barrier()
        sim.time_exchange(True)
        td0 = time.perf_counter()
        step(diag_ticks)
        barrier()
        serial_ms = (time.perf_counter() - td0) * 1e3 / diag_ticks
        exchange_ms = sim.time_exchange(False) / diag_ticks

    # ---- second half of the metric: rounds to 99 % convergence, measured after the timed region on
    # fresh user events, one at a time, under the same background load (every rank issues the same
    # calls; the originator's rank reads the Lamport time the event is going to get)
    rng = np.random.default_rng(99)
    rounds = []
    for i in range(0 if args.no_convergence else CONV_RUMOURS):
        node, key = int(rng.integers(0, n_total)), 0x7F000000 + i
        owner = node // args.nodes_per_gpu
        lt = raw.stats(node).event_time if owner == rank else 0
        if world > 1:
            t = torch.tensor([lt], dtype=torch.int64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            lt = int(t[0])
        sim.user_event(node, key, 64)
        got = None
        for r in range(1, CONV_MAX_ROUNDS + 1):
            step(1)
            seen, up = sim.convergence(_ffi.K_EVENT, key, lt)
            if seen * 100 >= up * 99:
                got = r
                break
        rounds.append(got if got is not None else CONV_MAX_ROUNDS + 1)
    load2 = load_now()
    out = None
    if rank == 0:
        value = n_total * args.steps / dt
        bt, bt2 = b_tick_v0(args.fanout), b_tick_layout(args.fanout)
        # dominant kernel = tick_kernel: one launch per tick; HIP events around the launches of the timed
        # region (sim_profile); ev_ms (everything on the stream, ops + push-pull included) for reference
        kern_s = prof_ms / 1e3 / max(1, prof_n) if prof_ms > 0 else ev_ms / 1e3 / args.steps  # (the oracle behind a CPU test has no kernel)
        achieved = args.nodes_per_gpu * bt / kern_s / 1e9
        traffic, traffic_src, traffic_cal = None, None, None
        for rel in PMC_TRAFFIC:
            if os.path.exists(os.path.join(ROOT, rel)):
                try:
                    doc = json.load(open(os.path.join(ROOT, rel)))
                    traffic, traffic_src = doc.get("hbm_bytes_per_launch"), rel
                    traffic_cal = doc.get("calibration")
                    break
                except Exception:
                    pass
        out = {
            "metric": "member-ticks/sec", "value": value, "unit": "member-ticks/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": f"{n_total} nodes ({args.nodes_per_gpu}/GPU), fan-out {args.fanout}, "
                                   f"{args.rate} API ops/tick evenly spaced, mix (0.55, 0.2, 0.15, 0.05, 0.05) of (user event, query, leave, crash+remove, crash+revive), "
                                   f"view_slots {args.view_slots}, rings {args.ring}, probe interval {args.probe_interval} ticks, push-pull interval "
                                   f"{args.push_pull_interval} ticks (x log2 scaling), reaper and queue checker on — BASELINE configs[2]; "
                                   f"{args.preroll} untimed pre-roll ticks under the same load before the warm-up (steady state)",
                       "parallelism": (f"node-id range shards x{world}, {args.chunks} chunk-wise all_to_all_single per tick, overlapped with compute"
                                       if world > 1 else "single GPU"),
                       "departure_from_survey_8d": "SURVEY.md §8d config 3 asks for 1 024 active rumours: a packet carries SIM_P = 4 records "
                                                   "(the survey's own P), so a node rebroadcasts f*P/limit = 16/28 = 0.57 records per tick and the "
                                                   f"cluster sustains about that many new rumours per tick; the bench injects {args.rate} operations "
                                                   "(~0.41 rumours) per tick = 72 % of it, ~10-15 rumours live at any time, zero model-bound drops "
                                                   "(DESIGN.md §7)",
                       "preroll": args.preroll,
                       "timed_ticks": [args.preroll + args.warmup, args.preroll + args.warmup + args.steps - 1],
                       "model_bound_drops": load2["drops"],
                       "load": {"records_per_packet_start": round(load0["records_per_packet"], 3),
                                "records_per_packet_end": round(load1["records_per_packet"], 3),
                                "queued_per_node_start": round(load0["queued_per_node"], 3),
                                "queued_per_node_end": round(load1["queued_per_node"], 3),
                                "records_per_packet_preroll_every_40_ticks": trace,
                                "nodes_up": load1["up"], "view_slots_in_use": load2["slots_in_use"],
                                "view_slots_recycled": load2["slots_recycled"]}},
            "rounds_to_99": ({"median": float(np.median(rounds)), "max": int(max(rounds)), "min": int(min(rounds)), "n": len(rounds),
                              "what": "gossip rounds until >= 99 % of running nodes have applied a fresh user event, under the bench load",
                              "fanout_model": "per-tick bijection (every node receives exactly `fanout` packets per round); memberlist's literal "
                                              "kRandomNodes (Poisson-like in-degree) needs one round more: 10 vs 9 at 64 Ki nodes, 12 vs 11 at "
                                              "1 Mi (CPU oracle, 1 000 rumours each, profiles/r02_fanout_model_*.json)"}
                             if rounds else None),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": traffic,
                         "kernel": "tick_kernel", "kernel_ms": kern_s * 1e3,
                         "kernel_ms_min": prof_min, "kernel_ms_max": prof_max, "kernel_launches": int(prof_n),
                         "kernel_timing": f"HIP event pair on every {profile_every}{'th' if profile_every > 1 else 'st'} tick_kernel dispatch of the timed region (hipExtLaunchKernelGGL start/stop events, on the launch stream)",
                         "stream_ms_per_step": ev_ms / args.steps, "b_tick_bytes": bt,
                         "achieved_revised": args.nodes_per_gpu * bt2 / kern_s / 1e9, "b_tick_layout_bytes": bt2,
                         "traffic_over_algorithmic": (traffic / (args.nodes_per_gpu * bt)) if traffic else None,
                         "traffic_source": traffic_src, "traffic_calibration": traffic_cal},
        }
        if world > 1:
            xb = raw.exchange_bytes()
            out["exchange"] = {"chunks": sim.chunks, "exchange_ms": exchange_ms, "kernel_ms": kern_s * 1e3,
                               "serial_ms_per_step": serial_ms, "overlapped_ms_per_step": dt / args.steps * 1e3,
                               "bytes_per_peer": xb // world, "bytes_per_gpu_per_tick": xb,
                               "bytes_leaving_gpu_per_tick": xb // world * (world - 1),
                               "what": f"the timed region runs each tick as {sim.chunks} chunk launches with the all-to-all of chunk c in flight "
                                       "while chunk c + 1 computes (overlapped_ms_per_step = ms_per_step); exchange_ms and serial_ms_per_step come "
                                       f"from {diag_ticks} further ticks with the collectives run one after the other between events (rank 0): "
                                       "exchange_ms = all-to-alls of one round, serial_ms_per_step = that round without overlap"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    if load2["drops"] and not args.allow_drops:
        # a run that hit a model bound is not a run of the protocol the parity tests cover: refuse it
        raise SystemExit(f"bench.py: model bound hit ({load2['drops']} drops: queue slots / bucket keys / timers) — result invalid")
    return out


def main():
    args = parse_args()
    run(args, backend=args.backend)


if __name__ == "__main__":
    main()
