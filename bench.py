#!/usr/bin/env python
"""bench.py — member-ticks/sec of the bulk SWIM/Serf gossip hot path on MI355X.

A "step" is one gossip tick of every simulated node.  N=1: BASELINE.json configs[2]
(1 Mi nodes, fan-out 4, HBM-roofline report), measured on BOTH fan-out models in one run: the HEADLINE (`value`,
`ms_per_step`, `roofline`) is memberlist's literal kRandomNodes peer selection — the reference's —,
the per-tick bijection (every node receives exactly `fanout` packets) is reported next to it under `fanout_models`.
N>1: one shard of 1 Mi nodes per GPU (weak scaling), the SAME two models, kRandomNodes the headline at every N: its packets
are packed per destination shard behind the tick's launch and travel as one equal-split RCCL all-to-all per round (the
bijection's all-to-all is issued chunk-wise and overlapped with compute).  Prints ONE JSON line on rank 0.

`python bench.py --gpus N` without a launcher (WORLD_SIZE unset) starts its N ranks itself (one process per GPU,
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set for them, rendezvous on 127.0.0.1) and supervises them: a rank that dies
or a run that stops making progress is torn down, retried once with `--chunks 1`, and reported as a JSON line with an
"error" field instead of a hang.  Under `python -m torch.distributed.run … bench.py --gpus N` it is a plain rank.

The timed region is steady state by construction: run() first rolls the cluster forward `--preroll` untimed ticks
under the same constant load (rumours live ~20 ticks, suspicion timers 120+ ticks at this size, so a cold cluster is
nearly idle), then does the `--warmup` and `--steps` the contract asks for.  The operation schedule has a fixed horizon
and the convergence window starts at a fixed tick, so neither member-ticks/s nor rounds-to-99 % depend on
--steps / --warmup.
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


# SURVEY.md §8d algorithmic bytes per member-tick (v0 layout): 2R + 2QE + 2fPE + 4(f+2)
def b_tick_v0(f, p=4):
    return 2 * 64 + 2 * 16 * 16 + 2 * f * p * 16 + 4 * (f + 2)   # (Q stays the survey's 16: the bytes a deeper queue moves are not priced)


# the same accounting for the frozen layout
def b_tick_layout(f):
    # rows r/w; sort keys r/w + the payload gathers of ONE packet (a node's f packets of a tick are the same packet);
    # that packet written once (48-byte cell + the map word) and fetched by its f receivers; one slot-map word and one
    # 16-byte head per received record
    return 2 * 64 + (2 * 16 * 4 + 4 * 16) + (48 + 4) + f * (48 + 4) + f * 4 * (4 + 16)


PMC_TRAFFIC = {"krandomnodes": ("profiles/r06_pmc_traffic_krandomnodes.json", "profiles/r05_pmc_traffic_krandomnodes.json", "profiles/r04_pmc_traffic_krandomnodes.json"),  # newest first
               "bijection": ("profiles/r06_pmc_traffic_bijection.json", "profiles/r05_pmc_traffic_bijection.json", "profiles/r04_pmc_traffic_bijection.json", "profiles/r03_pmc_traffic.json", "profiles/r02_pmc_traffic.json")}
# ... and over the launches of the LONG window (its own PMC passes: the bytes a launch moves follow the load of the ticks it covers)
PMC_TRAFFIC_LONG = {"krandomnodes": ("profiles/r06_pmc_traffic_krandomnodes_long.json", "profiles/r05_pmc_traffic_krandomnodes_long.json"),
                    "bijection": ("profiles/r06_pmc_traffic_bijection_long.json", "profiles/r05_pmc_traffic_bijection_long.json")}
PMC_TRAFFIC_SECOND = ("profiles/r06_pmc_traffic_second_load.json",)
LONG_WINDOW = 300  # ticks of the second timed window (with --steps < 300): long enough to hold a push-pull batch and recycling passes
# the DEVICE side of the tick kernel (state + queue, handlers + classification, the kernel itself): what a PMC profile is valid for
KERNEL_SOURCES = [os.path.join("serf_amd", "csrc", f) for f in ("serf_sim_state.inc", "serf_sim_handlers.inc", "serf_sim_tick.inc")]
# rounds-to-99 %: every user event the workload issues in CONV_WINDOW ticks starting CONV_OFFSET ticks after the
# pre-roll (at most CONV_RUMOURS of them), each followed for at most CONV_MAX_ROUNDS rounds
CONV_RUMOURS, CONV_MAX_ROUNDS, CONV_OFFSET, CONV_WINDOW = 64, 60, 400, 480


def long_window(args):
    return LONG_WINDOW if args.steps < LONG_WINDOW and not getattr(args, "no_long_window", False) else 0


def conv_start_tick(args):
    return args.preroll + max(CONV_OFFSET, args.warmup + args.steps + long_window(args))


def horizon(args):
    h = conv_start_tick(args) + CONV_WINDOW + CONV_MAX_ROUNDS + 1
    return (h + 39) // 40 * 40


def workload(args, n_total, model=None):
    """(config kwargs, operation schedule) of the benchmark — the same for the GPU run and the CPU baseline."""
    if model is None:  # (the tools: one model, by flag)
        model = "krandomnodes" if getattr(args, "fanout_model", "both") == "krandomnodes" else "bijection"
    from serf_amd import workload as wl

    kw = dict(fanout=args.fanout, view_slots=args.view_slots, event_ring=args.ring, query_ring=args.ring,
              probe_interval=args.probe_interval, push_pull_interval=args.push_pull_interval,
              reap_interval=75, queue_check_interval=150,  # options.rs defaults: reap 15 s, queue check 30 s, timeouts 24 h
              recycle_interval=args.recycle_interval)
    if getattr(args, "pkt_records", 4) != 4:
        kw["pkt_records"] = args.pkt_records
    if getattr(args, "ring_overflow", 0):
        kw["ring_overflow"] = args.ring_overflow
    if model == "krandomnodes":  # memberlist's literal kRandomNodes instead of the per-tick bijection (one GPU)
        from serf_amd import _ffi
        kw["flags"] = _ffi.CF_BASELINE_JOINED | _ffi.CF_RANDOM_FANOUT
    ops = wl.schedule(n_total, horizon(args), rate=args.rate, seed=3, mix=wl.BENCH_MIX,
                      max_member_subjects=args.view_slots // 2, even=True)
    return kw, ops


def kernel_source_sha16():
    try:
        return hashlib.sha256(b"".join(open(os.path.join(ROOT, f), "rb").read() for f in KERNEL_SOURCES)).hexdigest()[:16]
    except OSError:
        return None


def measured_traffic(model, long=False):
    """HBM bytes per tick-kernel launch from the newest committed PMC profile of that fan-out model — valid only for the
    kernel source it was measured on (the profile records the source's hash; a file without one is treated as stale)."""
    sha = kernel_source_sha16()
    for rel in (PMC_TRAFFIC_LONG if long else PMC_TRAFFIC)[model]:
        path = os.path.join(ROOT, rel)
        if not os.path.exists(path):
            continue
        try:
            doc = json.load(open(path))
        except Exception:
            continue
        prov = {"from_file": rel, "measured_at_commit": doc.get("commit"), "kernel_source_sha16": doc.get("kernel_source_sha16"),
                "kernel_source_sha16_now": sha, "workload": doc.get("workload"), "calibration": doc.get("calibration")}
        prov["matches_this_kernel"] = bool(sha and doc.get("kernel_source_sha16") == sha)
        prov["timed_ticks_of_the_profile"] = {"steps": doc.get("steps"), "warmup": doc.get("warmup")}
        return doc.get("hbm_bytes_per_launch"), prov
    return None, None


def cpu_baseline(args, model, parity_ticks, timed=True, seconds_budget=300.0):
    """The CPU oracle ("port") on the SAME configuration, fan-out model and schedule as the N=1 GPU run (rank 0 only): rolled
    through the same ticks; its state digest is taken at every tick of `parity_ticks` (the first timed tick of the GPU run
    and the tick right after its last timed one) for the parity block.  With `timed`, the ticks of the timed region — at most
    32 of them — are timed on all cores, and 8 more on one thread behind the last parity tick.  When the host cannot hold the
    configuration (it needs ~64 KiB of address space per node, a fraction of it resident) it falls back to a smaller cluster,
    says so, and the parity block is void."""
    from serf_amd import _ffi

    lib = _ffi.SimLib(os.path.join(ROOT, "oracle", "liboracle.so"), prefix="osim_")  # test infrastructure: the checker, timed
    n = args.nodes_per_gpu
    note = ""
    while True:
        kw, ops = workload(args, n, model)
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
    first, last = parity_ticks[0], parity_ticks[-1]

    def roll(to):
        nonlocal rolled
        while rolled < to:  # untimed, like the GPU run's pre-roll + warm-up — never shortened
            k = min(5, to - rolled)
            sim.step(k)
            rolled += k
            if time.perf_counter() - t0 > seconds_budget:
                raise SystemExit(f"bench.py: the CPU oracle needed more than {seconds_budget:.0f} s for {rolled} of {last} ticks "
                                 "on this host — cpu_baseline would not be timing the GPU's ticks; rerun with --no-cpu-baseline or a smaller --preroll")

    roll(first)
    t_roll = time.perf_counter() - t0
    digests = {first: sim.digest() if n == args.nodes_per_gpu else None}
    done, dt = 0, 0.0
    if timed:  # the first ticks of the GPU's timed region, on all cores
        ticks_all = min(32, last - first)
        t1 = time.perf_counter()
        while done < ticks_all and time.perf_counter() - t1 < 20.0:
            k = min(4, ticks_all - done)
            sim.step(k)
            done += k
        dt = time.perf_counter() - t1
        rolled += done
    for t in parity_ticks[1:]:
        roll(t)
        digests[t] = sim.digest() if n == args.nodes_per_gpu else None
    res = None
    if timed:
        lib.dll.osim_t_set_threads(1)  # SURVEY.md §8d asks for both legs
        t2 = time.perf_counter()
        done1 = 0
        while done1 < 8 and time.perf_counter() - t2 < 15.0:
            sim.step(1)
            done1 += 1
        dt1 = time.perf_counter() - t2
        lib.dll.osim_t_set_threads(0)
        drops = sim.cluster_stats()["overflow"]
        curve = {}
        host_cores = os.cpu_count()
        if getattr(args, "cpu_thread_curve", True):   # SURVEY.md §8d (ii) "all cores": what more threads than the cgroup quota buy (nothing)
            for nt in (2 * cores, 4 * cores, host_cores):
                if nt and nt > cores and nt <= (host_cores or 0) and nt not in curve:
                    lib.dll.osim_t_set_threads(nt)
                    t3 = time.perf_counter()
                    sim.step(2)
                    curve[nt] = n * 2 / (time.perf_counter() - t3)
            lib.dll.osim_t_set_threads(0)
        best = max([n * done / dt] + list(curve.values()))
        res = {"value": best, "unit": "member-ticks/s", "cores": cores if best == n * done / dt else max(curve, key=curve.get), "kind": "port",
               "host_cores": host_cores, "cpu_quota_threads": cores, "value_quota_threads": n * done / dt,
               "threads_curve": {str(k): v for k, v in sorted(curve.items())},
               "single_thread_value": n * done1 / dt1, "fanout_model": model,
               "sample_short": f"{n} nodes, same config+schedule; ticks {first}..{first + done - 1} on {cores} threads (cgroup quota; host shows {host_cores} CPUs: curve in detail), {done1} ticks on 1",
               "sample": f"same configuration, fan-out model ({model}) and schedule as the GPU run{note}: {n} nodes, view_slots {args.view_slots}, "
                         f"rings {args.ring}, fan-out {args.fanout}; {first} untimed pre-roll ticks ({t_roll:.1f} s), then ticks {first}..{first + done - 1} "
                         f"timed on {cores} threads (OpenMP over nodes), on to tick {last} for the second parity digest, and {done1} more on one "
                         f"thread; model_bound_drops {drops}"}
    sim.close()
    return res, digests


def second_load(args, lib, dev, torch):
    """N = 1 only, after the headline run: SURVEY.md §8d config 3 read the way the reference's packets allow — the same cluster under as
    many concurrently active rumours as the 1 400-byte packet budget carries (r6: the queue holds 64 entries, a full ring bucket
    continues in overflow rows — the old bounds of 16 / 6 stopped this load at 0.35 operations per tick).  The benchmark mix's user
    events average 264 bytes: a rumour has to be sent `retransmit_mult * ceil(log10(N + 1))` = 28 times by every node, a node sends
    4 packets x 87 sixteen-byte units per tick, so ~1 operation per tick saturates the REFERENCE's own packets at 1 Mi nodes —
    --second-rate (default 0.8) stays just below.  Packets of 16 records (4 pages): only the byte budget binds.  Measured like the
    headline (pre-roll, warm-up, HIP events on the launches); `live_rumours` = distinct (kind, subject / key) pairs in the queues
    at the end of the timed window (one dump of the queue array), `live_records` = distinct records (a suspicion counts once per
    accuser).  With --second-parity the CPU oracle is rolled through the same ticks and the digests are compared (minutes)."""
    import copy

    import numpy as np

    from serf_amd import _ffi

    a2 = copy.copy(args)
    a2.pkt_records, a2.rate, a2.preroll, a2.warmup, a2.steps = 16, args.second_rate, args.second_preroll, 20, args.second_steps
    a2.ring_overflow = args.second_ring_overflow
    model = "bijection" if args.fanout_model == "bijection" else "krandomnodes"   # the headline model
    kw, ops = workload(a2, args.nodes_per_gpu, model)
    sim = _ffi.Sim(lib, _ffi.make_config(args.nodes_per_gpu, **kw))
    sim.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    for o in ops:
        sim.inject(*o)
    sim.step(a2.preroll + a2.warmup)
    sim.sync()
    d0 = sim.digest() if args.second_parity else None
    sim.profile(1)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record()
    sim.step(a2.steps)
    ev1.record()
    sim.sync()
    dt = time.perf_counter() - t0
    (ms, mn, mx), cnt = sim.profile_read_stats()
    sim.profile(0)
    d1 = sim.digest() if args.second_parity else None
    cs = sim.cluster_stats()
    kern_s = ms / 1e3 / max(1, cnt)
    # the load, from the queues themselves
    q = sim.dump(_ffi.ARR_QUEUE)
    live = q[q["meta"] != 0xFFFFFFFF]
    kinds = ((live["meta"] >> 4) & 15).astype(np.uint64)
    live_rumours = int(len(np.unique(live["key"].astype(np.uint64) | (kinds << 32))))
    live_records = int(len(np.unique(np.stack([live["key"].astype(np.uint64) | (kinds << 32), live["val"]], 1), axis=0)))
    del q, live
    f, P = args.fanout, 16
    bt = b_tick_v0(f, P)
    alg = args.nodes_per_gpu * bt / kern_s / 1e9
    tr = None
    for rel in PMC_TRAFFIC_SECOND:
        path = os.path.join(ROOT, rel)
        if os.path.exists(path):
            try:
                doc = json.load(open(path))
                if doc.get("kernel_source_sha16") == kernel_source_sha16() and doc.get("rate") == a2.rate and doc.get("steps") == a2.steps:
                    tr = doc.get("hbm_bytes_per_launch")
            except Exception:
                pass
            break
    out = {"pkt_records": 16, "fanout_model": model, "rate": a2.rate, "steps": a2.steps, "preroll": a2.preroll, "ring_overflow": a2.ring_overflow,
           "value": args.nodes_per_gpu * a2.steps / dt, "unit": "member-ticks/s",
           "ms_per_step": dt / a2.steps * 1e3, "stream_ms_per_step": ev0.elapsed_time(ev1) / a2.steps,
           "kernel_ms": kern_s * 1e3, "kernel_ms_min": mn, "kernel_ms_max": mx,
           "b_tick_bytes": bt, "achieved": alg, "frac": alg / 8000.0, "traffic": tr, "frac_measured": (tr / kern_s / 1e9 / 8000.0) if tr else None,
           "model_bound_drops": int(cs["overflow"]) + int(cs["ops_dropped"]),
           "live_rumours": live_rumours, "live_records": live_records,
           "records_per_packet": round(cs["inbox_records"] / (args.fanout * args.nodes_per_gpu), 3),
           "queued_per_node": round(sum(cs["queued"]) / args.nodes_per_gpu, 3), "deepest_queue": int(cs["max_queue"]),
           "what": f"SURVEY §8d config 3 at the packet budget: same cluster and mix, {a2.rate} API ops/tick (~1 saturates the reference's 1 400-byte "
                   "packets at this size and mix), 16 records per packet, queue of 64 (16 hot + deep_queue_kernel), ring buckets with "
                   f"{a2.ring_overflow} overflow rows; frac = B_tick(f, P = 16) x nodes / tick-kernel time (the deep kernel's time is in ms_per_step)"}
    if args.second_parity:
        progress_ticks = [a2.preroll + a2.warmup, a2.preroll + a2.warmup + a2.steps]
        _, dig = cpu_baseline(a2, model, progress_ticks, timed=False, seconds_budget=3600.0)
        ok = [tuple(dig[t]) == tuple(d) for t, d in zip(progress_ticks, (d0, d1))]
        out["digest_match"] = all(ok)
        out["parity"] = {"ticks": progress_ticks, "digest_match_per_tick": ok, "gpu": [[f"{x:016x}" for x in d] for d in (d0, d1)],
                         "oracle": [[f"{x:016x}" for x in dig[t]] for t in progress_ticks]}
    sim.close()
    return out


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
    ap.add_argument("--pkt-records", type=int, default=4, help="records a gossip packet can carry (4, 8, 12 or 16: pages of 4)")
    ap.add_argument("--probe-interval", type=int, default=5, help="memberlist probe interval in ticks (0 = SWIM layer off)")
    ap.add_argument("--push-pull-interval", type=int, default=150, help="memberlist push_pull_interval in ticks before log2(N) scaling (0 = off)")
    ap.add_argument("--recycle-interval", type=int, default=75, help="view-slot recycling pass every this many ticks (0 = never)")
    ap.add_argument("--second-rate", type=float, default=0.8, help="API operations per tick of the second measured load (16 records per packet; ~1 saturates the 1 400-byte packets at 1 Mi nodes)")
    ap.add_argument("--second-preroll", type=int, default=160, help="untimed ticks of the second load before its 20 warm-up ticks")
    ap.add_argument("--second-steps", type=int, default=60, help="timed ticks of the second load")
    ap.add_argument("--second-ring-overflow", type=int, default=8, help="overflow rows per de-dup ring of the second load's cluster")
    ap.add_argument("--second-parity", action="store_true", help="roll the CPU oracle through the second load's ticks as well and compare digests (minutes of CPU)")
    ap.add_argument("--ring-overflow", type=int, default=0, help="overflow rows per de-dup ring and node of the headline cluster (sim_config.ring_overflow)")
    ap.add_argument("--fanout-model", choices=["both", "krandomnodes", "bijection"], default="both",
                    help="N = 1: which fan-out model(s) to measure.  both (default): memberlist's literal kRandomNodes — the reference's peer "
                         "selection, the HEADLINE — and the per-tick bijection next to it (fanout_models)")
    ap.add_argument("--random-fanout", action="store_true", help="shorthand for --fanout-model krandomnodes")
    ap.add_argument("--force-sharded", action="store_true",
                    help="N = 1: run the cluster as ONE shard through the sharded path (ShardedSim, exchange buffers, the round's all-to-all "
                         "over the collective library) — a one-rank rehearsal of the N > 1 line")
    ap.add_argument("--exchange", choices=["auto", "torch", "rccl"], default="auto",
                    help="sharded runs: who issues the round's all-to-all — the library itself over RCCL (sim_exchange_*), or torch.distributed")
    ap.add_argument("--no-second-load", action="store_true", help="skip the second measured load (N = 1: 16 records per packet at --second-rate)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--parity-all", action="store_true", help="roll the CPU oracle for the second fan-out model's parity digests too (default: the headline model only)")
    ap.add_argument("--no-long-window", action="store_true", help="skip the second timed window (profiling runs: the LAST launches are then the timed ones)")
    ap.add_argument("--no-convergence", action="store_true", help="skip the rounds-to-99 %% measurement (profiling runs)")
    ap.add_argument("--allow-drops", action="store_true", help="do not fail when the run hit a model bound (overflow > 0)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend of the N > 1 run (nccl = RCCL; gloo only for rehearsals)")
    ap.add_argument("--single-device", action="store_true",
                    help="rehearsal on a one-GPU box: every rank uses cuda:0 (with --backend gloo; RCCL refuses two ranks on one device)")
    ap.add_argument("--chunks", type=int, default=2,
                    help="N > 1: sender chunks per tick; the all-to-all of chunk c travels while chunk c + 1 computes (1 = one exchange after the kernel)")
    ap.add_argument("--watchdog", type=float, default=120.0,
                    help="N > 1: seconds without progress after which a rank gives up (a JSON line with \"error\" on rank 0, exit code 3)")
    ap.add_argument("--launch-timeout", type=float, default=1500.0, help="self-launched N > 1 run: seconds the supervisor waits for its ranks")
    a = ap.parse_args(argv)
    if a.random_fanout:
        a.fanout_model = "krandomnodes"
    if a.force_sharded and a.fanout_model == "both":
        a.fanout_model = "bijection"   # (the one-rank rehearsal: one model per run; --fanout-model krandomnodes: the packed slabs)
    return a


_OUT = None  # the process's REAL stdout once run() has claimed it (see claim_stdout)


def claim_stdout():
    """The contract is ONE JSON line on stdout.  Libraries below us write there too (RCCL prints a version banner through C
    stdio when a communicator is made, flushed at exit): keep a private handle on the real stdout for the JSON line and point
    file descriptor 1 at stderr for everybody else."""
    global _OUT
    if _OUT is None:
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        _OUT = os.fdopen(saved, "w")
    return _OUT


def emit(obj):
    print(json.dumps(obj), file=_OUT if _OUT is not None else sys.stdout, flush=True)


LINE_LIMIT = 4000      # bytes of the ONE JSON line (the driver keeps a tail of ~8 KB of stdout: a longer line loses its head)
DETAIL_FILE = "bench_detail.json"


def _r(x, nd=4):
    """a number rounded to `nd` significant digits (the line is a summary; bench_detail.json keeps full precision)"""
    if isinstance(x, float):
        return float(f"{x:.{nd}g}")
    return x


def compact(out):
    """The ONE line: the contract's keys + roofline + cpu_baseline + what a reader needs to judge the run, every text short.
    Everything else (prose, per-model blocks, digests, provenance, load traces) is `out` itself, written to DETAIL_FILE."""
    def roof(r):
        if not r:
            return None
        d = {k: _r(r.get(k)) for k in ("bound", "achieved", "peak", "unit", "frac", "frac_measured", "traffic", "kernel", "kernel_ms")}
        return {k: v for k, v in d.items() if v is not None or k in ("traffic", "frac_measured")}

    def rounds(r):
        return None if not r else {"median": r["median"], "p90": r["p90"], "max": r["max"], "n": r["n"]}

    cfg = out["config"]
    line = {k: _r(out[k], 6) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                       "vs_baseline", "dtype", "data") if k in out}
    line["config"] = {"workload": cfg["workload_short"], "fanout_model": cfg["fanout_model"], "parallelism": cfg["parallelism_short"],
                      "timed_ticks": cfg["timed_ticks"], "model_bound_drops": cfg["model_bound_drops"]}
    line["roofline"] = roof(out.get("roofline"))
    if out.get("cpu_baseline"):
        cb = out["cpu_baseline"]
        line["cpu_baseline"] = {k: _r(cb.get(k)) for k in ("value", "unit", "cores", "host_cores", "kind", "single_thread_value", "value_16_threads") if cb.get(k) is not None}
        line["cpu_baseline"]["sample"] = cb["sample_short"]
    if out.get("long_window"):
        lw = out["long_window"]
        line["value_long_window"] = _r(out["value_long_window"], 6)
        line["long_window"] = {"steps": lw["steps"], "ms_per_step": _r(lw["ms_per_step"]), "roofline": roof(lw["roofline"])}
    if out.get("parity"):
        line["parity"] = {"digest_match": out["parity"].get("digest_match"), "ticks": out["parity"].get("ticks"),
                          "models": [k for k, v in out["parity"].items() if isinstance(v, dict)]}
    line["rounds_to_99"] = rounds(out.get("rounds_to_99"))
    fm = {}
    for mo, d in out.get("fanout_models", {}).items():
        if mo == cfg["fanout_model"]:
            continue
        fm[mo] = {"value": _r(d["value"], 6), "ms_per_step": _r(d["ms_per_step"]), "kernel_ms": _r(d["kernel_ms"]),
                  "frac": _r(d["roofline"]["frac"]), "frac_measured": _r(d["roofline"]["frac_measured"]), "rounds_to_99": rounds(d["rounds_to_99"])}
        if "exchange" in d:
            fm[mo]["exchange"] = {"chunks": d["exchange"]["chunks"], "exchange_ms": _r(d["exchange"]["exchange_ms"])}
    if fm:
        line["fanout_models"] = fm
    if out.get("second_load"):
        sl = out["second_load"]
        line["second_load"] = {k: _r(sl.get(k)) for k in ("rate", "pkt_records", "live_rumours", "value", "ms_per_step", "kernel_ms", "frac", "frac_measured",
                                                          "model_bound_drops", "records_per_packet", "deepest_queue", "digest_match") if sl.get(k) is not None}
    if out.get("exchange"):
        x = out["exchange"]
        line["exchange"] = {k: _r(x.get(k)) for k in ("chunks", "exchange_ms", "kernel_ms", "serial_ms_per_step", "overlapped_ms_per_step",
                                                      "host_ms_per_step", "bytes_leaving_gpu_per_tick")}
    if out.get("distributed"):
        line["distributed"] = {k: out["distributed"].get(k) for k in ("backend", "world_size")}
    line["detail_file"] = DETAIL_FILE
    s = json.dumps(line)
    if len(s) > LINE_LIMIT:   # never again a line the driver cannot parse: shed the optional blocks, largest first
        for k in ("fanout_models", "second_load", "exchange", "long_window", "distributed"):
            line.pop(k, None)
            if len(json.dumps(line)) <= LINE_LIMIT:
                break
    return line


def emit_result(out):
    """full result -> DETAIL_FILE (next to bench.py; also under gpurun_out/ when that exists) and stderr; the compact line -> stdout"""
    txt = json.dumps(out, indent=1)
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, DETAIL_FILE), "w") as f:
                    f.write(txt)
            except OSError:
                pass
    sys.stderr.write("bench.py detail: " + json.dumps(out) + "\n")
    emit(compact(out))


class Progress:
    """Heartbeat of a rank: every phase of run() touches it; a watchdog thread ends the process when it goes stale
    (a collective that never completes would otherwise hang the whole job without a word)."""

    def __init__(self, limit, rank, meta):
        self.t, self.what, self.limit, self.rank, self.meta, self.done = time.monotonic(), "start", limit, rank, meta, False
        if limit > 0:
            threading.Thread(target=self._watch, daemon=True).start()

    def __call__(self, what):
        self.t, self.what = time.monotonic(), what

    def _watch(self):
        while not self.done:
            time.sleep(1.0)
            if not self.done and time.monotonic() - self.t > self.limit:
                if self.rank == 0:
                    emit(dict(self.meta, error=f"no progress for {self.limit:.0f} s in phase '{self.what}' (collective hung?)", value=None))
                sys.stderr.write(f"bench.py rank {self.rank}: watchdog: stuck in '{self.what}'\n")
                os._exit(3)


MODEL_WHAT = {
    "krandomnodes": "memberlist's literal kRandomNodes (uniform targets over the other nodes, no replacement: Poisson-like in-degree; the "
                    "tick's fan-out graph an explicit CSR, built per tick by the library's own bucket sort) — the REFERENCE's peer selection",
    "bijection": "per-tick pseudo-random bijection (every node sends `fanout` packets AND receives exactly `fanout`): an implicit, "
                 "coalescing-friendly fan-out graph — not the reference's peer selection",
}


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
        claim_stdout()
    progress = Progress(args.watchdog if world > 1 else 0.0, rank,
                        {"metric": "member-ticks/sec", "unit": "member-ticks/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup})
    if on_gpu:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
        if getattr(args, "single_device", False):
            local_rank = 0
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    if world > 1 or args.force_sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:  # a world of one rank: the N > 1 path (ShardedSim, exchange buffers, the collective library) on one GPU
            os.environ.setdefault("MASTER_PORT", "29577")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        progress("init_process_group")
        if on_gpu:
            dist.init_process_group(backend, device_id=dev)
        else:
            dist.init_process_group(backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    n_total = args.nodes_per_gpu * world
    if lib is None:
        lib = serf_amd.load()
    if args.fanout_model == "both":   # at every N: the reference's model first — the headline, ONE series from 1 to 8 GPUs —, the bijection next to it
        models = ["krandomnodes", "bijection"]
    else:
        models = [args.fanout_model]
    sharded = world > 1 or args.force_sharded

    class _HostEvent:  # CPU stand-in for torch.cuda.Event in the plumbing test
        def __init__(self, enable_timing=True):
            self.t = 0.0

        def record(self):
            self.t = time.perf_counter()

        def elapsed_time(self, other):
            return (other.t - self.t) * 1e3

    Event = torch.cuda.Event if on_gpu else _HostEvent

    def allsum(vals):
        if world == 1:
            return [int(v) for v in vals]
        t = torch.tensor([int(v) for v in vals], dtype=torch.int64, device=dev)
        dist.all_reduce(t)
        return [int(x) for x in t]

    def allmax(v):
        if world == 1:
            return int(v)
        t = torch.tensor([int(v)], dtype=torch.int64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return int(t[0])

    def measure(model):
        """One cluster on one fan-out model: pre-roll, warm-up, the timed K steps, the long window, rounds-to-99 %."""
        kw, ops = workload(args, n_total, model)
        progress(f"create ({model})")
        if sharded:
            sim = ShardedSim(lib, n_total, dev, chunks=args.chunks, exchange=args.exchange, **kw)
        else:
            sim = _ffi.Sim(lib, _ffi.make_config(n_total, **kw))
            if on_gpu:
                sim.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        raw = sim.sim if sharded else sim

        def step(k):
            # (a heartbeat every few ticks: a long pre-roll is progress, a stuck collective is not)
            while k > 0:
                j = min(k, 20)
                sim.step(j)
                progress(f"step (tick {raw.tick})")
                k -= j

        for t, op, node, a, b in ops:
            sim.inject(t, op, node, a, b)

        def barrier():
            if world > 1:
                dist.barrier()
            if on_gpu:
                torch.cuda.synchronize()
            else:
                raw.sync()
            progress("barrier")

        def load_now():  # cluster-wide load: records per packet in flight, queue entries per node, model-bound drops
            if sharded:
                sim.sync()  # the round's exchanges have landed
            cs = raw.cluster_stats()
            inbox, queued, drops, up, mxq = allsum([cs["inbox_records"], sum(cs["queued"]), cs["overflow"], cs["up"], 0])
            # operations skipped for want of a view slot are counted on the (replicated) schedule, the same on every rank
            return {"records_per_packet": inbox / (args.fanout * n_total), "queued_per_node": queued / n_total,
                    "drops": drops + int(cs["ops_dropped"]), "up": up, "slots_in_use": int(cs["slots_in_use"]),
                    "slots_recycled": int(cs["slots_recycled"]), "max_queue": allmax(cs["max_queue"])}

        def timed(k, profile_every):
            """exactly k steps between two barriers; the launches go to torch's current stream (sim_set_stream above), so one
            pair of torch events brackets them as well"""
            raw.profile(profile_every)
            barrier()
            t0 = time.perf_counter()
            ev0, ev1 = Event(enable_timing=True), Event(enable_timing=True)
            ev0.record()
            if os.environ.get("SERF_BENCH_CPROFILE"):  # where the host's share goes (measurements: tools/gpu/r6_onerank_ab.sh)
                import cProfile, pstats
                pr = cProfile.Profile()
                pr.runcall(sim.step, k)
                pstats.Stats(pr, stream=sys.stderr).sort_stats("cumulative").print_stats(30)
            else:
                sim.step(k)
            host = time.perf_counter() - t0  # the host's share: every launch of the k steps has been enqueued (nothing waited for)
            ev1.record()
            barrier()
            dt = time.perf_counter() - t0
            ev_ms = ev0.elapsed_time(ev1)
            if world > 1:
                t = torch.tensor([dt], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt = float(t[0])
            (ms, mn, mx), cnt = raw.profile_read_stats()
            raw.profile(False)
            kern_s = ms / 1e3 / max(1, cnt) if ms > 0 else ev_ms / 1e3 / k  # (the oracle behind a CPU test has no kernel)
            return {"dt": dt, "ev_ms": ev_ms, "kern_s": kern_s, "kmin": mn, "kmax": mx, "kn": int(cnt), "every": profile_every, "host_s": host}

        m = {"model": model, "ops": ops}
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
        m["trace"] = trace
        m["load0"] = load_now()
        # parity block: the state the timed region starts from and the state it ends in, digested (all 8 arrays); rank 0's
        # cpu_baseline() rolls the oracle through the same schedule to the same ticks and compares
        m["parity_ticks"] = [args.preroll + args.warmup, args.preroll + args.warmup + args.steps]
        want_parity = world == 1 and not args.no_cpu_baseline and (model == models[0] or args.parity_all)
        m["digests"] = {}
        if want_parity:
            m["digests"][m["parity_ticks"][0]] = raw.digest()
        # an event pair costs ~10 us of stream time: time a sample of the launches on long runs, all of them on short ones
        m["timed"] = timed(args.steps, 4 if args.steps >= 100 else 1)
        if want_parity:
            m["digests"][m["parity_ticks"][1]] = raw.digest()
        m["load1"] = load_now()
        # second window: 300 further ticks timed the same way — push-pull batch (every 300) and recycling passes (every 75) inside
        m["long"] = timed(long_window(args), 4) if long_window(args) else None
        # N > 1 diagnosis, outside the timed region: the same ticks with every all-to-all bracketed by events and run
        # synchronously (no overlap) — what one round's exchange costs on its own, next to the kernel
        m["exchange_ms"], diag_ticks = None, 20
        if sharded:
            barrier()
            sim.time_exchange(True)
            td0 = time.perf_counter()
            step(diag_ticks)
            barrier()
            m["serial_ms"] = (time.perf_counter() - td0) * 1e3 / diag_ticks
            m["exchange_ms"] = sim.time_exchange(False) / diag_ticks
            m["diag_ticks"] = diag_ticks
            m["chunks"] = sim.chunks
            m["exchange_bytes"] = raw.exchange_bytes()
            m["collectives"] = sim.collective_library()
        # ---- second half of the metric: rounds to 99 % convergence of the workload's OWN user events (no extra load),
        # every one issued in a fixed window of ticks, all outstanding ones polled with one launch per tick
        rounds, conv_first, conv_last = [], None, None
        if not args.no_convergence:
            c0 = conv_start_tick(args)
            step(c0 - raw.tick)
            evs = [(t, node, a) for (t, op, node, a, b) in ops if op == _ffi.OP_USER_EVENT and c0 <= t < c0 + CONV_WINDOW][:CONV_RUMOURS]
            by_tick = {}
            for t, node, key in evs:
                by_tick.setdefault(t, []).append((node, key))
            outstanding = {}  # key -> (ltime, tick issued)
            last = (max(by_tick) if by_tick else c0) + CONV_MAX_ROUNDS
            while raw.tick <= last and (outstanding or raw.tick <= (max(by_tick) if by_tick else c0)):
                t = raw.tick
                bump = {}
                for node, key in by_tick.get(t, ()):  # the Lamport time the event is going to get: its origin's event clock now
                    owner = node // args.nodes_per_gpu
                    lt = allmax(raw.stats(node).event_time if owner == rank else 0) + bump.get(node, 0)
                    bump[node] = bump.get(node, 0) + 1
                    outstanding[key] = (lt, t)
                step(1)
                keys = list(outstanding)
                if keys:
                    seen, up = sim.convergence_many([(_ffi.K_EVENT, k, outstanding[k][0]) for k in keys])
                    for k, sn in zip(keys, seen):
                        r = raw.tick - outstanding[k][1]
                        if sn * 100 >= up * 99 or r > CONV_MAX_ROUNDS:
                            rounds.append(r)
                            del outstanding[k]
            rounds += [CONV_MAX_ROUNDS + 1] * len(outstanding)
            conv_first, conv_last = c0, raw.tick
        m["rounds"], m["conv"] = rounds, (conv_first, conv_last)
        m["load2"] = load_now()
        # what the cluster holds at the end of the run: the planes of the view / the rings that ever got memory (sim_resident_planes:
        # slots handed out, Lamport times admitted) and what the device reports in use (everything on it: HIP context, torch, RCCL)
        fp = {}
        if "resident_planes" in raw.lib.f:
            rp = raw.resident_planes()
            fp = {"view_planes": list(rp["view"]), "event_ring_planes": list(rp["event_ring"]), "query_ring_planes": list(rp["query_ring"]),
                  "bytes_per_plane": rp["bytes_per_plane"],
                  "view_and_rings_resident_bytes": (rp["view"][0] + rp["event_ring"][0] + rp["query_ring"][0]) * rp["bytes_per_plane"],
                  "view_and_rings_whole_bytes": (rp["view"][1] + rp["event_ring"][1] + rp["query_ring"][1]) * rp["bytes_per_plane"]}
        if on_gpu:
            free, tot = torch.cuda.mem_get_info(dev)
            fp["device_bytes_in_use"] = int(tot - free)
            fp["device_GiB_in_use_per_Mi_nodes"] = round((tot - free) / 2 ** 30 / (args.nodes_per_gpu / 2 ** 20), 2)
        m["footprint"] = fp
        raw.close()  # two clusters need not sit next to each other
        return m

    def roofline_of(m):
        """the `roofline` object of one measured cluster (dominant kernel = tick_kernel: one launch per tick; HIP events around
        the launches of the timed region, sim_profile)"""
        model, t = m["model"], m["timed"]
        bt, bt2 = b_tick_v0(args.fanout), b_tick_layout(args.fanout)
        kern_s = t["kern_s"]
        algorithmic = args.nodes_per_gpu * bt / kern_s / 1e9
        layout = args.nodes_per_gpu * bt2 / kern_s / 1e9
        traffic, prov = measured_traffic(model)
        default_load = args.pkt_records == 4 and args.rate == 0.25 and args.nodes_per_gpu == 1 << 20 and world == 1
        # (the bytes a launch moves follow the load of the ticks it covers: the profile's figure is this run's only when the
        # same ticks are timed)
        same_ticks = bool(prov) and prov["timed_ticks_of_the_profile"] == {"steps": args.steps, "warmup": args.warmup}
        # `achieved` / `frac` follow the measurement contract: ALGORITHMIC bytes per launch (SURVEY.md §8d's per-member-tick
        # figure x the nodes one launch processes) over the kernel's average launch duration.  What is actually on the pins is
        # `traffic` (PMC counters, when they were collected on this kernel source, fan-out model, load and timed ticks) and
        # `frac_measured` = traffic over the same duration over the 8 TB/s peak — the number to read as bandwidth.
        if traffic and prov["matches_this_kernel"] and default_load and same_ticks:
            measured = {"achieved": traffic / kern_s / 1e9, "frac": traffic / kern_s / 1e9 / 8000.0, "unit": "GB/s",
                        "what": "HBM bytes per launch measured with rocprofv3 PMC counters on this kernel source (roofline.traffic) / average launch duration"}
        else:
            traffic = None  # a figure measured on another kernel source (or another load) is not this run's traffic
            measured = None
        every = t["every"]
        out = {"bound": "hbm", "achieved": algorithmic, "peak": 8000.0, "unit": "GB/s", "frac": algorithmic / 8000.0,
               "frac_measured": measured["frac"] if measured else None, "traffic": traffic, "measured": measured,
               "achieved_from": "the contract's formula: SURVEY.md §8d algorithmic bytes per member-tick x nodes per launch / average launch "
                                "duration; bytes actually moved: roofline.traffic and roofline.frac_measured (PMC counters) — null when no "
                                "PMC profile of this kernel source, fan-out model, load and timed ticks is on file",
               "kernel": "tick_kernel", "kernel_ms": kern_s * 1e3, "kernel_ms_min": t["kmin"], "kernel_ms_max": t["kmax"],
               "kernel_launches": t["kn"],
               "kernel_timing": f"HIP event pair on every {every}{'th' if every > 1 else 'st'} tick_kernel dispatch of the timed region "
                                "(hipExtLaunchKernelGGL start/stop events, on the launch stream)",
               "stream_ms_per_step": t["ev_ms"] / args.steps, "host_ms_per_step": t["host_s"] / args.steps * 1e3,
               "algorithmic": {"b_tick_bytes": bt, "achieved": algorithmic, "frac": algorithmic / 8000.0,
                               "what": "SURVEY.md §8d B_tick(f) = 2R + 2QE + 2fPE + 4(f+2) x nodes / kernel time"},
               "traffic_over_algorithmic": (traffic / (args.nodes_per_gpu * bt)) if traffic else None,
               "traffic_provenance": prov}
        if model == "bijection":
            out["layout"] = {"b_tick_bytes": bt2, "achieved": layout, "frac": layout / 8000.0,
                             "what": "the same accounting for the bijection's frozen layout (DESIGN.md §4): packets kept at the sender"}
        else:
            out["graph_build"] = ("the fan-out graph of a tick (rf_scatter + rf_rows: ~0.05 ms at 1 Mi nodes when they run alone; built two ticks "
                                  "ahead on a stream of its own, next to the tick kernels) is part of ms_per_step and of `value`, not of kernel_ms: "
                                  "profiles/r04_kernel_stats_krandomnodes.csv")
        return out

    def rounds_of(m):
        rounds, (conv_first, conv_last) = m["rounds"], m["conv"]
        if not rounds:
            return None
        return {"median": float(np.median(rounds)), "p90": float(np.percentile(rounds, 90)), "max": int(max(rounds)),
                "min": int(min(rounds)), "mean": float(np.mean(rounds)), "n": len(rounds),
                "histogram": {str(r): int(c) for r, c in zip(*np.unique(rounds, return_counts=True))},
                "window_ticks": [conv_first, conv_last], "fanout_model": m["model"],
                "parity": "memberlist half (queue order and limit, peer selection, loss) is parity-UNPINNED: DESIGN.md §6",
                "what": "gossip rounds until >= 99 % of running nodes have applied a user event, for every user event the workload itself "
                        f"issues in ticks [{conv_first}, {conv_first + CONV_WINDOW}) (a fixed window: independent of --steps / --warmup), "
                        "all outstanding events polled once per tick (sim_convergence_many)"}

    def exchange_of(m):
        """the `exchange` object of one measured sharded cluster"""
        xb = m["exchange_bytes"]
        packed = m["model"] == "krandomnodes"   # the random fan-out on shards: the slabs are packed from the packets the senders keep
        return {"chunks": m["chunks"], "exchange_ms": m["exchange_ms"], "kernel_ms": m["timed"]["kern_s"] * 1e3,
                           "collective": ("equal-split all-to-all of packed slabs (SIM_XCHG_PACKED: (target, sender, slot)-sorted 64-byte cells, "
                                          "one count byte per target; mean + 12 sigma of room per slab)") if packed else "equal-split all-to-all per chunk",
                           "serial_ms_per_step": m["serial_ms"], "overlapped_ms_per_step": m["timed"]["dt"] / args.steps * 1e3,
                           "host_ms_per_step": m["timed"]["host_s"] / args.steps * 1e3,  # the host's enqueue time per tick (sharded: the Python loop of serf_amd/shard.py)
                           "bytes_per_peer": xb // world, "bytes_per_gpu_per_tick": xb,
                           "bytes_leaving_gpu_per_tick": xb // world * (world - 1),
                           "bytes_arriving_per_gpu_per_tick": xb // world * (world - 1),
                           "packet_bytes_per_gpu_per_tick": args.fanout * args.nodes_per_gpu * (64 if packed else 48) * max(1, args.pkt_records // 4),
                           "what": f"the timed region runs each tick as {m['chunks']} chunk launches with the all-to-all of chunk c in flight "
                                   "while chunk c + 1 computes (overlapped_ms_per_step = ms_per_step); exchange_ms and serial_ms_per_step come "
                                   f"from {m['diag_ticks']} further ticks with the collectives run one after the other between events (rank 0): "
                                   "exchange_ms = all-to-alls of one round, serial_ms_per_step = that round without overlap"}

    def summary_of(m):
        t, lw = m["timed"], m["long"]
        d = {"what": MODEL_WHAT[m["model"]], "value": n_total * args.steps / t["dt"], "unit": "member-ticks/s",
             "ms_per_step": t["dt"] / args.steps * 1e3, "kernel_ms": t["kern_s"] * 1e3, "roofline": roofline_of(m),
             "rounds_to_99": rounds_of(m), "model_bound_drops": m["load2"]["drops"],
             "records_per_packet": [round(m["load0"]["records_per_packet"], 3), round(m["load1"]["records_per_packet"], 3)],
             "deepest_queue": m["load1"]["max_queue"], "footprint": m["footprint"]}
        if sharded:
            d["exchange"] = exchange_of(m)
        if lw:
            bt = b_tick_v0(args.fanout)
            alg = args.nodes_per_gpu * bt / lw["kern_s"] / 1e9
            tr, prov = measured_traffic(m["model"], long=True)
            ok = bool(tr) and prov["matches_this_kernel"] and args.pkt_records == 4 and args.rate == 0.25 and args.nodes_per_gpu == 1 << 20 and world == 1 \
                and prov["timed_ticks_of_the_profile"] == {"steps": long_window(args), "warmup": args.warmup + args.steps}
            d["long_window"] = {"steps": long_window(args), "value": n_total * long_window(args) / lw["dt"], "ms_per_step": lw["dt"] / long_window(args) * 1e3,
                                "kernel_ms": lw["kern_s"] * 1e3, "kernel_ms_max": lw["kmax"],
                                "roofline": {"bound": "hbm", "achieved": alg, "peak": 8000.0, "unit": "GB/s", "frac": alg / 8000.0,
                                             "traffic": tr if ok else None, "frac_measured": (tr / lw["kern_s"] / 1e9 / 8000.0) if ok else None,
                                             "traffic_provenance": prov,
                                             "what": "the same two figures as `roofline`, over the long window's launches: the contract's formula, and the "
                                                     "bytes the PMC counters saw on these very ticks over the kernel's time (null without a profile of this "
                                                     "kernel source and these ticks)"},
                                "timed_region_over_long_window_kernel_ms": t["kern_s"] / lw["kern_s"],
                                "ticks": [m["parity_ticks"][1], m["parity_ticks"][1] + long_window(args) - 1],
                                "what": "the ticks right behind the timed region, measured the same way: a push-pull batch (every 300 ticks at this "
                                        "size) and the recycling passes (every 75) fall inside"}
        return d

    res = {}
    for model in models:
        res[model] = measure(model)
    out = None
    drops = sum(res[mo]["load2"]["drops"] for mo in models)
    if rank == 0:
        head = res[models[0]]
        t = head["timed"]
        live = "about 10 rumours live at any time (0.41 new ones per tick, each alive ~20 ticks) — SURVEY.md §8d's config 3 asks for 1 024, which neither " \
               "the reference's 1 400-byte packets (~220) nor this model's 16-slot queue (~12) can carry at this size (DESIGN.md §7)"
        out = {
            "metric": "member-ticks/sec", "value": n_total * args.steps / t["dt"], "unit": "member-ticks/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": t["dt"] / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": f"{n_total} nodes ({args.nodes_per_gpu}/GPU), fan-out {args.fanout}; HEADLINE fan-out model: {models[0]} — {MODEL_WHAT[models[0]]}"
                                   + (f" (the other model, {models[1]}, under fanout_models)" if len(models) > 1 else "") + "; "
                                   f"{args.rate} API ops/tick evenly spaced, mix (0.55, 0.2, 0.15, 0.05, 0.05) of (user event, query, leave, crash+remove, crash+revive) evenly interleaved: {live}; "
                                   f"{args.pkt_records} records per packet, view_slots {args.view_slots}, rings {args.ring}, probe interval {args.probe_interval} ticks, push-pull interval "
                                   f"{args.push_pull_interval} ticks (x log2 scaling), reaper and queue checker on — BASELINE configs[2]; "
                                   f"{args.preroll} untimed pre-roll ticks under the same load before the warm-up (steady state); the timed region is the "
                                   f"{args.steps} ticks the driver asks for (not SURVEY §8d's 1 000)"
                                   + (f", the {long_window(args)} ticks behind it are measured as well (long_window)" if long_window(args) else ""),
                       "workload_short": f"BASELINE configs[2]: {n_total} nodes ({args.nodes_per_gpu}/GPU), fan-out {args.fanout}, {models[0]}, {args.rate} API ops/tick, "
                                         f"{args.pkt_records} records/packet, view_slots {args.view_slots}, rings {args.ring}, probe {args.probe_interval}, push-pull {args.push_pull_interval}, "
                                         f"{args.preroll} pre-roll ticks",
                       "parallelism_short": f"node-id range shards x{world}, {args.chunks} chunks, one all-to-all per round" if sharded else "single GPU",
                       "fanout_model": models[0],
                       "parallelism": ((f"node-id range shards x{world}; " + (f"kRandomNodes: packets packed per destination shard behind each of the {args.chunks} sender-chunk launches, an "
                                                                               "equal-split all-to-all of the packed slabs per chunk, overlapped with compute" if models[0] == "krandomnodes" else
                                                                               f"{args.chunks} chunk-wise all-to-all per tick, overlapped with compute"))
                                       if sharded else "single GPU"),
                       "preroll": args.preroll, "schedule_horizon": horizon(args),
                       "timed_ticks": [args.preroll + args.warmup, args.preroll + args.warmup + args.steps - 1],
                       "model_bound_drops": drops,
                       "load": {"records_per_packet_start": round(head["load0"]["records_per_packet"], 3),
                                "records_per_packet_end": round(head["load1"]["records_per_packet"], 3),
                                "queued_per_node_start": round(head["load0"]["queued_per_node"], 3),
                                "queued_per_node_end": round(head["load1"]["queued_per_node"], 3),
                                "deepest_queue": head["load1"]["max_queue"],
                                "records_per_packet_preroll_every_40_ticks": head["trace"],
                                "nodes_up": head["load1"]["up"], "view_slots_in_use": head["load2"]["slots_in_use"],
                                "view_slots_recycled": head["load2"]["slots_recycled"]},
                       "footprint": head["footprint"]},
            "rounds_to_99": rounds_of(head),
            "roofline": roofline_of(head),
            "fanout_models": {mo: summary_of(res[mo]) for mo in models},
        }
        if head["long"]:
            out["long_window"] = out["fanout_models"][models[0]]["long_window"]
            # the driver's K steps are a SAMPLE of the run: the long window (the 300 ticks behind them, a push-pull batch and four
            # recycling passes inside) is the representative figure — VERDICT r4 item 3
            out["value_long_window"] = out["long_window"]["value"]
            out["config"]["workload"] += (f"; kernel time of the timed region / of the long window = {out['long_window']['timed_region_over_long_window_kernel_ms']:.2f} "
                                          "(value_long_window is the representative rate)")
        if sharded:
            out["exchange"] = exchange_of(head)
            out["distributed"] = {"backend": dist.get_backend() if world > 1 else None, "world_size": world,
                                  "collective_library": head["collectives"]}
        if world == 1 and on_gpu and not args.no_second_load and args.pkt_records == 4 and not sharded:
            progress("second load")
            out["second_load"] = second_load(args, lib, dev, torch)
        if world == 1 and not args.no_cpu_baseline:
            out["parity"] = {}
            for i, mo in enumerate(models):
                if i > 0 and not args.parity_all:   # one oracle roll per run by default: the lease is for the GPU
                    continue
                progress(f"cpu_baseline ({mo})")
                m = res[mo]
                cb, cpu_dig = cpu_baseline(args, mo, m["parity_ticks"], timed=(i == 0))
                if i == 0:
                    out["cpu_baseline"] = cb
                pt = m["parity_ticks"]
                if any(cpu_dig[tk] is None for tk in pt):
                    out["parity"][mo] = {"ticks": pt, "digest_match": None, "why": "the host could not hold the configuration: oracle ran a smaller cluster"}
                else:
                    ok = [tuple(cpu_dig[tk]) == tuple(m["digests"][tk]) for tk in pt]
                    out["parity"][mo] = {"ticks": pt, "digest_match": all(ok), "digest_match_per_tick": ok, "arrays": 8,
                                         "what": "sim_state_digest of the GPU run right before the timed region AND right behind its last timed tick (rows, "
                                                 "queues, packets in flight, views, both rings, slot map + liveness, query tables) against the CPU "
                                                 "oracle rolled through the same schedule to the same ticks (oracle/liboracle.so, the checker): the "
                                                 "timed launches themselves are covered",
                                         "gpu": {str(tk): [f"{x:016x}" for x in m["digests"][tk]] for tk in pt},
                                         "oracle": {str(tk): [f"{x:016x}" for x in cpu_dig[tk]] for tk in pt}}
            out["parity"]["digest_match"] = all(v.get("digest_match") is not False for v in out["parity"].values() if isinstance(v, dict))
            out["parity"]["ticks"] = head["parity_ticks"]
        emit_result(out)
    progress.done = True
    if world > 1 or args.force_sharded:
        dist.destroy_process_group()
    if out is not None and out.get("parity", {}).get("digest_match") is False:
        raise SystemExit("bench.py: the GPU state at the ends of the timed region differs from the CPU oracle's — result invalid")
    if drops and not args.allow_drops:
        # a run that hit a model bound is not a run of the protocol the parity tests cover: refuse it
        raise SystemExit(f"bench.py: model bound hit ({drops} drops: queue slots / bucket keys / timers) — result invalid")
    return out


def self_launch(args, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks, wait, tear down on failure; one retry with
    `--chunks 1` (one synchronous exchange per tick: the simplest schedule) before giving up with an error line."""
    import socket

    def attempt(extra):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        procs = []
        for r in range(args.gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            # rank 0 inherits stdout (its JSON line is the result); the others keep stderr only
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(sys.argv[0])] + argv + extra, env=env,
                                          stdout=None if r == 0 else subprocess.DEVNULL))
        deadline = time.monotonic() + args.launch_timeout
        rc, why = 0, ""
        while True:
            codes = [p.poll() for p in procs]
            if any(c not in (None, 0) for c in codes):
                rc, why = next(c for c in codes if c not in (None, 0)), f"rank {next(i for i, c in enumerate(codes) if c not in (None, 0))} exited with a failure"
                break
            if all(c == 0 for c in codes):
                break
            if time.monotonic() > deadline:
                rc, why = 124, f"ranks still running after {args.launch_timeout:.0f} s"
                break
            time.sleep(0.2)
        for p in procs:  # our own children, by handle
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(10)
            except subprocess.TimeoutExpired:
                p.kill()
        return rc, why

    rc, why = attempt([])
    if rc and args.chunks > 1:
        sys.stderr.write(f"bench.py: {why} (exit code {rc}); retrying with --chunks 1\n")
        rc, why = attempt(["--chunks", "1"])
    if rc:
        print(json.dumps({"metric": "member-ticks/sec", "value": None, "unit": "member-ticks/s", "n_gpus": args.gpus, "steps": args.steps,
                          "warmup": args.warmup, "error": f"{why} (exit code {rc}), also with --chunks 1"}), flush=True)
    return rc


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args, sys.argv[1:]))
    run(args, backend=args.backend)


if __name__ == "__main__":
    main()
