#!/usr/bin/env python
"""BASELINE configs[4] on one GPU: 5 % churn + 1 % packet loss with the failure detector ON, convergence-round histogram.

`--churn-frac` of the nodes crash one after the other (one every `--churn-every` ticks), stay down `--down` ticks — long
enough to be suspected, confirmed and DECLARED FAILED by the SWIM layer (minimum suspicion timeout 120 ticks at 1 Mi
nodes) — and re-join (Serf::join: refuting incarnation + join intent); every gossip packet AND every probe leg is lost with
probability `--loss`.  Probes that fail on live nodes (0.26 per tick at 1 Mi nodes and 1 %) start false suspicions that
are refuted.  `--rumors` user events are injected at regular intervals from random running nodes; for each the number of
gossip rounds until >= 99 % of the running nodes have applied it is recorded.  The run is valid only if no model bound was
hit (`model_bound_drops` == 0).  The pace of the churn is set by the model's per-node capacity: SIM_S = 16 suspicion timers
(a crashed node is a running suspicion at every node for ~125 ticks); the queue holds 64 entries (r6).

Needs an MI355X.  Writes one JSON (default profiles/r03_config4_churn5_loss1_swim.json)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=1 << 20)
    ap.add_argument("--fanout", type=int, default=4)
    ap.add_argument("--loss", type=float, default=0.01)
    ap.add_argument("--churn-frac", type=float, default=0.05)
    ap.add_argument("--churn-every", type=int, default=24, help="ticks between two crashes")
    ap.add_argument("--down", type=int, default=160, help="ticks a crashed node stays down before it re-joins")
    ap.add_argument("--rumors", type=int, default=1000)
    ap.add_argument("--view-slots", type=int, default=1024)
    ap.add_argument("--ring", type=int, default=512)
    ap.add_argument("--probe-interval", type=int, default=5)
    ap.add_argument("--push-pull-interval", type=int, default=150)
    ap.add_argument("--recycle-interval", type=int, default=75)
    ap.add_argument("--pkt-records", type=int, default=4)
    ap.add_argument("--max-rounds", type=int, default=60)
    ap.add_argument("--random-fanout", action="store_true", help="gossip targets by memberlist's literal kRandomNodes (explicit per-tick CSR) instead of the bijection")
    ap.add_argument("--tcp-fallback", action="store_true", help="memberlist's stream-transport fallback ping (its default): packet loss alone never fails a probe")
    ap.add_argument("--nacks", action="store_true", help="memberlist's nack accounting for the health score")
    ap.add_argument("--reconnect-interval", type=int, default=0, help="Reconnector period in ticks (reference: 30 s = 150; 0 = off)")
    ap.add_argument("--gossip-to-the-dead", type=int, default=0, help="memberlist gossip_to_the_dead_time in ticks (lan: 30 s = 150; 0 = off)")
    ap.add_argument("--ring-overflow", type=int, default=8, help="overflow rows per de-dup ring and node (sim_config.ring_overflow)")
    ap.add_argument("--vshards", type=int, default=1, help="virtual shards of the fan-out map (the shape of one rank's share of a V-way sharded cluster)")
    ap.add_argument("--chunks", type=int, default=0, help="sender chunks per shard (the chunk-wise exchange's layout)")
    ap.add_argument("--lib", default=None, help="oracle: run the CPU oracle instead (small sizes; for checking the tool)")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r03_config4_churn5_loss1_swim.json"))
    args = ap.parse_args()

    import numpy as np
    from serf_amd import _ffi

    if args.lib == "oracle":
        from tests._oracle import load_oracle
        lib = load_oracle()
    else:
        import serf_amd
        lib = serf_amd.load()
    n = args.nodes
    kw = dict(fanout=args.fanout, view_slots=args.view_slots, event_ring=args.ring, query_ring=args.ring,
              probe_interval=args.probe_interval, push_pull_interval=args.push_pull_interval, loss=args.loss,
              reap_interval=75, queue_check_interval=150, recycle_interval=args.recycle_interval, pkt_records=args.pkt_records,
              reconnect_interval=args.reconnect_interval, gossip_to_the_dead=args.gossip_to_the_dead,
              tcp_fallback=args.tcp_fallback, nacks=args.nacks, vshards=args.vshards, chunks=args.chunks, ring_overflow=args.ring_overflow,
              **({"flags": _ffi.CF_BASELINE_JOINED | _ffi.CF_RANDOM_FANOUT} if args.random_fanout else {}),
              join_sync=True)   # Serf::join = memberlist.join: the re-joining node syncs with a peer (SIM_CF_JOIN_SYNC)
    sim = _ffi.Sim(lib, _ffi.make_config(n, **kw))
    rng = np.random.default_rng(5)
    n_churn = int(n * args.churn_frac)
    total = 20 + n_churn * args.churn_every + args.down + 400
    churned = rng.choice(n, n_churn, replace=False)
    crash_at = 20 + np.arange(n_churn) * args.churn_every
    for t, node in zip(crash_at.tolist(), churned.tolist()):
        sim.inject(t, _ffi.OP_CRASH, node)
        sim.inject(t + args.down, _ffi.OP_JOIN, node)
    down = {int(node): (int(t), int(t) + args.down) for t, node in zip(crash_at.tolist(), churned.tolist())}
    every = max(1, (total - 100) // args.rumors)
    rumor_ticks = [50 + i * every for i in range(args.rumors)]
    rounds, outstanding = [], {}
    stats = {"max_failed_entries": 0, "max_slots_in_use": 0, "max_queue": 0}
    t0 = time.perf_counter()
    ri = 0
    while sim.tick < total:
        t = sim.tick
        nxt = rumor_ticks[ri] if ri < len(rumor_ticks) else total
        if not outstanding and t < nxt:     # nothing to watch: run ahead to the next rumour
            sim.step(min(nxt, total) - t)
            cs = sim.cluster_stats()
            stats["max_failed_entries"] = max(stats["max_failed_entries"], int(cs["failed"]))
            stats["max_slots_in_use"] = max(stats["max_slots_in_use"], int(cs["slots_in_use"]))
            stats["max_queue"] = max(stats["max_queue"], int(cs["max_queue"]))
            continue
        if ri < len(rumor_ticks) and t == rumor_ticks[ri]:
            node = int(rng.integers(0, n))
            while node in down and down[node][0] - 2 <= t <= down[node][1] + 2:
                node = int(rng.integers(0, n))
            key = 0x40000000 + ri
            outstanding[key] = (sim.stats(node).event_time, t)
            sim.user_event(node, key, 64)
            ri += 1
        sim.step(1)
        keys = list(outstanding)
        seen, up = sim.convergence_many([(_ffi.K_EVENT, k, outstanding[k][0]) for k in keys])
        for k, s in zip(keys, seen):
            r = sim.tick - outstanding[k][1]
            if s * 100 >= up * 99:
                rounds.append(r)
                del outstanding[k]
            elif r > args.max_rounds:
                rounds.append(args.max_rounds + 1)
                del outstanding[k]
    sim.sync()
    dt = time.perf_counter() - t0
    r = np.array(rounds)
    rows = sim.dump(_ffi.ARR_ROWS)
    cs = sim.cluster_stats()
    ev_failed = int((rows["n_failed"] > 0).sum())
    out = {
        "what": "BASELINE configs[4] on one GPU: churn + packet loss with the SWIM layer on; rounds until >= 99 % of the running nodes have "
                "applied a user event",
        "config": {k: v for k, v in vars(args).items() if k not in ("out", "lib")}, "backend": lib.backend_name(),
        "ticks": int(sim.tick), "churn_events": int(n_churn), "churn_frac_of_nodes": n_churn / n, "rumors": int(len(r)),
        "rounds_to_99": {"median": float(np.median(r)), "p90": float(np.percentile(r, 90)), "p99": float(np.percentile(r, 99)),
                         "max": int(r.max()), "min": int(r.min()), "not_converged": int((r > args.max_rounds).sum())},
        "histogram": {int(k): int(v) for k, v in zip(*np.unique(r, return_counts=True))},
        "failure_detector": {"max_failed_entries_seen_cluster_wide": stats["max_failed_entries"],
                             "nodes_still_holding_a_failed_entry_at_end": ev_failed,
                             "refutations_incarnation_sum": int(rows["inc"].sum()),
                             "nodes_that_refuted": int((rows["inc"] > 0).sum()),
                             "awareness_nonzero_at_end": int((rows["awareness"] > 0).sum())},
        "model_bound_drops": int(cs["overflow"]), "ops_dropped_no_slot": int(cs["ops_dropped"]),
        "view_slots_recycled": int(cs["slots_recycled"]), "view_slots_in_use_at_end": int(cs["slots_in_use"]),
        "max_view_slots_in_use_seen": stats["max_slots_in_use"], "deepest_queue_seen": stats["max_queue"],
        "nodes_up_at_end": int(cs["up"]), "wall_s": dt, "member_ticks_per_s_incl_host_polling": n * int(sim.tick) / dt,
    }
    try:
        import torch
        free, tot = torch.cuda.mem_get_info()
        out["device_memory"] = {"in_use_bytes_at_end": int(tot - free), "total_bytes": int(tot),
                                "what": "hipMemGetInfo at the end of the run: this handle's arrays (views, rings, packets, rows) and the runtime's own"}
    except Exception:
        pass
    json.dump(out, open(args.out, "w"), indent=1)
    print(json.dumps({k: out[k] for k in ("ticks", "churn_events", "rounds_to_99", "model_bound_drops", "ops_dropped_no_slot", "failure_detector", "wall_s")}), "->", args.out)


if __name__ == "__main__":
    main()
