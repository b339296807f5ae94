#!/usr/bin/env python
"""BASELINE configs[4] AT ITS SIZE on the reference's fan-out model: 16 Mi nodes sharded 8-way, memberlist's literal kRandomNodes,
crash / re-join churn + 1 % packet loss with the failure detector on, convergence-round histogram — on ONE GPU: eight shard handles of
2 Mi nodes each (the handles a rank would hold on eight GPUs), driven exactly like eight ranks — sim_step_begin, the cross-shard
push-pull batches, sim_step_chunk (tick kernel + the pack of the round's slabs), the round's equal-split all-to-all of the packed
slabs (SIM_XCHG_PACKED) as device-to-device copies standing in for RCCL, the slot-less suspicions' hand-over —, one handle after the
other.  What it shows: the sharded form of the headline model exists at 16 Mi / 8 (round 4: sim_create refused it), its exchange moves
f * 64 * M * (V - 1) / V bytes per shard and round (+ 2 %), and rumours converge under churn and loss; what it cannot show is a rate
(eight shards take turns on one GPU).  The churn runs at the pace the view slots allow (one crash per `--churn-every` ticks, each node
down `--down` ticks, then Serf::join): a fraction of configs[4]'s 5 % in a run of this length — said in the JSON.

(r6) The pace of the churn is set by SIM_S = 16 suspicion timers per node (a crashed node is a running suspicion at every node for ~125
ticks), not by batching: 5 % of 16 Mi nodes at one crash per 12 ticks would be 1.0e7 cluster ticks of 25 ms — the full dose is run on the
headline model at 2 Mi nodes (tools/config4_run.py, profiles/r06_config4_2m_krandomnodes_churn5_loss1.json).

Needs an MI355X.  Writes one JSON (default profiles/r06_config4_16m_8shards_krandomnodes.json)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=1 << 24)
    ap.add_argument("--shards", type=int, default=8)
    ap.add_argument("--ticks", type=int, default=700)
    ap.add_argument("--churn-every", type=int, default=12)
    ap.add_argument("--down", type=int, default=170)
    ap.add_argument("--rumors", type=int, default=48)
    ap.add_argument("--loss", type=float, default=0.01)
    ap.add_argument("--view-slots", type=int, default=128)
    ap.add_argument("--max-rounds", type=int, default=60)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r06_config4_16m_8shards_krandomnodes.json"))
    args = ap.parse_args()

    import numpy as np
    import torch

    import serf_amd
    from serf_amd import _ffi
    from tests.test_parity_gpu import _push_pull_on_one_gpu, _suspicions_on_one_gpu, _packed_exchange_on_one_gpu

    lib = serf_amd.load()
    n, V = args.nodes, args.shards
    m = n // V
    kw = dict(fanout=4, view_slots=args.view_slots, event_ring=64, query_ring=32, probe_interval=5, push_pull_interval=150, loss=args.loss,
              reap_interval=75, queue_check_interval=150, pkt_records=16, tcp_fallback=True, nacks=True, join_sync=True, ring_overflow=8,
              flags=_ffi.CF_BASELINE_JOINED | _ffi.CF_RANDOM_FANOUT)
    t0 = time.perf_counter()
    shards, send, recv = [], [], []
    for g in range(V):
        s = _ffi.Sim(lib, _ffi.make_config(n, vshards=V, shard_rank=g, shard_count=V, **kw))
        kind, planes, pb, rb = s.exchange_layout()
        assert kind == _ffi.XCHG_PACKED
        send.append(torch.zeros(pb, dtype=torch.uint8, device="cuda"))
        recv.append(torch.zeros(rb, dtype=torch.uint8, device="cuda"))
        s.bind_exchange3(send[-1].data_ptr(), pb, recv[-1].data_ptr(), recv[-1].data_ptr(), rb)
        shards.append(s)
    torch.cuda.synchronize()
    free, tot = torch.cuda.mem_get_info()
    t_create = time.perf_counter() - t0
    rng = np.random.default_rng(5)
    n_churn = max(1, (args.ticks - args.down - 60) // args.churn_every)
    churned = rng.choice(n, n_churn, replace=False).tolist()
    down = {}
    for i, node in enumerate(churned):
        t = 20 + i * args.churn_every
        down[node] = (t, t + args.down)
        for s in shards:   # operations are replicated: every shard applies the ones of the nodes it owns
            s.inject(t, _ffi.OP_CRASH, node)
            s.inject(t + args.down, _ffi.OP_JOIN, node, int(rng.integers(0, n)))
    every = max(1, (args.ticks - 120) // args.rumors)
    rumor_ticks = [40 + i * every for i in range(args.rumors)]
    rounds, outstanding, ri = [], {}, 0
    stats = {"max_failed_entries": 0, "max_slots_in_use": 0, "max_queue": 0, "push_pull_batches": 0}
    t1 = time.perf_counter()
    while shards[0].tick < args.ticks:
        t = shards[0].tick
        if ri < len(rumor_ticks) and t == rumor_ticks[ri]:
            node = int(rng.integers(0, n))
            while node in down and down[node][0] - 2 <= t <= down[node][1] + 2:
                node = int(rng.integers(0, n))
            key = 0x40000000 + ri
            outstanding[key] = (shards[node // m].stats(node).event_time, t)
            for s in shards:
                s.user_event(node, key, 64)
            ri += 1
        for s in shards:
            s.step_begin()
        if shards[0].pp_due():
            _push_pull_on_one_gpu(shards)
            stats["push_pull_batches"] += 1
        for s in shards:
            s.step_chunk(0)
        for s in shards:
            s.step_end()
            s.sync()
        _packed_exchange_on_one_gpu(send, recv)      # the round's all-to-all: slab g of shard src -> slab src of shard g
        _suspicions_on_one_gpu(shards)
        torch.cuda.synchronize()
        keys = list(outstanding)
        if keys:
            per = [s.convergence_many([(_ffi.K_EVENT, k, outstanding[k][0]) for k in keys]) for s in shards]
            up = sum(p[1] for p in per)
            for j, k in enumerate(keys):
                seen = sum(p[0][j] for p in per)
                r = shards[0].tick - outstanding[k][1]
                if seen * 100 >= up * 99:
                    rounds.append(r)
                    del outstanding[k]
                elif r > args.max_rounds:
                    rounds.append(args.max_rounds + 1)
                    del outstanding[k]
        if t % 50 == 0:
            cs = [s.cluster_stats() for s in shards]
            stats["max_failed_entries"] = max(stats["max_failed_entries"], sum(int(c["failed"]) for c in cs))
            stats["max_slots_in_use"] = max(stats["max_slots_in_use"], int(cs[0]["slots_in_use"]))
            stats["max_queue"] = max(stats["max_queue"], max(int(c["max_queue"]) for c in cs))
    dt = time.perf_counter() - t1
    cs = [s.cluster_stats() for s in shards]
    r = np.array(rounds) if rounds else np.array([0])
    inc = [s.dump(_ffi.ARR_ROWS)["inc"] for s in shards]
    out = {
        "what": "BASELINE configs[4] at its size on the reference's fan-out model, on ONE GPU: 16 Mi nodes as 8 shard handles of 2 Mi nodes (the 8 ranks' "
                "handles), memberlist's kRandomNodes, the packed exchange (SIM_XCHG_PACKED) with device-to-device copies standing in for RCCL, "
                "crash / re-join churn + packet loss with the failure detector on; rounds until >= 99 % of the running nodes have applied a user event",
        "config": {k: v for k, v in vars(args).items() if k != "out"}, "backend": lib.backend_name(),
        "ticks": int(shards[0].tick), "churn_events": int(n_churn), "churn_frac_of_nodes": n_churn / n,
        "churn_note": "configs[4] asks for 5 %% churn: at one crash per %d ticks (the pace the view slots and suspicion timers allow) that is %.1e ticks; this run shows the "
                      "mechanism at the full size, not the full dose (the full dose on one rank's share: profiles/r04_config4_shard_size_2m_v8.json)" % (args.churn_every, 0.05 * n * args.churn_every),
        "rumors": int(len(rounds)),
        "rounds_to_99": {"median": float(np.median(r)), "p90": float(np.percentile(r, 90)), "max": int(r.max()), "min": int(r.min()), "not_converged": int((r > args.max_rounds).sum()),
                         "parity": "memberlist half (queue order and limit, peer selection, loss) is parity-UNPINNED: DESIGN.md §6"},
        "histogram": {int(k): int(v) for k, v in zip(*np.unique(r, return_counts=True))},
        "exchange": {"kind": "SIM_XCHG_PACKED", "bytes_per_shard_send_buffer": int(send[0].numel()), "slab_bytes": int(send[0].numel() // V),
                     "bytes_leaving_a_shard_per_round": int(send[0].numel() // V * (V - 1)),
                     "packet_bytes_per_shard_per_round": 4 * 4 * 64 * m,   # fan-out 4, 4 pages of 64-byte cells per packet
                     "round_4_all_gather_would_receive_per_shard_per_round": 4 * 4 * 64 * n},
        "failure_detector": {"max_failed_entries_seen_cluster_wide": stats["max_failed_entries"],
                             "refutations_incarnation_sum": int(sum(int(x.sum()) for x in inc)),
                             "nodes_that_refuted": int(sum(int((x > 0).sum()) for x in inc))},
        "push_pull_batches_across_shards": stats["push_pull_batches"],
        "model_bound_drops": int(sum(int(c["overflow"]) for c in cs)), "ops_dropped_no_slot": int(cs[0]["ops_dropped"]),
        "max_view_slots_in_use_seen": stats["max_slots_in_use"], "deepest_queue_seen": stats["max_queue"],
        "nodes_up_at_end": int(sum(int(c["up"]) for c in cs)),
        "create_s": t_create, "wall_s": dt, "ms_per_cluster_tick_8_shards_taking_turns_incl_host": dt / max(1, int(shards[0].tick)) * 1e3,
        "device_memory": {"in_use_bytes_after_create": int(tot - free), "in_use_bytes_at_the_end": int(tot - torch.cuda.mem_get_info()[0]), "total_bytes": int(tot),
                          "per_Mi_nodes_GiB_after_create": (tot - free) / (n / 2 ** 20) / 2 ** 30,
                          "resident_planes_shard_0": shards[0].resident_planes() if "resident_planes" in lib.f else None,
                          "note": "view planes get their memory as slots are handed out (every shard maps the same planes: the slot bookkeeping is replicated); a shard's rings are whole"},
    }
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    json.dump(out, open(args.out, "w"), indent=1)
    print(json.dumps({k: out[k] for k in ("ticks", "churn_events", "rumors", "rounds_to_99", "histogram", "model_bound_drops", "ops_dropped_no_slot", "wall_s",
                                          "ms_per_cluster_tick_8_shards_taking_turns_incl_host", "device_memory", "exchange", "failure_detector")}))
    for s in shards:
        s.close()


if __name__ == "__main__":
    main()
