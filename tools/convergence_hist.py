#!/usr/bin/env python
"""Rounds-to-99 % convergence histogram under packet loss and churn (BASELINE configs[4] style, on one GPU).

One user event is injected every `--every` ticks from a random running node; every `--churn-every` ticks a
node crashes and comes back `--down` ticks later (the SWIM layer suspects it, it refutes).  For every rumour
the number of gossip rounds until >= 99 % of the running nodes have applied it is recorded.
Writes a JSON histogram (default profiles/r01_convergence_hist.json).  Needs an MI355X.

Model bound: with the SWIM layer on, every churned node is a gossip subject that needs a view
slot and slots are not recycled, so the number of churn events is limited by --view-slots, not by a percentage
of N.  `--churn-frac 0.05 --probe-interval 0` is BASELINE configs[4]'s "5 % churn + 1 % loss" for the serf layer
alone: that fraction of the nodes crashes and comes back `--down` ticks later over the run (nobody detects
them, so nobody needs a slot for them); push-pull repairs what a node missed while it was down.
"""
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
    ap.add_argument("--rumors", type=int, default=1000)
    ap.add_argument("--every", type=int, default=4)
    ap.add_argument("--churn-every", type=int, default=40)
    ap.add_argument("--down", type=int, default=15)
    ap.add_argument("--view-slots", type=int, default=1024)
    ap.add_argument("--probe-interval", type=int, default=5, help="0 = SWIM layer off")
    ap.add_argument("--push-pull-interval", type=int, default=0)
    ap.add_argument("--churn-frac", type=float, default=0.0, help="fraction of the nodes that crash and come back over the run (needs --probe-interval 0)")
    ap.add_argument("--recycle-interval", type=int, default=0, help="view-slot recycling pass every this many ticks (SWIM layer on: lets the churn run over more subjects than view slots)")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r02_convergence_hist.json"))
    args = ap.parse_args()

    import numpy as np
    import serf_amd
    from serf_amd import _ffi

    n = args.nodes
    sim = serf_amd.create(n, fanout=args.fanout, view_slots=args.view_slots, event_ring=512, query_ring=512,
                          probe_interval=args.probe_interval, push_pull_interval=args.push_pull_interval, loss=args.loss,
                          recycle_interval=args.recycle_interval)
    rng = np.random.default_rng(5)
    total_ticks = args.rumors * args.every + 120
    if args.churn_frac > 0:
        assert args.probe_interval == 0, "--churn-frac runs the serf layer alone"
        n_churn = int(n * args.churn_frac)
        when = np.sort(rng.integers(10, total_ticks - args.down - 10, n_churn))
    else:
        n_churn = total_ticks // args.churn_every if args.recycle_interval else min(total_ticks // args.churn_every, args.view_slots - 8)
        when = 10 + np.arange(n_churn) * args.churn_every
    churned = rng.choice(n, n_churn, replace=False)
    for t, node in zip(when.tolist(), churned.tolist()):
        sim.inject(t, _ffi.OP_CRASH, node)
        sim.inject(t + args.down, _ffi.OP_REVIVE, node)
    down_until = {int(node): int(t) + args.down for t, node in zip(when.tolist(), churned.tolist())}
    inflight, rounds = [], []
    t0 = time.perf_counter()
    for tick in range(total_ticks):
        if tick % args.every == 0 and tick // args.every < args.rumors:
            node = int(rng.integers(0, n))
            while node in down_until and tick <= down_until[node] and tick >= down_until[node] - args.down:
                node = int(rng.integers(0, n))
            key = 0x40000000 + tick
            lt = sim.stats(node).event_time
            sim.user_event(node, key, 64)
            inflight.append([tick, key, lt])
        sim.step(1)
        keep = []
        for t_inj, key, lt in inflight:
            seen, up = sim.convergence(_ffi.K_EVENT, key, lt)
            if seen * 100 >= up * 99:
                rounds.append(tick - t_inj + 1)
            elif tick - t_inj >= 100:
                rounds.append(101)  # did not converge within 100 rounds (retransmits exhausted under loss)
            else:
                keep.append([t_inj, key, lt])
        inflight = keep
    dt = time.perf_counter() - t0
    r = np.array(rounds)
    hist = {int(k): int(v) for k, v in zip(*np.unique(r, return_counts=True))}
    rows = sim.dump(_ffi.ARR_ROWS)
    cs = sim.cluster_stats()
    out = {
        "config": vars(args), "ticks": total_ticks, "churn_events": int(n_churn), "rumors": int(len(r)),
        "rounds_to_99": {"median": float(np.median(r)), "p90": float(np.percentile(r, 90)), "p99": float(np.percentile(r, 99)),
                         "max": int(r.max()), "not_converged_in_100": int((r > 100).sum())},
        "histogram": hist,
        "refutations": int(rows["inc"].sum()), "nodes_ever_failed_somewhere": int((rows["n_failed"] > 0).sum()),
        "model_bound_drops": int(rows["overflow"].sum()), "ops_dropped_no_slot": int(cs["ops_dropped"]),
        "view_slots_recycled": int(cs["slots_recycled"]), "view_slots_in_use_at_end": int(cs["slots_in_use"]),
        "wall_s": dt, "member_ticks_per_s_incl_host_polling": n * total_ticks / dt,
    }
    json.dump(out, open(args.out, "w"), indent=1)
    print(json.dumps(out["rounds_to_99"]), "->", args.out)


if __name__ == "__main__":
    main()
