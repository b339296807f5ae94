#!/usr/bin/env python
"""Error bar on the fan-out model: rounds-to-99 % of the same rumours under the simulator's
per-tick bijection (every node receives exactly `fanout` packets per round) and under memberlist's literal
kRandomNodes (uniform targets without replacement, skip self — SURVEY.md App. B.2; in-degree Poisson-like),
both on the CPU oracle (round 2: the random mode existed only there; since round 3 the HIP library has SIM_CF_RANDOM_FANOUT
too — `bench.py --random-fanout`), serf layer alone, with packet loss.

    python -m tests.fanout_model_hist --nodes 65536 --rumors 1000 --out profiles/r02_fanout_model_64k.json

Lives under tests/ because it drives the oracle (test infrastructure); the product never runs it.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from serf_amd import _ffi  # noqa: E402
from tests._oracle import load_oracle  # noqa: E402

CF_RANDOM_FANOUT = 2


def run(lib, n, fanout, loss, rumors, every, random_fanout, seed):
    cfg = _ffi.make_config(n, fanout=fanout, view_slots=64, event_ring=512, query_ring=512, loss=loss,
                           flags=_ffi.CF_BASELINE_JOINED | (CF_RANDOM_FANOUT if random_fanout else 0))
    sim = _ffi.Sim(lib, cfg)
    rng = np.random.default_rng(seed)
    inflight, rounds = [], []
    total = rumors * every + 80
    for tick in range(total):
        if tick % every == 0 and tick // every < rumors:
            node = int(rng.integers(0, n))
            key = 0x40000000 + tick
            lt = sim.stats(node).event_time
            sim.user_event(node, key, 64)
            inflight.append((tick, key, lt))
        sim.step(1)
        keep = []
        for t_inj, key, lt in inflight:
            seen, up = sim.convergence(_ffi.K_EVENT, key, lt)
            if seen * 100 >= up * 99:
                rounds.append(tick - t_inj + 1)
            elif tick - t_inj >= 70:
                rounds.append(71)
            else:
                keep.append((t_inj, key, lt))
        inflight = keep
    drops = sim.cluster_stats()["overflow"]
    sim.close()
    r = np.array(rounds)
    return {"histogram": {int(k): int(v) for k, v in zip(*np.unique(r, return_counts=True))},
            "median": float(np.median(r)), "mean": float(r.mean()), "p90": float(np.percentile(r, 90)), "p99": float(np.percentile(r, 99)),
            "max": int(r.max()), "rumors": int(len(r)), "model_bound_drops": int(drops)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=1 << 16)
    ap.add_argument("--fanout", type=int, default=4)
    ap.add_argument("--loss", type=float, default=0.01)
    ap.add_argument("--rumors", type=int, default=1000)
    ap.add_argument("--every", type=int, default=4)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    lib = load_oracle()
    out = {"config": vars(a), "what": "rounds until >= 99 % of the nodes have applied a user event; CPU oracle, serf layer alone"}
    for name, rf in (("bijection", False), ("k_random_nodes", True)):
        t0 = time.time()
        out[name] = run(lib, a.nodes, a.fanout, a.loss, a.rumors, a.every, rf, seed=5)
        out[name]["wall_s"] = round(time.time() - t0, 1)
        print(name, {k: v for k, v in out[name].items() if k != "histogram"}, out[name]["histogram"], flush=True)
    out["shift_of_mean_rounds"] = out["k_random_nodes"]["mean"] - out["bijection"]["mean"]
    if a.out:
        json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
