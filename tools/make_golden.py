#!/usr/bin/env python
"""Generate tests/golden/digests.json: state digests of the CPU oracle on fixed seeded scenarios.

The reference cannot run here (no Rust toolchain, memberlist-core not vendored), so golden vectors cannot
come from the reference itself; the reference-derived known answers are the KATs in
tests/test_oracle_kat.py.  These digests freeze the oracle's behaviour on whole-cluster runs so that
(a) the oracle cannot drift silently and (b) the HIP path can be checked on the GPU box without the oracle
in the loop.  Re-run only when SIMSPEC changes on purpose.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from serf_amd import _ffi  # noqa: E402
from tests import _scenario as sc  # noqa: E402
from tests._oracle import load_oracle  # noqa: E402

CASES = [
    dict(name="cfg1_128_f3_dense", n=128, ticks=96, every=16, rate=0.6, seed=101, subjects=40,
         kw=dict(fanout=3, view_slots=0, event_ring=16, query_ring=8, leave_delay=6, probe_interval=5,
                 reap_interval=10, reconnect_timeout=30, tombstone_timeout=50, intent_timeout=20)),
    dict(name="ragged_257_f4_slots", n=257, ticks=96, every=16, rate=0.6, seed=102, subjects=40,
         kw=dict(fanout=4, view_slots=64, event_ring=16, query_ring=8, leave_delay=6, probe_interval=3, loss=0.05,
                 push_pull_interval=5)),
    dict(name="serf_only_1024_f4", n=1024, ticks=64, every=16, rate=1.0, seed=103, subjects=60,
         kw=dict(fanout=4, view_slots=64, event_ring=32, query_ring=32, leave_delay=6, probe_interval=0)),
    dict(name="cfg2_64k_f3", n=65536, ticks=64, every=32, rate=0.5, seed=104, subjects=100,
         kw=dict(fanout=3, view_slots=128, event_ring=64, query_ring=64, probe_interval=5)),
    dict(name="vshards4_2048_f4", n=2048, ticks=48, every=16, rate=0.8, seed=105, subjects=60,
         kw=dict(fanout=4, vshards=4, view_slots=96, event_ring=16, query_ring=8, leave_delay=6, probe_interval=4, loss=0.02)),
]


def run_case(lib, case):
    sim = _ffi.Sim(lib, _ffi.make_config(case["n"], **case["kw"]))
    sc.apply_schedule(sim, sc.schedule(case["n"], case["ticks"] // 2, rate=case["rate"], seed=case["seed"],
                                       max_member_subjects=case["subjects"]))
    out = []
    for t in range(0, case["ticks"], case["every"]):
        sim.step(case["every"])
        out.append([f"{x:016x}" for x in sim.digest()])
    sim.close()
    return out


if __name__ == "__main__":
    lib = load_oracle()
    doc = {"spec": f"DESIGN.md SIMSPEC (ABI {lib.abi_version()})", "cases": []}
    for case in CASES:
        doc["cases"].append(dict(case, digests=run_case(lib, case)))
    path = os.path.join(ROOT, "tests", "golden", "digests.json")
    json.dump(doc, open(path, "w"), indent=1)
    print("wrote", path)
