"""Committed golden vectors (tests/golden/digests.json, made by tools/make_golden.py): the oracle must
reproduce them on CPU, and the HIP path must reproduce them on the GPU box WITHOUT the oracle in the loop."""
import json
import os

import pytest

from serf_amd import _ffi
from tests import _scenario as sc

GOLDEN = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "digests.json")))


def replay(lib, case):
    sim = _ffi.Sim(lib, _ffi.make_config(case["n"], **case["kw"]))
    sc.apply_schedule(sim, sc.schedule(case["n"], case["ticks"] // 2, rate=case["rate"], seed=case["seed"],
                                       max_member_subjects=case["subjects"]))
    for i, want in enumerate(case["digests"]):
        sim.step(case["every"])
        got = [f"{x:016x}" for x in sim.digest()]
        assert got == want, f"{case['name']}: digest {i} (tick {(i + 1) * case['every']}) differs"
    sim.close()


@pytest.mark.parametrize("case", GOLDEN["cases"], ids=[c["name"] for c in GOLDEN["cases"]])
def test_oracle_reproduces_golden(oracle, case):
    replay(oracle, case)


@pytest.mark.gpu
@pytest.mark.parametrize("case", GOLDEN["cases"], ids=[c["name"] for c in GOLDEN["cases"]])
def test_hip_reproduces_golden(hiplib, case):
    replay(hiplib, case)
