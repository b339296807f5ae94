"""The C++ mirror of the reference's `Serf` API (serf_amd/host/serf.hpp) drives the HIP library through
the C ABI: the reference's event tests (tests/serf/event.rs:88-232) as a compiled host program."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "serf_amd", "host", "serf_example")


@pytest.mark.gpu
def test_cpp_host_example_runs_the_reference_event_scenarios():
    if not os.path.exists(EXE):
        pytest.skip("serf_example not built (python -c 'import __graft_entry__ as g; g.build()')")
    r = subprocess.run([EXE, "4096"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "reached 99%" in r.stdout
    rounds = int(r.stdout.split("after")[1].split()[0])
    assert 5 <= rounds <= 12, r.stdout  # log_4(4096) = 6 rounds of pure doubling, a few more with collisions
    assert "failed 1 left 1" in r.stdout
