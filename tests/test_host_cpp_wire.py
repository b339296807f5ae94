"""Host logic in C++ (serf_amd/host/wire.hpp, serf.hpp): the codec against serf_amd/wire.py byte for byte, its round
trips, and `Serf::user_event(name, payload, coalesce)` with the reference's size checks (api.rs:241-299).  No GPU: the
test program is compiled with every sim_* entry point renamed to the CPU oracle's osim_* — same C ABI, the oracle
stands in as the backend of a host-logic test."""
import os
import subprocess

from serf_amd import _ffi, wire

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_codec_matches_python_codec_and_user_event_api(oracle, tmp_path):
    exe = tmp_path / "wire_test"
    renames = [f"-Dsim_{s}=osim_{s}" for s in _ffi.ABI_SYMBOLS]
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), *renames, "-o", str(exe),
                           os.path.join(ROOT, "tests", "cpp", "wire_test.cpp"), "-L", os.path.join(ROOT, "oracle"), "-loracle",
                           f"-Wl,-rpath,{os.path.join(ROOT, 'oracle')}"])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    got = dict(line.split(" ", 1) for line in r.stdout.strip().splitlines() if " " in line)
    q_full = wire.Query(9, 0xDEADBEEF, 4242, 3, 200, 16000, b"ping", b"x")
    q_filt = wire.Query(9, 0xDEADBEEF, 4242, 3, 0, 16000, b"", b"", [b"f1", b"filter-two"])
    ev300 = wire.UserEvent(5, b"deploy", bytes([0xAB]) * 300, False)
    want = {
        "join_1_0": wire.encode_message(wire.Join(1, 0)),
        "join_300_1048575": wire.encode_message(wire.Join(300, 1048575)),
        "leave_big_77": wire.encode_message(wire.Leave(12345678901, 77)),
        "leave_big_77_prune": wire.encode_message(wire.Leave(12345678901, 77, True)),
        "event_empty": wire.encode_message(wire.UserEvent(5)),
        "event_deploy_cc": wire.encode_message(wire.UserEvent(5, b"deploy", b"v1.2.3", True)),
        "event_deploy_300": wire.encode_message(ev300),
        "query_full": wire.encode_message(q_full),
        "query_filters": wire.encode_message(q_filt),
    }
    for name, enc in want.items():
        assert got[name] == enc.hex(), name
    assert int(got["len_event_deploy_300"]) == wire.encoded_len(ev300) == len(want["event_deploy_300"])
    assert int(got["len_query_filters"]) == wire.encoded_len(q_filt)
    assert int(got["key_deploy"]) not in (0,)
    assert 3 <= int(got["event_rounds"]) <= 12
    assert r.stdout.strip().endswith("ok")
