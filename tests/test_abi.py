"""The C-ABI library loads and exports every symbol include/serf_sim.h declares; struct layouts seen
by the bindings match the header; the product fails loudly without a GPU.  No compute calls here."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import serf_amd
from serf_amd import _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "serf_sim.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|uint32_t|const char\s*\*)\s*(sim_[a-z_0-9]+)\s*\(", src)))


def test_header_declares_what_the_binding_binds():
    assert declared_symbols() == sorted("sim_" + s for s in _ffi.ABI_SYMBOLS)


def test_hip_library_exports_every_declared_symbol():
    lib = serf_amd.load()          # raises FileNotFoundError if the .so was not built
    dll = C.CDLL(lib.path)
    for sym in declared_symbols():
        assert hasattr(dll, sym), f"{sym} missing from {lib.path}"
    assert lib.backend_name() == "hip-gfx950" and lib.abi_version() == 15


def test_oracle_exports_the_same_interface(oracle):
    dll = C.CDLL(oracle.path)
    for sym in declared_symbols():
        assert hasattr(dll, "o" + sym), f"o{sym} missing from the oracle"
    assert oracle.backend_name() == "cpu-oracle" and oracle.abi_version() == 15


def test_struct_layouts_match_the_header(tmp_path):
    prog = tmp_path / "sz.c"
    prog.write_text('#include <stdio.h>\n#include "serf_sim.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                    "sizeof(sim_config),sizeof(sim_stats),sizeof(sim_event),sizeof(sim_row),sizeof(sim_record),"
                    "sizeof(sim_view),sizeof(sim_bucket),sizeof(sim_packet));return 0;}\n")
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(prog)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    want = [C.sizeof(_ffi.Config), C.sizeof(_ffi.Stats), C.sizeof(_ffi.Event), _ffi.ROW_DTYPE.itemsize,
            _ffi.REC_DTYPE.itemsize, _ffi.VIEW_DTYPE.itemsize, _ffi.BUCKET_DTYPE.itemsize, _ffi.PACKET_DTYPE.itemsize]
    assert got == want
    assert got[3:] == [112, 16, 32, 32, 48]


def test_product_has_no_cpu_fallback(hiplib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("this check is for boxes without a GPU")
    with pytest.raises(_ffi.SimError) as ei:
        _ffi.Sim(hiplib, _ffi.make_config(64))
    assert ei.value.code == _ffi.EDEVICE


def test_bad_configs_are_rejected(oracle):
    for kw in (dict(fanout=0), dict(fanout=5), dict(vshards=3), dict(event_ring=0), dict(retransmit_mult=40)):
        with pytest.raises(_ffi.SimError) as ei:
            _ffi.Sim(oracle, _ffi.make_config(64, **kw))
        assert ei.value.code == _ffi.EINVAL
    cfg = _ffi.make_config(64)
    cfg.struct_size = 12
    with pytest.raises(_ffi.SimError):
        _ffi.Sim(oracle, cfg)


def test_api_argument_errors(oracle):
    s = _ffi.Sim(oracle, _ffi.make_config(64, view_slots=2))
    with pytest.raises(_ffi.SimError):      # api.rs:246-262: user event larger than the hard limit
        s.user_event(1, 5, encoded_len=10 * 1024)
    with pytest.raises(_ffi.SimError):
        s.user_event(1, 0)                  # key 0 is the "empty" marker
    with pytest.raises(_ffi.SimError):
        s.leave(64)                         # no such node
    s.leave(1)
    s.leave(2)
    with pytest.raises(_ffi.SimError) as ei:
        s.leave(3)                          # third active subject, two view slots
    assert ei.value.code == _ffi.ENOSLOT
    with pytest.raises(_ffi.SimError) as ei:
        s.suspect_import(0, 0, 1)           # heads == NULL = "the ones the library's exchange carried": the oracle has no exchange
    assert ei.value.code == _ffi.EINVAL
