"""The fan-out map read from both ends (DESIGN.md SIMSPEC §2.3): the product keeps every packet at its SENDER and lets
the receiver of fan-out slot k fetch it, so it needs the map's inverse.  `sim_t_fanmap` (host arithmetic of the HIP
library, no device) evaluates the general forms the support kernels use: the targets must be the oracle's
(`fan_target`, oracle/serf_oracle.c), and the sources must be their inverse, slot by slot."""
import ctypes as C

import numpy as np
import pytest

import serf_amd
from serf_amd import _ffi

CASES = [  # (nodes, vshards, chunks, fanout)
    (1, 1, 0, 3), (2, 1, 0, 3), (3, 1, 0, 4), (100, 1, 0, 3), (128, 1, 0, 3), (777, 1, 0, 4),
    (512, 1, 0, 4), (1024, 1, 0, 4), (4096, 1, 0, 3), (2048, 4, 0, 4), (2048, 4, 2, 4), (8192, 4, 4, 4),
    (16384, 2, 2, 3), (4096, 4, 1, 1), (640, 2, 0, 2),
]


@pytest.mark.parametrize("n,v,c,f", CASES)
def test_sources_invert_targets_and_targets_are_the_oracles(oracle, n, v, c, f):
    hip = C.CDLL(serf_amd.load().path)
    cfg = _ffi.make_config(n, fanout=f, vshards=v, chunks=c, view_slots=8, event_ring=8, query_ring=8)
    o = _ffi.Sim(oracle, cfg)
    tg, sr, ot = (C.c_uint32 * 4)(), (C.c_uint32 * 4)(), (C.c_uint32 * 4)()
    for tick in (0, 1, 7, 123456789):
        targets = np.full((n, 4), -1, dtype=np.int64)
        sources = np.full((n, 4), -1, dtype=np.int64)
        feff = None
        for gid in range(n):
            k = hip.sim_t_fanmap(C.byref(cfg), C.c_uint64(tick), gid, tg, sr)
            ko = oracle.dll.osim_t_targets(o.h, C.c_uint64(tick), gid, ot)
            assert k == ko and k >= 0
            feff = k
            assert list(tg[:k]) == list(ot[:k]), (tick, gid)
            targets[gid, :k] = tg[:k]
            sources[gid, :k] = sr[:k]
        for k in range(feff):
            t = targets[:, k]
            assert sorted(t.tolist()) == list(range(n))      # every node receives exactly one packet per slot
            assert (t != np.arange(n)).all()                 # never from itself
            assert (sources[t, k] == np.arange(n)).all()     # the receiver finds its sender
    o.close()
