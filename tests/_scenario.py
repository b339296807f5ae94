"""Seeded synthetic workloads applied identically to any implementation of the ABI."""
import numpy as np

from serf_amd import _ffi


from serf_amd.workload import apply_schedule, schedule, with_filters  # noqa: E402,F401  (moved into the package: bench.py uses it)


ARRAYS = (_ffi.ARR_ROWS, _ffi.ARR_QUEUE, _ffi.ARR_INBOX, _ffi.ARR_VIEW, _ffi.ARR_ERING, _ffi.ARR_QRING, _ffi.ARR_SLOTMAP)
ARRAY_NAMES = ("rows", "queue", "inbox", "view", "event_ring", "query_ring", "slotmap")


def first_diff(a, b):
    """Index of the first differing element of two structured arrays (or None)."""
    av, bv = a.view(np.uint8).reshape(len(a), -1), b.view(np.uint8).reshape(len(b), -1)
    bad = np.nonzero((av != bv).any(axis=1))[0]
    return None if len(bad) == 0 else int(bad[0])


def assert_same_state(x, y, what=""):
    for which, name in zip(ARRAYS, ARRAY_NAMES):
        a, b = x.dump(which), y.dump(which)
        assert a.shape == b.shape, f"{what}: {name} shape {a.shape} vs {b.shape}"
        i = first_diff(a, b)
        assert i is None, f"{what}: {name}[{i}] differs: {a[i]} vs {b[i]}"


def free_port():
    """a TCP port nobody listens on right now (the rendezvous of a multi-process test: a port derived from the pid collided now and
    then when pytest-xdist ran two such tests side by side — one of them then waited for its 240 s join)"""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]
