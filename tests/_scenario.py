"""Seeded synthetic workloads applied identically to any implementation of the ABI."""
import numpy as np

from serf_amd import _ffi


def schedule(n_nodes, n_ticks, rate, seed=1234, mix=(0.5, 0.15, 0.15, 0.1, 0.1), max_member_subjects=None, even=False):
    """Return a list of (tick, op, node, a, b).

    mix = fractions of (user event, query, graceful leave, force-leave of a crashed node,
    crash+revive).  `rate` = expected rumors per tick, spread uniformly over the first
    n_ticks ticks.  Member-affecting ops use distinct subjects (each needs a view slot)."""
    rng = np.random.default_rng(seed)
    n_ops = max(1, int(round(rate * n_ticks)))
    ticks = np.sort(rng.integers(0, n_ticks, n_ops))
    if even:  # evenly spaced injections: a steady load for benchmarks (the draw above keeps the stream aligned)
        ticks = (np.arange(n_ops) * (n_ticks / n_ops)).astype(np.int64)
    kinds = rng.choice(5, n_ops, p=np.array(mix) / np.sum(mix))
    used = set()
    ops = []
    budget = max_member_subjects if max_member_subjects is not None else n_nodes // 4
    key = 1
    for t, k in zip(ticks.tolist(), kinds.tolist()):
        node = int(rng.integers(0, n_nodes))
        if k >= 2 and len(used) >= budget:
            k = 0
        if k == 0:
            ops.append((t, _ffi.OP_USER_EVENT, node, key, int(rng.integers(16, 512))))
            key += 1
        elif k == 1:
            ops.append((t, _ffi.OP_QUERY, node, key, int(rng.choice([0, _ffi.F_ACK, _ffi.F_ACK | _ffi.F_RESPOND, _ffi.F_ACK | _ffi.F_RESPOND | (2 << 8)]))))
            key += 1
        else:
            while node in used:
                node = int(rng.integers(0, n_nodes))
            used.add(node)
            if k == 2:
                ops.append((t, _ffi.OP_LEAVE, node, 0, 0))          # Serf::leave: intent ...
                ops.append((t + 8, _ffi.OP_LEAVE_FINISH, node, 0, 0))  # ... memberlist.leave ...
                ops.append((t + 16, _ffi.OP_CRASH, node, 0, 0))        # ... shutdown
                if rng.random() < 0.5:
                    ops.append((t + 30, _ffi.OP_JOIN, node, 0, 0))
            elif k == 3:
                other = int(rng.integers(0, n_nodes))
                if other == node:
                    other = (node + 1) % n_nodes
                ops.append((t, _ffi.OP_CRASH, node, 0, 0))
                ops.append((t + 2, _ffi.OP_FORCE_LEAVE, other, node, int(rng.random() < 0.3)))
            else:
                ops.append((t, _ffi.OP_CRASH, node, 0, 0))
                ops.append((t + int(rng.integers(3, 20)), _ffi.OP_REVIVE, node, 0, 0))
    ops.sort(key=lambda o: o[0])
    return ops


def apply_schedule(sim, ops):
    for t, op, node, a, b in ops:
        sim.inject(t, op, node, a, b)


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
