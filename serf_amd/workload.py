"""Seeded synthetic workloads: schedules of `Serf` API operations for a simulated cluster.

The reference has no workload generator (its tests drive a handful of nodes by hand,
serf-core/src/serf/base/tests/serf/*.rs); benchmarks and parity tests of the bulk path need one that
is a pure function of its arguments so that every implementation of the C ABI sees the same calls.
"""
import numpy as np

from . import _ffi

# the benchmark mix: (user event, query, graceful leave [+ rejoin], crash + remove_failed_node, crash + revive)
BENCH_MIX = (0.55, 0.2, 0.15, 0.05, 0.05)


def _interleaved(n_ops, mix):
    """The kinds of `n_ops` operations in the proportions `mix`, spread as evenly as the proportions allow (smooth
    weighted round-robin: add the weights, take the largest, subtract the total) — every stretch of the sequence has
    the mix of the whole, which a short timed window of a benchmark needs."""
    w = np.array(mix, dtype=np.float64) / np.sum(mix)
    cur = np.zeros(len(w))
    out = np.empty(n_ops, dtype=np.int64)
    for i in range(n_ops):
        cur += w
        k = int(np.argmax(cur))
        cur[k] -= 1.0
        out[i] = k
    return out


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
    if even:  # ... and the kinds interleaved in the proportions of the mix instead of drawn (same reason)
        kinds = _interleaved(n_ops, mix)
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


def with_filters(ops, n_nodes, seed=5, p=0.6, tag_changes=0, n_ticks=None):
    """QueryParam.filters and Serf::set_tags on top of a schedule (its own random stream: the base schedule is kept).
    With probability `p` a query gets filters — a Filter::Id list of 1..12 nodes, a tag-class mask, or both — as
    SIM_OP_QUERY_FILTER_* operations right before it; `tag_changes` SIM_OP_SET_TAGS are spread over the first
    `n_ticks` ticks.  Returns (ops, classes): `classes` = the start-up tag class of every node (Sim.init_tags)."""
    rng = np.random.default_rng(seed)
    classes = rng.integers(0, 8, n_nodes).astype(np.uint8)   # class 0 = no tags
    out = []
    for o in ops:
        if o[1] == _ffi.OP_QUERY and rng.random() < p:
            t, _, node, qid, _ = o
            how = int(rng.integers(0, 3))
            if how != 1:
                for x in rng.choice(n_nodes, int(rng.integers(1, _ffi.QF_IDS + 1)), replace=False).tolist():
                    out.append((t, _ffi.OP_QUERY_FILTER_ID, node, qid, int(x)))
            if how != 0:
                out.append((t, _ffi.OP_QUERY_FILTER_TAGS, node, qid, int(rng.integers(0, 256)) & ~1))
        out.append(o)
    last = n_ticks if n_ticks is not None else (max(o[0] for o in ops) + 1 if ops else 1)
    for _ in range(tag_changes):
        out.append((int(rng.integers(0, last)), _ffi.OP_SET_TAGS, int(rng.integers(0, n_nodes)), int(rng.integers(0, 8)), 0))
    out.sort(key=lambda o: o[0])
    return out, classes


def apply_schedule(sim, ops):
    for t, op, node, a, b in ops:
        sim.inject(t, op, node, a, b)
