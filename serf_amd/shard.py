"""Sharded runs: one process per GPU, node-id range partition, ONE all-to-all per gossip round.

Shard g owns nodes [g*M, (g+1)*M).  The tick kernel writes every outgoing packet into a send
buffer laid out [destination shard][fan-out slot][M/V packets]; because the per-tick fan-out
bijection assigns each block of M/V consecutive targets of a destination to exactly one source
shard (DESIGN.md SIMSPEC §2), the exchange is a dense, equal-split
``torch.distributed.all_to_all_single`` (RCCL over xGMI on MI355X; gloo in the CPU tests) with no
packing, no counts and no index lists.  The reference has no collective at all (its transport is
UDP/TCP inside memberlist); this replaces `memberlist.send`-style delivery for the simulation.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import _ffi


class ShardedSim:
    """`Serf`-shaped facade over one shard; every rank issues the same API calls (the slot map
    and the op schedule are replicated, each shard applies the ops of the nodes it owns)."""

    def __init__(self, lib: _ffi.SimLib, n_nodes: int, device: torch.device, group=None, **kw):
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.device = device
        kw.update(vshards=self.world, shard_rank=self.rank, shard_count=self.world)
        self.sim = _ffi.Sim(lib, _ffi.make_config(n_nodes, **kw))
        nbytes = self.sim.exchange_bytes()
        # plain byte tensors: torch only provides device memory + the collective
        self.send = torch.zeros(nbytes, dtype=torch.uint8, device=device)
        self.recv = torch.zeros(nbytes, dtype=torch.uint8, device=device)
        if device.type == "cuda":
            self.sim.set_stream(torch.cuda.current_stream(device).cuda_stream)
        self.sim.bind_exchange(self.send.data_ptr(), self.recv.data_ptr())
        self.n = n_nodes
        self.m = n_nodes // self.world
        self.lo = self.rank * self.m
        self._xt = None  # exchange timing: list of (start, end) event pairs while enabled

    def owns(self, node):
        return self.lo <= node < self.lo + self.m

    # replicated API calls -------------------------------------------------------------------
    def inject(self, tick, op, node, a=0, b=0):
        self.sim.inject(tick, op, node, a, b)

    def user_event(self, node, key, encoded_len=32):
        self.sim.user_event(node, key, encoded_len)

    def query(self, node, qid, flags=0):
        self.sim.query(node, qid, flags)

    def leave(self, node):
        self.sim.leave(node)

    def join(self, node, peer=0):
        self.sim.join(node, peer)

    # the hot loop -----------------------------------------------------------------------------
    def step(self, n_ticks=1):
        for _ in range(n_ticks):
            self.sim.step(1)  # reads self.recv (packets of the previous round), fills self.send
            if self._xt is None:
                dist.all_to_all_single(self.recv, self.send, group=self.group)
            else:
                e0, e1 = self._event(), self._event()
                e0.record()
                dist.all_to_all_single(self.recv, self.send, group=self.group)
                e1.record()
                self._xt.append((e0, e1))

    # measurement: time of the collective alone (events on the stream it is enqueued on) -------------
    def _event(self):
        if self.device.type == "cuda":
            return torch.cuda.Event(enable_timing=True)

        class _T:  # CPU stand-in (gloo is synchronous)
            def record(self):
                import time
                self.t = time.perf_counter()

            def elapsed_time(self, other):
                return (other.t - self.t) * 1e3
        return _T()

    def time_exchange(self, enable):
        """enable=True: start bracketing every all-to-all with events; enable=False: stop and return the
        summed milliseconds since it was enabled."""
        if enable:
            self._xt = []
            return 0.0
        pairs, self._xt = self._xt or [], None
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        return float(sum(a.elapsed_time(b) for a, b in pairs))

    def sync(self):
        self.sim.sync()
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)

    def convergence(self, kind, key, ltime):
        seen, up = self.sim.convergence(kind, key, ltime)
        t = torch.tensor([seen, up], dtype=torch.int64, device=self.device)
        dist.all_reduce(t, group=self.group)
        return int(t[0]), int(t[1])

    def query_status(self, query_id):
        acks, resp, is_open = self.sim.query_status(query_id)   # this shard's responders
        t = torch.tensor([acks, resp], dtype=torch.int64, device=self.device)
        dist.all_reduce(t, group=self.group)
        return int(t[0]), int(t[1]), is_open

    def close(self):
        self.sim.close()
