"""Sharded runs: one process per GPU, node-id range partition, the all-to-all of a gossip round overlapped with compute.

Shard g owns nodes [g*M, (g+1)*M).  The bijection's tick kernel writes every outgoing packet into a send buffer laid out
[sender chunk][destination shard][fan-out slot][M/V/C packets]: the fan-out map (DESIGN.md SIMSPEC §2.3) sends the
packets of one sender chunk for one (destination, slot) to exactly one dense slab, so

* the exchange is a dense, equal-split ``torch.distributed.all_to_all_single`` (RCCL over xGMI on MI355X; gloo in
  the CPU tests) with no packing, no counts and no index lists, and
* with ``chunks = C > 1`` a tick runs as C kernel launches, and the all-to-all of chunk c (1/C of the round's bytes)
  is issued asynchronously as soon as its launch is enqueued: RCCL's stream waits for that launch only, so the
  slabs of chunk c travel while chunk c + 1 computes.  Only the last chunk's exchange is exposed.  Every exchange of
  round t has to be complete before round t + 1 reads it, and the later chunks of round t + 1 still read round t's
  packets while the first chunks of t + 1 arrive, so the receive side is double-buffered (recv[t & 1]).

memberlist's literal kRandomNodes (SIM_CF_RANDOM_FANOUT) sends a packet to ANY node, so there is no dense slab a tick kernel could
write into: the packets stay in their senders' cells, every shard sorts the (target, sender, slot) triples of its OWN senders and
packs the packets bound for shard h into slab h in that order, one count byte per target next to them (sim_exchange_layout:
XCHG_PACKED); the exchange is the same equal-split all-to-all — f * M * (V - 1) / V packets leave a GPU per round —, the receiver's
row is V sorted runs.  With ``chunks = C`` the senders are cut into C ranges, each packed and exchanged behind its own launch (a row
is then V * C runs) — the same overlap as the bijection's.

The reference has no collective at all (its transport is UDP/TCP inside memberlist); this replaces
`memberlist.send`-style delivery for the simulation.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

from . import _ffi


class ShardedSim:
    """`Serf`-shaped facade over one shard; every rank issues the same API calls (the slot map
    and the op schedule are replicated, each shard applies the ops of the nodes it owns)."""

    def __init__(self, lib: _ffi.SimLib, n_nodes: int, device: torch.device, group=None, chunks: int = 1, exchange: str = "auto", **kw):
        """exchange: who issues the round's all-to-all — "rccl": the library itself (sim_exchange_*: grouped ncclSend / ncclRecv on
        a stream of its own, ordered against the chunk launches on the device), "torch": torch.distributed.all_to_all_single,
        "auto": the library when the tensors live on a GPU, the process group is RCCL's and the library has the entry points.
        A world of ONE rank runs the same path (SIM_CF_FORCE_SHARDED): the rehearsal of the N > 1 line on one GPU."""
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.device = device
        kw.update(vshards=self.world, shard_rank=self.rank, shard_count=self.world, chunks=chunks if chunks > 1 else 0)
        if self.world == 1:
            kw["force_sharded"] = True
        self.lib = lib
        self.sim = _ffi.Sim(lib, _ffi.make_config(n_nodes, **kw))
        nbytes = self.sim.exchange_bytes()
        self.chunks, self.chunk_bytes = self.sim.exchange_chunks()
        # what the round's exchange is (include/serf_sim.h sim_exchange_layout): always an equal-split all-to-all — of the slabs
        # the bijection's tick kernel writes, or (memberlist's kRandomNodes: XCHG_PACKED) of the slabs the library packs, behind the
        # tick's launch, from the packets the senders keep
        self.kind, self.planes, self.plane_bytes, recv_bytes = self.sim.exchange_layout()
        assert self.kind in (_ffi.XCHG_ALL_TO_ALL, _ffi.XCHG_PACKED) and self.planes == 1 and recv_bytes == nbytes
        # plain byte tensors: torch only provides device memory + the collective
        self.send = torch.zeros(nbytes, dtype=torch.uint8, device=device)
        self.recv = [torch.zeros(recv_bytes, dtype=torch.uint8, device=device) for _ in range(2 if self.chunks > 1 else 1)]
        if device.type == "cuda":
            self.sim.set_stream(torch.cuda.current_stream(device).cuda_stream)
        self.sim.bind_exchange3(self.send.data_ptr(), nbytes, self.recv[0].data_ptr(), self.recv[-1].data_ptr(), recv_bytes)
        backend = dist.get_backend(group)
        self.use_lib = exchange == "rccl" or (exchange == "auto" and device.type == "cuda" and backend == "nccl" and lib.exchange_library() is not None)
        if self.use_lib:
            # the communicator's id: made by rank 0 (ncclGetUniqueId through the library), handed round as plain bytes
            idt = torch.zeros(_ffi.EXCHANGE_ID_BYTES, dtype=torch.uint8)
            if self.rank == 0:
                idt = torch.frombuffer(bytearray(lib.exchange_unique_id()), dtype=torch.uint8).clone()
            if backend == "nccl":
                idt = idt.to(device)
            dist.broadcast(idt, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            self.sim.exchange_init(bytes(idt.cpu().numpy().tobytes()), self.rank, self.world)
        self.n = n_nodes
        self.m = n_nodes // self.world
        self.lo = self.rank * self.m
        # slot-less failed probes (SWIM layer on): every tick's lists are gathered on the device and read two ticks later
        self._poll_suspects = bool(kw.get("probe_interval", 0))
        # (r6) the library's own exchange carries the lists' heads to every peer and on into pinned host memory (sim_exchange_chunk;
        # sim_suspect_import(h, t, NULL, world) reads them): no second collective, no side stream, no copies between two ticks
        self._lib_heads = self.use_lib and self._poll_suspects and os.environ.get("SERF_SIM_XHEADS", "1") != "0"  # (= 0: the host's own hand-over below)
        if self._lib_heads:
            self._sq_inflight = []
        elif self._poll_suspects:
            w = _ffi.SREQ_HEAD_WORDS
            self._sq_send = [torch.zeros(w, dtype=torch.int32, device=device) for _ in range(4)]
            self._sq_gath = [torch.zeros(w * self.world, dtype=torch.int32, device=device) for _ in range(4)] if device.type == "cuda" else None
            self._sq_host = [torch.zeros(w * self.world, dtype=torch.int32) for _ in range(4)]
            if device.type == "cuda":
                self._sq_host = [x.pin_memory() for x in self._sq_host]
                self._sq_stream = torch.cuda.Stream(device)  # (r6, tried: at the highest priority — own hardware queue: kRandomNodes + 1.4 %, the bijection's chunk-wise exchange - 18 %)
            self._sq_inflight = []
        self._pending = []  # async all-to-alls of the round in flight
        self._xt = None     # exchange timing: list of (start, end) event pairs while enabled

    def owns(self, node):
        return self.lo <= node < self.lo + self.m

    # replicated API calls -------------------------------------------------------------------
    def inject(self, tick, op, node, a=0, b=0):
        self.sim.inject(tick, op, node, a, b)

    def user_event(self, node, key, encoded_len=32):
        self.sim.user_event(node, key, encoded_len)

    def query(self, node, qid, flags=0, ids=None, tag_mask=_ffi.NO_TAG_FILTER):
        self.sim.query(node, qid, flags, ids, tag_mask)

    def set_tags(self, node, tag_class):
        self.sim.set_tags(node, tag_class)

    def init_tags(self, classes, first=0):
        self.sim.init_tags(classes, first)

    def leave(self, node):
        self.sim.leave(node)

    def join(self, node, peer=0):
        self.sim.join(node, peer)

    # the hot loop -----------------------------------------------------------------------------
    def _drain(self):
        """Every exchange of the previous round has landed (and the compute stream knows it)."""
        if self.use_lib:
            self.sim.exchange_wait()  # on the device: the handle's stream waits for the exchange stream, the host does not
            return
        for w in self._pending:
            w.wait()
        self._pending = []

    def collective_library(self):
        """what moves the round's packets: "RCCL x.y.z (issued by the library: grouped ncclSend / ncclRecv)" or torch's backend"""
        if self.use_lib:
            return self.lib.exchange_library() + " — issued by libserf_sim (sim_exchange_chunk: grouped ncclSend / ncclRecv per peer)"
        backend = dist.get_backend(self.group)
        if backend == "nccl" and self.device.type == "cuda":
            return "RCCL " + ".".join(str(x) for x in torch.cuda.nccl.version()) + " — torch.distributed.all_to_all_single"
        return backend

    def snapshot(self):
        """sim_snapshot of this shard.  The lists of slot-less suspicions that are still travelling (all-gathers of the last two
        ticks) are waited for and imported first: they live in the host's buffers, not in the handle, and would be lost."""
        self._drain()
        if self._poll_suspects:
            while self._sq_inflight:
                self._suspicions_take()
        return self.sim.snapshot()

    def _recycle(self):
        """View-slot recycling needs every shard's verdict (include/serf_sim.h): scan locally, all-gather, keep the
        candidates no shard objects to and whose running nodes agree everywhere, apply the same list on every shard."""
        import numpy as np

        mine = self.sim.recycle_scan()  # [n, 12] uint32 words of sim_recycle_cand; n is the same on every shard
        n = len(mine)
        if n == 0:
            self.sim.recycle_apply(mine)
            return
        t = torch.from_numpy(mine.astype(np.int64)).to(self.device)
        allv = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(allv, t, group=self.group)
        a = torch.stack(allv).cpu().numpy().astype(np.uint32)  # [world, n, 12]
        keep = []
        for i in range(n):
            flags = a[:, i, 2]
            if (flags & 1).any() or not (flags & 2).any():
                continue
            refs = a[(flags & 2) != 0, i, 4:8]  # ltime.lo, ltime.hi, inc, bits of the shards that have running nodes
            if (refs == refs[0]).all():
                keep.append(a[np.nonzero(flags & 2)[0][0], i])
        self.sim.recycle_apply(np.array(keep, dtype=np.uint32).reshape(-1, 12))

    def _push_pull(self):
        """A push-pull batch whose pairs span shards (include/serf_sim.h): two rounds of pack -> all-to-all-v -> merge."""
        snd, rcv, rb = self.sim.pp_plan(self.world)
        for rnd in (1, 2):
            out_counts, in_counts = (snd, rcv) if rnd == 1 else (rcv, snd)
            send = torch.empty(max(1, sum(out_counts) * rb), dtype=torch.uint8, device=self.device)
            recv = torch.empty(max(1, sum(in_counts) * rb), dtype=torch.uint8, device=self.device)
            self.sim.pp_export(rnd, send.data_ptr())
            if self.device.type == "cuda":
                self.sim.sync()
            dist.all_to_all_single(recv[:sum(in_counts) * rb], send[:sum(out_counts) * rb],
                                   output_split_sizes=[c * rb for c in in_counts], input_split_sizes=[c * rb for c in out_counts],
                                   group=self.group)
            self.sim.pp_merge(rnd, recv.data_ptr())
        if self.device.type == "cuda":
            self.sim.sync()  # the buffers go away with this frame

    def _suspicions_out(self):
        """Probes that failed on a target without a view slot (include/serf_sim.h sim_suspect_export / _import): the head
        of this shard's list of the tick just ended goes into an all-gather on the device, the gathered heads follow into
        pinned host memory, and nobody waits: the result is read two ticks later (`_suspicions_in`)."""
        t = self.sim.tick - 1
        i = t % len(self._sq_send)
        self.sim.suspect_export(self._sq_send[i].data_ptr())
        if self.device.type == "cuda":
            # on a side stream: the compute stream never waits for this collective, only the side stream waits for the export
            self._sq_stream.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self._sq_stream):
                dist.all_gather_into_tensor(self._sq_gath[i], self._sq_send[i], group=self.group, async_op=True).wait()
                self._sq_host[i].copy_(self._sq_gath[i], non_blocking=True)
                done = torch.cuda.Event()
                done.record(self._sq_stream)
        else:
            done = dist.all_gather_into_tensor(self._sq_host[i], self._sq_send[i], group=self.group, async_op=True)
        self._sq_inflight.append((t, i, done))

    def _suspicions_in(self):
        """Before tick T begins: the gathered lists of tick T - 2 become SIM_OP_SUSPECT operations of tick T on EVERY shard
        (the schedule and the slot map are replicated) — the rule of a single-process run."""
        while self._sq_inflight and self._sq_inflight[0][0] + 2 <= self.sim.tick:
            self._suspicions_take()

    def _suspicions_take(self):
        t, i, done = self._sq_inflight.pop(0)
        if self._lib_heads:
            self.sim.suspect_import(t, 0, self.world)  # waits for the event behind the exchange of tick t (issued two ticks ago)
            return
        done.synchronize() if self.device.type == "cuda" else done.wait()  # issued two ticks ago
        self.sim.suspect_import(t, self._sq_host[i].data_ptr(), self.world)

    def step(self, n_ticks=1):
        for _ in range(n_ticks):
            self._drain()
            if self._poll_suspects:
                self._suspicions_in()
            if self.sim.recycle_due():
                self._recycle()
            rbuf = self.recv[self.sim.tick & 1] if self.chunks > 1 else self.recv[0]  # packets sent during tick t land in recv[t & 1]
            self.sim.step_begin()  # the tick's operations
            if self.sim.pp_due():
                self._push_pull()
            if self.chunks == 1:
                self.sim.step_chunk(0)  # reads recv (packets of the previous round), fills send
                self.sim.step_end()
                self._exchange(0, self.recv[0], self.send, False)
            else:
                for c in range(self.chunks):
                    self.sim.step_chunk(c)
                    lo = c * self.chunk_bytes
                    self._exchange(c, rbuf[lo:lo + self.chunk_bytes], self.send[lo:lo + self.chunk_bytes], True)
                self.sim.step_end()
            if self._lib_heads:
                self._sq_inflight.append((self.sim.tick - 1, None, None))  # (travelled with the last chunk's exchange)
            elif self._poll_suspects:
                self._suspicions_out()

    def restore(self, image):
        """sim_restore of this shard's image (collective: every rank restores its own).  With the random fan-out the image holds the
        shard's OWN cells (the packets in flight in their senders' cells); the library has packed them into the send buffer
        again (XCHG_PACKED) and the round's exchange is run once more before the next tick."""
        self._drain()
        self.sim.restore(image)
        if self.kind == _ffi.XCHG_PACKED and self.sim.tick > 0:
            rbuf = self.recv[(self.sim.tick - 1) & 1] if self.chunks > 1 else self.recv[0]
            for c in range(self.chunks):
                lo = c * self.chunk_bytes
                self._exchange(c, rbuf[lo:lo + self.chunk_bytes], self.send[lo:lo + self.chunk_bytes], False)

    def _exchange(self, c, recv, send, asynchronous):
        if self._xt is not None:  # measurement mode: bracket the collective with events, no overlap
            e0, e1 = self._event(), self._event()
            e0.record()
            if self.use_lib:
                self.sim.exchange_chunk(c)
                self.sim.exchange_wait()
            else:
                dist.all_to_all_single(recv, send, group=self.group)
            e1.record()
            self._xt.append((e0, e1))
        elif self.use_lib:
            self.sim.exchange_chunk(c)  # the library orders it behind chunk c's launch and ahead of the next tick, on the device
        elif asynchronous:
            self._pending.append(dist.all_to_all_single(recv, send, group=self.group, async_op=True))
        else:
            dist.all_to_all_single(recv, send, group=self.group)

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
        """enable=True: start bracketing every all-to-all with events (the exchanges then run synchronously, one
        after the other: this measures the collective, not the overlap); enable=False: stop and return the summed
        milliseconds since it was enabled."""
        self._drain()
        if enable:
            self._xt = []
            return 0.0
        pairs, self._xt = self._xt or [], None
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        return float(sum(a.elapsed_time(b) for a, b in pairs))

    def sync(self):
        self._drain()
        self.sim.sync()
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)

    def convergence(self, kind, key, ltime):
        self._drain()
        seen, up = self.sim.convergence(kind, key, ltime)
        t = torch.tensor([seen, up], dtype=torch.int64, device=self.device)
        dist.all_reduce(t, group=self.group)
        return int(t[0]), int(t[1])

    def convergence_many(self, rumours):
        self._drain()
        seen, up = self.sim.convergence_many(rumours)
        t = torch.tensor(list(seen) + [up], dtype=torch.int64, device=self.device)
        dist.all_reduce(t, group=self.group)
        return [int(x) for x in t[:-1]], int(t[-1])

    def query_status(self, query_id):
        self._drain()
        acks, resp, is_open = self.sim.query_status(query_id)   # this shard's responders
        t = torch.tensor([acks, resp], dtype=torch.int64, device=self.device)
        dist.all_reduce(t, group=self.group)
        return int(t[0]), int(t[1]), is_open

    def close(self):
        self._drain()
        for _, _, done in getattr(self, "_sq_inflight", []):  # gathers still writing into our buffers
            if done is None:
                continue  # (heads that travelled with the library's exchange: its buffers go with the handle)
            done.synchronize() if self.device.type == "cuda" else done.wait()
        self.sim.close()
