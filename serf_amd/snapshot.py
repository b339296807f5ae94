"""Reference-format snapshot of one simulated node — host side, over the drained event log.

serf-core's `Snapshotter` (serf-core/src/snapshot.rs) is an event-stream consumer: every event the node's `Serf`
delivers passes through it on its way to the user and leaves a record in an append-only file, from which a restarted
process recovers whom it knew to be alive and where its three Lamport clocks stood.  This module produces that
record stream — byte for byte in the reference's framing — for a WATCHED node of the simulation (`sim_watch` +
`sim_drain_events`), and restates `open_and_replay_snapshot` as the checker.

Record stream (snapshot.rs:117-126, 160-215): one type byte, then
    0 Alive     u32-LE length + encoded Node      (a Join member event; snapshot.rs:682-691)
    1 NotAlive  u32-LE length + encoded Node      (a Leave or Failed member event; :692-698)
    2 Clock, 3 EventClock, 4 QueryClock   u64-LE Lamport time
    5 Coordinate, 7 Comment               nothing  (off the simulated path)
    6 Leave                               nothing  (written by `leave()` unless rejoin_after_leave; :562-580)
Rules restated: a user event / query appends EventClock / QueryClock only when its time is newer than the last one
recorded (:659-679); after every member event, and periodically, `update_clock` appends Clock(clock.time() - 1) when
that is newer (:706-713); `compact` rewrites the file as the live nodes followed by the three clocks (:780-830).
Replay (:228-347): Alive inserts, NotAlive removes, the clock records overwrite, Leave clears everything unless
`rejoin_after_leave`, unknown record types are an error.

The canonical whole-cluster image (`sim_snapshot`) stays what oracle and HIP hand over to each other; this is the
per-node file a real `serf` process could be started from.  `Node` travels in serf_amd.wire's form (memberlist-proto is
not vendored: UPSTREAM-RECALL there).
"""
from __future__ import annotations

import struct

from . import _ffi, wire

ALIVE, NOT_ALIVE, CLOCK, EVENT_CLOCK, QUERY_CLOCK, COORDINATE, LEAVE, COMMENT = range(8)
# enum sim_event_type
EV_JOIN, EV_LEAVE, EV_FAILED, EV_UPDATE, EV_REAP, EV_USER, EV_QUERY = range(7)


def _node_record(kind: int, gid: int) -> bytes:
    node = wire.encode_node(gid)
    return bytes([kind]) + struct.pack("<I", len(node)) + node


def _clock_record(kind: int, t: int) -> bytes:
    return bytes([kind]) + struct.pack("<Q", t)


class Snapshotter:
    """`Snapshot::stream` (snapshot.rs:585-655) over simulator events of one observer."""

    def __init__(self, observer: int, rejoin_after_leave: bool = False, replay: "ReplayResult | None" = None):
        self.observer = observer
        self.rejoin_after_leave = rejoin_after_leave
        self.alive = set(replay.alive_nodes) if replay else set()
        self.last_clock = replay.last_clock if replay else 0
        self.last_event_clock = replay.last_event_clock if replay else 0
        self.last_query_clock = replay.last_query_clock if replay else 0
        self.buf = bytearray()
        self.left = False

    # ---- the event kinds (snapshot.rs:659-713) ----
    def user_event(self, ltime: int):
        if self.left or ltime <= self.last_event_clock:  # "stop recording events after a leave is issued" (:402)
            return
        self.last_event_clock = ltime
        self.buf += _clock_record(EVENT_CLOCK, ltime)

    def query(self, ltime: int):
        if self.left or ltime <= self.last_query_clock:
            return
        self.last_query_clock = ltime
        self.buf += _clock_record(QUERY_CLOCK, ltime)

    def member_event(self, ty: int, subject: int, clock_time: int):
        if self.left:
            return
        if ty == EV_JOIN:
            self.alive.add(subject)
            self.buf += _node_record(ALIVE, subject)
        elif ty in (EV_LEAVE, EV_FAILED):
            self.alive.discard(subject)
            self.buf += _node_record(NOT_ALIVE, subject)
        self.update_clock(clock_time)

    def update_clock(self, clock_time: int):
        if self.left:
            return
        last_seen = max(0, clock_time - 1)
        if last_seen > self.last_clock:
            self.last_clock = last_seen
            self.buf += _clock_record(CLOCK, last_seen)

    def leave(self):
        """snapshot.rs:562-580 (handle_leave): the Leave record is ALWAYS appended; only forgetting the live nodes
        depends on `rejoin_after_leave` ("if we plan to re-join, keep our state").  Nothing is recorded afterwards."""
        self.left = True
        if not self.rejoin_after_leave:
            self.alive.clear()
        self.buf += bytes([LEAVE])

    def feed(self, events, clock_time: int):
        """Events of sim_drain_events — (tick, observer, type, key, ltime) — of this observer, in order; `clock_time`
        is the observer's membership clock (`Stats.member_time`) when they are processed."""
        for _tick, obs, ty, key, ltime in events:
            if obs != self.observer or self.left:
                continue
            if ty == EV_USER:
                self.user_event(ltime)
            elif ty == EV_QUERY:
                self.query(ltime)
            elif ty in (EV_JOIN, EV_LEAVE, EV_FAILED):
                self.member_event(ty, key, clock_time)
        self.update_clock(clock_time)

    def compact(self) -> bytes:
        """snapshot.rs:780-830: the live nodes, then the three clocks."""
        out = bytearray()
        for gid in sorted(self.alive):
            out += _node_record(ALIVE, gid)
        out += _clock_record(CLOCK, self.last_clock) + _clock_record(EVENT_CLOCK, self.last_event_clock) + _clock_record(QUERY_CLOCK, self.last_query_clock)
        self.buf = out
        return bytes(out)

    def bytes(self) -> bytes:
        return bytes(self.buf)


class ReplayResult:
    def __init__(self, alive_nodes, last_clock, last_event_clock, last_query_clock, offset):
        self.alive_nodes, self.last_clock, self.last_event_clock, self.last_query_clock, self.offset = \
            alive_nodes, last_clock, last_event_clock, last_query_clock, offset


def replay(data: bytes, rejoin_after_leave: bool = False) -> ReplayResult:
    """open_and_replay_snapshot (snapshot.rs:228-347)."""
    alive, last_clock, last_event_clock, last_query_clock = set(), 0, 0, 0
    off = 0
    while off < len(data):
        kind = data[off]
        off += 1
        if kind in (ALIVE, NOT_ALIVE):
            (n,) = struct.unpack_from("<I", data, off)
            off += 4
            if off + n > len(data):
                raise ValueError("failed to replay snapshot: truncated node record")
            gid = wire.decode_node(data[off:off + n])
            off += n
            (alive.add if kind == ALIVE else alive.discard)(gid)
        elif kind in (CLOCK, EVENT_CLOCK, QUERY_CLOCK):
            if off + 8 > len(data):
                raise ValueError("failed to replay snapshot: truncated clock record")
            (t,) = struct.unpack_from("<Q", data, off)
            off += 8
            if kind == CLOCK:
                last_clock = t
            elif kind == EVENT_CLOCK:
                last_event_clock = t
            else:
                last_query_clock = t
        elif kind in (COORDINATE, COMMENT):
            continue
        elif kind == LEAVE:
            if rejoin_after_leave:  # "ignoring previous leave in snapshot"
                continue
            alive.clear()
            last_clock = last_event_clock = last_query_clock = 0
        else:
            raise ValueError(f"unrecognized snapshot record type: {kind}")
    return ReplayResult(alive, last_clock, last_event_clock, last_query_clock, len(data))


def snapshot_of(sim: "_ffi.Sim", observer: int, events, rejoin_after_leave: bool = False) -> Snapshotter:
    """The snapshotter a watched node would have written over `events` (its drained log), clocks read from the node."""
    s = Snapshotter(observer, rejoin_after_leave)
    s.feed(events, sim.stats(observer).member_time)
    return s
