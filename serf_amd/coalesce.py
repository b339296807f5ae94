"""Event coalescing for the host-side event stream (serf-core/src/coalesce.rs, coalesce/member.rs,
coalesce/user.rs), in simulation ticks instead of wall-clock time.

The stream is what `Sim.drain_events()` returns for watched observers: tuples
``(tick, observer, type, key, ltime)`` with ``type`` = MemberEventType (Join 0, Leave 1, Failed 2,
Update 3, Reap 4), 5 = user event, 6 = query.  A coalescer batches the events of ONE observer.

`coalesce_loop` follows `coalesce.rs:66-155`: an event the coalescer handles starts a quantum
(`coalesce_period`) if none is running and restarts the quiescence timer (`quiescent_period`); when either
expires everything coalesced so far is flushed; events it does not handle pass straight through.
"""
from __future__ import annotations

JOIN, LEAVE, FAILED, UPDATE, REAP, USER, QUERY = range(7)


class MemberEventCoalescer:
    """coalesce/member.rs:25-128: the latest event per member wins inside a window; a member whose
    latest event type equals the one reported last time is dropped, except for Update."""

    name = "member_event_coalescer"

    def __init__(self):
        self.last_events = {}     # member -> type reported by the previous flushes
        self.latest_events = {}   # member -> type seen in the current window (insertion ordered)

    def handle(self, ev):         # member.rs:52-54
        return ev[2] <= REAP

    def coalesce(self, ev):       # member.rs:56-72
        self.latest_events.pop(ev[3], None)
        self.latest_events[ev[3]] = ev[2]

    def flush(self, tick, observer):  # member.rs:74-110 -> one batch per event type: (tick, observer, type, [members])
        batches = {}
        for member, ty in self.latest_events.items():
            if self.last_events.get(member) == ty and ty != UPDATE:
                continue
            self.last_events[member] = ty
            batches.setdefault(ty, []).append(member)
        self.latest_events = {}
        return [(tick, observer, ty, members) for ty, members in batches.items()]


class UserEventCoalescer:
    """coalesce/user.rs:17-104: per event NAME only the events with the highest Lamport time survive a
    window.  The simulator identifies a user event by one 32-bit key for (name, payload); `name_of`
    maps a key to its name (default: the upper 24 bits), `is_cc` says whether an event asked to be
    coalesced (UserEventMessage.cc; default: all do)."""

    name = "user_event_coalescer"

    def __init__(self, name_of=lambda key: key >> 8, is_cc=lambda ev: True):
        self.name_of, self.is_cc = name_of, is_cc
        self.events = {}          # name -> (ltime, [events])

    def handle(self, ev):         # user.rs:45-50
        return ev[2] == USER and self.is_cc(ev)

    def coalesce(self, ev):       # user.rs:52-83
        name, ltime = self.name_of(ev[3]), ev[4]
        cur = self.events.get(name)
        if cur is None or cur[0] < ltime:
            self.events[name] = (ltime, [ev])
        elif cur[0] == ltime:
            cur[1].append(ev)

    def flush(self, tick, observer):  # user.rs:85-103
        out = [e for _, evs in self.events.values() for e in evs]
        self.events = {}
        return out


def coalesce_loop(events, coalescer, coalesce_period, quiescent_period, observer=None, end_tick=None):
    """Run `events` (one observer's stream, tick-ordered) through `coalescer`; returns the output stream.
    Flush ticks: quantum start + coalesce_period, or last handled event + quiescent_period (coalesce.rs:66-155)."""
    out = []
    quantum = quiescent = None

    def expire(upto):
        nonlocal quantum, quiescent
        while quantum is not None:
            due = min(quantum, quiescent)
            if upto is not None and due > upto:
                return
            out.extend(coalescer.flush(due, observer))
            quantum = quiescent = None

    for ev in events:
        if observer is None:
            observer = ev[1]
        expire(ev[0])
        if not coalescer.handle(ev):
            out.append(ev)
            continue
        if quantum is None:
            quantum = ev[0] + coalesce_period
        quiescent = ev[0] + quiescent_period
        coalescer.coalesce(ev)
    expire(end_tick)
    return out
