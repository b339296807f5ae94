"""ctypes binding of the C ABI declared in include/serf_sim.h.

The binding is generic over (shared-library path, symbol prefix): the product library
``serf_amd/csrc/libserf_sim.so`` exports ``sim_*``; a test may bind any other implementation of
the same ABI (the CPU oracle exports ``osim_*``) by passing its own path — nothing in this package
knows where such a library lives.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

P, Q, CKEYS, MAX_FANOUT = 4, 64, 6, 4
Q_HOT = 16  # SIM_Q_HOT: the entries the HIP tick kernel keeps in registers
DEFAULT_SEED = 0x5EEDC0DE5E4F0001

OK, EINVAL, ENOMEM, EDEVICE, ENOSLOT, ESTATE, ERANGE, ETOOBIG = 0, -1, -2, -3, -4, -5, -6, -7
_ERR = {EINVAL: "SIM_EINVAL", ENOMEM: "SIM_ENOMEM", EDEVICE: "SIM_EDEVICE", ENOSLOT: "SIM_ENOSLOT",
        ESTATE: "SIM_ESTATE", ERANGE: "SIM_ERANGE", ETOOBIG: "SIM_ETOOBIG"}

# enum sim_member_status (types/member.rs:54-87)
STATUS_NONE, STATUS_ALIVE, STATUS_LEAVING, STATUS_LEFT, STATUS_FAILED = 0, 1, 2, 3, 4
# enum sim_kind
K_JOIN, K_LEAVE, K_EVENT, K_QUERY, K_ALIVE, K_SUSPECT, K_DEAD = 1, 2, 3, 4, 5, 6, 7
# enum sim_event_type (event.rs:263-279, 367-378)
EV_JOIN, EV_LEAVE, EV_FAILED, EV_UPDATE, EV_REAP, EV_USER, EV_QUERY = range(7)
# enum sim_op
OP_USER_EVENT, OP_QUERY, OP_LEAVE, OP_JOIN, OP_FORCE_LEAVE, OP_CRASH, OP_REVIVE, OP_LEAVE_FINISH = 1, 2, 3, 4, 5, 6, 7, 8
OP_SET_TAGS, OP_QUERY_FILTER_ID, OP_QUERY_FILTER_TAGS, OP_DELIVER, OP_SUSPECT, OP_RECONNECT = 9, 10, 11, 12, 13, 14
OP_PRUNE = 17   # internal: the end of handle_prune's wait (CF_PRUNE_DELAY), scheduled by the library from the tick's request list
OP_QRESP, OP_WITNESS = 15, 16   # internal: scheduled by sim_deliver_message (a QueryResponse / a PushPull's clocks), refused by sim_inject
SUSPECT_REQ_MAX, SREQ_HEAD_WORDS = 4096, 8193
QF_IDS, TAG_CLASSES, NO_TAG_FILTER = 12, 32, 0xFFFFFFFF
# enum sim_array
ARR_ROWS, ARR_QUEUE, ARR_INBOX, ARR_VIEW, ARR_ERING, ARR_QRING, ARR_SLOTMAP = range(7)
# enum sim_swim_state (memberlist node state)
SWIM_ALIVE, SWIM_SUSPECT, SWIM_DEAD, SWIM_LEFT = 0, 1, 2, 3
CF_BASELINE_JOINED, CF_RANDOM_FANOUT, CF_AWARENESS_PROBE, CF_JOIN_SYNC, CF_TCP_FALLBACK, CF_NACKS, CF_FORCE_SHARDED, CF_PRUNE_DELAY = 1, 2, 4, 8, 16, 32, 64, 128
XCHG_ALL_TO_ALL, XCHG_ALL_GATHER, XCHG_PACKED = 0, 1, 2   # (ALL_GATHER: retired in ABI 13, never reported)
EXCHANGE_ID_BYTES = 128
F_NO_BROADCAST, F_ACK, F_RESPOND = 1, 2, 4


class SimError(RuntimeError):
    def __init__(self, code, what):
        super().__init__(f"{what}: {_ERR.get(code, code)}")
        self.code = code


class Config(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in (
        "struct_size", "n_nodes", "vshards", "shard_rank", "shard_count", "fanout", "view_slots",
        "event_ring", "query_ring", "retransmit_mult", "probe_interval", "suspicion_mult",
        "suspicion_max_mult", "indirect_checks", "loss_u32", "intent_timeout", "leave_delay",
        "reap_interval", "reconnect_timeout", "tombstone_timeout", "queue_check_interval", "max_queue_depth",
        "min_queue_depth", "push_pull_interval", "chunks", "recycle_interval", "pkt_records", "flags", "gossip_to_the_dead", "reconnect_interval",
        "ring_overflow", "reserved0")] + [("seed", C.c_uint64)]


class Stats(C.Structure):
    _fields_ = [("members", C.c_uint32), ("failed", C.c_uint32), ("left", C.c_uint32),
                ("health_score", C.c_uint32), ("member_time", C.c_uint64),
                ("event_time", C.c_uint64), ("query_time", C.c_uint64),
                ("intent_queue", C.c_uint32), ("event_queue", C.c_uint32),
                ("query_queue", C.c_uint32), ("swim_queue", C.c_uint32),
                ("serf_state", C.c_uint32), ("up", C.c_uint32), ("incarnation", C.c_uint32),
                ("queue_overflow", C.c_uint32)]


class ClusterStats(C.Structure):
    _fields_ = [("up", C.c_uint64), ("queued", C.c_uint64 * 4), ("overflow", C.c_uint64),
                ("inbox_records", C.c_uint64), ("failed", C.c_uint64), ("left", C.c_uint64), ("max_queue", C.c_uint64),
                ("ops_dropped", C.c_uint64), ("slots_in_use", C.c_uint64), ("slots_recycled", C.c_uint64),
                ("events_lost", C.c_uint64)]


class RecycleCand(C.Structure):
    _fields_ = [("subject", C.c_uint32), ("slot", C.c_uint32), ("flags", C.c_uint32), ("pad", C.c_uint32),
                ("ltime", C.c_uint64), ("inc", C.c_uint32), ("bits", C.c_uint32), ("conf", C.c_uint32 * 4)]


RECYCLE_BATCH = 64


class Event(C.Structure):
    _fields_ = [("tick", C.c_uint32), ("observer", C.c_uint32), ("type", C.c_uint32),
                ("key", C.c_uint32), ("ltime", C.c_uint64)]


ROW_DTYPE = np.dtype([("clock", "<u8"), ("event_clock", "<u8"), ("query_clock", "<u8"),
                      ("event_min", "<u8"), ("query_min", "<u8"), ("flags", "<u4"), ("inc", "<u4"),
                      ("n_known", "<u4"), ("n_failed", "<u4"), ("n_left", "<u4"),
                      ("next_seq", "<u4"), ("overflow", "<u4"), ("susp_next", "<u4"),
                      ("awareness", "<u4"), ("reap_next", "<u4"), ("susp", "<u2", (16,))])
REC_DTYPE = np.dtype([("key", "<u4"), ("meta", "<u4"), ("val", "<u8")])
PACKET_DTYPE = np.dtype([("key", "<u4", (4,)), ("val_lo", "<u4", (4,)), ("hi_meta", "<u4", (4,))])  # 12-byte wire records
VIEW_DTYPE = np.dtype([("ltime", "<u8"), ("inc", "<u4"), ("bits", "<u4"), ("conf", "<u4", (4,))])
BUCKET_DTYPE = np.dtype([("ltime", "<u8"), ("keys", "<u4", (CKEYS,))])
_ARR_DTYPE = {ARR_ROWS: ROW_DTYPE, ARR_QUEUE: REC_DTYPE, ARR_INBOX: PACKET_DTYPE, ARR_VIEW: VIEW_DTYPE,
              ARR_ERING: BUCKET_DTYPE, ARR_QRING: BUCKET_DTYPE, ARR_SLOTMAP: np.dtype("<u4")}

# every symbol include/serf_sim.h declares (without prefix)
ABI_SYMBOLS = ("create", "destroy", "set_stream", "join", "leave", "force_leave", "user_event",
               "query", "inject", "step", "sync", "tick", "members", "stats_get", "watch",
               "drain_events", "state_digest", "dump_state", "convergence", "convergence_many", "exchange_bytes",
               "bind_exchange", "snapshot", "restore", "query_status", "query_responders", "profile", "profile_read", "profile_read_stats", "cluster_stats_get", "resident_planes",
               "bind_exchange2", "bind_exchange3", "exchange_chunks", "exchange_layout", "step_begin", "step_chunk", "step_end",
               "recycle_due", "recycle_scan", "recycle_apply", "pp_due", "pp_plan", "pp_export", "pp_merge",
               "query_filtered", "set_tags", "init_tags", "inject_record", "deliver_message", "user_event_bytes", "peek_packet", "suspect_requests", "suspect_export", "suspect_import",
               "exchange_unique_id", "exchange_init", "exchange_chunk", "exchange_wait", "exchange_library",
               "abi_version", "backend_name")


def make_config(n_nodes, *, fanout=3, vshards=1, shard_rank=0, shard_count=1, view_slots=0,
                event_ring=512, query_ring=512, retransmit_mult=4, probe_interval=0,
                suspicion_mult=4, suspicion_max_mult=6, indirect_checks=3, loss=0.0,
                intent_timeout=0, leave_delay=30, reap_interval=0, reconnect_timeout=432000, tombstone_timeout=432000,
                queue_check_interval=0, max_queue_depth=4096, min_queue_depth=0, push_pull_interval=0, chunks=0, recycle_interval=0,
                pkt_records=0, gossip_to_the_dead=0, reconnect_interval=0, ring_overflow=0, tcp_fallback=False, nacks=False, awareness_probe=False, join_sync=False, force_sharded=False, prune_delay=False, flags=CF_BASELINE_JOINED, seed=DEFAULT_SEED):
    cfg = Config()
    cfg.struct_size = C.sizeof(Config)
    cfg.n_nodes, cfg.vshards, cfg.shard_rank, cfg.shard_count = n_nodes, vshards, shard_rank, shard_count
    cfg.fanout, cfg.view_slots, cfg.event_ring, cfg.query_ring = fanout, view_slots, event_ring, query_ring
    cfg.retransmit_mult, cfg.probe_interval = retransmit_mult, probe_interval
    cfg.suspicion_mult, cfg.suspicion_max_mult, cfg.indirect_checks = suspicion_mult, suspicion_max_mult, indirect_checks
    cfg.loss_u32 = min(0xFFFFFFFF, int(round(loss * 2 ** 32)))
    cfg.intent_timeout, cfg.leave_delay, cfg.flags, cfg.seed = intent_timeout, leave_delay, flags, seed
    cfg.reap_interval, cfg.reconnect_timeout, cfg.tombstone_timeout = reap_interval, reconnect_timeout, tombstone_timeout
    cfg.queue_check_interval, cfg.max_queue_depth, cfg.min_queue_depth = queue_check_interval, max_queue_depth, min_queue_depth
    cfg.push_pull_interval, cfg.chunks, cfg.recycle_interval = push_pull_interval, chunks, recycle_interval
    cfg.pkt_records = pkt_records  # 0 = 4 records (one page) per packet; 8 / 12 / 16 = more pages
    cfg.gossip_to_the_dead = gossip_to_the_dead
    cfg.reconnect_interval = reconnect_interval  # Reconnector (base.rs:612-681): 0 = off
    cfg.ring_overflow = ring_overflow  # overflow rows per de-dup ring and node (a full bucket's further keys); 0 = none
    if awareness_probe:
        cfg.flags |= CF_AWARENESS_PROBE
    if join_sync:
        cfg.flags |= CF_JOIN_SYNC
    if tcp_fallback:
        cfg.flags |= CF_TCP_FALLBACK   # memberlist's stream-transport fallback ping: packet loss alone never fails a probe
    if nacks:
        cfg.flags |= CF_NACKS          # awareness += relays that were asked and did not nack
    if force_sharded:
        cfg.flags |= CF_FORCE_SHARDED  # one rank of the N > 1 path: exchange buffers, the sharded kernel, host-driven push-pull
    if prune_delay:
        cfg.flags |= CF_PRUNE_DELAY    # handle_prune's wait (base.rs:1628-1653): a Leaving member's forced erase comes leave_delay ticks later
    return cfg


class SimLib:
    """One loaded implementation of the ABI."""

    def __init__(self, path, prefix="sim_", optional=()):
        if not os.path.exists(path):
            raise FileNotFoundError(
                f"{path} is missing — build it first (python -c 'import __graft_entry__ as g; g.build()')")
        self.path, self.prefix = path, prefix
        self.dll = C.CDLL(path)
        self.f = {}
        H, u32, u64, vp = C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p
        sig = {
            "create": (C.c_int, [C.POINTER(Config), C.POINTER(H)]),
            "destroy": (C.c_int, [H]),
            "set_stream": (C.c_int, [H, vp]),
            "join": (C.c_int, [H, u32, u32]),
            "leave": (C.c_int, [H, u32]),
            "force_leave": (C.c_int, [H, u32, u32, C.c_int]),
            "user_event": (C.c_int, [H, u32, u32, u32, C.c_int]),
            "query": (C.c_int, [H, u32, u32, u32]),
            "query_filtered": (C.c_int, [H, u32, u32, u32, C.POINTER(u32), u32, u32]),
            "set_tags": (C.c_int, [H, u32, u32]),
            "init_tags": (C.c_int, [H, u32, u32, vp]),
            "inject": (C.c_int, [H, u64, u32, u32, u32, u32]),
            "step": (C.c_int, [H, u32]),
            "sync": (C.c_int, [H]),
            "tick": (C.c_int, [H, C.POINTER(u64)]),
            "members": (C.c_int, [H, u32, vp, vp, u32]),
            "stats_get": (C.c_int, [H, u32, C.POINTER(Stats)]),
            "watch": (C.c_int, [H, u32]),
            "drain_events": (C.c_int, [H, C.POINTER(Event), u32, C.POINTER(u32)]),
            "state_digest": (C.c_int, [H, C.POINTER(u64 * 8)]),
            "dump_state": (C.c_int, [H, u32, vp, C.c_size_t, C.POINTER(C.c_size_t)]),
            "convergence": (C.c_int, [H, u32, u32, u64, C.POINTER(u64), C.POINTER(u64)]),
            "convergence_many": (C.c_int, [H, u32, C.POINTER(u32), C.POINTER(u32), C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]),
            "exchange_bytes": (C.c_int, [H, C.POINTER(C.c_size_t)]),
            "bind_exchange": (C.c_int, [H, vp, vp]),
            "bind_exchange2": (C.c_int, [H, vp, vp, vp]),
            "bind_exchange3": (C.c_int, [H, vp, C.c_size_t, vp, vp, C.c_size_t]),
            "exchange_chunks": (C.c_int, [H, C.POINTER(u32), C.POINTER(C.c_size_t)]),
            "exchange_layout": (C.c_int, [H, C.POINTER(u32), C.POINTER(u32), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
            "resident_planes": (C.c_int, [H, C.POINTER(u32), C.POINTER(C.c_uint64)]),
            "step_begin": (C.c_int, [H]),
            "step_chunk": (C.c_int, [H, u32]),
            "step_end": (C.c_int, [H]),
            "pp_due": (C.c_int, [H]),
            "pp_plan": (C.c_int, [H, C.POINTER(u32), C.POINTER(u32), C.POINTER(C.c_size_t)]),
            "pp_export": (C.c_int, [H, C.c_int, vp]),
            "pp_merge": (C.c_int, [H, C.c_int, vp]),
            "recycle_due": (C.c_int, [H]),
            "recycle_scan": (C.c_int, [H, C.POINTER(RecycleCand), u32, C.POINTER(u32)]),
            "recycle_apply": (C.c_int, [H, C.POINTER(RecycleCand), u32]),
            "snapshot": (C.c_int, [H, vp, C.c_size_t, C.POINTER(C.c_size_t)]),
            "restore": (C.c_int, [H, vp, C.c_size_t]),
            "query_status": (C.c_int, [H, u32, C.POINTER(u64), C.POINTER(u64), C.POINTER(C.c_int)]),
            "query_responders": (C.c_int, [H, u32, C.c_int, C.POINTER(u32), u32, C.POINTER(u32)]),
            "profile": (C.c_int, [H, C.c_int]),
            "profile_read": (C.c_int, [H, C.POINTER(C.c_double), C.POINTER(u64)]),
            "profile_read_stats": (C.c_int, [H, C.POINTER(C.c_double * 3), C.POINTER(u64)]),
            "cluster_stats_get": (C.c_int, [H, C.POINTER(ClusterStats)]),
            "inject_record": (C.c_int, [H, u64, u32, vp]),
            "deliver_message": (C.c_int, [H, u32, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]),
            "user_event_bytes": (C.c_int, [H, u32, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int]),
            "peek_packet": (C.c_int, [H, u32, u32, vp, C.c_size_t, C.POINTER(C.c_size_t)]),
            "suspect_requests": (C.c_int, [H, vp, u32, C.POINTER(u32)]),
            "suspect_export": (C.c_int, [H, vp]),
            "suspect_import": (C.c_int, [H, u64, vp, u32]),
            "exchange_unique_id": (C.c_int, [vp]),
            "exchange_init": (C.c_int, [H, vp, u32, u32]),
            "exchange_chunk": (C.c_int, [H, u32]),
            "exchange_wait": (C.c_int, [H]),
            "exchange_library": (C.c_int, [C.c_char_p, C.c_size_t]),
            "abi_version": (u32, []),
            "backend_name": (C.c_char_p, []),
        }
        for name, (res, args) in sig.items():
            if name in optional and not hasattr(self.dll, prefix + name):
                continue   # (tools/ab.py: a build of an older ABI next to the current one)
            fn = getattr(self.dll, prefix + name)  # AttributeError if the symbol is not exported
            fn.restype, fn.argtypes = res, args
            self.f[name] = fn

    def backend_name(self):
        return self.f["backend_name"]().decode()

    def exchange_unique_id(self) -> bytes:
        """ncclGetUniqueId through the library (rank 0); the bytes go to every rank, then Sim.exchange_init."""
        buf = C.create_string_buffer(EXCHANGE_ID_BYTES)
        rc = self.f["exchange_unique_id"](buf)
        if rc:
            raise SimError(rc, "sim_exchange_unique_id")
        return buf.raw

    def exchange_library(self):
        """"RCCL x.y.z" of the collective library behind sim_exchange_* (None: the implementation has none)."""
        buf = C.create_string_buffer(64)
        return buf.value.decode() if self.f["exchange_library"](buf, 64) == 0 else None

    def abi_version(self):
        return self.f["abi_version"]()


class Sim:
    """A simulated cluster behind the C ABI; method names follow ``Serf`` (serf/api.rs)."""

    def __init__(self, lib: SimLib, cfg: Config):
        self.lib, self.cfg = lib, cfg
        self.h = C.c_void_p()
        rc = lib.f["create"](C.byref(cfg), C.byref(self.h))
        if rc:
            self.h = None
            raise SimError(rc, "sim_create")
        self.n = cfg.n_nodes

    def _ck(self, rc, what):
        if rc < 0:
            raise SimError(rc, what)
        return rc

    def close(self):
        if self.h:
            self.lib.f["destroy"](self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- Serf API (api.rs) ----
    def join(self, node, peer=0):
        self._ck(self.lib.f["join"](self.h, node, peer), "sim_join")

    def leave(self, node):
        self._ck(self.lib.f["leave"](self.h, node), "sim_leave")

    def remove_failed_node(self, node, subject, prune=False):
        self._ck(self.lib.f["force_leave"](self.h, node, subject, int(prune)), "sim_force_leave")

    def user_event(self, node, key, encoded_len=32, coalesce=False):
        self._ck(self.lib.f["user_event"](self.h, node, key, encoded_len, int(coalesce)), "sim_user_event")

    def query(self, node, query_id, flags=0, ids=None, tag_mask=NO_TAG_FILTER):
        """Serf::query (api.rs:304).  ids / tag_mask = QueryParam.filters in the form the ABI takes them
        (serf_amd.filters.TagTable.compile turns Filter::Id / Filter::Tag lists into this pair)."""
        if ids is None and tag_mask == NO_TAG_FILTER:
            self._ck(self.lib.f["query"](self.h, node, query_id, flags), "sim_query")
            return
        ids = list(ids or [])
        arr = (C.c_uint32 * max(1, len(ids)))(*ids)
        self._ck(self.lib.f["query_filtered"](self.h, node, query_id, flags, arr, len(ids), tag_mask), "sim_query_filtered")

    def set_tags(self, node, tag_class):
        """Serf::set_tags (api.rs:219) in the tag-class model (include/serf_sim.h)."""
        self._ck(self.lib.f["set_tags"](self.h, node, tag_class), "sim_set_tags")

    def inject(self, tick, op, node, a=0, b=0):
        self._ck(self.lib.f["inject"](self.h, tick, op, node, a, b), "sim_inject")

    def init_tags(self, classes, first=0):
        """Options::with_tags for the nodes [first, first + len(classes)): start-up tags, no gossip."""
        import numpy as np
        arr = np.ascontiguousarray(classes, dtype=np.uint8)
        self._ck(self.lib.f["init_tags"](self.h, first, len(arr), arr.ctypes.data), "sim_init_tags")

    # ---- the byte boundary of the delegate (delegate.rs:157-163, 317-384) ----
    def inject_record(self, tick, node, key, meta, val):
        """`node` receives the record (key, wire bits of meta, val) at the start of tick `tick` as if it came in a packet."""
        rec = np.zeros(1, REC_DTYPE)
        rec["key"], rec["meta"], rec["val"] = key, meta, val
        self._ck(self.lib.f["inject_record"](self.h, tick, node, rec.ctypes.data), "sim_inject_record")

    def deliver_message(self, node, data: bytes):
        """SerfDelegate::notify_message(buf): one framed serf message in the reference's encoding; returns bytes consumed."""
        used = C.c_size_t()
        self._ck(self.lib.f["deliver_message"](self.h, node, data, len(data), C.byref(used)), "sim_deliver_message")
        return used.value

    def user_event_bytes(self, node, name: bytes, payload: bytes, coalesce=False):
        """Serf::user_event(name, payload, coalesce) (api.rs:241-299) with the bytes themselves."""
        self._ck(self.lib.f["user_event_bytes"](self.h, node, name, len(name), payload, len(payload), int(coalesce)), "sim_user_event_bytes")

    def peek_packet(self, node, k) -> bytes:
        """SerfDelegate::broadcast_messages as bytes: the serf messages of the packet `node` sent in slot k last tick."""
        n = C.c_size_t()
        self._ck(self.lib.f["peek_packet"](self.h, node, k, None, 0, C.byref(n)), "sim_peek_packet")
        buf = np.zeros(max(1, n.value), np.uint8)
        self._ck(self.lib.f["peek_packet"](self.h, node, k, buf.ctypes.data, n.value, C.byref(n)), "sim_peek_packet")
        return buf[:n.value].tobytes()

    def set_stream(self, stream_ptr):
        self._ck(self.lib.f["set_stream"](self.h, C.c_void_p(stream_ptr)), "sim_set_stream")

    # ---- the round's all-to-all issued by the library itself over RCCL (include/serf_sim.h sim_exchange_*) ----
    def exchange_init(self, unique_id: bytes, rank, world):
        self._ck(self.lib.f["exchange_init"](self.h, unique_id, rank, world), "sim_exchange_init")

    def exchange_chunk(self, chunk):
        self._ck(self.lib.f["exchange_chunk"](self.h, chunk), "sim_exchange_chunk")

    def exchange_wait(self):
        self._ck(self.lib.f["exchange_wait"](self.h), "sim_exchange_wait")

    def step(self, n=1):
        self._ck(self.lib.f["step"](self.h, n), "sim_step")

    def sync(self):
        self._ck(self.lib.f["sync"](self.h), "sim_sync")

    @property
    def tick(self):
        t = C.c_uint64()
        self._ck(self.lib.f["tick"](self.h, C.byref(t)), "sim_tick")
        return t.value

    def members(self, observer):
        st = np.zeros(self.n, np.uint8)
        lt = np.zeros(self.n, np.uint64)
        self._ck(self.lib.f["members"](self.h, observer, st.ctypes.data, lt.ctypes.data, self.n), "sim_members")
        return st, lt

    def stats(self, node):
        s = Stats()
        self._ck(self.lib.f["stats_get"](self.h, node, C.byref(s)), "sim_stats_get")
        return s

    def watch(self, observer):
        self._ck(self.lib.f["watch"](self.h, observer), "sim_watch")

    def drain_events(self, cap=65536):
        buf = (Event * cap)()
        n = C.c_uint32()
        self._ck(self.lib.f["drain_events"](self.h, buf, cap, C.byref(n)), "sim_drain_events")
        return [(e.tick, e.observer, e.type, e.key, e.ltime) for e in buf[:n.value]]

    def digest(self):
        d = (C.c_uint64 * 8)()
        self._ck(self.lib.f["state_digest"](self.h, C.byref(d)), "sim_state_digest")
        return tuple(int(x) for x in d)

    def dump(self, which, dtype=None):
        """One state array of the local shard as a structured numpy array (`dtype` overrides the product's record
        layout: an implementation built with other capacity constants has wider rows / buckets)."""
        n = C.c_size_t()
        self._ck(self.lib.f["dump_state"](self.h, which, None, 0, C.byref(n)), "sim_dump_state")
        buf = np.zeros(n.value, np.uint8)
        self._ck(self.lib.f["dump_state"](self.h, which, buf.ctypes.data, n.value, C.byref(n)), "sim_dump_state")
        return buf.view(_ARR_DTYPE[which] if dtype is None else dtype)

    def convergence(self, kind, key, ltime):
        seen, up = C.c_uint64(), C.c_uint64()
        self._ck(self.lib.f["convergence"](self.h, kind, key, ltime, C.byref(seen), C.byref(up)), "sim_convergence")
        return seen.value, up.value

    def convergence_many(self, rumours):
        """([seen_i], up) for a list of up to 64 (kind, key, ltime) in one pass over the nodes."""
        n = len(rumours)
        kinds = (C.c_uint32 * max(1, n))(*[r[0] for r in rumours])
        keys = (C.c_uint32 * max(1, n))(*[r[1] for r in rumours])
        lts = (C.c_uint64 * max(1, n))(*[r[2] for r in rumours])
        seen, up = (C.c_uint64 * max(1, n))(), C.c_uint64()
        self._ck(self.lib.f["convergence_many"](self.h, n, kinds, keys, lts, seen, C.byref(up)), "sim_convergence_many")
        return [int(x) for x in seen[:n]], up.value

    def snapshot(self):
        """Canonical image of the whole simulated cluster (bytes); restores into any implementation of the ABI."""
        n = C.c_size_t()
        self._ck(self.lib.f["snapshot"](self.h, None, 0, C.byref(n)), "sim_snapshot")
        buf = np.zeros(n.value, np.uint8)
        self._ck(self.lib.f["snapshot"](self.h, buf.ctypes.data, n.value, C.byref(n)), "sim_snapshot")
        return buf

    def restore(self, image):
        image = np.ascontiguousarray(image, dtype=np.uint8)
        self._ck(self.lib.f["restore"](self.h, image.ctypes.data, image.size), "sim_restore")

    def query_status(self, query_id):
        """(acks, responses, still_open) of a running query, as its origin counts them (query.rs:240-303)."""
        a, r, o = C.c_uint64(), C.c_uint64(), C.c_int()
        self._ck(self.lib.f["query_status"](self.h, query_id, C.byref(a), C.byref(r), C.byref(o)), "sim_query_status")
        return a.value, r.value, bool(o.value)

    def query_responders(self, query_id, which):
        """Ids of this shard's nodes whose ack (which = 0) / response (which = 1) reached the origin of the running query —
        what QueryResponse::ack_rx / response_rx deliver (query.rs:201-212), ascending."""
        n = C.c_uint32()
        self._ck(self.lib.f["query_responders"](self.h, query_id, which, None, 0, C.byref(n)), "sim_query_responders")
        out = (C.c_uint32 * max(1, n.value))()
        self._ck(self.lib.f["query_responders"](self.h, query_id, which, out, n.value, C.byref(n)), "sim_query_responders")
        return list(out[:n.value])

    def profile(self, enable=True):
        self._ck(self.lib.f["profile"](self.h, int(enable)), "sim_profile")

    def profile_read(self):
        """(summed tick-kernel milliseconds, launches) since the last read."""
        ms, n = C.c_double(), C.c_uint64()
        self._ck(self.lib.f["profile_read"](self.h, C.byref(ms), C.byref(n)), "sim_profile_read")
        return ms.value, n.value

    def profile_read_stats(self):
        """(sum, min, max) of the timed tick-kernel launches in milliseconds, and their number."""
        ms, n = (C.c_double * 3)(), C.c_uint64()
        self._ck(self.lib.f["profile_read_stats"](self.h, C.byref(ms), C.byref(n)), "sim_profile_read_stats")
        return (ms[0], ms[1], ms[2]), n.value

    def resident_planes(self):
        """{array: (planes with memory, planes)} for view / event ring / query ring, and the bytes of one plane"""
        out = (C.c_uint32 * 6)()
        bpp = C.c_uint64()
        self._ck(self.lib.f["resident_planes"](self.h, out, C.byref(bpp)), "sim_resident_planes")
        return {"view": (out[0], out[1]), "event_ring": (out[2], out[3]), "query_ring": (out[4], out[5]), "bytes_per_plane": bpp.value}

    def cluster_stats(self):
        """Load figures summed over the local shard's nodes (queue depths by class, model-bound drops, ...)."""
        s = ClusterStats()
        self._ck(self.lib.f["cluster_stats_get"](self.h, C.byref(s)), "sim_cluster_stats_get")
        return {"up": s.up, "queued": [int(x) for x in s.queued], "overflow": s.overflow,
                "inbox_records": s.inbox_records, "failed": s.failed, "left": s.left, "max_queue": s.max_queue,
                "ops_dropped": s.ops_dropped, "slots_in_use": s.slots_in_use, "slots_recycled": s.slots_recycled,
                "events_lost": s.events_lost}

    def exchange_bytes(self):
        n = C.c_size_t()
        self._ck(self.lib.f["exchange_bytes"](self.h, C.byref(n)), "sim_exchange_bytes")
        return n.value

    def bind_exchange3(self, send_ptr, send_bytes, recv0_ptr, recv1_ptr, recv_bytes):
        """sim_bind_exchange2 with the sizes of the caller's buffers (SIM_EINVAL when one is too small)"""
        self._ck(self.lib.f["bind_exchange3"](self.h, C.c_void_p(send_ptr), send_bytes, C.c_void_p(recv0_ptr), C.c_void_p(recv1_ptr), recv_bytes), "sim_bind_exchange3")

    def bind_exchange2(self, send_ptr, recv0_ptr, recv1_ptr):
        self._ck(self.lib.f["bind_exchange2"](self.h, C.c_void_p(send_ptr), C.c_void_p(recv0_ptr), C.c_void_p(recv1_ptr)), "sim_bind_exchange2")

    def exchange_chunks(self):
        """(number of sender chunks, bytes of one chunk's region of the exchange buffers)."""
        c, n = C.c_uint32(), C.c_size_t()
        self._ck(self.lib.f["exchange_chunks"](self.h, C.byref(c), C.byref(n)), "sim_exchange_chunks")
        return c.value, n.value

    def exchange_layout(self):
        """(kind, planes, bytes of the send buffer, bytes of the receive buffer) — every kind is an equal-split all-to-all;
        XCHG_ALL_TO_ALL: the slabs of the bijection, written by the tick kernel; XCHG_PACKED: the random fan-out on a shard — the
        slabs are packed from the packets the senders keep, and after a restore the exchange has to be run once more."""
        k, p, a, b = C.c_uint32(), C.c_uint32(), C.c_size_t(), C.c_size_t()
        self._ck(self.lib.f["exchange_layout"](self.h, C.byref(k), C.byref(p), C.byref(a), C.byref(b)), "sim_exchange_layout")
        return k.value, p.value, a.value, b.value

    def pp_due(self):
        return self._ck(self.lib.f["pp_due"](self.h), "sim_pp_due") > 0

    def pp_plan(self, n_shards):
        """(records to send to each peer in round 1, records to receive from each peer, bytes per record)."""
        snd, rcv, rb = (C.c_uint32 * n_shards)(), (C.c_uint32 * n_shards)(), C.c_size_t()
        self._ck(self.lib.f["pp_plan"](self.h, snd, rcv, C.byref(rb)), "sim_pp_plan")
        return list(snd), list(rcv), rb.value

    def pp_export(self, rnd, send_ptr):
        self._ck(self.lib.f["pp_export"](self.h, rnd, C.c_void_p(send_ptr)), "sim_pp_export")

    def pp_merge(self, rnd, recv_ptr):
        self._ck(self.lib.f["pp_merge"](self.h, rnd, C.c_void_p(recv_ptr)), "sim_pp_merge")

    def suspect_requests(self):
        """(prober, target) pairs of the probes that failed on a slot-less target in the tick just ended, sorted by
        prober (sharded hosts: gather, merge, inject OP_SUSPECT on every shard); empties the shard's list."""
        buf = np.zeros((SUSPECT_REQ_MAX, 2), np.uint32)
        n = C.c_uint32()
        self._ck(self.lib.f["suspect_requests"](self.h, buf.ctypes.data, SUSPECT_REQ_MAX, C.byref(n)), "sim_suspect_requests")
        return buf[:n.value].copy()

    def suspect_export(self, out_ptr):
        """Enqueue a copy of the head (SREQ_HEAD_WORDS u32: count + pairs) of the just-ended tick's request list into
        caller memory (device memory for the HIP library)."""
        self._ck(self.lib.f["suspect_export"](self.h, C.c_void_p(out_ptr)), "sim_suspect_export")

    def suspect_import(self, of_tick, heads_ptr, world):
        """The gathered heads of all shards for tick `of_tick` (host memory): merged and scheduled for of_tick + 2."""
        self._ck(self.lib.f["suspect_import"](self.h, of_tick, C.c_void_p(heads_ptr), world), "sim_suspect_import")

    def recycle_due(self):
        return self._ck(self.lib.f["recycle_due"](self.h), "sim_recycle_due") > 0

    def recycle_scan(self):
        """This shard's verdict on the recycling candidates of the next tick: numpy uint32 array [n, 12] (the
        sim_recycle_cand records as words), ready for an all-gather."""
        buf = (RecycleCand * RECYCLE_BATCH)()
        n = C.c_uint32()
        self._ck(self.lib.f["recycle_scan"](self.h, buf, RECYCLE_BATCH, C.byref(n)), "sim_recycle_scan")
        return np.frombuffer(buf, dtype=np.uint32, count=n.value * 12).reshape(n.value, 12).copy()

    def recycle_apply(self, words):
        words = np.ascontiguousarray(words, dtype=np.uint32).reshape(-1, 12)
        buf = (RecycleCand * max(1, len(words))).from_buffer_copy(words.tobytes() if len(words) else bytes(C.sizeof(RecycleCand)))
        self._ck(self.lib.f["recycle_apply"](self.h, buf, len(words)), "sim_recycle_apply")

    def step_begin(self):
        self._ck(self.lib.f["step_begin"](self.h), "sim_step_begin")

    def step_chunk(self, chunk):
        self._ck(self.lib.f["step_chunk"](self.h, chunk), "sim_step_chunk")

    def step_end(self):
        self._ck(self.lib.f["step_end"](self.h), "sim_step_end")

    def bind_exchange(self, send_ptr, recv_ptr):
        self._ck(self.lib.f["bind_exchange"](self.h, C.c_void_p(send_ptr), C.c_void_p(recv_ptr)), "sim_bind_exchange")
