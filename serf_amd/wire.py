"""Wire codec of the serf messages on the simulated path — host side.

Restates the byte layout of serf-core's message types so that (a) the encoded length a simulated record carries
(`sim_record.meta` length field, the unit TransmitLimitedQueue sorts and budgets by) is the length the reference would
put on the wire, and (b) a drained simulator event / record can be turned into the bytes a real `serf` process would
accept, and back.

Layout, from the reference sources (all lengths below are exact; the functions are checked by round-trip tests the
way serf-core/src/types/tests.rs:27-110 checks the Rust ones):

* framing (types/message.rs:397-428 `encode_message`): one type byte `merge(LengthDelimited, TAG)`, the body length as
  a varint (`(encoded_len as u32).encode`), the body.  TAGs: leave 1, join 2, push_pull 3, user_event 4, query 5,
  query_response 6, conflict_response 7, relay 8 (types/message.rs:17-24).
* JoinMessage (types/join.rs:123-158): `LTIME_BYTE varint(ltime)  ID_BYTE <id, length-delimited>`; tags ltime 1, id 2.
* LeaveMessage (types/leave.rs:138-195): `LTIME_BYTE varint(ltime)  [PRUNE_BYTE 0x01]  ID_BYTE <id>`; tags 1, 2, 3.
* UserEventMessage (types/user_event/message.rs:205-272): `LTIME_BYTE varint  [CC_BYTE 0x01]  [NAME_BYTE <name>]
  [PAYLOAD_BYTE <payload>]`; tags ltime 1, cc 2, name 3, payload 4; empty name / payload and cc = false are omitted.
* QueryMessage (types/query.rs:404-527): `LTIME varint  ID varint(u32)  FROM <node>  (FILTERS <filter>)*  FLAGS
  varint(u32)  RELAY_FACTOR byte  TIMEOUT varint  [NAME <name>]  [PAYLOAD <payload>]`; tags 1..9.
* PushPullMessage (types/push_pull.rs:100-111): `LTIME varint  (STATUS_LTIMES <{id, ltime}>)*  (LEFT_MEMBERS <id>)*
  EVENT_LTIME varint  (EVENTS <user events>)*  QUERY_LTIME varint`; tags 1..6.

What the reference does NOT contain is memberlist-proto (memberlist-core 0.8.1, Cargo.toml:39-41, not vendored): the
bit layout of `merge(wire_type, tag)`, the numeric values of `WireType`, and how `Node<I, A>` and `Duration` encode.
They are isolated below (`WIRE_*`, `merge`, `encode_node`, `encode_duration`) as UPSTREAM-RECALL assumptions — the
protobuf convention `(tag << 3) | wire_type`, which is the only single-byte reading under which the reference's own
tags (up to 10) do not collide with the wire type; LEB128 varints — and none of the LENGTHS depends on them: every tag
is one byte whatever its bits are.  A simulated node id travels as the UTF-8 decimal string of its number.
"""
from __future__ import annotations

from dataclasses import dataclass, field

# ---- memberlist-proto assumptions (UPSTREAM-RECALL; see module docstring) -------------------------------------------
WIRE_BYTE, WIRE_VARINT, WIRE_LEN = 0, 1, 2


def merge(wire_type: int, tag: int) -> int:
    return ((tag << 3) | wire_type) & 0xFF


def split(b: int):
    return b & 7, b >> 3


def varint(v: int) -> bytes:
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def varint_len(v: int) -> int:
    n = 1
    while v >= 0x80:
        v >>= 7
        n += 1
    return n


def read_varint(buf: bytes, off: int):
    v = shift = 0
    while True:
        if off >= len(buf):
            raise ValueError("truncated varint")
        b = buf[off]
        off += 1
        v |= (b & 0x7F) << shift
        if not b & 0x80:
            return v, off
        shift += 7
        if shift > 63:
            raise ValueError("varint too long")


def ld(data: bytes) -> bytes:  # length-delimited
    return varint(len(data)) + data


def read_ld(buf: bytes, off: int):
    n, off = read_varint(buf, off)
    if off + n > len(buf):
        raise ValueError("truncated length-delimited field")
    return bytes(buf[off:off + n]), off + n


def node_id(gid: int) -> bytes:
    """Wire form of a simulated node's id."""
    return str(int(gid)).encode()


def encode_node(gid: int) -> bytes:
    """`Node<I, A>` of a simulated node: id + a 6-byte socket address (assumption: {id: tag 1, addr: tag 2})."""
    addr = bytes([10, (gid >> 16) & 0xFF, (gid >> 8) & 0xFF, gid & 0xFF]) + (7946).to_bytes(2, "big")
    return bytes([merge(WIRE_LEN, 1)]) + ld(node_id(gid)) + bytes([merge(WIRE_LEN, 2)]) + ld(addr)


def parse_node_id(data: bytes) -> int:
    """a simulated node's id: the decimal string of its number (digits only, at most 2^32 - 1)"""
    if not data or not all(0x30 <= c <= 0x39 for c in data):
        raise ValueError("node id is not a decimal number")
    v = int(data)
    if v > 0xFFFFFFFF:
        raise ValueError("node id out of range")
    return v


def decode_node(buf: bytes) -> int:
    d = _known(buf, {KB(1, WIRE_LEN): "id", KB(2, WIRE_LEN): "addr"}, required=("id",))
    return parse_node_id(d["id"])


def encode_duration(ms: int) -> bytes:
    return varint(ms)


# ---- message types (types/message.rs:17-44) -----------------------------------------------------------------------
LEAVE, JOIN, PUSH_PULL, USER_EVENT, QUERY, QUERY_RESPONSE, CONFLICT_RESPONSE, RELAY = 1, 2, 3, 4, 5, 6, 7, 8


@dataclass
class Join:
    ltime: int
    id: int

    def body(self) -> bytes:
        return bytes([merge(WIRE_VARINT, 1)]) + varint(self.ltime) + bytes([merge(WIRE_LEN, 2)]) + ld(node_id(self.id))


@dataclass
class Leave:
    ltime: int
    id: int
    prune: bool = False

    def body(self) -> bytes:
        out = bytes([merge(WIRE_VARINT, 1)]) + varint(self.ltime)
        if self.prune:
            out += bytes([merge(WIRE_BYTE, 2), 1])
        return out + bytes([merge(WIRE_LEN, 3)]) + ld(node_id(self.id))


@dataclass
class UserEvent:
    ltime: int
    name: bytes = b""
    payload: bytes = b""
    cc: bool = False

    def body(self) -> bytes:
        out = bytes([merge(WIRE_VARINT, 1)]) + varint(self.ltime)
        if self.cc:
            out += bytes([merge(WIRE_BYTE, 2), 1])
        if self.name:
            out += bytes([merge(WIRE_LEN, 3)]) + ld(self.name)
        if self.payload:
            out += bytes([merge(WIRE_LEN, 4)]) + ld(self.payload)
        return out


@dataclass
class Query:
    ltime: int
    id: int
    from_node: int
    flags: int = 0
    relay_factor: int = 0
    timeout_ms: int = 0
    name: bytes = b""
    payload: bytes = b""
    filters: list = field(default_factory=list)

    def body(self) -> bytes:
        out = bytes([merge(WIRE_VARINT, 1)]) + varint(self.ltime)
        out += bytes([merge(WIRE_VARINT, 2)]) + varint(self.id)
        out += bytes([merge(WIRE_LEN, 3)]) + ld(encode_node(self.from_node))
        for f in self.filters:
            out += bytes([merge(WIRE_LEN, 4)]) + ld(f)
        out += bytes([merge(WIRE_VARINT, 5)]) + varint(self.flags)
        out += bytes([merge(WIRE_VARINT, 6), self.relay_factor & 0xFF])
        out += bytes([merge(WIRE_VARINT, 7)]) + encode_duration(self.timeout_ms)
        if self.name:
            out += bytes([merge(WIRE_LEN, 8)]) + ld(self.name)
        if self.payload:
            out += bytes([merge(WIRE_LEN, 9)]) + ld(self.payload)
        return out


@dataclass
class PushPull:
    """SerfDelegate::local_state (delegate.rs:386-425): clocks, every member's status_ltime, the left members, the event buffer."""
    ltime: int
    status_ltimes: dict = field(default_factory=dict)   # node id -> status_ltime
    left_members: list = field(default_factory=list)
    event_ltime: int = 0
    events: list = field(default_factory=list)          # [(ltime, [(name, payload), ...]), ...]
    query_ltime: int = 0

    def body(self) -> bytes:
        out = bytes([merge(WIRE_VARINT, 1)]) + varint(self.ltime)
        for nid, lt in self.status_ltimes.items():
            pair = bytes([merge(WIRE_LEN, 1)]) + ld(node_id(nid)) + bytes([merge(WIRE_VARINT, 2)]) + varint(lt)
            out += bytes([merge(WIRE_LEN, 2)]) + ld(pair)
        for nid in self.left_members:
            out += bytes([merge(WIRE_LEN, 3)]) + ld(node_id(nid))
        out += bytes([merge(WIRE_VARINT, 4)]) + varint(self.event_ltime)
        for lt, evs in self.events:   # UserEvents (types/user_event/user_events.rs): ltime + repeated {name, payload}
            ue = bytes([merge(WIRE_VARINT, 1)]) + varint(lt)
            for name, payload in evs:
                one = b""
                if name:
                    one += bytes([merge(WIRE_LEN, 1)]) + ld(name)
                if payload:
                    one += bytes([merge(WIRE_LEN, 2)]) + ld(payload)
                ue += bytes([merge(WIRE_LEN, 2)]) + ld(one)
            out += bytes([merge(WIRE_LEN, 5)]) + ld(ue)
        return out + bytes([merge(WIRE_VARINT, 6)]) + varint(self.query_ltime)


@dataclass
class QueryResponse:
    """QueryResponseMessage (types/query/response.rs:9-19, 244-300): ltime 1, id 2, from 3 (Node), flags 4 (QueryFlag: ACK = 1),
    payload 5 (omitted when empty)."""
    ltime: int
    id: int
    from_node: int
    flags: int = 0
    payload: bytes = b""

    def body(self) -> bytes:
        out = bytes([merge(WIRE_VARINT, 1)]) + varint(self.ltime) + bytes([merge(WIRE_VARINT, 2)]) + varint(self.id)
        out += bytes([merge(WIRE_LEN, 3)]) + ld(encode_node(self.from_node)) + bytes([merge(WIRE_VARINT, 4)]) + varint(self.flags)
        if self.payload:
            out += bytes([merge(WIRE_LEN, 5)]) + ld(self.payload)
        return out


@dataclass
class Relay:
    """encode_relay_message (types/message.rs:431-470): RELAY_MESSAGE_BYTE, then — with NO length of its own —
    RELAY_NODE_BYTE <node, length-delimited>, RELAY_MSG_BYTE, and the wrapped message framed as usual, to the end of the buffer."""
    node: int          # whom the receiver is asked to forward the message to
    msg: object        # the wrapped message (relay_response wraps a QueryResponse: query.rs:523-601)


@dataclass
class ConflictResponse:
    """ConflictResponseMessage (types/conflict.rs): carried opaquely — SerfDelegate::notify_message has no arm for it (delegate.rs:286-288)"""
    body: bytes = b""


_TAG_OF = {Leave: LEAVE, Join: JOIN, PushPull: PUSH_PULL, UserEvent: USER_EVENT, Query: QUERY, QueryResponse: QUERY_RESPONSE}


def encode_message(msg) -> bytes:
    """types/message.rs:397-428: type byte, varint body length, body."""
    if isinstance(msg, Relay):
        return (bytes([merge(WIRE_LEN, RELAY), merge(WIRE_LEN, 1)]) + ld(encode_node(msg.node)) + bytes([merge(WIRE_LEN, 2)])
                + encode_message(msg.msg))
    body = msg.body()
    return bytes([merge(WIRE_LEN, _TAG_OF[type(msg)])]) + varint(len(body)) + body


def encoded_len(msg) -> int:
    body = len(msg.body())
    return 1 + varint_len(body) + body


def KB(tag: int, wt: int) -> int:
    """a field's key byte"""
    return (tag << 3) | wt


def _fields(body: bytes, raw1: int = 0):
    """(key byte, value) of every field of a body.  raw1: a key byte whose value is ONE raw byte although its wire type says
    Varint (QueryMessage.relay_factor: types/query.rs:484-490 writes `buf[offset] = self.relay_factor`)."""
    off = 0
    while off < len(body):
        kb = body[off]
        off += 1
        ty = kb & 7
        if (raw1 and kb == raw1) or ty == WIRE_BYTE:
            if off >= len(body):
                raise ValueError("truncated byte field")
            v, off = body[off], off + 1
        elif ty == WIRE_VARINT:
            v, off = read_varint(body, off)
        elif ty == WIRE_LEN:
            v, off = read_ld(body, off)
        else:
            raise ValueError(f"wire type {ty} cannot be skipped")
        yield kb, v


def _known(body: bytes, table: dict, required=(), repeated=(), raw1: int = 0) -> dict:
    """The decoders' rule — the reference's (types/join.rs:58-105 and its siblings): a field is opened by ONE key byte; a known key
    byte that comes twice is an error (DecodeError::duplicate_field) unless the field repeats, any other key byte is skipped by
    its wire type, and the fields the reference unwraps without a default must have come (DecodeError::missing_field)."""
    out = {name: [] for name in repeated}
    for kb, v in _fields(body, raw1):
        name = table.get(kb)
        if name is None:
            continue
        if name in repeated:
            out[name].append(v)
        elif name in out:
            raise ValueError(f"duplicate field {name}")
        else:
            out[name] = v
    for name in required:
        if name not in out:
            raise ValueError(f"missing field {name}")
    return out


def decode_message(buf: bytes):
    """Inverse of encode_message; returns (message, bytes consumed)."""
    if not buf:
        raise ValueError("empty buffer")
    ty, tag = split(buf[0])
    if ty != WIRE_LEN:
        raise ValueError("message type byte is not length-delimited")
    if tag == RELAY:   # no length of its own: node, then the wrapped message to the end of the buffer
        if len(buf) < 3 or buf[1] != merge(WIRE_LEN, 1):
            raise ValueError("relay message without a node")
        node, off = read_ld(buf, 2)
        if off >= len(buf) or buf[off] != merge(WIRE_LEN, 2):
            raise ValueError("relay message without a message")
        inner, used = decode_message(buf[off + 1:])
        return Relay(decode_node(node), inner), off + 1 + used
    body, end = read_ld(buf, 1)
    V, L, B = WIRE_VARINT, WIRE_LEN, WIRE_BYTE
    if tag == JOIN:
        d = _known(body, {KB(1, V): "ltime", KB(2, L): "id"}, required=("ltime", "id"))
        msg = Join(d["ltime"], parse_node_id(d["id"]))
    elif tag == LEAVE:
        d = _known(body, {KB(1, V): "ltime", KB(2, B): "prune", KB(3, L): "id"}, required=("ltime", "id"))
        msg = Leave(d["ltime"], parse_node_id(d["id"]), bool(d.get("prune", 0)))
    elif tag == USER_EVENT:
        d = _known(body, {KB(1, V): "ltime", KB(2, B): "cc", KB(3, L): "name", KB(4, L): "payload"}, required=("ltime",))
        msg = UserEvent(d["ltime"], d.get("name", b""), d.get("payload", b""), bool(d.get("cc", 0)))
    elif tag == QUERY:
        d = _known(body, {KB(1, V): "ltime", KB(2, V): "id", KB(3, L): "from", KB(4, L): "filters", KB(5, V): "flags", KB(6, V): "relay_factor",
                          KB(7, V): "timeout", KB(8, L): "name", KB(9, L): "payload"},
                   required=("ltime", "id", "from", "flags", "relay_factor", "timeout"), repeated=("filters",), raw1=KB(6, V))
        if d["id"] > 0xFFFFFFFF:
            raise ValueError("query id out of range")
        msg = Query(d["ltime"], d["id"], decode_node(d["from"]), d["flags"], d["relay_factor"], d["timeout"], d.get("name", b""),
                    d.get("payload", b""), d["filters"])
    elif tag == QUERY_RESPONSE:
        d = _known(body, {KB(1, V): "ltime", KB(2, V): "id", KB(3, L): "from", KB(4, V): "flags", KB(5, L): "payload"},
                   required=("ltime", "id", "from", "flags"))
        if d["id"] > 0xFFFFFFFF:
            raise ValueError("query id out of range")
        msg = QueryResponse(d["ltime"], d["id"], decode_node(d["from"]), d["flags"], d.get("payload", b""))
    elif tag == PUSH_PULL:
        d = _known(body, {KB(1, V): "ltime", KB(2, L): "status", KB(3, L): "left", KB(4, V): "event_ltime", KB(5, L): "events", KB(6, V): "query_ltime"},
                   required=("ltime", "event_ltime", "query_ltime"), repeated=("status", "left", "events"))
        msg = PushPull(d["ltime"], event_ltime=d["event_ltime"], query_ltime=d["query_ltime"])
        msg.status_list = []   # in first-insertion order; a repeated id keeps its place and takes the LAST value (the reference's IndexMap)
        for v in d["status"]:
            e = _known(v, {KB(1, L): "id", KB(2, V): "ltime"}, required=("id",))
            nid, lt = parse_node_id(e["id"]), e.get("ltime", 0)
            msg.status_ltimes[nid] = lt
        msg.status_list = list(msg.status_ltimes.items())
        msg.left_members = [parse_node_id(v) for v in d["left"]]
        for v in d["events"]:
            e = _known(v, {KB(1, V): "ltime", KB(2, L): "events"}, required=("ltime",), repeated=("events",))   # user_events.rs:102: missing_field
            evs = []
            for v2 in e["events"]:
                u = _known(v2, {KB(1, L): "name", KB(2, L): "payload"})
                evs.append((u.get("name", b""), u.get("payload", b"")))
            msg.events.append((e.get("ltime", 0), evs))
    elif tag == CONFLICT_RESPONSE:
        msg = ConflictResponse(body)
    else:
        raise ValueError(f"message tag {tag} is not on the simulated path")
    return msg, end


# ---- what the simulator is told -----------------------------------------------------------------------------------
def user_event_len(ltime: int, name: bytes, payload: bytes, cc: bool = False) -> int:
    """The `encoded_len` argument of sim_user_event / Serf::user_event: the framed wire length of the message."""
    return encoded_len(UserEvent(ltime, name, payload, cc))


def check_user_event_size(name: bytes, payload: bytes, limit: int = 512):
    """api.rs:246-262: `name.len() + payload.len() > user_event_size_limit` is refused before anything is encoded."""
    if len(name) + len(payload) > limit:
        raise ValueError(f"user event exceeds limit of {limit} bytes before encoding")
