// wire.hpp — C++ codec of the serf messages on the simulated path, header-only.
//
// The native twin of serf_amd/wire.py: the byte layout of serf-core's message types, so that the C++ host
// (serf.hpp) prices a user event or a query the way the reference does (`encoded_message_len`, api.rs:241-299) and
// can turn a simulated record into the bytes a real `serf` process would accept, and back.
//
//   framing       types/message.rs:397-428   type byte merge(LengthDelimited, TAG), varint body length, body;
//                                            TAGs leave 1, join 2, push_pull 3, user_event 4, query 5 (message.rs:17-24)
//   JoinMessage   types/join.rs:123-158      LTIME varint | ID <id>                         tags 1, 2
//   LeaveMessage  types/leave.rs:138-195     LTIME varint | [PRUNE 0x01] | ID <id>          tags 1, 2, 3
//   UserEvent     types/user_event/message.rs:205-272
//                                            LTIME varint | [CC 0x01] | [NAME <..>] | [PAYLOAD <..>]   tags 1..4
//   QueryMessage  types/query.rs:404-527     LTIME | ID varint(u32) | FROM <node> | (FILTERS <..>)* | FLAGS varint |
//                                            RELAY_FACTOR one raw byte | TIMEOUT varint | [NAME] | [PAYLOAD]   tags 1..9
//
// memberlist-proto (memberlist-core 0.8.1, not vendored) defines the bits of `merge(wire_type, tag)`, the WireType
// values and the encoding of Node / Duration: UPSTREAM-RECALL assumptions, isolated in `merge`, `encode_node`,
// `WIRE_*` exactly as in wire.py — no LENGTH depends on them (every tag is one byte whatever its bits are).
// A simulated node id travels as the UTF-8 decimal string of its number.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <algorithm>
#include <vector>

namespace serf {
namespace wire {

using Bytes = std::vector<uint8_t>;

enum : uint8_t { WIRE_BYTE = 0, WIRE_VARINT = 1, WIRE_LEN = 2 };
enum : uint8_t { LEAVE = 1, JOIN = 2, PUSH_PULL = 3, USER_EVENT = 4, QUERY = 5, QUERY_RESPONSE = 6, CONFLICT_RESPONSE = 7, RELAY = 8 };

inline uint8_t merge(uint8_t wire_type, uint8_t tag) { return (uint8_t)((tag << 3) | wire_type); }
inline std::pair<uint8_t, uint8_t> split(uint8_t b) { return {(uint8_t)(b & 7), (uint8_t)(b >> 3)}; }

inline void put_varint(Bytes& out, uint64_t v) {
  for (;;) {
    uint8_t b = v & 0x7F;
    v >>= 7;
    if (v) out.push_back(b | 0x80);
    else { out.push_back(b); return; }
  }
}
inline size_t varint_len(uint64_t v) {
  size_t n = 1;
  while (v >= 0x80) { v >>= 7; ++n; }
  return n;
}
inline uint64_t read_varint(const Bytes& buf, size_t& off) {
  uint64_t v = 0;
  for (unsigned shift = 0;; shift += 7) {
    if (off >= buf.size()) throw std::invalid_argument("truncated varint");
    if (shift > 63) throw std::invalid_argument("varint too long");
    uint8_t b = buf[off++];
    v |= (uint64_t)(b & 0x7F) << shift;
    if (!(b & 0x80)) return v;
  }
}
inline void put_ld(Bytes& out, const uint8_t* p, size_t n) {  // length-delimited
  put_varint(out, n);
  out.insert(out.end(), p, p + n);
}
inline void put_ld(Bytes& out, const Bytes& b) { put_ld(out, b.data(), b.size()); }
inline void put_ld(Bytes& out, const std::string& s) { put_ld(out, (const uint8_t*)s.data(), s.size()); }
inline Bytes read_ld(const Bytes& buf, size_t& off) {
  uint64_t n = read_varint(buf, off);
  if (n > buf.size() - off) throw std::invalid_argument("truncated length-delimited field");
  Bytes out(buf.begin() + off, buf.begin() + off + n);
  off += n;
  return out;
}

inline std::string node_id(uint32_t gid) { return std::to_string(gid); }
inline uint32_t parse_node_id(const Bytes& b) {
  if (b.empty()) throw std::invalid_argument("empty node id");
  uint64_t v = 0;
  for (uint8_t c : b) {
    if (c < '0' || c > '9') throw std::invalid_argument("node id is not a decimal number");
    v = v * 10 + (c - '0');
    if (v > 0xFFFFFFFFull) throw std::invalid_argument("node id out of range");
  }
  return (uint32_t)v;
}
// Node<I, A> of a simulated node: id + a 6-byte socket address (assumption: {id: tag 1, addr: tag 2})
inline Bytes encode_node(uint32_t gid) {
  Bytes out;
  out.push_back(merge(WIRE_LEN, 1));
  put_ld(out, node_id(gid));
  const uint8_t addr[6] = {10, (uint8_t)(gid >> 16), (uint8_t)(gid >> 8), (uint8_t)gid, (uint8_t)(7946 >> 8), (uint8_t)(7946 & 0xFF)};
  out.push_back(merge(WIRE_LEN, 2));
  put_ld(out, addr, 6);
  return out;
}
// ---- decoding, by the reference's rule (types/join.rs:58-105 and its siblings): a body is a run of fields, each opened by ONE key
// byte (tag << 3 | wire type); a decoder knows the key bytes of its message — a known one that comes twice is an error
// (DecodeError::duplicate_field), any other key byte is skipped by its wire type (Byte: one raw byte; Varint; LengthDelimited;
// anything else cannot be skipped: error), and the fields the reference unwraps without a default must have come
// (DecodeError::missing_field).
constexpr uint8_t KB(unsigned tag, unsigned wt) { return (uint8_t)((tag << 3) | wt); }
struct Field {
  uint8_t kb = 0;  // the key byte
  uint8_t tag = 0;
  uint64_t v = 0;  // varint / byte fields
  Bytes data;      // length-delimited fields
};
// raw1: a key byte whose value is ONE raw byte although its wire type says Varint (QueryMessage.relay_factor,
// types/query.rs:484-490); 0: none
inline std::vector<Field> fields(const Bytes& body, uint8_t raw1 = 0) {
  std::vector<Field> out;
  size_t off = 0;
  while (off < body.size()) {
    Field f;
    f.kb = body[off++];
    f.tag = (uint8_t)(f.kb >> 3);
    const uint8_t ty = f.kb & 7;
    if ((raw1 && f.kb == raw1) || ty == WIRE_BYTE) {
      if (off >= body.size()) throw std::invalid_argument("truncated byte field");
      f.v = body[off++];
    } else if (ty == WIRE_VARINT) f.v = read_varint(body, off);
    else if (ty == WIRE_LEN) f.data = read_ld(body, off);
    else throw std::invalid_argument("a wire type that cannot be skipped");
    out.push_back(std::move(f));
  }
  return out;
}
inline void once(uint32_t& seen, uint32_t bit) {
  if (seen & bit) throw std::invalid_argument("duplicate field");
  seen |= bit;
}
inline void need(uint32_t seen, uint32_t mask) {
  if ((seen & mask) != mask) throw std::invalid_argument("missing field");
}
inline uint32_t decode_node(const Bytes& buf) {
  uint32_t gid = 0, seen = 0;
  for (const Field& f : fields(buf)) {
    if (f.kb == KB(1, WIRE_LEN)) { once(seen, 1); gid = parse_node_id(f.data); }
    else if (f.kb == KB(2, WIRE_LEN)) once(seen, 2);
  }
  need(seen, 1);
  return gid;
}

struct Join {
  uint64_t ltime = 0;
  uint32_t id = 0;
  Bytes body() const {
    Bytes out;
    out.push_back(merge(WIRE_VARINT, 1)); put_varint(out, ltime);
    out.push_back(merge(WIRE_LEN, 2)); put_ld(out, node_id(id));
    return out;
  }
  static constexpr uint8_t TAG = JOIN;
};
struct Leave {
  uint64_t ltime = 0;
  uint32_t id = 0;
  bool prune = false;
  Bytes body() const {
    Bytes out;
    out.push_back(merge(WIRE_VARINT, 1)); put_varint(out, ltime);
    if (prune) { out.push_back(merge(WIRE_BYTE, 2)); out.push_back(1); }
    out.push_back(merge(WIRE_LEN, 3)); put_ld(out, node_id(id));
    return out;
  }
  static constexpr uint8_t TAG = LEAVE;
};
struct UserEvent {
  uint64_t ltime = 0;
  Bytes name, payload;
  bool cc = false;
  Bytes body() const {
    Bytes out;
    out.push_back(merge(WIRE_VARINT, 1)); put_varint(out, ltime);
    if (cc) { out.push_back(merge(WIRE_BYTE, 2)); out.push_back(1); }
    if (!name.empty()) { out.push_back(merge(WIRE_LEN, 3)); put_ld(out, name); }
    if (!payload.empty()) { out.push_back(merge(WIRE_LEN, 4)); put_ld(out, payload); }
    return out;
  }
  static constexpr uint8_t TAG = USER_EVENT;
};
struct Query {
  uint64_t ltime = 0;
  uint32_t id = 0, from_node = 0, flags = 0;
  uint8_t relay_factor = 0;
  uint64_t timeout_ms = 0;
  Bytes name, payload;
  std::vector<Bytes> filters;
  Bytes body() const {
    Bytes out;
    out.push_back(merge(WIRE_VARINT, 1)); put_varint(out, ltime);
    out.push_back(merge(WIRE_VARINT, 2)); put_varint(out, id);
    out.push_back(merge(WIRE_LEN, 3)); put_ld(out, encode_node(from_node));
    for (const Bytes& f : filters) { out.push_back(merge(WIRE_LEN, 4)); put_ld(out, f); }
    out.push_back(merge(WIRE_VARINT, 5)); put_varint(out, flags);
    // types/query.rs:484-490: the tag says Varint, the value is ONE raw byte (`buf[offset] = self.relay_factor`)
    out.push_back(merge(WIRE_VARINT, 6)); out.push_back(relay_factor);
    out.push_back(merge(WIRE_VARINT, 7)); put_varint(out, timeout_ms);
    if (!name.empty()) { out.push_back(merge(WIRE_LEN, 8)); put_ld(out, name); }
    if (!payload.empty()) { out.push_back(merge(WIRE_LEN, 9)); put_ld(out, payload); }
    return out;
  }
  static constexpr uint8_t TAG = QUERY;
};

// types/message.rs:397-428: type byte, varint body length, body
template <class M>
inline Bytes encode_message(const M& m) {
  Bytes body = m.body(), out;
  out.push_back(merge(WIRE_LEN, M::TAG));
  put_varint(out, body.size());
  out.insert(out.end(), body.begin(), body.end());
  return out;
}
template <class M>
inline size_t encoded_len(const M& m) {  // crate::types::encoded_message_len
  size_t body = m.body().size();
  return 1 + varint_len(body) + body;
}

// The framed message at the head of `buf`: its TAG and body; `consumed` = bytes used
inline std::pair<uint8_t, Bytes> unframe(const Bytes& buf, size_t& consumed) {
  if (buf.empty()) throw std::invalid_argument("empty buffer");
  auto [ty, tag] = split(buf[0]);
  if (ty != WIRE_LEN) throw std::invalid_argument("message type byte is not length-delimited");
  size_t off = 1;
  Bytes body = read_ld(buf, off);
  consumed = off;
  return {tag, std::move(body)};
}
inline Join decode_join(const Bytes& body) {  // types/join.rs:58-105: ltime, id — both required
  Join m;
  uint32_t seen = 0;
  for (const Field& f : fields(body)) {
    if (f.kb == KB(1, WIRE_VARINT)) { once(seen, 1); m.ltime = f.v; }
    else if (f.kb == KB(2, WIRE_LEN)) { once(seen, 2); m.id = parse_node_id(f.data); }
  }
  need(seen, 3);
  return m;
}
inline Leave decode_leave(const Bytes& body) {  // types/leave.rs:60-118: ltime, prune (optional), id
  Leave m;
  uint32_t seen = 0;
  for (const Field& f : fields(body)) {
    if (f.kb == KB(1, WIRE_VARINT)) { once(seen, 1); m.ltime = f.v; }
    else if (f.kb == KB(2, WIRE_BYTE)) { once(seen, 2); m.prune = f.v != 0; }
    else if (f.kb == KB(3, WIRE_LEN)) { once(seen, 4); m.id = parse_node_id(f.data); }
  }
  need(seen, 5);
  return m;
}
inline UserEvent decode_user_event(const Bytes& body) {  // types/user_event/message.rs:100-190: ltime required; cc, name, payload
  UserEvent m;
  uint32_t seen = 0;
  for (const Field& f : fields(body)) {
    if (f.kb == KB(1, WIRE_VARINT)) { once(seen, 1); m.ltime = f.v; }
    else if (f.kb == KB(2, WIRE_BYTE)) { once(seen, 2); m.cc = f.v != 0; }
    else if (f.kb == KB(3, WIRE_LEN)) { once(seen, 4); m.name = f.data; }
    else if (f.kb == KB(4, WIRE_LEN)) { once(seen, 8); m.payload = f.data; }
  }
  need(seen, 1);
  return m;
}
inline Query decode_query(const Bytes& body) {  // types/query.rs:200-370: ltime, id, from, flags, relay_factor, timeout required
  Query m;
  uint32_t seen = 0;
  for (const Field& f : fields(body, KB(6, WIRE_VARINT))) {
    switch (f.kb) {
      case KB(1, WIRE_VARINT): once(seen, 1); m.ltime = f.v; break;
      case KB(2, WIRE_VARINT): once(seen, 2); if (f.v > 0xFFFFFFFFull) throw std::invalid_argument("query id out of range"); m.id = (uint32_t)f.v; break;
      case KB(3, WIRE_LEN): once(seen, 4); m.from_node = decode_node(f.data); break;
      case KB(4, WIRE_LEN): m.filters.push_back(f.data); break;
      case KB(5, WIRE_VARINT): once(seen, 16); m.flags = (uint32_t)f.v; break;
      case KB(6, WIRE_VARINT): once(seen, 32); m.relay_factor = (uint8_t)f.v; break;
      case KB(7, WIRE_VARINT): once(seen, 64); m.timeout_ms = f.v; break;
      case KB(8, WIRE_LEN): once(seen, 128); m.name = f.data; break;
      case KB(9, WIRE_LEN): once(seen, 256); m.payload = f.data; break;
      default: break;
    }
  }
  need(seen, 1 | 2 | 4 | 16 | 32 | 64);
  return m;
}

// QueryResponseMessage (types/query/response.rs:9-19): ltime 1, id 2, from 3 (Node), flags 4 (QueryFlag: ACK = 1), payload 5
struct QueryResponse {
  uint64_t ltime = 0;
  uint32_t id = 0, from_node = 0, flags = 0;
  Bytes payload;
};
inline QueryResponse decode_query_response(const Bytes& body) {  // types/query/response.rs:100-243: ltime, id, from, flags required
  QueryResponse m;
  uint32_t seen = 0;
  for (const Field& f : fields(body)) {
    switch (f.kb) {
      case KB(1, WIRE_VARINT): once(seen, 1); m.ltime = f.v; break;
      case KB(2, WIRE_VARINT): once(seen, 2); if (f.v > 0xFFFFFFFFull) throw std::invalid_argument("query id out of range"); m.id = (uint32_t)f.v; break;
      case KB(3, WIRE_LEN): once(seen, 4); m.from_node = decode_node(f.data); break;
      case KB(4, WIRE_VARINT): once(seen, 8); m.flags = (uint32_t)f.v; break;
      case KB(5, WIRE_LEN): once(seen, 16); m.payload = f.data; break;
      default: break;
    }
  }
  need(seen, 15);
  return m;
}
// A Relay (types/message.rs:431-470) at the head of `buf` — RELAY_MESSAGE_BYTE, RELAY_NODE_BYTE <node>, RELAY_MSG_BYTE, then the
// wrapped message framed as usual to the end of the buffer (no length of its own): the node to forward to and where the
// wrapped message starts
inline std::pair<uint32_t, size_t> unwrap_relay(const Bytes& buf) {
  if (buf.size() < 3 || buf[0] != merge(WIRE_LEN, RELAY) || buf[1] != merge(WIRE_LEN, 1)) throw std::invalid_argument("not a relay message");
  size_t off = 2;
  Bytes node = read_ld(buf, off);
  if (off >= buf.size() || buf[off] != merge(WIRE_LEN, 2)) throw std::invalid_argument("relay message without a message");
  return {decode_node(node), off + 1};
}
// PushPullMessage (types/push_pull.rs): ltime 1, status_ltimes 2 {id 1, ltime 2}, left_members 3, event_ltime 4, events 5
// {ltime 1, events 2 {name 1, payload 2}}, query_ltime 6 — in message order
struct PushPull {
  uint64_t ltime = 0, event_ltime = 0, query_ltime = 0;
  std::vector<std::pair<uint32_t, uint64_t>> status_ltimes;
  std::vector<uint32_t> left_members;
  std::vector<std::pair<uint64_t, std::vector<std::pair<Bytes, Bytes>>>> events;
};
inline PushPull decode_push_pull(const Bytes& body) {  // types/push_pull.rs:150-320: the three clocks required
  PushPull m;
  uint32_t seen = 0;
  for (const Field& f : fields(body)) {
    switch (f.kb) {
      case KB(1, WIRE_VARINT): once(seen, 1); m.ltime = f.v; break;
      case KB(4, WIRE_VARINT): once(seen, 2); m.event_ltime = f.v; break;
      case KB(6, WIRE_VARINT): once(seen, 4); m.query_ltime = f.v; break;
      case KB(2, WIRE_LEN): {  // one entry of the status map: {id, ltime}
        uint32_t id = 0, es = 0;
        uint64_t lt = 0;
        for (const Field& g : fields(f.data)) {
          if (g.kb == KB(1, WIRE_LEN)) { once(es, 1); id = parse_node_id(g.data); }
          else if (g.kb == KB(2, WIRE_VARINT)) { once(es, 2); lt = g.v; }
        }
        need(es, 1);
        {  // the reference's map is an IndexMap (types/push_pull.rs): a repeated id keeps its place and takes the LAST value
          auto it = std::find_if(m.status_ltimes.begin(), m.status_ltimes.end(), [&](const std::pair<uint32_t, uint64_t>& e) { return e.first == id; });
          if (it != m.status_ltimes.end()) it->second = lt; else m.status_ltimes.emplace_back(id, lt);
        }
        break;
      }
      case KB(3, WIRE_LEN): m.left_members.push_back(parse_node_id(f.data)); break;
      case KB(5, WIRE_LEN): {  // UserEvents{ltime, events {name, payload}} (types/user_event.rs): one bucket of the event buffer
        uint64_t lt = 0;
        uint32_t bs = 0;
        std::vector<std::pair<Bytes, Bytes>> evs;
        for (const Field& g : fields(f.data)) {
          if (g.kb == KB(1, WIRE_VARINT)) { once(bs, 1); lt = g.v; }
          else if (g.kb == KB(2, WIRE_LEN)) {
            Bytes name, payload;
            uint32_t us = 0;
            for (const Field& e : fields(g.data)) {
              if (e.kb == KB(1, WIRE_LEN)) { once(us, 1); name = e.data; }
              else if (e.kb == KB(2, WIRE_LEN)) { once(us, 2); payload = e.data; }
            }
            evs.emplace_back(std::move(name), std::move(payload));
          }
        }
        need(bs, 1);  // types/user_event/user_events.rs:102: DecodeError::missing_field("UserEvents", "ltime")
        m.events.emplace_back(lt, std::move(evs));
        break;
      }
      default: break;
    }
  }
  need(seen, 7);
  return m;
}

// What the simulator is told about a user event: the 32-bit identity of (name, payload) its de-dup ring compares
// (base.rs:783-813 compares name and payload; FNV-1a over name, a separator, payload; never 0 = "no key") and the
// framed wire length TransmitLimitedQueue sorts and budgets by.
inline uint32_t event_key(const Bytes& name, const Bytes& payload) {
  uint32_t h = 2166136261u;
  auto eat = [&](uint8_t b) { h = (h ^ b) * 16777619u; };
  for (uint8_t b : name) eat(b);
  eat(0xFF);
  for (uint8_t b : payload) eat(b);
  return h ? h : 1u;
}
inline size_t user_event_len(uint64_t ltime, const Bytes& name, const Bytes& payload, bool cc) {
  UserEvent m;
  m.ltime = ltime; m.name = name; m.payload = payload; m.cc = cc;
  return encoded_len(m);
}

}  // namespace wire
}  // namespace serf
