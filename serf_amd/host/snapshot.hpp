// snapshot.hpp — reference-format snapshot of one simulated node, header-only; the native twin of
// serf_amd/snapshot.py.
//
// serf-core's `Snapshotter` (serf-core/src/snapshot.rs) is an event-stream consumer: every event the node's `Serf`
// delivers leaves a record in an append-only file, from which a restarted process recovers whom it knew to be alive and
// where its three Lamport clocks stood.  Record stream (snapshot.rs:117-126, 160-215): one type byte, then
//   0 Alive / 1 NotAlive   u32-LE length + encoded Node (Join; Leave or Failed member events, :682-698)
//   2 Clock / 3 EventClock / 4 QueryClock   u64-LE Lamport time
//   5 Coordinate / 7 Comment   nothing (off the simulated path);  6 Leave   nothing (:562-580)
// A user event / query appends EventClock / QueryClock only when its time is newer than the last one recorded
// (:659-679); after every member event `update_clock` appends Clock(clock.time() - 1) when that is newer (:706-713);
// `compact` rewrites the file as the live nodes followed by the three clocks (:780-830).  `replay` restates
// open_and_replay_snapshot (:228-347).  `Node` travels in wire.hpp's form.
#pragma once
#include <cstdint>
#include <set>
#include <stdexcept>
#include <vector>

#include "../../include/serf_sim.h"
#include "wire.hpp"

namespace serf {
namespace snapshot {

enum : uint8_t { ALIVE = 0, NOT_ALIVE = 1, CLOCK = 2, EVENT_CLOCK = 3, QUERY_CLOCK = 4, COORDINATE = 5, LEAVE = 6, COMMENT = 7 };
using Bytes = wire::Bytes;

inline void put_node_record(Bytes& out, uint8_t kind, uint32_t gid) {
  Bytes node = wire::encode_node(gid);
  out.push_back(kind);
  uint32_t n = (uint32_t)node.size();
  for (int i = 0; i < 4; ++i) out.push_back((uint8_t)(n >> (8 * i)));
  out.insert(out.end(), node.begin(), node.end());
}
inline void put_clock_record(Bytes& out, uint8_t kind, uint64_t t) {
  out.push_back(kind);
  for (int i = 0; i < 8; ++i) out.push_back((uint8_t)(t >> (8 * i)));
}

struct ReplayResult {
  std::set<uint32_t> alive_nodes;
  uint64_t last_clock = 0, last_event_clock = 0, last_query_clock = 0;
  size_t offset = 0;
};
// open_and_replay_snapshot (snapshot.rs:228-347)
inline ReplayResult replay(const Bytes& data, bool rejoin_after_leave = false) {
  ReplayResult r;
  size_t off = 0;
  while (off < data.size()) {
    uint8_t kind = data[off++];
    if (kind == ALIVE || kind == NOT_ALIVE) {
      if (data.size() - off < 4) throw std::invalid_argument("failed to replay snapshot: truncated node record");
      uint32_t n = 0;
      for (int i = 0; i < 4; ++i) n |= (uint32_t)data[off + i] << (8 * i);
      off += 4;
      if (n > data.size() - off) throw std::invalid_argument("failed to replay snapshot: truncated node record");
      uint32_t gid = wire::decode_node(Bytes(data.begin() + off, data.begin() + off + n));
      off += n;
      if (kind == ALIVE) r.alive_nodes.insert(gid);
      else r.alive_nodes.erase(gid);
    } else if (kind == CLOCK || kind == EVENT_CLOCK || kind == QUERY_CLOCK) {
      if (data.size() - off < 8) throw std::invalid_argument("failed to replay snapshot: truncated clock record");
      uint64_t t = 0;
      for (int i = 0; i < 8; ++i) t |= (uint64_t)data[off + i] << (8 * i);
      off += 8;
      (kind == CLOCK ? r.last_clock : kind == EVENT_CLOCK ? r.last_event_clock : r.last_query_clock) = t;
    } else if (kind == COORDINATE || kind == COMMENT) {
      continue;
    } else if (kind == LEAVE) {
      if (rejoin_after_leave) continue;  // "ignoring previous leave in snapshot"
      r.alive_nodes.clear();
      r.last_clock = r.last_event_clock = r.last_query_clock = 0;
    } else {
      throw std::invalid_argument("unrecognized snapshot record type");
    }
  }
  r.offset = data.size();
  return r;
}

// `Snapshot::stream` (snapshot.rs:585-655) over simulator events of one observer
class Snapshotter {
 public:
  explicit Snapshotter(uint32_t observer, bool rejoin_after_leave = false, const ReplayResult* from = nullptr)
      : observer_(observer), rejoin_after_leave_(rejoin_after_leave) {
    if (from) { alive_ = from->alive_nodes; last_clock_ = from->last_clock; last_event_clock_ = from->last_event_clock; last_query_clock_ = from->last_query_clock; }
  }
  void user_event(uint64_t ltime) {  // snapshot.rs:659-668; "stop recording events after a leave is issued" (:402)
    if (left_ || ltime <= last_event_clock_) return;
    last_event_clock_ = ltime;
    put_clock_record(buf_, EVENT_CLOCK, ltime);
  }
  void query(uint64_t ltime) {       // snapshot.rs:670-679
    if (left_ || ltime <= last_query_clock_) return;
    last_query_clock_ = ltime;
    put_clock_record(buf_, QUERY_CLOCK, ltime);
  }
  void member_event(uint32_t type, uint32_t subject, uint64_t clock_time) {  // snapshot.rs:682-704
    if (left_) return;
    if (type == SIM_EV_JOIN) { alive_.insert(subject); put_node_record(buf_, ALIVE, subject); }
    else if (type == SIM_EV_LEAVE || type == SIM_EV_FAILED) { alive_.erase(subject); put_node_record(buf_, NOT_ALIVE, subject); }
    update_clock(clock_time);
  }
  void update_clock(uint64_t clock_time) {  // snapshot.rs:706-713
    if (left_) return;
    uint64_t last_seen = clock_time ? clock_time - 1 : 0;
    if (last_seen > last_clock_) { last_clock_ = last_seen; put_clock_record(buf_, CLOCK, last_seen); }
  }
  void leave() {  // snapshot.rs:562-580: the record is always appended; only the live set depends on rejoin_after_leave
    left_ = true;
    if (!rejoin_after_leave_) alive_.clear();
    buf_.push_back(LEAVE);
  }
  // events of Cluster::drain_events() of this observer, in order; `clock_time` = the observer's Stats.member_time
  void feed(const std::vector<sim_event>& events, uint64_t clock_time) {
    for (const sim_event& e : events) {
      if (e.observer != observer_ || left_) continue;
      if (e.type == SIM_EV_USER) user_event(e.ltime);
      else if (e.type == SIM_EV_QUERY) query(e.ltime);
      else if (e.type == SIM_EV_JOIN || e.type == SIM_EV_LEAVE || e.type == SIM_EV_FAILED) member_event(e.type, e.key, clock_time);
    }
    update_clock(clock_time);
  }
  const Bytes& compact() {  // snapshot.rs:780-830: the live nodes, then the three clocks
    Bytes out;
    for (uint32_t gid : alive_) put_node_record(out, ALIVE, gid);
    put_clock_record(out, CLOCK, last_clock_);
    put_clock_record(out, EVENT_CLOCK, last_event_clock_);
    put_clock_record(out, QUERY_CLOCK, last_query_clock_);
    buf_ = std::move(out);
    return buf_;
  }
  const Bytes& bytes() const { return buf_; }

 private:
  uint32_t observer_;
  bool rejoin_after_leave_, left_ = false;
  std::set<uint32_t> alive_;
  uint64_t last_clock_ = 0, last_event_clock_ = 0, last_query_clock_ = 0;
  Bytes buf_;
};

}  // namespace snapshot
}  // namespace serf
