// coalesce.hpp — event coalescing for the host-side event stream (serf-core/src/coalesce.rs, coalesce/member.rs,
// coalesce/user.rs) in simulation ticks, header-only; the native twin of serf_amd/coalesce.py.
//
// The stream is what Cluster::drain_events() returns for subscribed nodes (sim_event: tick, observer, type, key,
// ltime; type = MemberEventType Join 0, Leave 1, Failed 2, Update 3, Reap 4; 5 = user event, 6 = query).  A coalescer
// batches the events of ONE observer.  `coalesce_loop` follows coalesce.rs:66-155: an event the coalescer handles starts
// a quantum (`coalesce_period`) if none is running and restarts the quiescence timer (`quiescent_period`); when either
// expires everything coalesced so far is flushed; events it does not handle pass straight through.
#pragma once
#include <algorithm>
#include <cstdint>
#include <functional>
#include <map>
#include <utility>
#include <vector>

#include "../../include/serf_sim.h"

namespace serf {
namespace coalesce {

// One output item: a single event passed through / flushed (members empty), or a member batch (coalesce/member.rs
// flushes ONE event per type carrying all members of that type)
struct Out {
  uint32_t tick, observer, type, key;
  uint64_t ltime;
  std::vector<uint32_t> members;  // batch of a MemberEventCoalescer flush
  bool batch;
};
inline Out single(const sim_event& e) { return Out{e.tick, e.observer, e.type, e.key, e.ltime, {}, false}; }

// coalesce/member.rs:25-128: the latest event per member wins inside a window; a member whose latest event type equals
// the one reported last time is dropped, except for Update
class MemberEventCoalescer {
 public:
  bool handle(const sim_event& e) const { return e.type <= SIM_EV_REAP; }  // member.rs:52-54
  void coalesce(const sim_event& e) {                                     // member.rs:56-72 (latest wins, order of last sight)
    auto it = std::find_if(latest_.begin(), latest_.end(), [&](const std::pair<uint32_t, uint32_t>& p) { return p.first == e.key; });
    if (it != latest_.end()) latest_.erase(it);
    latest_.emplace_back(e.key, e.type);
  }
  std::vector<Out> flush(uint32_t tick, uint32_t observer) {              // member.rs:74-110
    std::vector<std::pair<uint32_t, std::vector<uint32_t>>> batches;      // in order of first appearance of the type
    for (auto& [member, ty] : latest_) {
      auto l = last_.find(member);
      if (l != last_.end() && l->second == ty && ty != SIM_EV_UPDATE) continue;
      last_[member] = ty;
      auto b = std::find_if(batches.begin(), batches.end(), [&](const std::pair<uint32_t, std::vector<uint32_t>>& p) { return p.first == ty; });
      if (b == batches.end()) { batches.emplace_back(ty, std::vector<uint32_t>{}); b = batches.end() - 1; }
      b->second.push_back(member);
    }
    latest_.clear();
    std::vector<Out> out;
    for (auto& [ty, members] : batches) out.push_back(Out{tick, observer, ty, 0, 0, members, true});
    return out;
  }

 private:
  std::map<uint32_t, uint32_t> last_;                     // member -> type reported by the previous flushes
  std::vector<std::pair<uint32_t, uint32_t>> latest_;     // (member, type) seen in the current window
};

// coalesce/user.rs:17-104: per event NAME only the events with the highest Lamport time survive a window.  The
// simulator identifies a user event by one 32-bit key for (name, payload); `name_of` maps a key to its name
// (default: the upper 24 bits), `is_cc` says whether an event asked to be coalesced (default: all do)
class UserEventCoalescer {
 public:
  std::function<uint32_t(uint32_t)> name_of = [](uint32_t key) { return key >> 8; };
  std::function<bool(const sim_event&)> is_cc = [](const sim_event&) { return true; };
  bool handle(const sim_event& e) const { return e.type == SIM_EV_USER && is_cc(e); }  // user.rs:45-50
  void coalesce(const sim_event& e) {                                                  // user.rs:52-83
    uint32_t name = name_of(e.key);
    auto it = std::find_if(events_.begin(), events_.end(), [&](const Slot& s) { return s.name == name; });
    if (it == events_.end()) events_.push_back(Slot{name, e.ltime, {e}});
    else if (it->ltime < e.ltime) { it->ltime = e.ltime; it->evs.assign(1, e); }
    else if (it->ltime == e.ltime) it->evs.push_back(e);
  }
  std::vector<Out> flush(uint32_t, uint32_t) {                                         // user.rs:85-103
    std::vector<Out> out;
    for (const Slot& s : events_)
      for (const sim_event& e : s.evs) out.push_back(single(e));
    events_.clear();
    return out;
  }

 private:
  struct Slot { uint32_t name; uint64_t ltime; std::vector<sim_event> evs; };
  std::vector<Slot> events_;  // in order of first sight of the name
};

// coalesce.rs:66-155 over one observer's tick-ordered stream; `end_tick` < 0: everything still held is flushed at its
// own deadline, otherwise only what is due by `end_tick`
template <class C>
inline std::vector<Out> coalesce_loop(const std::vector<sim_event>& events, C& coalescer, uint32_t coalesce_period,
                                      uint32_t quiescent_period, uint32_t observer, int64_t end_tick = -1) {
  std::vector<Out> out;
  bool running = false;
  uint64_t quantum = 0, quiescent = 0;
  auto expire = [&](bool bounded, uint64_t upto) {
    while (running) {
      uint64_t due = std::min(quantum, quiescent);
      if (bounded && due > upto) return;
      for (Out& o : coalescer.flush((uint32_t)due, observer)) out.push_back(std::move(o));
      running = false;
    }
  };
  for (const sim_event& e : events) {
    expire(true, e.tick);
    if (!coalescer.handle(e)) { out.push_back(single(e)); continue; }
    if (!running) { quantum = (uint64_t)e.tick + coalesce_period; running = true; }
    quiescent = (uint64_t)e.tick + quiescent_period;
    coalescer.coalesce(e);
  }
  expire(end_tick >= 0, end_tick >= 0 ? (uint64_t)end_tick : 0);
  return out;
}

}  // namespace coalesce
}  // namespace serf
