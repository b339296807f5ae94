// serf.hpp — C++ host-side mirror of the reference's `Serf` API (serf-core/src/serf/api.rs) for the
// bulk simulation path, header-only, on top of the C ABI of include/serf_sim.h.
//
// The reference's host language is Rust; this image has no Rust toolchain, so the host above the C
// ABI is C++ (the Rust binding a maintainer would add is in INTEGRATION.md).  Names and argument
// meaning follow the reference: a `Cluster` owns all N simulated nodes (one `Serf::new` each,
// api.rs:25), `Cluster::node(i)` is the `Serf` handle of node i and has the reference's methods.
// Errors: the reference returns `Result<_, Error>` (error.rs:64-81); here a `serf::Error` exception
// carries the SIM_E* code.
#pragma once
#include <algorithm>
#include <cstdint>
#include <functional>
#include <iterator>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/serf_sim.h"
#include "wire.hpp"

namespace serf {

struct Error : std::runtime_error {
  int code;
  Error(int c, const char* what) : std::runtime_error(std::string(what) + ": SIM error " + std::to_string(c)), code(c) {}
};
inline void check(int rc, const char* what) {
  if (rc < 0) throw Error(rc, what);
}

// MemberStatus — types/member.rs:54-87
enum class MemberStatus : uint8_t { None = 0, Alive = 1, Leaving = 2, Left = 3, Failed = 4 };
// MemberEventType — event.rs:263-279 (+ user / query events, event.rs:367-378)
enum class EventType : uint32_t { Join = 0, Leave = 1, Failed = 2, Update = 3, Reap = 4, User = 5, Query = 6 };

struct Member {  // types/member.rs:135-185 (id + status; tags/addresses are not on the simulated path)
  uint32_t id;
  MemberStatus status;
  uint64_t status_ltime;
};
using Stats = sim_stats;  // api.rs:586-602
using Event = sim_event;

// Options — options.rs:16-469 (the fields the simulated path reads) + MemberlistOptions::lan()/wan()
struct Options {
  sim_config c{};
  explicit Options(uint32_t n_nodes) {
    c.struct_size = sizeof(sim_config);
    c.n_nodes = n_nodes; c.vshards = 1; c.shard_rank = 0; c.shard_count = 1;
    c.fanout = 3;                                  // lan(): gossip_nodes
    c.view_slots = 0;
    c.event_ring = 512; c.query_ring = 512;        // options.rs:516-517
    c.retransmit_mult = 4; c.probe_interval = 5;   // lan(): 1 s / 200 ms
    c.suspicion_mult = 4; c.suspicion_max_mult = 6; c.indirect_checks = 3;
    c.loss_u32 = 0; c.intent_timeout = 0;
    c.leave_delay = 30;                            // broadcast_timeout 5 s + leave_propagate_delay 1 s
    c.reap_interval = 75;                          // options.rs:506 reap_interval 15 s
    c.reconnect_timeout = c.tombstone_timeout = 432000;  // 24 h (options.rs:507-508)
    c.queue_check_interval = 150; c.max_queue_depth = 4096; c.min_queue_depth = 0;  // options.rs:512-514
    c.push_pull_interval = 150;                    // lan(): 30 s
    c.reconnect_interval = 150;                    // options.rs:507 reconnect_interval 30 s (Reconnector, base.rs:612-681)
    c.ring_overflow = 8;                           // a de-dup bucket is a Vec in the reference (base.rs:801-813): 6 keys + 8 rows of 6 per ring here
    c.flags = SIM_CF_BASELINE_JOINED | SIM_CF_TCP_FALLBACK | SIM_CF_NACKS;  // memberlist lan(): disable_tcp_pings = false, nacks (protocol >= 4)
    c.seed = SIM_DEFAULT_SEED;
  }
  static Options lan(uint32_t n) { return Options(n); }
  static Options wan(uint32_t n) {  // gossip 500 ms, probe 5 s, fan-out 4, suspicion_mult 6 (k capped at 3)
    Options o(n);
    o.c.fanout = 4; o.c.probe_interval = 10; o.c.suspicion_mult = 6;
    return o;
  }
  Options& with_fanout(uint32_t f) { c.fanout = f; return *this; }
  Options& with_view_slots(uint32_t a) { c.view_slots = a; return *this; }
  Options& with_event_buffer_size(uint32_t b) { c.event_ring = b; return *this; }
  Options& with_query_buffer_size(uint32_t b) { c.query_ring = b; return *this; }
  Options& with_packet_loss(double p) { c.loss_u32 = p >= 1.0 ? 0xFFFFFFFFu : (uint32_t)(p * 4294967296.0); return *this; }
  Options& with_disable_tcp_pings(bool off) { c.flags = off ? (c.flags & ~SIM_CF_TCP_FALLBACK) : (c.flags | SIM_CF_TCP_FALLBACK); return *this; }  // memberlist Options::disable_tcp_pings
  Options& with_reconnect_interval(uint32_t ticks) { c.reconnect_interval = ticks; return *this; }  // options.rs:162 (0 = no Reconnector)
  Options& with_ring_overflow(uint32_t rows) { c.ring_overflow = rows; return *this; }  // overflow rows per de-dup ring and node (0: a full bucket treats new keys as seen)
  // handle_prune's wait (serf/base.rs:1628-1653): a pruning leave intent about a Leaving member erases it leave_delay ticks later — the reference's
  // behaviour; off by default (the erase in the tick of the intent), needs the SWIM layer (probe_interval > 0)
  Options& with_prune_wait(bool on) { c.flags = on ? (c.flags | SIM_CF_PRUNE_DELAY) : (c.flags & ~SIM_CF_PRUNE_DELAY); return *this; }
  Options& with_seed(uint64_t s) { c.seed = s; return *this; }
};

class Cluster;

// One simulated node: the reference's `Serf<T, D>` (serf.rs:171-254) for the simulated path.
class Serf {
 public:
  uint32_t local_id() const { return id_; }                                   // api.rs:100
  inline std::vector<Member> members() const;                                 // api.rs:136
  inline Stats stats() const;                                                 // api.rs:150
  inline size_t num_members() const { return stats().members; }               // api.rs:188
  // api.rs:241-299 with the reference's arguments: the size checks before and after encoding (UserEventLimitTooLarge /
  // UserEventTooLarge / RawUserEventTooLarge, error.rs:247-259 -> SIM_ETOOBIG), the message priced by the codec
  // (`encoded_message_len`, wire.hpp) at the node's current event clock; the simulator is told the 32-bit identity of
  // (name, payload) and the framed length.  `max_user_event_size` = options.rs:528 (512), hard limit serf.rs:44 (9 KiB).
  inline void user_event(const std::string& name, const std::vector<uint8_t>& payload, bool coalesce,
                         size_t max_user_event_size = 512);
  // the same with the identity and the length worked out by the caller
  inline void user_event(uint32_t event_key, uint32_t encoded_len = 32, bool coalesce = true);
  inline void query(uint32_t query_id, uint32_t flags = 0);                   // api.rs:304
  // QueryParam.filters (query.rs:37-93, should_process_query query.rs:439-521): `ids` = Filter::Id (at most
  // SIM_QF_IDS), `tag_mask` = the AND of the Filter::Tag masks over tag classes (0xFFFFFFFF = no tag filter)
  inline void query(uint32_t query_id, uint32_t flags, const std::vector<uint32_t>& ids, uint32_t tag_mask = 0xFFFFFFFFu);
  inline void set_tags(uint32_t tag_class);                                   // api.rs:219
  struct QueryStatus { uint64_t acks, responses; bool open; };                // QueryResponse, query.rs:117-303
  inline QueryStatus query_status(uint32_t query_id) const;
  // What the caller of Serf::query holds in the reference (QueryResponse, query.rs:117-236): who acked, who responded and
  // with what, each responder once (the `acks` / `responses` sets of handle_query_response, query.rs:240-303), until the
  // deadline or close().  The simulator tracks the responders (include/serf_sim.h sim_query_responders); a response's
  // PAYLOAD is what the responding node's user hands to respond() — user data, not simulated state — so it comes from the
  // responder function the caller registers (default: empty payloads).  Polling: every call returns the responders that
  // arrived since the previous call, like draining ack_rx / response_rx.
  struct NodeResponse { uint32_t from; std::vector<uint8_t> payload; };       // query.rs:373-385
  class QueryResponse {
   public:
    uint32_t id() const { return id_; }                                       // query.rs:129-133
    bool finished() const;                                                    // query.rs:215-219
    void close() { closed_ = true; }                                          // query.rs:223-237: no further deliveries
    std::vector<uint32_t> acks();                                             // ack_rx (query.rs:203-205)
    std::vector<NodeResponse> responses();                                    // response_rx (query.rs:209-212)
    void on_respond(std::function<std::vector<uint8_t>(uint32_t)> f) { payload_of_ = std::move(f); }
   private:
    friend class Serf;
    QueryResponse(Cluster* c, uint32_t id);   // (below Cluster: it notes the tick the query's deadline cannot outlast)
    std::vector<uint32_t> fresh(int which, std::vector<uint32_t>& seen);
    Cluster* c_;
    uint32_t id_;
    bool closed_ = false;
    mutable bool started_ = false;  // the tracker has been seen under this id (the tick that executes the query has run)
    uint64_t give_up_ = 0;          // a tick by which the query has certainly run out: its tick + 16 * digits10(N) (query.rs:421-427) and some
    std::vector<uint32_t> seen_acks_, seen_resp_;                             // ascending (QueryResponseCore.acks / .responses)
    std::function<std::vector<uint8_t>(uint32_t)> payload_of_;
  };
  // api.rs:304 as the reference shapes it: the query goes out and the caller keeps the QueryResponse
  inline QueryResponse query_response(uint32_t query_id, uint32_t flags = 0) { query(query_id, flags); return QueryResponse(c_, query_id); }
  inline void join(uint32_t peer);                                            // api.rs:318
  // api.rs:366-420: joins through the first of `peers` (memberlist.join_many contacts them in turn and a simulated peer
  // always answers); returns how many were contacted
  inline size_t join_many(const std::vector<uint32_t>& peers) { if (peers.empty()) return 0; join(peers.front()); return 1; }
  inline void leave();                                                        // api.rs:422
  inline void shutdown();                                                     // api.rs:525: the process stops (tests/serf/event.rs:112 shuts a node down)
  inline uint32_t state() const { return stats().serf_state; }                // api.rs:113 SerfState (serf.rs:80-89) as enum sim_serf_state
  inline void remove_failed_node(uint32_t id);                                // api.rs:505
  inline void remove_failed_node_prune(uint32_t id);                          // api.rs:513
  inline void subscribe();                                                    // EventSubscriber, event.rs:430-491

 private:
  friend class Cluster;
  Serf(Cluster* c, uint32_t id) : c_(c), id_(id) {}
  Cluster* c_;
  uint32_t id_;
};

class Cluster {
 public:
  explicit Cluster(const Options& o) : n_(o.c.n_nodes) { check(sim_create(&o.c, &h_), "sim_create"); }
  ~Cluster() { if (h_) sim_destroy(h_); }
  // Options::with_tags for every node at once (start-up tags; include/serf_sim.h sim_init_tags)
  void init_tags(const std::vector<uint8_t>& classes, uint32_t first = 0) {
    check(sim_init_tags(h_, first, (uint32_t)classes.size(), classes.data()), "sim_init_tags");
  }
  Cluster(const Cluster&) = delete;
  Cluster& operator=(const Cluster&) = delete;
  Serf node(uint32_t id) { return Serf(this, id); }
  uint32_t size() const { return n_; }
  void step(uint32_t ticks = 1) { check(sim_step(h_, ticks), "sim_step"); }   // the gossip loop
  void sync() { check(sim_sync(h_), "sim_sync"); }
  uint64_t tick() const { uint64_t t = 0; check(sim_tick(h_, &t), "sim_tick"); return t; }
  // churn / fault injection (the reference tests shutdown() nodes: tests/serf/event.rs:112)
  void crash(uint32_t node, uint64_t at_tick) { check(sim_inject(h_, at_tick, SIM_OP_CRASH, node, 0, 0), "sim_inject"); }
  void revive(uint32_t node, uint64_t at_tick) { check(sim_inject(h_, at_tick, SIM_OP_REVIVE, node, 0, 0), "sim_inject"); }
  std::vector<Event> drain_events() {
    std::vector<Event> ev(1 << 16);
    uint32_t n = 0;
    check(sim_drain_events(h_, ev.data(), (uint32_t)ev.size(), &n), "sim_drain_events");
    ev.resize(n);
    return ev;
  }
  // fraction of running nodes that have applied the rumour (rounds-to-99 % = first tick >= 0.99)
  double convergence(uint32_t kind, uint32_t key, uint64_t ltime) {
    uint64_t seen = 0, up = 0;
    check(sim_convergence(h_, kind, key, ltime, &seen, &up), "sim_convergence");
    return up ? (double)seen / (double)up : 0.0;
  }
  // checkpoint / resume (canonical image; snapshot.rs:117-126,228-347 is the per-node analogue)
  std::vector<uint8_t> snapshot() {
    size_t n = 0;
    check(sim_snapshot(h_, nullptr, 0, &n), "sim_snapshot");
    std::vector<uint8_t> img(n);
    check(sim_snapshot(h_, img.data(), img.size(), &n), "sim_snapshot");
    return img;
  }
  void restore(const std::vector<uint8_t>& img) { check(sim_restore(h_, img.data(), img.size()), "sim_restore"); }
  sim_handle* raw() { return h_; }

 private:
  sim_handle* h_ = nullptr;
  uint32_t n_;
};

inline std::vector<Member> Serf::members() const {
  std::vector<uint8_t> st(c_->size());
  std::vector<uint64_t> lt(c_->size());
  check(sim_members(c_->raw(), id_, st.data(), lt.data(), c_->size()), "sim_members");
  std::vector<Member> out;
  for (uint32_t i = 0; i < c_->size(); ++i)
    if (st[i] != 0) out.push_back(Member{i, (MemberStatus)st[i], lt[i]});
  return out;
}
inline Stats Serf::stats() const {
  Stats s;
  check(sim_stats_get(c_->raw(), id_, &s), "sim_stats_get");
  return s;
}
inline void Serf::user_event(uint32_t key, uint32_t len, bool cc) { check(sim_user_event(c_->raw(), id_, key, len, cc), "sim_user_event"); }
inline void Serf::user_event(const std::string& name, const std::vector<uint8_t>& payload, bool cc, size_t max_user_event_size) {
  constexpr size_t USER_EVENT_SIZE_LIMIT = 9 * 1024;  // serf.rs:44
  size_t before = name.size() + payload.size();
  if (before > max_user_event_size) throw Error(SIM_ETOOBIG, "user event exceeds configured limit before encoding");
  if (before > USER_EVENT_SIZE_LIMIT) throw Error(SIM_ETOOBIG, "user event exceeds sane limit before encoding");
  wire::Bytes nm(name.begin(), name.end());
  size_t len = wire::user_event_len(stats().event_time, nm, payload, cc);
  if (len > max_user_event_size || len > USER_EVENT_SIZE_LIMIT) throw Error(SIM_ETOOBIG, "encoded user event exceeds limit");
  user_event(wire::event_key(nm, payload), (uint32_t)len, cc);
}
inline void Serf::query(uint32_t qid, uint32_t flags) { check(sim_query(c_->raw(), id_, qid, flags), "sim_query"); }
inline void Serf::query(uint32_t qid, uint32_t flags, const std::vector<uint32_t>& ids, uint32_t tag_mask) {
  check(sim_query_filtered(c_->raw(), id_, qid, flags, ids.data(), (uint32_t)ids.size(), tag_mask), "sim_query_filtered");
}
inline void Serf::set_tags(uint32_t tag_class) { check(sim_set_tags(c_->raw(), id_, tag_class), "sim_set_tags"); }
inline Serf::QueryStatus Serf::query_status(uint32_t qid) const {
  QueryStatus st{0, 0, false};
  int open = 0;
  check(sim_query_status(c_->raw(), qid, &st.acks, &st.responses, &open), "sim_query_status");
  st.open = open != 0;
  return st;
}
// (SIM_EINVAL from the two tracker reads means "this id does not own its tracker entry": before the tick that executes the
// query has run — the channels exist and are empty, query.rs:201-212 — or after a newer query with the same residue took the
// entry over — the reference's receivers would simply be closed.  Neither is an error of the caller.)
inline bool Serf::QueryResponse::finished() const {
  if (closed_) return true;
  uint64_t a, r;
  int open = 0;
  const int rc = sim_query_status(c_->raw(), id_, &a, &r, &open);
  // not started yet: still open; evicted: finished — and an id that never gets to own its tracker entry (taken over by a newer query
  // of the same residue before this object ever saw it; an id that was never valid) is finished once its deadline is past, so that
  // `while (!r.finished()) step()` ends
  if (rc == SIM_EINVAL) return started_ || c_->tick() > give_up_;
  check(rc, "sim_query_status");
  started_ = true;
  return !open;
}
inline Serf::QueryResponse::QueryResponse(Cluster* c, uint32_t id) : c_(c), id_(id) {
  uint32_t n = c->size(), digits = 0;
  while (n) { ++digits; n /= 10; }
  give_up_ = c->tick() + 2u + 16u * digits;
}
inline std::vector<uint32_t> Serf::QueryResponse::fresh(int which, std::vector<uint32_t>& seen) {
  std::vector<uint32_t> out;
  if (closed_) return out;
  uint32_t n = 0;
  const int rc = sim_query_responders(c_->raw(), id_, which, nullptr, 0, &n);
  if (rc == SIM_EINVAL) return out;  // nobody yet (the query's tick has not run) or nobody any more (tracker taken over)
  check(rc, "sim_query_responders");
  started_ = true;
  std::vector<uint32_t> all(n);
  if (n) check(sim_query_responders(c_->raw(), id_, which, all.data(), n, &n), "sim_query_responders");
  std::set_difference(all.begin(), all.end(), seen.begin(), seen.end(), std::back_inserter(out));  // both ascending
  seen = std::move(all);
  return out;
}
inline std::vector<uint32_t> Serf::QueryResponse::acks() { return fresh(0, seen_acks_); }
inline std::vector<Serf::NodeResponse> Serf::QueryResponse::responses() {
  std::vector<NodeResponse> out;
  for (uint32_t from : fresh(1, seen_resp_)) out.push_back(NodeResponse{from, payload_of_ ? payload_of_(from) : std::vector<uint8_t>{}});
  return out;
}
inline void Serf::join(uint32_t peer) { check(sim_join(c_->raw(), id_, peer), "sim_join"); }
inline void Serf::leave() { check(sim_leave(c_->raw(), id_), "sim_leave"); }
inline void Serf::shutdown() { check(sim_inject(c_->raw(), c_->tick(), SIM_OP_CRASH, id_, 0, 0), "sim_inject"); }
inline void Serf::remove_failed_node(uint32_t id) { check(sim_force_leave(c_->raw(), id_, id, 0), "sim_force_leave"); }
inline void Serf::remove_failed_node_prune(uint32_t id) { check(sim_force_leave(c_->raw(), id_, id, 1), "sim_force_leave"); }
inline void Serf::subscribe() { check(sim_watch(c_->raw(), id_), "sim_watch"); }

}  // namespace serf
