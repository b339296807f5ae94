// host_stream_test.cpp — serf_amd/host/coalesce.hpp and snapshot.hpp on an event stream read from stdin
// (one event per line: tick observer type key ltime); tests/test_host_cpp_streams.py feeds the same seeded streams to
// the Python modules (which restate coalesce/*.rs and snapshot.rs and carry the reference's own tests) and compares.
//   host_stream_test member <coalesce_period> <quiescent_period> <observer>
//   host_stream_test user   <coalesce_period> <quiescent_period> <observer>
//   host_stream_test snapshot <observer> <clock_time> <rejoin_after_leave> <leave_after_n_events>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../serf_amd/host/coalesce.hpp"
#include "../../serf_amd/host/snapshot.hpp"

using namespace serf;

static std::vector<sim_event> read_events() {
  std::vector<sim_event> ev;
  unsigned t, o, ty, k;
  unsigned long long lt;
  while (scanf("%u %u %u %u %llu", &t, &o, &ty, &k, &lt) == 5) {
    sim_event e;
    memset(&e, 0, sizeof e);
    e.tick = t; e.observer = o; e.type = ty; e.key = k; e.ltime = lt;
    ev.push_back(e);
  }
  return ev;
}
static void print(const std::vector<coalesce::Out>& out) {
  for (const coalesce::Out& o : out) {
    if (o.batch) {
      printf("B %u %u %u", o.tick, o.observer, o.type);
      for (uint32_t m : o.members) printf(" %u", m);
      printf("\n");
    } else {
      printf("E %u %u %u %u %llu\n", o.tick, o.observer, o.type, o.key, (unsigned long long)o.ltime);
    }
  }
}
int main(int argc, char** argv) {
  if (argc < 2) return 2;
  std::vector<sim_event> ev = read_events();
  if (!strcmp(argv[1], "member") || !strcmp(argv[1], "user")) {
    uint32_t cp = (uint32_t)atoi(argv[2]), qp = (uint32_t)atoi(argv[3]), obs = (uint32_t)atoi(argv[4]);
    if (argv[1][0] == 'm') { coalesce::MemberEventCoalescer c; print(coalesce::coalesce_loop(ev, c, cp, qp, obs)); }
    else { coalesce::UserEventCoalescer c; print(coalesce::coalesce_loop(ev, c, cp, qp, obs)); }
    return 0;
  }
  if (!strcmp(argv[1], "snapshot")) {
    uint32_t obs = (uint32_t)atoi(argv[2]);
    uint64_t clock_time = strtoull(argv[3], nullptr, 10);
    bool rejoin = atoi(argv[4]) != 0;
    size_t leave_after = (size_t)atoi(argv[5]);
    snapshot::Snapshotter s(obs, rejoin);
    std::vector<sim_event> first(ev.begin(), ev.begin() + std::min(leave_after, ev.size())), rest(ev.begin() + std::min(leave_after, ev.size()), ev.end());
    s.feed(first, clock_time);
    if (leave_after < ev.size()) { s.leave(); s.feed(rest, clock_time + 5); }
    auto hex = [](const char* what, const snapshot::Bytes& b) { printf("%s ", what); for (uint8_t x : b) printf("%02x", x); printf("\n"); };
    hex("stream", s.bytes());
    snapshot::ReplayResult r = snapshot::replay(s.bytes(), rejoin);
    printf("replay %llu %llu %llu", (unsigned long long)r.last_clock, (unsigned long long)r.last_event_clock, (unsigned long long)r.last_query_clock);
    for (uint32_t g : r.alive_nodes) printf(" %u", g);
    printf("\n");
    snapshot::Bytes compacted = s.compact();
    hex("compact", compacted);
    snapshot::ReplayResult r2 = snapshot::replay(compacted, rejoin);
    printf("replay_compact %llu %llu %llu %zu\n", (unsigned long long)r2.last_clock, (unsigned long long)r2.last_event_clock,
           (unsigned long long)r2.last_query_clock, r2.alive_nodes.size());
    bool threw = false;
    try { snapshot::Bytes bad = s.bytes(); bad.push_back(42); snapshot::replay(bad); } catch (const std::invalid_argument&) { threw = true; }
    printf("bad_record_refused %d\n", threw ? 1 : 0);
    return 0;
  }
  return 2;
}
