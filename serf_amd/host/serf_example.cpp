// serf_example.cpp — the reference's event tests, written against the C++ mirror of `Serf`:
// a user event reaches everybody, a crashed node is reported Failed, a leaving node Left
// (serf-core/src/serf/base/tests/serf/event.rs:88-232).  Needs an MI355X: `./serf_example [n_nodes]`.
#include <cstdio>
#include <cstdlib>

#include "serf.hpp"

int main(int argc, char** argv) {
  uint32_t n = argc > 1 ? (uint32_t)atoi(argv[1]) : 4096;
  try {
    serf::Cluster cl(serf::Options::lan(n).with_view_slots(64));
    serf::Serf s0 = cl.node(0), s7 = cl.node(7);
    s0.subscribe();
    s7.user_event(/*event_key=*/42, /*encoded_len=*/64);
    cl.node(9).leave();
    cl.crash(11, /*at_tick=*/2);
    uint32_t rounds = 0;
    while (cl.convergence(SIM_K_EVENT, 42, 1) < 0.99 && rounds < 200) { cl.step(); ++rounds; }
    printf("user event reached 99%% of %u nodes after %u rounds\n", n, rounds);
    // a query only the "web" nodes (tag class 1: every fourth node) answer, and of those only three by id
    std::vector<uint8_t> classes(n);
    for (uint32_t i = 0; i < n; ++i) classes[i] = i % 4 == 0 ? 1 : 2;
    cl.init_tags(classes);
    s7.query(/*query_id=*/7, SIM_F_ACK, /*ids=*/{0, 4, 5}, /*tag_mask=*/1u << 1);
    cl.step(60);
    serf::Serf::QueryStatus qs = s7.query_status(7);
    printf("filtered query: %llu acks (nodes 0 and 4: id listed and class 1)\n", (unsigned long long)qs.acks);
    if (qs.acks != 2) return 1;
    cl.step(1500);
    serf::Stats st = s0.stats();
    printf("node 0: members %u failed %u left %u, clocks %llu/%llu/%llu\n", st.members, st.failed, st.left,
           (unsigned long long)st.member_time, (unsigned long long)st.event_time, (unsigned long long)st.query_time);
    for (const serf::Event& e : cl.drain_events())
      printf("  tick %u observer %u event %u key %u ltime %llu\n", e.tick, e.observer, e.type, e.key, (unsigned long long)e.ltime);
    return st.failed == 1 && st.left == 1 ? 0 : 1;
  } catch (const serf::Error& e) {
    fprintf(stderr, "%s\n", e.what());
    return 2;
  }
}
