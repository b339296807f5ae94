// wire_test.cpp — host-logic test of serf_amd/host/wire.hpp and of Serf::user_event(name, payload, coalesce)
// (serf_amd/host/serf.hpp).  No GPU: tests/test_host_cpp_wire.py compiles it with every sim_* entry point renamed to
// the CPU oracle's osim_* (the same C ABI; the oracle is the stand-in backend of a host-logic test, nothing else), runs
// it, and compares the printed encodings with serf_amd/wire.py byte for byte.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../serf_amd/host/serf.hpp"

using namespace serf;
using wire::Bytes;

static void hex(const char* what, const Bytes& b) {
  printf("%s ", what);
  for (uint8_t x : b) printf("%02x", x);
  printf("\n");
}
static Bytes bytes(const char* s) { return Bytes(s, s + strlen(s)); }
#define REQUIRE(x) do { if (!(x)) { fprintf(stderr, "REQUIRE failed: %s (line %d)\n", #x, __LINE__); return 1; } } while (0)

int main() {
  // ---- encodings (compared with wire.py by the caller) ----
  wire::Join j; j.ltime = 1; j.id = 0;
  hex("join_1_0", wire::encode_message(j));
  j.ltime = 300; j.id = 1048575;
  hex("join_300_1048575", wire::encode_message(j));
  wire::Leave l; l.ltime = 12345678901ull; l.id = 77;
  hex("leave_big_77", wire::encode_message(l));
  l.prune = true;
  hex("leave_big_77_prune", wire::encode_message(l));
  wire::UserEvent e; e.ltime = 5;
  hex("event_empty", wire::encode_message(e));
  e.name = bytes("deploy"); e.payload = bytes("v1.2.3"); e.cc = true;
  hex("event_deploy_cc", wire::encode_message(e));
  e.cc = false; e.payload = Bytes(300, 0xAB);
  hex("event_deploy_300", wire::encode_message(e));
  wire::Query q; q.ltime = 9; q.id = 0xDEADBEEFu; q.from_node = 4242; q.flags = 3; q.relay_factor = 200; q.timeout_ms = 16000;
  q.name = bytes("ping"); q.payload = bytes("x");
  hex("query_full", wire::encode_message(q));
  q.filters.push_back(bytes("f1")); q.filters.push_back(bytes("filter-two")); q.name.clear(); q.payload.clear(); q.relay_factor = 0;
  hex("query_filters", wire::encode_message(q));
  printf("len_event_deploy_300 %zu\n", wire::encoded_len(e));
  printf("len_query_filters %zu\n", wire::encoded_len(q));
  printf("key_deploy %u\n", wire::event_key(bytes("deploy"), bytes("v1.2.3")));

  // ---- round trips (types/tests.rs:27-110 style) ----
  {
    size_t used = 0;
    Bytes buf = wire::encode_message(l);
    auto [tag, body] = wire::unframe(buf, used);
    REQUIRE(tag == wire::LEAVE && used == buf.size());
    wire::Leave l2 = wire::decode_leave(body);
    REQUIRE(l2.ltime == l.ltime && l2.id == l.id && l2.prune);
    buf = wire::encode_message(q);
    auto [tag2, body2] = wire::unframe(buf, used);
    REQUIRE(tag2 == wire::QUERY && used == buf.size());
    wire::Query q2 = wire::decode_query(body2);
    REQUIRE(q2.ltime == 9 && q2.id == 0xDEADBEEFu && q2.from_node == 4242 && q2.flags == 3 && q2.timeout_ms == 16000);
    REQUIRE(q2.filters.size() == 2 && q2.filters[1] == bytes("filter-two") && q2.name.empty());
    buf = wire::encode_message(e);
    auto [tag3, body3] = wire::unframe(buf, used);
    wire::UserEvent e2 = wire::decode_user_event(body3);
    REQUIRE(tag3 == wire::USER_EVENT && e2.ltime == 5 && e2.name == bytes("deploy") && e2.payload.size() == 300 && !e2.cc);
    bool threw = false;
    try { Bytes cut(buf.begin(), buf.begin() + 10); size_t u; wire::unframe(cut, u); } catch (const std::invalid_argument&) { threw = true; }
    REQUIRE(threw);  // a truncated message is refused, not read past its end
  }

  // ---- Serf::user_event(name, payload, coalesce): api.rs:241-299 ----
  try {
    Cluster cl(Options::lan(256).with_view_slots(16));
    Serf s3 = cl.node(3);
    Bytes payload = bytes("v1.2.3");
    uint64_t t0 = s3.stats().event_time;
    s3.user_event("deploy", payload, true);
    cl.step();                                                                 // (operations execute at the start of a tick)
    REQUIRE(s3.stats().event_time == t0 + 1);                                  // event_clock.increment()
    uint32_t key = wire::event_key(bytes("deploy"), payload);
    uint32_t rounds = 1;
    while (cl.convergence(SIM_K_EVENT, key, t0) < 0.99 && rounds < 100) { cl.step(); ++rounds; }
    REQUIRE(rounds >= 3 && rounds <= 12);                                      // log_3(256) = 5 rounds of pure tripling
    printf("event_rounds %u\n", rounds);
    int code = 0;
    try { s3.user_event(std::string(500, 'n'), Bytes(20, 1), false); } catch (const Error& er) { code = er.code; }
    REQUIRE(code == SIM_ETOOBIG);                                              // UserEventLimitTooLarge: 520 > 512 before encoding
    code = 0;
    try { s3.user_event(std::string(505, 'n'), Bytes(5, 1), false); } catch (const Error& er) { code = er.code; }
    REQUIRE(code == SIM_ETOOBIG);                                              // RawUserEventTooLarge: 510 fits, its encoding does not
    code = 0;
    try { s3.user_event(std::string(400, 'n'), Bytes(90, 1), false); } catch (const Error& er) { code = er.code; }
    REQUIRE(code == 0);                                                        // 490 + framing <= 512
    cl.step();
    REQUIRE(s3.stats().event_time == t0 + 2);
    // ---- Serf::query -> QueryResponse (api.rs:304, query.rs:117-303): who acked, who responded, with what ----
    Serf s9 = cl.node(9);
    Serf::QueryResponse qr = s9.query_response(77, SIM_F_ACK | SIM_F_RESPOND);
    qr.on_respond([](uint32_t from) { return Bytes{(uint8_t)(from & 0xFF), (uint8_t)(from >> 8)}; });  // every node answers with its id
    std::vector<uint32_t> ackers;
    std::vector<Serf::NodeResponse> answers;
    for (int t = 0; t < 14; ++t) {
      cl.step();
      for (uint32_t a : qr.acks()) ackers.push_back(a);                       // each call: the responders new since the last one
      for (auto& nr : qr.responses()) answers.push_back(nr);
    }
    std::vector<uint32_t> once = ackers;
    std::sort(once.begin(), once.end());
    REQUIRE(std::adjacent_find(once.begin(), once.end()) == once.end());       // every responder exactly once (QueryResponseCore.acks)
    REQUIRE(ackers.size() == 256 && answers.size() == 256);                    // lossless cluster: everybody, the origin included
    Serf::QueryStatus qs = s9.query_status(77);
    REQUIRE(qs.acks == ackers.size() && qs.responses == answers.size());
    for (auto& nr : answers) REQUIRE(nr.payload.size() == 2 && (uint32_t)(nr.payload[0] | (nr.payload[1] << 8)) == nr.from);
    REQUIRE(!qr.finished());                                                   // deadline 16 * ceil(log10(257)) = 48 ticks
    qr.close();
    REQUIRE(qr.finished() && qr.acks().empty());
    printf("query_acks %zu query_responses %zu\n", ackers.size(), answers.size());
  } catch (const Error& er) {
    fprintf(stderr, "unexpected %s\n", er.what());
    return 2;
  }
  printf("ok\n");
  return 0;
}
