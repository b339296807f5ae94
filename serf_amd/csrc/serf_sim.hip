// serf_sim.hip — MI355X (gfx950) implementation of include/serf_sim.h.
//
// One simulated node per lane, one fused kernel per gossip tick.  Per tick each lane
//   1. streams its packed row (three 16-byte groups: the 3 Lamport clocks, SerfState bits and
//      counters, queue bookkeeping; a fourth group with the memberlist fields when the SWIM layer
//      is on) and the 16 sort keys of its TransmitLimitedQueue out of HBM,
//   2. reads the fan-out packets addressed to it (48 B each: 12-byte wire records) — on one GPU it FETCHES them from
//      their senders through the inverse of the fan-out map (a sender keeps one copy of each distinct packet, step 6),
//      in a sharded run from the exchange buffer — four records at a
//      time, first issuing the four independent de-dup lookups of a packet (slot map -> view
//      column entry, or event/query ring bucket) and only then running the handlers in arrival
//      order, so a packet costs two memory round trips instead of eight,
//   3. runs every piggyback record through the serf-core handlers (Lamport witness, join/leave
//      intent vs. status_time, user-event / query de-dup rings: serf-core/src/serf/delegate.rs:
//      157-315, serf/base.rs:750-1572) and memberlist's alive/suspect/dead rules below them,
//   4. advances its suspicion timers and, every probe interval, probes one peer,
//   5. keeps the queue as 16 sort keys in registers — (class, transmits, length, id, slot) packed
//      in 32 bits, sorted by min/max networks — while the 16-byte records themselves never move
//      (slot-stable payload array), drains `fanout` packets of SIM_P records
//      (delegate.rs:317-384, memberlist-core App. B.1) and
//   6. sends packet k to the peer chosen by this tick's fixed-point-free pseudo-random bijection — every node
//      receives exactly one packet per slot, so no atomics and no ordering ambiguity (DESIGN.md SIMSPEC).  One GPU:
//      a node's f packets of a tick are nearly always the SAME packet, so it writes each distinct packet once, next
//      to itself, plus a map word (slot -> cell), and the receivers come and get it (step 2): a quarter of the packet
//      writes.  Sharded: the packets go into the exchange buffer, one dense slab per (destination, slot).
// Integer / byte work only: the roofline is HBM bandwidth, there is nothing for MFMA to do.
//
// There is deliberately no CPU fallback in this file: without a usable HIP device sim_create
// returns SIM_EDEVICE.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <rccl/rccl.h>  // the round's all-to-all issued by the library itself (sim_exchange_*): grouped ncclSend / ncclRecv over xGMI

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/serf_sim.h"
#include "../host/wire.hpp"  // the reference's message encoding (host side only: sim_deliver_message / sim_peek_packet)
#include <unordered_map>

typedef uint32_t u32;
typedef uint64_t u64;

#define NOSLOT 0xFFFFFFFFu
#define STAMP_MASK 0x1FFFFFu
#define BLOCK 256
// The tick kernel runs one wave per block: nothing in it is shared between waves (LDS staging is per lane, the
// transposes per quad), and the smaller the unit the scheduler hands out, the shorter the tail at the end of a launch
// in which 16 384 waves go through 4 096 slots (measured: 256 -> 64 threads per block = -2.4 %).
#ifndef TBLOCK
#define TBLOCK 64
#endif
#define KEMPTY 0xFFFFFFFFu  // empty sort key
// broadcasts one node can park in one tick: every received record can ask for one rebroadcast,
// every suspicion timer can fire (dead) and the probe can fail (suspect): Dev::npend = fanout * pkt_records + SIM_S + 1


// One translation unit, eight files (r5: the 5 700-line file, split where its sections ended).  It stays ONE unit on purpose: the
// handlers are force-inlined into the tick kernel across these files — separate device translation units would need relocatable
// device code, which changes the code the compiler generates for the hot kernel — and the host side launches the kernel
// templates it instantiates.  bench.py stamps its PMC profiles with the hash of the DEVICE files (state, handlers, tick).
#include "serf_sim_state.inc"
#include "serf_sim_handlers.inc"
#include "serf_sim_tick.inc"
#include "serf_sim_kernels.inc"
#include "serf_sim_host.inc"

extern "C" {

#include "serf_sim_api.inc"
#include "serf_sim_exchange.inc"

}  // extern "C"
