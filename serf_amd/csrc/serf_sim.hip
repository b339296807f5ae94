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

// ------------------------------------------------------------------------------------------------
// hashing / permutation (same arithmetic as the spec; host and device)
// ------------------------------------------------------------------------------------------------
__host__ __device__ static inline u64 mix64(u64 z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
enum { STREAM_PERM = 1, STREAM_OFF = 2, STREAM_ROT = 3, STREAM_LOSS = 4, STREAM_PROBE = 5, STREAM_QUERY = 6, STREAM_RFAN = 7, STREAM_RHO = 8 };
enum { PD_TARGET = 0, PD_PING = 1, PD_ACK = 2, PD_RELAY0 = 3, PD_RECONNECT = 30 /* + 1: which failed member */ };
#define SREQ_RECONNECT 0x80000000u  // request-list entries of the Reconnector: (node, target | this)
static inline u64 rng_base(u64 seed, u64 stream, u64 a) {
  return mix64(mix64(seed ^ (stream * 0xD6E8FEB86659FD93ull)) ^ a);
}
static inline u64 rng4(u64 seed, u64 stream, u64 a, u64 b) { return mix64(rng_base(seed, stream, a) ^ b); }

struct TickP {  // per-tick parameters, passed by value (lands in SGPRs)
  u64 tick;
  u64 loss_base, probe_base, query_base;
  u64 rfan_base;  // random fan-out: the base of this tick's target draws
  u32 M, mask, shift, feff, V, blk, loss_u32, first;  // first: tick 0 has no inbox yet
  u32 n_slots;  // length of d.walk: view slots in use (the Reaper and the push-pull merge walk them)
  u32 zero_;    // always 0 (opaque to the compiler)
  u32 abl;      // -DTICK_ABLATE measurement builds only: parts of the tick to leave out
  // fan-out map (SIMSPEC §2.3): C sender chunks, sub = blk / C cells per (chunk, destination, slot) slab, blocks of B
  // nodes (64, or 1 for small / ragged shards), nbc = V * sub / B blocks per chunk, bmask/bshift: bit width of the
  // block permutation
  u32 C, sub, B, nbc, bmask, bshift;
  u32 N, gmask, gshift;  // push-pull pairs come from a permutation of all N nodes
  u32 mul[3], add[3], imul[3];
  u32 off[SIM_MAX_FANOUT], rot[SIM_MAX_FANOUT], rho[SIM_MAX_FANOUT];
  u32 prot[SIM_MAX_FANOUT], prho[SIM_MAX_FANOUT];  // rot[] / rho[] of the previous tick (sharded reads)
};

static u32 modinv32(u32 a) {
  u32 x = a;
  for (int i = 0; i < 5; ++i) x *= 2u - a * x;
  return x;
}
static void tickp_make(TickP* p, const sim_config* c, u64 tick) {
  memset(p, 0, sizeof *p);
  p->tick = tick;
  p->V = c->vshards;
  p->M = c->n_nodes / c->vshards;
  p->blk = p->M / p->V;
  u32 nbits = 0;
  while (nbits < 32 && (1ull << nbits) < p->M) ++nbits;
  if (nbits < 1) nbits = 1;
  p->mask = nbits >= 32 ? 0xFFFFFFFFu : ((1u << nbits) - 1u);
  p->shift = (nbits + 1) / 2;
  p->N = c->n_nodes;
  {
    u32 gb = 0;
    while (gb < 32 && (1ull << gb) < p->N) ++gb;
    if (gb < 1) gb = 1;
    p->gmask = gb >= 32 ? 0xFFFFFFFFu : ((1u << gb) - 1u);
    p->gshift = (gb + 1) / 2;
  }
  p->C = c->chunks ? c->chunks : 1;
  p->sub = p->blk / p->C;
  p->B = (p->sub % 64u == 0 && (u64)p->V * p->sub / 64u >= 8u) ? 64u : 1u;
  p->nbc = (u32)((u64)p->V * p->sub / p->B);
  {
    u32 bb = 0;
    while (bb < 32 && (1ull << bb) < p->nbc) ++bb;
    if (bb < 1) bb = 1;
    p->bmask = bb >= 32 ? 0xFFFFFFFFu : ((1u << bb) - 1u);
    p->bshift = (bb + 1) / 2;
  }
  p->feff = std::min(c->fanout, p->nbc - 1);
  for (int r = 0; r < 3; ++r) {
    u64 w = rng4(c->seed, STREAM_PERM, tick, (u64)r);
    p->mul[r] = (u32)w | 1u;
    p->add[r] = (u32)(w >> 32);
    p->imul[r] = modinv32(p->mul[r]);
  }
  for (u32 k = 0; k < p->feff; ++k) {
    u64 u = rng4(c->seed, STREAM_OFF, tick, k);
    u32 ck = 1u + (u32)(u % (u64)(p->nbc - 1));
    for (;;) {
      bool clash = false;
      for (u32 j = 0; j < k; ++j) clash |= (p->off[j] == ck);
      if (!clash) break;
      ck = ck % (p->nbc - 1) + 1u;
    }
    p->off[k] = ck;
    p->rot[k] = (u32)(rng4(c->seed, STREAM_ROT, tick, k) % (u64)p->V);
    p->rho[k] = (u32)(rng4(c->seed, STREAM_RHO, tick, k) % (u64)p->C);
  }
  p->loss_base = rng_base(c->seed, STREAM_LOSS, tick);
  p->probe_base = rng_base(c->seed, STREAM_PROBE, tick);
  p->query_base = rng_base(c->seed, STREAM_QUERY, tick);
  p->rfan_base = rng_base(c->seed, STREAM_RFAN, tick);
  p->loss_u32 = c->loss_u32;
  p->first = (tick == 0);
}

__device__ static inline u32 perm_f(const TickP& p, u32 x) {
  x = (x * p.mul[0] + p.add[0]) & p.mask;
  x ^= x >> p.shift;
  x = (x * p.mul[1] + p.add[1]) & p.mask;
  x ^= x >> p.shift;
  x = (x * p.mul[2] + p.add[2]) & p.mask;
  return x;
}
__device__ static inline u32 perm_fi(const TickP& p, u32 y) {
  y = ((y - p.add[2]) * p.imul[2]) & p.mask;
  y ^= y >> p.shift;
  y = ((y - p.add[1]) * p.imul[1]) & p.mask;
  y ^= y >> p.shift;
  y = ((y - p.add[0]) * p.imul[0]) & p.mask;
  return y;
}
// the block permutation pi of the fan-out map: the same rounds on the bit width of nbc, cycle-walking into [0, nbc)
__host__ __device__ static inline u32 pi_f(const TickP& p, u32 x) {
  do {
    x = (x * p.mul[0] + p.add[0]) & p.bmask;
    x ^= x >> p.bshift;
    x = (x * p.mul[1] + p.add[1]) & p.bmask;
    x ^= x >> p.bshift;
    x = (x * p.mul[2] + p.add[2]) & p.bmask;
  } while (x >= p.nbc);
  return x;
}
__host__ __device__ static inline u32 pi_inv(const TickP& p, u32 y) {
  do {
    y = ((y - p.add[2]) * p.imul[2]) & p.bmask;
    y ^= y >> p.bshift;
    y = ((y - p.add[1]) * p.imul[1]) & p.bmask;
    y ^= y >> p.bshift;
    y = ((y - p.add[0]) * p.imul[0]) & p.bmask;
  } while (y >= p.nbc);
  return y;
}
// The fan-out map in its general (per-node) form, both directions (SIMSPEC §2.3, oracle fan_target).  A node is
// (shard g, in-shard index ll); ll = (vblock bb0, sub-slab s0, offset r0).  The tick kernel evaluates the same map
// with the block permutation on the scalar unit; these are what the support kernels (and the host-side test hook) use.
#define SEL4(i, a, b, c, d) ((i) == 0 ? (a) : (i) == 1 ? (b) : (i) == 2 ? (c) : (d))
__host__ __device__ static inline u32 fan_scramble(u32 y, u32 k) { return ((y + 1u) * 0x9E3779B1u + k * 0x85EBCA6Bu) >> 26; }
__host__ __device__ static inline void fan_target_g(const TickP& p, u32 g, u32 ll, u32 k, u32& h, u32& t) {
  u32 bb0 = ll / p.blk, w = ll - bb0 * p.blk, s0 = w / p.sub, r0 = w - s0 * p.sub;
  u32 u = bb0 * p.sub + r0, j = u / p.B, i = u - j * p.B;
  u32 y = pi_f(p, j) + p.off[k];
  if (y >= p.nbc) y -= p.nbc;
  u32 j2 = pi_inv(p, y);
  u32 i2 = p.B == 64u ? (i ^ fan_scramble(y, k)) : 0u;
  u32 u2 = j2 * p.B + i2, bb = u2 / p.sub, r = u2 - bb * p.sub;
  u32 s = s0 + p.rho[k];
  if (s >= p.C) s -= p.C;
  h = (g + p.V - ((bb + p.rot[k]) % p.V)) % p.V;
  t = bb * p.blk + s * p.sub + r;
}
// the node whose k-th packet lands at (shard h, in-shard index t): the inverse of fan_target_g in (g, ll) for fixed k
// (off_k, rot_k, rho_k = p.off[k], p.rot[k], p.rho[k]: picked by the caller, a kernel must not index its arguments dynamically)
#define PICK4(a, k) SEL4(k, (a)[0], (a)[1], (a)[2], (a)[3])
__host__ __device__ __attribute__((always_inline)) static inline void fan_source_g(const TickP& p, u32 off_k, u32 rot_k, u32 rho_k, u32 h, u32 t, u32 k, u32& g, u32& ll) {
  u32 bb = t / p.blk, w = t - bb * p.blk, s = w / p.sub, r = w - s * p.sub;
  u32 u2 = bb * p.sub + r, j2 = u2 / p.B, i2 = u2 - j2 * p.B;
  u32 y = pi_f(p, j2);
  u32 yy = y >= off_k ? y - off_k : y + p.nbc - off_k;
  u32 j = pi_inv(p, yy);
  u32 i = p.B == 64u ? (i2 ^ fan_scramble(y, k)) : 0u;
  u32 u = j * p.B + i, bb0 = u / p.sub, r0 = u - bb0 * p.sub;
  u32 s0 = s >= rho_k ? s - rho_k : s + p.C - rho_k;
  g = (h + (bb + rot_k) % p.V) % p.V;
  ll = bb0 * p.blk + s0 * p.sub + r0;
}
// the permutation of all N nodes the push-pull matching comes from (host and device: the sharded host plans with it)
__host__ __device__ static inline u32 sigma_g_inv(const TickP& p, u32 y) {
  do {
    y = ((y - p.add[2]) * p.imul[2]) & p.gmask;
    y ^= y >> p.gshift;
    y = ((y - p.add[1]) * p.imul[1]) & p.gmask;
    y ^= y >> p.gshift;
    y = ((y - p.add[0]) * p.imul[0]) & p.gmask;
  } while (y >= p.N);
  return y;
}
__device__ static inline u32 sigma(const TickP& p, u32 x) {
#ifdef TICK_TIMING_IDENTITY  // measurement only: coalesced fan-out (target = node + off) instead of the bijection
  return x;
#endif
  do x = perm_f(p, x); while (x >= p.M);
  return x;
}
__device__ static inline u32 sigma_inv(const TickP& p, u32 y) {
#ifdef TICK_TIMING_IDENTITY
  return y;
#endif
  do y = perm_fi(p, y); while (y >= p.M);
  return y;
}

// ------------------------------------------------------------------------------------------------
// device state (data layout in HBM: DESIGN.md §3)
// ------------------------------------------------------------------------------------------------
#define SREQ_HEAD 62u  // pairs of a tick's request list the kernel also writes into pinned host memory (a longer list is fetched when it is read)
struct Dev {
  // row groups, one uint4 per node each
  uint4* R0;  // {clock.lo, clock.hi, event_clock.lo, event_clock.hi}
  uint4* R1;  // {query_clock.lo, query_clock.hi, flags, n_known}
  uint4* R2;  // {n_failed, n_left, next_seq | used-slot mask << 16, overflow}
  uint4* R3;  // {incarnation, susp_next, awareness, reap_next}          (memberlist layer / Reaper)
  uint4* R4;  // [Nl][2]: susp[16] x u16, view slot + 1 of each running suspicion timer (memberlist layer; rare paths only)
  uint4* R5;  // {event_min.lo, event_min.hi, query_min.lo, query_min.hi} (read when SIM_RF_MINTIME)
  uint4* qkeys;  // [4][Nl]  the 16 sort keys of a node's queue, ascending, 4 per uint4
  uint4* qpay;   // [Q][Nl]  slot-stable wire records {key, wire meta, val.lo, val.hi}
  uint4* pend;   // [npend][Nl] broadcasts requested by the handlers of the running tick, arrival order
  // Local mode: what a node sent, kept at the SENDER (SIMSPEC §2.3 read from the other end).  A node's f packets of one
  // tick are nearly always the same packet (a queue of at most SIM_P entries sends the same records f times), so it
  // writes each DISTINCT packet once — obox[j][sender], 3 x uint4: keys, value low words, value high bits + meta — and
  // one word omap[sender] = for every fan-out slot the index j of the cell that holds its packet (0xFF: nothing sent
  // or lost).  The receiver of slot k looks up its sender (the inverse of the map) and fetches the cell.
  uint4* obox[2];        // [f * PG][Nl] cells (pages), double buffered by tick parity
  u32* omap[2];          // [Nl]
  uint4 *xsend, *xrecv;  // sharded mode: [V][f][blk] packets
  // Entries are split into two planes of 16 bytes per (row, node): the HEAD the hot path checks every record against
  // — view {ltime.lo, ltime.hi, inc, bits}, ring bucket {ltime.lo, ltime.hi, k0, k1} — at arr[row * Nl + l], dense
  // across the nodes of a wave, and the rarely touched TAIL — view conf[4], bucket k2..k5 — `*tail` uint4s further on.
  uint4* view;   // [2][A][Nl]
  uint4* ering;  // [2][Bev][Nl]
  uint4* qring;  // [2][Bq][Nl]
  size_t vtail, etail, qtail;  // A * Nl, Bev * Nl, Bq * Nl
  u32* slot_of;     // [N]
  u32* subject_of;  // [A]
  u32* walk;        // [n_slots] allocated slots in ascending SUBJECT order (the order of every walk over the view)
  u32* upmap;       // [ceil(N/32)] ground-truth liveness of every node (all shards)
  uint4* qtab;      // [SIM_QT] running queries {qid, origin, deadline, flags}; then [SIM_QT][4] their filters
                    // {qid, n_ids, tag mask, 0, ids[12]} (QFILT); then [N] bytes, every node's tag class (TAGCLASS)
  u32* qbits;       // [SIM_QT][2][ceil(N/32)] who acked / responded, by global node id
  uint4* nullcell;  // 2 x uint4 of zeros: where the prefetch of a record without a lookup points
  uint8_t* skipmask;  // [Nl] gossip_to_the_dead: bit k = do not send packet k this tick (written by gossip_skip_kernel)
  // SIM_CF_RANDOM_FANOUT (memberlist's literal kRandomNodes, App. B.2): the fan-out graph of a tick as an explicit CSR — for
  // receiver l the packets addressed to it are rsrc[rcsr[l] .. rcsr[l + 1]), each entry sender * 4 + slot, in (sender, slot)
  // order: the order memberlist's packets would be handed to the oracle in.  A function of (seed, tick) alone, built ahead of
  // the tick by the rf_* kernels (a two-level bucket sort) on a stream of their own, while the tick before runs; the host
  // points rcsr / rsrc at the graph of the packets this tick RECEIVES (sim_handle::rf_rcsr / rf_rsrc, three buffers each by
  // sending tick).  The packets stay with their senders as in the bijection's local mode, in 64-byte cells (RF_CELL_U4) whose
  // cell 0 carries the sender's map word in its fourth quarter — no omap array in this mode.
  u32* rcsr;       // [Nl + 1]
  u32* rsrc;       // [f * Nl] (a shard: the packets for its nodes — f * Nl on average, room for more)
  u32 rfan;        // the mode is on
  // where a receiver finds the cells, and how many senders a plane of cells spans: the handle's own cells of the tick before
  // (obox[cur], Nl senders) — or, on a shard, the receive buffer the cells of EVERY shard were gathered into (N senders: the
  // entries of rsrc then carry global sender ids).  Set per tick by the host.
  const uint4* rfrd;
  u32 NC;
  u32* sreq;        // [1 + 2 * SIM_SUSPECT_REQ_MAX]: count, then the (prober, target) pairs of the running tick's slot-less
                    // failed probes (one of three buffers, by tick mod 3: the host reads a tick's list one tick later)
  u32* sreq_next;   // the count word of the NEXT tick's buffer: zeroed by this tick's kernel (no memset between two ticks)
  u32* sreq_hh;     // pinned HOST memory, [SREQ_HEAD] pairs, 0xFFFFFFFF-terminated: the head of the same list, written
                    // straight to where the host reads it (no copy between two ticks); null in sharded handles
  sim_event* events;
  u32* ev_count;
  u32 ev_cap;
  u32 N, Nl, M, V, A, Bev, Bq, f, shard0, shard_rank, sharded, retransmit_mult;
  u32 P, PG, fp;  // records a packet can carry (sim_config.pkt_records), its pages of SIM_P records, fp = f * PG cells per node
  u32 npend;      // rows of `pend`: f * P + SIM_S + 1
  u32 bev_mask, bq_mask;  // B - 1 when B is a power of two (> 1), else 0
  u32 swim, PI, kconf, ic, T[SIM_MAX_CONF];
  u32 loss_u32;
  u32 aw_probe;  // SIM_CF_AWARENESS_PROBE
  u32 tcp_fallback, nacks;  // SIM_CF_TCP_FALLBACK, SIM_CF_NACKS
  u32 gttd;      // gossip_to_the_dead in ticks (0 = off)
  u32 r3on;  // R3 is live: SWIM layer or Reaper configured
  u32 reap_interval, reconnect_timeout, tombstone_timeout, intent_timeout;
  u32 reconnect_interval;  // Reconnector period in ticks (0 = off, or no SWIM layer: nobody ever fails)
  u32 queue_check_interval, max_queue_depth, min_queue_depth;
};

__device__ static inline u32 digits10(u32 n) {
  u32 d = 0;
  d += n >= 1u; d += n >= 10u; d += n >= 100u; d += n >= 1000u; d += n >= 10000u;
  d += n >= 100000u; d += n >= 1000000u; d += n >= 10000000u; d += n >= 100000000u;
  d += n >= 1000000000u;
  return d;
}

// A queue entry's sort key: [27:26] class [25:20] transmits [19:14] 63-len [13:4] 1023-seq [3:0] slot.
// Ascending order of the key is TransmitLimitedQueue's drain order (App. B.1); the canonical
// `meta` of include/serf_sim.h is (key >> 4) << 8 | (wire meta & 0xFF).
struct Node {  // one node's state in registers
  u64 clock, eclock, qclock;
  u32 flags, nknown, nfailed, nleft, next_seq, used, overflow;
  u32 inc, susp_next, awareness, reap_next;
  u32 dirty;  // DR* bits: row groups that must be written back
  u32 npend;  // broadcasts parked in d.pend[] by this tick's handlers, queued once they are all done
};
typedef u32 (&SK)[SIM_Q];  // the 16 sort keys of a node's queue, ascending (phase 2 of the tick only)
enum { DR0 = 1, DR1 = 2, DR2 = 4, DR3 = 8 };  // R0 {clock, event_clock} R1 {query_clock, flags, n_known} R2 {n_failed, n_left, seq/used, overflow} R3 {inc, susp_next, awareness}

struct Ctx {
  const Dev& d;
  u32 l, gid;
  u32 tick;   // low 32 bits of the tick
  u64 qbase;  // per-tick base of the query-response loss draws
};
// A handler invocation queues at most one broadcast (the rebroadcast of the record it was given, or
// the refutation it answers with); it is parked here so that the kernel has ONE queue_broadcast
// site per call site of the handlers instead of one per `return true`.
struct Ins {
  u32 has, key, wmeta;
  u32 wide;  // the handler also wrote the node's OWN view entry (refutation), not only the record's
  u64 val;
};
__device__ static inline void ins_set(Ins& q, u32 key, u32 wmeta, u64 val) {
  q.has = 1; q.key = key; q.wmeta = wmeta; q.val = val;
}

__device__ static inline u32 kind_class(u32 kind) {
  return (kind == SIM_K_JOIN || kind == SIM_K_LEAVE) ? 1u : kind == SIM_K_QUERY ? 2u : kind == SIM_K_EVENT ? 3u : 0u;
}
__host__ __device__ static inline u32 wire_meta(u32 kind, u32 flags, u32 len_bytes) {
  u32 len64 = (len_bytes + 15u) / 16u;
  if (len64 > 63u) len64 = 63u;
  return ((63u - len64) << 18) | ((kind & 15u) << 4) | (flags & 15u);
}

// 16-byte accesses through a native vector type: one global_load/store_dwordx4 each, also when
// the access sits under a condition (a conditional uint4 load is otherwise split per component)
typedef u32 v4u __attribute__((ext_vector_type(4)));
__device__ static inline uint4 ld4(const uint4* p) {
  v4u v = *reinterpret_cast<const v4u*>(p);
  return make_uint4(v.x, v.y, v.z, v.w);
}

// ---- row / queue load-store -------------------------------------------------------------------
__device__ static inline void node_load(const Dev& d, u32 l, Node& n) {
  uint4 r0 = d.R0[l], r1 = d.R1[l], r2 = d.R2[l];
  uint4 r3 = make_uint4(0, 0, 0, 0);
  if (d.r3on) r3 = ld4(&d.R3[l]);
  n.clock = (u64)r0.x | ((u64)r0.y << 32);
  n.eclock = (u64)r0.z | ((u64)r0.w << 32);
  n.qclock = (u64)r1.x | ((u64)r1.y << 32);
  n.flags = r1.z; n.nknown = r1.w;
  n.nfailed = r2.x; n.nleft = r2.y; n.next_seq = r2.z & 0xFFFFu; n.used = r2.z >> 16; n.overflow = r2.w;
  n.inc = r3.x; n.susp_next = r3.y; n.awareness = r3.z; n.reap_next = r3.w;
  n.dirty = 0;
  n.npend = 0;
}
__device__ static inline void keys_load(const Dev& d, u32 l, u32 cnt0, SK sk) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    uint4 k = make_uint4(KEMPTY, KEMPTY, KEMPTY, KEMPTY);
    if (cnt0 > (u32)(4 * g)) k = ld4(&d.qkeys[(size_t)g * d.Nl + l]);
    sk[4 * g] = k.x; sk[4 * g + 1] = k.y; sk[4 * g + 2] = k.z; sk[4 * g + 3] = k.w;
  }
}
// the key groups that hold, or held, queue entries (a drain changes every live key)
__device__ static inline void keys_store(const Dev& d, u32 l, u32 cnt0, u32 used, const u32 (&sk)[SIM_Q]) {
  u32 cnt = max(cnt0, (u32)__popc(used));
#pragma unroll
  for (int g = 0; g < 4; ++g)
    if (cnt > (u32)(4 * g)) d.qkeys[(size_t)g * d.Nl + l] = make_uint4(sk[4 * g], sk[4 * g + 1], sk[4 * g + 2], sk[4 * g + 3]);
}
__device__ static inline bool ne4(const uint4& a, const uint4& b) { return a.x != b.x || a.y != b.y || a.z != b.z || a.w != b.w; }
// Write back the row groups some handler touched (every mutation site sets its DR* bit).
__device__ static inline void node_store(const Dev& d, u32 l, const Node& n) {
  if (n.dirty & DR0) d.R0[l] = make_uint4((u32)n.clock, (u32)(n.clock >> 32), (u32)n.eclock, (u32)(n.eclock >> 32));
  if (n.dirty & DR1) d.R1[l] = make_uint4((u32)n.qclock, (u32)(n.qclock >> 32), n.flags, n.nknown);
  if (n.dirty & DR2) d.R2[l] = make_uint4(n.nfailed, n.nleft, (n.next_seq & 0xFFFFu) | (n.used << 16), n.overflow);
  if (n.dirty & DR3) d.R3[l] = make_uint4(n.inc, n.susp_next, n.awareness, n.reap_next);
}

// ---- TransmitLimitedQueue on sort keys ----------------------------------------------------------
__device__ static inline void cas32(u32& a, u32& b) {
  u32 lo = min(a, b), hi = max(a, b);
  a = lo;
  b = hi;
}
__device__ static inline u32 sk_get(const u32 (&sk)[SIM_Q], u32 i) {  // sk[i] for a wave-uniform i, as a select tree
  u32 a = (i & 1) ? sk[1] : sk[0], b = (i & 1) ? sk[3] : sk[2], c = (i & 1) ? sk[5] : sk[4], d = (i & 1) ? sk[7] : sk[6];
  u32 e = (i & 1) ? sk[9] : sk[8], f = (i & 1) ? sk[11] : sk[10], g = (i & 1) ? sk[13] : sk[12], h = (i & 1) ? sk[15] : sk[14];
  u32 ab = (i & 2) ? b : a, cd = (i & 2) ? d : c, ef = (i & 2) ? f : e, gh = (i & 2) ? h : g;
  u32 lo = (i & 4) ? cd : ab, hi = (i & 4) ? gh : ef;
  return (i & 8) ? hi : lo;
}
// queue_broadcast (memberlist TransmitLimitedQueue, App. B.1): fresh id, a class-0 broadcast
// invalidates the queued class-0 broadcast about the same node, the entry that drains last falls
// off a full pool (counted as overflow); the record goes to the lowest free payload slot.
__device__ static void q_insert(const Ctx& c, Node& n, SK sk, u32 key, u32 wmeta, u64 val) {
  const Dev& d = c.d;
  u32 kind = (wmeta >> 4) & 15u, cls = kind_class(kind);
  u32 seq = n.next_seq++;
  n.dirty |= DR2;  // next_seq, used, overflow
  u32 k32 = (cls << 26) | (((wmeta >> 18) & 63u) << 14) | ((1023u - seq) << 4);
  if (cls == 0 && n.used) {
    // class-0 entries drain first, so they sit at the front of the key array; at most one of
    // them is about `key` (every insert removes its predecessor).  Rolled loop: one load in
    // flight at a time instead of sixteen address/value register pairs.
    u32 pos = SIM_Q, slot = 0;
#pragma unroll 1
    for (u32 i = 0; i < SIM_Q; ++i) {
      u32 k = sk_get(sk, i);
      if (k == KEMPTY || (k >> 26) != 0) break;
      u32 sl = k & 15u;
      if (d.qpay[(size_t)sl * d.Nl + c.l].x == key) { pos = i; slot = sl; break; }
    }
    if (pos < SIM_Q) {
      n.used &= ~(1u << slot);
#pragma unroll
      for (int i = 0; i < (int)SIM_Q - 1; ++i) sk[i] = ((u32)i >= pos) ? sk[i + 1] : sk[i];
      sk[SIM_Q - 1] = KEMPTY;
    }
  }
  if (sk[SIM_Q - 1] != KEMPTY) {  // pool full
    n.overflow++;
    if (k32 > sk[SIM_Q - 1]) return;  // the newcomer drains last: it is the one dropped
    n.used &= ~(1u << (sk[SIM_Q - 1] & 15u));
    sk[SIM_Q - 1] = KEMPTY;
  }
  u32 slot = (u32)__ffs((int)(~n.used & 0xFFFFu)) - 1u;
  n.used |= 1u << slot;
  k32 |= slot;
  d.qpay[(size_t)slot * d.Nl + c.l] = make_uint4(key, wmeta & SIM_META_WIRE_MASK, (u32)val, (u32)(val >> 32));
#pragma unroll
  for (int i = SIM_Q - 1; i >= 1; --i) {
    bool below = sk[i - 1] > k32;  // predecessor sorts after the newcomer => shift it right
    bool here = !below && sk[i] > k32;
    sk[i] = below ? sk[i - 1] : (here ? k32 : sk[i]);
  }
  if (sk[0] > k32) sk[0] = k32;
}

// bitonic sort network on the 16 sort keys (ascending)
__device__ static inline void sort16(SK sk) {
#pragma unroll
  for (int k = 2; k <= (int)SIM_Q; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
#pragma unroll
      for (int i = 0; i < (int)SIM_Q; ++i) {
        int l = i ^ j;
        if (l > i) {
          bool up = (i & k) == 0;
          u32 lo = min(sk[i], sk[l]), hi = max(sk[i], sk[l]);
          sk[i] = up ? lo : hi;
          sk[l] = up ? hi : lo;
        }
      }
    }
  }
}
// get_broadcasts for one packet (App. B.1; delegate.rs:317-384 `limit`): entries in drain order, every one that still
// fits the packet's byte budget (SIM_PKT_UNITS 16-byte units), at most SIM_P of them; transmits+1, drop at the
// retransmit limit, then restore the sorted order.  Returns the payload slots of the emitted entries, 0xFF where there
// is none.  Nearly always the first SIM_P entries fit together (records are tens of bytes): that case keeps the cheap
// 4-sort + bitonic merge; only when some lane of the wave holds large user events does the wave take the general walk.
__device__ static inline u32 q_round(Node& n, SK sk, u32 limit) {
  u32 head_units = 0;
#pragma unroll
  for (int p = 0; p < (int)SIM_P; ++p) head_units += sk[p] != KEMPTY ? 63u - ((sk[p] >> 14) & 63u) : 0u;
  if (__any(head_units > SIM_PKT_UNITS)) {
    u32 free_u = SIM_PKT_UNITS, cnt = 0, slots = 0xFFFFFFFFu;
#pragma unroll
    for (int i = 0; i < (int)SIM_Q; ++i) {
      u32 k = sk[i], len = 63u - ((k >> 14) & 63u);
      bool take = (k != KEMPTY) & (cnt < SIM_P) & (len <= free_u);
      u32 t = ((k >> 20) & 63u) + 1u;
      bool drop = t >= limit;
      if (take) {
        free_u -= len;
        slots = (slots & ~(0xFFu << (8 * cnt))) | ((k & 15u) << (8 * cnt));
        cnt++;
        if (drop) { n.used &= ~(1u << (k & 15u)); n.dirty |= DR2; }
        sk[i] = drop ? KEMPTY : k + (1u << 20);
      }
    }
    sort16(sk);
    return slots;
  }
  u32 a[SIM_P], slots = 0;
#pragma unroll
  for (int p = 0; p < (int)SIM_P; ++p) {
    u32 k = sk[p];
    bool valid = k != KEMPTY;
    u32 t = ((k >> 20) & 63u) + 1u;
    bool drop = t >= limit;
    slots |= (valid ? (k & 15u) : 0xFFu) << (8 * p);
    if (valid && drop) { n.used &= ~(1u << (k & 15u)); n.dirty |= DR2; }
    a[p] = valid ? (drop ? KEMPTY : k + (1u << 20)) : k;
  }
  cas32(a[0], a[1]); cas32(a[2], a[3]); cas32(a[0], a[2]); cas32(a[1], a[3]); cas32(a[1], a[2]);
  u32 s[SIM_Q];
#pragma unroll
  for (int i = 0; i < 12; ++i) s[i] = sk[i + 4];
  s[12] = a[3]; s[13] = a[2]; s[14] = a[1]; s[15] = a[0];
#pragma unroll
  for (int dd = 8; dd >= 1; dd >>= 1) {
#pragma unroll
    for (int i = 0; i < (int)SIM_Q; ++i)
      if ((i & dd) == 0) cas32(s[i], s[i + dd]);
  }
#pragma unroll
  for (int i = 0; i < (int)SIM_Q; ++i) sk[i] = s[i];
  return slots;
}

// The same round for a wave in which no lane holds more than 8 entries (sk[8 ..] all empty — the caller checks it with a
// ballot; at the benchmark load the deepest queue is 7): the four re-keyed head entries merge with four others, a
// 12-exchange network instead of 32.
__device__ static inline u32 q_round8(Node& n, SK sk, u32 limit) {
  u32 head_units = 0;
#pragma unroll
  for (int p = 0; p < (int)SIM_P; ++p) head_units += sk[p] != KEMPTY ? 63u - ((sk[p] >> 14) & 63u) : 0u;
  if (__any(head_units > SIM_PKT_UNITS)) return q_round(n, sk, limit);  // large messages: the general walk
  u32 a[SIM_P], slots = 0;
#pragma unroll
  for (int p = 0; p < (int)SIM_P; ++p) {
    u32 k = sk[p];
    bool valid = k != KEMPTY;
    bool drop = ((k >> 20) & 63u) + 1u >= limit;
    slots |= (valid ? (k & 15u) : 0xFFu) << (8 * p);
    if (valid && drop) { n.used &= ~(1u << (k & 15u)); n.dirty |= DR2; }
    a[p] = valid ? (drop ? KEMPTY : k + (1u << 20)) : k;
  }
  cas32(a[0], a[1]); cas32(a[2], a[3]); cas32(a[0], a[2]); cas32(a[1], a[3]); cas32(a[1], a[2]);
  u32 s[8] = {sk[4], sk[5], sk[6], sk[7], a[3], a[2], a[1], a[0]};  // ascending, then descending: bitonic
#pragma unroll
  for (int dd = 4; dd >= 1; dd >>= 1) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if ((i & dd) == 0) cas32(s[i], s[i + dd]);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) sk[i] = s[i];
  return slots;
}

// get_broadcasts for a packet of up to P records (P <= SIM_Q: sim_config.pkt_records > SIM_P): the same walk without
// the four-record shortcut.  The payload slots of the taken entries come back as nibbles, in drain order.
__device__ static inline void q_round_mp(Node& n, SK sk, u32 limit, u32 P, u64& nib, u32& cnt) {
  u32 free_u = SIM_PKT_UNITS, c = 0;
  u64 nb = 0;
#pragma unroll
  for (int i = 0; i < (int)SIM_Q; ++i) {
    u32 k = sk[i], len = 63u - ((k >> 14) & 63u);
    bool take = (k != KEMPTY) & (c < P) & (len <= free_u);
    bool drop = ((k >> 20) & 63u) + 1u >= limit;
    if (take) {
      free_u -= len;
      nb |= (u64)(k & 15u) << (4u * c);
      c++;
      if (drop) { n.used &= ~(1u << (k & 15u)); n.dirty |= DR2; }
      sk[i] = drop ? KEMPTY : k + (1u << 20);
    }
  }
  sort16(sk);
  nib = nb;
  cnt = c;
}

// rare: renumber the queue ids when the 10-bit id space is nearly used up (order-preserving)
__device__ static void q_renorm(Node& n, SK sk) {
  u32 o[SIM_Q], cnt = 0;
#pragma unroll
  for (int i = 0; i < (int)SIM_Q; ++i) o[i] = sk[i];
#pragma unroll
  for (int i = 0; i < (int)SIM_Q; ++i) {
    if (o[i] == KEMPTY) continue;
    u32 fi = (o[i] >> 4) & 0x3FFu, rank = 0;  // field = 1023 - seq: larger field = older
#pragma unroll
    for (int j = 0; j < (int)SIM_Q; ++j) rank += (o[j] != KEMPTY && ((o[j] >> 4) & 0x3FFu) > fi) ? 1u : 0u;
    sk[i] = (o[i] & ~(0x3FFu << 4)) | ((1023u - rank) << 4);
    cnt++;
  }
  n.next_seq = cnt;
  n.dirty |= DR2;
}

// Park a broadcast request; phase 2 of the tick queues them in this order (queue_broadcast order
// is arrival order, and no handler looks at the queue, so deferring is exact).
__device__ static __forceinline__ void pend_push(const Ctx& c, Node& n, const Ins& q) {
  if (n.npend >= c.d.npend) { n.overflow++; n.dirty |= DR2; return; }  // cannot happen (npend is the per-tick maximum); never write past the array
  c.d.pend[(size_t)n.npend * c.d.Nl + c.l] = make_uint4(q.key, q.wmeta, (u32)q.val, (u32)(q.val >> 32));
  n.npend++;
}

// ---- view / ring access -------------------------------------------------------------------------
__device__ static inline u32 vb_make(u32 known, u32 status, u32 swim, u32 intent, u32 nconf, u32 stamp) {
  return (known & 1u) | ((status & 7u) << 1) | ((swim & 3u) << 4) | ((intent & 3u) << 6) | ((nconf & 7u) << 8) | (stamp << 11);
}
__device__ static inline u32 vb_set_status(u32 b, u32 s) { return (b & ~(7u << 1)) | ((s & 7u) << 1); }
__device__ static inline u32 vb_set_swim(u32 b, u32 w) { return (b & ~(3u << 4)) | ((w & 3u) << 4); }
__device__ static inline u32 vb_set_intent(u32 b, u32 t) { return (b & ~(3u << 6)) | ((t & 3u) << 6); }
__device__ static inline u32 vb_set_nconf(u32 b, u32 k) { return (b & ~(7u << 8)) | ((k & 7u) << 8); }
__device__ static inline u32 vb_set_stamp(u32 b, u32 st) { return (b & 0x7FFu) | (st << 11); }
#define E_LTIME(e) ((u64)(e).x | ((u64)(e).y << 32))
#define E_SET_LTIME(e, t) ((e).x = (u32)(t), (e).y = (u32)((t) >> 32))

__device__ static inline uint4* view_slot_ptr(const Ctx& c, u32 a) { return c.d.view + ((size_t)a * c.d.Nl + c.l); }
__device__ static inline uint4* view_ptr(const Ctx& c, u32 subject) {
  if (subject >= c.d.N) return nullptr;
  u32 a = c.d.slot_of[subject];
  if (a == NOSLOT) return nullptr;
  return view_slot_ptr(c, a);
}
__device__ static inline u32 ring_idx(u64 lt, u32 B, u32 mask) {
  if (mask) return (u32)lt & mask;
  return (lt >> 32) ? (u32)(lt % B) : ((u32)lt % B);
}
__device__ static inline uint4* ering_ptr(const Ctx& c, u64 lt) {
  return c.d.ering + ((size_t)ring_idx(lt, c.d.Bev, c.d.bev_mask) * c.d.Nl + c.l);
}
__device__ static inline uint4* qring_ptr(const Ctx& c, u64 lt) {
  return c.d.qring + ((size_t)ring_idx(lt, c.d.Bq, c.d.bq_mask) * c.d.Nl + c.l);
}

// Event stream of watched observers (event.rs:325-378); appended in program order per node, the
// host orders the log by (tick, observer).
__device__ static inline void emit_event(const Ctx& c, const Node& n, u32 type, u32 key, u64 ltime) {
  if (!(n.flags & SIM_RF_WATCHED)) return;
  u32 i = atomicAdd(c.d.ev_count, 1u);
  if (i < c.d.ev_cap) {
    sim_event e;
    e.tick = c.tick; e.observer = c.gid; e.type = type; e.key = key; e.ltime = ltime;
    c.d.events[i] = e;
  }
}

// ---- serf-core handlers --------------------------------------------------------------------------
__device__ static inline void witness(Node& n, u64& c, u64 t, u32 group) {  // types/clock.rs:155-172
  if (t >= c) { c = t + 1; n.dirty |= group; }
}
// Reaper bookkeeping (see reap_run): earliest tick at which an entry of this node out-lives its timeout
__device__ static inline void reap_arm(const Ctx& c, Node& n, u32 age, u32 timeout) {
  if (!c.d.reap_interval) return;
  u32 due = c.tick - age + timeout + 1u;
  if (!n.reap_next || due < n.reap_next) { n.reap_next = due; n.dirty |= DR3; }
}
// upsert_intent: base.rs:1835-1866
__device__ static inline bool upsert_intent(uint4& e, u32 ty, u64 ltime, u32 stamp) {
  if (SIM_VB_INTENT(e.w)) {
    if (ltime > E_LTIME(e)) {
      e.w = vb_set_stamp(vb_set_intent(e.w, ty), stamp);
      E_SET_LTIME(e, ltime);
      return true;
    }
    return false;
  }
  e.w = vb_set_stamp(vb_set_intent(e.w, ty), stamp);
  E_SET_LTIME(e, ltime);
  return true;
}
// erase_node!: base.rs:499-518
__device__ static inline void erase_member(const Ctx& c, Node& n, uint4* p, const uint4& e, u32 subject) {
  u32 st = SIM_VB_STATUS(e.w);
  if (st == SIM_STATUS_FAILED && n.nfailed) n.nfailed--;
  if (st == SIM_STATUS_LEFT && n.nleft) n.nleft--;
  p[0] = make_uint4(0, 0, 0, 0);
  p[c.d.vtail] = make_uint4(0, 0, 0, 0);
  if (n.nknown) n.nknown--;
  n.dirty |= DR1 | DR2;
  emit_event(c, n, SIM_EV_REAP, subject, 0);
}
// handle_node_join_intent: base.rs:1338-1373.  (p, e) = the subject's view entry, e preloaded.
__device__ static bool handle_join_intent(const Ctx& c, Node& n, u32 subject, u64 ltime, uint4* p, uint4& e, bool& dirty) {
  witness(n, n.clock, ltime, DR0);
  if (!p) return false;
  if (e.w & SIM_VB_KNOWN) {
    if (ltime <= E_LTIME(e)) return false;
    E_SET_LTIME(e, ltime);
    if (SIM_VB_STATUS(e.w) == SIM_STATUS_LEAVING) e.w = vb_set_status(e.w, SIM_STATUS_ALIVE);
    p[0] = e;
    dirty = true;
    return true;
  }
  bool rb = upsert_intent(e, 1, ltime, c.tick & STAMP_MASK);
  if (rb) {
    p[0] = e;
    dirty = true;
    if (c.d.intent_timeout) reap_arm(c, n, 0, c.d.intent_timeout);
  }
  return rb;
}
// broadcast_join: base.rs:381-397
__device__ static void broadcast_join(const Ctx& c, Node& n, u64 ltime, bool& dirty, Ins& ins) {
  witness(n, n.clock, ltime, DR0);
  uint4* p = view_ptr(c, c.gid);
  uint4 own = p ? p[0] : make_uint4(0, 0, 0, 0);
  handle_join_intent(c, n, c.gid, ltime, p, own, dirty);
  ins.wide = 1;
  ins_set(ins, c.gid, wire_meta(SIM_K_JOIN, 0, 16), ltime);
}
// handle_node_leave_intent: base.rs:1442-1572
__device__ static bool handle_leave_intent(const Ctx& c, Node& n, u32 subject, u64 ltime, bool prune, uint4* p, uint4& e, bool& dirty, Ins& ins) {
  u32 state = SIM_RF_STATE(n.flags);
  witness(n, n.clock, ltime, DR0);
  if (!p) return false;
  if (!(e.w & SIM_VB_KNOWN)) {
    bool rb = upsert_intent(e, 2, ltime, c.tick & STAMP_MASK);
    if (rb) {
      p[0] = e;
      dirty = true;
      if (c.d.intent_timeout) reap_arm(c, n, 0, c.d.intent_timeout);
    }
    return rb;
  }
  if (ltime <= E_LTIME(e)) return false;
  if (subject == c.gid && state == SIM_SERF_ALIVE) {  // refute: base.rs:1470-1480
    broadcast_join(c, n, n.clock, dirty, ins);
    return false;
  }
  E_SET_LTIME(e, ltime);
  u32 st = SIM_VB_STATUS(e.w);
  bool rb = true;
  if (st == SIM_STATUS_NONE) {
    rb = false;
  } else if (st == SIM_STATUS_ALIVE) {
    e.w = vb_set_status(e.w, SIM_STATUS_LEAVING);
  } else if (st == SIM_STATUS_LEAVING || st == SIM_STATUS_LEFT) {
  } else if (st == SIM_STATUS_FAILED) {
    e.w = vb_set_status(e.w, SIM_STATUS_LEFT);
    if (n.nfailed) n.nfailed--;
    n.nleft++;
    n.dirty |= DR2;
    reap_arm(c, n, (c.tick - SIM_VB_STAMP(e.w)) & STAMP_MASK, c.d.tombstone_timeout);
    emit_event(c, n, SIM_EV_LEAVE, subject, 0);
  } else {
    e.w = vb_set_status(e.w, SIM_STATUS_LEAVING);
  }
  dirty = true;
  if (prune && rb) { erase_member(c, n, p, e, subject); e = make_uint4(0, 0, 0, 0); }  // handle_prune: base.rs:1628-1653
  else p[0] = e;
  return rb;
}
// handle_node_join (memberlist notify_join): base.rs:1206-1334; works on the entry in registers
__device__ static void node_join_e(const Ctx& c, Node& n, uint4& e, u32 subject) {
  if (e.w & SIM_VB_KNOWN) {
    u32 old = SIM_VB_STATUS(e.w);
    e.w = vb_set_stamp(vb_set_status(e.w, SIM_STATUS_ALIVE), 0);
    if (old == SIM_STATUS_FAILED && n.nfailed) n.nfailed--;
    if (old == SIM_STATUS_LEFT && n.nleft) n.nleft--;
    n.dirty |= DR2;
  } else {
    u32 status = SIM_STATUS_ALIVE, it = SIM_VB_INTENT(e.w);
    u64 lt = 0;
    if (it == 1) lt = E_LTIME(e);
    if (it == 2) { lt = E_LTIME(e); status = SIM_STATUS_LEAVING; }
    E_SET_LTIME(e, lt);
    e.w = vb_make(1, status, SIM_VB_SWIM(e.w), 0, 0, 0);
    n.nknown++;
    n.dirty |= DR1;
  }
  emit_event(c, n, SIM_EV_JOIN, subject, 0);
}
// handle_node_leave (memberlist notify_leave): base.rs:1375-1440
__device__ static void node_leave_e(const Ctx& c, Node& n, uint4& e, u32 subject) {
  if (!(e.w & SIM_VB_KNOWN)) return;
  u32 st = SIM_VB_STATUS(e.w), stamp = c.tick & STAMP_MASK;
  if (st == SIM_STATUS_LEAVING) {
    e.w = vb_set_stamp(vb_set_status(e.w, SIM_STATUS_LEFT), stamp);
    n.nleft++;
    n.dirty |= DR2;
    reap_arm(c, n, 0, c.d.tombstone_timeout);
    emit_event(c, n, SIM_EV_LEAVE, subject, 0);
  } else if (st == SIM_STATUS_ALIVE) {
    e.w = vb_set_stamp(vb_set_status(e.w, SIM_STATUS_FAILED), stamp);
    n.nfailed++;
    n.dirty |= DR2;
    reap_arm(c, n, 0, c.d.reconnect_timeout);
    emit_event(c, n, SIM_EV_FAILED, subject, 0);
  }
}
__device__ static inline bool bucket_add(uint4* p, size_t tail, uint4& b0, u32 key, bool same_lt, bool check_lt, Node& n, bool& seen) {
  // returns true when the key was added; `seen` when it was already there (subject to ltime for queries)
  seen = false;
  bool m = !check_lt || same_lt;
  if (m && (b0.z == key || b0.w == key)) { seen = true; return false; }
  if (b0.w == 0) { b0.w = key; p[0] = b0; return true; }
  uint4 b1 = p[tail];
  if (m && (b1.x == key || b1.y == key || b1.z == key || b1.w == key)) { seen = true; return false; }
  if (b1.x == 0) b1.x = key;
  else if (b1.y == 0) b1.y = key;
  else if (b1.z == 0) b1.z = key;
  else if (b1.w == 0) b1.w = key;
  else { n.overflow++; n.dirty |= DR2; return false; }  // model bound: bucket full => treated as seen
  p[tail] = b1;
  return true;
}
// handle_user_event: base.rs:750-837 (quirk U1 kept).  (p, b0) = ring bucket of ltime, preloaded.
__device__ static bool handle_user_event(const Ctx& c, Node& n, u32 key, u64 ltime, uint4* p, uint4& b0, bool& dirty) {
  witness(n, n.eclock, ltime, DR0);
  if (n.flags & SIM_RF_MINTIME) {
    uint4 mn = c.d.R5[c.l];
    if (ltime < ((u64)mn.x | ((u64)mn.y << 32))) return false;
  }
  u64 B = c.d.Bev, cur = n.eclock;
  if (cur > B && ltime < cur - B) return false;
  if (b0.z) {  // bucket present: keys[0] != 0
    bool seen;
    if (!bucket_add(p, c.d.etail, b0, key, false, false, n, seen)) return false;
  } else {
    b0 = make_uint4((u32)ltime, (u32)(ltime >> 32), key, 0);
    p[0] = b0;
  }
  dirty = true;
  emit_event(c, n, SIM_EV_USER, key, ltime);
  return true;
}
#define QFILT(d) ((d).qtab + SIM_QT)
#define TAGCLASS(d) (reinterpret_cast<uint8_t*>((d).qtab + SIM_QT + SIM_QT * (SIM_QF_WORDS / 4)))
#define QTAB_U4(n) ((size_t)SIM_QT + (size_t)SIM_QT * (SIM_QF_WORDS / 4) + ((size_t)(n) + 15) / 16)
// should_process_query (query.rs:439-521): every filter must match — the node's id is in the Filter::Id list, its tag
// class is in the mask the host made of the Filter::Tag expressions (include/serf_sim.h).  Rare path: runs once per
// (node, query), after the de-dup.
// (`h` = the filter entry's first word group, `tcls` = the node's tag class, `t` in query_respond = the tracker entry: the
// caller loads the three together — they do not depend on each other, and fetched one inside the other they were three of
// the four dependent round trips of a first-seen query, with the whole wave waiting: profiles/r03_experiments.md)
__device__ static bool query_should_process(const Dev& d, u32 gid, u32 id, const uint4& h, u32 tcls) {
  const uint4* f = QFILT(d) + (size_t)(id % SIM_QT) * (SIM_QF_WORDS / 4);
  if (h.x != id) return true;  // no filters on record for this query
  if (h.z != 0xFFFFFFFFu && !((h.z >> tcls) & 1u)) return false;
  if (!h.y) return true;
  const u32* ids = reinterpret_cast<const u32*>(f + 1);
  for (u32 i = 0; i < h.y; ++i)
    if (ids[i] == gid) return true;
  return false;
}
// Responder half of handle_query (base.rs:1075-1154) and origin half (base.rs:1158-1204,
// query.rs:240-303) — see oracle query_respond: one bit per (running query, node) for acks, one for
// responses; the counts are popcounts taken when somebody asks (no hot atomic counter).
__device__ static void query_respond(const Ctx& c, u32 id, u32 flags, const uint4& t) {
  const Dev& d = c.d;
  if (!(flags & (SIM_F_ACK | SIM_F_RESPOND))) return;
  u32 j = id % SIM_QT;
  if (t.x != id) return;  // "reply for non-running query"
  if (c.tick > t.z || !((d.upmap[t.y >> 5] >> (t.y & 31)) & 1u)) return;
  u64 base = mix64(c.qbase ^ ((u64)id << 32));
  size_t words = ((size_t)d.N + 31) / 32;
  u32 relay = (t.w >> 8) & 7u;  // QueryMessage.relay_factor (query.rs:523-601)
  if (d.N < relay + 1) relay = 0;
  for (u32 which = 0; which < 2; ++which) {
    if (!(flags & (which ? SIM_F_RESPOND : SIM_F_ACK))) continue;
    u64 lane = (u64)c.gid * 64u + which * 32u;
#define QLOST(i) (d.loss_u32 && (u32)(mix64(base ^ (lane + (i))) >> 32) < d.loss_u32)
    bool ok = !QLOST(0);  // memberlist.send straight to the origin (base.rs:1097)
    for (u32 r = 0; !ok && r < relay; ++r) {  // relay_response: via a random live member, two more legs
      u32 via = (u32)(((mix64(base ^ (lane + 1 + 3 * r)) >> 32) * (u64)d.N) >> 32);
      if (via == c.gid || !((d.upmap[via >> 5] >> (via & 31)) & 1u)) continue;
      ok = !QLOST(2 + 3 * r) && !QLOST(3 + 3 * r);
    }
#undef QLOST
    if (!ok) continue;
    atomicOr(&d.qbits[((size_t)j * 2 + which) * words + (c.gid >> 5)], 1u << (c.gid & 31));
  }
}
// handle_query, de-dup part: base.rs:972-1073 (quirks Q1, Q2 kept)
__device__ static bool handle_query(const Ctx& c, Node& n, u32 id, u64 ltime, u32 flags, uint4* p, uint4& b0, bool& dirty) {
  witness(n, n.qclock, ltime, DR1);
  if (n.flags & SIM_RF_MINTIME) {
    uint4 mn = c.d.R5[c.l];
    if (ltime < ((u64)mn.z | ((u64)mn.w << 32))) return false;
  }
  u64 cur = n.qclock, qt = c.d.Bq;
  if (cur > qt && qt < cur - qt) return false;
  if (b0.z) {
    bool seen;
    if (!bucket_add(p, c.d.qtail, b0, id, E_LTIME(b0) == ltime, true, n, seen)) return false;
  } else {
    b0 = make_uint4((u32)ltime, (u32)(ltime >> 32), id, 0);
    p[0] = b0;
  }
  dirty = true;
  // base.rs:1062-1073: a node the filters exclude still rebroadcasts what it sees for the first time
  const uint4 fh = ld4(QFILT(c.d) + (size_t)(id % SIM_QT) * (SIM_QF_WORDS / 4)), trk = ld4(&c.d.qtab[id % SIM_QT]);
  const u32 tcls = TAGCLASS(c.d)[c.gid];
  if (!query_should_process(c.d, c.gid, id, fh, tcls)) return !(flags & SIM_F_NO_BROADCAST);
  query_respond(c, id, flags, trk);
  emit_event(c, n, SIM_EV_QUERY, id, ltime);
  return !(flags & SIM_F_NO_BROADCAST);
}

// ---- memberlist SWIM layer (oracle/serf_oracle.c swim_*) ------------------
__device__ static inline void aw_delta(Node& n, int dlt) {
  int a = (int)n.awareness + dlt;
  n.awareness = a < 0 ? 0u : a > (int)SIM_MAX_AWARENESS ? SIM_MAX_AWARENESS : (u32)a;
  n.dirty |= DR3;
}
// R4 holds SIM_S = 16 sixteen-bit entries per node (two uint4): view slot + 1 of each running suspicion timer (rare paths:
// plain 2-byte accesses)
static_assert(SIM_S == 16u, "R4 is laid out as two uint4 per node");
__device__ static inline uint16_t* susp_of(const Ctx& c) { return reinterpret_cast<uint16_t*>(&c.d.R4[2 * (size_t)c.l]); }
__device__ static inline u32 pk_word(const uint4& v, u32 p) { return p == 0 ? v.x : p == 1 ? v.y : p == 2 ? v.z : v.w; }
// (the sixteen entries are read as the two uint4 they are — one round trip — and looked at in registers; entry by entry
// they were up to sixteen dependent 2-byte loads, with the whole wave waiting for each: the handlers that touch the timers
// and the timer walk of the tick kernel run at a few active lanes)
__device__ static inline void susp_load(const Ctx& c, uint4& a, uint4& b) {
  const uint4* r4 = &c.d.R4[2 * (size_t)c.l];
  a = ld4(r4); b = ld4(r4 + 1);
}
__device__ static inline u32 susp_get(const uint4& a, const uint4& b, u32 j) {  // entry j of the two words groups
  u32 w = (j & 8u) ? pk_word(b, (j >> 1) & 3u) : pk_word(a, (j >> 1) & 3u);
  return (j & 1u) ? (w >> 16) : (w & 0xFFFFu);
}
__device__ static inline void susp_forget(const Ctx& c, u32 slot) {
  uint16_t* sp = susp_of(c);
  uint4 a, b;
  susp_load(c, a, b);
#pragma unroll
  for (u32 j = 0; j < SIM_S; ++j)
    if (susp_get(a, b, j) == slot + 1) sp[j] = 0;
}
__device__ static inline void susp_track(const Ctx& c, Node& n, u32 slot, u32 deadline) {
  uint16_t* sp = susp_of(c);
  uint4 a, b;
  susp_load(c, a, b);
  u32 j = SIM_S;
#pragma unroll
  for (int i = (int)SIM_S - 1; i >= 0; --i)
    if (!susp_get(a, b, (u32)i)) j = (u32)i;  // the first free entry
  if (j == SIM_S) { n.overflow++; n.dirty |= DR2; return; }  // model bound: the timer is not tracked
  sp[j] = (uint16_t)(slot + 1);
  if (!n.susp_next || deadline < n.susp_next) { n.susp_next = deadline; n.dirty |= DR3; }
}
__device__ static void swim_refute(const Ctx& c, Node& n, u32 accused_inc, Ins& ins) {
  u32 inc = n.inc + 1;
  if (accused_inc >= inc) inc = accused_inc + 1;
  n.inc = inc;
  n.dirty |= DR3;
  uint4* p = view_ptr(c, c.gid);
  if (p) { uint4 e = p[0]; e.z = inc; p[0] = e; }
  aw_delta(n, +1);
  ins.wide = 1;
  ins_set(ins, c.gid, wire_meta(SIM_K_ALIVE, 0, 64), inc);
}
__device__ static void swim_alive(const Ctx& c, Node& n, u32 subject, u32 inc, u32 wmeta, uint4* p, uint4& e, bool& dirty, Ins& ins) {
  if (!p) return;
  if (subject == c.gid) {
    if (inc <= n.inc) return;
    dirty = true;
    swim_refute(c, n, inc, ins);
    return;
  }
  if (!(e.w & SIM_VB_KNOWN)) {  // new member: notify_join
    e.w = vb_set_swim(e.w, SIM_SWIM_ALIVE);
    node_join_e(c, n, e, subject);
    e.z = inc;
    p[0] = e;
    dirty = true;
    ins_set(ins, subject, wmeta, inc);
    return;
  }
  if (inc <= e.z) return;
  u32 old = SIM_VB_SWIM(e.w);
  if (old == SIM_SWIM_SUSPECT) susp_forget(c, c.d.slot_of[subject]);
  e.z = inc;
  e.w = vb_set_nconf(vb_set_swim(e.w, SIM_SWIM_ALIVE), 0);
  ins_set(ins, subject, wmeta, inc);
  if (old == SIM_SWIM_DEAD || old == SIM_SWIM_LEFT) node_join_e(c, n, e, subject);
  else if (wmeta & SIM_F_META) emit_event(c, n, SIM_EV_UPDATE, subject, inc);  // notify_update -> handle_node_update, base.rs:1576-1624
  p[0] = e;
  dirty = true;
}
__device__ static void swim_suspect(const Ctx& c, Node& n, u32 subject, u32 inc, u32 from, u32 wmeta, uint4* p, uint4& e, bool& dirty, Ins& ins) {
  const Dev& d = c.d;
  if (!p || !(e.w & SIM_VB_KNOWN)) return;
  if (inc < e.z) return;
  u64 val = (u64)inc | ((u64)from << 32);
  if (SIM_VB_SWIM(e.w) == SIM_SWIM_SUSPECT) {  // a timer exists: try to confirm
    u32 k = SIM_VB_NCONF(e.w);
    if (k >= d.kconf) return;
    uint4 cf = p[d.vtail];
    if (cf.x == from) return;
    if (k >= 1 && cf.y == from) return;
    if (k >= 2 && cf.z == from) return;
    if (k >= 3 && cf.w == from) return;
    if (k == 0) cf.y = from;
    else if (k == 1) cf.z = from;
    else cf.w = from;
    p[d.vtail] = cf;
    e.w = vb_set_nconf(e.w, k + 1);
    p[0] = e;
    dirty = true;
    u32 deadline = c.tick - ((c.tick - SIM_VB_STAMP(e.w)) & STAMP_MASK) + d.T[k + 1];
    if (n.susp_next && deadline < n.susp_next) { n.susp_next = deadline; n.dirty |= DR3; }
    ins_set(ins, subject, wmeta, val);
    return;
  }
  if (SIM_VB_SWIM(e.w) != SIM_SWIM_ALIVE) return;
  if (subject == c.gid) { dirty = true; swim_refute(c, n, inc, ins); return; }
  ins_set(ins, subject, wmeta, val);
  e.z = inc;
  e.w = vb_set_stamp(vb_set_nconf(vb_set_swim(e.w, SIM_SWIM_SUSPECT), 0), c.tick & STAMP_MASK);
  p[0] = e;
  p[d.vtail] = make_uint4(from, 0, 0, 0);
  dirty = true;
  susp_track(c, n, d.slot_of[subject], c.tick + d.T[0]);
}
__device__ static void swim_dead(const Ctx& c, Node& n, u32 subject, u32 inc, u32 from, u32 wmeta, uint4* p, uint4& e, bool& dirty, Ins& ins) {
  if (!p || !(e.w & SIM_VB_KNOWN)) return;
  if (inc < e.z) return;
  u32 old = SIM_VB_SWIM(e.w);
  if (old == SIM_SWIM_SUSPECT) {  // cancel the timer
    susp_forget(c, c.d.slot_of[subject]);
    e.w = vb_set_nconf(e.w, 0);
    p[0] = e;
    dirty = true;
  }
  if (old == SIM_SWIM_DEAD || old == SIM_SWIM_LEFT) return;
  u32 st = SIM_RF_STATE(n.flags);
  if (subject == c.gid && st != SIM_SERF_LEAVING && st != SIM_SERF_LEFT) {  // not leaving: refute
    dirty = true;
    swim_refute(c, n, inc, ins);
    return;
  }
  ins_set(ins, subject, wmeta, (u64)inc | ((u64)from << 32));
  e.z = inc;
  e.w = vb_set_swim(e.w, from == subject ? SIM_SWIM_LEFT : SIM_SWIM_DEAD);
  node_leave_e(c, n, e, subject);  // notify_leave
  p[0] = e;
  dirty = true;
}
// suspicion timers (B.5): fire -> deadNode(inc, from = self).  One call examines timer j;
// `next` accumulates the earliest deadline still pending.
__device__ static void swim_timer_j(const Ctx& c, Node& n, u32 j, u32 a /* entry j of the node's timer list as the walk began */, u32& next, Ins& ins) {
  const Dev& d = c.d;
  u32 now = c.tick;
  uint16_t* sp = susp_of(c);
  if (!a) return;
  uint4* p = view_slot_ptr(c, a - 1);
  uint4 e = p[0];
  // (a slot that was recycled while this process was down: its subject is back at the baseline entry and
  // subject_of[] says NOSLOT — which must not be used as a subject: slot_of[NOSLOT] is 16 GiB past the table)
  if (d.subject_of[a - 1] == NOSLOT || SIM_VB_SWIM(e.w) != SIM_SWIM_SUSPECT) {
    sp[j] = 0;
    return;
  }
  u32 age = (now - SIM_VB_STAMP(e.w)) & STAMP_MASK;
  u32 T = d.T[SIM_VB_NCONF(e.w)];
  if (age >= T) {
    bool dirty = false;
    swim_dead(c, n, d.subject_of[a - 1], e.z, c.gid, wire_meta(SIM_K_DEAD, 0, 32), p, e, dirty, ins);
  } else {
    u32 deadline = now - age + T;
    if (!next || deadline < next) next = deadline;
  }
}
// probe (B.3)
__device__ static inline u64 probe_draw(const TickP& tp, u32 gid, u32 j) { return mix64(tp.probe_base ^ ((u64)gid * 32u + j)); }
__device__ static inline u32 draw_below(u64 draw, u32 n) { return (u32)(((draw >> 32) * (u64)n) >> 32); }
__device__ static inline bool leg_lost(const TickP& tp, u32 gid, u32 j) {
  return tp.loss_u32 && (u32)(probe_draw(tp, gid, j) >> 32) < tp.loss_u32;
}
__device__ static inline bool up_of(const Dev& d, u32 gid) { return (d.upmap[gid >> 5] >> (gid & 31)) & 1u; }
__device__ static void swim_probe(const Ctx& c, Node& n, const TickP& tp, const uint4* base, Ins& ins) {
  const Dev& d = c.d;
  // SIM_CF_AWARENESS_PROBE: the probe interval scales with the health score (memberlist probeNode: ScaleTimeout)
#ifndef TICK_LEAN
  if (d.aw_probe && ((c.tick + (c.gid >> 6)) / d.PI) % (n.awareness + 1u)) return;
#endif
  u32 t = draw_below(probe_draw(tp, c.gid, PD_TARGET), d.N - 1);
  if (t >= c.gid) ++t;
  uint4* p = view_ptr(c, t);
  uint4 e = p ? p[0] : base[(size_t)t * 2];
  if (!(e.w & SIM_VB_KNOWN)) return;
  u32 sw = SIM_VB_SWIM(e.w);
  if (sw == SIM_SWIM_DEAD || sw == SIM_SWIM_LEFT) return;
  bool ok = false;
  if (up_of(d, t)) {
    ok = !leg_lost(tp, c.gid, PD_PING) && !leg_lost(tp, c.gid, PD_ACK);
    for (u32 j = 0; !ok && j < d.ic && j < 4; ++j) {
      u32 r = draw_below(probe_draw(tp, c.gid, PD_RELAY0 + 5 * j), d.N);
      if (r == c.gid || r == t || !up_of(d, r)) continue;
      ok = !leg_lost(tp, c.gid, PD_RELAY0 + 5 * j + 1) && !leg_lost(tp, c.gid, PD_RELAY0 + 5 * j + 2) &&
           !leg_lost(tp, c.gid, PD_RELAY0 + 5 * j + 3) && !leg_lost(tp, c.gid, PD_RELAY0 + 5 * j + 4);
    }
  }
  if (!ok && d.tcp_fallback && up_of(d, t)) ok = true;  // SIM_CF_TCP_FALLBACK: the stream ping next to the indirect ones gets through
  if (ok) { aw_delta(n, -1); return; }
  if (d.nacks) {  // SIM_CF_NACKS: the score rises by the relays that were asked and did not nack (none asked: + 1)
    int expected = 0, nk = 0;
    for (u32 j = 0; j < d.ic && j < 4; ++j) {
      u32 r = draw_below(probe_draw(tp, c.gid, PD_RELAY0 + 5 * j), d.N);
      if (r == c.gid || r == t) continue;
      ++expected;
      if (up_of(d, r) && !leg_lost(tp, c.gid, PD_RELAY0 + 5 * j + 1) && !leg_lost(tp, c.gid, PD_RELAY0 + 5 * j + 4)) ++nk;
    }
    aw_delta(n, expected ? expected - nk : 1);
  } else aw_delta(n, +1);
  if (!p) {  // no view slot to hold the suspicion yet: taken up next tick, once the target has one (SIM_OP_SUSPECT)
#ifndef TICK_LEAN
    u32 i = atomicAdd(d.sreq, 1u);
    if (i < SIM_SUSPECT_REQ_MAX) { d.sreq[1 + 2 * i] = c.gid; d.sreq[2 + 2 * i] = t; }
    if (d.sreq_hh && i < SREQ_HEAD) { d.sreq_hh[2 * i + 1] = t; d.sreq_hh[2 * i] = c.gid; }
#endif
    return;
  }
  bool dirty = false;
  swim_suspect(c, n, t, e.z, c.gid, wire_meta(SIM_K_SUSPECT, 0, 32), p, e, dirty, ins);
}

// SIM_CF_RANDOM_FANOUT: a node's gossip targets of a tick (memberlist kRandomNodes, App. B.2; oracle rf_draw) — uniform over all N
// nodes, skip self and duplicates, up to 3 N tries: a function of (seed, tick, node).  Returns how many of the feff slots found one.
__device__ static inline u32 rf_draw(u64 rb, u32 gid, u32 N, u32 feff, u32 (&chosen)[SIM_MAX_FANOUT]) {
  u32 nc = 0;
  chosen[0] = chosen[1] = chosen[2] = chosen[3] = NOSLOT;
  for (u32 i = 0; i < 3u * N && nc < feff; ++i) {
    u32 t = (u32)(((mix64(rb ^ ((u64)gid * 4096u + i)) >> 32) * (u64)N) >> 32);
    bool dup = t == gid;
#pragma unroll
    for (u32 j = 0; j < SIM_MAX_FANOUT; ++j) dup |= (j < nc) & (chosen[j] == t);
    if (!dup) {
      if (nc == 0) chosen[0] = t; else if (nc == 1) chosen[1] = t; else if (nc == 2) chosen[2] = t; else chosen[3] = t;
      ++nc;
    }
  }
  return nc;
}
// ---- SerfDelegate::notify_message: delegate.rs:183-300 -----------------------------------------------
__device__ static inline bool member_kind(u32 kind) {
  return kind == SIM_K_JOIN || kind == SIM_K_LEAVE || kind >= SIM_K_ALIVE;
}
// address of the state a record is checked against (null: nothing to look at)
// Branch-free on the per-lane values (divergent branches cost scalar exec-mask work on every
// record): the ring index and the view address are both formed, then selected by kind.
// The three plane bases, pinned in scalar registers at kernel entry.  Without this the compiler turns
// `isring ? (isq ? d.qring : d.ering) : d.view` into ONE per-lane load from the kernel-argument segment at a selected
// offset: a global-memory round trip between the slot lookup and the head load of every packet.
// (The same happens to a struct of the three pointers — it goes to scratch and is indexed there — so the rings travel
// as byte distances from the view plane: differences of pinned values are not loads and cannot be folded into one.)
__device__ static inline u64 pin_uniform(const void* p) {
  u32 lo = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(uintptr_t)p), hi = (u32)__builtin_amdgcn_readfirstlane((int)(u32)((uintptr_t)p >> 32));
  return ((u64)hi << 32) | lo;
}
__device__ static __forceinline__ uint4* lookup_ptr(const Ctx& c, u64 vbase, u64 eoff, u64 qoff, u32 kind, u32 key, u64 val, u32 slot) {
  const Dev& d = c.d;
  bool isq = kind == SIM_K_QUERY, isring = isq || kind == SIM_K_EVENT;
  u32 B = isq ? d.Bq : d.Bev, mask = isq ? d.bq_mask : d.bev_mask;
  u32 idx = (u32)val & mask;
  if (!(d.bev_mask && d.bq_mask)) idx = ring_idx(val, B, mask);  // uniform: a ring size that is not a power of two
  size_t row = isring ? (size_t)idx : (size_t)slot;
  // (through a global-address-space pointer: a bare integer -> pointer cast would make every access FLAT)
  uint4* basep = (uint4*)(__attribute__((address_space(1))) uint4*)(vbase + (isring ? (isq ? qoff : eoff) : 0ull));
  bool none = !isring && (kind == SIM_K_EMPTY || slot == NOSLOT);
  uint4* p = basep + (row * d.Nl + c.l);
  return none ? nullptr : p;
}
// slot of a member record's subject (NOSLOT for other kinds and for ids out of range)
__device__ static __forceinline__ u32 slot_load(const Dev& d, u32 kind, u32 key) {
  u32 s = NOSLOT;
  if (member_kind(kind) && key < d.N) s = d.slot_of[key];
  return s;
}
// Reconnector (base.rs:612-681) — see oracle reconnect_run: with probability failed / alive the node attempts a
// memberlist.join with one of its failed members, drawn uniformly (the idx-th in subject order).  A push-pull needs both
// nodes, so the attempt goes on the tick's request list, (node, target | 1 << 31), and the host replays it two ticks later
// as SIM_OP_RECONNECT (sim_step_begin).  Rare: once per failed member, reconnect interval and CLUSTER.
__device__ static void reconnect_run(const Ctx& c, const Node& n, const TickP& tp, u32 n_slots) {
  const Dev& d = c.d;
  u32 nf = n.nfailed, gone = nf + n.nleft, alive = n.nknown > gone ? n.nknown - gone : 0u;
  if (!alive) alive = 1u;
  u32 r = (u32)(probe_draw(tp, c.gid, PD_RECONNECT) >> 32);
  if ((u64)r * alive > ((u64)nf << 32)) return;  // "forgoing reconnect for random throttling"
  u32 idx = draw_below(probe_draw(tp, c.gid, PD_RECONNECT + 1), nf), target = NOSLOT;
#pragma unroll 1
  for (u32 wi = 0; wi < n_slots; ++wi) {
    u32 a = d.walk[wi];
    uint4 e = view_slot_ptr(c, a)[0];
    if (!(e.w & SIM_VB_KNOWN) || SIM_VB_STATUS(e.w) != SIM_STATUS_FAILED) continue;
    if (idx-- == 0) { target = d.subject_of[a]; break; }
  }
  if (target == NOSLOT || target == c.gid) return;
  u32 i = atomicAdd(d.sreq, 1u);
  if (i < SIM_SUSPECT_REQ_MAX) { d.sreq[1 + 2 * i] = c.gid; d.sreq[2 + 2 * i] = target | SREQ_RECONNECT; }
  if (d.sreq_hh && i < SREQ_HEAD) { d.sreq_hh[2 * i + 1] = target | SREQ_RECONNECT; d.sreq_hh[2 * i] = c.gid; }
}
// Reaper::run (base.rs:483-610; reap! 521-553; reap_intents 1820-1822) — see oracle reap_run
__device__ static void reap_run(const Ctx& c, Node& n, u32 n_slots) {
  const Dev& d = c.d;
  u32 now = c.tick, next = 0;
#pragma unroll 1
  for (u32 wi = 0; wi < n_slots; ++wi) {  // in subject order: independent of how the slots were handed out
    u32 a = d.walk[wi];
    uint4* p = view_slot_ptr(c, a);
    uint4 e = p[0];
    u32 age = (now - SIM_VB_STAMP(e.w)) & STAMP_MASK, timeout;
    if (e.w & SIM_VB_KNOWN) {
      u32 st = SIM_VB_STATUS(e.w);
      if (st == SIM_STATUS_FAILED) timeout = d.reconnect_timeout;
      else if (st == SIM_STATUS_LEFT) timeout = d.tombstone_timeout;
      else continue;
      if (age > timeout) { erase_member(c, n, p, e, d.subject_of[a]); continue; }
    } else if (SIM_VB_INTENT(e.w) && d.intent_timeout) {
      timeout = d.intent_timeout;
      if (age > timeout) { p[0] = make_uint4(0, 0, 0, 0); p[d.vtail] = make_uint4(0, 0, 0, 0); continue; }
    } else {
      continue;
    }
    u32 due = now - age + timeout + 1u;
    if (!next || due < next) next = due;
  }
  n.reap_next = next;
  n.dirty |= DR3;
}
// QueueChecker (base.rs:683-740) — see oracle queue_check.  The entries of a class are contiguous
// in the sorted key array; the ones that drain last go.
__device__ static void queue_check(const Ctx& c, Node& n, SK sk) {
  const Dev& d = c.d;
  u32 mx = d.max_queue_depth;
  if (d.min_queue_depth > 0) mx = max(2u * n.nknown, d.min_queue_depth);
  bool changed = false;
#pragma unroll 1
  for (u32 cls = 1; cls <= 3; ++cls) {
    u32 cnt = 0;
#pragma unroll
    for (int i = 0; i < (int)SIM_Q; ++i) cnt += (sk[i] != KEMPTY && (sk[i] >> 26) == cls) ? 1u : 0u;
#pragma unroll
    for (int i = SIM_Q - 1; i >= 0; --i) {
      bool drop = cnt > mx && sk[i] != KEMPTY && (sk[i] >> 26) == cls;
      if (drop) { n.used &= ~(1u << (sk[i] & 15u)); sk[i] = KEMPTY; --cnt; changed = true; }
    }
  }
  if (__any(changed)) {  // re-sort
    n.dirty |= changed ? DR2 : 0u;
    sort16(sk);
  }
}

// Fast classification of one record against the prefetched head `e` of the state it is checked
// against (null-ness of the lookup in `has`).  Returns true when the handler would change nothing
// but the Lamport clock it witnesses — a duplicate, an old message, a subject without a view slot —
// which is the fate of ~95 % of all records; the caller then applies the witness and is done.
// Anything else (a new rumour, a refutation, a confirmation...) is left to the full handlers.
// The conditions are the early `return false` exits of the handlers, in the handlers' order.
__device__ static __forceinline__ bool fast_noop(const Ctx& c, const Node& n, u32 kind, const uint4& r, bool has, const uint4& e) {
  // Straight-line predicate logic (& and |, no ?: on booleans, no short-circuit): this runs for
  // every record of every node, and the lane masks combine on the scalar unit.
  const Dev& d = c.d;
  u64 lt = (u64)r.z | ((u64)r.w << 32), elt = E_LTIME(e);
  bool isev = kind == SIM_K_EVENT, isq = kind == SIM_K_QUERY, isring = isev | isq;
  bool isjl = (kind - SIM_K_JOIN) < 2u, isswim = (kind - SIM_K_ALIVE) < 3u;
  bool known = e.w & SIM_VB_KNOWN;
  // rings — handle_user_event base.rs:750-837, handle_query base.rs:972-1073
  u64 clk = isev ? n.eclock : n.qclock, B = isev ? d.Bev : d.Bq;
  u64 cur = lt >= clk ? lt + 1 : clk, lo = cur - B;
  bool old = (cur > B) & ((isev & (lt < lo)) | (isq & (B < lo)));
  bool dup = (e.z != 0) & (isev | (elt == lt)) & ((e.z == r.x) | (e.w == r.x));
  bool ring_fast = !(n.flags & SIM_RF_MINTIME) & (old | dup);
  // intents — base.rs:1338-1373, 1442-1572
  bool jl_fast = !has | ((lt <= elt) & (known | (SIM_VB_INTENT(e.w) != 0)));
  // memberlist — App. B.4
  u32 sw = SIM_VB_SWIM(e.w);
  bool self = r.x == c.gid;
  bool alive_fast = (self & (r.z <= n.inc)) | (!self & known & (r.z <= e.z));
  bool gone = sw >= SIM_SWIM_DEAD;  // dead or left
  bool sd_fast = !known | (r.z < e.z) | gone | ((kind == SIM_K_SUSPECT) & (sw == SIM_SWIM_SUSPECT) & (SIM_VB_NCONF(e.w) >= d.kconf));
  bool swim_fast = !d.swim | !has | ((kind == SIM_K_ALIVE) ? alive_fast : sd_fast);
  return (isring & ring_fast) | (isjl & jl_fast) | (isswim & swim_fast) | !(isring | isjl | isswim);
}
// A record can be retired without a handler when it is a no-op (fast_noop) that does not even advance the Lamport
// clock it witnesses (it has been seen before: the common duplicate).  The property survives whatever the handlers of
// earlier records do to the node — clocks and incarnations only grow — as long as they leave the record's own entry alone.
__device__ static __forceinline__ bool fast_retire(const Ctx& c, const Node& n, u32 kind, const uint4& r, bool has, const uint4& e) {
  u64 lt = (u64)r.z | ((u64)r.w << 32);
  bool adv = ((kind == SIM_K_EVENT) & (lt >= n.eclock)) | ((kind == SIM_K_QUERY) & (lt >= n.qclock)) |
             (((kind - SIM_K_JOIN) < 2u) & (lt >= n.clock));
  return fast_noop(c, n, kind, r, has, e) & !adv;
}
// The second look of the random fan-out's classification: `t` = the TAIL of the entry whose head `e` could not settle the record
// (ring bucket keys k2 .. k5; a view entry's confirmers).  True when the handler would change nothing, not even a clock:
//   user event / query  the key is in the tail (bucket_add finds it: "seen"); a query only in a bucket of its own Lamport time
//                       (quirk Q2: in a bucket of another time every copy is appended again)
//   suspect             a suspicion is running and `from` is among the confirmers counted so far (swim_suspect: return)
__device__ static __forceinline__ bool tail_retire(const Ctx& c, const Node& n, u32 kind, const uint4& r, const uint4& e, const uint4& t) {
  const u64 lt = (u64)r.z | ((u64)r.w << 32);
  const bool isev = kind == SIM_K_EVENT, isq = kind == SIM_K_QUERY;
  const u64 clk = isev ? n.eclock : n.qclock, B = isev ? c.d.Bev : c.d.Bq;
  const u64 cur = lt >= clk ? lt + 1 : clk;
  const bool old = (cur > B) & ((isev & (lt < cur - B)) | (isq & (B < cur - B)));  // (retired by the first look already; kept so that the two agree)
  const bool in_tail = (t.x == r.x) | (t.y == r.x) | (t.z == r.x) | (t.w == r.x);
  const bool ring_ok = (isev | isq) & !(n.flags & SIM_RF_MINTIME) & (lt < clk) & (e.z != 0u) & (old | ((isev | (E_LTIME(e) == lt)) & in_tail));
  const u32 k = SIM_VB_NCONF(e.w), from = r.w;
  const bool conf = (t.x == from) | ((k >= 1u) & (t.y == from)) | ((k >= 2u) & (t.z == from)) | ((k >= 3u) & (t.w == from));
  const bool susp_ok = (kind == SIM_K_SUSPECT) & ((e.w & SIM_VB_KNOWN) != 0u) & (SIM_VB_SWIM(e.w) == SIM_SWIM_SUSPECT) & (r.z >= e.z) & conf;
  return ring_ok | susp_ok;
}
__device__ static __forceinline__ void dispatch(const Ctx& c, Node& n, const uint4& r, uint4* p, uint4& e, bool& dirty, Ins& ins) {
  u32 kind = SIM_META_KIND(r.y), flags = SIM_META_FLAGS(r.y);
  u64 val = (u64)r.z | ((u64)r.w << 32);
  bool rb = false;
  if (kind == SIM_K_EVENT) rb = handle_user_event(c, n, r.x, val, p, e, dirty);
  else if (kind == SIM_K_QUERY) rb = handle_query(c, n, r.x, val, flags, p, e, dirty);
  else if (kind == SIM_K_JOIN) rb = handle_join_intent(c, n, r.x, val, p, e, dirty);
  else if (kind == SIM_K_LEAVE) rb = handle_leave_intent(c, n, r.x, val, flags & SIM_F_PRUNE, p, e, dirty, ins);
  else if (c.d.swim) {  // memberlist's own broadcasts are handled below the serf delegate
    if (kind == SIM_K_ALIVE) swim_alive(c, n, r.x, r.z, r.y, p, e, dirty, ins);
    else if (kind == SIM_K_SUSPECT) swim_suspect(c, n, r.x, r.z, r.w, r.y, p, e, dirty, ins);
    else if (kind == SIM_K_DEAD) swim_dead(c, n, r.x, r.z, r.w, r.y, p, e, dirty, ins);
  }
  if (rb) ins_set(ins, r.x, r.y, val);  // re-queue the original message unchanged (delegate.rs:294-300)
}
__device__ static inline uint4 sel4(u32 i, const uint4& a, const uint4& b, const uint4& c, const uint4& d) {
  return make_uint4(SEL4(i, a.x, b.x, c.x, d.x), SEL4(i, a.y, b.y, c.y, d.y), SEL4(i, a.z, b.z, c.z, d.z), SEL4(i, a.w, b.w, c.w, d.w));
}

// ------------------------------------------------------------------------------------------------
// the tick kernel
// ------------------------------------------------------------------------------------------------
#ifdef TICK_TIMING
__device__ unsigned long long g_tt[32];
// wave-uniform accumulation in scalar registers; one set of atomics per wave at the very end
#define TT(i) do { unsigned long long t_ = __builtin_amdgcn_s_memtime(); tacc[i] += t_ - tprev; tprev = t_; } while (0)
#define TCNT(i, v) do { tacc[i] += (v); } while (0)  // wave-uniform event counts next to the cycle counters (12 .. 15)
#else
#define TT(i)
#define TCNT(i, v)
#endif
#ifdef TICK_ABLATE
static u32 g_ablate = 0;
#define ABL(bit) (tp.abl & (bit))
#else
#define ABL(bit) false
#endif
#ifndef TICK_OCC
#define TICK_OCC 4
#endif
// A record between its 16-byte working form {key, wire meta, val} and its 12 bytes in a packet (include/serf_sim.h
// sim_packet: key, value bits 31..0, value bits 47..32 | len64 | kind | flags; SUSPECT / DEAD carry inc : 24 | from : 24)
#define PK_U4 3u  // a packet cell is three uint4: the four keys, the four low words, the four high words
#define RF_TMAX 512u   // random fan-out, balanced classification: incoming packets of one wave's 64 nodes the LDS tables hold (mean 256)
#define RF_STASH 208u  // ... and records in need of a handler whose unpacked form and entry pointer are kept for it (mean 75; a rumour's
                       // wavefront brings 300 and more — profiles/r05_tick_series_before.json —: as many as 10 KiB of LDS per wave hold)
#define RF_LDS_U4 ((4u * RF_TMAX + 2u * RF_TMAX + 8u * RF_STASH + 32u * TBLOCK + 16u * RF_STASH) / 16u)  // sum | own | sidx | st_p | nst | st_r
#define RF_CELL_U4 4u  // random fan-out: a sender's cell is 64 bytes — the packet's 48 and, in cell 0, the sender's map word (slot -> cell): entry -> cell is ONE scattered line
__device__ static inline uint4 wire_unpack(u32 key, u32 lo, u32 hm) {
  u32 meta = (((hm >> 8) & 0x3Fu) << 18) | (hm & 0xFFu), hi = hm >> 16;
  bool two = ((hm >> 5) & 7u) == 3u;  // kind 6 or 7
  return make_uint4(key, meta, two ? (lo & 0xFFFFFFu) : lo, two ? ((lo >> 24) | (hi << 8)) : hi);
}
__device__ static inline void wire_pack(const uint4& r, u32& key, u32& lo, u32& hm) {
  bool two = ((r.y >> 5) & 7u) == 3u;
  key = r.x;
  lo = two ? ((r.z & 0xFFFFFFu) | (r.w << 24)) : r.z;
  hm = ((two ? (r.w >> 8) : r.w) << 16) | (((r.y >> 18) & 0x3Fu) << 8) | (r.y & 0xFFu);
}
// (B64: local mode with 64-node blocks — the sharded instantiations read tp.B at run time and pass false)
// (MP: packets of more than one page, sim_config.pkt_records > SIM_P: the deliver loop walks the pages of a packet, the
//  drain takes up to d.P entries per packet; with MP = false all of that folds back to the one-page kernel)
// (the body of the kernel as a function of the block index: written this way the compiler keeps 76 instead of 116 bytes of
// scratch per lane — 2 % of the tick, profiles/r03_experiments.md)
template <bool SHARDED, int F, bool B64, bool MP, bool RF>
__device__ __forceinline__ void tick_block(const Dev& d, const TickP& tp, const TickP& ptp, const u32 cur, const uint4* base, const u32 chunk, const u32 cnt, const u32 bx) {
#ifdef TICK_TIMING
  unsigned long long tacc[32] = {0};
  unsigned long long tprev = __builtin_amdgcn_s_memtime();
#endif
  // LDS staging of the per-node inbox: the packet being delivered, the slot of each record's
  // subject's entry and the head of that entry (10 KiB per wave: 16 waves per CU = all of its 160 KiB)
  // (one allocation, carved: the RF one-page instantiation stages nothing per record — its tables fit lds_r and lds_e, 8 KiB per
  // wave = 20 waves per CU —, so for it lds_p is only a name for lds_r's memory that dead code refers to)
  constexpr bool kNoP = RF && !MP;
  __shared__ uint4 lds_all[kNoP ? RF_LDS_U4 : 2 * SIM_P * TBLOCK + SIM_P * TBLOCK / 2];
  static_assert(RF_LDS_U4 * 16u <= 10240u && RF_LDS_U4 >= 3u * TBLOCK && RF_STASH < 255u && (6u * RF_TMAX + 8u * RF_STASH) % 16u == 0u, "RF tables: 10 KiB per wave (16 waves per CU), store_cell's three columns, 8-bit stash indices");
  uint4 (&lds_r)[SIM_P][TBLOCK] = *reinterpret_cast<uint4 (*)[SIM_P][TBLOCK]>(&lds_all[0]);
  uint4 (&lds_e)[SIM_P][TBLOCK] = *reinterpret_cast<uint4 (*)[SIM_P][TBLOCK]>(&lds_all[SIM_P * TBLOCK]);
  uint4* (&lds_p)[SIM_P][TBLOCK] = *reinterpret_cast<uint4* (*)[SIM_P][TBLOCK]>(&lds_all[kNoP ? 0 : 2 * SIM_P * TBLOCK]);  // where each record's entry lives (null: nothing to look at)
  const u32 tid = threadIdx.x;
  // one launch covers `cnt` nodes: the whole shard (chunk == ~0), or sender chunk `chunk` of a sharded run = the nodes
  // whose offset inside their vblock lies in sub-slab `chunk` (V ranges of `sub` consecutive nodes)
  const u32 idx = bx * TBLOCK + threadIdx.x;
  if (idx == 0 && d.swim) *d.sreq_next = 0;  // the next tick's request list starts empty (its buffer was read a tick ago)
  if (idx >= cnt) return;
  u32 l = idx;
  if (SHARDED && chunk != 0xFFFFFFFFu) {
    u32 b = idx / tp.sub;
    l = b * tp.blk + chunk * tp.sub + (idx - b * tp.sub);
  }
#ifdef TICK_ABLATE
  if (ABL(0xFF00u) && bx < 1024u) {  // experiment: stagger the first generation of blocks
    u32 slot = (bx >> 8) & 3u, per = (tp.abl >> 8) & 0xFFu;
    for (u32 i = 0; i < slot * per; ++i) __builtin_amdgcn_s_sleep(127);
  }
#endif
  u32 gid = d.shard0 + l;
  // V == 1 (one shard holds everything) is wave-uniform: no divisions by run-time values on that path
  u32 g = tp.V == 1 ? 0u : gid / tp.M, ll = gid - g * tp.M;
  Ctx c{d, l, gid, (u32)tp.tick, tp.query_base};
  const u64 vbase = pin_uniform(d.view), eoff = pin_uniform(d.ering) - vbase, qoff = pin_uniform(d.qring) - vbase;
  const uint4 zero = make_uint4(0, 0, 0, 0);
  // Local mode: the packet of fan-out slot k is fetched from its sender (Dev::obox).  The receiver is (vblock rbb,
  // sub-slab rs, offset rr); the senders of a wave's 64 consecutive receivers are the 64 nodes of ONE block (the map of
  // the PREVIOUS tick, `ptp`, run backwards: block pi^-1(pi(j2) - off_k), positions XOR-scrambled), so the block
  // permutation runs on the scalar unit and the wave reads one 3 KiB run.
  // (B64, 64-node blocks: pi(j2) and the four sender blocks are wave-uniform and pinned in SGPRs; small or ragged
  // shards, B = 1, work the four senders out once through the general per-node form of the map.)
  u32 ry = 0, sj0 = 0, sj1 = 0, sj2 = 0, sj3 = 0;
  if (!SHARDED && B64 && !RF && tp.feff) {
    u32 ru2 = ll;  // index inside the receiving chunk: (vblock, offset)
    if (tp.V != 1 || tp.C != 1) {
      u32 rbb = ll / tp.blk, w = ll - rbb * tp.blk;
      ru2 = rbb * tp.sub + w % tp.sub;
    }
    ry = (u32)__builtin_amdgcn_readfirstlane((int)pi_f(ptp, (u32)__builtin_amdgcn_readfirstlane((int)(ru2 >> 6))));
    auto blk_of = [&](u32 k) __attribute__((always_inline)) -> u32 {
      if (k >= tp.feff) return 0u;
      u32 yy = ry >= ptp.off[k] ? ry - ptp.off[k] : ry + tp.nbc - ptp.off[k];
      return (u32)__builtin_amdgcn_readfirstlane((int)pi_inv(ptp, yy));
    };
    sj0 = blk_of(0); sj1 = blk_of(1); sj2 = blk_of(2); sj3 = blk_of(3);
  }
  if (!SHARDED && !B64 && !RF) {
#pragma unroll 1
    for (u32 k = 0; k < tp.feff; ++k) {
      u32 gs, sl;
      fan_source_g(ptp, PICK4(ptp.off, k), PICK4(ptp.rot, k), PICK4(ptp.rho, k), g, ll, k, gs, sl);
      u32 v = gs * tp.M + sl;
      if (k == 0) sj0 = v; else if (k == 1) sj1 = v; else if (k == 2) sj2 = v; else sj3 = v;
    }
  }
  auto src_of = [&](u32 k) __attribute__((always_inline)) -> u32 {  // local index of the node whose k-th packet is addressed to this one
    if (!B64) return SEL4(k, sj0, sj1, sj2, sj3);
    u32 u = SEL4(k, sj0, sj1, sj2, sj3) * 64u + ((ll & 63u) ^ fan_scramble(ry, k));
    if (tp.V == 1 && tp.C == 1) return u;
    u32 rbb = ll / tp.blk, w = ll - rbb * tp.blk, rs = w / tp.sub;
    u32 bb0 = u / tp.sub, r0 = u - bb0 * tp.sub;
    u32 rho = PICK4(ptp.rho, k), s0 = rs >= rho ? rs - rho : rs + tp.C - rho;
    u32 gs = (g + (rbb + PICK4(ptp.rot, k)) % tp.V) % tp.V;
    return gs * tp.M + bb0 * tp.blk + s0 * tp.sub + r0;
  };
  // local mode: byte k = where the sender of slot k put that packet: first page << 2 | pages - 1 (0xFF: nothing sent)
  u32 jw = 0xFFFFFFFFu;
  // page pg of the packet of fan-out slot k (sharded: the block of the receive buffer the source shard filled)
  // RF (literal kRandomNodes): the node's incoming packets are the entries rin0 .. rin0 + rcnt of the tick's CSR (Dev::rsrc:
  // sender * 4 + slot, in (sender, slot) order: the order the oracle hands them over in); "slot k" of the deliver loop is then
  // the k-th incoming packet, and the loop runs as often as the lane of the wave with the most packets needs (in-degree is
  // Poisson-like: mean f).  A packet is FETCHED from its sender, like in the bijection's local mode — but here a sender's cells
  // are 64 bytes and cell 0 carries the sender's map word in its spare quarter, so entry -> cell is all there is: ONE
  // scattered line per packet (99 % of the packets sit in their sender's cell 0; another cell, or a further page, is a second
  // fetch).  rf_e = the entry of the packet whose first page is in rn .. / rf_map, rf_en = the next one (requested an
  // iteration ahead), rf_jb = where the packet being delivered sits (first page << 2 | pages - 1; 0xFF: nothing).
  u32 rin0 = 0, rcnt = 0, rf_e = NOSLOT, rf_en = NOSLOT, rf_map = 0, rf_jb = 0xFFu, rf_snd = 0;
  if (RF) { rin0 = d.rcsr[l]; rcnt = d.rcsr[l + 1] - rin0; }
  auto cell_of = [&](u32 k, u32 pg) __attribute__((always_inline)) -> const uint4* {
    if (RF) {  // first page: the sender's cell 0, asked for before its map word is known; further pages: where the map said
      if (!MP || pg == 0) return rf_e == NOSLOT ? d.nullcell : d.rfrd + (size_t)(rf_e >> 2) * RF_CELL_U4;
      return (rf_jb == 0xFFu || pg > (rf_jb & 3u)) ? d.nullcell : d.rfrd + ((size_t)((rf_jb >> 2) + pg) * d.NC + rf_snd) * RF_CELL_U4;
    }
    if (SHARDED) {  // [sender chunk][source shard][slot * PG + page][sub] (oracle xcell)
      u32 b = l / tp.blk, w = l - b * tp.blk, sl = w / tp.sub;
      u32 src = (d.shard_rank + b + tp.prot[k]) % tp.V;
      u32 ch = (sl + tp.C - tp.prho[k]) % tp.C;
      return d.xrecv + ((((size_t)ch * tp.V + src) * d.fp + (MP ? k * d.PG + pg : k)) * tp.sub + (w - sl * tp.sub)) * PK_U4;
    }
    u32 jb = (jw >> (8u * k)) & 0xFFu;
    bool none = jb == 0xFFu || (MP && pg > (jb & 3u));
    return none ? d.nullcell : d.obox[cur] + ((size_t)((jb >> 2) + (MP ? pg : 0u)) * d.Nl + src_of(k)) * PK_U4;
  };
  // pages the wave walks for slot k: the most any of its lanes received (a lane with fewer reads the zero cell)
  auto wave_np = [&](u32 k) __attribute__((always_inline)) -> u32 {
    if (!MP) return 1u;
    if (SHARDED || RF) return d.PG;
    u32 jb = (jw >> (8u * k)) & 0xFFu, np = jb == 0xFFu ? 0u : (jb & 3u) + 1u, w = 1u;
    if (__any(np >= 2u)) w = 2u;
    if (__any(np >= 3u)) w = 3u;
    if (__any(np >= 4u)) w = 4u;
    return w;
  };
  // The first packet is requested together with the node's row (it does not depend on it: a node that turns out to be
  // down has loaded 48 bytes for nothing), every further one a packet ahead: keys, low words, high words.  Local mode:
  // together with the four senders' map words; a sender's first packet can only be in its cell 0, so that cell is
  // requested before the map word is known and dropped if the word says "nothing sent".
  const uint4* cell;
  u32 om0 = 0xFFFFFFFFu, om1 = 0xFFFFFFFFu, om2 = 0xFFFFFFFFu, om3 = 0xFFFFFFFFu;
  u32 rf_npk = 0;  // RF: packets the wave walks (the most any lane received)
  // RF, one page per packet: BALANCED classification (below) when the whole wave is here and its packets fit the LDS tables
  bool bal = false;   // (wave-uniform)
  u32 rb_base = 0, rb_T = 0;
  if (RF) {
    u32 w = rcnt;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) w = max(w, (u32)__shfl_xor((int)w, o, 64));
    rf_npk = w;
    if (!MP && (bx + 1u) * TBLOCK <= cnt) {
      rb_base = (u32)__builtin_amdgcn_readfirstlane((int)rin0);
      rb_T = (u32)__builtin_amdgcn_readlane((int)(rin0 + rcnt), 63) - rb_base;
      bal = rb_T <= RF_TMAX && !ABL(2) && !ABL(64);
    }
    if (MP) {
      rf_e = rcnt > 0 ? d.rsrc[rin0] : NOSLOT;
      rf_en = rcnt > 1 ? d.rsrc[rin0 + 1] : NOSLOT;
    }
    cell = MP ? cell_of(0, 0) : d.nullcell;
  } else if (SHARDED) cell = tp.first ? d.nullcell : cell_of(0, 0);
  else {
    // (the four senders first, then the four loads back to back from selected addresses: a load inside a branch gets
    // its own s_waitcnt — four round trips before the row was even asked for)
    const u32* om = d.omap[cur];
    const u32* none = reinterpret_cast<const u32*>(d.nullcell);
    const u32 s0 = src_of(0), s1 = src_of(1), s2 = src_of(2), s3 = src_of(3);
    const u32 *a0 = tp.feff > 0 ? om + s0 : none, *a1 = tp.feff > 1 ? om + s1 : none;
    const u32 *a2 = tp.feff > 2 ? om + s2 : none, *a3 = tp.feff > 3 ? om + s3 : none;
    cell = tp.feff ? d.obox[cur] + (size_t)s0 * PK_U4 : d.nullcell;
    om0 = *a0; om1 = *a1; om2 = *a2; om3 = *a3;
  }
  uint4 rn = ld4(cell), rn1 = ld4((SHARDED && tp.first) ? cell : cell + 1), rn2 = ld4((SHARDED && tp.first) ? cell : cell + 2);
  if (RF) rf_map = reinterpret_cast<const u32*>(cell)[12];  // (the zero cell: "cell 0, one page" of nothing)
  Node n;
  node_load(d, l, n);
  if (!SHARDED && !RF) {
    jw = (tp.feff > 0 ? om0 & 0xFFu : 0xFFu) | (tp.feff > 1 ? om1 & 0xFF00u : 0xFF00u) |
         (tp.feff > 2 ? om2 & 0xFF0000u : 0xFF0000u) | (tp.feff > 3 ? om3 & 0xFF000000u : 0xFF000000u);
    if ((jw & 0xFFu) == 0xFFu) rn = rn1 = rn2 = zero;
  }
  bool up = n.flags & SIM_RF_UP;
  TT(0);
  // ---- RF: balanced classification.  With a random in-degree a node-per-lane deliver loop runs as often as the wave's
  // busiest node has packets (9.4 times at 1 Mi nodes, fan-out 4) at 4 / 9.4 of its lanes — and the loop is issue-bound.  But
  // the wave's packets are ONE run of the tick's CSR (its nodes' rows, back to back): here lane j takes packet base + 64 r + j,
  // whoever it is for — 4.2 rounds at full width — and judges its four records against the OWNER's clocks (staged in LDS) the
  // way the deliver loop's fast path does.  What that leaves (~5 % of the records: a new rumour, a refutation ...) is noted per
  // packet (LDS: a mask and, for member records, a hash of the subject) with the unpacked record and its entry pointer in a
  // stash; then every node walks the notes of ITS packets in arrival order and runs the handlers (phase 1 below: one record
  // per lane and iteration, its entry read afresh).  Exact because a record judged to change nothing — not even a clock — at
  // the start of the tick still changes nothing at its turn: clocks and incarnations only grow, ring buckets only gain keys,
  // and a member entry is only ever written by a handler of a record about the same subject — such a later record is sent
  // through the handlers as well (the hash, conservatively).
  // (A wave that is not all here — the last one of a ragged shard — or whose packets do not fit the tables — clusters of a few
  // nodes — skips the classification: every record of every packet goes through the handlers, which is always right.)
  u64 rb_need = 0;  // records of this node's packets 0 .. 15 that need a handler: bit 4 k + q
  u64 rb_dup = 0;   // ... of those, literal copies of an EARLIER record of this node's list (the same rumour in another packet of the tick)
  if (RF && !MP && !bal) rb_need = rcnt >= 16u ? ~0ull : (1ull << (4u * rcnt)) - 1ull;
  if (RF && !MP && bal) {
    u32* const sum = reinterpret_cast<u32*>(&lds_all[0]);           // [RF_TMAX] slow mask | 4 x 6-bit subject hash (0: not a member record)
    uint8_t* const own = reinterpret_cast<uint8_t*>(sum + RF_TMAX); // [RF_TMAX] the lane a packet is for
    uint8_t* const sidx = own + RF_TMAX;                            // [RF_TMAX] first stash entry of the packet's slow records (0xFF: none)
    u64* const st_p = reinterpret_cast<u64*>(sidx + RF_TMAX);       // [RF_STASH] entry pointers
    uint4* const nst = reinterpret_cast<uint4*>(st_p + RF_STASH);   // [64][2] the nodes' clocks, flags, incarnation as the tick begins
    uint4* const st_r = nst + 2 * TBLOCK;                           // [RF_STASH] unpacked records
    const u32 rown = rin0 - rb_base;
#pragma unroll 1
    for (u32 i = 0; i < rf_npk; ++i)
      if (i < rcnt) own[rown + i] = (uint8_t)tid;
    nst[2 * tid] = make_uint4((u32)n.eclock, (u32)(n.eclock >> 32), (u32)n.qclock, (u32)(n.qclock >> 32));
    nst[2 * tid + 1] = make_uint4((u32)n.clock, (u32)(n.clock >> 32), n.flags, n.inc);
    __builtin_amdgcn_wave_barrier();
    TT(1);
    u32 nstash = 0;  // (wave-uniform)
    const u32 l0 = l - tid, g0 = gid - tid;
    const u64 lt_mask = (1ull << tid) - 1ull;
    u32 ea = tid < rb_T ? d.rsrc[rb_base + tid] : NOSLOT, eb = 64u + tid < rb_T ? d.rsrc[rb_base + 64u + tid] : NOSLOT;
    const uint4* cp = ea == NOSLOT ? d.nullcell : d.rfrd + (size_t)(ea >> 2) * RF_CELL_U4;
    uint4 a0 = ld4(cp), a1 = ld4(cp + 1), a2 = ld4(cp + 2);
    u32 am = reinterpret_cast<const u32*>(cp)[12];
#pragma unroll 1
    for (u32 c0 = 0; c0 < rb_T; c0 += 64u) {
      const u32 cc = c0 + tid, e = ea;
      const bool live = cc < rb_T;
      u32 jb = e == NOSLOT ? 0xFFu : (am >> (8u * (e & 3u))) & 0xFFu;
      uint4 ck = a0, cl = a1, ch = a2;
      if (jb == 0xFFu) ck = cl = ch = zero;
      const bool far = jb != 0xFFu && (jb >> 2) != 0u;  // not the sender's cell 0 (1 % of the packets)
      if (__any(far)) {
        const uint4* c2 = far ? d.rfrd + ((size_t)(jb >> 2) * d.NC + (e >> 2)) * RF_CELL_U4 : d.nullcell;
        const uint4 b0 = ld4(c2), b1 = ld4(c2 + 1), b2 = ld4(c2 + 2);
        if (far) { ck = b0; cl = b1; ch = b2; }
      }
      // the next round's cell is asked for behind this round's slot-map loads, the entry of the round after it right away
      ea = eb;
      eb = c0 + 128u + tid < rb_T ? d.rsrc[rb_base + c0 + 128u + tid] : NOSLOT;
      cp = ea == NOSLOT ? d.nullcell : d.rfrd + (size_t)(ea >> 2) * RF_CELL_U4;
      auto prefetch = [&]() __attribute__((always_inline)) { a0 = ld4(cp); a1 = ld4(cp + 1); a2 = ld4(cp + 2); am = reinterpret_cast<const u32*>(cp)[12]; };
      if (!__any(((ch.x | ch.y | ch.z | ch.w) & 0xF0u) != 0)) {  // nothing in any of the round's packets
        prefetch();
        if (live) { sum[cc] = 0; sidx[cc] = 0xFFu; }
        continue;
      }
      const u32 k0 = SIM_META_KIND(ch.x), k1 = SIM_META_KIND(ch.y), k2 = SIM_META_KIND(ch.z), k3 = SIM_META_KIND(ch.w);
      const u32 s0 = slot_load(d, k0, ck.x), s1 = slot_load(d, k1, ck.y), s2 = slot_load(d, k2, ck.z), s3 = slot_load(d, k3, ck.w);
      prefetch();
      const uint4 r0 = wire_unpack(ck.x, cl.x, ch.x), r1 = wire_unpack(ck.y, cl.y, ch.y);
      const uint4 r2 = wire_unpack(ck.z, cl.z, ch.z), r3 = wire_unpack(ck.w, cl.w, ch.w);
      const u32 o = live ? (u32)own[cc] : 0u;
      const Ctx co{d, l0 + o, g0 + o, (u32)tp.tick, tp.query_base};
      uint4* const p0 = lookup_ptr(co, vbase, eoff, qoff, k0, r0.x, (u64)r0.z | ((u64)r0.w << 32), s0);
      uint4* const p1 = lookup_ptr(co, vbase, eoff, qoff, k1, r1.x, (u64)r1.z | ((u64)r1.w << 32), s1);
      uint4* const p2 = lookup_ptr(co, vbase, eoff, qoff, k2, r2.x, (u64)r2.z | ((u64)r2.w << 32), s2);
      uint4* const p3 = lookup_ptr(co, vbase, eoff, qoff, k3, r3.x, (u64)r3.z | ((u64)r3.w << 32), s3);
      const uint4 e0 = ld4(p0 ? p0 : d.nullcell), e1 = ld4(p1 ? p1 : d.nullcell), e2 = ld4(p2 ? p2 : d.nullcell), e3 = ld4(p3 ? p3 : d.nullcell);
      // all four heads are asked for before any of them is looked at: left to itself the scheduler sinks each load to its first
      // use — to save registers — and the round makes three or four trips to memory, one after the other, instead of one
      __builtin_amdgcn_sched_barrier(0);
      const uint4 na = nst[2 * o], nb = nst[2 * o + 1];
      Node no;
      no.eclock = (u64)na.x | ((u64)na.y << 32); no.qclock = (u64)na.z | ((u64)na.w << 32);
      no.clock = (u64)nb.x | ((u64)nb.y << 32); no.flags = nb.z; no.inc = nb.w;
      u32 m = (fast_retire(co, no, k0, r0, p0 != nullptr, e0) ? 0u : 1u) | (fast_retire(co, no, k1, r1, p1 != nullptr, e1) ? 0u : 2u) |
              (fast_retire(co, no, k2, r2, p2 != nullptr, e2) ? 0u : 4u) | (fast_retire(co, no, k3, r3, p3 != nullptr, e3) ? 0u : 8u);
      if (!live || !(no.flags & SIM_RF_UP)) m = 0;  // (packets for a process that is down are dropped)
      // Second look (r5): two kinds of records are no-ops that the 16-byte head cannot show — a user event / query whose key sits
      // in the TAIL of its ring bucket (three and more rumours of one Lamport time share a bucket), and a suspect message from a
      // confirmer the entry has already counted (the confirmers are in the tail).  Both circulate for the whole life of their
      // rumour and reach every node in every packet: 250 of a wave's 256 record positions went to the handlers in such ticks, 20
      // iterations, each of them a no-op (profiles/r05_tick_series_*.json: the 0.5 - 0.6 ms ticks), and the stash overflowed.
      // One candidate per lane and turn, rolled, head (again: it sits in L2) and tail requested together: the four heads are
      // dead by now, so the loop lives in their registers (looking at all four tails at once cost 115 spilled registers).
      // A record retired here is judged like one the first look retires: against the state as the tick began — the walk below
      // lists it again if an earlier listed record is about the same subject.
      {
        auto wants = [&](u32 kind, const uint4& e, u32 bit) __attribute__((always_inline)) -> u32 {
          const bool ring = kind == SIM_K_EVENT || kind == SIM_K_QUERY;
          return (bit != 0u && ((ring && e.z != 0u && e.w != 0u) ||
                                (kind == SIM_K_SUSPECT && (e.w & SIM_VB_KNOWN) && SIM_VB_SWIM(e.w) == SIM_SWIM_SUSPECT))) ? bit : 0u;
        };
        u32 wm = wants(k0, e0, m & 1u) | wants(k1, e1, m & 2u) | wants(k2, e2, m & 4u) | wants(k3, e3, m & 8u);
#pragma unroll 1
        while (__any(wm != 0u)) {
          const u32 q = wm ? (u32)__ffs((int)wm) - 1u : 0u;
          const bool w = wm != 0u;
          wm &= wm - 1u;
          const u32 kq = SEL4(q, k0, k1, k2, k3);
          const uint4 rq = sel4(q, r0, r1, r2, r3);
          uint4* const pq = q == 0u ? p0 : q == 1u ? p1 : q == 2u ? p2 : p3;
          const uint4* const tq = pq + (kq == SIM_K_QUERY ? d.qtail : kq == SIM_K_EVENT ? d.etail : d.vtail);
          const uint4 eh = ld4(w ? pq : d.nullcell), et = ld4(w ? tq : d.nullcell);
          if (w && tail_retire(co, no, kq, rq, eh, et)) m &= ~(1u << q);
          TCNT(24, 1);  // turns of the second look
        }
      }
      // per member record: a 4-bit hash of its subject (+ 1) and, bit 5, whether its handler can UNDO what made a later record
      // about the same subject a no-op (see the walk below): an alive message, a leave intent with the prune flag
      auto hsh = [](u32 kind, u32 key, u32 meta) __attribute__((always_inline)) -> u32 {
        const u32 trig = (kind == SIM_K_ALIVE || (kind == SIM_K_LEAVE && (SIM_META_FLAGS(meta) & SIM_F_PRUNE))) ? 32u : 0u;
        return member_kind(kind) ? (((key * 0x9E3779B1u) >> 28) + 1u) | trig : 0u;
      };
      const u32 note = m | (hsh(k0, r0.x, r0.y) << 4) | (hsh(k1, r1.x, r1.y) << 10) | (hsh(k2, r2.x, r2.y) << 16) | (hsh(k3, r3.x, r3.y) << 22);
      // stash entries for the slow records: a prefix sum of popcount(m) over the lanes, from three ballots
      const u32 pm = (u32)__popc(m);
      const u64 q0 = __ballot(pm & 1u), q1 = __ballot(pm & 2u), q2 = __ballot(pm & 4u);
      const u32 first = nstash + (u32)__popcll(q0 & lt_mask) + 2u * (u32)__popcll(q1 & lt_mask) + 4u * (u32)__popcll(q2 & lt_mask);
      nstash += (u32)__popcll(q0) + 2u * (u32)__popcll(q1) + 4u * (u32)__popcll(q2);
      const bool keep = m != 0u && first + pm <= RF_STASH;
      if (keep) {
        u32 j = first;
        if (m & 1u) { st_r[j] = r0; st_p[j] = (u64)(uintptr_t)p0; ++j; }
        if (m & 2u) { st_r[j] = r1; st_p[j] = (u64)(uintptr_t)p1; ++j; }
        if (m & 4u) { st_r[j] = r2; st_p[j] = (u64)(uintptr_t)p2; ++j; }
        if (m & 8u) { st_r[j] = r3; st_p[j] = (u64)(uintptr_t)p3; ++j; }
      }
      if (live) { sum[cc] = note; sidx[cc] = keep ? (uint8_t)first : (uint8_t)0xFFu; }
      TCNT(13, 1);  // rounds that looked anything up
#ifdef TICK_TIMING
#pragma unroll 1
      for (u32 kd = 1; kd <= 7u; ++kd) {  // records left for the handlers, by kind (17 .. 23)
        u32 w = ((m & 1u) && k0 == kd ? 1u : 0u) + ((m & 2u) && k1 == kd ? 1u : 0u) + ((m & 4u) && k2 == kd ? 1u : 0u) + ((m & 8u) && k3 == kd ? 1u : 0u);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) w += (u32)__shfl_xor((int)w, o, 64);
        tacc[16 + kd] += w;
      }
#endif
    }
    __builtin_amdgcn_wave_barrier();
    TT(2);
    TCNT(14, nstash);  // records left for the handlers by the classification
    // every node over the notes of its own packets, in arrival order: a member record that was judged a no-op against the state
    // as the tick began needs a handler all the same when an EARLIER listed record about the same subject can undo what made it
    // one.  (r5) Which handlers can: a record is retired because its Lamport time / incarnation is not newer than the entry's
    // (both only grow), because the subject is unknown, gone, or fully confirmed, or because its confirmer is counted.  An alive
    // message (a new incarnation: unknown -> known, gone / suspect -> alive) and a leave intent with the prune flag (the entry is
    // erased) undo such a verdict; a suspect, a dead, a join intent or a plain leave intent cannot — they raise the incarnation /
    // the time, count a confirmer, or take the member from alive to suspect to gone, and every one of the verdicts above survives
    // that.  (Until r5 every listed member record re-listed the later ones about its subject: at a suspicion's flood — every
    // packet carries suspect / dead records about the same node — one new confirmation at a node with nine packets made a wave run
    // 19 handler iterations for its one lane: profiles/r05_tick_series_2nd_look.json, ticks 376 - 382.)
    u32 hot = 0;
    u64 fl = 0;   // stash indices of the first copies seen so far (eight of them, a byte each)
    u32 nf = 0;
    const u32 wmax = min(rf_npk, 16u);
#pragma unroll 1
    for (u32 i = 0; i < wmax; ++i) {
      if (i < rcnt) {
        const u32 sn = sum[rown + i];
        u32 m = sn & 15u;
#pragma unroll
        for (u32 q = 0; q < SIM_P; ++q) {
          const u32 hf = (sn >> (4u + 6u * q)) & 63u, hq = hf & 31u;
          if (hq && ((hot >> hq) & 1u)) m |= 1u << q;
          if ((hf & 32u) && ((m >> q) & 1u)) hot |= 1u << hq;
        }
        rb_need |= (u64)m << (4u * i);
        // The same rumour arrives in several of a node's packets of one tick — at a rumour's wavefront every node hears it for the
        // first time from two or three senders at once, and every copy was judged "new" against the state as the tick began
        // (profiles/r05_tick_series_before.json: 300 records per wave and tick for the handlers, 20 iterations).  A literal copy
        // of an earlier record of the node's own list changes nothing once that record's handler has run (the handlers below are
        // idempotent for: user events and queries — the key is in the bucket by then —, join intents and leave intents about others
        // without prune — the entry's Lamport time is the record's by then —, memberlist's alive / suspect / dead about others — the
        // incarnation is the record's, the confirmer is counted, the member is gone by then), PROVIDED nothing of what `poison` watches for happens
        // in between (handler loop below): noted here, skipped there.  Compared in the stash: only stashed records take part.
        const u32 m0 = sn & 15u, si = (u32)sidx[rown + i];
        if (m0 != 0u && si != 0xFFu) {
          u32 mm = m0, j = si;
#pragma unroll 1
          while (mm) {
            const u32 q = (u32)__ffs((int)mm) - 1u;
            mm &= mm - 1u;
            const uint4 r = st_r[j];
            const u32 kd = SIM_META_KIND(r.y);
            const bool elig = kd == SIM_K_EVENT || kd == SIM_K_QUERY || kd == SIM_K_JOIN || (kd >= SIM_K_ALIVE && r.x != gid) ||
                              (kd == SIM_K_LEAVE && !(SIM_META_FLAGS(r.y) & SIM_F_PRUNE) && r.x != gid);
            if (elig) {
              bool isdup = false;
#pragma unroll 1
              for (u32 f = 0; f < nf; ++f) {
                const uint4 a = st_r[(u32)(fl >> (8u * f)) & 0xFFu];
                isdup |= a.x == r.x && a.y == r.y && a.z == r.z && a.w == r.w;
              }
              if (isdup) rb_dup |= 1ull << (4u * i + q);
              else if (nf < 8u) { fl |= (u64)j << (8u * nf); ++nf; }
            }
            ++j;
          }
        }
      }
    }
#ifdef TICK_TIMING
    {
      u32 w = (u32)__popcll(rb_dup);
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) w += (u32)__shfl_xor((int)w, o, 64);
      TCNT(16, w);  // literal copies found, per wave
    }
#endif
    TT(3);
  }
  // ---- phase 1: deliver.  The queue is not touched: handlers park their broadcasts in d.pend.
  if (up && !ABL(2)) {
    if (RF && !MP) {
      // the records the balanced pass left for the handlers, in arrival order, one per lane and iteration
      const u32* const sum = reinterpret_cast<const u32*>(&lds_all[0]);
      const uint8_t* const sidx = reinterpret_cast<const uint8_t*>(sum + RF_TMAX) + RF_TMAX;
      const u64* const st_p = reinterpret_cast<const u64*>(sidx + RF_TMAX);
      const uint4* const st_r = reinterpret_cast<const uint4*>(st_p + RF_STASH) + 2 * TBLOCK;
      const u32 rown = rin0 - rb_base;
      // a record that was not stashed (a later record about a subject of an earlier one; a stash that ran full; a packet
      // beyond a node's sixteenth): entry -> cell -> slot map, on the spot
      auto reload = [&](u32 k, u32 q, uint4& r, uint4*& ptr) __attribute__((always_inline)) {
        const u32 ent = d.rsrc[rin0 + k], snd = ent >> 2;
        const u32 mp = reinterpret_cast<const u32*>(d.rfrd + (size_t)snd * RF_CELL_U4)[12];
        const u32 jb = (mp >> (8u * (ent & 3u))) & 0xFFu;
        r = make_uint4(0, 0, 0, 0);
        ptr = nullptr;
        if (jb != 0xFFu) {
          const u32* cw = reinterpret_cast<const u32*>(d.rfrd + ((size_t)(jb >> 2) * d.NC + snd) * RF_CELL_U4);
          r = wire_unpack(cw[q], cw[4u + q], cw[8u + q]);
          const u32 kind = SIM_META_KIND(r.y);
          ptr = lookup_ptr(c, vbase, eoff, qoff, kind, r.x, (u64)r.z | ((u64)r.w << 32), slot_load(d, kind, r.x));
        }
      };
      // (tried: the head of the NEXT record requested while this record's handler runs — ten more live registers across the
      // handlers, 10 spilled VGPRs, + 6 %: profiles/r04_experiments.md)
      // `poison`: from here on no copy is skipped any more (running a record whose handler changes nothing is always right; not
      // running one is right only while the handlers are idempotent): a model bound was hit (a full ring bucket counts EVERY copy
      // it turns away), a query met a bucket of another Lamport time (quirk Q2: the bucket keeps its time, so every copy is
      // appended again), a leave intent pruned an entry or was about the node itself (erase / refute: not idempotent)
      const u32 ovf0 = n.overflow;
      bool poison = false;
      auto run = [&](const uint4& r, uint4* ptr) __attribute__((always_inline)) {
        uint4 e = ld4(ptr ? ptr : d.nullcell);
        Ins ins;
        ins.has = ins.wide = 0;
        bool dirty = false;
        dispatch(c, n, r, ptr, e, dirty, ins);
        if (ins.has) pend_push(c, n, ins);
        const u32 kd = SIM_META_KIND(r.y);
        poison |= (n.overflow != ovf0) | ((kd == SIM_K_QUERY) & (E_LTIME(e) != ((u64)r.z | ((u64)r.w << 32)))) |
                  ((kd == SIM_K_LEAVE) & (((SIM_META_FLAGS(r.y) & SIM_F_PRUNE) != 0u) | (r.x == gid)));
      };
      // (packets beyond a node's sixteenth — never at any realistic size — have no bits in rb_need: every record of theirs goes
      // through the handlers, after the others, by way of the cursor xcur = 4 k + q)
      u32 xcur = 64u;
      const u32 xend = rcnt > 16u ? 4u * rcnt : 0u;
#pragma unroll 1
      for (;;) {
        // the next record of the list that is not a copy to be skipped; the copies in front of it are dropped with it (they were
        // no-ops at their turn: nothing poisoned the list before them)
        const u64 cand = poison ? rb_need : (rb_need & ~rb_dup);
        if (ABL(32) || !__any(cand != 0 || xcur < xend)) break;
        TCNT(12, 1);
        u32 bit = NOSLOT;
        if (cand) { bit = (u32)__ffsll((unsigned long long)cand) - 1u; rb_need &= ~((2ull << bit) - 1ull); }
        else {
          rb_need = 0;
          if (xcur < xend) bit = xcur++;
        }
        if (bit != NOSLOT) {
          const u32 k = bit >> 2, q = bit & 3u;
          const bool noted = bal && k < 16u;
          const u32 m0 = noted ? sum[rown + k] & 15u : 0u, si = noted ? (u32)sidx[rown + k] : 0xFFu;
          const bool st = ((m0 >> q) & 1u) && si != 0xFFu;
          uint4 r;
          uint4* ptr;
          if (st) {
            const u32 j = si + (u32)__popc(m0 & ((1u << q) - 1u));
            r = st_r[j];
            ptr = (uint4*)(__attribute__((address_space(1))) uint4*)st_p[j];
          } else reload(k, q, r, ptr);
          TCNT(15, __popcll(__ballot(!st)));  // records fetched again (not stashed)
          if (SIM_META_KIND(r.y) != SIM_K_EMPTY) run(r, ptr);
        }
      }
      TT(5);
    } else if (!SHARDED || !tp.first) {  // (not compiled into the RF one-page instantiation: it has ONE site that calls the handlers)
      // the pages of the f packets, in order: packet k's page 0, 1, ... then packet k + 1 (one page each unless MP)
      u32 k = 0, pg = 0, wnp = wave_np(0);
      const u32 npk = RF ? rf_npk : d.f;  // packets to walk
#ifdef TICK_NEXT_SLOTS
      // The slot-map lookups of the NEXT page travel while this page is classified: they are issued as soon as the next
      // cell's words are here (right behind the wait for this page's heads), and parked in the top 16 bits of this lane's
      // staged entry pointers (48-bit addresses) so that they cost no register across the handler loop.  (16-bit slots:
      // not with a dense view of more than 65 534 subjects.)
      const bool use_ns = d.A <= 65534u;
      bool have_ns = false;  // (wave-uniform) lds_p[i][tid] >> 48 = slot of record i of the page whose words are in rn ..
#endif
      while (k < npk) {
        if (RF) {
          // the page in rn .. : a first page came from its sender's cell 0 on spec — now that the map word is here: nothing
          // sent (or lost) -> an empty page; the packet is another of the sender's cells (1 % of them) -> fetched now
          if (!MP || pg == 0) {
            rf_jb = rf_e == NOSLOT ? 0xFFu : (rf_map >> (8u * (rf_e & 3u))) & 0xFFu;
            rf_snd = rf_e >> 2;
            if (rf_jb == 0xFFu) rn = rn1 = rn2 = zero;
            const bool far = rf_jb != 0xFFu && (rf_jb >> 2) != 0u;
            if (__any(far)) {
              const uint4* c2 = far ? d.rfrd + ((size_t)(rf_jb >> 2) * d.NC + rf_snd) * RF_CELL_U4 : d.nullcell;
              const uint4 a0 = ld4(c2), a1 = ld4(c2 + 1), a2 = ld4(c2 + 2);
              if (far) { rn = a0; rn1 = a1; rn2 = a2; }
            }
          }
        }
        if (MP) {
          if (++pg >= wnp) { ++k; pg = 0; if (k < npk) wnp = wave_np(k); }
        } else ++k;
        if (RF && (!MP || pg == 0)) {  // on to the next packet: its entry was asked for an iteration ago; ask for the one after it
          rf_e = rf_en;
          rf_en = k + 1u < rcnt ? d.rsrc[rin0 + k + 1u] : NOSLOT;
        }
        // from here on (k, pg) is the page AFTER the one being delivered (whose three words are in rn, rn1, rn2)
        u32 slow;  // records of this page that need a handler
        // ---- stage the packet in LDS (one 16-byte column per record and lane: conflict-free) ----
        // phase A: the four records, then their four independent lookups — slot map for member
        // records, then the 16-byte head of the view entry / ring bucket each record is checked
        // against.  Everything lands in this lane's LDS cells so that the handler loop below can
        // index it by record number without holding 40 registers across the handlers.
        {
          const uint4 ck = rn, cl = rn1, ch = rn2;
          // The next packet is requested right BEHIND this packet's slot-map loads (which need only the keys and the
          // kinds, straight from the wire words), not before them: vmcnt counts in issue order, so a prefetch issued first
          // is waited for by the slot-map wait — an HBM round trip where an L2 one would do, and the heads' round trip
          // on top; issued behind the slot-map loads it travels together with the head loads.
          // (its address is worked out up here: what that needs may come back from scratch, and a scratch reload
          // between two loads makes the second wait for the first)
          cell = k < npk ? cell_of(k, pg) : d.nullcell;
          auto prefetch = [&]() __attribute__((always_inline)) {
            rn = ld4(cell); rn1 = ld4(cell + 1); rn2 = ld4(cell + 2);
            if (RF && (!MP || pg == 0)) rf_map = reinterpret_cast<const u32*>(cell)[12];
          };
          TT(1);
          // wave-ballot early out: nobody in this wave received anything in packet k (an empty record is all zero)
#ifdef TICK_NEXT_SLOTS
          if (!__any(((ch.x | ch.y | ch.z | ch.w) & 0xF0u) != 0)) { prefetch(); have_ns = false; continue; }
#else
          if (!__any(((ch.x | ch.y | ch.z | ch.w) & 0xF0u) != 0)) { prefetch(); continue; }
#endif
          u32 k0 = SIM_META_KIND(ch.x), k1 = SIM_META_KIND(ch.y), k2 = SIM_META_KIND(ch.z), k3 = SIM_META_KIND(ch.w);
          if (ABL(64)) { n.dirty |= (k0 ^ k1 ^ k2 ^ k3) & tp.zero_; prefetch(); continue; }
          u32 s0, s1, s2, s3;
#ifdef TICK_NEXT_SLOTS
          if (have_ns) {
            const u64* lp = reinterpret_cast<const u64*>(&lds_p[0][0]);
            auto dec = [](u64 v) __attribute__((always_inline)) -> u32 { u32 x = (u32)(v >> 48); return x == 0xFFFFu ? NOSLOT : x; };
            s0 = dec(lp[0 * TBLOCK + tid]); s1 = dec(lp[1 * TBLOCK + tid]); s2 = dec(lp[2 * TBLOCK + tid]); s3 = dec(lp[3 * TBLOCK + tid]);
          } else
#endif
          {
            s0 = slot_load(d, k0, ck.x);
            s1 = slot_load(d, k1, ck.y);
            s2 = slot_load(d, k2, ck.z);
            s3 = slot_load(d, k3, ck.w);
          }
          prefetch();
          uint4 r0 = wire_unpack(ck.x, cl.x, ch.x), r1 = wire_unpack(ck.y, cl.y, ch.y);
          uint4 r2 = wire_unpack(ck.z, cl.z, ch.z), r3 = wire_unpack(ck.w, cl.w, ch.w);
          lds_r[0][tid] = r0; lds_r[1][tid] = r1; lds_r[2][tid] = r2; lds_r[3][tid] = r3;
          TT(2);
          uint4* p0 = lookup_ptr(c, vbase, eoff, qoff, k0, r0.x, (u64)r0.z | ((u64)r0.w << 32), s0);
          uint4* p1 = lookup_ptr(c, vbase, eoff, qoff, k1, r1.x, (u64)r1.z | ((u64)r1.w << 32), s1);
          uint4* p2 = lookup_ptr(c, vbase, eoff, qoff, k2, r2.x, (u64)r2.z | ((u64)r2.w << 32), s2);
          uint4* p3 = lookup_ptr(c, vbase, eoff, qoff, k3, r3.x, (u64)r3.z | ((u64)r3.w << 32), s3);
          lds_p[0][tid] = p0; lds_p[1][tid] = p1; lds_p[2][tid] = p2; lds_p[3][tid] = p3;
          uint4 e0 = ld4(p0 ? p0 : d.nullcell), e1 = ld4(p1 ? p1 : d.nullcell), e2 = ld4(p2 ? p2 : d.nullcell), e3 = ld4(p3 ? p3 : d.nullcell);
          TT(3);
          lds_e[0][tid] = e0; lds_e[1][tid] = e1; lds_e[2][tid] = e2; lds_e[3][tid] = e3;
#ifdef TICK_NEXT_SLOTS
          // (the store above waited for the heads AND for the next cell: its keys and kinds are in rn / rn2)
          u32 ns0 = NOSLOT, ns1 = NOSLOT, ns2 = NOSLOT, ns3 = NOSLOT;
          const bool park = use_ns && k < d.f;
          if (park) {
            ns0 = slot_load(d, SIM_META_KIND(rn2.x), rn.x); ns1 = slot_load(d, SIM_META_KIND(rn2.y), rn.y);
            ns2 = slot_load(d, SIM_META_KIND(rn2.z), rn.z); ns3 = slot_load(d, SIM_META_KIND(rn2.w), rn.w);
          }
#endif
          // classify all four against the state as it is now: straight-line, no state is touched
          slow = (fast_retire(c, n, k0, r0, p0 != nullptr, e0) ? 0u : 1u) | (fast_retire(c, n, k1, r1, p1 != nullptr, e1) ? 0u : 2u) |
                 (fast_retire(c, n, k2, r2, p2 != nullptr, e2) ? 0u : 4u) | (fast_retire(c, n, k3, r3, p3 != nullptr, e3) ? 0u : 8u);
#ifdef TICK_NEXT_SLOTS
          if (park) {
            u64* lp = reinterpret_cast<u64*>(&lds_p[0][0]);
            lp[0 * TBLOCK + tid] = (u64)(uintptr_t)p0 | ((u64)(ns0 & 0xFFFFu) << 48);
            lp[1 * TBLOCK + tid] = (u64)(uintptr_t)p1 | ((u64)(ns1 & 0xFFFFu) << 48);
            lp[2 * TBLOCK + tid] = (u64)(uintptr_t)p2 | ((u64)(ns2 & 0xFFFFu) << 48);
            lp[3 * TBLOCK + tid] = (u64)(uintptr_t)p3 | ((u64)(ns3 & 0xFFFFu) << 48);
          }
          have_ns = park;
#endif
          TT(4);
        }
        // phase B: the records that need a handler, in arrival order, one rolled loop = one copy of the handler code.
        // Duplicates, old messages and subjects without a view slot (~95 % of all records) were retired by
        // fast_retire above: they change nothing, not even a clock, so it does not matter that they were judged
        // before the handlers of earlier records ran — EXCEPT when such a handler writes the entry a later
        // record was judged against (a pruned member, a first alive before a suspect ...): that record is looked at
        // again.  Every lane walks its own list, so a wave runs the handlers as many times as its busiest lane has
        // work (once or twice per packet), not once per record position.
        // `wptr`: the one entry a handler of this packet has written so far; `wall`: more than one,
        // or the node's own entry as well (refutation) — only then is a staged head stale.
        if (ABL(32)) continue;
        uint4* wptr = nullptr;
        bool wall = false;
        TCNT(13, 1);                                          // pages delivered with at least one record in the wave
        TCNT(14, __popcll(__ballot(slow != 0)));              // lanes with a record that needs a handler
        TCNT(15, __any(slow != 0) ? 1 : 0);                   // pages whose handler loop ran at all
#pragma unroll 1
        while (__any(slow != 0)) {
          TCNT(12, 1);                                        // iterations of the handler loop
          if (slow) {
            u32 p = (u32)__ffs((int)slow) - 1u;
            slow &= slow - 1u;
            uint4 r = lds_r[p][tid];
#ifdef TICK_NEXT_SLOTS
            uint4* ptr = (uint4*)(__attribute__((address_space(1))) uint4*)((uintptr_t)lds_p[p][tid] & 0x0000FFFFFFFFFFFFull);
#else
            uint4* ptr = (uint4*)(__attribute__((address_space(1))) uint4*)lds_p[p][tid];  // (global, not flat, accesses in the handlers)
#endif
            uint4 e = lds_e[p][tid];
            if (ptr && (wall || ptr == wptr)) e = ld4(ptr);
            Ins ins;
            ins.has = ins.wide = 0;
            bool dirty = false;
            dispatch(c, n, r, ptr, e, dirty, ins);
            if (dirty) {
#pragma unroll 1
              for (u32 q = p + 1; q < SIM_P; ++q)
#ifdef TICK_NEXT_SLOTS
                if (ins.wide || (ptr && (uint4*)((uintptr_t)lds_p[q][tid] & 0x0000FFFFFFFFFFFFull) == ptr)) slow |= 1u << q;
#else
                if (ins.wide || (ptr && lds_p[q][tid] == ptr)) slow |= 1u << q;
#endif
              wall |= ins.wide || (wptr != nullptr && wptr != ptr);
              wptr = ptr;
            }
            if (ins.has) pend_push(c, n, ins);
          }
        }
        TT(5);
      }
    }
    if (d.swim) {
      // suspicion timers (4 slots), then the probe: five producers, one queue_broadcast site
      bool due = n.susp_next && (u32)tp.tick >= n.susp_next;
      // probe phase is shared by the 64 nodes of an id-aligned group: wave-uniform when shard0 % 64 == 0
      bool probing = d.N >= 2 && ((u32)tp.tick + (gid >> 6)) % d.PI == 0;
      if (__any(due || probing)) {
        u32 next = 0;
        // the timer list, read once (one round trip instead of one per entry).  Nothing in the walk changes another entry than
        // the one it is looking at: a timer that fires forgets its own slot (swim_dead -> susp_forget), nothing starts one
        uint4 ta = make_uint4(0, 0, 0, 0), tb = ta;
        if (due) susp_load(c, ta, tb);
#pragma unroll 1
        for (u32 j = 0; j <= SIM_S; ++j) {
          Ins ins;
          ins.has = ins.wide = 0;
          if (j < SIM_S) { if (due) swim_timer_j(c, n, j, susp_get(ta, tb, j), next, ins); }
          else {
            if (due) { n.susp_next = next; n.dirty |= DR3; }
            if (probing) swim_probe(c, n, tp, base, ins);
          }
          if (ins.has) pend_push(c, n, ins);
        }
      }
    }
  }
  if (up && d.reap_interval) {  // Reaper: wave-uniform phase, per-lane due check
    bool due = ((u32)tp.tick + (gid >> 6)) % d.reap_interval == 0 && n.reap_next && (u32)tp.tick >= n.reap_next;
    if (due) reap_run(c, n, tp.n_slots);
  }
  if (up && d.reconnect_interval && ((u32)tp.tick + (gid >> 6)) % d.reconnect_interval == 0 && n.nfailed) reconnect_run(c, n, tp, tp.n_slots);
  TT(6);
  if (ABL(1)) { if (up) node_store(d, l, n); return; }
  // ---- phase 2: queue.  Load the sort keys, queue what phase 1 parked, drain `fanout` packets.
  u32 sk[SIM_Q];
  u32 cnt0 = __popc(n.used);
  if (up) {
    // the first two parked broadcasts travel with the sort keys (one round trip instead of up to three; a lane with
    // nothing parked reads the zero cell: a load inside a branch gets its own s_waitcnt)
    uint4 pq0 = ld4(n.npend > 0 ? &d.pend[l] : d.nullcell), pq1 = ld4(n.npend > 1 ? &d.pend[(size_t)d.Nl + l] : d.nullcell);
    keys_load(d, l, cnt0, sk);
    if (n.next_seq > 1023u - 64u) q_renorm(n, sk);
    if (__any(n.npend > 0)) {
      if (n.npend > 0) q_insert(c, n, sk, pq0.x, pq0.y, (u64)pq0.z | ((u64)pq0.w << 32));
      if (__any(n.npend > 1)) {
        if (n.npend > 1) q_insert(c, n, sk, pq1.x, pq1.y, (u64)pq1.z | ((u64)pq1.w << 32));
#pragma unroll 1
        for (u32 i = 2; i < n.npend; ++i) {
          uint4 q = ld4(&d.pend[(size_t)i * d.Nl + l]);
          q_insert(c, n, sk, q.x, q.y, (u64)q.z | ((u64)q.w << 32));
        }
      }
    }
    if (d.queue_check_interval && ((u32)tp.tick + (gid >> 6)) % d.queue_check_interval == 0) queue_check(c, n, sk);
  }
  TT(7);
  u32 limit = up ? d.retransmit_mult * digits10(n.nknown) : 0;
  // gossip_to_the_dead_time (App. B.2): the slots whose target this node believed dead for too long when the tick began
  // (gossip_skip_kernel, launched ahead of the tick only when the option is on) — those packets are not sent
#ifndef TICK_LEAN
  u32 skipm = d.gttd ? (u32)d.skipmask[l] : 0u;
  if (RF && d.N < 256u) {
    // a slot that drew no target sends nothing (oracle tick_node).  Only a cluster with fewer other nodes than the fan-out, or
    // a very small one out of luck, has such slots: with 256 nodes or more, 3 N tries that fail to find `fanout` distinct
    // peers have a probability below 1e-1000 and are not looked for
    u32 ch[SIM_MAX_FANOUT];
    const u32 nc = rf_draw(tp.rfan_base, gid, d.N, tp.feff, ch);
    skipm |= (0xFu << nc) & 0xFu;
  }
#else
  const u32 skipm = 0u;  // (measurement build: the round's optional memberlist switches compiled out of the tick kernel)
#endif
  const bool coop = (bx + 1u) * TBLOCK <= cnt;  // every lane of the block is here
  // One 48-byte cell per lane that has one (`wr`), written quad-cooperatively when the whole wave is here:
  // three lanes of a quad write one whole cell per store instruction (lane i < 3 writes part i of quad-mate
  // j's packet): the texture addresser sees 48 contiguous bytes per quad and L2 one write per cell instead of
  // three.  The transpose goes through this wave's columns of lds_r (free in phase 2), XOR-swizzled so that
  // neither side has bank conflicts.  A lane without a cell to write hands its quad a null address.
  // (RF: the cells are 64 bytes and every store writes all four quarters — one whole, aligned burst —, the fourth being the
  // node's map word `mapw`: it is cell 0's that the receivers read, the copies in the other cells are never looked at)
  auto store_cell = [&](uint4* dst, bool wr, const uint4& wk, const uint4& wl, const uint4& wh, u32 mapw) __attribute__((always_inline)) {
    if (coop) {
      lds_r[0][tid] = wk; lds_r[1][tid ^ 1] = wl; lds_r[2][tid ^ 2] = wh;
      __builtin_amdgcn_wave_barrier();
      u32 qi = tid & 3u, qb = tid & ~3u, part = qi < 3u ? qi : 2u;  // the fourth lane of a quad has nothing to write
      u32 dlo = wr ? (u32)(uintptr_t)dst : 0u, dhi = wr ? (u32)((uintptr_t)dst >> 32) : 0u;
#define COOP_STORE(j)                                                                              \
      {                                                                                            \
        uint4 v = lds_r[part][(qb + j) ^ part];                                                    \
        u32 lo = (u32)__builtin_amdgcn_mov_dpp((int)dlo, j * 0x55, 0xF, 0xF, true);                \
        u32 hi = (u32)__builtin_amdgcn_mov_dpp((int)dhi, j * 0x55, 0xF, 0xF, true);                \
        if (RF) { u32 mw = (u32)__builtin_amdgcn_mov_dpp((int)mapw, j * 0x55, 0xF, 0xF, true); if (qi == 3u) v = make_uint4(mw, 0u, 0u, 0u); } \
        if ((RF || qi < 3u) && (lo | hi) != 0u) ((uint4*)(((uintptr_t)hi << 32) | lo))[qi] = v;   \
      }
      COOP_STORE(0) COOP_STORE(1) COOP_STORE(2) COOP_STORE(3)
#undef COOP_STORE
      __builtin_amdgcn_wave_barrier();
    } else if (wr) {
      dst[0] = wk; dst[1] = wl; dst[2] = wh;
      if (RF) dst[3] = make_uint4(mapw, 0u, 0u, 0u);
    }
  };
  if (MP) {
    // ---- packets of up to d.P records (pages of SIM_P): drain, then every DISTINCT packet page by page ----
    u64 nib[F];
    u32 cn[F];
#pragma unroll
    for (int k = 0; k < F; ++k) {
      nib[k] = 0; cn[k] = 0;
      if (up && (u32)k < tp.feff) {
        u64 nb; u32 c;
        q_round_mp(n, sk, limit, d.P, nb, c);
        bool lost = (tp.loss_u32 && (u32)(mix64(tp.loss_base ^ ((u64)gid * 4u + k)) >> 32) < tp.loss_u32) || ((skipm >> k) & 1u);
        if (!lost) { nib[k] = nb; cn[k] = c; }
      }
    }
    u32 bb0 = 0, s0 = 0, r0 = ll;
    if (SHARDED && (tp.V != 1 || tp.C != 1)) {
      bb0 = ll / tp.blk;
      u32 w = ll - bb0 * tp.blk;
      s0 = w / tp.sub;
      r0 = w - s0 * tp.sub;
    }
    const u32 uu = bb0 * tp.sub + r0;
    u32 fj = uu, fi = 0;
    if (SHARDED && tp.B == 64u) { fj = (u32)__builtin_amdgcn_readfirstlane((int)(uu >> 6)); fi = uu & 63u; }
    const u32 pj = (SHARDED && tp.feff) ? pi_f(tp, fj) : 0;
    u32 jout = 0xFFFFFFFFu, used = 0;  // local mode: the map word, pages handed out so far
    // RF: the map word travels IN cell 0, which is written first: the word is worked out before anything is stored (the same
    // arithmetic as in the loop below, on registers)
    u32 jfin = 0xFFFFFFFFu;
    if (RF) {
      u32 usedp = 0;
#pragma unroll
      for (int k = 0; k < F; ++k) {
        if ((u32)k >= tp.feff || cn[k] == 0u) continue;
        u32 jenc = 0xFFu;
        bool nw = true;
#pragma unroll
        for (int q = k - 1; q >= 0; --q)
          if (cn[q] == cn[k] && nib[q] == nib[k]) { jenc = (jfin >> (8 * q)) & 0xFFu; nw = false; }
        if (nw) { const u32 npk_ = (cn[k] + SIM_P - 1u) / SIM_P; jenc = (usedp << 2) | (npk_ - 1u); usedp += npk_; }
        jfin = (jfin & ~(0xFFu << (8 * k))) | (jenc << (8 * k));
      }
    }
#pragma unroll
    for (int k = 0; k < F; ++k) {
      if ((u32)k >= tp.feff) break;
      const bool has = cn[k] != 0u;  // (nothing queued, or the packet was lost: no cell)
      const u32 np = (cn[k] + SIM_P - 1u) / SIM_P;
      u32 first = 0;
      bool isnew = SHARDED;
      uint4* dbase;  // page 0 of this packet's cells; page pg is pstride uint4s further on
      size_t pstride;
      if (SHARDED) {
        u32 y = pj + tp.off[k];
        if (y >= tp.nbc) y -= tp.nbc;
        u32 j2 = pi_inv(tp, y);
        u32 u2 = j2;
        if (tp.B == 64u) u2 = j2 * 64u + (fi ^ fan_scramble(y, (u32)k));
        u32 bb = 0, r = u2, h = 0;
        if (tp.V != 1 || tp.C != 1) {
          bb = u2 / tp.sub;
          r = u2 - bb * tp.sub;
          h = (g + tp.V - ((bb + tp.rot[k]) % tp.V)) % tp.V;
        }
        dbase = d.xsend + ((((size_t)s0 * tp.V + h) * d.fp + (size_t)k * d.PG) * tp.sub + r) * PK_U4;
        pstride = (size_t)tp.sub * PK_U4;
      } else {
        u32 jenc = 0xFFu;
        if (has) {
          isnew = true;
#pragma unroll
          for (int q = k - 1; q >= 0; --q)
            if (cn[q] == cn[k] && nib[q] == nib[k]) { jenc = (jout >> (8 * q)) & 0xFFu; isnew = false; }
          if (isnew) { first = used; used += np; jenc = (first << 2) | (np - 1u); }
        }
        jout = (jout & ~(0xFFu << (8 * k))) | (jenc << (8 * k));
        dbase = d.obox[cur ^ 1] + ((size_t)first * d.Nl + l) * (RF ? RF_CELL_U4 : PK_U4);
        pstride = (size_t)d.Nl * (RF ? RF_CELL_U4 : PK_U4);
      }
      u32 wmax = SHARDED ? d.PG : 0u;  // pages the wave writes for this slot (uniform)
      if (!SHARDED) {
        if (__any(isnew)) wmax = 1u;
        if (__any(isnew && np >= 2u)) wmax = 2u;
        if (__any(isnew && np >= 3u)) wmax = 3u;
        if (__any(isnew && np >= 4u)) wmax = 4u;
      }
#pragma unroll 1
      for (u32 pgi = 0; pgi < wmax; ++pgi) {
        const bool wr = SHARDED || (isnew && pgi < np);
        uint4 pk[SIM_P];
#pragma unroll
        for (int p = 0; p < (int)SIM_P; ++p) {
          u32 idx = 4u * pgi + (u32)p;
          bool valid = has && isnew && idx < cn[k];
          u32 sl = (u32)(nib[k] >> (4u * (idx & 15u))) & 15u;
          pk[p] = ld4(valid ? &d.qpay[(size_t)sl * d.Nl + l] : d.nullcell);
        }
        uint4 wk, wl, wh;
        wire_pack(pk[0], wk.x, wl.x, wh.x); wire_pack(pk[1], wk.y, wl.y, wh.y);
        wire_pack(pk[2], wk.z, wl.z, wh.z); wire_pack(pk[3], wk.w, wl.w, wh.w);
        store_cell(dbase + (size_t)pgi * pstride, wr, wk, wl, wh, jfin);
      }
    }
    if (!SHARDED && !RF) d.omap[cur ^ 1][l] = jout;
    // RF: a node that sent nothing has not written its cell 0 — its map word (all 0xFF) has to stand there all the same
    if (RF && jfin == 0xFFFFFFFFu) d.obox[cur ^ 1][(size_t)l * RF_CELL_U4 + 3u] = make_uint4(0xFFFFFFFFu, 0u, 0u, 0u);
  } else {
  // all F drain rounds first (pure register work on the sort keys) ...
  u32 slots[F];
#ifdef TICK_NO_ROUND8
  const bool small = false;
#else
  const bool small = !__any(up && sk[8] != KEMPTY);  // (wave-uniform) nobody holds more than 8 entries
#endif
#pragma unroll
  for (int k = 0; k < F; ++k) {
    slots[k] = 0xFFFFFFFFu;
    if (up && (u32)k < tp.feff) {
      u32 s = small ? q_round8(n, sk, limit) : q_round(n, sk, limit);
      bool lost = (tp.loss_u32 && (u32)(mix64(tp.loss_base ^ ((u64)gid * 4u + k)) >> 32) < tp.loss_u32) || ((skipm >> k) & 1u);
      if (!lost) slots[k] = s;
    }
  }
  TT(8);
  // ... then the payload of the first packet (four gathers in flight; a lane with nothing to fetch reads the zero cell:
  // a load inside a branch gets its own s_waitcnt, i.e. its own round trip).  A queue of at most SIM_P entries sends
  // the same records in every round, so packet k + 1 is packet k except where another payload slot moved into a
  // position — those are gathered while packet k is being stored, behind a wave-uniform branch that is rarely taken.
  // (Holding all F packets in registers at once — 64 VGPRs — made the compiler spill every gather as it arrived:
  // sixteen round trips, one after the other.)
  uint4 pkc[SIM_P];
#pragma unroll
  for (int p = 0; p < (int)SIM_P; ++p) {
    u32 s = (slots[0] >> (8 * p)) & 0xFFu;
    pkc[p] = ld4((s != 0xFFu && !ABL(8)) ? &d.qpay[(size_t)s * d.Nl + l] : d.nullcell);
  }
  TT(9);
  // ... then the F scatters: packet k goes to the inbox cell of T_k(l) (SIMSPEC §2.3, oracle fan_target).  The sender is
  // (vblock bb0, sub-slab s0 = its chunk, offset r0); u = its index inside the chunk = (block j, position i).  With
  // 64-node blocks j is the same for the whole wave: the block permutation runs on the scalar unit, and the wave's 64
  // packets of one slot land in 64 consecutive cells.
  u32 bb0 = 0, s0 = 0, r0 = ll;
  if (SHARDED && (tp.V != 1 || tp.C != 1)) {
    bb0 = ll / tp.blk;
    u32 w = ll - bb0 * tp.blk;
    s0 = w / tp.sub;
    r0 = w - s0 * tp.sub;
  }
  const u32 uu = bb0 * tp.sub + r0;
  u32 fj = uu, fi = 0;
  if (SHARDED && tp.B == 64u) { fj = (u32)__builtin_amdgcn_readfirstlane((int)(uu >> 6)); fi = uu & 63u; }
  const u32 pj = (SHARDED && tp.feff) ? pi_f(tp, fj) : 0;
  // local mode: which of this node's cells holds the packet of every slot (0xFF: nothing sent), distinct packets so far
  u32 jout = 0xFFFFFFFFu, ndist = 0;
  // RF: the map word travels IN cell 0, which is written first: the word is worked out before anything is stored
  u32 jfin = 0xFFFFFFFFu;
  if (RF) {
    u32 nd = 0;
#pragma unroll
    for (int k = 0; k < F; ++k) {
      if ((u32)k >= tp.feff || slots[k] == 0xFFFFFFFFu) continue;
      u32 j = 0xFFu;
#pragma unroll
      for (int q = k - 1; q >= 0; --q)
        if (slots[q] == slots[k]) j = ((jfin >> (8 * q)) & 0xFFu) >> 2;
      if (j == 0xFFu) j = nd++;
      jfin = (jfin & ~(0xFFu << (8 * k))) | ((j << 2) << (8 * k));
    }
  }
#pragma unroll
  for (int k = 0; k < F; ++k) {
    if ((u32)k >= tp.feff || ABL(4)) break;
    // the payloads packet k + 1 does not share with packet k, in flight while packet k goes out
    uint4 pkn[SIM_P];
    bool fetch = false;
    if (k + 1 < F) {
#pragma unroll
      for (int p = 0; p < (int)SIM_P; ++p) {
        u32 s = (slots[k + 1] >> (8 * p)) & 0xFFu;
        fetch |= s != 0xFFu && s != ((slots[k] >> (8 * p)) & 0xFFu);
      }
      fetch = __any(fetch && !ABL(8));
      if (fetch) {
#pragma unroll
        for (int p = 0; p < (int)SIM_P; ++p) {
          u32 s = (slots[k + 1] >> (8 * p)) & 0xFFu;
          bool again = s == ((slots[k] >> (8 * p)) & 0xFFu);
          pkn[p] = ld4((s != 0xFFu && !again) ? &d.qpay[(size_t)s * d.Nl + l] : d.nullcell);
        }
      }
    }
    uint4* dst;
    bool wr = true;  // this lane has a cell to write for slot k
    if (SHARDED) {
      u32 y = pj + tp.off[k];
      if (y >= tp.nbc) y -= tp.nbc;
      u32 j2 = pi_inv(tp, y);
      u32 u2 = j2;
      if (tp.B == 64u) u2 = j2 * 64u + (fi ^ fan_scramble(y, (u32)k));
      u32 bb = 0, r = u2, h = 0;
      if (tp.V != 1 || tp.C != 1) {
        bb = u2 / tp.sub;
        r = u2 - bb * tp.sub;
        h = (g + tp.V - ((bb + tp.rot[k]) % tp.V)) % tp.V;
      }
      dst = d.xsend + ((((size_t)s0 * tp.V + h) * d.f + k) * tp.sub + r) * PK_U4;
    } else {
      // the same payload slots in the same positions = the same packet (payloads do not change while the queue
      // drains; transmit counts do not travel): point at the cell that already holds it
      u32 j = 0xFFu;
      wr = slots[k] != 0xFFFFFFFFu;
      if (wr) {
#pragma unroll
        for (int q = k - 1; q >= 0; --q)
          if (slots[q] == slots[k]) { j = ((jout >> (8 * q)) & 0xFFu) >> 2; wr = false; }
        if (wr) j = ndist++;
      }
      jout = (jout & ~(0xFFu << (8 * k))) | ((j == 0xFFu ? 0xFFu : j << 2) << (8 * k));  // first page << 2 | pages - 1
      dst = d.obox[cur ^ 1] + ((size_t)j * d.Nl + l) * (RF ? RF_CELL_U4 : PK_U4);
    }
    const bool store = SHARDED || __any(wr);  // (wave-uniform)
    uint4 wk, wl, wh;  // the packet in its wire form
    if (store) {
      wire_pack(pkc[0], wk.x, wl.x, wh.x); wire_pack(pkc[1], wk.y, wl.y, wh.y);
      wire_pack(pkc[2], wk.z, wl.z, wh.z); wire_pack(pkc[3], wk.w, wl.w, wh.w);
    }
    if (k + 1 < F) {  // packet k + 1: what stays in place is kept, what moved in was fetched, an empty position is zero
#pragma unroll
      for (int p = 0; p < (int)SIM_P; ++p) {
        u32 s = (slots[k + 1] >> (8 * p)) & 0xFFu;
        bool again = s == ((slots[k] >> (8 * p)) & 0xFFu);
        if (fetch) { if (!again) pkc[p] = pkn[p]; }
        else if (s == 0xFFu) pkc[p] = zero;
      }
    }
    if (!store) continue;
    store_cell(dst, wr, wk, wl, wh, jfin);
  }
  if (!SHARDED && !RF && !ABL(4)) d.omap[cur ^ 1][l] = jout;
  // RF: a node that sent nothing has not written its cell 0 — its map word (all 0xFF) has to stand there all the same
  if (RF && jfin == 0xFFFFFFFFu && !ABL(4)) d.obox[cur ^ 1][(size_t)l * RF_CELL_U4 + 3u] = make_uint4(0xFFFFFFFFu, 0u, 0u, 0u);
  }
  TT(10);
  if (up && !ABL(16)) {
    node_store(d, l, n);
    keys_store(d, l, cnt0, n.used, sk);
  }
  TT(11);
#ifdef TICK_TIMING
  if ((threadIdx.x & 63) == 0)
    for (int i = 0; i < 32; ++i) atomicAdd(&g_tt[i], tacc[i]);
#endif
}
#ifndef TICK_OCC_RF
#define TICK_OCC_RF 4
#endif
// the RF one-page instantiation under its own launch bounds (8 KiB of LDS per wave: 20 waves per CU if the registers allow 5 per SIMD)
template <int F>
__global__ __launch_bounds__(TBLOCK, TICK_OCC_RF) void tick_kernel_rf(Dev d, TickP tp, TickP ptp, u32 cur, const uint4* base, u32 chunk, u32 cnt) {
  tick_block<false, F, false, false, true>(d, tp, ptp, cur, base, chunk, cnt, blockIdx.x);
}
template <bool SHARDED, int F, bool B64, bool MP, bool RF = false>
__global__ __launch_bounds__(TBLOCK, TICK_OCC) void tick_kernel(Dev d, TickP tp, TickP ptp, u32 cur, const uint4* base, u32 chunk, u32 cnt) {
#ifdef TICK_PERSIST
  // experiment (withdrawn, profiles/r03_experiments.md): as many blocks as fit on the GPU at once, each walking its share of the node blocks
  const u32 nb = (cnt + TBLOCK - 1) / TBLOCK;
#pragma unroll 1
  for (u32 bx = blockIdx.x; bx < nb; bx += gridDim.x) tick_block<SHARDED, F, B64, MP, RF>(d, tp, ptp, cur, base, chunk, cnt, bx);
#else
  tick_block<SHARDED, F, B64, MP, RF>(d, tp, ptp, cur, base, chunk, cnt, blockIdx.x);
#endif
}

// ------------------------------------------------------------------------------------------------
// operations (user-facing API acting on one node): one thread, a handful of ops per launch
// ------------------------------------------------------------------------------------------------
struct OpBatch {
  u32 n;
  u32 op[8], node[8], a[8], b[8];
  u32 c[8];  // SIM_OP_QUERY: tracker index
  u64 val[8];  // SIM_OP_DELIVER: the record's value
};
__device__ static inline void up_set(const Dev& d, u32 gid, bool up) {
  u32 w = d.upmap[gid >> 5];
  d.upmap[gid >> 5] = up ? (w | (1u << (gid & 31))) : (w & ~(1u << (gid & 31)));
}
__global__ void ops_kernel(Dev d, OpBatch ob, u64 tick, u32 has_alive, u64 qbase, u32 q_timeout) {
  if (threadIdx.x || blockIdx.x) return;
  for (u32 i = 0; i < ob.n; ++i) {
    u32 gid = ob.node[i], op = ob.op[i];
    // ground-truth liveness is replicated on every shard (probes read it)
    if (op == SIM_OP_CRASH) up_set(d, gid, false);
    if (op == SIM_OP_REVIVE || op == SIM_OP_JOIN) up_set(d, gid, true);
    // base.rs:905-930: the QueryResponse is registered before the query goes out (every shard counts its own nodes)
    if (op == SIM_OP_QUERY) d.qtab[ob.c[i]] = make_uint4(ob.a[i], gid, (u32)tick + q_timeout, ob.b[i]);
    if (op == SIM_OP_SET_TAGS) TAGCLASS(d)[gid] = (uint8_t)ob.a[i];  // replicated like liveness: a table the host fills
    if (op == SIM_OP_QRESP) {  // handle_query_response (base.rs:1158-1204): an ack / a response that came in over the byte boundary.
      // Trackers and liveness are replicated; the responder's bit lives on the shard that owns the responder
      const u32 j = ob.a[i] % SIM_QT, from = ob.b[i] & 0xFFFFFFu, which = (ob.b[i] >> 31) ? 0u : 1u;
      const size_t words = ((size_t)d.N + 31) / 32;
      const uint4 t = d.qtab[j];
      if (from >= d.shard0 && from < d.shard0 + d.Nl && up_of(d, gid) && t.x == ob.a[i] && t.y == gid && (u32)tick <= t.z)
        d.qbits[((size_t)j * 2 + which) * words + (from >> 5)] |= 1u << (from & 31);
      continue;
    }
    if (gid < d.shard0 || gid >= d.shard0 + d.Nl) continue;
    u32 l = gid - d.shard0;
    Ctx c{d, l, gid, (u32)tick, qbase};
    Node n;
    node_load(d, l, n);
    u32 sk[SIM_Q];
    u32 cnt0 = __popc(n.used);
    keys_load(d, l, cnt0, sk);
    if (n.next_seq > 1023u - 64u) q_renorm(n, sk);
    n.dirty = DR0 | DR1 | DR2 | DR3;  // ops are rare: write the whole row back
    bool up = n.flags & SIM_RF_UP, dirty = false;
    u32 a = ob.a[i], b = ob.b[i];
    Ins ins, ins2;  // ins: what the handler queues; ins2: the op's own broadcast, queued after it
    ins.has = ins2.has = ins.wide = ins2.wide = 0;
    switch (op) {
      case SIM_OP_USER_EVENT:  // api.rs:241-299
        if (up) {
          u64 lt = n.eclock;
          n.eclock++;
          uint4* p = ering_ptr(c, lt);
          uint4 e_ = p[0];
          handle_user_event(c, n, a, lt, p, e_, dirty);
          ins_set(ins2, a, wire_meta(SIM_K_EVENT, (b >> 31) ? SIM_F_CC : 0u, b & 0x7FFFFFFFu), lt);
        }
        break;
      case SIM_OP_QUERY:  // base.rs:875-942
        if (up) {
          u64 lt = n.qclock;
          uint4* p = qring_ptr(c, lt);
          uint4 e_ = p[0];
          handle_query(c, n, a, lt, b, p, e_, dirty);
          ins_set(ins2, a, wire_meta(SIM_K_QUERY, b, 48), lt);
        }
        break;
      case SIM_OP_LEAVE:  // api.rs:422-460
        if (up && SIM_RF_STATE(n.flags) == SIM_SERF_ALIVE) {
          n.flags = (n.flags & ~(3u << 1)) | (SIM_SERF_LEAVING << 1);
          u64 lt = n.clock;
          n.clock++;
          uint4* p = view_ptr(c, gid);
          uint4 e_ = p ? p[0] : make_uint4(0, 0, 0, 0);
          handle_leave_intent(c, n, gid, lt, false, p, e_, dirty, ins);
          if (has_alive) ins_set(ins2, gid, wire_meta(SIM_K_LEAVE, 0, 16), lt);
        }
        break;
      case SIM_OP_LEAVE_FINISH:  // api.rs:474-497: memberlist.leave (dead{self, from = self}), state = Left
        if (up && SIM_RF_STATE(n.flags) == SIM_SERF_LEAVING) {
          if (d.swim) {
            uint4* p = view_ptr(c, gid);
            uint4 e_ = p ? p[0] : make_uint4(0, 0, 0, 0);
            swim_dead(c, n, gid, n.inc, gid, wire_meta(SIM_K_DEAD, 0, 32), p, e_, dirty, ins);
          }
          n.flags = (n.flags & ~(3u << 1)) | (SIM_SERF_LEFT << 1);
        }
        break;
      case SIM_OP_JOIN:  // api.rs:318-364
        n.flags |= SIM_RF_UP;
        n.flags = (n.flags & ~(3u << 1)) | (SIM_SERF_ALIVE << 1);
        if (d.swim) {
          uint4* p = view_ptr(c, gid);
          u32 old = SIM_SWIM_ALIVE, accused = n.inc;
          if (p) {
            uint4 e = p[0];
            old = SIM_VB_SWIM(e.w);
            accused = e.z;
            e.w = vb_set_nconf(vb_set_swim(e.w, SIM_SWIM_ALIVE), 0);
            p[0] = e;
          }
          swim_refute(c, n, accused, ins);
          aw_delta(n, -1);
          if (p && (old == SIM_SWIM_DEAD || old == SIM_SWIM_LEFT)) {
            uint4 e = p[0];
            node_join_e(c, n, e, gid);
            p[0] = e;
          }
        }
        broadcast_join(c, n, n.clock, dirty, ins2);
        break;
      case SIM_OP_FORCE_LEAVE:  // base.rs:452-480
        if (up) {
          u64 lt = n.clock;
          uint4* p = view_ptr(c, a);
          uint4 e_ = p ? p[0] : make_uint4(0, 0, 0, 0);
          handle_leave_intent(c, n, a, lt, b != 0, p, e_, dirty, ins);
          if (has_alive) ins_set(ins2, a, wire_meta(SIM_K_LEAVE, b ? SIM_F_PRUNE : 0, 16), lt);
        }
        break;
      case SIM_OP_SET_TAGS:  // api.rs:219-235: memberlist.update_node = next incarnation + an alive broadcast
        if (up && d.swim) {
          swim_refute(c, n, n.inc, ins);
          ins.wmeta |= SIM_F_META;  // the meta differs from the one the previous incarnation carried
          aw_delta(n, -1);  // not an accusation
        }
        break;
      case SIM_OP_CRASH: n.flags &= ~SIM_RF_UP; break;
      case SIM_OP_REVIVE: n.flags |= SIM_RF_UP; break;
      case SIM_OP_SUSPECT:  // the suspicion of a probe that failed last tick on a then slot-less target (swim_probe)
        if (up && d.swim) {
          uint4* p = view_ptr(c, a);
          if (p) {
            uint4 e_ = p[0];
            swim_suspect(c, n, a, e_.z, gid, wire_meta(SIM_K_SUSPECT, 0, 32), p, e_, dirty, ins);
          }
        }
        break;
      case SIM_OP_DELIVER:  // a record from outside the cluster: notify_message (delegate.rs:157-315) / memberlist's own handling
        if (up) {
          u64 val = ob.val[i];
          const bool mute = b & SIM_DELIVER_MUTE;
          b &= SIM_META_WIRE_MASK;
          uint4 r = make_uint4(a, b, (u32)val, (u32)(val >> 32));
          u32 kind = SIM_META_KIND(b);
          u64 vbase = (u64)(uintptr_t)d.view;
          uint4* p = lookup_ptr(c, vbase, (u64)(uintptr_t)d.ering - vbase, (u64)(uintptr_t)d.qring - vbase, kind, a, val, slot_load(d, kind, a));
          uint4 e_ = p ? p[0] : make_uint4(0, 0, 0, 0);
          if (mute) {  // out of a PushPull message: merge_remote_state (delegate.rs:495-552) — the handlers' verdicts are dropped, a
                       // refutation (broadcast_join: `ins`, set inside the handler) is not
            if (kind == SIM_K_LEAVE) (void)handle_leave_intent(c, n, a, val, false, p, e_, dirty, ins);
            else if (kind == SIM_K_JOIN) (void)handle_join_intent(c, n, a, val, p, e_, dirty);
            else if (kind == SIM_K_EVENT) (void)handle_user_event(c, n, a, val, p, e_, dirty);
          } else dispatch(c, n, r, p, e_, dirty, ins);
        }
        break;
      case SIM_OP_WITNESS:  // a PushPull message's clocks (delegate.rs:466-480)
        if (up) {
          if (a == 0) witness(n, n.clock, ob.val[i], DR0);
          else if (a == 1) witness(n, n.eclock, ob.val[i], DR1);
          else witness(n, n.qclock, ob.val[i], DR1);
        }
        break;
      default: break;
    }
    if (ins.has) q_insert(c, n, sk, ins.key, ins.wmeta, ins.val);
    if (ins2.has) q_insert(c, n, sk, ins2.key, ins2.wmeta, ins2.val);
    node_store(d, l, n);
    keys_store(d, l, cnt0, n.used, sk);
    __threadfence();  // the next op of this batch may touch the same node
  }
}

// ------------------------------------------------------------------------------------------------
// SIM_CF_RANDOM_FANOUT — memberlist's kRandomNodes (App. B.2; oracle rf_draw / rf_group): the tick's fan-out graph
// ------------------------------------------------------------------------------------------------
// Every node draws its `fanout` gossip targets uniformly over the other nodes, without replacement: a function of (seed,
// tick, node) alone, so the graph of tick t can be built before — or while — anything else of tick t runs, and any kernel
// that needs a target draws it again instead of reading it (a handful of mix64 per node against 16 bytes of HBM).
// What the tick kernel needs of it: for every receiver the row rsrc[rcsr[t] .. rcsr[t + 1]) of the (sender, slot) pairs
// p = 4 l + k that drew it, p ascending: the oracle hands a node its packets in (sender, slot) order.  That is a sort of
// f * N pairs by (target, p); the keys are uniform, so it is done as a two-level bucket sort written for this job (no
// library on the per-tick path), two launches per tick:
//   rf_scatter  every workgroup draws the targets of its SPW senders, counts them per level-1 bucket (= 2^LB consecutive
//               targets) in LDS, reserves its share of every bucket's REGION with one global atomic per (workgroup, bucket),
//               lays the pairs down in LDS bucket by bucket and writes them out from there: l1[b * bcap + ...], runs of
//               ~ 4 SPW / NB pairs stored by consecutive lanes.  The regions have a
//               fixed capacity bcap (mean 4 * 2^LB + 12 sigma: uniform draws never fill one); what does not fit all the
//               same goes onto an overflow list.  Any order inside a bucket will do: the order that counts is restored by
//               rank, not by stability — so the result does not depend on who won which atomic.
//   rf_rows     one workgroup per bucket: its place in the output = the sum of the totals of the buckets before it; an LDS
//               counting sort by target -> the bucket's part of rcsr; every pair ranks itself among the ~ f pairs of its row
//               (p ascending) -> its place in rsrc, laid down in LDS and written out as one run.  Pairs and targets stay in
//               registers between the passes.  A bucket that does not fit the LDS tables (never with uniform draws; forced
//               by the tests through SERF_RF_CAP) ranks straight from global memory.
struct RfP {
  u64 rb;       // rng_base(seed, STREAM_RFAN, tick)
  u32 N, Nl, shard0, feff, f;
  u32 Ns;       // senders whose targets are drawn: the handle's own Nl (global id shard0 + l; pair ids p = 4 * l + slot are local)
  u32 rcap;     // entries rsrc holds
  u32 LB, NB;   // level-1 buckets: ranges of 2^LB consecutive targets — NB = ceil(Nl / 2^LB) of them; on a shard (r5: the sort is
                // the SENDING side's, over the targets of the shard's own senders anywhere in the cluster) NBh = ceil(M / 2^LB) per
                // destination shard, NB = V * NBh: a bucket never straddles two shards
  u32 V, M, NBh;  // destination shards, their size, buckets per destination (one handle that holds every node: 1, N, NB)
  u32 PB;       // bits of a pair id p = 4 l + k; a scattered entry is (target - bucket start) << PB | p — 32 bits when they fit
  u32 NWG;      // workgroups of rf_scatter (RfSpw senders each)
  u32 cap;      // pairs rf_rows can rank in LDS (a multiple of RFR, at most RF_EPT * RFR)
  u32 bcap;     // pairs a bucket's region of l1 holds
  u32 ocap;     // entries of the overflow list
};
#define RFB 1024          // threads of an rf_scatter workgroup
#define RF_GCS 1u         // stride of the buckets' fill counters in words (a line each — 32 — made the atomics slower: 55.7 vs 47.6 us, profiles/r04_experiments.md)
// senders of an rf_scatter workgroup, by entry width: their pairs are staged in LDS (6 / 10 bytes a pair) and leave as runs of
// 4 SPW / NB entries — 4096 senders: 102 KiB of LDS with 32-bit entries; 64-bit entries (above 4 Mi nodes) take 2048
template <typename E> struct RfSpw { static constexpr u32 v = sizeof(E) == 4 ? 4096u : 2048u; };
#ifndef RFR
#define RFR 512           // threads of an rf_rows workgroup
#endif
#define RF_LB_MAX 13u     // at most 8192 rows per level-1 bucket (64 KiB of rf_rows' LDS; one handle that holds every node stays at 11 or below)
#define RF_EPT 24u        // pairs one thread of rf_rows keeps in registers: cap <= RF_EPT * RFR
// exclusive prefix over the 64 lanes of a wave (`total` = the sum)
__device__ static inline u32 wave_excl_scan(u32 v, u32& total) {
  u32 x = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    u32 y = (u32)__shfl_up((int)x, o, 64);
    if ((threadIdx.x & 63u) >= (u32)o) x += y;
  }
  total = (u32)__shfl((int)x, 63, 64);
  return x - v;
}
// gcur[NB]: pairs in each bucket so far (zero at launch: rf_rows of the build before zeroed it); ovf: [0] = entries, then (bucket, entry) pairs
// LDS of rf_scatter (dynamic): cnt[NB] | lst[NB] | cur[NB] | stage: E[f * SPW] | stb: u16[f * SPW]
template <typename E>
static inline size_t rf_scatter_lds(const RfP& r) { return (((size_t)3u * r.NB * 4u + 7u) & ~(size_t)7u) + (size_t)SIM_MAX_FANOUT * RfSpw<E>::v * (sizeof(E) + 2u); }
template <typename E>
__global__ __launch_bounds__(RFB) void rf_scatter_kernel(RfP r, u32* gcur, E* l1, E* ovf) {
  constexpr u32 SPW = RfSpw<E>::v;
  extern __shared__ u32 rf_lds[];
  // cnt: the workgroup's pairs per bucket, then its base in the bucket's region; lst: where the bucket's run starts in the
  // staging area; cur: cursors.  The pairs are laid down in LDS bucket by bucket and leave as runs: consecutive lanes store
  // consecutive entries of a region (scattered 4-byte stores straight from the drawing lanes cost 30 of this kernel's 48 us)
  u32 *cnt = rf_lds, *lst = cnt + r.NB, *cur = lst + r.NB;
  E* stage = reinterpret_cast<E*>(reinterpret_cast<char*>(rf_lds) + (((size_t)3u * r.NB * 4u + 7u) & ~(size_t)7u));
  uint16_t* stb = reinterpret_cast<uint16_t*>(stage + (size_t)SIM_MAX_FANOUT * SPW);
  __shared__ u32 wsum[RFB / 64u];
  for (u32 b = threadIdx.x; b < 3u * r.NB; b += RFB) rf_lds[b] = 0;
  __syncthreads();
  const u32 l0 = blockIdx.x * SPW;
  u32 tg[SPW / RFB][SIM_MAX_FANOUT], nc[SPW / RFB];
#pragma unroll
  for (u32 j = 0; j < SPW / RFB; ++j) {
    const u32 l = l0 + j * RFB + threadIdx.x;  // the sender, local index (global id shard0 + l)
    nc[j] = l < r.Ns ? rf_draw(r.rb, r.shard0 + l, r.N, r.feff, tg[j]) : 0u;
#pragma unroll
    for (u32 k = 0; k < SIM_MAX_FANOUT; ++k)
      if (k < nc[j]) {  // from here on a target is (bucket << 16 | offset in the bucket): the bucket never straddles two shards
        const u32 t = tg[j][k], h = r.V == 1u ? 0u : t / r.M, tl = t - h * r.M;
        const u32 b = h * r.NBh + (tl >> r.LB);
        tg[j][k] = (b << 16) | (tl & ((1u << r.LB) - 1u));
        atomicAdd(&cnt[b], 1u);
      }
  }
  __syncthreads();
  {  // exclusive prefix of the counts (every thread a stretch of buckets), and the workgroup's share of every region
    const u32 per = (r.NB + RFB - 1u) / RFB, b0 = threadIdx.x * per, b1 = min(b0 + per, r.NB);
    u32 sum = 0;
    for (u32 b = b0; b < b1; ++b) sum += cnt[b];
    u32 wtot;
    u32 run = wave_excl_scan(sum, wtot);
    if ((threadIdx.x & 63u) == 0) wsum[threadIdx.x >> 6] = wtot;
    __syncthreads();
    for (u32 w = 0; w < (threadIdx.x >> 6); ++w) run += wsum[w];
    for (u32 b = b0; b < b1; ++b) {
      const u32 c = cnt[b];
      lst[b] = run;
      run += c;
      cnt[b] = c ? atomicAdd(&gcur[(size_t)b * RF_GCS], c) : 0u;
    }
  }
  __syncthreads();
#pragma unroll
  for (u32 j = 0; j < SPW / RFB; ++j) {
    const u32 l = l0 + j * RFB + threadIdx.x;
#pragma unroll
    for (u32 k = 0; k < SIM_MAX_FANOUT; ++k) {
      if (k < nc[j]) {  // (a slot without a target — fewer other nodes than the fan-out — is in nobody's row)
        const u32 b = tg[j][k] >> 16, at = lst[b] + atomicAdd(&cur[b], 1u);
        stage[at] = ((E)(tg[j][k] & 0xFFFFu) << r.PB) | (E)(4u * l + k);
        stb[at] = (uint16_t)b;
      }
    }
  }
  __syncthreads();
  u32 total = 0;
  for (u32 w = 0; w < RFB / 64u; ++w) total += wsum[w];
  for (u32 i = threadIdx.x; i < total; i += RFB) {  // any order inside a bucket will do (rf_rows ranks)
    const u32 b = stb[i], at = cnt[b] + (i - lst[b]);
    const E e = stage[i];
    if (at < r.bcap) l1[(size_t)b * r.bcap + at] = e;
    else {
      const u32 o = atomicAdd(reinterpret_cast<u32*>(ovf), 1u);
      if (o < r.ocap) { ovf[1u + 2u * o] = (E)b; ovf[2u + 2u * o] = e; }
    }
  }
}
// LDS of rf_rows (dynamic): cnt[R + 1] | cur[R] | rowp[cap]
static inline size_t rf_rows_lds(const RfP& r) { return ((size_t)(2u << r.LB) + 1u + r.cap) * 4u + 16u; }
template <typename E>
__global__ __launch_bounds__(RFR) void rf_rows_kernel(RfP r, u32* gcur, u32* gcur_next, const E* l1, E* ovf, E* ovf_next, u32* rcsr, u32* rsrc, uint8_t* cntb, u32* btot, u32* xflag) {
  // (cntb != null: the SENDING side's sort of a shard — instead of row starts, one count byte per target of every destination
  // shard, cntb[h * M + t], and the bucket's total, btot[b]: what travels with the packets; rsrc = the sorted pair ids)
  extern __shared__ u32 rf_lds[];
  const u32 R = 1u << r.LB;
  u32 *cnt = rf_lds, *cur = cnt + R + 1u, *rowp = cur + R;
  __shared__ u32 wtot[RFR / 64u], s_base;
  const u32 b = blockIdx.x, n = gcur[(size_t)b * RF_GCS], nreg = min(n, r.bcap);
  const u32 hb = b / r.NBh, t0 = (b - hb * r.NBh) << r.LB, nrows = min(R, r.M - t0);
  const E pmask = ((E)1 << r.PB) - (E)1;
  // the bucket's place in the output: behind everything the buckets before it hold
  {
    u32 part = 0;
    for (u32 i = threadIdx.x; i < b; i += RFR) part += gcur[(size_t)i * RF_GCS];
    u32 wsum;
    (void)wave_excl_scan(part, wsum);
    if ((threadIdx.x & 63u) == 0) wtot[threadIdx.x >> 6] = wsum;
  }
  for (u32 i = threadIdx.x; i <= R; i += RFR) cnt[i] = 0;
  for (u32 i = threadIdx.x; i < R; i += RFR) cur[i] = 0;
  __syncthreads();
  if (threadIdx.x == 0) {
    u32 sum = 0;
    for (u32 w = 0; w < RFR / 64u; ++w) sum += wtot[w];
    s_base = sum;
    gcur_next[(size_t)b * RF_GCS] = 0;  // the next build's counters start from zero (it runs behind this one, on the same stream)
    if (b == 0) *reinterpret_cast<u32*>(ovf_next) = 0;
  }
  // entry i of the bucket: the first nreg in its region of l1, the rest on the overflow list (any order)
  auto entry = [&](u32 i) __attribute__((always_inline)) -> E {
    if (i < nreg) return l1[(size_t)b * r.bcap + i];
    u32 seen = 0, tot = min(*reinterpret_cast<const u32*>(ovf), r.ocap);
    for (u32 j = 0; j < tot; ++j)
      if (ovf[1u + 2u * j] == (E)b) { if (seen == i - nreg) return ovf[2u + 2u * j]; ++seen; }
    return (E)0;  // (cannot happen: the list holds the rest of this bucket — it cannot run full, sim_create sizes it for every pair)
  };
  const bool fits = n <= r.cap;
  u32 pv[RF_EPT];
  uint16_t tv[RF_EPT];
#pragma unroll
  for (u32 j = 0; j < RF_EPT; ++j) {
    const u32 i = threadIdx.x + j * RFR;
    pv[j] = NOSLOT; tv[j] = 0;
    if (fits && i < n) {
      const E e = entry(i);
      pv[j] = (u32)(e & pmask);
      tv[j] = (uint16_t)(e >> r.PB);
      atomicAdd(&cnt[tv[j]], 1u);
    }
  }
  if (!fits)
    for (u32 i = threadIdx.x; i < n; i += RFR) atomicAdd(&cnt[(u32)(entry(i) >> r.PB)], 1u);
  __syncthreads();
  const u32 base = s_base;
  {  // exclusive prefix over the bucket's rows: every thread R / RFR consecutive ones (R <= 4096; a smaller R: one each)
    const u32 per = (R + RFR - 1u) / RFR, i0 = threadIdx.x * per;
    u32 v[(1u << RF_LB_MAX) / RFR], sum = 0;
#pragma unroll
    for (u32 j = 0; j < (1u << RF_LB_MAX) / RFR; ++j) { v[j] = (j < per && i0 + j < R) ? cnt[i0 + j] : 0u; sum += v[j]; }
    u32 wsum, run = wave_excl_scan(sum, wsum);
    if ((threadIdx.x & 63u) == 0) wtot[threadIdx.x >> 6] = wsum;  // (everybody is past the sync behind the first use of wtot)
    __syncthreads();
    for (u32 w = 0; w < (threadIdx.x >> 6); ++w) run += wtot[w];
#pragma unroll
    for (u32 j = 0; j < (1u << RF_LB_MAX) / RFR; ++j)
      if (j < per && i0 + j < R) { cnt[i0 + j] = run; run += v[j]; }
    if (threadIdx.x == RFR - 1u) cnt[R] = run;
  }
  __syncthreads();
  if (cntb) {
    for (u32 i = threadIdx.x; i < nrows; i += RFR) {
      const u32 c = cnt[i + 1u] - cnt[i];
      if (c > 255u) atomicOr(xflag, 1u);  // (a node drawn by more than 255 of one shard's senders in one tick)
      cntb[(size_t)hb * r.M + t0 + i] = (uint8_t)min(c, 255u);
    }
    if (threadIdx.x == 0) btot[b] = n;
  } else {
    for (u32 i = threadIdx.x; i < nrows; i += RFR) rcsr[t0 + i] = base + cnt[i];
    if (b == r.NB - 1u && threadIdx.x == 0) rcsr[r.Nl] = base + n;
  }
  if (fits) {
#pragma unroll
    for (u32 j = 0; j < RF_EPT; ++j)
      if (pv[j] != NOSLOT) rowp[cnt[tv[j]] + atomicAdd(&cur[tv[j]], 1u)] = pv[j];
    __syncthreads();
#pragma unroll
    for (u32 j = 0; j < RF_EPT; ++j) {
      if (pv[j] != NOSLOT) {
        const u32 lo = cnt[tv[j]], hi = cnt[tv[j] + 1u];
        u32 rank = 0;
        for (u32 q = lo; q < hi; ++q) rank += rowp[q] < pv[j] ? 1u : 0u;
        tv[j] = (uint16_t)(lo + rank);  // (its place in the bucket: below 2^16, the row is not needed any more)
      }
    }
    __syncthreads();  // the rows have been read: the sorted bucket goes into the same LDS, then out as one run
#pragma unroll
    for (u32 j = 0; j < RF_EPT; ++j)
      if (pv[j] != NOSLOT) rowp[tv[j]] = pv[j];
    __syncthreads();
    for (u32 i = threadIdx.x; i < n; i += RFR)
      if (base + i < r.rcap) rsrc[base + i] = rowp[i];  // (a shard: more packets than rsrc has room for — mean + 12 sigma — cannot happen)
  } else {  // the bucket does not fit the tables: rank every pair against the whole bucket, straight from global memory
    for (u32 i = threadIdx.x; i < n; i += RFR) {
      const E e = entry(i);
      u32 rank = 0;
      for (u32 j = 0; j < n; ++j) {
        const E q = entry(j);
        rank += ((q >> r.PB) == (e >> r.PB) && (q & pmask) < (e & pmask)) ? 1u : 0u;
      }
      const u32 at = base + cnt[(u32)(e >> r.PB)] + rank;
      if (at < r.rcap) rsrc[at] = (u32)(e & pmask);
    }
  }
}
// ---- SIM_CF_RANDOM_FANOUT on a shard (r5): the packed exchange (include/serf_sim.h SIM_XCHG_PACKED) ----------------------------
// A packet goes to ANY node of the cluster.  The packets stay in their senders' cells (as on one GPU); per tick every shard
//   * sorts the (target, sender, slot) triples of its OWN senders by global target (rf_scatter / rf_rows above, two ticks ahead on
//     the build stream): rsrc = the sorted pair ids p = 4 l + k, cntb = one count byte per target of every destination shard,
//     btot = the buckets' totals, xoff = where every destination's pairs start (rfx_soff_kernel);
//   * PACKS, behind the tick kernel, the packets bound for shard h into slab h of the send buffer in that order (rfx_pack_kernel:
//     pair -> sender's cell 0 and its map word -> the packet's pages, one scattered 64-byte read per packet — the read the
//     receiver does on one GPU), the count bytes and bucket totals of h next to them (rfx_meta_kernel);
//   * after the round's all-to-all turns the V slabs it received into the tick's CSR (rfx_index_kernel): a node's row = V runs,
//     source shards ascending = senders ascending; an entry of rsrc = the packet's cell in the receive buffer, so the tick
//     kernel reads it exactly as it reads a sender's cell 0 (map word: "this cell, n pages"; pages adjacent: Dev::NC = 1).
// A slab, in 64-byte units: header {packets, overflow flag, tick} | count bytes [M] | bucket totals [NBh] u32 | cells [cap * PG].
struct RfxL {  // the layout of one slab (the same on every shard of a run)
  u32 V, M, NBh, LB, PG, cap;
  u32 cnt_u, tot_u, cell_u, slab_u;  // offsets of the three sections and the slab's size, in 64-byte units
};
static RfxL rfx_layout(u32 V, u32 M, u32 NBh, u32 LB, u32 PG, u32 f) {
  RfxL x;
  x.V = V; x.M = M; x.NBh = NBh; x.LB = LB; x.PG = PG;
  x.cap = serf_rf_slab_cap(f, M, V);
  x.cnt_u = 1u;
  x.tot_u = x.cnt_u + (M + 63u) / 64u;
  x.cell_u = x.tot_u + (NBh + 15u) / 16u;
  x.slab_u = x.cell_u + x.cap * PG;
  return x;
}
// where every destination's pairs start in the sorted list (xoff[0 .. V]); xoff[V + 1 + h] = 1 when slab h cannot hold them
__global__ void rfx_soff_kernel(RfxL x, const u32* btot, u32* xoff, u32* xflag) {
  __shared__ u32 part[64];
  for (u32 h = threadIdx.x; h < x.V; h += blockDim.x) {
    u32 sum = 0;
    for (u32 j = 0; j < x.NBh; ++j) sum += btot[(size_t)h * x.NBh + j];
    part[h] = sum;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    u32 run = 0;
    for (u32 h = 0; h < x.V; ++h) {
      xoff[h] = run;
      run += part[h];
      xoff[x.V + 1u + h] = part[h] > x.cap ? 1u : 0u;
      if (part[h] > x.cap) atomicOr(xflag, 2u);
    }
    xoff[x.V] = run;
  }
}
// the slabs' headers, count bytes and bucket totals (what the receiver makes its rows from)
__global__ void rfx_meta_kernel(RfxL x, const uint8_t* cntb, const u32* btot, const u32* xoff, u32 tick, uint4* send) {
  const size_t per = (size_t)x.M + (size_t)x.NBh * 4u + 64u;  // bytes of one slab's meta, header last
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < per * x.V; i += (size_t)gridDim.x * blockDim.x) {
    const u32 h = (u32)(i / per);
    const size_t o = i - (size_t)h * per;
    uint8_t* slab = reinterpret_cast<uint8_t*>(send + (size_t)h * x.slab_u * 4u);
    if (o < x.M) slab[(size_t)x.cnt_u * 64u + o] = cntb[(size_t)h * x.M + o];
    else if (o < (size_t)x.M + (size_t)x.NBh * 4u) {
      const size_t w = o - x.M;
      if ((w & 3u) == 0u) reinterpret_cast<u32*>(slab + (size_t)x.tot_u * 64u)[w >> 2] = btot[(size_t)h * x.NBh + (w >> 2)];
    } else {
      const size_t w = o - x.M - (size_t)x.NBh * 4u;
      if ((w & 3u) == 0u) {
        const u32 n = xoff[h + 1u] - xoff[h], over = xoff[x.V + 1u + h];
        const u32 i4 = (u32)(w >> 2);
        reinterpret_cast<u32*>(slab)[i4] = i4 == 0u ? min(n, x.cap) : i4 == 1u ? over : i4 == 2u ? tick : 0u;
      }
    }
  }
}
// pair i of the sorted list -> its packet, copied into its place in the slab of its destination: four lanes per packet, one
// 16-byte quarter each, consecutive pairs -> consecutive cells (dense 64-byte writes); the fourth quarter of a packet's first
// cell is its map word ("this cell, n pages" — or all 0xFF: nothing was sent / the packet was lost)
__global__ void rfx_pack_kernel(RfxL x, const u32* rsrc, const u32* xoff, const uint4* cells, u32 Nl, uint4* send) {
  const u32 total = xoff[x.V];
  const u32 q = threadIdx.x & 3u;
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2; i < total; i += ((size_t)gridDim.x * blockDim.x) >> 2) {
    u32 h = 0;
    while (h + 1u < x.V && xoff[h + 1u] <= (u32)i) ++h;
    const u32 pos = (u32)i - xoff[h];
    if (pos >= x.cap) continue;  // (the slab is full: its header says so and the step fails)
    const u32 p = rsrc[i], l = p >> 2, k = p & 3u;
    const u32 mw = cells[(size_t)l * RF_CELL_U4 + 3u].x, jb = (mw >> (8u * k)) & 0xFFu;
    const u32 pages = jb == 0xFFu ? 0u : (jb & 3u) + 1u;
    uint4* dst = send + ((size_t)h * x.slab_u + x.cell_u + (size_t)pos * x.PG) * 4u;
    for (u32 pg = 0; pg < x.PG; ++pg) {
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (q == 3u) v.x = pg == 0u ? (pages ? (0xFFFFFF00u | (pages - 1u)) : 0xFFFFFFFFu) : 0xFFFFFFFFu;
      else if (pg < pages) v = cells[((size_t)((jb >> 2) + pg) * Nl + l) * RF_CELL_U4 + q];
      dst[(size_t)pg * 4u + q] = v;
    }
  }
}
// the receiving side: one workgroup per bucket of 2^LB targets — the bucket's place in every source's slab and in the rows is
// the sum of the totals before it (they travelled with the packets); inside it, prefix sums over the count bytes
#define RFX_T 256u
__global__ __launch_bounds__(RFX_T) void rfx_index_kernel(RfxL x, const uint4* recv, u32 Nl, u32 rcap, u32* rcsr, u32* rsrc, u32* xflag) {
  extern __shared__ u32 rfx_lds[];  // rs[R]: where the next entry of every row goes
  __shared__ u32 wsum[RFX_T / 64u], s_base[2];
  u32* rs = rfx_lds;
  const u32 R = 1u << x.LB, b = blockIdx.x, t0 = b << x.LB, nrows = min(R, x.M - t0);
  const u32 per = (R + RFX_T - 1u) / RFX_T, i0 = threadIdx.x * per, i1 = min(i0 + per, nrows);
  auto slab = [&](u32 g) __attribute__((always_inline)) -> const uint8_t* { return reinterpret_cast<const uint8_t*>(recv + (size_t)g * x.slab_u * 4u); };
  auto block_excl = [&](u32 v, u32& total) __attribute__((always_inline)) -> u32 {  // exclusive prefix over the workgroup's threads
    u32 wt, run = wave_excl_scan(v, wt);
    __syncthreads();
    if ((threadIdx.x & 63u) == 0) wsum[threadIdx.x >> 6] = wt;
    __syncthreads();
    total = 0;
    for (u32 w = 0; w < RFX_T / 64u; ++w) { if (w < (threadIdx.x >> 6)) run += wsum[w]; total += wsum[w]; }
    return run;
  };
  // the rows' starts: everything the buckets before this one hold, over all sources, then the prefix of the rows' lengths
  u32 before = 0;
  for (u32 g = 0; g < x.V; ++g) {
    const u32* tot = reinterpret_cast<const u32*>(slab(g) + (size_t)x.tot_u * 64u);
    for (u32 j = threadIdx.x; j < b; j += RFX_T) before += tot[j];
    if (threadIdx.x == 0 && b == 0) {  // (one workgroup looks at the headers: a sender whose slab ran full says so here)
      const u32* hd = reinterpret_cast<const u32*>(slab(g));
      if (hd[1] != 0u || hd[0] > x.cap) atomicOr(xflag, 4u);
    }
  }
  u32 mine = 0;
  for (u32 i = i0; i < i1; ++i) {
    u32 c = 0;
    for (u32 g = 0; g < x.V; ++g) c += slab(g)[(size_t)x.cnt_u * 64u + t0 + i];
    mine += c;
  }
  u32 tot_before, tot_rows;
  (void)block_excl(before, tot_before);
  u32 run = block_excl(mine, tot_rows) + tot_before;
  for (u32 i = i0; i < i1; ++i) {
    u32 c = 0;
    for (u32 g = 0; g < x.V; ++g) c += slab(g)[(size_t)x.cnt_u * 64u + t0 + i];
    rs[i] = run;
    rcsr[t0 + i] = run;
    run += c;
  }
  if (b == gridDim.x - 1u && threadIdx.x == RFX_T - 1u) rcsr[Nl] = tot_before + tot_rows;
  if (threadIdx.x == 0 && tot_before + tot_rows > rcap) atomicOr(xflag, 8u);  // (more packets than rsrc has room for: mean + 12 sigma)
  __syncthreads();
  // source by source: the entries of a row's run from source g = the cells of g's slab from the row's offset on
  for (u32 g = 0; g < x.V; ++g) {
    const u32* tot = reinterpret_cast<const u32*>(slab(g) + (size_t)x.tot_u * 64u);
    const uint8_t* cb = slab(g) + (size_t)x.cnt_u * 64u + t0;
    u32 bef = 0, sum = 0;
    for (u32 j = threadIdx.x; j < b; j += RFX_T) bef += tot[j];
    for (u32 i = i0; i < i1; ++i) sum += cb[i];
    u32 tb, ts;
    (void)block_excl(bef, tb);
    u32 off = block_excl(sum, ts) + tb;
    const u32 cell0 = g * x.slab_u + x.cell_u;  // (64-byte units from the start of the receive buffer)
    for (u32 i = i0; i < i1; ++i) {
      const u32 c = cb[i];
      u32 at = rs[i];
      for (u32 k = 0; k < c; ++k, ++at, ++off)
        if (at < rcap) rsrc[at] = (cell0 + off * x.PG) << 2;
      rs[i] = at;
    }
  }
}
// gossip_to_the_dead_time with random fan-out (oracle tick_node): bit k of the node's skip byte = its view, as the tick begins,
// says the target it drew for slot k has been dead / left for longer than that.  Its own launch, only when the option is on.
__global__ void rf_skip_kernel(Dev d, TickP tp, RfP r, const uint4* base) {
  for (size_t l = (size_t)blockIdx.x * blockDim.x + threadIdx.x; l < d.Nl; l += (size_t)gridDim.x * blockDim.x) {
    u32 ch[SIM_MAX_FANOUT], m = 0;
    const u32 nc = rf_draw(r.rb, d.shard0 + (u32)l, r.N, r.feff, ch);
    for (u32 k = 0; k < nc; ++k) {
      const u32 key = SEL4(k, ch[0], ch[1], ch[2], ch[3]), a = d.slot_of[key];
      uint4 e = a == NOSLOT ? base[(size_t)key * 2] : d.view[(size_t)a * d.Nl + l];
      u32 sw = SIM_VB_SWIM(e.w);
      if ((e.w & SIM_VB_KNOWN) && (sw == SIM_SWIM_DEAD || sw == SIM_SWIM_LEFT) && ((((u32)tp.tick - SIM_VB_STAMP(e.w)) & STAMP_MASK) > d.gttd)) m |= 1u << k;
    }
    d.skipmask[l] = (uint8_t)m;
  }
}
// gossip_to_the_dead_time (App. B.2; oracle gossip_skips): for every node and fan-out slot, does the node's view — as it is when
// the tick begins — say that the packet's target has been dead / left for longer than that?  Its own launch, ahead of
// the tick kernel and only when the option is on: the tick kernel then reads one byte per node.
__global__ void gossip_skip_kernel(Dev d, TickP tp, const uint4* base) {
  for (size_t l = (size_t)blockIdx.x * blockDim.x + threadIdx.x; l < d.Nl; l += (size_t)gridDim.x * blockDim.x) {
    u32 gid = d.shard0 + (u32)l, g = gid / tp.M, ll = gid - g * tp.M, m = 0;
    for (u32 k = 0; k < tp.feff; ++k) {
      u32 hh, tt;
      fan_target_g(tp, g, ll, k, hh, tt);
      u32 target = hh * tp.M + tt, a = d.slot_of[target];
      uint4 e = a == NOSLOT ? base[(size_t)target * 2] : d.view[(size_t)a * d.Nl + l];
      u32 sw = SIM_VB_SWIM(e.w);
      if ((e.w & SIM_VB_KNOWN) && (sw == SIM_SWIM_DEAD || sw == SIM_SWIM_LEFT) && ((((u32)tp.tick - SIM_VB_STAMP(e.w)) & STAMP_MASK) > d.gttd)) m |= 1u << k;
    }
    d.skipmask[l] = (uint8_t)m;
  }
}
// SIM_CF_JOIN_SYNC (oracle join_sync): memberlist.join = a push-pull with the peer — the joining node adopts the view of a
// running node of its own shard.  One block: the partner is picked by thread 0, the entries are copied in parallel, the
// suspicion timers of the adopted entries are listed in walk order, chunk by chunk.
__global__ void join_sync_kernel(Dev d, u32 n_slots, u32 gid, u32 peer, u32 tick) {
  __shared__ u32 s_partner;
  __shared__ u32 s_flag[BLOCK];  // view slot + 1 of an adopted entry that is suspect, 0 otherwise
  __shared__ u32 s_nt, s_next, s_ovf;
  const u32 M = d.M, base = (gid / M) * M, l = gid - d.shard0;
  if (!threadIdx.x) {
    u32 partner = NOSLOT;
    for (u32 i = 0; i < M && partner == NOSLOT; ++i) {
      u32 cand = base + (peer % M + i) % M;
      if (cand != gid && up_of(d, cand)) partner = cand;
    }
    s_partner = partner;
    s_nt = 0; s_next = 0xFFFFFFFFu; s_ovf = 0;
  }
  __syncthreads();
  if (s_partner == NOSLOT) return;
  const u32 lp = s_partner - d.shard0;
  uint16_t* sp = reinterpret_cast<uint16_t*>(&d.R4[2 * (size_t)l]);
  if (threadIdx.x < SIM_S) sp[threadIdx.x] = 0;
  for (u32 w0 = 0; w0 < n_slots; w0 += BLOCK) {
    u32 wi = w0 + threadIdx.x;
    u32 flag = 0, deadline = 0;
    if (wi < n_slots) {
      u32 a = d.walk[wi];
      uint4* e = d.view + ((size_t)a * d.Nl + l);
      const uint4* pe = d.view + ((size_t)a * d.Nl + lp);
      uint4 h = pe[0];
      if (d.subject_of[a] == gid) {  // its own entry stays its own
        uint4 mine = e[0];
        if (h.z > mine.z) { mine.z = h.z; e[0] = mine; }
      } else {
        e[0] = h;
        e[d.vtail] = pe[d.vtail];
        if ((h.w & SIM_VB_KNOWN) && SIM_VB_SWIM(h.w) == SIM_SWIM_SUSPECT) {  // the adopted suspicion keeps running here
          flag = a + 1;
          deadline = tick - ((tick - SIM_VB_STAMP(h.w)) & STAMP_MASK) + d.T[SIM_VB_NCONF(h.w)];
        }
      }
    }
    s_flag[threadIdx.x] = flag;
    __syncthreads();
    if (!threadIdx.x)
      for (u32 i = 0; i < BLOCK; ++i)
        if (s_flag[i]) {
          if (s_nt == SIM_S) s_ovf++;
          else sp[s_nt++] = (uint16_t)s_flag[i];
        }
    __syncthreads();
    // earliest deadline over the TRACKED timers: thread 0 cannot see the deadlines, so every flagged thread checks whether
    // its slot made it into the list
    if (flag) {
      bool tracked = false;
      for (u32 j = 0; j < SIM_S; ++j) tracked |= sp[j] == (uint16_t)flag;
      if (tracked) atomicMin(&s_next, deadline);
    }
    __syncthreads();
  }
  if (!threadIdx.x) {
    uint4 r0 = d.R0[l], r1 = d.R1[l], r2 = d.R2[l], r3 = d.R3[l];
    const uint4 p0 = d.R0[lp], p1 = d.R1[lp], p2 = d.R2[lp], p3 = d.R3[lp];
    u32 a_me = d.slot_of[gid];
    u32 pst = SIM_STATUS_NONE, mst = SIM_STATUS_NONE;
    if (a_me != NOSLOT) {
      uint4 pm = d.view[(size_t)a_me * d.Nl + lp], mm = d.view[(size_t)a_me * d.Nl + l];
      if (pm.w & SIM_VB_KNOWN) pst = SIM_VB_STATUS(pm.w);
      if (mm.w & SIM_VB_KNOWN) mst = SIM_VB_STATUS(mm.w);
    }
    r3.y = s_next == 0xFFFFFFFFu ? 0u : s_next;  // susp_next
    r3.w = p3.w;         // reap_next
    r1.w = p1.w;         // n_known
    r2.x = p2.x - (pst == SIM_STATUS_FAILED ? 1u : 0u) + (mst == SIM_STATUS_FAILED ? 1u : 0u);
    r2.y = p2.y - (pst == SIM_STATUS_LEFT ? 1u : 0u) + (mst == SIM_STATUS_LEFT ? 1u : 0u);
    r2.w += s_ovf;
    u64 c = (u64)r0.x | ((u64)r0.y << 32), ec = (u64)r0.z | ((u64)r0.w << 32), qc = (u64)r1.x | ((u64)r1.y << 32);
    u64 pc = (u64)p0.x | ((u64)p0.y << 32), pec = (u64)p0.z | ((u64)p0.w << 32), pqc = (u64)p1.x | ((u64)p1.y << 32);
    if (pc > 0 && pc - 1 >= c) c = pc;       // witness(remote - 1): delegate.rs:466-480
    if (pec > 0 && pec - 1 >= ec) ec = pec;
    if (pqc > 0 && pqc - 1 >= qc) qc = pqc;
    r0 = make_uint4((u32)c, (u32)(c >> 32), (u32)ec, (u32)(ec >> 32));
    r1.x = (u32)qc; r1.y = (u32)(qc >> 32);
    d.R0[l] = r0; d.R1[l] = r1; d.R2[l] = r2; d.R3[l] = r3;
  }
}

// ------------------------------------------------------------------------------------------------
// push-pull anti-entropy (memberlist pushPull, App. B.6; SerfDelegate::local_state / merge_remote_state,
// serf-core/src/serf/delegate.rs:386-554) — SIMSPEC §2.10, oracle pp_round/pp_pair/pp_merge.
// One lane per synchronising pair; a batch holds 1/PP_GROUPS of all pairs, so the launch fills the chip
// and happens once per (scaled interval / PP_GROUPS) ticks: off the per-tick critical path.
// ------------------------------------------------------------------------------------------------
#define PP_GROUPS 8u
// What one side of a push-pull ships to the other (memberlist's node states + SerfDelegate::local_state,
// delegate.rs:386-425): the three clocks, the 16-byte heads of the view entries in walk (subject) order, the event
// ring.  The partner's copy is read either in place (it lives on this shard) or from a flat record a remote shard
// exported (sim_pp_export): [0] {clock, event_clock} [1] {query_clock, 0} [2 .. 2+ns) heads [..] Bev x {head, tail}.
struct PPLocal {
  const Dev& d;
  u32 lr;
  __device__ void clocks(u64& c, u64& e, u64& q) const {
    uint4 r0 = d.R0[lr], r1 = d.R1[lr];
    c = (u64)r0.x | ((u64)r0.y << 32); e = (u64)r0.z | ((u64)r0.w << 32); q = (u64)r1.x | ((u64)r1.y << 32);
  }
  __device__ uint4 head(u32 wi) const { return d.view[(size_t)d.walk[wi] * d.Nl + lr]; }
  __device__ uint4 bucket_head(u32 idx) const { return d.ering[(size_t)idx * d.Nl + lr]; }
  __device__ uint4 bucket_tail(u32 idx) const { return d.ering[d.etail + (size_t)idx * d.Nl + lr]; }
};
struct PPRecord {
  const uint4* rec;
  u32 ns;
  __device__ void clocks(u64& c, u64& e, u64& q) const {
    uint4 r0 = rec[0], r1 = rec[1];
    c = (u64)r0.x | ((u64)r0.y << 32); e = (u64)r0.z | ((u64)r0.w << 32); q = (u64)r1.x | ((u64)r1.y << 32);
  }
  __device__ uint4 head(u32 wi) const { return rec[2 + wi]; }
  __device__ uint4 bucket_head(u32 idx) const { return rec[2 + ns + 2 * (size_t)idx]; }
  __device__ uint4 bucket_tail(u32 idx) const { return rec[2 + ns + 2 * (size_t)idx + 1]; }
};
// local <- remote: memberlist mergeState, then merge_remote_state(is_join = false)
template <class Remote>
__device__ static void pp_merge(const Dev& d, const TickP& tp, u32 ll, const Remote& rm) {
  Ctx c{d, ll, d.shard0 + ll, (u32)tp.tick, tp.query_base};
  Node n;
  node_load(d, ll, n);
  u32 sk[SIM_Q];
  u32 cnt0 = __popc(n.used);
  keys_load(d, ll, cnt0, sk);
  if (n.next_seq > 1023u - 64u) q_renorm(n, sk);
  const uint4 zero = make_uint4(0, 0, 0, 0);
  bool dirty = false;
  if (d.swim) {  // alive as alive, left as dead{from = node}, suspect and dead as suspect
#pragma unroll 1
    for (u32 wi = 0; wi < tp.n_slots; ++wi) {
      u32 a = d.walk[wi];
      uint4 re = rm.head(wi);
      if (!(re.w & SIM_VB_KNOWN)) continue;
      if (n.next_seq > 1023u - 64u) q_renorm(n, sk);  // a merge can queue one broadcast per view slot
      u32 subj = d.subject_of[a], sw = SIM_VB_SWIM(re.w), inc = re.z;
      uint4* p = view_slot_ptr(c, a);
      uint4 e = p[0];
      Ins ins;
      ins.has = ins.wide = 0;
      if (sw == SIM_SWIM_ALIVE) swim_alive(c, n, subj, inc, wire_meta(SIM_K_ALIVE, 0, 64), p, e, dirty, ins);
      else if (sw == SIM_SWIM_LEFT) swim_dead(c, n, subj, inc, subj, wire_meta(SIM_K_DEAD, 0, 32), p, e, dirty, ins);
      else swim_suspect(c, n, subj, inc, c.gid, wire_meta(SIM_K_SUSPECT, 0, 32), p, e, dirty, ins);
      if (ins.has) q_insert(c, n, sk, ins.key, ins.wmeta, ins.val);
    }
  }
  u64 rclock, reclock, rqclock;
  rm.clocks(rclock, reclock, rqclock);
  if (rclock > 0) witness(n, n.clock, rclock - 1, DR0);     // delegate.rs:466-480
  if (reclock > 0) witness(n, n.eclock, reclock - 1, DR0);
  if (rqclock > 0) witness(n, n.qclock, rqclock - 1, DR1);
#pragma unroll 1
  for (u32 pass = 0; pass < 2; ++pass) {  // left members first (at status_ltime + 1), then the join intents
#pragma unroll 1
    for (u32 wi = 0; wi < tp.n_slots; ++wi) {
      u32 a = d.walk[wi];
      uint4 re = rm.head(wi);
      if (!(re.w & SIM_VB_KNOWN)) continue;
      bool left = SIM_VB_STATUS(re.w) == SIM_STATUS_LEFT;
      if (left != (pass == 0)) continue;
      if (left && n.next_seq > 1023u - 64u) q_renorm(n, sk);
      u32 subj = d.subject_of[a];
      uint4* p = view_slot_ptr(c, a);
      uint4 e = p[0];
      Ins ins;
      ins.has = ins.wide = 0;
      if (left) handle_leave_intent(c, n, subj, E_LTIME(re) + 1, false, p, e, dirty, ins);  // delegate.rs:495-512
      else handle_join_intent(c, n, subj, E_LTIME(re), p, e, dirty);                        // delegate.rs:515-526
      if (ins.has) q_insert(c, n, sk, ins.key, ins.wmeta, ins.val);                         // a refutation
    }
  }
#pragma unroll 1
  for (u32 idx = 0; idx < d.Bev; ++idx) {  // replay the remote event buffer: delegate.rs:540-552
    uint4 b0 = rm.bucket_head(idx);
    if (!b0.z) continue;
    uint4 b1 = b0.w ? rm.bucket_tail(idx) : zero;
    u64 lt = E_LTIME(b0);
    u32 keys[SIM_C] = {b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
    for (int k = 0; k < (int)SIM_C; ++k) {
      if (!keys[k]) break;
      uint4* p = ering_ptr(c, lt);
      uint4 e_ = p[0];
      handle_user_event(c, n, keys[k], lt, p, e_, dirty);
    }
  }
  node_store(d, ll, n);
  keys_store(d, ll, cnt0, n.used, sk);
}
// The pairs of a batch: the tick's matching {sigma_N^-1(2 pi), sigma_N^-1(2 pi + 1)} over ALL N nodes — memberlist's
// pushPull picks any peer (App. B.6), whichever shard it lives on — restricted to class pi mod PP_GROUPS.  The node
// with the even sigma value merges first, then the other merges its updated state.  This kernel: every shard local.
__global__ void pushpull_kernel(Dev d, TickP tp, u32 cls, u32 n_pairs) {
  u32 j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_pairs) return;
  u32 pi = cls + j * PP_GROUPS;
  if (2 * (u64)pi + 1 >= tp.N) return;
  u32 la = sigma_g_inv(tp, 2 * pi), lb = sigma_g_inv(tp, 2 * pi + 1);
  if (!up_of(d, la) || !up_of(d, lb)) return;  // a TCP exchange needs both ends
  pp_merge(d, tp, la, PPLocal{d, lb});
  pp_merge(d, tp, lb, PPLocal{d, la});
}
// sharded runs (host-planned, sim_pp_*): the in-shard pairs ...
__global__ void pp_local_kernel(Dev d, TickP tp, const u32* la, const u32* lb, u32 n) {
  u32 j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  pp_merge(d, tp, la[j], PPLocal{d, lb[j]});
  pp_merge(d, tp, lb[j], PPLocal{d, la[j]});
}
// ... a handful of pairs handed over by value (the Reconnector's attempts of a tick: pairwise disjoint, `a` merges first) ...
struct PairBatch { u32 n; u32 a[8], b[8]; };
__global__ void pp_pairs_kernel(Dev d, TickP tp, PairBatch pb) {
  u32 j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= pb.n) return;
  pp_merge(d, tp, pb.a[j], PPLocal{d, pb.b[j]});
  pp_merge(d, tp, pb.b[j], PPLocal{d, pb.a[j]});
}
// ... the records this shard ships (one block per record: `ns` view heads + the event ring) ...
__global__ void pp_export_kernel(Dev d, const u32* list, u32 ns, uint4* out, size_t rec_u4) {
  u32 l = list[blockIdx.x];
  uint4* o = out + (size_t)blockIdx.x * rec_u4;
  for (size_t i = threadIdx.x; i < rec_u4; i += blockDim.x) {
    uint4 v;
    if (i == 0) v = d.R0[l];
    else if (i == 1) { uint4 r1 = d.R1[l]; v = make_uint4(r1.x, r1.y, 0, 0); }
    else if (i < 2 + (size_t)ns) v = d.view[(size_t)d.walk[i - 2] * d.Nl + l];
    else {
      size_t k = i - 2 - ns;
      v = d.ering[((k & 1) ? d.etail : 0) + (k >> 1) * d.Nl + l];
    }
    o[i] = v;
  }
}
// ... and the merges from the records it received
__global__ void pp_cross_kernel(Dev d, TickP tp, const u32* list, u32 n, const uint4* recs, size_t rec_u4) {
  u32 j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  pp_merge(d, tp, list[j], PPRecord{recs + (size_t)j * rec_u4, tp.n_slots});
}

// ------------------------------------------------------------------------------------------------
// support kernels: fills, canonical forms, digest, members, convergence, stats
// ------------------------------------------------------------------------------------------------
__global__ void fill_u32(u32* p, size_t n, u32 v) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void fill_u4(uint4* p, size_t n, uint4 v) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
// column a of the view := the subject's baseline entry
__global__ void fill_view_col(uint4* view, size_t tail, size_t Nl, u32 a, uint4 e0, uint4 e1) {
  for (size_t l = blockIdx.x * (size_t)blockDim.x + threadIdx.x; l < Nl; l += (size_t)gridDim.x * blockDim.x) {
    view[(size_t)a * Nl + l] = e0;
    view[tail + (size_t)a * Nl + l] = e1;
  }
}
// the per-subject baseline table stays entry-interleaved ([N][2]): it is read by probes and status queries only
__global__ void fill_base(uint4* base, size_t n, uint4 e0, uint4 e1) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    base[i * 2] = e0;
    base[i * 2 + 1] = e1;
  }
}
// canonical (interleaved, 32-byte) form of `count` entries starting at entry `first` of a split array, and back
__global__ void canon_entries_kernel(const uint4* arr, size_t tail, size_t first, size_t count, uint4* out) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
    out[i * 2] = arr[first + i];
    out[i * 2 + 1] = arr[tail + first + i];
  }
}
__global__ void uncanon_entries_kernel(uint4* arr, size_t tail, size_t first, size_t count, const uint4* in) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
    arr[first + i] = in[i * 2];
    arr[tail + first + i] = in[i * 2 + 1];
  }
}
// keep d.walk sorted by subject: shift [pos, count) up by one, put slot a at pos (one thread; slot allocation is rare)
__device__ static inline void walk_insert(u32* walk, u32 count, u32 pos, u32 a) {
  for (u32 i = count; i > pos; --i) walk[i] = walk[i - 1];
  walk[pos] = a;
}
// A subject takes view slot a: its column of the view := the subject's baseline entry, the slot joins the walk order, the
// two slot maps point at each other.  ONE launch (it used to be four, 4 - 8 us of stream time each, in front of the tick of
// every operation that names a new subject).
__global__ void slot_alloc_kernel(uint4* view, size_t tail, size_t Nl, u32 a, uint4 e0, uint4 e1, u32* walk, u32 count, u32 pos,
                                  u32* slot_of_x, u32* subject_of_a, u32 subject) {
  if (!blockIdx.x && !threadIdx.x) {
    walk_insert(walk, count, pos, a);
    *slot_of_x = a;
    *subject_of_a = subject;
  }
  for (size_t l = blockIdx.x * (size_t)blockDim.x + threadIdx.x; l < Nl; l += (size_t)gridDim.x * blockDim.x) {
    view[(size_t)a * Nl + l] = e0;
    view[tail + (size_t)a * Nl + l] = e1;
  }
}
__global__ void fill_iota(u32* p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (u32)i;
}
__global__ void init_dense_self(Dev d) {  // new_in's synthetic notify_join(local): self known, Alive @ 0
  for (size_t l = blockIdx.x * (size_t)blockDim.x + threadIdx.x; l < d.Nl; l += (size_t)gridDim.x * blockDim.x) {
    u32 gid = d.shard0 + (u32)l;
    d.view[(size_t)gid * d.Nl + l] = make_uint4(0, 0, 0, 1u | (SIM_STATUS_ALIVE << 1));
  }
}

// canonical sim_row (12 x u64 words) of node l
#define ROW_W 14  // u64 words of a canonical sim_row
static_assert(sizeof(sim_row) == ROW_W * 8, "canonical row");
__device__ static inline void canon_row(const Dev& d, size_t l, u64 (&w)[ROW_W]) {
  uint4 r0 = d.R0[l], r1 = d.R1[l], r2 = d.R2[l], r3 = d.R3[l], r4 = d.R4[2 * l], r4b = d.R4[2 * l + 1], r5 = d.R5[l];
  w[0] = (u64)r0.x | ((u64)r0.y << 32);
  w[1] = (u64)r0.z | ((u64)r0.w << 32);
  w[2] = (u64)r1.x | ((u64)r1.y << 32);
  w[3] = (u64)r5.x | ((u64)r5.y << 32);
  w[4] = (u64)r5.z | ((u64)r5.w << 32);
  w[5] = (u64)r1.z | ((u64)r3.x << 32);              // flags, inc
  w[6] = (u64)r1.w | ((u64)r2.x << 32);              // n_known, n_failed
  w[7] = (u64)r2.y | ((u64)(r2.z & 0xFFFFu) << 32);  // n_left, next_seq
  w[8] = (u64)r2.w | ((u64)r3.y << 32);              // overflow, susp_next
  w[9] = (u64)r3.z | ((u64)r3.w << 32);              // awareness, reap_next
  w[10] = (u64)r4.x | ((u64)r4.y << 32);
  w[11] = (u64)r4.z | ((u64)r4.w << 32);
  w[12] = (u64)r4b.x | ((u64)r4b.y << 32);
  w[13] = (u64)r4b.z | ((u64)r4b.w << 32);
}
// canonical sim_record i (drain order) of node l
__device__ static inline uint4 canon_qrec(const Dev& d, size_t l, u32 i, u32 cnt) {
  if (i >= cnt) return make_uint4(0u, SIM_META_EMPTY, 0u, 0u);
  uint4 kq = d.qkeys[(size_t)(i >> 2) * d.Nl + l];
  u32 k = (i & 3) == 0 ? kq.x : (i & 3) == 1 ? kq.y : (i & 3) == 2 ? kq.z : kq.w;
  uint4 pay = d.qpay[(size_t)(k & 15u) * d.Nl + l];
  return make_uint4(pay.x, ((k >> 4) << 8) | (pay.y & 0xFFu), pay.z, pay.w);
}
__global__ void canon_rows_kernel(Dev d, u64* out) {
  for (size_t l = blockIdx.x * (size_t)blockDim.x + threadIdx.x; l < d.Nl; l += (size_t)gridDim.x * blockDim.x) {
    u64 w[ROW_W];
    canon_row(d, l, w);
    for (int i = 0; i < ROW_W; ++i) out[l * ROW_W + i] = w[i];
  }
}
__global__ void canon_queue_kernel(Dev d, uint4* out) {
  for (size_t l = blockIdx.x * (size_t)blockDim.x + threadIdx.x; l < d.Nl; l += (size_t)gridDim.x * blockDim.x) {
    u32 cnt = __popc(d.R2[l].z >> 16);
    for (u32 i = 0; i < SIM_Q; ++i) out[l * SIM_Q + i] = canon_qrec(d, l, i, cnt);
  }
}

// ---- checkpoint / resume: canonical image -> physical layout ----------------------------------------
__global__ void restore_rows_kernel(Dev d, const u64* in /* [Nl][ROW_W] canonical sim_row */) {
  for (size_t l = blockIdx.x * (size_t)blockDim.x + threadIdx.x; l < d.Nl; l += (size_t)gridDim.x * blockDim.x) {
    const u64* w = in + l * ROW_W;
    u32 used = d.R2[l].z >> 16;  // set by restore_queue_kernel, which runs first
    d.R0[l] = make_uint4((u32)w[0], (u32)(w[0] >> 32), (u32)w[1], (u32)(w[1] >> 32));
    d.R1[l] = make_uint4((u32)w[2], (u32)(w[2] >> 32), (u32)w[5], (u32)w[6]);
    d.R2[l] = make_uint4((u32)(w[6] >> 32), (u32)w[7], ((u32)(w[7] >> 32) & 0xFFFFu) | (used << 16), (u32)w[8]);
    d.R3[l] = make_uint4((u32)(w[5] >> 32), (u32)(w[8] >> 32), (u32)w[9], (u32)(w[9] >> 32));
    d.R4[2 * l] = make_uint4((u32)w[10], (u32)(w[10] >> 32), (u32)w[11], (u32)(w[11] >> 32));
    d.R4[2 * l + 1] = make_uint4((u32)w[12], (u32)(w[12] >> 32), (u32)w[13], (u32)(w[13] >> 32));
    d.R5[l] = make_uint4((u32)w[3], (u32)(w[3] >> 32), (u32)w[4], (u32)(w[4] >> 32));
  }
}
__global__ void restore_queue_kernel(Dev d, const uint4* in /* [Nl][Q] canonical sim_record, drain order */) {
  for (size_t l = blockIdx.x * (size_t)blockDim.x + threadIdx.x; l < d.Nl; l += (size_t)gridDim.x * blockDim.x) {
    u32 k[SIM_Q], cnt = 0;
    for (u32 i = 0; i < SIM_Q; ++i) {
      uint4 r = in[l * SIM_Q + i];
      if (r.y == SIM_META_EMPTY) { k[i] = KEMPTY; continue; }
      k[i] = ((r.y >> 8) << 4) | i;  // payload slot = rank
      d.qpay[(size_t)i * d.Nl + l] = make_uint4(r.x, r.y & SIM_META_WIRE_MASK, r.z, r.w);
      cnt++;
    }
    for (u32 g = 0; g < 4; ++g) d.qkeys[(size_t)g * d.Nl + l] = make_uint4(k[4 * g], k[4 * g + 1], k[4 * g + 2], k[4 * g + 3]);
    uint4 r2 = d.R2[l];
    r2.z = (r2.z & 0xFFFFu) | (((1u << cnt) - 1u) << 16);
    d.R2[l] = r2;
  }
}

__device__ static inline u64 dig(u64 w, u64 idx) { return mix64(w ^ (idx * 0xD1342543DE82EF95ull)); }
__device__ static inline void block_sum_add(u64 v, u64* out) {
  __shared__ u64 sm[BLOCK / 64];
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_down(v, o, 64);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    u64 t = 0;
    for (int i = 0; i < BLOCK / 64; ++i) t += sm[i];
    atomicAdd((unsigned long long*)out, (unsigned long long)t);
  }
  __syncthreads();
}
// digest of a flat array whose physical word order IS the canonical order
__global__ void digest_flat(const u64* w, size_t n_words, u64* out) {
  u64 acc = 0;
  for (size_t i = blockIdx.x * (size_t)BLOCK + threadIdx.x; i < n_words; i += (size_t)gridDim.x * BLOCK) acc += dig(w[i], i);
  block_sum_add(acc, out);
}
// digest of a split array in its canonical (interleaved) word order: entry e = words 4e, 4e+1 (head), 4e+2, 4e+3 (tail)
__global__ void digest_split(const uint4* arr, size_t tail, size_t n_entries, u64* out) {
  u64 acc = 0;
  for (size_t e = blockIdx.x * (size_t)BLOCK + threadIdx.x; e < n_entries; e += (size_t)gridDim.x * BLOCK) {
    uint4 h = arr[e], t = arr[tail + e];
    acc += dig((u64)h.x | ((u64)h.y << 32), e * 4) + dig((u64)h.z | ((u64)h.w << 32), e * 4 + 1);
    acc += dig((u64)t.x | ((u64)t.y << 32), e * 4 + 2) + dig((u64)t.z | ((u64)t.w << 32), e * 4 + 3);
  }
  block_sum_add(acc, out);
}
// running queries: tracker table, then the ack / response bitmaps (canonical order = physical order)
// ... then the filters (32-bit words) and the tag classes (bytes)
__global__ void digest_queries(const uint4* qtab, const u32* qbits, size_t n_bits_words, u32 N, u64* out) {
  u64 acc = 0;
  const size_t n0 = 2 * SIM_QT, n1 = n0 + n_bits_words, n2 = n1 + (size_t)SIM_QT * SIM_QF_WORDS, n3 = n2 + N;
  const u32* filt = reinterpret_cast<const u32*>(qtab + SIM_QT);
  const uint8_t* tags = reinterpret_cast<const uint8_t*>(qtab + SIM_QT + SIM_QT * (SIM_QF_WORDS / 4));
  for (size_t i = blockIdx.x * (size_t)BLOCK + threadIdx.x; i < n3; i += (size_t)gridDim.x * BLOCK) {
    if (i < n0) {
      uint4 t = qtab[i >> 1];
      u64 w = (i & 1) ? ((u64)t.z | ((u64)t.w << 32)) : ((u64)t.x | ((u64)t.y << 32));
      acc += dig(w, i);
    } else if (i < n1) {
      acc += dig((u64)qbits[i - n0], i);
    } else if (i < n2) {
      acc += dig((u64)filt[i - n1], i);
    } else {
      acc += dig((u64)tags[i - n2], i);
    }
  }
  block_sum_add(acc, out);
}
// one filter entry, by value (the host keeps the table and is its only writer)
struct QFiltEnt { uint4 w[SIM_QF_WORDS / 4]; };
__global__ void qfilt_set_kernel(uint4* dst, QFiltEnt e) {
  if (threadIdx.x < SIM_QF_WORDS / 4 && !blockIdx.x) dst[threadIdx.x] = e.w[threadIdx.x];
}
__global__ void query_count_kernel(const u32* bits, size_t words, u64* out /*[2]*/) {
  u64 a = 0, r = 0;
  for (size_t i = blockIdx.x * (size_t)BLOCK + threadIdx.x; i < words; i += (size_t)gridDim.x * BLOCK) {
    a += __popc(bits[i]);
    r += __popc(bits[words + i]);
  }
  block_sum_add(a, out);
  block_sum_add(r, out + 1);
}
// aux digest: slot map, then the liveness bitmap (bits past N masked)
__global__ void digest_aux(const u32* slot_of, const u32* upmap, u32 N, u64* out) {
  u64 acc = 0;
  size_t nw = ((size_t)N + 31) / 32;
  for (size_t i = blockIdx.x * (size_t)BLOCK + threadIdx.x; i < (size_t)N + nw; i += (size_t)gridDim.x * BLOCK) {
    if (i < N) {
      acc += dig((u64)slot_of[i], i);
    } else {
      size_t j = i - N;
      u32 w = upmap[j];
      if (j == N / 32 && (N & 31)) w &= (1u << (N & 31)) - 1u;
      acc += dig((u64)w, i);
    }
  }
  block_sum_add(acc, out);
}
__global__ void digest_rows_queue(Dev d, u64* out_rows, u64* out_queue) {
  u64 ar = 0, aq = 0;
  for (size_t l = blockIdx.x * (size_t)BLOCK + threadIdx.x; l < d.Nl; l += (size_t)gridDim.x * BLOCK) {
    u64 w[ROW_W];
    canon_row(d, l, w);
    for (int i = 0; i < ROW_W; ++i) ar += dig(w[i], l * ROW_W + i);
    u32 cnt = __popc(d.R2[l].z >> 16);
    for (u32 q = 0; q < SIM_Q; ++q) {
      uint4 e = canon_qrec(d, l, q, cnt);
      aq += dig((u64)e.x | ((u64)e.y << 32), (l * SIM_Q + q) * 2);
      aq += dig((u64)e.z | ((u64)e.w << 32), (l * SIM_Q + q) * 2 + 1);
    }
  }
  block_sum_add(ar, out_rows);
  block_sum_add(aq, out_queue);
}

__global__ void members_kernel(Dev d, const uint4* base, u32 obs_l, uint8_t* st, u64* lt) {
  for (size_t s = blockIdx.x * (size_t)blockDim.x + threadIdx.x; s < d.N; s += (size_t)gridDim.x * blockDim.x) {
    u32 a = d.slot_of[s];
    uint4 e = a == NOSLOT ? base[s * 2] : d.view[(size_t)a * d.Nl + obs_l];
    bool known = e.w & SIM_VB_KNOWN;
    st[s] = known ? (uint8_t)SIM_VB_STATUS(e.w) : (uint8_t)SIM_STATUS_NONE;
    lt[s] = known ? E_LTIME(e) : 0;
  }
}
__global__ void convergence_kernel(Dev d, const uint4* base, u32 kind, u32 key, u64 ltime, u64* out /*[2]*/) {
  u64 seen = 0, upc = 0;
  for (size_t l = blockIdx.x * (size_t)BLOCK + threadIdx.x; l < d.Nl; l += (size_t)gridDim.x * BLOCK) {
    if (!(d.R1[l].z & SIM_RF_UP)) continue;
    upc++;
    if (kind == SIM_K_JOIN || kind == SIM_K_LEAVE) {
      u32 a = d.slot_of[key];
      uint4 e = a == NOSLOT ? base[(size_t)key * 2] : d.view[(size_t)a * d.Nl + l];
      seen += ((e.w & SIM_VB_KNOWN) && E_LTIME(e) >= ltime);
    } else {
      const uint4* ring = kind == SIM_K_EVENT ? d.ering : d.qring;
      u32 B = kind == SIM_K_EVENT ? d.Bev : d.Bq;
      const uint4* p = ring + ((size_t)(ltime % B) * d.Nl + l);
      uint4 b0 = p[0], b1 = p[kind == SIM_K_EVENT ? d.etail : d.qtail];
      seen += (b0.z == key) | (b0.w == key) | (b1.x == key) | (b1.y == key) | (b1.z == key) | (b1.w == key);
    }
  }
  block_sum_add(seen, out);
  block_sum_add(upc, out + 1);
}
// the same for up to SIM_CONV_MAX rumours in one pass: out[0] = running nodes, out[1 + i] = those that have applied rumour i
struct ConvSet { u32 n; u32 kind[SIM_CONV_MAX], key[SIM_CONV_MAX]; u64 ltime[SIM_CONV_MAX]; };
__global__ void convergence_many_kernel(Dev d, const uint4* base, ConvSet cs, u64* out) {
  __shared__ u32 cnt[SIM_CONV_MAX + 1];
  for (u32 i = threadIdx.x; i <= SIM_CONV_MAX; i += BLOCK) cnt[i] = 0;
  __syncthreads();
  const size_t rounds = ((size_t)d.Nl + (size_t)gridDim.x * BLOCK - 1) / ((size_t)gridDim.x * BLOCK);
  for (size_t it = 0; it < rounds; ++it) {  // whole waves stay together: the ballots below need every lane
    size_t l = (it * gridDim.x + blockIdx.x) * (size_t)BLOCK + threadIdx.x;
    bool up = l < d.Nl && (d.R1[l].z & SIM_RF_UP);
    u64 m = __ballot(up);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(&cnt[0], (u32)__popcll(m));
    for (u32 i = 0; i < cs.n; ++i) {
      bool hit = false;
      if (up) {
        u32 kind = cs.kind[i], key = cs.key[i];
        u64 ltime = cs.ltime[i];
        if (kind == SIM_K_JOIN || kind == SIM_K_LEAVE) {
          u32 a = d.slot_of[key];
          uint4 e = a == NOSLOT ? base[(size_t)key * 2] : d.view[(size_t)a * d.Nl + l];
          hit = (e.w & SIM_VB_KNOWN) && E_LTIME(e) >= ltime;
        } else {
          const uint4* ring = kind == SIM_K_EVENT ? d.ering : d.qring;
          u32 B = kind == SIM_K_EVENT ? d.Bev : d.Bq;
          const uint4* p = ring + ((size_t)(ltime % B) * d.Nl + l);
          uint4 b0 = p[0];
          hit = (b0.z == key) | (b0.w == key);
          if (!hit && b0.w) {  // the tail plane only when the head is full and does not hold the key
            uint4 b1 = p[kind == SIM_K_EVENT ? d.etail : d.qtail];
            hit = (b1.x == key) | (b1.y == key) | (b1.z == key) | (b1.w == key);
          }
        }
      }
      u64 hm = __ballot(hit);
      if ((threadIdx.x & 63) == 0 && hm) atomicAdd(&cnt[1 + i], (u32)__popcll(hm));
    }
  }
  __syncthreads();
  for (u32 i = threadIdx.x; i <= cs.n; i += BLOCK)
    if (cnt[i]) atomicAdd((unsigned long long*)(out + i), (unsigned long long)cnt[i]);
}
__global__ void stats_kernel(Dev d, u32 l, sim_stats* o) {
  if (threadIdx.x || blockIdx.x) return;
  sim_stats s;
  memset(&s, 0, sizeof s);
  uint4 r0 = d.R0[l], r1 = d.R1[l], r2 = d.R2[l], r3 = d.R3[l];
  s.members = r1.w; s.failed = r2.x; s.left = r2.y;
  s.health_score = r3.z;
  s.member_time = (u64)r0.x | ((u64)r0.y << 32);
  s.event_time = (u64)r0.z | ((u64)r0.w << 32);
  s.query_time = (u64)r1.x | ((u64)r1.y << 32);
  u32 cnt = __popc(r2.z >> 16);
  for (u32 q = 0; q < cnt; ++q) {
    uint4 kq = d.qkeys[(size_t)(q >> 2) * d.Nl + l];
    u32 k = (q & 3) == 0 ? kq.x : (q & 3) == 1 ? kq.y : (q & 3) == 2 ? kq.z : kq.w;
    u32 cls = k >> 26;
    if (cls == 0) s.swim_queue++; else if (cls == 1) s.intent_queue++; else if (cls == 2) s.query_queue++; else s.event_queue++;
  }
  s.serf_state = SIM_RF_STATE(r1.z); s.up = r1.z & SIM_RF_UP; s.incarnation = r3.x;
  s.queue_overflow = r2.w;
  *o = s;
}
// cluster-wide load figures: out[0] up, [1..4] queue entries by class, [5] overflow, [6] records in flight,
// [7] failed, [8] left, [9] deepest queue (max)
// Two levels: every workgroup leaves its ten partial figures in part[blockIdx.x][10] (no atomics: 4096 workgroups adding to
// ten words cost 0.4 ms), cluster_stats_fold adds them up.
#define CSTAT_WG 1024
__global__ void cluster_stats_kernel(Dev d, const uint4* inbox, u64* part) {
  u64 a[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  u32 mx = 0;
  for (size_t l = blockIdx.x * (size_t)BLOCK + threadIdx.x; l < d.Nl; l += (size_t)gridDim.x * BLOCK) {
    uint4 r1 = d.R1[l], r2 = d.R2[l];
    a[0] += (r1.z & SIM_RF_UP) ? 1u : 0u;
    u32 cnt = __popc(r2.z >> 16);
    mx = max(mx, cnt);
    for (u32 q = 0; q < cnt; ++q) {
      uint4 kq = d.qkeys[(size_t)(q >> 2) * d.Nl + l];
      u32 k = (q & 3) == 0 ? kq.x : (q & 3) == 1 ? kq.y : (q & 3) == 2 ? kq.z : kq.w;
      a[1 + (k >> 26)] += 1;
    }
    a[5] += r2.w; a[7] += r2.x; a[8] += r2.y;
    if (inbox)
      for (u32 k = 0; k < d.fp; ++k)
        for (u32 p = 0; p < SIM_P; ++p) a[6] += SIM_META_KIND(pk_word(inbox[((size_t)k * d.Nl + l) * PK_U4 + 2], p)) != SIM_K_EMPTY;
  }
  __shared__ u64 sm[10][BLOCK / 64];
  for (int i = 0; i < 9; ++i) {
    u64 v = a[i];
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) sm[i][threadIdx.x >> 6] = v;
  }
  for (int o = 32; o >= 1; o >>= 1) mx = max(mx, (u32)__shfl_down((int)mx, o, 64));
  if ((threadIdx.x & 63) == 0) sm[9][threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x < 10) {
    u64 t = 0;
    for (int i = 0; i < BLOCK / 64; ++i) t = threadIdx.x == 9 ? max(t, sm[9][i]) : t + sm[threadIdx.x][i];
    part[(size_t)blockIdx.x * 10 + threadIdx.x] = t;
  }
}
__global__ void cluster_stats_fold(const u64* part, u32 nwg, u64* out) {  // one workgroup of 64 x 10 threads: figure = threadIdx.y
  const u32 i = threadIdx.y;
  u64 t = 0;
  for (u32 w = threadIdx.x; w < nwg; w += 64u) t = i == 9 ? max(t, part[(size_t)w * 10 + i]) : t + part[(size_t)w * 10 + i];
  for (int o = 32; o >= 1; o >>= 1) {
    const u64 y = __shfl_down(t, o, 64);
    t = i == 9 ? max(t, y) : t + y;
  }
  if (threadIdx.x == 0) out[i] = t;
}
// single-word / single-entry updates of device tables with the value passed by value (no host buffer to outlive)
__global__ void poke_u32(u32* p, u32 v) { if (!threadIdx.x && !blockIdx.x) *p = v; }
__global__ void poke_base(uint4* base, u32 x, uint4 e0, uint4 e1) { if (!threadIdx.x && !blockIdx.x) { base[(size_t)x * 2] = e0; base[(size_t)x * 2 + 1] = e1; } }
// ---- view-slot recycling scans (SIMSPEC §2.6; oracle recycle_scan) ----
// subjects that a running node still has a queued record or a suspicion timer about, or that a packet in flight mentions
// (packets in flight: sharded, the receive buffer `inbox`; local mode, inbox == null and the cells the senders kept —
// every cell a map word points at is delivered to somebody, so the set of subjects is the same)
__global__ void recycle_refd_kernel(Dev d, const uint4* inbox, u32 cur, uint8_t* refd, u32* first_up) {
  u32 lo = 0xFFFFFFFFu;
  for (size_t l = blockIdx.x * (size_t)BLOCK + threadIdx.x; l < d.Nl; l += (size_t)gridDim.x * BLOCK) {
    if (!inbox && !d.sharded) {
      u32 jw = d.omap[cur][l];
      for (u32 k = 0; k < d.f; ++k) {
        u32 jb = (jw >> (8u * k)) & 0xFFu;  // first page << 2 | pages - 1
        bool again = jb == 0xFFu;
        for (u32 q = 0; q < k; ++q) again |= ((jw >> (8u * q)) & 0xFFu) == jb;
        if (again) continue;
        for (u32 pg = 0; pg <= (jb & 3u); ++pg) {
          const uint4* cellp = d.obox[cur] + ((size_t)((jb >> 2) + pg) * d.Nl + l) * PK_U4;
          uint4 ck = cellp[0], ch = cellp[2];
          for (u32 p = 0; p < SIM_P; ++p) {
            u32 key = pk_word(ck, p);
            if (member_kind(SIM_META_KIND(pk_word(ch, p))) && key < d.N) refd[key] = 1;
          }
        }
      }
    }
    if (inbox)
      for (u32 k = 0; k < d.fp; ++k)
        for (u32 p = 0; p < SIM_P; ++p) {
          const uint4* cellp = inbox + ((size_t)k * d.Nl + l) * PK_U4;
          u32 key = pk_word(cellp[0], p);
          if (member_kind(SIM_META_KIND(pk_word(cellp[2], p))) && key < d.N) refd[key] = 1;
        }
    uint4 r1 = d.R1[l];
    if (!(r1.z & SIM_RF_UP)) continue;
    lo = min(lo, (u32)l);
    u32 cnt = __popc(d.R2[l].z >> 16);
    for (u32 q = 0; q < cnt; ++q) {
      uint4 kq = d.qkeys[(size_t)(q >> 2) * d.Nl + l];
      u32 k = (q & 3) == 0 ? kq.x : (q & 3) == 1 ? kq.y : (q & 3) == 2 ? kq.z : kq.w;
      uint4 pay = d.qpay[(size_t)(k & 15u) * d.Nl + l];
      if (member_kind(SIM_META_KIND(pay.y)) && pay.x < d.N) refd[pay.x] = 1;
    }
    const uint16_t* sp = reinterpret_cast<const uint16_t*>(&d.R4[2 * l]);
    for (u32 j = 0; j < SIM_S; ++j)
      if (sp[j] && d.subject_of[sp[j] - 1] != NOSLOT) refd[d.subject_of[sp[j] - 1]] = 1;
  }
  if (lo != 0xFFFFFFFFu) atomicMin(first_up, lo);
}
// candidate c = blockIdx.y: does every running node hold the head the first running node holds?
__global__ void recycle_view_kernel(Dev d, const u32* cand_slots, const u32* first_up, uint4* out_ref, u32* out_bad) {
  u32 c = blockIdx.y, a = cand_slots[c], l0 = *first_up;
  if (l0 == 0xFFFFFFFFu) return;
  uint4 ref = d.view[(size_t)a * d.Nl + l0];
  ref.w &= 0x7FFu;  // the stamp of a settled entry is dead data
  if (!blockIdx.x && !threadIdx.x) out_ref[c] = ref;
  bool bad = false;
  for (size_t l = blockIdx.x * (size_t)BLOCK + threadIdx.x; l < d.Nl; l += (size_t)gridDim.x * BLOCK) {
    if (!(d.R1[l].z & SIM_RF_UP)) continue;
    uint4 e = d.view[(size_t)a * d.Nl + l];
    e.w &= 0x7FFu;
    bad |= ne4(e, ref);
  }
  if (bad) out_bad[c] = 1;
}
// Local mode, off the hot path: Dev::obox / omap (packets kept at their senders) <-> the canonical receiver-indexed
// inbox[k][node].  `p` = the parameters of the tick the packets were sent in; `valid` = 0 at tick 0 (nothing in flight).
__global__ void materialize_kernel(Dev d, TickP p, u32 cur, u32 valid, uint4* out) {
  size_t n = (size_t)d.fp * d.Nl;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    u32 kk = (u32)(i / d.Nl), l = (u32)(i - (size_t)kk * d.Nl), k = kk / d.PG, pg = kk - k * d.PG;
    uint4 a = make_uint4(0, 0, 0, 0), b = a, c = a;
    if (d.rfan) {  // random fan-out: the canonical form is the oracle's — packets in the SENDER's cells, [slot][sender]
      const u32 jb = valid ? (d.obox[cur][(size_t)l * RF_CELL_U4 + 3u].x >> (8u * k)) & 0xFFu : 0xFFu;  // the map word sits in cell 0
      if (jb != 0xFFu && pg <= (jb & 3u)) {
        const uint4* cp = d.obox[cur] + ((size_t)((jb >> 2) + pg) * d.Nl + l) * RF_CELL_U4;
        a = cp[0]; b = cp[1]; c = cp[2];
      }
    } else if (valid && k < p.feff) {
      u32 h = l / p.M, t = l - h * p.M, g, ll;
      fan_source_g(p, PICK4(p.off, k), PICK4(p.rot, k), PICK4(p.rho, k), h, t, k, g, ll);
      u32 s = g * p.M + ll;
      u32 jb = (d.omap[cur][s] >> (8u * k)) & 0xFFu;  // first page << 2 | pages - 1
      if (jb != 0xFFu && pg <= (jb & 3u)) {
        const uint4* cp = d.obox[cur] + ((size_t)((jb >> 2) + pg) * d.Nl + s) * PK_U4;
        a = cp[0]; b = cp[1]; c = cp[2];
      }
    }
    out[i * PK_U4] = a; out[i * PK_U4 + 1] = b; out[i * PK_U4 + 2] = c;
  }
}
__global__ void unmaterialize_kernel(Dev d, TickP p, u32 cur, u32 valid, const uint4* in) {
  for (size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x; s < d.Nl; s += (size_t)gridDim.x * blockDim.x) {
    if (d.rfan) {  // random fan-out: the image is sender-indexed already; every slot gets its own pages k * PG ..., the map word goes into cell 0
      u32 jw = 0xFFFFFFFFu;
      for (u32 k = 0; valid && k < d.f; ++k) {
        u32 np = 0;
        for (u32 pg = 0; pg < d.PG; ++pg) {
          const uint4* cp = in + ((size_t)(k * d.PG + pg) * d.Nl + s) * PK_U4;
          uint4 a = cp[0], b = cp[1], c = cp[2];
          if (((c.x | c.y | c.z | c.w) & 0xF0u) == 0) break;
          uint4* op = d.obox[cur] + ((size_t)(k * d.PG + pg) * d.Nl + s) * RF_CELL_U4;
          op[0] = a; op[1] = b; op[2] = c;
          np = pg + 1;
        }
        if (np) jw = (jw & ~(0xFFu << (8u * k))) | ((((k * d.PG) << 2) | (np - 1u)) << (8u * k));
      }
      d.obox[cur][(size_t)s * RF_CELL_U4 + 3u] = make_uint4(jw, 0u, 0u, 0u);
      continue;
    }
    u32 jw = 0xFFFFFFFFu;
    if (valid) {
      u32 g = (u32)s / p.M, ll = (u32)s - g * p.M;
      for (u32 k = 0; k < p.feff; ++k) {  // every slot gets its own pages k * PG ...: an image does not say which packets were the same
        u32 h, t, np = 0;
        fan_target_g(p, g, ll, k, h, t);
        for (u32 pg = 0; pg < d.PG; ++pg) {
          const uint4* cp = in + ((size_t)(k * d.PG + pg) * d.Nl + (size_t)h * p.M + t) * PK_U4;
          uint4 a = cp[0], b = cp[1], c = cp[2];
          if (((c.x | c.y | c.z | c.w) & 0xF0u) == 0) break;  // no record in it: pages fill up in order
          uint4* op = d.obox[cur] + ((size_t)(k * d.PG + pg) * d.Nl + s) * PK_U4;
          op[0] = a; op[1] = b; op[2] = c;
          np = pg + 1;
        }
        if (np) jw = (jw & ~(0xFFu << (8u * k))) | ((((k * d.PG) << 2) | (np - 1u)) << (8u * k));
      }
    }
    d.omap[cur][s] = jw;
  }
}
__global__ void set_flag_bits(uint4* R1, u32 l, u32 bits) {
  if (threadIdx.x == 0 && blockIdx.x == 0) R1[l].z |= bits;
}

// ------------------------------------------------------------------------------------------------
// host side: the C ABI
// ------------------------------------------------------------------------------------------------
struct OpEnt {
  u64 tick;
  u32 op, node, a, b;
  u64 val;  // SIM_OP_DELIVER: the record's value (a = key, b = wire bits of meta); 0 otherwise
};

struct sim_handle {
  sim_config cfg;
  Dev d;
  u64 tick;
  u32 dense, n_slots;
  hipStream_t stream;
  std::vector<u32> slot_of, subject_of;
  std::vector<u32> walk;  // host copy of d.walk
  std::vector<u32> alloc_tick;  // [A] tick at which the slot was handed out
  u32 n_alloc;                  // slots in use
  u64 ops_dropped, slots_recycled;
  u64 events_lost;              // events the bounded device log dropped (counted when they are drained)
  u32 recycle_at;               // the tick whose recycling pass has already run
  u32 pp_done_at;               // the tick whose push-pull batch the sharded host has already run
  // the batch being driven by the sharded host: in-shard pairs, and the cross-shard pairs grouped by peer shard in
  // ascending pair order — r1: this shard owns the even node `a` (receives b in round 1, sends a in round 2); s1: owns `b`
  std::vector<u32> pp_local_a, pp_local_b, pp_r1, pp_s1;
  std::vector<u32> rc_a, rc_b;  // Reconnector: the reconnect attempts that run as push-pull pairs in THIS tick (global ids; initiator, target)
  u32* d_pp;                    // the four lists on the device, back to back
  std::vector<sim_view> base;
  uint4* d_base;  // [N][2]
  std::vector<OpEnt> ops;
  size_t op_cursor;
  u64* d_scratch;  // 16 x u64
  uint8_t* d_mst;
  u64* d_mlt;
  sim_stats* d_stats;
  std::vector<void*> allocs;
  TickP prev;
  bool bound;
  int device;
  u32 qt_cursor, q_timeout;  // running-query trackers (SIM_QT, round robin); query timeout in ticks
  std::vector<u32> qfilt;    // [SIM_QT][SIM_QF_WORDS] host copy of the query filters (the host is their only writer)
  // content of the user events the library was told in bytes (sim_deliver_message, sim_user_event_bytes): what
  // sim_peek_packet encodes for their keys
  std::unordered_map<u32, std::pair<serf::wire::Bytes, serf::wire::Bytes>> evreg;
  u32 profiling;  // 0 = off, n = HIP events around every n-th tick-kernel launch
  u64 prof_seq;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof;  // one event pair per tick-kernel launch
  // slot-less failed probes (SIMSPEC §2.7): device lists by tick parity, their heads copied to pinned host memory behind
  // every tick's launch; the list of tick t is read at the end of tick t + 1 — by then the copy has long landed, nobody
  // waits for a kernel — and replayed as operations of tick t + 2
  u32* sreq_buf[3];
  u32* sreq_host[3];       // pinned, written by the kernel itself: the first SREQ_HEAD pairs, unused ones 0xFFFFFFFF
  hipEvent_t sreq_ev[3];
  hipEvent_t sreq_wait[3]; // what marks "tick t's kernel has finished": sreq_ev (recorded behind the launch, or riding on the dispatch
                           // itself as its stop event) or the stop event of the tick's timing pair; null: the stream was synchronised since
  bool sreq_on_dispatch;   // this tick's launch carried its completion event: sim_step_end records nothing
  u64 sreq_tick[3];        // the tick whose list sits in the buffer (~0: none / consumed)
  u32 pp_step;  // push-pull batches: every pp_step ticks one of PP_GROUPS pair classes synchronises (0 = off)
  TickP cur_tp;            // parameters of the tick between sim_step_begin and sim_step_end
  bool in_tick, tick_timed, tick_bracket;
  hipEvent_t tick_ev0;
  uint4* rbuf[2];          // sharded: packets sent during tick t are received into rbuf[t & 1]
  // local mode: the packets in flight in their canonical receiver-indexed form inbox[k][node] (what the oracle keeps,
  // what dumps, digests and images hold), produced from Dev::obox on demand; mat_tick = the tick it was made for
  uint4* inbox_mat;
  u64 mat_tick;
  // SIM_CF_RANDOM_FANOUT: scratch of the per-tick graph build (rf_* kernels)
  // the round's all-to-all over RCCL, issued by the library (sim_exchange_*): the communicator, a stream of its own for the
  // collectives (it waits for one chunk's launch, the handle's stream waits for all of a round's exchanges before the next
  // tick reads them), the events that carry those two orderings
  ncclComm_t xcomm;
  hipStream_t xstream;
  hipEvent_t xev_go, xev_done;
  u32 xworld;
  bool xpending;  // exchanges issued since the handle's stream last waited for them
  u32 *rf_gcur[2];     // bucket fill counters (two: a build zeroes the next one's)
  void *rf_ovf[2], *rf_l1;  // overflow lists (two, likewise) and the buckets' regions: u32 entries when a pair id and a target's offset fit, u64 otherwise
  bool rf_wide;        // ... u64
  u32 rf_par;                          // which of the two this build uses
  RfP rfp;  // the parameters that do not change from tick to tick
  // The graph of tick s is a function of (seed, s): it is built on a stream of its own, TWO ticks ahead — enqueued when tick
  // s - 1 begins, read by tick s + 1 — so that no tick ever waits for a build (one tick ahead, the build ran in the tail of
  // the tick kernel and the next tick waited 25 us for rf_rows).  rf_rcsr[s % 3] / rf_rsrc[s % 3] = the rows of the packets SENT
  // during tick s: while tick t reads buffer (t - 1) % 3 the builds of t and t + 1 may still be writing the other two.
  // rf_q[i] = the tick whose graph buffer i holds or is getting (~0: none), rf_done[i] marks its build.
  u32* rf_rcsr[3];
  u32* rf_rsrc[3];
  // ... on a shard (SIM_XCHG_PACKED): the graph of tick s is the SENDING side's — rf_rsrc[s % 3] = the shard's own (target, sender,
  // slot) triples sorted, as pair ids; rf_cntb / rf_btot / rf_xoff [s % 3] = count bytes per target of every destination, bucket
  // totals, where every destination's pairs start — needed when tick s has computed (the pack).  The receiving side's rows
  // (rx_rcsr / rx_rsrc) are made at the start of a tick from the slabs the round's exchange delivered.
  uint8_t* rf_cntb[3];
  u32* rf_btot[3];
  u32* rf_xoff[3];
  u32 *rx_rcsr, *rx_rsrc;
  u32 rx_cap;          // entries rx_rsrc holds (f * Nl, 12 sigma and some)
  RfxL rfx;
  u32* xflag;          // pinned host memory the rf / rfx kernels write to: a slab, a count byte or rsrc overflowed -> SIM_ERANGE
  hipStream_t rf_stream;
  hipEvent_t rf_done[3], rf_go[2];
  u64 rf_q[3];
  bool rf_sync;
};

#define HCHECK(x)                                                                        \
  do {                                                                                   \
    hipError_t e_ = (x);                                                                 \
    if (e_ != hipSuccess) {                                                              \
      fprintf(stderr, "serf_sim: %s failed: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
      return SIM_EDEVICE;                                                                \
    }                                                                                    \
  } while (0)

static u32 h_digits10(u32 n) {
  u32 d = 0;
  while (n) { ++d; n /= 10; }
  return d;
}
static int cfg_check(const sim_config* c) {
  if (!c || c->struct_size != sizeof(sim_config)) return SIM_EINVAL;
  if (c->n_nodes < 1 || c->vshards < 1 || c->n_nodes % c->vshards) return SIM_EINVAL;
  u32 M = c->n_nodes / c->vshards;
  if (c->vshards > 1 && (M % c->vshards || M <= SIM_MAX_FANOUT)) return SIM_EINVAL;
  if (c->shard_count != 1 && c->shard_count != c->vshards) return SIM_EINVAL;
  if (c->shard_rank >= c->shard_count) return SIM_EINVAL;
  if ((c->flags & SIM_CF_FORCE_SHARDED) && c->shard_count != c->vshards) return SIM_EINVAL;  // one rank of the N > 1 path: V == 1
  if (c->fanout < 1 || c->fanout > SIM_MAX_FANOUT) return SIM_EINVAL;
  if (c->chunks > 1 && ((M / c->vshards) % c->chunks || (M / c->vshards) / c->chunks < 1)) return SIM_EINVAL;
  // memberlist's literal kRandomNodes (variable in-degree, an explicit CSR per tick): one chunk per tick (shards: SIM_XCHG_PACKED)
  if ((c->flags & SIM_CF_RANDOM_FANOUT) && c->chunks > 1) return SIM_EINVAL;
  if (c->event_ring < 1 || c->query_ring < 1) return SIM_EINVAL;
  if (c->pkt_records && (c->pkt_records % SIM_P || c->pkt_records > SIM_PKT_RECORDS_MAX)) return SIM_EINVAL;
  if (c->retransmit_mult * h_digits10(c->n_nodes) > 63u) return SIM_EINVAL;
  if (c->n_nodes > (1u << 24)) return SIM_EINVAL;  // SUSPECT / DEAD carry the accuser's id in 24 bits on the wire (sim_packet)
  if (c->probe_interval) {  // suspicion timers name view slots with 16 bits
    u32 A = (c->view_slots == 0 || c->view_slots >= c->n_nodes) ? c->n_nodes : c->view_slots;
    if (A > 65534u) return SIM_EINVAL;
  }
  return SIM_OK;
}
// Suspicion parameters in ticks (memberlist suspicion.go / util.go); the same
// arithmetic, in the same order, as the spec (DESIGN.md SIMSPEC §6).
static void swim_params(const sim_config* c, u32* swim, u32* k_out, u32 T[SIM_MAX_CONF]) {
  *swim = c->probe_interval > 0;
  u32 k = c->suspicion_mult >= 2 ? c->suspicion_mult - 2 : 0;
  if (k > SIM_MAX_CONF - 1) k = SIM_MAX_CONF - 1;
  if (c->n_nodes < 2 || c->n_nodes - 2 < k) k = 0;
  double scale = std::log10(c->n_nodes > 1 ? (double)c->n_nodes : 1.0);
  if (scale < 1.0) scale = 1.0;
  u64 mn = (u64)c->suspicion_mult * (u64)std::floor(scale * 1000.0) * c->probe_interval / 1000u;
  if (mn < 1) mn = 1;
  u64 mx = (u64)c->suspicion_max_mult * mn;
  if (mx < mn) mx = mn;
  for (u32 i = 0; i < SIM_MAX_CONF; ++i) {
    double t = (double)mn;
    if (k >= 1 && i <= k) {
      double frac = std::log((double)i + 1.0) / std::log((double)k + 1.0);
      t = std::floor((double)mx - frac * (double)(mx - mn));
      if (t < (double)mn) t = (double)mn;
    }
    T[i] = t > 2000000.0 ? 2000000u : (u32)t;
  }
  *k_out = k;
}

template <typename T>
static int dalloc(sim_handle* h, T** p, size_t n) {
  void* v = nullptr;
  if (hipMalloc(&v, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) return SIM_ENOMEM;
  h->allocs.push_back(v);
  *p = (T*)v;
  return SIM_OK;
}
static inline int grid_for(size_t n) { return (int)std::min<size_t>((n + BLOCK - 1) / BLOCK, 8192); }
#define EV_CAP (1u << 20)

extern "C" {

#ifdef TICK_ABLATE
int sim_debug_ablate(unsigned mask) { g_ablate = mask; return SIM_OK; }
static struct AblEnv { AblEnv() { if (const char* e = getenv("SERF_ABLATE")) g_ablate = (u32)strtoul(e, nullptr, 0); } } g_abl_env;
#endif
#ifdef TICK_TIMING
int sim_debug_timing(unsigned long long* out16, int reset) {
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_tt), 32 * 8) != hipSuccess) return SIM_EDEVICE;  // (the caller's buffer holds 32 words)
  if (reset) { unsigned long long z[32] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tt), z, 32 * 8); }
  return SIM_OK;
}
#endif
uint32_t sim_abi_version(void) { return SIM_ABI_VERSION; }
const char* sim_backend_name(void) { return "hip-gfx950"; }

int sim_destroy(sim_handle* h) {
  if (!h) return SIM_EINVAL;
  (void)hipStreamSynchronize(h->stream);
  if (h->xstream) (void)hipStreamSynchronize(h->xstream);
  if (h->xcomm) (void)ncclCommDestroy(h->xcomm);
  if (h->xstream) (void)hipStreamDestroy(h->xstream);
  if (h->xev_go) (void)hipEventDestroy(h->xev_go);
  if (h->xev_done) (void)hipEventDestroy(h->xev_done);
  if (h->rf_stream) { (void)hipStreamSynchronize(h->rf_stream); (void)hipStreamDestroy(h->rf_stream); }
  for (int i = 0; i < 3; ++i) if (h->rf_done[i]) (void)hipEventDestroy(h->rf_done[i]);
  for (int i = 0; i < 2; ++i) if (h->rf_go[i]) (void)hipEventDestroy(h->rf_go[i]);
  for (auto& pr : h->prof) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
  if (h->d_pp) (void)hipFree(h->d_pp);
  for (int i = 0; i < 3; ++i) {
    if (h->sreq_host[i]) (void)hipHostFree(h->sreq_host[i]);
    if (h->sreq_ev[i]) (void)hipEventDestroy(h->sreq_ev[i]);
  }
  if (h->xflag) (void)hipHostFree(h->xflag);
  for (void* p : h->allocs) (void)hipFree(p);
  delete h;
  return SIM_OK;
}

int sim_create(const sim_config* cfg, sim_handle** out) {
  int rc = cfg_check(cfg);
  if (rc) return rc;
  if (!out) return SIM_EINVAL;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
    fprintf(stderr, "serf_sim: no HIP device available (this library has no CPU fallback)\n");
    return SIM_EDEVICE;
  }
  sim_handle* h = new sim_handle();
  h->cfg = *cfg;
  h->tick = 0;
  h->stream = nullptr;
  h->op_cursor = 0;
  h->bound = false;
  h->n_alloc = 0; h->ops_dropped = h->slots_recycled = 0; h->recycle_at = 0xFFFFFFFFu;
  h->pp_done_at = 0xFFFFFFFFu; h->d_pp = nullptr;
  for (int i = 0; i < 3; ++i) { h->sreq_host[i] = nullptr; h->sreq_ev[i] = nullptr; h->sreq_buf[i] = nullptr; h->sreq_wait[i] = nullptr; }
  h->sreq_on_dispatch = false;
  h->in_tick = false;
  h->tick_timed = false;
  h->rbuf[0] = h->rbuf[1] = nullptr;
  h->xcomm = nullptr; h->xstream = nullptr; h->xev_go = h->xev_done = nullptr; h->xworld = 0; h->xpending = false;
  h->profiling = 0;
  h->prof_seq = 0;
  memset(&h->prev, 0, sizeof h->prev);
  (void)hipGetDevice(&h->device);
  Dev& d = h->d;
  memset(&d, 0, sizeof d);
  d.N = cfg->n_nodes; d.V = cfg->vshards; d.M = d.N / d.V;
  d.sharded = cfg->shard_count > 1 || (cfg->flags & SIM_CF_FORCE_SHARDED);  // (the flag: ONE shard run as a shard — the N > 1 path with a single rank)
  d.Nl = d.sharded ? d.M : d.N;
  d.shard0 = d.sharded ? cfg->shard_rank * d.M : 0;
  d.shard_rank = cfg->shard_rank;
  h->dense = (cfg->view_slots == 0 || cfg->view_slots >= d.N);
  d.A = h->dense ? d.N : cfg->view_slots;
  d.Bev = cfg->event_ring; d.Bq = cfg->query_ring; d.f = cfg->fanout;
  d.P = cfg->pkt_records ? cfg->pkt_records : SIM_P;
  d.PG = d.P / SIM_P;
  d.fp = d.f * d.PG;
  d.npend = d.f * d.P + SIM_S + 1u + ((cfg->flags & SIM_CF_RANDOM_FANOUT) ? SIM_RF_PEND_EXTRA : 0u);
  d.vtail = (size_t)d.A * d.Nl; d.etail = (size_t)d.Bev * d.Nl; d.qtail = (size_t)d.Bq * d.Nl;
  d.bev_mask = (d.Bev > 1 && !(d.Bev & (d.Bev - 1))) ? d.Bev - 1 : 0;
  d.bq_mask = (d.Bq > 1 && !(d.Bq & (d.Bq - 1))) ? d.Bq - 1 : 0;
  d.retransmit_mult = cfg->retransmit_mult;
  swim_params(cfg, &d.swim, &d.kconf, d.T);
  d.PI = cfg->probe_interval;
  d.ic = cfg->indirect_checks;
  d.reap_interval = cfg->reap_interval; d.reconnect_timeout = cfg->reconnect_timeout;
  d.reconnect_interval = cfg->probe_interval ? cfg->reconnect_interval : 0u;
  d.tombstone_timeout = cfg->tombstone_timeout; d.intent_timeout = cfg->intent_timeout;
  d.queue_check_interval = cfg->queue_check_interval; d.max_queue_depth = cfg->max_queue_depth;
  d.min_queue_depth = cfg->min_queue_depth;
  d.r3on = d.swim || d.reap_interval;
  d.loss_u32 = cfg->loss_u32;
  d.aw_probe = (cfg->flags & SIM_CF_AWARENESS_PROBE) ? 1u : 0u;
  d.tcp_fallback = (cfg->flags & SIM_CF_TCP_FALLBACK) ? 1u : 0u;
  d.nacks = (cfg->flags & SIM_CF_NACKS) ? 1u : 0u;
  d.gttd = cfg->gossip_to_the_dead;
  d.rfan = (cfg->flags & SIM_CF_RANDOM_FANOUT) ? 1u : 0u;
  h->qt_cursor = 0;
  h->q_timeout = 16u * h_digits10(cfg->n_nodes);  // query.rs:421-427, query_timeout_mult = 16 (options.rs:518)
  h->pp_step = 0;
  if (cfg->push_pull_interval) {  // memberlist pushPullScale: x (ceil(log2 N - 5) + 1) above 32 nodes
    u64 mult = 1;
    if (cfg->n_nodes > 32) mult = (u64)std::ceil(std::log2((double)cfg->n_nodes) - 5.0) + 1;
    u64 st = (u64)cfg->push_pull_interval * mult / PP_GROUPS;
    h->pp_step = st < 1 ? 1u : st > 0x7FFFFFFFu ? 0x7FFFFFFFu : (u32)st;
  }
  d.ev_cap = EV_CAP;
  size_t Nl = d.Nl, nup = ((size_t)d.N + 31) / 32;
#define DA(ptr, n)                                   \
  if ((rc = dalloc(h, &(ptr), (n))) != SIM_OK) {     \
    sim_destroy(h);                                  \
    return rc;                                       \
  }
  DA(d.R0, Nl) DA(d.R1, Nl) DA(d.R2, Nl) DA(d.R3, Nl) DA(d.R4, 2 * Nl) DA(d.R5, Nl)
  DA(d.qkeys, 4 * Nl) DA(d.qpay, (size_t)SIM_Q * Nl) DA(d.pend, (size_t)d.npend * Nl)
  h->inbox_mat = nullptr;
  h->mat_tick = ~0ull;
  if (!d.sharded) {
    const size_t cu4 = d.rfan ? RF_CELL_U4 : PK_U4;  // (random fan-out: 64-byte cells, the map word inside cell 0)
    DA(d.obox[0], (size_t)d.fp * Nl * cu4) DA(d.obox[1], (size_t)d.fp * Nl * cu4)
    if (!d.rfan) { DA(d.omap[0], Nl) DA(d.omap[1], Nl) }
    DA(h->inbox_mat, (size_t)d.fp * Nl * PK_U4)
  } else if (d.rfan) {  // a shard with the random fan-out: the packets stay in their senders' cells here too (one buffer: nobody
    // reads the cells of the tick before — the receivers read the slabs the exchange delivered)
    DA(d.obox[0], (size_t)d.fp * Nl * RF_CELL_U4)
    d.obox[1] = d.obox[0];
    DA(h->inbox_mat, (size_t)d.fp * Nl * PK_U4)
  }
  DA(d.view, (size_t)d.A * Nl * 2)
  DA(d.ering, (size_t)d.Bev * Nl * 2)
  DA(d.qring, (size_t)d.Bq * Nl * 2)
  DA(d.slot_of, d.N) DA(d.subject_of, d.A) DA(d.walk, d.A) DA(d.upmap, nup) DA(d.nullcell, 4) DA(d.skipmask, (d.gttd || d.rfan) ? Nl : 1) DA(h->sreq_buf[0], 1 + 2 * SIM_SUSPECT_REQ_MAX) DA(h->sreq_buf[1], 1 + 2 * SIM_SUSPECT_REQ_MAX) DA(h->sreq_buf[2], 1 + 2 * SIM_SUSPECT_REQ_MAX)
  DA(d.qtab, QTAB_U4(d.N)) DA(d.qbits, (size_t)SIM_QT * 2 * nup)
  DA(d.events, (size_t)EV_CAP) DA(d.ev_count, 1)
  DA(h->d_base, (size_t)d.N * 2)
  DA(h->d_scratch, 16)
  DA(h->d_mst, d.N)
  DA(h->d_mlt, d.N)
  DA(h->d_stats, 1)
  d.rcsr = d.rsrc = nullptr;
  for (int i = 0; i < 3; ++i) { h->rf_rcsr[i] = h->rf_rsrc[i] = nullptr; h->rf_cntb[i] = nullptr; h->rf_btot[i] = h->rf_xoff[i] = nullptr; }
  h->rx_rcsr = h->rx_rsrc = nullptr; h->xflag = nullptr;
  memset(&h->rfx, 0, sizeof h->rfx);
  h->rf_stream = nullptr; h->rf_go[0] = h->rf_go[1] = nullptr;
  for (int i = 0; i < 3; ++i) { h->rf_done[i] = nullptr; h->rf_q[i] = ~0ull; }
  h->rf_gcur[0] = h->rf_gcur[1] = nullptr; h->rf_ovf[0] = h->rf_ovf[1] = h->rf_l1 = nullptr; h->rf_par = 0; h->rf_wide = false;
  memset(&h->rfp, 0, sizeof h->rfp);
  if (d.rfan) {  // the fan-out graph as a CSR, rebuilt every tick (SIM_CF_RANDOM_FANOUT)
    RfP& r = h->rfp;
    r.N = d.N; r.Nl = d.Nl; r.shard0 = d.shard0; r.f = d.f;
    r.Ns = d.Nl;  // every handle sorts the pairs of its OWN senders: by (local) target — or, a shard, by target anywhere in the cluster
    r.V = d.sharded ? d.V : 1u; r.M = d.sharded ? d.M : d.N;
    if (d.sharded && d.V > 64u) { sim_destroy(h); return SIM_EINVAL; }
    // the sorted list: every pair of the handle's senders, f * Nl at most.  The rows a shard RECEIVES (rx_rsrc): f * Nl on average
    const size_t np = (size_t)d.f * Nl, nrx = (size_t)d.f * Nl + (size_t)(12.0 * std::sqrt((double)d.f * Nl)) + 4096u;
    r.rcap = (u32)np;
    // level-1 buckets of 2^LB targets.  A scattered entry carries the pair id (PB bits) and the target's offset in its bucket
    // (LB bits) — rf_rows does not draw again —; 32-bit entries when both fit (1 Mi nodes: 22 + 10), 64-bit ones otherwise.
    // A shard's buckets are per destination (V * ceil(M / 2^LB) of them): narrow entries while that keeps rf_scatter's three
    // tables in 48 KiB of LDS, else wide ones and about 2048 buckets
    r.PB = 1;
    while ((1ull << r.PB) < 4ull * r.Ns) r.PB++;
    auto nb_of = [&](u32 lb) { return (size_t)r.V * (((size_t)r.M + (1u << lb) - 1u) >> lb); };
    r.LB = std::min<u32>(11u, 32u - std::min<u32>(r.PB, 24u));
    if (const char* e = getenv("SERF_RF_LB")) r.LB = std::min<u32>(RF_LB_MAX, std::max<u32>(8u, (u32)strtoul(e, nullptr, 0)));
    while (nb_of(r.LB) > (r.V > 1u && r.PB + r.LB > 32u ? 2048u : 4096u) && r.LB < RF_LB_MAX) r.LB++;  // rf_scatter's three tables: 48 KiB of LDS next to its staging area
    if (nb_of(r.LB) > 4096u) { sim_destroy(h); return SIM_EINVAL; }
    h->rf_wide = r.PB + r.LB > 32u || getenv("SERF_RF_WIDE") != nullptr;
    r.NBh = (u32)(nb_of(r.LB) / r.V);
    r.NB = r.V * r.NBh;
    r.NWG = (u32)(((size_t)r.Ns + (h->rf_wide ? RfSpw<u64>::v : RfSpw<u32>::v) - 1u) / (h->rf_wide ? RfSpw<u64>::v : RfSpw<u32>::v));
    // pairs of a bucket: f per row on one handle; a shard's f * M pairs spread over all V * NBh buckets
    const double mean = (double)d.f * (double)(1u << r.LB) / (double)r.V;
    r.cap = std::min<u32>(std::max<u32>(((u32)(mean + 16.0 * std::sqrt(mean)) + 64u + RFR - 1u) / RFR * RFR, RFR), RF_EPT * RFR);  // 16 sigma and more of room
    if (r.V == 1u) r.cap = std::min<u32>(6u << r.LB, RF_EPT * RFR);
    if (const char* e = getenv("SERF_RF_CAP")) r.cap = std::max<u32>(RFR, std::min<u32>(r.cap, (u32)strtoul(e, nullptr, 0)) / RFR * RFR);  // tests: force rf_rows' slow path
    // a bucket's region of l1: the mean and 12 sigma; what does not fit goes onto the overflow list
    {
      r.bcap = (u32)(mean + 12.0 * std::sqrt(mean)) + 64u;
      if (const char* e = getenv("SERF_RF_BCAP")) r.bcap = std::max<u32>(1u, (u32)strtoul(e, nullptr, 0));  // tests: force the overflow list
      r.ocap = (u32)np;  // (every pair would fit: the list cannot run full)
    }
    if (h->rf_wide ? (hipFuncSetAttribute(reinterpret_cast<const void*>(rf_rows_kernel<u64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)rf_rows_lds(r)) != hipSuccess ||
                      hipFuncSetAttribute(reinterpret_cast<const void*>(rf_scatter_kernel<u64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)rf_scatter_lds<u64>(r)) != hipSuccess)
                   : (hipFuncSetAttribute(reinterpret_cast<const void*>(rf_rows_kernel<u32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)rf_rows_lds(r)) != hipSuccess ||
                      hipFuncSetAttribute(reinterpret_cast<const void*>(rf_scatter_kernel<u32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)rf_scatter_lds<u32>(r)) != hipSuccess)) { sim_destroy(h); return SIM_EDEVICE; }
    if (hipHostMalloc((void**)&h->xflag, 64) != hipSuccess) { sim_destroy(h); return SIM_ENOMEM; }
    *h->xflag = 0;
    if (d.sharded) {
      h->rfx = rfx_layout(d.V, d.M, r.NBh, r.LB, d.PG, d.f);
      if ((size_t)d.V * h->rfx.slab_u >= ((size_t)1 << 30) ||
          hipFuncSetAttribute(reinterpret_cast<const void*>(rfx_index_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)((4u << r.LB) + 16u)) != hipSuccess) { sim_destroy(h); return SIM_EINVAL; }
      for (int i = 0; i < 3; ++i) { DA(h->rf_cntb[i], (size_t)d.N) DA(h->rf_btot[i], r.NB) DA(h->rf_xoff[i], 2 * (size_t)d.V + 2) }
      DA(h->rx_rcsr, Nl + 1) DA(h->rx_rsrc, nrx)
      if (hipMemset(h->rx_rcsr, 0, (Nl + 1) * 4) != hipSuccess) { sim_destroy(h); return SIM_EDEVICE; }  // tick 0 receives nothing
    }
    h->rf_sync = getenv("SERF_RF_SYNC") != nullptr;  // measurements: build on the tick's own stream, nothing overlaps
    for (int i = 0; i < 3; ++i) { DA(h->rf_rcsr[i], Nl + 1) DA(h->rf_rsrc[i], np) }
    h->rx_cap = (u32)nrx;
    {
      const size_t esz = h->rf_wide ? 2 : 1;  // (in u32 words)
      u32* p32 = nullptr;
      for (int i = 0; i < 2; ++i) { DA(h->rf_gcur[i], (size_t)r.NB * RF_GCS) DA(p32, (1 + 2 * (size_t)r.ocap) * esz) h->rf_ovf[i] = p32; }
      DA(p32, (size_t)r.NB * r.bcap * esz)
      h->rf_l1 = p32;
    }
    bool ok = true;
    for (int i = 0; i < 3; ++i) ok = ok && hipMemset(h->rf_rcsr[i], 0, (Nl + 1) * 4) == hipSuccess;  // tick 0 receives nothing
    for (int i = 0; i < 2; ++i) ok = ok && hipMemset(h->rf_gcur[i], 0, (size_t)r.NB * RF_GCS * 4) == hipSuccess && hipMemset(h->rf_ovf[i], 0, 8) == hipSuccess;
    ok = ok && hipStreamCreateWithFlags(&h->rf_stream, hipStreamNonBlocking) == hipSuccess &&
         hipEventCreateWithFlags(&h->rf_done[0], hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&h->rf_done[1], hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&h->rf_done[2], hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&h->rf_go[0], hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&h->rf_go[1], hipEventDisableTiming) == hipSuccess;
    if (!ok) { sim_destroy(h); return SIM_EDEVICE; }
  }
#undef DA
  bool joined = cfg->flags & SIM_CF_BASELINE_JOINED;
  hipStream_t s = h->stream;
  auto zero = [&](void* p, size_t bytes) { return hipMemsetAsync(p, 0, bytes, s); };
  HCHECK(zero(d.R2, Nl * 16)); HCHECK(zero(d.R3, Nl * 16)); HCHECK(zero(d.R4, Nl * 32)); HCHECK(zero(d.R5, Nl * 16));
  HCHECK(zero(d.qpay, (size_t)SIM_Q * Nl * 16));
  HCHECK(zero(d.ev_count, 4));
  HCHECK(zero(d.nullcell, 64));
  d.sreq = h->sreq_buf[0];
  for (int i = 0; i < 3; ++i) {
    HCHECK(zero(h->sreq_buf[i], 4));
    HCHECK(hipHostMalloc((void**)&h->sreq_host[i], 2 * SREQ_HEAD * 4));
    memset(h->sreq_host[i], 0xFF, 2 * SREQ_HEAD * 4);
    HCHECK(hipEventCreateWithFlags(&h->sreq_ev[i], hipEventDisableTiming));
    h->sreq_tick[i] = ~0ull;
  }
  HCHECK(zero(d.qtab, QTAB_U4(d.N) * 16)); HCHECK(zero(d.qbits, (size_t)SIM_QT * 2 * nup * 4));
  if (!d.sharded && !d.rfan) {  // nothing has been sent yet
    HCHECK(zero(d.obox[0], (size_t)d.fp * Nl * sizeof(sim_packet))); HCHECK(zero(d.obox[1], (size_t)d.fp * Nl * sizeof(sim_packet)));
    HCHECK(hipMemsetAsync(d.omap[0], 0xFF, Nl * 4, s)); HCHECK(hipMemsetAsync(d.omap[1], 0xFF, Nl * 4, s));
  }
  if (d.rfan) {  // (all 0xFF: every map word says "nothing sent")
    HCHECK(hipMemsetAsync(d.obox[0], 0xFF, (size_t)d.fp * Nl * RF_CELL_U4 * 16, s));
    if (d.obox[1] != d.obox[0]) HCHECK(hipMemsetAsync(d.obox[1], 0xFF, (size_t)d.fp * Nl * RF_CELL_U4 * 16, s));
  }
  HCHECK(zero(d.view, (size_t)d.A * Nl * 32));
  HCHECK(zero(d.ering, (size_t)d.Bev * Nl * 32));
  HCHECK(zero(d.qring, (size_t)d.Bq * Nl * 32));
  HCHECK(hipMemsetAsync(d.upmap, 0xFF, nup * 4, s));
  // base.rs:196-205: every clock starts at 1 (+ the own join at ltime 1 when pre-joined)
  fill_u4<<<grid_for(Nl), BLOCK, 0, s>>>(d.R0, Nl, make_uint4(joined ? 2 : 1, 0, 1, 0));
  fill_u4<<<grid_for(Nl), BLOCK, 0, s>>>(d.R1, Nl, make_uint4(1, 0, SIM_RF_UP | (SIM_SERF_ALIVE << 1), joined ? d.N : 1));
  fill_u4<<<grid_for(4 * Nl), BLOCK, 0, s>>>(d.qkeys, 4 * Nl, make_uint4(KEMPTY, KEMPTY, KEMPTY, KEMPTY));
  // slot map + baseline
  h->slot_of.assign(d.N, NOSLOT);
  h->subject_of.assign(d.A, NOSLOT);
  h->alloc_tick.assign(d.A, 0);
  h->qfilt.assign((size_t)SIM_QT * SIM_QF_WORDS, 0);
  sim_view b0;
  memset(&b0, 0, sizeof b0);
  if (joined) { b0.ltime = 1; b0.bits = 1u | (SIM_STATUS_ALIVE << 1); }
  h->base.assign(d.N, b0);
  uint4 e0 = make_uint4((u32)b0.ltime, (u32)(b0.ltime >> 32), b0.inc, b0.bits), e1 = make_uint4(0, 0, 0, 0);
  fill_base<<<grid_for(d.N), BLOCK, 0, s>>>(h->d_base, d.N, e0, e1);
  if (h->dense) {
    h->n_slots = h->n_alloc = d.N;
    h->walk.resize(d.N);
    for (u32 i = 0; i < d.N; ++i) h->slot_of[i] = h->subject_of[i] = h->walk[i] = i;
    fill_iota<<<grid_for(d.N), BLOCK, 0, s>>>(d.walk, d.N);
    if (joined) {
      size_t tot = (size_t)d.A * Nl;
      fill_view_col<<<grid_for(tot), BLOCK, 0, s>>>(d.view, d.vtail, tot, 0, e0, e1);
    } else {
      init_dense_self<<<grid_for(Nl), BLOCK, 0, s>>>(d);
    }
  } else {
    h->n_slots = 0;
  }
  HCHECK(hipMemcpyAsync(d.slot_of, h->slot_of.data(), (size_t)d.N * 4, hipMemcpyHostToDevice, s));
  HCHECK(hipMemcpyAsync(d.subject_of, h->subject_of.data(), (size_t)d.A * 4, hipMemcpyHostToDevice, s));
  HCHECK(hipStreamSynchronize(s));
  HCHECK(hipGetLastError());
  *out = h;
  return SIM_OK;
}

int sim_set_stream(sim_handle* h, void* st) {
  if (!h) return SIM_EINVAL;
  (void)hipStreamSynchronize(h->stream);
  h->stream = (hipStream_t)st;
  return SIM_OK;
}

static const uint4* cur_inbox(sim_handle* h);
static int sreq_take(sim_handle* h, u64 t, uint32_t* out, uint32_t cap_pairs, uint32_t* n_pairs);
static int inject_val(sim_handle* h, uint64_t tick, uint32_t op, uint32_t node, uint32_t a, uint32_t b, uint64_t val);
static void walk_upload(sim_handle* h) {  // h->walk -> d.walk (synchronous: the host vector changes again later)
  if (!h->walk.empty()) (void)hipMemcpy(h->d.walk, h->walk.data(), h->walk.size() * 4, hipMemcpyHostToDevice);
}
static int ensure_slot(sim_handle* h, u32 subject) {
  Dev& d = h->d;
  if (subject >= d.N) return SIM_EINVAL;
  if (h->slot_of[subject] != NOSLOT) return SIM_OK;
  u32 a = 0;
  while (a < d.A && h->subject_of[a] != NOSLOT) ++a;  // the lowest free slot
  if (a == d.A) return SIM_ENOSLOT;
  if (a >= h->n_slots) h->n_slots = a + 1;
  h->n_alloc++;
  h->slot_of[subject] = a;
  h->subject_of[a] = subject;
  h->alloc_tick[a] = (u32)h->tick;
  const sim_view& b = h->base[subject];
  uint4 e0 = make_uint4((u32)b.ltime, (u32)(b.ltime >> 32), b.inc, b.bits);
  uint4 e1 = make_uint4(b.conf[0], b.conf[1], b.conf[2], b.conf[3]);
  u32 pos = (u32)h->walk.size();
  while (pos > 0 && h->subject_of[h->walk[pos - 1]] > subject) --pos;
  slot_alloc_kernel<<<grid_for(d.Nl), BLOCK, 0, h->stream>>>(d.view, d.vtail, d.Nl, a, e0, e1, d.walk, (u32)h->walk.size(), pos,
                                                             d.slot_of + subject, d.subject_of + a, subject);
  h->walk.insert(h->walk.begin() + pos, a);
  return SIM_OK;
}
// the subject an operation needs a view slot for (NOSLOT: none) — SIMSPEC §2.6
static u32 op_subject(const sim_handle* h, u32 op, u32 node, u32 a, u32 b) {
  switch (op) {
    case SIM_OP_LEAVE: case SIM_OP_JOIN: case SIM_OP_LEAVE_FINISH: return node;
    case SIM_OP_FORCE_LEAVE: return a;
    case SIM_OP_CRASH: case SIM_OP_REVIVE: case SIM_OP_SET_TAGS: return h->d.swim ? node : NOSLOT;
    case SIM_OP_SUSPECT: return h->d.swim ? a : NOSLOT;
    case SIM_OP_DELIVER: {  // a member record from outside is about subject `a`
      u32 kind = SIM_META_KIND(b);
      if (kind == SIM_K_JOIN || kind == SIM_K_LEAVE) return a;
      return (kind >= SIM_K_ALIVE && h->d.swim) ? a : NOSLOT;
    }
    default: return NOSLOT;
  }
}

// ---- view-slot recycling (SIMSPEC §2.6; oracle recycle_*) -------------------------------------------------------------
static bool recycle_is_due(const sim_handle* h) {
  u32 R = h->cfg.recycle_interval;
  return R && !h->dense && h->tick > 0 && h->tick % R == 0 && h->recycle_at != (u32)h->tick;
}
static u32 recycle_candidates(const sim_handle* h, sim_recycle_cand* out) {
  u32 n = 0, R = h->cfg.recycle_interval, now = (u32)h->tick;
  for (u32 a = 0; a < h->n_slots; ++a) {
    if (h->subject_of[a] == NOSLOT || h->alloc_tick[a] + R > now) continue;
    u32 pos = n < SIM_RECYCLE_BATCH ? n : SIM_RECYCLE_BATCH;
    while (pos > 0 && h->alloc_tick[out[pos - 1].slot] > h->alloc_tick[a]) --pos;
    if (pos >= SIM_RECYCLE_BATCH) continue;
    u32 last = n < SIM_RECYCLE_BATCH ? n : SIM_RECYCLE_BATCH - 1;
    for (u32 i = last; i > pos; --i) out[i] = out[i - 1];
    memset(&out[pos], 0, sizeof out[pos]);
    out[pos].slot = a;
    out[pos].subject = h->subject_of[a];
    if (n < SIM_RECYCLE_BATCH) ++n;
  }
  return n;
}
static int recycle_scan(sim_handle* h, sim_recycle_cand* c, u32 n) {
  if (!n) return SIM_OK;
  Dev& d = h->d;
  hipStream_t s = h->stream;
  uint8_t* refd = nullptr;
  u32* scr = nullptr;  // [0] first running node, [1 .. n] slots, [1 + 64 .. ] bad flags, then the refs (16-byte aligned)
  if (hipMalloc((void**)&refd, d.N) != hipSuccess) return SIM_ENOMEM;
  if (hipMalloc((void**)&scr, (4 + 2 * SIM_RECYCLE_BATCH) * 4 + SIM_RECYCLE_BATCH * 16) != hipSuccess) { (void)hipFree(refd); return SIM_ENOMEM; }
  u32 hs[4 + 2 * SIM_RECYCLE_BATCH];
  memset(hs, 0, sizeof hs);
  hs[0] = 0xFFFFFFFFu;
  for (u32 i = 0; i < n; ++i) hs[4 + i] = c[i].slot;
  uint4* d_ref = (uint4*)(scr + 4 + 2 * SIM_RECYCLE_BATCH);
  std::vector<uint8_t> hrefd(d.N);
  uint4 href[SIM_RECYCLE_BATCH];
  u32 hbad[SIM_RECYCLE_BATCH];
  hipError_t e = hipMemsetAsync(refd, 0, d.N, s);
  if (e == hipSuccess) e = hipMemcpyAsync(scr, hs, sizeof hs, hipMemcpyHostToDevice, s);
  if (e == hipSuccess) e = hipMemsetAsync(d_ref, 0, SIM_RECYCLE_BATCH * 16, s);
  if (e == hipSuccess) {
    recycle_refd_kernel<<<grid_for(d.Nl), BLOCK, 0, s>>>(d, (d.sharded || d.rfan) ? cur_inbox(h) : nullptr, (u32)(h->tick & 1), refd, scr);
    recycle_view_kernel<<<dim3((unsigned)std::min<size_t>((d.Nl + BLOCK - 1) / BLOCK, 1024), n), BLOCK, 0, s>>>(d, scr + 4, scr, d_ref, scr + 4 + SIM_RECYCLE_BATCH);
    e = hipMemcpyAsync(hs, scr, sizeof hs, hipMemcpyDeviceToHost, s);
  }
  if (e == hipSuccess) e = hipMemcpyAsync(href, d_ref, sizeof href, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipMemcpyAsync(hrefd.data(), refd, d.N, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  (void)hipFree(refd);
  (void)hipFree(scr);
  HCHECK(e);
  memcpy(hbad, hs + 4 + SIM_RECYCLE_BATCH, sizeof hbad);
  for (u32 i = 0; i < n; ++i) {
    c[i].flags = hrefd[c[i].subject] ? 1u : 0u;
    if (hs[0] == 0xFFFFFFFFu) continue;  // no running node on this shard
    memset(&c[i].ref, 0, sizeof c[i].ref);
    c[i].ref.ltime = (u64)href[i].x | ((u64)href[i].y << 32);
    c[i].ref.inc = href[i].z;
    c[i].ref.bits = href[i].w;
    c[i].flags |= 2u;
    const sim_view& r = c[i].ref;
    // settled = forgotten altogether, or known + Alive for serf and for memberlist, nothing buffered or pending
    bool settled = (r.bits == 0 && r.ltime == 0 && r.inc == 0) ||
                   ((r.bits & SIM_VB_KNOWN) && SIM_VB_STATUS(r.bits) == SIM_STATUS_ALIVE && SIM_VB_SWIM(r.bits) == SIM_SWIM_ALIVE &&
                    !SIM_VB_INTENT(r.bits) && !SIM_VB_NCONF(r.bits));
    if (!settled || hbad[i]) c[i].flags |= 1u;
  }
  return SIM_OK;
}
static int recycle_apply(sim_handle* h, const sim_recycle_cand* c, u32 n) {
  Dev& d = h->d;
  {  // a candidate that was examined and could not go goes to the back of the line (oracle recycle_apply)
    sim_recycle_cand ex[SIM_RECYCLE_BATCH];
    u32 ne = recycle_candidates(h, ex);
    for (u32 j = 0; j < ne; ++j) {
      bool agreed = false;
      for (u32 i = 0; i < n; ++i) agreed |= c[i].subject == ex[j].subject;
      if (!agreed) h->alloc_tick[ex[j].slot] = (u32)h->tick;
    }
  }
  for (u32 i = 0; i < n; ++i) {
    u32 x = c[i].subject;
    if (x >= d.N) return SIM_EINVAL;
    u32 a = h->slot_of[x];
    if (a == NOSLOT) continue;
    h->base[x] = c[i].ref;
    const sim_view& b = h->base[x];
    poke_base<<<1, 64, 0, h->stream>>>(h->d_base, x, make_uint4((u32)b.ltime, (u32)(b.ltime >> 32), b.inc, b.bits),
                                        make_uint4(b.conf[0], b.conf[1], b.conf[2], b.conf[3]));
    poke_u32<<<1, 64, 0, h->stream>>>(d.slot_of + x, NOSLOT);
    poke_u32<<<1, 64, 0, h->stream>>>(d.subject_of + a, NOSLOT);
    h->slot_of[x] = NOSLOT;
    h->subject_of[a] = NOSLOT;
    h->n_alloc--;
    h->slots_recycled++;
  }
  while (h->n_slots > 0 && h->subject_of[h->n_slots - 1] == NOSLOT) h->n_slots--;
  h->walk.clear();
  for (u32 x = 0; x < d.N; ++x)
    if (h->slot_of[x] != NOSLOT) h->walk.push_back(h->slot_of[x]);
  HCHECK(hipStreamSynchronize(h->stream));
  walk_upload(h);
  return SIM_OK;
}
static int recycle_local(sim_handle* h) {  // every shard is in this process: decide here
  sim_recycle_cand c[SIM_RECYCLE_BATCH];
  u32 n = recycle_candidates(h, c), m = 0;
  int rc = recycle_scan(h, c, n);
  if (rc) return rc;
  for (u32 i = 0; i < n; ++i)
    if ((c[i].flags & 3u) == 2u) c[m++] = c[i];
  rc = recycle_apply(h, c, m);
  h->recycle_at = (u32)h->tick;
  return rc;
}
// what an operation's arguments have to satisfy before it may reach ops_kernel (sim_inject, and every pending operation
// of an image being restored)
static int op_validate(u32 N, u32 op, u32 node, u32 a, u32 b) {
  if (node >= N) return SIM_EINVAL;
  switch (op) {
    case SIM_OP_USER_EVENT: if (!a) return SIM_EINVAL; if ((b & 0x7FFFFFFFu) > 9 * 1024) return SIM_ETOOBIG; break;  // bit 31: cc
    case SIM_OP_QUERY: if (!a) return SIM_EINVAL; break;
    case SIM_OP_LEAVE: case SIM_OP_JOIN: case SIM_OP_LEAVE_FINISH: case SIM_OP_CRASH: case SIM_OP_REVIVE: break;
    case SIM_OP_FORCE_LEAVE: if (a >= N) return SIM_EINVAL; break;
    case SIM_OP_SET_TAGS: if (a >= SIM_TAG_CLASSES) return SIM_EINVAL; break;
    case SIM_OP_QUERY_FILTER_ID: if (!a || b >= N) return SIM_EINVAL; break;
    case SIM_OP_QUERY_FILTER_TAGS: if (!a) return SIM_EINVAL; break;
    case SIM_OP_SUSPECT: case SIM_OP_RECONNECT: if (a >= N) return SIM_EINVAL; break;
    case SIM_OP_DELIVER: {
      u32 kind = SIM_META_KIND(b);
      if (kind < SIM_K_JOIN || kind > SIM_K_DEAD || (b & ~(SIM_META_WIRE_MASK | SIM_DELIVER_MUTE))) return SIM_EINVAL;
      if ((b & SIM_DELIVER_MUTE) && kind != SIM_K_JOIN && kind != SIM_K_LEAVE && kind != SIM_K_EVENT) return SIM_EINVAL;
      if (kind == SIM_K_EVENT || kind == SIM_K_QUERY) { if (!a) return SIM_EINVAL; }
      else if (a >= N) return SIM_EINVAL;
      break;
    }
    case SIM_OP_QRESP: if (!a || (b & 0xFFFFFFu) >= N || (b & 0x7F000000u)) return SIM_EINVAL; break;
    case SIM_OP_WITNESS: if (a > 2u) return SIM_EINVAL; break;
    default: return SIM_EINVAL;
  }
  return SIM_OK;
}
static int inject_val(sim_handle* h, uint64_t tick, uint32_t op, uint32_t node, uint32_t a, uint32_t b, uint64_t val) {
  if (!h) return SIM_EINVAL;
  if (tick < h->tick) tick = h->tick;
  if (op == SIM_OP_SUSPECT && (a & SREQ_RECONNECT)) { op = SIM_OP_RECONNECT; a &= ~SREQ_RECONNECT; }  // an entry of the request list, as it stands there
  int rc = op_validate(h->d.N, op, node, a, b);
  if (rc) return rc;
  // an operation that executes now gets its view slot now (and SIM_ENOSLOT if there is none); one scheduled for a
  // later tick gets it when it executes — and is dropped and counted if none is free then (SIMSPEC §2.6)
  // (a SIM_OP_SUSPECT always takes its slot when it executes: it is scheduled by the library / the sharded host, and a
  // full view must count it as dropped the same way in both)
  if (op != SIM_OP_SUSPECT && tick <= h->tick && op_subject(h, op, node, a, b) != NOSLOT) rc = ensure_slot(h, op_subject(h, op, node, a, b));
  if (rc) return rc;
  size_t pos = h->ops.size();
  h->ops.push_back(OpEnt{tick, op, node, a, b, val});
  while (pos > h->op_cursor && h->ops[pos - 1].tick > tick) { std::swap(h->ops[pos], h->ops[pos - 1]); --pos; }
  return SIM_OK;
}
int sim_inject(sim_handle* h, uint64_t tick, uint32_t op, uint32_t node, uint32_t a, uint32_t b) {
  if (op == SIM_OP_DELIVER || op == SIM_OP_QRESP || op == SIM_OP_WITNESS) return SIM_EINVAL;  // internal, with a value: sim_inject_record / sim_deliver_message
  return inject_val(h, tick, op, node, a, b, 0);
}
// ---- the byte boundary of the delegate (include/serf_sim.h; oracle: the same entry points with its own C codec) ----
int sim_inject_record(sim_handle* h, uint64_t tick, uint32_t node, const sim_record* rec) {
  if (!h || !rec) return SIM_EINVAL;
  return inject_val(h, tick, SIM_OP_DELIVER, node, rec->key, rec->meta & SIM_META_WIRE_MASK, rec->val);
}
int sim_user_event_bytes(sim_handle* h, uint32_t node, const uint8_t* name, size_t nlen, const uint8_t* payload, size_t plen, int cc) {
  if (!h || (nlen && !name) || (plen && !payload)) return SIM_EINVAL;
  if (nlen + plen > 512) return SIM_ETOOBIG;  // api.rs:246-262 user_event_size_limit
  namespace w = serf::wire;
  w::Bytes nm(name, name + nlen), pl(payload, payload + plen);
  u32 key = w::event_key(nm, pl);
  h->evreg.emplace(key, std::make_pair(nm, pl));  // the first content under a key stays
  return sim_user_event(h, node, key, (uint32_t)w::user_event_len(1, nm, pl, cc != 0), cc);
}
// `relayed`: the message is the inside of a Relay
static int deliver_one(sim_handle* h, uint32_t node, const uint8_t* buf, size_t len, size_t* consumed, bool relayed) {
  if (!h || !buf || !len || node >= h->d.N) return SIM_EINVAL;
  namespace w = serf::wire;
  try {
    w::Bytes in(buf, buf + len);
    if (in[0] == w::merge(w::WIRE_LEN, w::RELAY)) {
      // Relay (types/message.rs:431-470): `node` forwards the wrapped message to the node it names as it is (delegate.rs:262-313:
      // memberlist.send) — if it is running; a process that is down forwards nothing
      if (relayed) return SIM_EINVAL;
      auto [dest, off] = w::unwrap_relay(in);
      if (dest >= h->d.N || off >= len) return SIM_EINVAL;
      const u32 in_tag = buf[off] >> 3;
      if (in_tag == w::PUSH_PULL || in_tag == w::RELAY) return SIM_EINVAL;  // a push-pull does not travel as a user message; no nesting
      u32 word = 0;  // ground-truth liveness as of the end of the last tick (rare call: one word, one wait)
      HCHECK(hipMemcpyAsync(&word, h->d.upmap + (node >> 5), 4, hipMemcpyDeviceToHost, h->stream));
      HCHECK(hipStreamSynchronize(h->stream));
      size_t in_used = 0;
      int rc = SIM_OK;
      if ((word >> (node & 31)) & 1u) rc = deliver_one(h, dest, buf + off, len - off, &in_used, true);
      else {  // dropped with its relay: the inner frame is still walked so that the caller learns its length
        w::Bytes inner(buf + off, buf + len);
        (void)w::unframe(inner, in_used);
      }
      if (rc == SIM_OK && consumed) *consumed = off + in_used;
      return rc;
    }
    size_t used = 0;
    auto [tag, body] = w::unframe(in, used);
    sim_record rec;
    memset(&rec, 0, sizeof rec);
    int rc = SIM_OK;
    if (tag == w::QUERY_RESPONSE) {  // -> SIM_OP_QRESP at the origin
      w::QueryResponse m = w::decode_query_response(body);
      if (m.from_node >= h->d.N || !m.id) return SIM_EINVAL;
      rc = inject_val(h, h->tick, SIM_OP_QRESP, node, m.id, m.from_node | ((m.flags & 1u) ? 0x80000000u : 0u), 0);
      if (rc == SIM_OK && consumed) *consumed = used;
      return rc;
    }
    if (tag == w::CONFLICT_RESPONSE) {  // notify_message has no arm for it ("receive unexpected message type", delegate.rs:286-288)
      if (consumed) *consumed = used;
      return SIM_OK;
    }
    if (tag == w::PUSH_PULL) {  // what merge_remote_state (delegate.rs:427-554) does with it
      w::PushPull m = w::decode_push_pull(body);
      for (auto& st : m.status_ltimes)
        if (st.first >= h->d.N) return SIM_EINVAL;
      for (u32 id : m.left_members)
        if (id >= h->d.N) return SIM_EINVAL;
      const u64 clk[3] = {m.ltime, m.event_ltime, m.query_ltime};
      for (u32 i = 0; i < 3 && rc == SIM_OK; ++i)  // "we subtract 1 since no message with that clock has been sent yet"
        if (clk[i] > 0) rc = inject_val(h, h->tick, SIM_OP_WITNESS, node, i, 0, clk[i] - 1);
      auto is_left = [&](u32 id) { return std::find(m.left_members.begin(), m.left_members.end(), id) != m.left_members.end(); };
      for (size_t i = 0; i < m.left_members.size() && rc == SIM_OK; ++i) {  // the left members first, one past their status time
        size_t j = 0;
        while (j < m.status_ltimes.size() && m.status_ltimes[j].first != m.left_members[i]) ++j;
        if (j < m.status_ltimes.size())
          rc = inject_val(h, h->tick, SIM_OP_DELIVER, node, m.left_members[i], wire_meta(SIM_K_LEAVE, 0, 16) | SIM_DELIVER_MUTE, m.status_ltimes[j].second + 1);
      }
      for (size_t j = 0; j < m.status_ltimes.size() && rc == SIM_OK; ++j)  // every other member: an artificial join message at its status time
        if (!is_left(m.status_ltimes[j].first))
          rc = inject_val(h, h->tick, SIM_OP_DELIVER, node, m.status_ltimes[j].first, wire_meta(SIM_K_JOIN, 0, 16) | SIM_DELIVER_MUTE, m.status_ltimes[j].second);
      for (auto& bucket : m.events)  // the event buffer, replayed in order
        for (auto& ev : bucket.second) {
          if (rc != SIM_OK) break;
          u32 key = w::event_key(ev.first, ev.second);
          h->evreg.emplace(key, std::make_pair(ev.first, ev.second));
          rc = inject_val(h, h->tick, SIM_OP_DELIVER, node, key, wire_meta(SIM_K_EVENT, 0, 32) | SIM_DELIVER_MUTE, bucket.first);
        }
      if (rc == SIM_OK && consumed) *consumed = used;
      return rc;
    }
    if (tag == w::JOIN) {
      w::Join m = w::decode_join(body);
      if (m.id >= h->d.N) return SIM_EINVAL;
      rec.key = m.id; rec.val = m.ltime; rec.meta = wire_meta(SIM_K_JOIN, 0, (u32)used);
    } else if (tag == w::LEAVE) {
      w::Leave m = w::decode_leave(body);
      if (m.id >= h->d.N) return SIM_EINVAL;
      rec.key = m.id; rec.val = m.ltime; rec.meta = wire_meta(SIM_K_LEAVE, m.prune ? SIM_F_PRUNE : 0, (u32)used);
    } else if (tag == w::USER_EVENT) {
      w::UserEvent m = w::decode_user_event(body);
      rec.key = w::event_key(m.name, m.payload); rec.val = m.ltime;
      rec.meta = wire_meta(SIM_K_EVENT, m.cc ? SIM_F_CC : 0, (u32)used);
      h->evreg.emplace(rec.key, std::make_pair(m.name, m.payload));
    } else if (tag == w::QUERY) {
      w::Query m = w::decode_query(body);
      if (!m.id) return SIM_EINVAL;
      rec.key = m.id; rec.val = m.ltime;
      rec.meta = wire_meta(SIM_K_QUERY, ((m.flags & 1u) ? SIM_F_ACK : 0u) | ((m.flags & 2u) ? SIM_F_NO_BROADCAST : 0u), 48);  // every query is priced at 48 B
      std::vector<u32> ids;
      for (const w::Bytes& f : m.filters) {  // types/filter.rs:176-262: Id = (id_byte <id>)*, Tag = tag_byte <TagFilter>
        size_t off = 0;
        while (off < f.size()) {
          if ((f[off++] >> 3) != 1) return SIM_EINVAL;  // a tag expression: evaluated by the host (sim_query_filtered)
          u32 g = w::parse_node_id(w::read_ld(f, off));
          if (g >= h->d.N || ids.size() == SIM_QF_IDS) return SIM_EINVAL;
          ids.push_back(g);
        }
      }
      for (u32 g : ids)
        if ((rc = inject_val(h, h->tick, SIM_OP_QUERY_FILTER_ID, node, rec.key, g, 0)) != SIM_OK) return rc;
    } else {
      return SIM_EINVAL;  // not a message of the simulated path
    }
    rc = sim_inject_record(h, h->tick, node, &rec);
    if (rc == SIM_OK && consumed) *consumed = used;
    return rc;
  } catch (const std::exception&) {
    return SIM_EINVAL;
  }
}
int sim_deliver_message(sim_handle* h, uint32_t node, const uint8_t* buf, size_t len, size_t* consumed) {
  return deliver_one(h, node, buf, len, consumed, false);
}
int sim_join(sim_handle* h, uint32_t node, uint32_t peer) { return sim_inject(h, h ? h->tick : 0, SIM_OP_JOIN, node, peer, 0); }
int sim_leave(sim_handle* h, uint32_t node) {
  if (!h) return SIM_EINVAL;
  // api.rs:422-499: leave intent now; memberlist.leave after broadcast_timeout; the caller's
  // shutdown() (api.rs:525) after leave_propagate_delay — both modelled as leave_delay ticks
  int rc = sim_inject(h, h->tick, SIM_OP_LEAVE, node, 0, 0);
  if (rc) return rc;
  rc = sim_inject(h, h->tick + h->cfg.leave_delay + 1, SIM_OP_LEAVE_FINISH, node, 0, 0);
  if (rc) return rc;
  return sim_inject(h, h->tick + 2 * h->cfg.leave_delay + 2, SIM_OP_CRASH, node, 0, 0);
}
int sim_force_leave(sim_handle* h, uint32_t node, uint32_t subject, int prune) {
  return sim_inject(h, h ? h->tick : 0, SIM_OP_FORCE_LEAVE, node, subject, prune ? 1u : 0u);
}
int sim_user_event(sim_handle* h, uint32_t node, uint32_t key, uint32_t len, int cc) {
  // UserEventMessage.cc (types/user_event/message.rs) travels in the record's flag bits
  return sim_inject(h, h ? h->tick : 0, SIM_OP_USER_EVENT, node, key, (len & 0x7FFFFFFFu) | (cc ? 0x80000000u : 0u));
}
int sim_query(sim_handle* h, uint32_t node, uint32_t id, uint32_t flags) {
  return sim_inject(h, h ? h->tick : 0, SIM_OP_QUERY, node, id, flags);
}
int sim_query_filtered(sim_handle* h, uint32_t node, uint32_t id, uint32_t flags, const uint32_t* ids, uint32_t n_ids, uint32_t tag_mask) {
  if (!h || !id || node >= h->d.N || (n_ids && !ids)) return SIM_EINVAL;
  if (n_ids > SIM_QF_IDS) return SIM_ETOOBIG;
  for (u32 i = 0; i < n_ids; ++i)
    if (ids[i] >= h->d.N) return SIM_EINVAL;
  int rc = SIM_OK;
  for (u32 i = 0; i < n_ids && rc == SIM_OK; ++i) rc = sim_inject(h, h->tick, SIM_OP_QUERY_FILTER_ID, node, id, ids[i]);
  if (rc == SIM_OK && tag_mask != 0xFFFFFFFFu) rc = sim_inject(h, h->tick, SIM_OP_QUERY_FILTER_TAGS, node, id, tag_mask);
  return rc ? rc : sim_inject(h, h->tick, SIM_OP_QUERY, node, id, flags);
}
int sim_init_tags(sim_handle* h, uint32_t first, uint32_t count, const uint8_t* classes) {
  if (!h || !classes || first > h->d.N || count > h->d.N - first) return SIM_EINVAL;
  for (u32 i = 0; i < count; ++i)
    if (classes[i] >= SIM_TAG_CLASSES) return SIM_EINVAL;
  if (!count) return SIM_OK;
  HCHECK(hipMemcpyAsync(TAGCLASS(h->d) + first, classes, count, hipMemcpyHostToDevice, h->stream));
  HCHECK(hipStreamSynchronize(h->stream));  // the caller's buffer is free again when this returns
  return SIM_OK;
}
int sim_set_tags(sim_handle* h, uint32_t node, uint32_t tag_class) {
  return sim_inject(h, h ? h->tick : 0, SIM_OP_SET_TAGS, node, tag_class, 0);
}

// One tick = sim_step_begin (operations, push-pull batch, tick parameters), one tick-kernel launch per sender chunk
// (sim_step_chunk; a single launch when there is one chunk or all shards are local), sim_step_end.  sim_step does all
// of it; a sharded host that wants the exchange of chunk c in flight while chunk c + 1 computes drives the three
// calls itself (serf_amd/shard.py).
// ---- cross-shard push-pull, driven by the sharded host (include/serf_sim.h; oracle pp_plan / pp_export / pp_merge) ----
static bool pp_batch_class(const sim_handle* h, u32* cls) {
  if (!h->pp_step || h->tick == 0 || h->tick % h->pp_step) return false;
  *cls = (u32)((h->tick / h->pp_step) % PP_GROUPS);
  return true;
}
int sim_pp_due(const sim_handle* h) {
  u32 cls;
  if (!h) return SIM_EINVAL;
  // (a reconnect attempt is a push-pull pair as well: known once sim_step_begin has resolved the tick's operations)
  return (h->d.sharded && (pp_batch_class(h, &cls) || (h->in_tick && !h->rc_a.empty())) && h->pp_done_at != (u32)h->tick) ? 1 : 0;
}
int sim_pp_plan(sim_handle* h, uint32_t* send1, uint32_t* recv1, size_t* record_bytes) {
  u32 cls = 0;
  if (!h || !send1 || !recv1 || !record_bytes) return SIM_EINVAL;
  const bool batch = h->in_tick && pp_batch_class(h, &cls);
  if (!h->in_tick || !h->d.sharded || (!batch && h->rc_a.empty())) return SIM_ESTATE;  // after sim_step_begin: the tick's operations come first
  Dev& d = h->d;
  const TickP& tp = h->cur_tp;
  const u32 V = d.V, me = d.shard_rank, M = d.M;
  std::vector<u32> up(((size_t)d.N + 31) / 32);  // ground-truth liveness after this tick's operations
  HCHECK(hipMemcpyAsync(up.data(), d.upmap, up.size() * 4, hipMemcpyDeviceToHost, h->stream));
  HCHECK(hipStreamSynchronize(h->stream));
  auto is_up = [&](u32 g) { return (up[g >> 5] >> (g & 31)) & 1u; };
  for (u32 v = 0; v < V; ++v) send1[v] = recv1[v] = 0;
  h->pp_local_a.clear(); h->pp_local_b.clear();
  std::vector<std::vector<u32>> r1(V), s1(V);
  auto place = [&](u32 ga, u32 gb) {
    if (!is_up(ga) || !is_up(gb)) return;
    u32 oa = ga / M, ob = gb / M;
    if (oa == me && ob == me) { h->pp_local_a.push_back(ga - d.shard0); h->pp_local_b.push_back(gb - d.shard0); }
    else if (oa == me) r1[ob].push_back(ga - d.shard0);
    else if (ob == me) s1[oa].push_back(gb - d.shard0);
  };
  if (batch)
    for (u32 pi = cls; 2 * (u64)pi + 1 < tp.N; pi += PP_GROUPS) place(sigma_g_inv(tp, 2 * pi), sigma_g_inv(tp, 2 * pi + 1));
  else  // the tick's reconnect attempts (sim_step_begin): the initiator is `a`, it merges first
    for (size_t i = 0; i < h->rc_a.size(); ++i) place(h->rc_a[i], h->rc_b[i]);
  h->pp_r1.clear(); h->pp_s1.clear();
  for (u32 v = 0; v < V; ++v) {
    recv1[v] = (u32)r1[v].size(); send1[v] = (u32)s1[v].size();
    h->pp_r1.insert(h->pp_r1.end(), r1[v].begin(), r1[v].end());
    h->pp_s1.insert(h->pp_s1.end(), s1[v].begin(), s1[v].end());
  }
  // the four lists on the device: local a | local b | r1 | s1
  size_t tot = h->pp_local_a.size() * 2 + h->pp_r1.size() + h->pp_s1.size();
  if (h->d_pp) { (void)hipFree(h->d_pp); h->d_pp = nullptr; }
  if (tot) {
    if (hipMalloc((void**)&h->d_pp, tot * 4) != hipSuccess) return SIM_ENOMEM;
    std::vector<u32> all;
    all.reserve(tot);
    all.insert(all.end(), h->pp_local_a.begin(), h->pp_local_a.end());
    all.insert(all.end(), h->pp_local_b.begin(), h->pp_local_b.end());
    all.insert(all.end(), h->pp_r1.begin(), h->pp_r1.end());
    all.insert(all.end(), h->pp_s1.begin(), h->pp_s1.end());
    HCHECK(hipMemcpy(h->d_pp, all.data(), tot * 4, hipMemcpyHostToDevice));
  }
  *record_bytes = (2 + (size_t)tp.n_slots + 2 * (size_t)d.Bev) * 16;
  return SIM_OK;
}
int sim_pp_export(sim_handle* h, int round, void* send) {
  if (!h || (round != 1 && round != 2)) return SIM_EINVAL;
  if (!h->in_tick || !h->d.sharded) return SIM_ESTATE;
  Dev& d = h->d;
  const TickP& tp = h->cur_tp;
  size_t nl = h->pp_local_a.size();
  const u32* list = h->d_pp + 2 * nl + (round == 1 ? h->pp_r1.size() : 0);
  u32 n = (u32)(round == 1 ? h->pp_s1.size() : h->pp_r1.size());
  if (!n) return SIM_OK;
  if (!send) return SIM_EINVAL;
  size_t rec_u4 = 2 + (size_t)tp.n_slots + 2 * (size_t)d.Bev;
  pp_export_kernel<<<n, 256, 0, h->stream>>>(d, list, tp.n_slots, (uint4*)send, rec_u4);
  HCHECK(hipGetLastError());
  return SIM_OK;
}
int sim_pp_merge(sim_handle* h, int round, const void* recv) {
  if (!h || (round != 1 && round != 2)) return SIM_EINVAL;
  if (!h->in_tick || !h->d.sharded) return SIM_ESTATE;
  Dev& d = h->d;
  const TickP& tp = h->cur_tp;
  size_t nl = h->pp_local_a.size();
  size_t rec_u4 = 2 + (size_t)tp.n_slots + 2 * (size_t)d.Bev;
  if (round == 1) {
    if (nl) pp_local_kernel<<<(unsigned)((nl + 63) / 64), 64, 0, h->stream>>>(d, tp, h->d_pp, h->d_pp + nl, (u32)nl);
    u32 n = (u32)h->pp_r1.size();
    if (n && !recv) return SIM_EINVAL;
    if (n) pp_cross_kernel<<<(n + 63) / 64, 64, 0, h->stream>>>(d, tp, h->d_pp + 2 * nl, n, (const uint4*)recv, rec_u4);
  } else {
    u32 n = (u32)h->pp_s1.size();
    if (n && !recv) return SIM_EINVAL;
    if (n) pp_cross_kernel<<<(n + 63) / 64, 64, 0, h->stream>>>(d, tp, h->d_pp + 2 * nl + h->pp_r1.size(), n, (const uint4*)recv, rec_u4);
    h->pp_done_at = (u32)h->tick;
  }
  HCHECK(hipGetLastError());
  return SIM_OK;
}
int sim_recycle_due(const sim_handle* h) { return h ? (recycle_is_due(h) ? 1 : 0) : SIM_EINVAL; }
int sim_recycle_scan(sim_handle* h, sim_recycle_cand* out, uint32_t cap, uint32_t* n) {
  if (!h || !out || !n || cap < SIM_RECYCLE_BATCH) return SIM_EINVAL;
  if (h->in_tick) return SIM_ESTATE;
  *n = recycle_candidates(h, out);
  return recycle_scan(h, out, *n);
}
int sim_recycle_apply(sim_handle* h, const sim_recycle_cand* agreed, uint32_t n) {
  if (!h || (n && !agreed)) return SIM_EINVAL;
  if (h->in_tick) return SIM_ESTATE;
  int rc = recycle_apply(h, agreed, n);
  h->recycle_at = (u32)h->tick;
  return rc;
}
// the fan-out graph of `tick` (random fan-out), on stream `s`: rf_rcsr / rf_rsrc [tick % 3] := the rows of the packets sent during `tick`
static int rf_build(sim_handle* h, u64 tick, hipStream_t s) {
  RfP r = h->rfp;
  TickP tp;
  tickp_make(&tp, &h->cfg, tick);
  r.rb = rng_base(h->cfg.seed, STREAM_RFAN, tick);
  r.feff = tp.feff;
  u32 *rcsr = h->rf_rcsr[tick % 3], *rsrc = h->rf_rsrc[tick % 3];
  uint8_t* cntb = h->d.sharded ? h->rf_cntb[tick % 3] : nullptr;  // a shard: the sending side's sort (SIM_XCHG_PACKED)
  u32* btot = h->d.sharded ? h->rf_btot[tick % 3] : nullptr;
  const u32 par = h->rf_par;
  h->rf_par ^= 1u;
  if (h->rf_wide) {
    rf_scatter_kernel<u64><<<r.NWG, RFB, rf_scatter_lds<u64>(r), s>>>(r, h->rf_gcur[par], (u64*)h->rf_l1, (u64*)h->rf_ovf[par]);
    rf_rows_kernel<u64><<<r.NB, RFR, rf_rows_lds(r), s>>>(r, h->rf_gcur[par], h->rf_gcur[par ^ 1u], (const u64*)h->rf_l1, (u64*)h->rf_ovf[par], (u64*)h->rf_ovf[par ^ 1u], rcsr, rsrc, cntb, btot, h->xflag);
  } else {
    rf_scatter_kernel<u32><<<r.NWG, RFB, rf_scatter_lds<u32>(r), s>>>(r, h->rf_gcur[par], (u32*)h->rf_l1, (u32*)h->rf_ovf[par]);
    rf_rows_kernel<u32><<<r.NB, RFR, rf_rows_lds(r), s>>>(r, h->rf_gcur[par], h->rf_gcur[par ^ 1u], (const u32*)h->rf_l1, (u32*)h->rf_ovf[par], (u32*)h->rf_ovf[par ^ 1u], rcsr, rsrc, cntb, btot, h->xflag);
  }
  if (h->d.sharded) rfx_soff_kernel<<<1, 64, 0, s>>>(h->rfx, btot, h->rf_xoff[tick % 3], h->xflag);
  HCHECK(hipGetLastError());
  return SIM_OK;
}
// random fan-out on a shard: the packets sent during tick `t` (in their senders' cells) -> the V slabs of the send buffer, on the
// handle's stream, behind the tick's launch (SIM_XCHG_PACKED).  The sort of tick t was enqueued on the build stream a tick ago
// (right after a restore, or with SERF_RF_SYNC: it is built here and now).
static int rfx_pack(sim_handle* h, u64 t) {
  Dev& d = h->d;
  if (h->rf_q[t % 3] != t) {
    int rc = rf_build(h, t, h->stream);
    if (rc) return rc;
    h->rf_q[t % 3] = t;
  } else HCHECK(hipStreamWaitEvent(h->stream, h->rf_done[t % 3], 0));
  const RfxL& x = h->rfx;
  rfx_meta_kernel<<<grid_for(((size_t)x.M + (size_t)x.NBh * 4u + 64u) * x.V), BLOCK, 0, h->stream>>>(x, h->rf_cntb[t % 3], h->rf_btot[t % 3], h->rf_xoff[t % 3], (u32)t, d.xsend);
  rfx_pack_kernel<<<grid_for((size_t)d.f * d.Nl * 4u), BLOCK, 0, h->stream>>>(x, h->rf_rsrc[t % 3], h->rf_xoff[t % 3], d.obox[0], d.Nl, d.xsend);
  HCHECK(hipGetLastError());
  return SIM_OK;
}
int sim_step_begin(sim_handle* h) {
  if (!h) return SIM_EINVAL;
  Dev& d = h->d;
  if (h->in_tick || (d.sharded && !h->bound)) return SIM_ESTATE;
  if (h->xflag && *h->xflag) return SIM_ERANGE;  // a slab of the random fan-out's exchange (or a count byte, or the rows) overflowed
  if (h->xpending) { int rc = sim_exchange_wait(h); if (rc) return rc; }  // the packets of the round before have landed
  if (d.swim && !d.sharded) {
    // every shard is here: the slot-less suspicions / reconnect attempts of the tick BEFORE the one that just ended are
    // replayed now — behind whatever the caller scheduled for this tick so far, which is where a sharded host
    // (sim_suspect_import at the start of its step) puts them too
    static thread_local std::vector<u32> buf(2 * SIM_SUSPECT_REQ_MAX);
    u32 n = 0;
    int rc = sim_suspect_requests(h, buf.data(), SIM_SUSPECT_REQ_MAX, &n);
    if (rc) return rc;
    for (u32 i = 0; i < n; ++i)
      if ((rc = inject_val(h, h->tick, SIM_OP_SUSPECT, buf[2 * i], buf[2 * i + 1], 0, 0)) != SIM_OK) return rc;
  }
  if (recycle_is_due(h)) {
    if (d.sharded) return SIM_ESTATE;  // the host runs the pass first (it needs every shard's verdict)
    int rc = recycle_local(h);
    if (rc) return rc;
  }
  if (d.swim) {  // this tick's request list: its count was zeroed by the previous tick's kernel (or never used); the two
    // other buffers hold the lists of the two ticks before until they have been read
    d.sreq = h->sreq_buf[h->tick % 3];
    d.sreq_next = h->sreq_buf[(h->tick + 1) % 3];
    d.sreq_hh = d.sharded ? nullptr : h->sreq_host[h->tick % 3];
  }
  TickP& tp = h->cur_tp;
  tickp_make(&tp, &h->cfg, h->tick);
#ifdef TICK_ABLATE
  tp.abl = g_ablate;
#endif
  for (u32 k = 0; k < SIM_MAX_FANOUT; ++k) { tp.prot[k] = h->prev.rot[k]; tp.prho[k] = h->prev.rho[k]; }
  if (d.sharded) d.xrecv = h->rbuf[(h->tick + 1) & 1];  // what was sent during tick - 1
  std::vector<u32> rc_req;  // this tick's reconnect attempts (node, target), in schedule order
  h->rc_a.clear(); h->rc_b.clear();
  while (h->op_cursor < h->ops.size() && h->ops[h->op_cursor].tick <= h->tick) {
    OpBatch ob;
    memset(&ob, 0, sizeof ob);
    while (ob.n < 8 && h->op_cursor < h->ops.size() && h->ops[h->op_cursor].tick <= h->tick) {
      const OpEnt& e = h->ops[h->op_cursor++];
      if (e.op == SIM_OP_RECONNECT) { rc_req.push_back(e.node); rc_req.push_back(e.a); continue; }  // resolved below, once the tick's operations have run
      if (e.op == SIM_OP_QUERY_FILTER_ID || e.op == SIM_OP_QUERY_FILTER_TAGS || e.op == SIM_OP_QUERY) {
        // the query's filter entry: started by the first filter operation that names the query, SEALED by its
        // SIM_OP_QUERY (word 3), replaced by whatever names another query with the same residue — or the same id again
        // once the entry is sealed: a query issued a second time under an id starts from no filters
        u32* f = h->qfilt.data() + (size_t)(e.a % SIM_QT) * SIM_QF_WORDS;
        bool changed = false;
        if (f[0] != e.a || (f[3] & 1u)) { memset(f, 0, SIM_QF_WORDS * 4); f[0] = e.a; f[2] = 0xFFFFFFFFu; changed = true; }
        if (e.op == SIM_OP_QUERY_FILTER_ID) {
          if (f[1] == SIM_QF_IDS) { h->ops_dropped++; continue; }  // model bound: the id does not fit
          f[4 + f[1]++] = e.b; changed = true;
        } else if (e.op == SIM_OP_QUERY_FILTER_TAGS) { f[2] &= e.b; changed = true; }
        else { f[3] |= 1u; changed = true; }
        if (changed) {
          QFiltEnt qe;
          memcpy(&qe, f, sizeof qe);
          qfilt_set_kernel<<<1, 64, 0, h->stream>>>(QFILT(d) + (size_t)(e.a % SIM_QT) * (SIM_QF_WORDS / 4), qe);
        }
        if (e.op != SIM_OP_QUERY) continue;
      }
      u32 x = op_subject(h, e.op, e.node, e.a, e.b);
      if (x != NOSLOT && ensure_slot(h, x) != SIM_OK) { h->ops_dropped++; continue; }  // no free view slot: the operation does not happen
      if (e.op == SIM_OP_JOIN && (h->cfg.flags & SIM_CF_JOIN_SYNC) && e.node >= d.shard0 && e.node < d.shard0 + d.Nl) {
        // memberlist.join comes first: what was batched so far runs, then the joining node adopts its partner's view
        if (ob.n) ops_kernel<<<1, 64, 0, h->stream>>>(d, ob, h->tick, d.N > 1 ? 1u : 0u, tp.query_base, h->q_timeout);
        memset(&ob, 0, sizeof ob);
        join_sync_kernel<<<1, BLOCK, 0, h->stream>>>(d, (u32)h->walk.size(), e.node, e.a, (u32)h->tick);
      }
      ob.op[ob.n] = e.op; ob.node[ob.n] = e.node; ob.a[ob.n] = e.a; ob.b[ob.n] = e.b; ob.val[ob.n] = e.val;
      if (e.op == SIM_OP_QUERY) {  // a fresh tracker: who acked / responded starts empty
        u32 j = e.a % SIM_QT;
        size_t words = ((size_t)d.N + 31) / 32;
        ob.c[ob.n] = j;
        HCHECK(hipMemsetAsync(d.qbits + (size_t)j * 2 * words, 0, 2 * words * 4, h->stream));
      }
      ob.n++;
    }
    if (ob.n) ops_kernel<<<1, 64, 0, h->stream>>>(d, ob, h->tick, d.N > 1 ? 1u : 0u, tp.query_base, h->q_timeout);
  }
  tp.n_slots = (u32)h->walk.size();  // after the operations: they may have taken slots
  if (!rc_req.empty()) {
    // The tick's SIM_OP_RECONNECT operations -> the push-pull pairs that run in this tick (oracle rc_resolve): an attempt
    // whose initiator or target is not running fails and is forgotten; the pairs of a tick are disjoint and do not share
    // the tick with a push-pull batch — an attempt that would goes back on the schedule for the next tick.
    std::vector<u32> up(((size_t)d.N + 31) / 32);  // ground-truth liveness after this tick's operations (rare path: a copy and a wait)
    HCHECK(hipMemcpyAsync(up.data(), d.upmap, up.size() * 4, hipMemcpyDeviceToHost, h->stream));
    HCHECK(hipStreamSynchronize(h->stream));
    auto is_up = [&](u32 g) { return (up[g >> 5] >> (g & 31)) & 1u; };
    u32 cls;
    const bool batch = pp_batch_class(h, &cls);
    for (size_t i = 0; i + 1 < rc_req.size(); i += 2) {
      const u32 a = rc_req[i], b = rc_req[i + 1];
      if (a == b || !is_up(a) || !is_up(b)) continue;
      bool busy = batch;
      for (size_t j = 0; j < h->rc_a.size() && !busy; ++j) busy = h->rc_a[j] == a || h->rc_b[j] == a || h->rc_a[j] == b || h->rc_b[j] == b;
      if (busy) { int rc = inject_val(h, h->tick + 1, SIM_OP_RECONNECT, a, b, 0, 0); if (rc) return rc; }
      else { h->rc_a.push_back(a); h->rc_b.push_back(b); }
    }
  }
  if (!d.sharded && h->pp_step && h->tick > 0 && h->tick % h->pp_step == 0) {  // (sharded: the host runs the batch, sim_pp_*)
    u32 cls = (u32)((h->tick / h->pp_step) % PP_GROUPS);
    u32 half = tp.N / 2, n_pairs = half > cls ? (half - cls + PP_GROUPS - 1) / PP_GROUPS : 0;
    if (n_pairs) pushpull_kernel<<<(n_pairs + 63) / 64, 64, 0, h->stream>>>(d, tp, cls, n_pairs);
  }
  if (!d.sharded)  // the Reconnector's push-pulls of this tick (none on a batch tick)
    for (size_t i = 0; i < h->rc_a.size(); i += 8) {
      PairBatch pb;
      memset(&pb, 0, sizeof pb);
      for (size_t j = i; j < h->rc_a.size() && j < i + 8; ++j) { pb.a[pb.n] = h->rc_a[j]; pb.b[pb.n++] = h->rc_b[j]; }
      pp_pairs_kernel<<<1, 64, 0, h->stream>>>(d, tp, pb);
    }
  // Timing of the tick's launch(es) with HIP events.  One launch per tick: the pair rides on the dispatch itself
  // (hipExtLaunchKernelGGL: start / stop = the kernel's own begin and end, no barrier packets in the stream — two
  // hipEventRecord calls around every launch cost 10 us of stream time each tick).  Several chunk launches per tick
  // (sharded, C > 1): the pair brackets them with hipEventRecord.
  if (d.gttd && !d.rfan) gossip_skip_kernel<<<grid_for(d.Nl), BLOCK, 0, h->stream>>>(d, tp, h->d_base);  // whom not to gossip to this tick
  if (d.rfan) {
    // kRandomNodes: the graph of the packets this tick RECEIVES — sent during tick - 1 — has to stand before the tick kernel
    // reads it.  It was enqueued on the build stream two ticks ago (right after a restore, at tick 1, or with SERF_RF_SYNC: it
    // is built here and now); the graphs of THIS tick's packets and the next tick's are enqueued now if they are not yet — as
    // soon as everything enqueued so far has finished: they overwrite a buffer the tick before this one read.
    if (d.sharded) {
      // a shard: the rows of this tick come from the slabs the round's exchange delivered (sim_exchange_wait above / the host's
      // collective has completed): one launch (rfx_index_kernel); an entry = a cell of the receive buffer
      const uint4* rb = h->rbuf[(h->tick + 1) & 1];
      if (h->tick > 0) rfx_index_kernel<<<h->rfx.NBh, RFX_T, (4u << h->rfx.LB) + 16u, h->stream>>>(h->rfx, rb, d.Nl, h->rx_cap, h->rx_rcsr, h->rx_rsrc, h->xflag);
      d.rcsr = h->rx_rcsr;  // (tick 0: zeros — nothing has been sent)
      d.rsrc = h->rx_rsrc;
      d.rfrd = rb;
      d.NC = 1u;            // the pages of a packet are adjacent cells of its slab
    } else {
      if (h->tick > 0) {
        const u64 s = h->tick - 1;
        if (h->rf_q[s % 3] != s) {
          int rc = rf_build(h, s, h->stream);
          if (rc) return rc;
          h->rf_q[s % 3] = s;
        } else HCHECK(hipStreamWaitEvent(h->stream, h->rf_done[s % 3], 0));
      }
      d.rcsr = h->rf_rcsr[(h->tick + 2) % 3];  // (tick 0: a buffer of zeros — nothing has been sent)
      d.rsrc = h->rf_rsrc[(h->tick + 2) % 3];
      d.rfrd = d.obox[h->tick & 1];  // ... and the cells those packets sit in: this handle's own of the tick before
      d.NC = d.Nl;
    }
    if (!h->rf_sync) {
      bool waited = false;
      for (u64 s = h->tick; s <= h->tick + 1; ++s) {
        if (h->rf_q[s % 3] == s) continue;
        if (!waited) {
          hipEvent_t go = h->rf_go[h->tick & 1];
          HCHECK(hipEventRecord(go, h->stream));
          HCHECK(hipStreamWaitEvent(h->rf_stream, go, 0));
          waited = true;
        }
        int rc = rf_build(h, s, h->rf_stream);
        if (rc) return rc;
        HCHECK(hipEventRecord(h->rf_done[s % 3], h->rf_stream));
        h->rf_q[s % 3] = s;
      }
    }
    if (d.gttd) {
      RfP r = h->rfp;
      r.rb = rng_base(h->cfg.seed, STREAM_RFAN, h->tick); r.feff = tp.feff;
      rf_skip_kernel<<<grid_for(d.Nl), BLOCK, 0, h->stream>>>(d, tp, r, h->d_base);
    }
  }
  h->tick_timed = h->profiling && (h->prof_seq++ % h->profiling) == 0;
  h->tick_bracket = h->tick_timed && d.sharded && tp.C > 1;
  if (h->tick_bracket) {
    HCHECK(hipEventCreate(&h->tick_ev0));
    HCHECK(hipEventRecord(h->tick_ev0, h->stream));
  }
  h->in_tick = true;
  return SIM_OK;
}
static int tick_launch(sim_handle* h, u32 chunk) {
  Dev& d = h->d;
  const TickP& tp = h->cur_tp;
  const TickP& ptp = h->tick ? h->prev : h->cur_tp;  // the map the packets in flight were sent with (tick 0: none are)
  u32 cnt = chunk == 0xFFFFFFFFu ? d.Nl : tp.V * tp.sub;
  int grid = (int)((cnt + TBLOCK - 1) / TBLOCK);
#ifdef TICK_PERSIST
  if (grid > TICK_PERSIST) grid = TICK_PERSIST;
#endif
  u32 cur = (u32)(h->tick & 1);
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (h->tick_timed && !h->tick_bracket) {
    HCHECK(hipEventCreate(&e0));
    HCHECK(hipEventCreate(&e1));
    h->prof.emplace_back(e0, e1);
  }
  // One launch per tick and a request list to read behind it (local mode, SWIM on): the event the host waits on before it
  // reads the list rides on the dispatch as its stop event — a hipEventRecord behind every launch is a marker packet of
  // its own, ~5 us of stream time per tick.
  h->sreq_on_dispatch = false;
  if (d.swim && !d.sharded && chunk == 0xFFFFFFFFu) {
    if (!e1) e1 = h->sreq_ev[h->tick % 3];
    h->sreq_wait[h->tick % 3] = e1;
    h->sreq_on_dispatch = true;
  }
#define LAUNCH_TICK_(SH, FF, BB, PP)                                                                                     \
  do {                                                                                                                   \
    if (e1) hipExtLaunchKernelGGL((tick_kernel<SH, FF, BB, PP>), dim3(grid), dim3(TBLOCK), 0, h->stream, e0, e1, 0, d, tp, ptp, cur, \
                                  (const uint4*)h->d_base, chunk, cnt);                                                  \
    else tick_kernel<SH, FF, BB, PP><<<grid, TBLOCK, 0, h->stream>>>(d, tp, ptp, cur, h->d_base, chunk, cnt);            \
  } while (0)
#define LAUNCH_TICK(SH, FF, BB) do { if (d.PG > 1u) LAUNCH_TICK_(SH, FF, BB, true); else LAUNCH_TICK_(SH, FF, BB, false); } while (0)
#define LAUNCH_LOCAL(FF)                                                                                                 \
  do {                                                                                                                   \
    if (d.rfan && d.PG > 1u) {                                                                                           \
      if (e1) hipExtLaunchKernelGGL((tick_kernel<false, FF, false, true, true>), dim3(grid), dim3(TBLOCK), 0, h->stream, e0, e1, 0, d, tp, ptp, \
                                    cur, (const uint4*)h->d_base, chunk, cnt);                                           \
      else tick_kernel<false, FF, false, true, true><<<grid, TBLOCK, 0, h->stream>>>(d, tp, ptp, cur, h->d_base, chunk, cnt); \
    } else if (d.rfan) {                                                                                                 \
      if (e1) hipExtLaunchKernelGGL((tick_kernel_rf<FF>), dim3(grid), dim3(TBLOCK), 0, h->stream, e0, e1, 0, d, tp, ptp, \
                                    cur, (const uint4*)h->d_base, chunk, cnt);                                           \
      else tick_kernel_rf<FF><<<grid, TBLOCK, 0, h->stream>>>(d, tp, ptp, cur, h->d_base, chunk, cnt); \
    } else if (tp.B == 64u) LAUNCH_TICK(false, FF, true);                                                                \
    else LAUNCH_TICK(false, FF, false);                                                                                  \
  } while (0)
  switch (tp.feff + ((d.sharded && !d.rfan) ? 4u : 0u)) {  // one instantiation per fan-out: the drain loop is fully unrolled
    // (the random fan-out on a shard runs the local instantiation: its packets stay in its cells, the exchange gathers them)
    case 0: case 1: LAUNCH_LOCAL(1); break;
    case 2: LAUNCH_LOCAL(2); break;
    case 3: LAUNCH_LOCAL(3); break;
    case 4: LAUNCH_LOCAL(4); break;
    case 5: LAUNCH_TICK(true, 1, false); break;
    case 6: LAUNCH_TICK(true, 2, false); break;
    case 7: LAUNCH_TICK(true, 3, false); break;
    default: LAUNCH_TICK(true, 4, false); break;
  }
#undef LAUNCH_LOCAL
#undef LAUNCH_TICK
#undef LAUNCH_TICK_
  HCHECK(hipGetLastError());
  return SIM_OK;
}
int sim_step_chunk(sim_handle* h, uint32_t chunk) {
  if (!h) return SIM_EINVAL;
  if (!h->in_tick) return SIM_ESTATE;
  if (!h->d.sharded || chunk >= h->cur_tp.C) return SIM_EINVAL;
  if (sim_pp_due(h) > 0) return SIM_ESTATE;  // the push-pull batch of this tick comes first (its pairs span shards: the host runs it)
  int rc = tick_launch(h, h->cur_tp.C == 1 ? 0xFFFFFFFFu : chunk);
  if (rc == SIM_OK && h->d.rfan) rc = rfx_pack(h, h->tick);  // the slabs of the round's exchange, from the cells the launch fills
  return rc;
}
int sim_step_end(sim_handle* h) {
  if (!h) return SIM_EINVAL;
  if (!h->in_tick) return SIM_ESTATE;
  if (h->tick_bracket) {
    hipEvent_t ev1 = nullptr;
    HCHECK(hipEventCreate(&ev1));
    HCHECK(hipEventRecord(ev1, h->stream));
    h->prof.emplace_back(h->tick_ev0, ev1);
  }
  h->prev = h->cur_tp;
  h->tick++;
  h->in_tick = false;
  // Slot-less failed probes: the head of this tick's list follows the launch into pinned memory; the list of the tick
  // BEFORE is read now (its copy landed a whole tick ago) and, every shard being here, replayed next tick.
  if (h->d.swim) {
    const u64 t = h->tick - 1;  // the tick that just ended
    if (!h->sreq_on_dispatch) {
      HCHECK(hipEventRecord(h->sreq_ev[t % 3], h->stream));
      h->sreq_wait[t % 3] = h->sreq_ev[t % 3];
    }
    h->sreq_on_dispatch = false;
    h->sreq_tick[t % 3] = t;
  }
  return SIM_OK;
}
// the list of one finished tick out of its buffer (sorted by prober); marks it read
static int sreq_take(sim_handle* h, u64 t, uint32_t* out, uint32_t cap_pairs, uint32_t* n_pairs) {
  *n_pairs = 0;
  const u32 b = (u32)(t % 3);
  if (h->sreq_tick[b] != t) return SIM_OK;  // nothing recorded for that tick, or read already
  h->sreq_tick[b] = ~0ull;
  if (h->sreq_wait[b]) HCHECK(hipEventSynchronize(h->sreq_wait[b]));
  u32* hh = h->sreq_host[b];
  u32 n = 0;
  if (!h->d.sharded) {  // the kernel wrote the head of the list here itself
    while (n < SREQ_HEAD && hh[2 * n] != 0xFFFFFFFFu) ++n;
    if (!n) return SIM_OK;
  }
  if (h->d.sharded || n == SREQ_HEAD) {  // a long list (or no host copy): the buffer on the device is untouched until the tick after next has run
    HCHECK(hipMemcpyAsync(&n, h->sreq_buf[b], 4, hipMemcpyDeviceToHost, h->stream));
    HCHECK(hipStreamSynchronize(h->stream));
  }
  auto forget = [&]() { if (!h->d.sharded) memset(hh, 0xFF, 2 * SREQ_HEAD * 4); };
  if (!n) return SIM_OK;
  if (n > SIM_SUSPECT_REQ_MAX) { h->ops_dropped += n; forget(); return SIM_OK; }  // model bound: the whole tick's list is dropped
  if (n > cap_pairs || !out) { h->sreq_tick[b] = t; return SIM_ERANGE; }
  if (!h->d.sharded && n <= SREQ_HEAD) memcpy(out, hh, (size_t)n * 8);
  else {
    HCHECK(hipMemcpyAsync(out, h->sreq_buf[b] + 1, (size_t)n * 8, hipMemcpyDeviceToHost, h->stream));
    HCHECK(hipStreamSynchronize(h->stream));
  }
  forget();
  std::vector<std::pair<u32, u32>> v(n);
  for (u32 i = 0; i < n; ++i) v[i] = {out[2 * i], out[2 * i + 1]};
  std::sort(v.begin(), v.end());  // a node probes once per tick: probers are distinct
  for (u32 i = 0; i < n; ++i) { out[2 * i] = v[i].first; out[2 * i + 1] = v[i].second; }
  *n_pairs = n;
  return SIM_OK;
}
int sim_suspect_export(sim_handle* h, void* out) {  // the head of the list of the tick that just ended -> device memory of the caller
  if (!h || !out || h->in_tick || !h->tick) return SIM_EINVAL;
  static_assert(SIM_SREQ_HEAD_WORDS * 4 <= (1 + 2 * SIM_SUSPECT_REQ_MAX) * 4, "the head is a prefix of the list buffer");
  if (!h->d.swim) { HCHECK(hipMemsetAsync(out, 0, SIM_SREQ_HEAD_WORDS * 4, h->stream)); return SIM_OK; }
  const u64 t = h->tick - 1;
  HCHECK(hipMemcpyAsync(out, h->sreq_buf[t % 3], SIM_SREQ_HEAD_WORDS * 4, hipMemcpyDeviceToDevice, h->stream));
  h->sreq_tick[t % 3] = ~0ull;  // handed over: nothing for sim_suspect_requests to read
  return SIM_OK;
}
int sim_suspect_import(sim_handle* h, uint64_t of_tick, const uint32_t* heads, uint32_t world) {
  if (!h || !heads || !world || h->in_tick || of_tick + 2 < h->tick) return SIM_EINVAL;
  std::vector<std::pair<u32, u32>> v;
  for (u32 w = 0; w < world; ++w) {
    const u32* hd = heads + (size_t)w * SIM_SREQ_HEAD_WORDS;
    if (hd[0] > SIM_SREQ_HEAD_PAIRS) { h->ops_dropped += hd[0]; continue; }  // model bound: that shard's list is dropped
    for (u32 i = 0; i < hd[0]; ++i) v.emplace_back(hd[1 + 2 * i], hd[2 + 2 * i]);
  }
  std::sort(v.begin(), v.end());
  for (auto& pr : v) {
    int rc = inject_val(h, of_tick + 2, SIM_OP_SUSPECT, pr.first, pr.second, 0, 0);
    if (rc) return rc;
  }
  return SIM_OK;
}
int sim_suspect_requests(sim_handle* h, uint32_t* out, uint32_t cap_pairs, uint32_t* n_pairs) {
  if (!h || !n_pairs || h->in_tick) return SIM_EINVAL;
  *n_pairs = 0;
  if (!h->d.swim || h->tick < 2) return SIM_OK;
  return sreq_take(h, h->tick - 2, out, cap_pairs, n_pairs);  // the requests of the tick BEFORE the one that just ended
}
int sim_step(sim_handle* h, uint32_t n_ticks) {
  if (!h) return SIM_EINVAL;
  Dev& d = h->d;
  if (d.sharded && !h->bound) return SIM_ESTATE;
  if (d.sharded && n_ticks > 1) return SIM_EINVAL;  // the caller has to move send -> recv between two ticks
  for (u32 it = 0; it < n_ticks; ++it) {
    if (sim_pp_due(h) > 0) return SIM_ESTATE;  // needs the host between begin and end (cross-shard push-pull batch)
    int rc = sim_step_begin(h);
    if (rc) return rc;
    if (d.sharded && h->cur_tp.C > 1) {
      for (u32 c = 0; c < h->cur_tp.C && rc == SIM_OK; ++c) rc = tick_launch(h, c);
    } else {
      rc = tick_launch(h, 0xFFFFFFFFu);
      if (rc == SIM_OK && d.sharded && d.rfan) rc = rfx_pack(h, h->tick);
    }
    int rc2 = sim_step_end(h);
    if (rc) return rc;
    if (rc2) return rc2;
  }
  return SIM_OK;
}
int sim_sync(sim_handle* h) {
  if (!h) return SIM_EINVAL;
  HCHECK(hipStreamSynchronize(h->stream));
  return SIM_OK;
}
int sim_tick(const sim_handle* h, uint64_t* t) {
  if (!h || !t) return SIM_EINVAL;
  *t = h->tick;
  return SIM_OK;
}

int sim_members(sim_handle* h, uint32_t obs, uint8_t* st, uint64_t* lt, uint32_t cap) {
  if (!h) return SIM_EINVAL;
  Dev& d = h->d;
  if (obs < d.shard0 || obs >= d.shard0 + d.Nl) return SIM_EINVAL;
  if (cap < d.N) return SIM_ERANGE;
  members_kernel<<<grid_for(d.N), BLOCK, 0, h->stream>>>(d, h->d_base, obs - d.shard0, h->d_mst, h->d_mlt);
  if (st) HCHECK(hipMemcpyAsync(st, h->d_mst, d.N, hipMemcpyDeviceToHost, h->stream));
  if (lt) HCHECK(hipMemcpyAsync(lt, h->d_mlt, (size_t)d.N * 8, hipMemcpyDeviceToHost, h->stream));
  HCHECK(hipStreamSynchronize(h->stream));
  return SIM_OK;
}
int sim_stats_get(sim_handle* h, uint32_t node, sim_stats* o) {
  if (!h || !o) return SIM_EINVAL;
  Dev& d = h->d;
  if (node < d.shard0 || node >= d.shard0 + d.Nl) return SIM_EINVAL;
  stats_kernel<<<1, 64, 0, h->stream>>>(d, node - d.shard0, h->d_stats);
  HCHECK(hipMemcpyAsync(o, h->d_stats, sizeof(sim_stats), hipMemcpyDeviceToHost, h->stream));
  HCHECK(hipStreamSynchronize(h->stream));
  return SIM_OK;
}
int sim_watch(sim_handle* h, uint32_t obs) {
  if (!h) return SIM_EINVAL;
  Dev& d = h->d;
  if (obs < d.shard0 || obs >= d.shard0 + d.Nl) return SIM_EINVAL;
  set_flag_bits<<<1, 64, 0, h->stream>>>(d.R1, obs - d.shard0, SIM_RF_WATCHED);
  return SIM_OK;
}
int sim_drain_events(sim_handle* h, sim_event* out, uint32_t cap, uint32_t* n) {
  if (!h || !n) return SIM_EINVAL;
  Dev& d = h->d;
  HCHECK(hipStreamSynchronize(h->stream));
  u32 cnt = 0;
  HCHECK(hipMemcpy(&cnt, d.ev_count, 4, hipMemcpyDeviceToHost));
  if (cnt > d.ev_cap) { h->events_lost += cnt - d.ev_cap; cnt = d.ev_cap; }
  std::vector<sim_event> ev(cnt);
  if (cnt) HCHECK(hipMemcpy(ev.data(), d.events, (size_t)cnt * sizeof(sim_event), hipMemcpyDeviceToHost));
  // per node the log is in program order; across nodes the oracle's order is (tick, observer)
  std::stable_sort(ev.begin(), ev.end(), [](const sim_event& a, const sim_event& b) {
    return a.tick != b.tick ? a.tick < b.tick : a.observer < b.observer;
  });
  u32 m = std::min(cnt, cap);
  if (out && m) memcpy(out, ev.data(), (size_t)m * sizeof(sim_event));
  // keep what did not fit
  u32 rest = cnt - m;
  if (rest) HCHECK(hipMemcpy(d.events, ev.data() + m, (size_t)rest * sizeof(sim_event), hipMemcpyHostToDevice));
  HCHECK(hipMemcpy(d.ev_count, &rest, 4, hipMemcpyHostToDevice));
  *n = m;
  return SIM_OK;
}

// The packets in flight, receiver-indexed ([f][Nl] cells).  Sharded: the receive buffer.  Local mode: Dev::obox turned
// inside out on h->stream (every user launches on that stream afterwards); the map is the one of the tick they were sent in.
static const uint4* cur_inbox(sim_handle* h) {
  if (h->d.sharded && !h->d.rfan) return h->rbuf[(h->tick + 1) & 1];
  if (h->mat_tick != h->tick) {
    const Dev& d = h->d;
    TickP p;
    tickp_make(&p, &h->cfg, h->tick ? h->tick - 1 : 0);
    materialize_kernel<<<grid_for((size_t)d.fp * d.Nl), BLOCK, 0, h->stream>>>(d, p, (u32)(h->tick & 1), h->tick ? 1u : 0u, h->inbox_mat);
    h->mat_tick = h->tick;
  }
  return h->inbox_mat;
}
int sim_state_digest(sim_handle* h, uint64_t out[8]) {
  if (!h || !out) return SIM_EINVAL;
  Dev& d = h->d;
  hipStream_t s = h->stream;
  HCHECK(hipMemsetAsync(h->d_scratch, 0, 16 * 8, s));
  digest_rows_queue<<<grid_for(d.Nl), BLOCK, 0, s>>>(d, h->d_scratch + 0, h->d_scratch + 1);
  size_t nw;
  if (cur_inbox(h)) { nw = (size_t)d.fp * d.Nl * (sizeof(sim_packet) / 8); digest_flat<<<grid_for(nw), BLOCK, 0, s>>>((const u64*)cur_inbox(h), nw, h->d_scratch + 2); }
  digest_split<<<grid_for(d.vtail), BLOCK, 0, s>>>(d.view, d.vtail, d.vtail, h->d_scratch + 3);
  digest_split<<<grid_for(d.etail), BLOCK, 0, s>>>(d.ering, d.etail, d.etail, h->d_scratch + 4);
  digest_split<<<grid_for(d.qtail), BLOCK, 0, s>>>(d.qring, d.qtail, d.qtail, h->d_scratch + 5);
  digest_aux<<<grid_for((size_t)d.N + d.N / 32 + 1), BLOCK, 0, s>>>(d.slot_of, d.upmap, d.N, h->d_scratch + 6);
  nw = (size_t)SIM_QT * 2 * (((size_t)d.N + 31) / 32);
  digest_queries<<<grid_for(nw + 2 * SIM_QT + SIM_QT * SIM_QF_WORDS + d.N), BLOCK, 0, s>>>(d.qtab, d.qbits, nw, d.N, h->d_scratch + 7);
  HCHECK(hipMemcpyAsync(out, h->d_scratch, 8 * 8, hipMemcpyDeviceToHost, s));
  HCHECK(hipStreamSynchronize(s));
  return SIM_OK;
}

// Split arrays (view, rings) <-> their canonical interleaved form, in chunks of at most 64 MiB of device scratch
static int split_copy(sim_handle* h, uint4* arr, size_t tail, size_t n_entries, void* host, bool download) {
  const size_t CH = (size_t)1 << 21;  // entries per chunk
  if (!n_entries) return SIM_OK;
  uint4* tmp = nullptr;
  if (hipMalloc((void**)&tmp, std::min(n_entries, CH) * 32) != hipSuccess) return SIM_ENOMEM;
  hipError_t e = hipSuccess;
  for (size_t first = 0; first < n_entries && e == hipSuccess; first += CH) {
    size_t cnt = std::min(CH, n_entries - first);
    uint8_t* hp = (uint8_t*)host + first * 32;
    if (download) {
      canon_entries_kernel<<<grid_for(cnt), BLOCK, 0, h->stream>>>(arr, tail, first, cnt, tmp);
      e = hipMemcpyAsync(hp, tmp, cnt * 32, hipMemcpyDeviceToHost, h->stream);
    } else {
      e = hipMemcpyAsync(tmp, hp, cnt * 32, hipMemcpyHostToDevice, h->stream);
      if (e == hipSuccess) uncanon_entries_kernel<<<grid_for(cnt), BLOCK, 0, h->stream>>>(arr, tail, first, cnt, tmp);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);  // tmp is reused by the next chunk
  }
  (void)hipFree(tmp);
  HCHECK(e);
  return SIM_OK;
}
int sim_dump_state(sim_handle* h, uint32_t which, void* buf, size_t cap, size_t* bytes) {
  if (!h || !bytes) return SIM_EINVAL;
  Dev& d = h->d;
  size_t Nl = d.Nl, n;
  const void* src = nullptr;
  switch (which) {
    case SIM_ARR_ROWS: n = Nl * sizeof(sim_row); break;
    case SIM_ARR_QUEUE: n = Nl * SIM_Q * sizeof(sim_record); break;
    case SIM_ARR_INBOX: src = cur_inbox(h); n = (size_t)d.fp * Nl * sizeof(sim_packet); break;
    case SIM_ARR_VIEW: src = d.view; n = (size_t)d.A * Nl * sizeof(sim_view); break;
    case SIM_ARR_ERING: src = d.ering; n = (size_t)d.Bev * Nl * sizeof(sim_bucket); break;
    case SIM_ARR_QRING: src = d.qring; n = (size_t)d.Bq * Nl * sizeof(sim_bucket); break;
    case SIM_ARR_SLOTMAP: src = d.slot_of; n = (size_t)d.N * 4; break;
    default: return SIM_EINVAL;
  }
  *bytes = n;
  if (!buf) return SIM_OK;
  if (cap < n) return SIM_ERANGE;
  HCHECK(hipStreamSynchronize(h->stream));
  if (which == SIM_ARR_ROWS || which == SIM_ARR_QUEUE) {  // canonical form is assembled on the device
    void* tmp = nullptr;
    if (hipMalloc(&tmp, std::max<size_t>(n, 16)) != hipSuccess) return SIM_ENOMEM;
    if (which == SIM_ARR_ROWS) canon_rows_kernel<<<grid_for(Nl), BLOCK, 0, h->stream>>>(d, (u64*)tmp);
    else canon_queue_kernel<<<grid_for(Nl), BLOCK, 0, h->stream>>>(d, (uint4*)tmp);
    // copy on the handle's stream: a non-blocking stream does not order against the null stream
    hipError_t e = hipMemcpyAsync(buf, tmp, n, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    (void)hipFree(tmp);
    HCHECK(e);
    return SIM_OK;
  }
  if (which == SIM_ARR_VIEW) return split_copy(h, d.view, d.vtail, d.vtail, buf, true);
  if (which == SIM_ARR_ERING) return split_copy(h, d.ering, d.etail, d.etail, buf, true);
  if (which == SIM_ARR_QRING) return split_copy(h, d.qring, d.qtail, d.qtail, buf, true);
  if (!src) { memset(buf, 0, n); return SIM_OK; }
  HCHECK(hipMemcpyAsync(buf, src, n, hipMemcpyDeviceToHost, h->stream));
  HCHECK(hipStreamSynchronize(h->stream));
  return SIM_OK;
}

int sim_convergence(sim_handle* h, uint32_t kind, uint32_t key, uint64_t ltime, uint64_t* seen, uint64_t* up) {
  if (!h || !seen || !up) return SIM_EINVAL;
  Dev& d = h->d;
  if (kind == SIM_K_JOIN || kind == SIM_K_LEAVE) { if (key >= d.N) return SIM_EINVAL; }
  else if (kind != SIM_K_EVENT && kind != SIM_K_QUERY) return SIM_EINVAL;
  hipStream_t s = h->stream;
  HCHECK(hipMemsetAsync(h->d_scratch + 8, 0, 16, s));
  convergence_kernel<<<grid_for(d.Nl), BLOCK, 0, s>>>(d, h->d_base, kind, key, ltime, h->d_scratch + 8);
  u64 r[2];
  HCHECK(hipMemcpyAsync(r, h->d_scratch + 8, 16, hipMemcpyDeviceToHost, s));
  HCHECK(hipStreamSynchronize(s));
  *seen = r[0];
  *up = r[1];
  return SIM_OK;
}

int sim_peek_packet(sim_handle* h, uint32_t node, uint32_t k, uint8_t* buf, size_t cap, size_t* len) {
  if (!h || !len || k >= h->d.f) return SIM_EINVAL;
  Dev& d = h->d;
  if (node < d.shard0 || node >= d.shard0 + d.Nl || h->in_tick) return SIM_EINVAL;
  namespace w = serf::wire;
  w::Bytes out;
  const TickP& p = h->prev;  // the map the packets in flight were sent with
  if (h->tick > 0 && k < p.feff) {
    u32 g = node / p.M, ll = node - g * p.M, hh, lp;
    fan_target_g(p, g, ll, k, hh, lp);
    std::vector<sim_packet> pages(d.PG);
    const uint4* src = (d.sharded && !d.rfan) ? d.xsend : cur_inbox(h);
    for (u32 pg = 0; pg < d.PG; ++pg) {
      size_t cell = d.rfan    ? (((size_t)k * d.PG + pg) * d.Nl + (node - d.shard0))  // random fan-out: the canonical inbox is sender-indexed
                    : d.sharded ? ((((size_t)((ll % p.blk) / p.sub) * p.V + hh) * d.fp + (size_t)k * d.PG + pg) * p.sub + lp % p.sub)
                              : (((size_t)k * d.PG + pg) * d.Nl + (size_t)hh * p.M + lp);
      HCHECK(hipMemcpyAsync(&pages[pg], src + cell * PK_U4, sizeof(sim_packet), hipMemcpyDeviceToHost, h->stream));
    }
    HCHECK(hipStreamSynchronize(h->stream));
    for (u32 pg = 0; pg < d.PG; ++pg)
      for (u32 r = 0; r < SIM_P; ++r) {
        const sim_packet& pk = pages[pg];
        u32 hm = pk.hi_meta[r], kind = (hm >> 4) & 15u, flags = hm & 15u;
        if (kind == SIM_K_EMPTY || kind >= SIM_K_ALIVE) continue;  // memberlist's own records are not serf messages
        u64 ltime = (u64)pk.val_lo[r] | ((u64)(hm >> 16) << 32);
        w::Bytes m;
        if (kind == SIM_K_JOIN) {
          w::Join j; j.ltime = ltime; j.id = pk.key[r];
          m = w::encode_message(j);
        } else if (kind == SIM_K_LEAVE) {
          w::Leave l; l.ltime = ltime; l.id = pk.key[r]; l.prune = flags & SIM_F_PRUNE;
          m = w::encode_message(l);
        } else if (kind == SIM_K_EVENT) {
          w::UserEvent e; e.ltime = ltime; e.cc = flags & SIM_F_CC;
          auto it = h->evreg.find(pk.key[r]);
          if (it != h->evreg.end()) { e.name = it->second.first; e.payload = it->second.second; }
          else { char nm[16]; int n = snprintf(nm, sizeof nm, "#%08x", pk.key[r]); e.name.assign(nm, nm + n); }
          m = w::encode_message(e);
        } else {
          w::Query q; q.ltime = ltime; q.id = pk.key[r];
          q.flags = ((flags & SIM_F_ACK) ? 1u : 0u) | ((flags & SIM_F_NO_BROADCAST) ? 2u : 0u);
          uint4 tj;
          HCHECK(hipMemcpy(&tj, d.qtab + pk.key[r] % SIM_QT, sizeof tj, hipMemcpyDeviceToHost));
          if (tj.x == pk.key[r]) { q.from_node = tj.y; q.relay_factor = (uint8_t)((tj.w >> 8) & 7u); }
          q.timeout_ms = (u64)h->q_timeout * 200u;  // gossip intervals of 200 ms
          q.name = {'#', 'q'};
          m = w::encode_message(q);
        }
        out.insert(out.end(), m.begin(), m.end());
      }
  }
  *len = out.size();
  if (!buf) return SIM_OK;
  if (out.size() > cap) return SIM_ERANGE;
  if (!out.empty()) memcpy(buf, out.data(), out.size());
  return SIM_OK;
}

int sim_convergence_many(sim_handle* h, uint32_t n, const uint32_t* kinds, const uint32_t* keys, const uint64_t* ltimes,
                         uint64_t* seen, uint64_t* up) {
  if (!h || !up || n > SIM_CONV_MAX || (n && (!kinds || !keys || !ltimes || !seen))) return SIM_EINVAL;
  Dev& d = h->d;
  ConvSet cs;
  memset(&cs, 0, sizeof cs);
  cs.n = n;
  for (u32 i = 0; i < n; ++i) {
    if (kinds[i] == SIM_K_JOIN || kinds[i] == SIM_K_LEAVE) { if (keys[i] >= d.N) return SIM_EINVAL; }
    else if (kinds[i] != SIM_K_EVENT && kinds[i] != SIM_K_QUERY) return SIM_EINVAL;
    else if (!keys[i]) return SIM_EINVAL;
    cs.kind[i] = kinds[i]; cs.key[i] = keys[i]; cs.ltime[i] = ltimes[i];
  }
  hipStream_t s = h->stream;
  u64* scr = nullptr;
  if (hipMalloc((void**)&scr, (SIM_CONV_MAX + 1) * 8) != hipSuccess) return SIM_ENOMEM;
  u64 r[SIM_CONV_MAX + 1];
  hipError_t e = hipMemsetAsync(scr, 0, sizeof r, s);
  if (e == hipSuccess) {
    convergence_many_kernel<<<grid_for(d.Nl), BLOCK, 0, s>>>(d, h->d_base, cs, scr);
    e = hipMemcpyAsync(r, scr, sizeof r, hipMemcpyDeviceToHost, s);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  (void)hipFree(scr);
  HCHECK(e);
  *up = r[0];
  for (u32 i = 0; i < n; ++i) seen[i] = r[1 + i];
  return SIM_OK;
}

// ---- checkpoint / resume (canonical image; layout documented in oracle/serf_oracle.c and DESIGN.md) ----
struct snap_header {
  uint32_t magic, abi;
  sim_config cfg;
  uint64_t tick;
  uint32_t n_slots, n_pending_ops;
  uint64_t ops_dropped, slots_recycled;
};
#define SNAP_MAGIC 0x53465253u
#define SNAP_SECTIONS 16
static void snap_lengths(const sim_handle* h, size_t len[SNAP_SECTIONS]) {
  const Dev& d = h->d;
  size_t nup = ((size_t)d.N + 31) / 32;
  size_t n[SNAP_SECTIONS] = {(size_t)d.Nl * sizeof(sim_row), (size_t)d.Nl * SIM_Q * sizeof(sim_record),
                             (size_t)d.fp * d.Nl * sizeof(sim_packet), (size_t)d.A * d.Nl * sizeof(sim_view),
                             (size_t)d.Bev * d.Nl * sizeof(sim_bucket), (size_t)d.Bq * d.Nl * sizeof(sim_bucket),
                             (size_t)d.N * 4, (size_t)d.A * 4, (size_t)d.N * sizeof(sim_view), nup * 4,
                             (size_t)SIM_QT * 16, (size_t)SIM_QT * 2 * nup * 4, (h->ops.size() - h->op_cursor) * sizeof(OpEnt),
                             (size_t)d.A * 4, (size_t)SIM_QT * SIM_QF_WORDS * 4, (size_t)d.N};
  memcpy(len, n, sizeof n);
}
int sim_snapshot(sim_handle* h, void* buf, size_t cap, size_t* bytes) {
  if (!h || !bytes) return SIM_EINVAL;
  if (h->in_tick) return SIM_ESTATE;  // between sim_step_begin and sim_step_end the state is half a tick ahead of `tick`
  Dev& d = h->d;
  if (d.swim && !d.sharded) {  // slot-less failed probes not yet replayed: into the schedule, so that the image holds them
    std::vector<u32> rq(2 * SIM_SUSPECT_REQ_MAX);
    for (u64 back = 2; back >= 1; --back) {
      if (h->tick < back) continue;
      u32 n = 0;
      int rc = sreq_take(h, h->tick - back, rq.data(), SIM_SUSPECT_REQ_MAX, &n);
      if (rc) return rc;
      for (u32 i = 0; i < n; ++i)
        if ((rc = inject_val(h, h->tick + 2 - back, SIM_OP_SUSPECT, rq[2 * i], rq[2 * i + 1], 0, 0)) != SIM_OK) return rc;
    }
  }
  size_t len[SNAP_SECTIONS], tot = sizeof(snap_header);
  snap_lengths(h, len);
  for (int i = 0; i < SNAP_SECTIONS; ++i) tot += 8 + len[i];
  *bytes = tot;
  if (!buf) return SIM_OK;
  if (cap < tot) return SIM_ERANGE;
  HCHECK(hipStreamSynchronize(h->stream));
  snap_header hd;
  memset(&hd, 0, sizeof hd);
  hd.magic = SNAP_MAGIC; hd.abi = SIM_ABI_VERSION; hd.cfg = h->cfg; hd.tick = h->tick; hd.n_slots = h->n_slots;
  hd.n_pending_ops = (uint32_t)(h->ops.size() - h->op_cursor);
  hd.ops_dropped = h->ops_dropped; hd.slots_recycled = h->slots_recycled;
  uint8_t* o = (uint8_t*)buf;
  memcpy(o, &hd, sizeof hd); o += sizeof hd;
  const uint32_t dumps[6] = {SIM_ARR_ROWS, SIM_ARR_QUEUE, SIM_ARR_INBOX, SIM_ARR_VIEW, SIM_ARR_ERING, SIM_ARR_QRING};
  for (int i = 0; i < SNAP_SECTIONS; ++i) {
    uint64_t n = len[i];
    memcpy(o, &n, 8); o += 8;
    if (i < 6) {
      size_t got = 0;
      int rc = sim_dump_state(h, dumps[i], o, n, &got);
      if (rc) return rc;
    } else if (i == 6) memcpy(o, h->slot_of.data(), n);
    else if (i == 7) memcpy(o, h->subject_of.data(), n);
    else if (i == 8) memcpy(o, h->base.data(), n);
    else if (i == 9) { HCHECK(hipMemcpyAsync(o, d.upmap, n, hipMemcpyDeviceToHost, h->stream)); HCHECK(hipStreamSynchronize(h->stream)); }
    else if (i == 10) { HCHECK(hipMemcpyAsync(o, d.qtab, n, hipMemcpyDeviceToHost, h->stream)); HCHECK(hipStreamSynchronize(h->stream)); }
    else if (i == 11) { HCHECK(hipMemcpyAsync(o, d.qbits, n, hipMemcpyDeviceToHost, h->stream)); HCHECK(hipStreamSynchronize(h->stream)); }
    else if (i == 12) { if (n) memcpy(o, h->ops.data() + h->op_cursor, n); }
    else if (i == 13) memcpy(o, h->alloc_tick.data(), n);
    else if (i == 14) memcpy(o, h->qfilt.data(), n);
    else { HCHECK(hipMemcpyAsync(o, TAGCLASS(d), n, hipMemcpyDeviceToHost, h->stream)); HCHECK(hipStreamSynchronize(h->stream)); }
    o += n;
  }
  return SIM_OK;
}
int sim_restore(sim_handle* h, const void* buf, size_t bytes) {
  if (!h || !buf || bytes < sizeof(snap_header)) return SIM_EINVAL;
  if (h->tick != 0 || !h->ops.empty()) return SIM_ESTATE;
  Dev& d = h->d;
  snap_header hd;
  memcpy(&hd, buf, sizeof hd);
  if (hd.magic != SNAP_MAGIC || hd.abi != SIM_ABI_VERSION || memcmp(&hd.cfg, &h->cfg, sizeof(sim_config))) return SIM_EINVAL;
  // ---- pass 1: validate the header and every section length before anything is touched ----
  if (hd.n_slots > d.A) return SIM_EINVAL;                          // reap_run / pp_merge walk a < n_slots
  if ((size_t)hd.n_pending_ops > bytes / sizeof(OpEnt)) return SIM_EINVAL;  // sizes a host allocation
  size_t len[SNAP_SECTIONS];
  snap_lengths(h, len);
  len[12] = (size_t)hd.n_pending_ops * sizeof(OpEnt);
  const uint8_t* sec[SNAP_SECTIONS];
  {
    const uint8_t* in = (const uint8_t*)buf + sizeof hd;
    const uint8_t* end = (const uint8_t*)buf + bytes;
    for (int i = 0; i < SNAP_SECTIONS; ++i) {
      uint64_t n;
      if ((size_t)(end - in) < 8) return SIM_EINVAL;
      memcpy(&n, in, 8); in += 8;
      if (n != len[i] || (size_t)(end - in) < n) return SIM_EINVAL;
      sec[i] = in;
      in += n;
    }
  }
  uint4* inbox_dst = (d.sharded && !d.rfan) ? h->rbuf[(hd.tick + 1) & 1] : h->inbox_mat;
  if ((len[2] && !inbox_dst) || (d.sharded && d.rfan && !h->bound)) return SIM_ESTATE;  // sharded: bind the exchange buffers first
  {  // slot maps must be consistent with n_slots (they index the view) and with each other (the recycling scan and
     // ensure_slot go from a slot to its subject and back)
    const u32* so = (const u32*)sec[6];
    const u32* sj = (const u32*)sec[7];
    for (u32 i = 0; i < d.N; ++i)
      if (so[i] != NOSLOT && (so[i] >= (h->dense ? d.A : hd.n_slots) || sj[so[i]] != i)) return SIM_EINVAL;
    for (u32 a = 0; a < d.A; ++a)
      if (sj[a] != NOSLOT && (sj[a] >= d.N || so[sj[a]] != a)) return SIM_EINVAL;
  }
  {  // the pending schedule goes straight to ops_kernel: the same checks sim_inject applies
    const OpEnt* po = (const OpEnt*)sec[12];
    for (u32 i = 0; i < hd.n_pending_ops; ++i) {
      OpEnt e;
      memcpy(&e, po + i, sizeof e);
      if (op_validate(d.N, e.op, e.node, e.a, e.b) != SIM_OK) return SIM_EINVAL;
    }
  }
  for (u32 j = 0; j < SIM_QT; ++j)  // the kernel loops over n_ids and shifts by the class
    if (((const u32*)sec[14])[(size_t)j * SIM_QF_WORDS + 1] > SIM_QF_IDS) return SIM_EINVAL;
  for (u32 i = 0; i < d.N; ++i)
    if (sec[15][i] >= SIM_TAG_CLASSES) return SIM_EINVAL;
  // ---- pass 2: device copies (the handle's host state is committed only after they succeed) ----
  hipStream_t s = h->stream;
  void* tmp_rows = nullptr;
  void* tmp_queue = nullptr;
  int rc = SIM_OK;
#define RCHECK(x) do { if (rc == SIM_OK && (x) != hipSuccess) rc = SIM_EDEVICE; } while (0)
  if (hipMalloc(&tmp_rows, std::max<size_t>(len[0], 16)) != hipSuccess) rc = SIM_ENOMEM;
  if (rc == SIM_OK && hipMalloc(&tmp_queue, std::max<size_t>(len[1], 16)) != hipSuccess) rc = SIM_ENOMEM;
  auto up = [&](void* dst, int i) { if (len[i]) RCHECK(hipMemcpyAsync(dst, sec[i], len[i], hipMemcpyHostToDevice, s)); };
  if (rc == SIM_OK) {
    up(tmp_rows, 0); up(tmp_queue, 1); up(inbox_dst, 2);
    if (rc == SIM_OK) rc = split_copy(h, d.view, d.vtail, d.vtail, (void*)sec[3], false);
    if (rc == SIM_OK) rc = split_copy(h, d.ering, d.etail, d.etail, (void*)sec[4], false);
    if (rc == SIM_OK) rc = split_copy(h, d.qring, d.qtail, d.qtail, (void*)sec[5], false);
    up(d.slot_of, 6); up(d.subject_of, 7); up(h->d_base, 8); up(d.upmap, 9); up(d.qtab, 10); up(d.qbits, 11);
    up(QFILT(d), 14); up(TAGCLASS(d), 15);
    // canonical rows / queue -> packed row groups, sort keys + slot-stable payloads
    RCHECK(hipMemsetAsync(d.R2, 0, (size_t)d.Nl * 16, s));
    if (rc == SIM_OK) {
      restore_queue_kernel<<<grid_for(d.Nl), BLOCK, 0, s>>>(d, (const uint4*)tmp_queue);
      restore_rows_kernel<<<grid_for(d.Nl), BLOCK, 0, s>>>(d, (const u64*)tmp_rows);
      if (!d.sharded || d.rfan) {  // the packets in flight go back to their senders (one cell per slot)
        TickP p;
        tickp_make(&p, &h->cfg, hd.tick ? hd.tick - 1 : 0);
        // (random fan-out: back to their places in their targets' rows — the graph of the tick they were sent in is a
        // function of (seed, tick) and is built again first)
        // (random fan-out: the graph of the packets in flight is a function of (seed, tick - 1): sim_step_begin builds it again)
        unmaterialize_kernel<<<grid_for(d.Nl), BLOCK, 0, s>>>(d, p, (u32)(hd.tick & 1), hd.tick ? 1u : 0u, h->inbox_mat);
      }
      RCHECK(hipGetLastError());
    }
    RCHECK(hipStreamSynchronize(s));
  }
#undef RCHECK
  if (tmp_rows) (void)hipFree(tmp_rows);
  if (tmp_queue) (void)hipFree(tmp_queue);
  if (rc != SIM_OK) return rc;
  h->tick = hd.tick;
  h->mat_tick = (d.sharded && !d.rfan) ? ~0ull : hd.tick;  // the image's inbox section is what sits in inbox_mat
  h->n_slots = hd.n_slots;
  if (hd.tick > 0) tickp_make(&h->prev, &h->cfg, hd.tick - 1);  // the parameters the packets in flight were sent with
  memcpy(h->slot_of.data(), sec[6], len[6]);
  memcpy(h->subject_of.data(), sec[7], len[7]);
  memcpy(h->base.data(), sec[8], len[8]);
  h->walk.clear();
  for (u32 subj = 0; subj < d.N; ++subj)
    if (h->slot_of[subj] != NOSLOT) h->walk.push_back(h->slot_of[subj]);
  if (!h->walk.empty()) HCHECK(hipMemcpy(d.walk, h->walk.data(), h->walk.size() * 4, hipMemcpyHostToDevice));
  h->ops.assign(hd.n_pending_ops, OpEnt{0, 0, 0, 0, 0});
  if (len[12]) memcpy(h->ops.data(), sec[12], len[12]);
  memcpy(h->alloc_tick.data(), sec[13], len[13]);
  memcpy(h->qfilt.data(), sec[14], len[14]);
  h->n_alloc = (u32)h->walk.size();
  h->ops_dropped = hd.ops_dropped; h->slots_recycled = hd.slots_recycled;
  h->recycle_at = 0xFFFFFFFFu;
  h->pp_done_at = 0xFFFFFFFFu;
  h->op_cursor = 0;
  if (h->rf_stream) HCHECK(hipStreamSynchronize(h->rf_stream));  // a build of the run that is being replaced may still be writing the scratch
  for (int i = 0; i < 3; ++i) h->rf_q[i] = ~0ull;
  if (h->xflag) *h->xflag = 0;
  if (d.sharded && d.rfan && hd.tick > 0) {
    // the packets in flight are back in their senders' cells: packed again — the host runs the round's exchange once more
    // before the next tick (SIM_XCHG_PACKED)
    int prc = rfx_pack(h, hd.tick - 1);
    if (prc) return prc;
    HCHECK(hipStreamSynchronize(h->stream));
  }
  return SIM_OK;
}
int sim_query_status(sim_handle* h, uint32_t qid, uint64_t* acks, uint64_t* responses, int* open) {
  if (!h || !acks || !responses || !open || !qid) return SIM_EINVAL;
  Dev& d = h->d;
  hipStream_t s = h->stream;
  u32 j = qid % SIM_QT;
  uint4 tj;
  HCHECK(hipMemcpyAsync(&tj, d.qtab + j, sizeof tj, hipMemcpyDeviceToHost, s));
  HCHECK(hipStreamSynchronize(s));
  if (tj.x != qid) return SIM_EINVAL;
  size_t words = ((size_t)d.N + 31) / 32;
  HCHECK(hipMemsetAsync(h->d_scratch + 10, 0, 16, s));
  query_count_kernel<<<grid_for(words), BLOCK, 0, s>>>(d.qbits + (size_t)j * 2 * words, words, h->d_scratch + 10);
  u64 r[2];
  HCHECK(hipMemcpyAsync(r, h->d_scratch + 10, 16, hipMemcpyDeviceToHost, s));
  HCHECK(hipStreamSynchronize(s));
  *acks = r[0];
  *responses = r[1];
  *open = (u32)h->tick <= tj.z;
  return SIM_OK;
}
int sim_query_responders(sim_handle* h, uint32_t qid, int which, uint32_t* out, uint32_t cap, uint32_t* n) {
  if (!h || !n || !qid || (which != 0 && which != 1) || (cap && !out)) return SIM_EINVAL;
  Dev& d = h->d;
  hipStream_t s = h->stream;
  u32 j = qid % SIM_QT;
  uint4 tj;
  HCHECK(hipMemcpyAsync(&tj, d.qtab + j, sizeof tj, hipMemcpyDeviceToHost, s));
  HCHECK(hipStreamSynchronize(s));
  if (tj.x != qid) return SIM_EINVAL;
  size_t words = ((size_t)d.N + 31) / 32;
  std::vector<u32> bits(words);  // one bit per node: the bitmap itself is the compact form (128 KiB at 1 Mi nodes)
  HCHECK(hipMemcpyAsync(bits.data(), d.qbits + ((size_t)j * 2 + (size_t)which) * words, words * 4, hipMemcpyDeviceToHost, s));
  HCHECK(hipStreamSynchronize(s));
  u32 k = 0;
  for (u32 g = d.shard0; g < d.shard0 + d.Nl; ++g)
    if ((bits[g >> 5] >> (g & 31)) & 1u) { if (k < cap) out[k] = g; ++k; }
  *n = k;
  return SIM_OK;
}
int sim_profile(sim_handle* h, int enable) {
  if (!h) return SIM_EINVAL;
  h->profiling = enable > 0 ? (u32)enable : 0u;
  h->prof_seq = 0;
  return SIM_OK;
}
int sim_profile_read(sim_handle* h, double* ms, uint64_t* launches) {
  if (!h || !ms || !launches) return SIM_EINVAL;
  HCHECK(hipStreamSynchronize(h->stream));
  double tot = 0.0;
  for (auto& pr : h->prof) {
    float t = 0.f;
    HCHECK(hipEventElapsedTime(&t, pr.first, pr.second));
    tot += t;
    (void)hipEventDestroy(pr.first);
    (void)hipEventDestroy(pr.second);
  }
  *ms = tot;
  *launches = h->prof.size();
  for (int i = 0; i < 3; ++i) h->sreq_wait[i] = nullptr;  // (the stream was synchronised above: every tick has finished)
  h->prof.clear();
  return SIM_OK;
}
int sim_profile_read_stats(sim_handle* h, double out_ms[3], uint64_t* launches) {
  if (!h || !out_ms || !launches) return SIM_EINVAL;
  HCHECK(hipStreamSynchronize(h->stream));
  double tot = 0.0, mn = 0.0, mx = 0.0;
  bool first = true;
  for (auto& pr : h->prof) {
    float t = 0.f;
    HCHECK(hipEventElapsedTime(&t, pr.first, pr.second));
    tot += t;
    mn = first ? t : std::min<double>(mn, t);
    mx = first ? t : std::max<double>(mx, t);
    first = false;
    (void)hipEventDestroy(pr.first);
    (void)hipEventDestroy(pr.second);
  }
  out_ms[0] = tot; out_ms[1] = mn; out_ms[2] = mx;
  *launches = h->prof.size();
  for (int i = 0; i < 3; ++i) h->sreq_wait[i] = nullptr;  // (the stream was synchronised above: every tick has finished)
  h->prof.clear();
  return SIM_OK;
}
int sim_cluster_stats_get(sim_handle* h, sim_cluster_stats* out) {
  if (!h || !out) return SIM_EINVAL;
  Dev& d = h->d;
  hipStream_t s = h->stream;
  u64* scr = nullptr;  // [10] the figures, then [workgroups][10] partial ones
  const u32 nwg = (u32)std::min<int>(grid_for(d.Nl), CSTAT_WG);
  if (hipMalloc((void**)&scr, (size_t)(1 + nwg) * 10 * 8) != hipSuccess) return SIM_ENOMEM;
  u64 r[10];
  hipError_t e = hipSuccess;
  {
    // sharded: the packets in flight sit in the receive buffer, [src][k][blk] = f * Nl cells as well
    cluster_stats_kernel<<<nwg, BLOCK, 0, s>>>(d, cur_inbox(h), scr + 10);
    cluster_stats_fold<<<1, dim3(64, 10), 0, s>>>(scr + 10, nwg, scr);
    e = hipMemcpyAsync(r, scr, sizeof r, hipMemcpyDeviceToHost, s);
  }
  u32 evc = 0;
  if (e == hipSuccess) e = hipMemcpyAsync(&evc, d.ev_count, 4, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  (void)hipFree(scr);
  HCHECK(e);
  out->up = r[0];
  for (int i = 0; i < 4; ++i) out->queued[i] = r[1 + i];
  out->overflow = r[5]; out->inbox_records = r[6]; out->failed = r[7]; out->left = r[8]; out->max_queue = r[9];
  out->ops_dropped = h->ops_dropped; out->slots_in_use = h->n_alloc; out->slots_recycled = h->slots_recycled;
  out->events_lost = h->events_lost + (evc > d.ev_cap ? evc - d.ev_cap : 0);
  return SIM_OK;
}
// bytes of the handle's send buffer (and of each receive buffer): the slabs [C][V dst][fp][M / V / C] of 48-byte packets — or,
// random fan-out on a shard, the V packed slabs (RfxL: header, count bytes, bucket totals, 64-byte cells)
static size_t xsend_bytes(const sim_handle* h) {
  const Dev& d = h->d;
  if (!d.sharded) return 0;
  return d.rfan ? (size_t)d.V * h->rfx.slab_u * 64u : (size_t)d.fp * d.M * sizeof(sim_packet);
}
int sim_exchange_bytes(const sim_handle* h, size_t* bytes) {
  if (!h || !bytes) return SIM_EINVAL;
  *bytes = xsend_bytes(h);
  return SIM_OK;
}
int sim_exchange_layout(const sim_handle* h, uint32_t* kind, uint32_t* planes, size_t* send_plane_bytes, size_t* recv_bytes) {
  if (!h || !kind || !planes || !send_plane_bytes || !recv_bytes) return SIM_EINVAL;
  const Dev& d = h->d;
  *kind = (d.sharded && d.rfan) ? SIM_XCHG_PACKED : SIM_XCHG_ALL_TO_ALL;
  *planes = 1;
  *send_plane_bytes = *recv_bytes = xsend_bytes(h);
  return SIM_OK;
}
int sim_bind_exchange2(sim_handle* h, void* send, void* recv0, void* recv1) {
  if (!h || !h->d.sharded || !send || !recv0 || !recv1) return SIM_EINVAL;
  h->d.xsend = (uint4*)send;
  h->rbuf[0] = (uint4*)recv0;
  h->rbuf[1] = (uint4*)recv1;
  h->d.xrecv = h->rbuf[(h->tick + 1) & 1];
  // (all zero: no packets — the bijection's empty cells, the random fan-out's empty slabs)
  HCHECK(hipMemsetAsync(send, 0, xsend_bytes(h), h->stream));
  HCHECK(hipMemsetAsync(recv0, 0, xsend_bytes(h), h->stream));
  if (recv1 != recv0) HCHECK(hipMemsetAsync(recv1, 0, xsend_bytes(h), h->stream));
  h->bound = true;
  return SIM_OK;
}
int sim_bind_exchange(sim_handle* h, void* send, void* recv) { return sim_bind_exchange2(h, send, recv, recv); }
int sim_bind_exchange3(sim_handle* h, void* send, size_t send_bytes, void* recv0, void* recv1, size_t recv_bytes) {
  if (!h || !h->d.sharded || send_bytes < xsend_bytes(h) || recv_bytes < xsend_bytes(h)) return SIM_EINVAL;
  return sim_bind_exchange2(h, send, recv0, recv1);
}
int sim_exchange_chunks(const sim_handle* h, uint32_t* chunks, size_t* bytes_per_chunk) {
  if (!h || !chunks || !bytes_per_chunk) return SIM_EINVAL;
  u32 C = h->cfg.chunks ? h->cfg.chunks : 1;
  *chunks = h->d.sharded ? C : 1;
  *bytes_per_chunk = xsend_bytes(h) / C;
  return SIM_OK;
}

// ---- the round's all-to-all over RCCL, issued by the library (include/serf_sim.h sim_exchange_*) ----
#define NCHECK(x)                                                                                          \
  do {                                                                                                     \
    ncclResult_t r_ = (x);                                                                                 \
    if (r_ != ncclSuccess) {                                                                               \
      fprintf(stderr, "serf_sim: %s failed: %s (%s:%d)\n", #x, ncclGetErrorString(r_), __FILE__, __LINE__); \
      return SIM_EDEVICE;                                                                                  \
    }                                                                                                      \
  } while (0)
static_assert(sizeof(ncclUniqueId) <= SIM_EXCHANGE_ID_BYTES, "the communicator id travels as SIM_EXCHANGE_ID_BYTES bytes");
int sim_exchange_unique_id(uint8_t* id_out) {
  if (!id_out) return SIM_EINVAL;
  ncclUniqueId id;
  NCHECK(ncclGetUniqueId(&id));
  memset(id_out, 0, SIM_EXCHANGE_ID_BYTES);
  memcpy(id_out, &id, sizeof id);
  return SIM_OK;
}
int sim_exchange_library(char* buf, size_t cap) {
  if (!buf || cap < 16) return SIM_EINVAL;
  int v = 0;
  NCHECK(ncclGetVersion(&v));  // major * 10000 + minor * 100 + patch
  snprintf(buf, cap, "RCCL %d.%d.%d", v / 10000, (v / 100) % 100, v % 100);
  return SIM_OK;
}
int sim_exchange_init(sim_handle* h, const uint8_t* id, uint32_t rank, uint32_t world) {
  if (!h || !id) return SIM_EINVAL;
  Dev& d = h->d;
  if (!d.sharded || world != h->cfg.shard_count || rank != h->cfg.shard_rank) return SIM_EINVAL;
  if (!h->bound || h->xcomm) return SIM_ESTATE;
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof uid);
  HCHECK(hipSetDevice(h->device));
  NCHECK(ncclCommInitRank(&h->xcomm, (int)world, uid, (int)rank));
  HCHECK(hipStreamCreateWithFlags(&h->xstream, hipStreamNonBlocking));
  HCHECK(hipEventCreateWithFlags(&h->xev_go, hipEventDisableTiming));
  HCHECK(hipEventCreateWithFlags(&h->xev_done, hipEventDisableTiming));
  h->xworld = world;
  return SIM_OK;
}
int sim_exchange_chunk(sim_handle* h, uint32_t chunk) {
  if (!h) return SIM_EINVAL;
  if (!h->xcomm) return SIM_ESTATE;
  Dev& d = h->d;
  const u32 C = h->cfg.chunks ? h->cfg.chunks : 1, V = h->xworld;
  if (chunk >= C) return SIM_EINVAL;
  // (called behind sim_step_chunk(chunk), before or after sim_step_end: the packets sent during tick t land in recv[t & 1])
  const u64 t = h->in_tick ? h->tick : h->tick - 1;
  const size_t chunk_bytes = xsend_bytes(h) / C, slab = chunk_bytes / V;  // (random fan-out: the V packed slabs, one chunk)
  const uint8_t* send = reinterpret_cast<const uint8_t*>(d.xsend) + (size_t)chunk * chunk_bytes;
  uint8_t* recv = reinterpret_cast<uint8_t*>(h->rbuf[t & 1]) + (size_t)chunk * chunk_bytes;
  // the exchange stream waits for what the handle's stream holds now — this chunk's launch —, not for the chunks after it
  HCHECK(hipEventRecord(h->xev_go, h->stream));
  HCHECK(hipStreamWaitEvent(h->xstream, h->xev_go, 0));
  NCHECK(ncclGroupStart());
  for (u32 p = 0; p < V; ++p) {
    NCHECK(ncclSend(send + (size_t)p * slab, slab, ncclUint8, (int)p, h->xcomm, h->xstream));
    NCHECK(ncclRecv(recv + (size_t)p * slab, slab, ncclUint8, (int)p, h->xcomm, h->xstream));
  }
  NCHECK(ncclGroupEnd());
  h->xpending = true;
  return SIM_OK;
}
int sim_exchange_wait(sim_handle* h) {
  if (!h) return SIM_EINVAL;
  if (!h->xcomm) return SIM_ESTATE;
  if (h->xpending) {
    HCHECK(hipEventRecord(h->xev_done, h->xstream));
    HCHECK(hipStreamWaitEvent(h->stream, h->xev_done, 0));
    h->xpending = false;
  }
  return SIM_OK;
}

// Test hook (host arithmetic only, no device): the fan-out map of `tick` for global node `gid` — its targets and the
// nodes whose packets land on it, both through the general forms the support kernels use.  Returns feff.
int sim_t_fanmap(const sim_config* cfg, uint64_t tick, uint32_t gid, uint32_t* targets, uint32_t* sources) {
  if (!cfg || !targets || !sources || !cfg->vshards || gid >= cfg->n_nodes) return SIM_EINVAL;
  TickP p;
  tickp_make(&p, cfg, tick);
  u32 g = gid / p.M, ll = gid - g * p.M;
  for (u32 k = 0; k < p.feff; ++k) {
    u32 h, t, sg, sl;
    fan_target_g(p, g, ll, k, h, t);
    targets[k] = h * p.M + t;
    fan_source_g(p, p.off[k], p.rot[k], p.rho[k], g, ll, k, sg, sl);
    sources[k] = sg * p.M + sl;
  }
  return (int)p.feff;
}

}  // extern "C"
