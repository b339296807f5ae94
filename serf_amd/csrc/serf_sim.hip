// serf_sim.hip — MI355X (gfx950) implementation of include/serf_sim.h.
//
// One simulated node per lane.  Per gossip tick each lane
//   1. streams its own row (3 Lamport clocks, state bits) and its retransmit queue (16 x 16 B,
//      SoA [slot][node] so a wave's loads are 1 KiB contiguous) out of HBM,
//   2. reads the fan-out packets addressed to it (inbox[k][node], 64 B each, coalesced),
//   3. runs every piggyback record through the serf-core handlers (Lamport witness, join/leave
//      intent vs. status_time, user-event / query de-dup rings) against its column of the
//      slot-major view table (view[slot][node]) — the rebroadcast decision of
//      serf-core/src/serf/delegate.rs:157-315 and serf/base.rs:750-1572,
//   4. keeps its TransmitLimitedQueue sorted in registers (static-index insertion / merge
//      networks, no scratch), drains `fanout` packets of SIM_P records from it
//      (delegate.rs:317-384, memberlist-core App. B.1) and
//   5. pushes each packet into the inbox cell of the peer chosen by this tick's fixed-point-free
//      pseudo-random bijection — exactly one writer per cell, so no atomics and no ordering
//      ambiguity (DESIGN.md SIMSPEC).
// Integer / byte work only: the roofline is HBM bandwidth, there is nothing for MFMA to do.
//
// There is deliberately no CPU fallback in this file: without a usable HIP device sim_create
// returns SIM_EDEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/serf_sim.h"

typedef uint32_t u32;
typedef uint64_t u64;

#define NOSLOT 0xFFFFFFFFu
#define STAMP_MASK 0x1FFFFFu
#define BLOCK 256

// ------------------------------------------------------------------------------------------------
// hashing / permutation (same arithmetic as the spec; host and device)
// ------------------------------------------------------------------------------------------------
__host__ __device__ static inline u64 mix64(u64 z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
enum { STREAM_PERM = 1, STREAM_OFF = 2, STREAM_ROT = 3, STREAM_LOSS = 4, STREAM_PROBE = 5 };
static inline u64 rng_base(u64 seed, u64 stream, u64 a) {
  return mix64(mix64(seed ^ (stream * 0xD6E8FEB86659FD93ull)) ^ a);
}
static inline u64 rng4(u64 seed, u64 stream, u64 a, u64 b) { return mix64(rng_base(seed, stream, a) ^ b); }

struct TickP {  // per-tick parameters, passed by value (lands in SGPRs)
  u64 tick;
  u64 loss_base;
  u32 M, mask, shift, feff, V, blk, loss_u32, first;  // first: tick 0 has no inbox yet
  u32 mul[3], add[3], imul[3];
  u32 off[SIM_MAX_FANOUT], rot[SIM_MAX_FANOUT];
  u32 prot[SIM_MAX_FANOUT];  // rot[] of the previous tick (sharded reads)
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
  p->feff = std::min(c->fanout, p->M - 1);
  for (int r = 0; r < 3; ++r) {
    u64 w = rng4(c->seed, STREAM_PERM, tick, (u64)r);
    p->mul[r] = (u32)w | 1u;
    p->add[r] = (u32)(w >> 32);
    p->imul[r] = modinv32(p->mul[r]);
  }
  for (u32 k = 0; k < p->feff; ++k) {
    u64 u = rng4(c->seed, STREAM_OFF, tick, k);
    u32 ck = 1u + (u32)(u % (u64)(p->M - 1));
    for (;;) {
      bool clash = false;
      for (u32 j = 0; j < k; ++j) clash |= (p->off[j] == ck);
      if (!clash) break;
      ck = ck % (p->M - 1) + 1u;
    }
    p->off[k] = ck;
    p->rot[k] = (u32)(rng4(c->seed, STREAM_ROT, tick, k) % (u64)p->V);
  }
  p->loss_base = rng_base(c->seed, STREAM_LOSS, tick);
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
__device__ static inline u32 sigma(const TickP& p, u32 x) {
  do x = perm_f(p, x); while (x >= p.M);
  return x;
}
__device__ static inline u32 sigma_inv(const TickP& p, u32 y) {
  do y = perm_fi(p, y); while (y >= p.M);
  return y;
}

// ------------------------------------------------------------------------------------------------
// device state
// ------------------------------------------------------------------------------------------------
struct Dev {
  // rows, SoA
  u64 *clock, *eclock, *qclock, *emin, *qmin;
  u32 *flags, *inc, *nknown, *nfailed, *nleft, *seqcnt, *overflow, *suspnext, *awareness, *probepend;
  uint4* queue;     // [Q][Nl]
  uint4* inbox[2];  // [f][Nl] packets of 4 x uint4 (local mode)
  uint4 *xsend, *xrecv;  // sharded mode: [V][f][blk] packets
  uint4* view;      // [A][Nl] entries of 2 x uint4
  uint4* ering;     // [Bev][Nl] buckets of 2 x uint4
  uint4* qring;     // [Bq][Nl]
  u32* slot_of;     // [N]
  u32 N, Nl, M, V, A, Bev, Bq, f, shard0, shard_rank, sharded, retransmit_mult;
};

// seqcnt packs next_seq (low 16) and the number of valid queue entries (high 16)
__device__ static inline u32 digits10(u32 n) {
  u32 d = 0;
  d += n >= 1u; d += n >= 10u; d += n >= 100u; d += n >= 1000u; d += n >= 10000u;
  d += n >= 100000u; d += n >= 1000000u; d += n >= 10000000u; d += n >= 100000000u;
  d += n >= 1000000000u;
  return d;
}

struct Node {  // one node's state in registers
  u64 clock, eclock, qclock;
  u32 flags, nknown, nfailed, nleft, next_seq, overflow, sc0;
  uint4 q[SIM_Q];  // {key, meta, val.lo, val.hi}, sorted by meta; empty = meta 0xFFFFFFFF
};

#define QEMPTY make_uint4(0u, SIM_META_EMPTY, 0u, 0u)

__device__ static inline u32 kind_class(u32 kind) {
  return (kind == SIM_K_JOIN || kind == SIM_K_LEAVE) ? 1u : kind == SIM_K_QUERY ? 2u : kind == SIM_K_EVENT ? 3u : 0u;
}
__host__ __device__ static inline u32 wire_meta(u32 kind, u32 flags, u32 len_bytes) {
  u32 len64 = (len_bytes + 15u) / 16u;
  if (len64 > 63u) len64 = 63u;
  return ((63u - len64) << 18) | ((kind & 15u) << 4) | (flags & 15u);
}

// queue_broadcast (memberlist TransmitLimitedQueue, App. B.1): sorted insertion with a fresh id;
// the entry that falls off the end of the Q-slot pool is counted as overflow.
__device__ static inline void q_insert(Node& n, u32 key, u32 wmeta, u64 val) {
  u32 kind = (wmeta >> 4) & 15u;
  u32 seq = n.next_seq++;
  u32 meta = (kind_class(kind) << 30) | (wmeta & SIM_META_WIRE_MASK) | ((1023u - seq) << 8);
  uint4 rec = make_uint4(key, meta, (u32)val, (u32)(val >> 32));
  if (n.q[SIM_Q - 1].y != SIM_META_EMPTY || meta > n.q[SIM_Q - 1].y) {
    // pool full (or the newcomer ranks last of a full pool): one record is dropped
    if (n.q[SIM_Q - 1].y != SIM_META_EMPTY) n.overflow++;
  }
#pragma unroll
  for (int i = SIM_Q - 1; i >= 1; --i) {
    bool below = n.q[i - 1].y > meta;  // predecessor sorts after the newcomer => shift it right
    bool here = !below && n.q[i].y > meta;
    uint4 v = below ? n.q[i - 1] : (here ? rec : n.q[i]);
    n.q[i] = v;
  }
  if (n.q[0].y > meta) n.q[0] = rec;
}

// compare-and-swap on the drain key
__device__ static inline void cas(uint4& a, uint4& b) {
  bool sw = a.y > b.y;
  uint4 lo = sw ? b : a, hi = sw ? a : b;
  a = lo;
  b = hi;
}

// get_broadcasts for one packet: the first SIM_P entries in drain order, transmits+1, drop at the
// retransmit limit, then restore the sorted order (4-sort + bitonic merge, static indices only).
__device__ static inline void q_emit(Node& n, u32 limit, uint4 (&pk)[SIM_P]) {
#pragma unroll
  for (int p = 0; p < (int)SIM_P; ++p) {
    uint4 e = n.q[p];
    bool valid = e.y != SIM_META_EMPTY;
    pk[p] = valid ? make_uint4(e.x, e.y & SIM_META_WIRE_MASK, e.z, e.w) : make_uint4(0, 0, 0, 0);
    u32 t = ((e.y >> 24) & 0x3Fu) + 1u;
    bool drop = t >= limit;
    uint4 bumped = make_uint4(e.x, (e.y & ~(0x3Fu << 24)) | (t << 24), e.z, e.w);
    n.q[p] = valid ? (drop ? QEMPTY : bumped) : e;
  }
  // sort the 4 touched entries
  cas(n.q[0], n.q[1]); cas(n.q[2], n.q[3]); cas(n.q[0], n.q[2]); cas(n.q[1], n.q[3]); cas(n.q[1], n.q[2]);
  // bitonic sequence: q[4..15] ascending followed by the 4 touched entries descending
  uint4 s[SIM_Q];
#pragma unroll
  for (int i = 0; i < 12; ++i) s[i] = n.q[i + 4];
  s[12] = n.q[3]; s[13] = n.q[2]; s[14] = n.q[1]; s[15] = n.q[0];
#pragma unroll
  for (int d = 8; d >= 1; d >>= 1) {
#pragma unroll
    for (int i = 0; i < SIM_Q; ++i)
      if ((i & d) == 0) cas(s[i], s[i + d]);
  }
#pragma unroll
  for (int i = 0; i < SIM_Q; ++i) n.q[i] = s[i];
}

// rare: renumber the queue ids when the 10-bit id space is nearly used up
__device__ static void q_renorm(Node& n) {
  u32 cnt = 0;
  uint4 o[SIM_Q];
#pragma unroll
  for (int i = 0; i < SIM_Q; ++i) o[i] = n.q[i];
#pragma unroll
  for (int i = 0; i < SIM_Q; ++i) {
    bool vi = o[i].y != SIM_META_EMPTY;
    u32 si = SIM_META_SEQ(o[i].y), rank = 0;
#pragma unroll
    for (int j = 0; j < SIM_Q; ++j) rank += (o[j].y != SIM_META_EMPTY && SIM_META_SEQ(o[j].y) < si) ? 1u : 0u;
    if (vi) {
      n.q[i].y = (o[i].y & ~(0x3FFu << 8)) | ((1023u - rank) << 8);
      cnt++;
    }
  }
  n.next_seq = cnt;
}

// ---- row / queue load-store -------------------------------------------------------------------
__device__ static inline void node_load(const Dev& d, u32 l, Node& n) {
  n.clock = d.clock[l]; n.eclock = d.eclock[l]; n.qclock = d.qclock[l];
  n.flags = d.flags[l]; n.nknown = d.nknown[l]; n.nfailed = d.nfailed[l]; n.nleft = d.nleft[l];
  n.overflow = d.overflow[l];
  u32 sc = d.seqcnt[l];
  n.sc0 = sc;
  n.next_seq = sc & 0xFFFFu;
  u32 cnt = sc >> 16;
#pragma unroll
  for (int i = 0; i < SIM_Q; ++i) n.q[i] = ((u32)i < cnt) ? d.queue[(size_t)i * d.Nl + l] : QEMPTY;
}
__device__ static inline void node_store(const Dev& d, u32 l, const Node& n, const Node& o) {
  if (n.clock != o.clock) d.clock[l] = n.clock;
  if (n.eclock != o.eclock) d.eclock[l] = n.eclock;
  if (n.qclock != o.qclock) d.qclock[l] = n.qclock;
  if (n.flags != o.flags) d.flags[l] = n.flags;
  if (n.nknown != o.nknown) d.nknown[l] = n.nknown;
  if (n.nfailed != o.nfailed) d.nfailed[l] = n.nfailed;
  if (n.nleft != o.nleft) d.nleft[l] = n.nleft;
  if (n.overflow != o.overflow) d.overflow[l] = n.overflow;
  u32 cnt = 0;
#pragma unroll
  for (int i = 0; i < SIM_Q; ++i) {
    cnt += n.q[i].y != SIM_META_EMPTY;
    uint4 a = n.q[i], b = o.q[i];
    if (a.x != b.x || a.y != b.y || a.z != b.z || a.w != b.w) d.queue[(size_t)i * d.Nl + l] = a;
  }
  u32 sc = (n.next_seq & 0xFFFFu) | (cnt << 16);
  if (sc != o.sc0) d.seqcnt[l] = sc;
}

// ---- handlers ----------------------------------------------------------------------------------
struct Ctx {
  const Dev& d;
  u32 l, gid;
  u32 stamp;  // tick & STAMP_MASK
};

__device__ static inline void witness(u64& c, u64 t) {  // types/clock.rs:155-172
  if (t >= c) c = t + 1;
}
__device__ static inline u32 vb_set_status(u32 b, u32 s) { return (b & ~(7u << 1)) | ((s & 7u) << 1); }
__device__ static inline u32 vb_set_intent(u32 b, u32 t) { return (b & ~(3u << 6)) | ((t & 3u) << 6); }
__device__ static inline u32 vb_set_stamp(u32 b, u32 st) { return (b & 0x7FFu) | (st << 11); }

__device__ static inline uint4* view_ptr(const Ctx& c, u32 subject) {
  if (subject >= c.d.N) return nullptr;
  u32 a = c.d.slot_of[subject];
  if (a == NOSLOT) return nullptr;
  return c.d.view + ((size_t)a * c.d.Nl + c.l) * 2;
}
#define E_LTIME(e) ((u64)(e).x | ((u64)(e).y << 32))
#define E_SET_LTIME(e, t) ((e).x = (u32)(t), (e).y = (u32)((t) >> 32))

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
__device__ static inline void erase_member(Node& n, uint4* p, const uint4& e) {
  u32 st = SIM_VB_STATUS(e.w);
  if (st == SIM_STATUS_FAILED && n.nfailed) n.nfailed--;
  if (st == SIM_STATUS_LEFT && n.nleft) n.nleft--;
  p[0] = make_uint4(0, 0, 0, 0);
  p[1] = make_uint4(0, 0, 0, 0);
  if (n.nknown) n.nknown--;
}
// handle_node_join_intent: base.rs:1338-1373
__device__ static bool handle_join_intent(const Ctx& c, Node& n, u32 subject, u64 ltime) {
  witness(n.clock, ltime);
  uint4* p = view_ptr(c, subject);
  if (!p) return false;
  uint4 e = p[0];
  if (e.w & SIM_VB_KNOWN) {
    if (ltime <= E_LTIME(e)) return false;
    E_SET_LTIME(e, ltime);
    if (SIM_VB_STATUS(e.w) == SIM_STATUS_LEAVING) e.w = vb_set_status(e.w, SIM_STATUS_ALIVE);
    p[0] = e;
    return true;
  }
  bool rb = upsert_intent(e, 1, ltime, c.stamp);
  if (rb) p[0] = e;
  return rb;
}
// broadcast_join: base.rs:381-397
__device__ static void broadcast_join(const Ctx& c, Node& n, u64 ltime) {
  witness(n.clock, ltime);
  handle_join_intent(c, n, c.gid, ltime);
  q_insert(n, c.gid, wire_meta(SIM_K_JOIN, 0, 16), ltime);
}
// handle_node_leave_intent: base.rs:1442-1572
__device__ static bool handle_leave_intent(const Ctx& c, Node& n, u32 subject, u64 ltime, bool prune) {
  u32 state = SIM_RF_STATE(n.flags);
  witness(n.clock, ltime);
  uint4* p = view_ptr(c, subject);
  if (!p) return false;
  uint4 e = p[0];
  if (!(e.w & SIM_VB_KNOWN)) {
    bool rb = upsert_intent(e, 2, ltime, c.stamp);
    if (rb) p[0] = e;
    return rb;
  }
  if (ltime <= E_LTIME(e)) return false;
  if (subject == c.gid && state == SIM_SERF_ALIVE) {  // refute: base.rs:1470-1480
    broadcast_join(c, n, n.clock);
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
  } else {
    e.w = vb_set_status(e.w, SIM_STATUS_LEAVING);
  }
  if (prune && rb) erase_member(n, p, e);  // handle_prune: base.rs:1628-1653
  else p[0] = e;
  return rb;
}
// handle_user_event: base.rs:750-837 (quirk U1 kept)
__device__ static bool handle_user_event(const Ctx& c, Node& n, u32 key, u64 ltime) {
  witness(n.eclock, ltime);
  if (ltime < c.d.emin[c.l]) return false;
  u64 B = c.d.Bev, cur = n.eclock;
  if (cur > B && ltime < cur - B) return false;
  u32 idx = (u32)(ltime % B);
  uint4* p = c.d.ering + ((size_t)idx * c.d.Nl + c.l) * 2;
  uint4 b0 = p[0];
  if (b0.z) {  // bucket present: keys[0] != 0
    uint4 b1 = p[1];
    u32 k[SIM_C] = {b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
    u32 cnt = 0;
    bool dup = false;
#pragma unroll
    for (int i = 0; i < (int)SIM_C; ++i) {
      dup |= (k[i] == key);  // key != 0, so empty slots never match
      cnt += k[i] != 0;
    }
    if (dup) return false;
    if (cnt == SIM_C) { n.overflow++; return false; }
    if (cnt == 1) b0.w = key;
    else if (cnt == 2) b1.x = key;
    else if (cnt == 3) b1.y = key;
    else if (cnt == 4) b1.z = key;
    else b1.w = key;
    if (cnt == 1) p[0] = b0; else p[1] = b1;
  } else {
    p[0] = make_uint4((u32)ltime, (u32)(ltime >> 32), key, 0);
  }
  return true;
}
// handle_query, de-dup part: base.rs:972-1073 (quirks Q1, Q2 kept)
__device__ static bool handle_query(const Ctx& c, Node& n, u32 id, u64 ltime, u32 flags) {
  witness(n.qclock, ltime);
  if (ltime < c.d.qmin[c.l]) return false;
  u64 cur = n.qclock, qt = c.d.Bq;
  if (cur > qt && qt < cur - qt) return false;
  u32 idx = (u32)(ltime % qt);
  uint4* p = c.d.qring + ((size_t)idx * c.d.Nl + c.l) * 2;
  uint4 b0 = p[0];
  if (b0.z) {
    uint4 b1 = p[1];
    u32 k[SIM_C] = {b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
    bool same = (E_LTIME(b0) == ltime);
    u32 cnt = 0;
    bool dup = false;
#pragma unroll
    for (int i = 0; i < (int)SIM_C; ++i) {
      dup |= (same && k[i] == id);
      cnt += k[i] != 0;
    }
    if (dup) return false;
    if (cnt == SIM_C) { n.overflow++; return false; }
    if (cnt == 1) b0.w = id;
    else if (cnt == 2) b1.x = id;
    else if (cnt == 3) b1.y = id;
    else if (cnt == 4) b1.z = id;
    else b1.w = id;
    if (cnt == 1) p[0] = b0; else p[1] = b1;
  } else {
    p[0] = make_uint4((u32)ltime, (u32)(ltime >> 32), id, 0);
  }
  return !(flags & SIM_F_NO_BROADCAST);
}
// SerfDelegate::notify_message: delegate.rs:183-300
__device__ static inline void dispatch(const Ctx& c, Node& n, const uint4& r) {
  u32 kind = SIM_META_KIND(r.y), flags = SIM_META_FLAGS(r.y);
  u64 val = (u64)r.z | ((u64)r.w << 32);
  bool rb = false;
  if (kind == SIM_K_LEAVE) rb = handle_leave_intent(c, n, r.x, val, flags & SIM_F_PRUNE);
  else if (kind == SIM_K_JOIN) rb = handle_join_intent(c, n, r.x, val);
  else if (kind == SIM_K_EVENT) rb = handle_user_event(c, n, r.x, val);
  else if (kind == SIM_K_QUERY) rb = handle_query(c, n, r.x, val, flags);
  if (rb) q_insert(n, r.x, r.y, val);  // re-queue the original message unchanged
}

// ------------------------------------------------------------------------------------------------
// the tick kernel
// ------------------------------------------------------------------------------------------------
template <bool SHARDED>
__global__ __launch_bounds__(BLOCK) void tick_kernel(Dev d, TickP tp, u32 cur) {
  u32 l = blockIdx.x * BLOCK + threadIdx.x;
  if (l >= d.Nl) return;
  u32 gid = d.shard0 + l;
  u32 g = gid / tp.M, ll = gid - g * tp.M;
  Ctx c{d, l, gid, (u32)tp.tick & STAMP_MASK};
  u32 flags0 = d.flags[l];
  bool up = flags0 & SIM_RF_UP;
  Node n, o;
  uint4 zero = make_uint4(0, 0, 0, 0);
  if (up) {
    node_load(d, l, n);
    o = n;
    if (n.next_seq > 1023u - 64u) q_renorm(n);
    if (!tp.first) {
      for (u32 k = 0; k < d.f; ++k) {
        const uint4* cell;
        if (SHARDED) {
          u32 b = l / tp.blk;
          u32 src = (d.shard_rank + b + tp.prot[k]) % tp.V;
          cell = d.xrecv + (((size_t)src * d.f + k) * tp.blk + (l - b * tp.blk)) * 4;
        } else {
          cell = d.inbox[cur] + ((size_t)k * d.Nl + l) * 4;
        }
        uint4 r0 = cell[0], r1 = cell[1], r2 = cell[2], r3 = cell[3];
        for (u32 p = 0; p < SIM_P; ++p) {
          uint4 r = p == 0 ? r0 : p == 1 ? r1 : p == 2 ? r2 : r3;
          if (SIM_META_KIND(r.y) != SIM_K_EMPTY) dispatch(c, n, r);
        }
      }
    }
  }
  u32 limit = up ? d.retransmit_mult * digits10(n.nknown) : 0;
  u32 sx = tp.feff ? sigma(tp, ll) : 0;
  for (u32 k = 0; k < tp.feff; ++k) {
    uint4 pk[SIM_P] = {zero, zero, zero, zero};
    if (up) {
      q_emit(n, limit, pk);
      if (tp.loss_u32 && (u32)(mix64(tp.loss_base ^ ((u64)gid * 4u + k)) >> 32) < tp.loss_u32)
        pk[0] = pk[1] = pk[2] = pk[3] = zero;
    }
    u32 y = sx + tp.off[k];
    if (y >= tp.M) y -= tp.M;
    u32 t = sigma_inv(tp, y);
    u32 b = t / tp.blk;
    u32 h = (g + tp.V - ((b + tp.rot[k]) % tp.V)) % tp.V;
    uint4* dst;
    if (SHARDED) dst = d.xsend + (((size_t)h * d.f + k) * tp.blk + (t - b * tp.blk)) * 4;
    else dst = d.inbox[cur ^ 1] + ((size_t)k * d.Nl + (size_t)h * tp.M + t) * 4;
    dst[0] = pk[0]; dst[1] = pk[1]; dst[2] = pk[2]; dst[3] = pk[3];
  }
  if (up) node_store(d, l, n, o);
}

// ------------------------------------------------------------------------------------------------
// operations (user-facing API acting on one node): one thread, a handful of ops per launch
// ------------------------------------------------------------------------------------------------
struct OpBatch {
  u32 n;
  u32 op[8], node[8], a[8], b[8];
};
__global__ void ops_kernel(Dev d, OpBatch ob, u64 tick, u32 has_alive) {
  if (threadIdx.x || blockIdx.x) return;
  for (u32 i = 0; i < ob.n; ++i) {
    u32 gid = ob.node[i];
    if (gid < d.shard0 || gid >= d.shard0 + d.Nl) continue;
    u32 l = gid - d.shard0;
    Ctx c{d, l, gid, (u32)tick & STAMP_MASK};
    Node n, o;
    node_load(d, l, n);
    o = n;
    if (n.next_seq > 1023u - 64u) q_renorm(n);
    bool up = n.flags & SIM_RF_UP;
    u32 a = ob.a[i], b = ob.b[i];
    switch (ob.op[i]) {
      case SIM_OP_USER_EVENT:  // api.rs:241-299
        if (up) {
          u64 lt = n.eclock;
          n.eclock++;
          handle_user_event(c, n, a, lt);
          q_insert(n, a, wire_meta(SIM_K_EVENT, 0, b), lt);
        }
        break;
      case SIM_OP_QUERY:  // base.rs:875-942
        if (up) {
          u64 lt = n.qclock;
          handle_query(c, n, a, lt, b);
          q_insert(n, a, wire_meta(SIM_K_QUERY, b, 32), lt);
        }
        break;
      case SIM_OP_LEAVE:  // api.rs:422-460
        if (up && SIM_RF_STATE(n.flags) == SIM_SERF_ALIVE) {
          n.flags = (n.flags & ~(3u << 1)) | (SIM_SERF_LEAVING << 1);
          u64 lt = n.clock;
          n.clock++;
          handle_leave_intent(c, n, gid, lt, false);
          if (has_alive) q_insert(n, gid, wire_meta(SIM_K_LEAVE, 0, 16), lt);
        }
        break;
      case SIM_OP_LEAVE_FINISH:  // api.rs:474-497
        if (SIM_RF_STATE(n.flags) == SIM_SERF_LEAVING) {
          n.flags = (n.flags & ~(3u << 1)) | (SIM_SERF_LEFT << 1);
          n.flags &= ~SIM_RF_UP;
        }
        break;
      case SIM_OP_JOIN:  // api.rs:318-364
        n.flags |= SIM_RF_UP;
        n.flags = (n.flags & ~(3u << 1)) | (SIM_SERF_ALIVE << 1);
        broadcast_join(c, n, n.clock);
        break;
      case SIM_OP_FORCE_LEAVE:  // base.rs:452-480
        if (up) {
          u64 lt = n.clock;
          handle_leave_intent(c, n, a, lt, b != 0);
          if (has_alive) q_insert(n, a, wire_meta(SIM_K_LEAVE, b ? SIM_F_PRUNE : 0, 16), lt);
        }
        break;
      case SIM_OP_CRASH: n.flags &= ~SIM_RF_UP; break;
      case SIM_OP_REVIVE: n.flags |= SIM_RF_UP; break;
      default: break;
    }
    node_store(d, l, n, o);
    __threadfence();  // the next op of this batch may touch the same node
  }
}

// ------------------------------------------------------------------------------------------------
// support kernels: fills, digest, members, convergence, stats
// ------------------------------------------------------------------------------------------------
__global__ void fill_u32(u32* p, size_t n, u32 v) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void fill_u64(u64* p, size_t n, u64 v) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void fill_u4(uint4* p, size_t n, uint4 v) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
// column a of the view := the subject's baseline entry
__global__ void fill_view_col(uint4* view, size_t Nl, u32 a, uint4 e0, uint4 e1) {
  for (size_t l = blockIdx.x * (size_t)blockDim.x + threadIdx.x; l < Nl; l += (size_t)gridDim.x * blockDim.x) {
    view[((size_t)a * Nl + l) * 2] = e0;
    view[((size_t)a * Nl + l) * 2 + 1] = e1;
  }
}
__global__ void init_dense_self(Dev d) {  // new_in's synthetic notify_join(local): self known, Alive @ 0
  for (size_t l = blockIdx.x * (size_t)blockDim.x + threadIdx.x; l < d.Nl; l += (size_t)gridDim.x * blockDim.x) {
    u32 gid = d.shard0 + (u32)l;
    d.view[((size_t)gid * d.Nl + l) * 2] = make_uint4(0, 0, 0, 1u | (SIM_STATUS_ALIVE << 1));
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
__global__ void digest_u32(const u32* w, size_t n, u64* out) {
  u64 acc = 0;
  for (size_t i = blockIdx.x * (size_t)BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLOCK) acc += dig((u64)w[i], i);
  block_sum_add(acc, out);
}
// rows (canonical AoS sim_row, 10 words per node) and queue (canonical [node][Q])
__global__ void digest_rows_queue(Dev d, u64* out_rows, u64* out_queue) {
  u64 ar = 0, aq = 0;
  for (size_t l = blockIdx.x * (size_t)BLOCK + threadIdx.x; l < d.Nl; l += (size_t)gridDim.x * BLOCK) {
    u32 sc = d.seqcnt[l];
    u64 w[10] = {d.clock[l], d.eclock[l], d.qclock[l], d.emin[l], d.qmin[l],
                 (u64)d.flags[l] | ((u64)d.inc[l] << 32), (u64)d.nknown[l] | ((u64)d.nfailed[l] << 32),
                 (u64)d.nleft[l] | ((u64)(sc & 0xFFFFu) << 32), (u64)d.overflow[l] | ((u64)d.suspnext[l] << 32),
                 (u64)d.awareness[l] | ((u64)d.probepend[l] << 32)};
    for (int i = 0; i < 10; ++i) ar += dig(w[i], l * 10 + i);
    u32 cnt = sc >> 16;
    for (u32 q = 0; q < SIM_Q; ++q) {
      uint4 e = q < cnt ? d.queue[(size_t)q * d.Nl + l] : QEMPTY;
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
    uint4 e = a == NOSLOT ? base[s * 2] : d.view[((size_t)a * d.Nl + obs_l) * 2];
    bool known = e.w & SIM_VB_KNOWN;
    st[s] = known ? (uint8_t)SIM_VB_STATUS(e.w) : (uint8_t)SIM_STATUS_NONE;
    lt[s] = known ? E_LTIME(e) : 0;
  }
}
__global__ void convergence_kernel(Dev d, const uint4* base, u32 kind, u32 key, u64 ltime, u64* out /*[2]*/) {
  u64 seen = 0, upc = 0;
  for (size_t l = blockIdx.x * (size_t)BLOCK + threadIdx.x; l < d.Nl; l += (size_t)gridDim.x * BLOCK) {
    if (!(d.flags[l] & SIM_RF_UP)) continue;
    upc++;
    if (kind == SIM_K_JOIN || kind == SIM_K_LEAVE) {
      u32 a = d.slot_of[key];
      uint4 e = a == NOSLOT ? base[(size_t)key * 2] : d.view[((size_t)a * d.Nl + l) * 2];
      seen += ((e.w & SIM_VB_KNOWN) && E_LTIME(e) >= ltime);
    } else {
      const uint4* ring = kind == SIM_K_EVENT ? d.ering : d.qring;
      u32 B = kind == SIM_K_EVENT ? d.Bev : d.Bq;
      const uint4* p = ring + ((size_t)(ltime % B) * d.Nl + l) * 2;
      uint4 b0 = p[0], b1 = p[1];
      seen += (b0.z == key) | (b0.w == key) | (b1.x == key) | (b1.y == key) | (b1.z == key) | (b1.w == key);
    }
  }
  block_sum_add(seen, out);
  block_sum_add(upc, out + 1);
}
__global__ void stats_kernel(Dev d, u32 l, sim_stats* o) {
  if (threadIdx.x || blockIdx.x) return;
  sim_stats s;
  memset(&s, 0, sizeof s);
  s.members = d.nknown[l]; s.failed = d.nfailed[l]; s.left = d.nleft[l];
  s.health_score = d.awareness[l];
  s.member_time = d.clock[l]; s.event_time = d.eclock[l]; s.query_time = d.qclock[l];
  u32 cnt = d.seqcnt[l] >> 16;
  for (u32 q = 0; q < cnt; ++q) {
    u32 cls = d.queue[(size_t)q * d.Nl + l].y >> 30;
    if (cls == 0) s.swim_queue++; else if (cls == 1) s.intent_queue++; else if (cls == 2) s.query_queue++; else s.event_queue++;
  }
  s.serf_state = SIM_RF_STATE(d.flags[l]); s.up = d.flags[l] & SIM_RF_UP; s.incarnation = d.inc[l];
  s.queue_overflow = d.overflow[l];
  *o = s;
}
__global__ void set_flag_bits(u32* flags, u32 l, u32 bits) {
  if (threadIdx.x == 0 && blockIdx.x == 0) flags[l] |= bits;
}

// ------------------------------------------------------------------------------------------------
// host side: the C ABI
// ------------------------------------------------------------------------------------------------
struct OpEnt {
  u64 tick;
  u32 op, node, a, b;
};

struct sim_handle {
  sim_config cfg;
  Dev d;
  u64 tick;
  u32 dense, n_slots;
  hipStream_t stream;
  std::vector<u32> slot_of, subject_of;
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
  if (c->fanout < 1 || c->fanout > SIM_MAX_FANOUT) return SIM_EINVAL;
  if (c->event_ring < 1 || c->query_ring < 1) return SIM_EINVAL;
  if (c->retransmit_mult * h_digits10(c->n_nodes) > 63u) return SIM_EINVAL;
  return SIM_OK;
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

extern "C" {

uint32_t sim_abi_version(void) { return SIM_ABI_VERSION; }
const char* sim_backend_name(void) { return "hip-gfx950"; }

int sim_destroy(sim_handle* h) {
  if (!h) return SIM_EINVAL;
  (void)hipStreamSynchronize(h->stream);
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
  (void)hipGetDevice(&h->device);
  Dev& d = h->d;
  memset(&d, 0, sizeof d);
  d.N = cfg->n_nodes; d.V = cfg->vshards; d.M = d.N / d.V;
  d.sharded = cfg->shard_count > 1;
  d.Nl = d.sharded ? d.M : d.N;
  d.shard0 = d.sharded ? cfg->shard_rank * d.M : 0;
  d.shard_rank = cfg->shard_rank;
  h->dense = (cfg->view_slots == 0 || cfg->view_slots >= d.N);
  d.A = h->dense ? d.N : cfg->view_slots;
  d.Bev = cfg->event_ring; d.Bq = cfg->query_ring; d.f = cfg->fanout;
  d.retransmit_mult = cfg->retransmit_mult;
  size_t Nl = d.Nl;
#define DA(ptr, n)                                   \
  if ((rc = dalloc(h, &(ptr), (n))) != SIM_OK) {     \
    sim_destroy(h);                                  \
    return rc;                                       \
  }
  DA(d.clock, Nl) DA(d.eclock, Nl) DA(d.qclock, Nl) DA(d.emin, Nl) DA(d.qmin, Nl)
  DA(d.flags, Nl) DA(d.inc, Nl) DA(d.nknown, Nl) DA(d.nfailed, Nl) DA(d.nleft, Nl) DA(d.seqcnt, Nl)
  DA(d.overflow, Nl) DA(d.suspnext, Nl) DA(d.awareness, Nl) DA(d.probepend, Nl)
  DA(d.queue, (size_t)SIM_Q * Nl)
  if (!d.sharded) { DA(d.inbox[0], (size_t)d.f * Nl * 4) DA(d.inbox[1], (size_t)d.f * Nl * 4) }
  DA(d.view, (size_t)d.A * Nl * 2)
  DA(d.ering, (size_t)d.Bev * Nl * 2)
  DA(d.qring, (size_t)d.Bq * Nl * 2)
  DA(d.slot_of, d.N)
  DA(h->d_base, (size_t)d.N * 2)
  DA(h->d_scratch, 16)
  DA(h->d_mst, d.N)
  DA(h->d_mlt, d.N)
  DA(h->d_stats, 1)
#undef DA
  bool joined = cfg->flags & SIM_CF_BASELINE_JOINED;
  hipStream_t s = h->stream;
  auto zero = [&](void* p, size_t bytes) { return hipMemsetAsync(p, 0, bytes, s); };
  HCHECK(zero(d.emin, Nl * 8)); HCHECK(zero(d.qmin, Nl * 8)); HCHECK(zero(d.inc, Nl * 4));
  HCHECK(zero(d.nfailed, Nl * 4)); HCHECK(zero(d.nleft, Nl * 4)); HCHECK(zero(d.seqcnt, Nl * 4));
  HCHECK(zero(d.overflow, Nl * 4)); HCHECK(zero(d.suspnext, Nl * 4)); HCHECK(zero(d.awareness, Nl * 4));
  HCHECK(zero(d.probepend, Nl * 4));
  if (!d.sharded) { HCHECK(zero(d.inbox[0], (size_t)d.f * Nl * 64)); HCHECK(zero(d.inbox[1], (size_t)d.f * Nl * 64)); }
  HCHECK(zero(d.view, (size_t)d.A * Nl * 32));
  HCHECK(zero(d.ering, (size_t)d.Bev * Nl * 32));
  HCHECK(zero(d.qring, (size_t)d.Bq * Nl * 32));
  fill_u64<<<grid_for(Nl), BLOCK, 0, s>>>(d.clock, Nl, joined ? 2 : 1);  // base.rs:196-205 (+ own join)
  fill_u64<<<grid_for(Nl), BLOCK, 0, s>>>(d.eclock, Nl, 1);
  fill_u64<<<grid_for(Nl), BLOCK, 0, s>>>(d.qclock, Nl, 1);
  fill_u32<<<grid_for(Nl), BLOCK, 0, s>>>(d.flags, Nl, SIM_RF_UP | (SIM_SERF_ALIVE << 1));
  fill_u32<<<grid_for(Nl), BLOCK, 0, s>>>(d.nknown, Nl, joined ? d.N : 1);
  fill_u4<<<grid_for((size_t)SIM_Q * Nl), BLOCK, 0, s>>>(d.queue, (size_t)SIM_Q * Nl, QEMPTY);
  // slot map + baseline
  h->slot_of.assign(d.N, NOSLOT);
  h->subject_of.assign(d.A, NOSLOT);
  sim_view b0;
  memset(&b0, 0, sizeof b0);
  if (joined) { b0.ltime = 1; b0.bits = 1u | (SIM_STATUS_ALIVE << 1); }
  h->base.assign(d.N, b0);
  uint4 e0 = make_uint4((u32)b0.ltime, (u32)(b0.ltime >> 32), b0.inc, b0.bits), e1 = make_uint4(0, 0, 0, 0);
  fill_view_col<<<grid_for(d.N), BLOCK, 0, s>>>(h->d_base, d.N, 0, e0, e1);  // d_base is one "column" of N entries
  if (h->dense) {
    h->n_slots = d.N;
    for (u32 i = 0; i < d.N; ++i) h->slot_of[i] = h->subject_of[i] = i;
    if (joined) {
      // every entry of the dense table = baseline
      size_t tot = (size_t)d.A * Nl;
      fill_view_col<<<grid_for(tot), BLOCK, 0, s>>>(d.view, tot, 0, e0, e1);
    } else {
      init_dense_self<<<grid_for(Nl), BLOCK, 0, s>>>(d);
    }
  } else {
    h->n_slots = 0;
  }
  HCHECK(hipMemcpyAsync(d.slot_of, h->slot_of.data(), (size_t)d.N * 4, hipMemcpyHostToDevice, s));
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

static int ensure_slot(sim_handle* h, u32 subject) {
  Dev& d = h->d;
  if (subject >= d.N) return SIM_EINVAL;
  if (h->slot_of[subject] != NOSLOT) return SIM_OK;
  if (h->n_slots >= d.A) return SIM_ENOSLOT;
  u32 a = h->n_slots++;
  h->slot_of[subject] = a;
  h->subject_of[a] = subject;
  const sim_view& b = h->base[subject];
  uint4 e0 = make_uint4((u32)b.ltime, (u32)(b.ltime >> 32), b.inc, b.bits);
  uint4 e1 = make_uint4(b.conf[0], b.conf[1], b.conf[2], b.conf[3]);
  fill_view_col<<<grid_for(d.Nl), BLOCK, 0, h->stream>>>(d.view, d.Nl, a, e0, e1);
  HCHECK(hipMemcpyAsync(d.slot_of + subject, &h->slot_of[subject], 4, hipMemcpyHostToDevice, h->stream));
  return SIM_OK;
}

int sim_inject(sim_handle* h, uint64_t tick, uint32_t op, uint32_t node, uint32_t a, uint32_t b) {
  if (!h || node >= h->d.N) return SIM_EINVAL;
  if (tick < h->tick) tick = h->tick;
  int rc = SIM_OK;
  switch (op) {
    case SIM_OP_USER_EVENT: if (!a) return SIM_EINVAL; if (b > 9 * 1024) return SIM_ETOOBIG; break;
    case SIM_OP_QUERY: if (!a) return SIM_EINVAL; break;
    case SIM_OP_LEAVE: case SIM_OP_JOIN: case SIM_OP_LEAVE_FINISH: rc = ensure_slot(h, node); break;
    case SIM_OP_FORCE_LEAVE: rc = ensure_slot(h, a); break;
    case SIM_OP_CRASH: case SIM_OP_REVIVE: break;
    default: return SIM_EINVAL;
  }
  if (rc) return rc;
  size_t pos = h->ops.size();
  h->ops.push_back(OpEnt{tick, op, node, a, b});
  while (pos > h->op_cursor && h->ops[pos - 1].tick > tick) { std::swap(h->ops[pos], h->ops[pos - 1]); --pos; }
  return SIM_OK;
}
int sim_join(sim_handle* h, uint32_t node, uint32_t peer) { return sim_inject(h, h ? h->tick : 0, SIM_OP_JOIN, node, peer, 0); }
int sim_leave(sim_handle* h, uint32_t node) {
  if (!h) return SIM_EINVAL;
  int rc = sim_inject(h, h->tick, SIM_OP_LEAVE, node, 0, 0);
  if (rc) return rc;
  return sim_inject(h, h->tick + h->cfg.leave_delay + 1, SIM_OP_LEAVE_FINISH, node, 0, 0);
}
int sim_force_leave(sim_handle* h, uint32_t node, uint32_t subject, int prune) {
  return sim_inject(h, h ? h->tick : 0, SIM_OP_FORCE_LEAVE, node, subject, prune ? 1u : 0u);
}
int sim_user_event(sim_handle* h, uint32_t node, uint32_t key, uint32_t len, int cc) {
  (void)cc;
  return sim_inject(h, h ? h->tick : 0, SIM_OP_USER_EVENT, node, key, len);
}
int sim_query(sim_handle* h, uint32_t node, uint32_t id, uint32_t flags) {
  return sim_inject(h, h ? h->tick : 0, SIM_OP_QUERY, node, id, flags);
}

int sim_step(sim_handle* h, uint32_t n_ticks) {
  if (!h) return SIM_EINVAL;
  Dev& d = h->d;
  if (d.sharded && !h->bound) return SIM_ESTATE;
  for (u32 it = 0; it < n_ticks; ++it) {
    TickP tp;
    tickp_make(&tp, &h->cfg, h->tick);
    for (u32 k = 0; k < SIM_MAX_FANOUT; ++k) tp.prot[k] = h->prev.rot[k];
    while (h->op_cursor < h->ops.size() && h->ops[h->op_cursor].tick <= h->tick) {
      OpBatch ob;
      memset(&ob, 0, sizeof ob);
      while (ob.n < 8 && h->op_cursor < h->ops.size() && h->ops[h->op_cursor].tick <= h->tick) {
        const OpEnt& e = h->ops[h->op_cursor++];
        ob.op[ob.n] = e.op; ob.node[ob.n] = e.node; ob.a[ob.n] = e.a; ob.b[ob.n] = e.b;
        ob.n++;
      }
      ops_kernel<<<1, 64, 0, h->stream>>>(d, ob, h->tick, d.N > 1 ? 1u : 0u);
    }
    int grid = (int)((d.Nl + BLOCK - 1) / BLOCK);
    u32 cur = (u32)(h->tick & 1);
    if (d.sharded) tick_kernel<true><<<grid, BLOCK, 0, h->stream>>>(d, tp, cur);
    else tick_kernel<false><<<grid, BLOCK, 0, h->stream>>>(d, tp, cur);
    h->prev = tp;
    h->tick++;
  }
  HCHECK(hipGetLastError());
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
  set_flag_bits<<<1, 64, 0, h->stream>>>(d.flags, obs - d.shard0, SIM_RF_WATCHED);
  return SIM_OK;
}
int sim_drain_events(sim_handle* h, sim_event* out, uint32_t cap, uint32_t* n) {
  (void)out; (void)cap;
  if (!h || !n) return SIM_EINVAL;
  *n = 0;  // event log of watched observers: not yet surfaced by the HIP path
  return SIM_OK;
}

static const uint4* cur_inbox(const sim_handle* h) {
  return h->d.sharded ? h->d.xrecv : h->d.inbox[h->tick & 1];
}
int sim_state_digest(sim_handle* h, uint64_t out[8]) {
  if (!h || !out) return SIM_EINVAL;
  Dev& d = h->d;
  hipStream_t s = h->stream;
  HCHECK(hipMemsetAsync(h->d_scratch, 0, 16 * 8, s));
  digest_rows_queue<<<grid_for(d.Nl), BLOCK, 0, s>>>(d, h->d_scratch + 0, h->d_scratch + 1);
  size_t nw;
  if (cur_inbox(h)) { nw = (size_t)d.f * d.Nl * 8; digest_flat<<<grid_for(nw), BLOCK, 0, s>>>((const u64*)cur_inbox(h), nw, h->d_scratch + 2); }
  nw = (size_t)d.A * d.Nl * 4; digest_flat<<<grid_for(nw), BLOCK, 0, s>>>((const u64*)d.view, nw, h->d_scratch + 3);
  nw = (size_t)d.Bev * d.Nl * 4; digest_flat<<<grid_for(nw), BLOCK, 0, s>>>((const u64*)d.ering, nw, h->d_scratch + 4);
  nw = (size_t)d.Bq * d.Nl * 4; digest_flat<<<grid_for(nw), BLOCK, 0, s>>>((const u64*)d.qring, nw, h->d_scratch + 5);
  digest_u32<<<grid_for(d.N), BLOCK, 0, s>>>(d.slot_of, d.N, h->d_scratch + 6);
  HCHECK(hipMemcpyAsync(out, h->d_scratch, 8 * 8, hipMemcpyDeviceToHost, s));
  HCHECK(hipStreamSynchronize(s));
  out[7] = 0;
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
    case SIM_ARR_INBOX: src = cur_inbox(h); n = (size_t)d.f * Nl * sizeof(sim_packet); break;
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
  if (which == SIM_ARR_ROWS) {
    std::vector<u64> c64(Nl);
    std::vector<u32> c32(Nl);
    sim_row* r = (sim_row*)buf;
    memset(r, 0, n);
#define G64(field, ptr) HCHECK(hipMemcpy(c64.data(), ptr, Nl * 8, hipMemcpyDeviceToHost)); for (size_t i = 0; i < Nl; ++i) r[i].field = c64[i];
#define G32(field, ptr) HCHECK(hipMemcpy(c32.data(), ptr, Nl * 4, hipMemcpyDeviceToHost)); for (size_t i = 0; i < Nl; ++i) r[i].field = c32[i];
    G64(clock, d.clock) G64(event_clock, d.eclock) G64(query_clock, d.qclock) G64(event_min, d.emin) G64(query_min, d.qmin)
    G32(flags, d.flags) G32(inc, d.inc) G32(n_known, d.nknown) G32(n_failed, d.nfailed) G32(n_left, d.nleft)
    G32(next_seq, d.seqcnt) G32(overflow, d.overflow) G32(susp_next, d.suspnext) G32(awareness, d.awareness) G32(probe_pending, d.probepend)
#undef G64
#undef G32
    for (size_t i = 0; i < Nl; ++i) r[i].next_seq &= 0xFFFFu;
    return SIM_OK;
  }
  if (which == SIM_ARR_QUEUE) {
    std::vector<sim_record> t((size_t)SIM_Q * Nl);
    std::vector<u32> sc(Nl);
    HCHECK(hipMemcpy(t.data(), d.queue, t.size() * sizeof(sim_record), hipMemcpyDeviceToHost));
    HCHECK(hipMemcpy(sc.data(), d.seqcnt, Nl * 4, hipMemcpyDeviceToHost));
    sim_record* o = (sim_record*)buf;
    for (size_t l = 0; l < Nl; ++l)
      for (u32 q = 0; q < SIM_Q; ++q) {
        if (q < (sc[l] >> 16)) o[l * SIM_Q + q] = t[(size_t)q * Nl + l];
        else { o[l * SIM_Q + q].key = 0; o[l * SIM_Q + q].meta = SIM_META_EMPTY; o[l * SIM_Q + q].val = 0; }
      }
    return SIM_OK;
  }
  if (!src) { memset(buf, 0, n); return SIM_OK; }
  HCHECK(hipMemcpy(buf, src, n, hipMemcpyDeviceToHost));
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

int sim_exchange_bytes(const sim_handle* h, size_t* bytes) {
  if (!h || !bytes) return SIM_EINVAL;
  *bytes = h->d.sharded ? (size_t)h->d.f * h->d.M * sizeof(sim_packet) : 0;
  return SIM_OK;
}
int sim_bind_exchange(sim_handle* h, void* send, void* recv) {
  if (!h || !h->d.sharded || !send || !recv) return SIM_EINVAL;
  h->d.xsend = (uint4*)send;
  h->d.xrecv = (uint4*)recv;
  size_t n = (size_t)h->d.f * h->d.M * sizeof(sim_packet);
  HCHECK(hipMemsetAsync(send, 0, n, h->stream));
  HCHECK(hipMemsetAsync(recv, 0, n, h->stream));
  h->bound = true;
  return SIM_OK;
}

}  // extern "C"
