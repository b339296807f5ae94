/*
 * serf_oracle.c — CPU ORACLE for the bulk SWIM/Serf gossip simulator.
 *
 * *** TEST INFRASTRUCTURE ONLY. ***  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library.  The product (serf_amd/csrc, HIP) never calls it.
 *
 * What this is: a plain-C restatement of the serf-core 0.5.1 state machine on the simulated hot
 * path, each function citing the reference file:line it follows (paths relative to
 * /root/reference/serf-core/src), driven by the tick loop of DESIGN.md ("SIMSPEC").
 *
 * Parity status:
 *   - serf-layer handlers (Lamport clock, join/leave intents, intent buffer, user-event and query
 *     de-dup rings, notify_join/notify_leave, push-pull merge, reaper, queue cap) are PINNED by the
 *     reference's own known-answer tests, restated in tests/test_oracle_kat.py.
 *   - memberlist-core 0.8.1 (TransmitLimitedQueue order/limit, gossip peer selection, probe and
 *     suspicion timing) is NOT in /root/reference: its published algorithm is restated from
 *     SURVEY.md App. B and is "parity unpinned".
 *
 * Build: make -C oracle   (gcc -O3 -march=native -std=c11 -fopenmp)
 */
#include "../include/serf_sim.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define API(name) osim_##name

/* Worker threads for the node loop: min(online CPUs, cgroup CPU quota), overridable with
 * ORACLE_THREADS.  (GPU boxes expose 256 CPUs under a 16-CPU quota; one OpenMP thread per visible
 * CPU would spend its life in contended barriers.) */
static int g_threads = 0;
static int oracle_threads(void) {
  if (g_threads) return g_threads;
  int n = 1;
#ifdef _OPENMP
  n = omp_get_num_procs();
#endif
  FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r");
  if (f) {
    long long q = 0, per = 0;
    if (fscanf(f, "%lld %lld", &q, &per) == 2 && q > 0 && per > 0) {
      int lim = (int)((q + per - 1) / per);
      if (lim >= 1 && lim < n) n = lim;
    }
    fclose(f);
  }
  const char* e = getenv("ORACLE_THREADS");
  if (e && atoi(e) > 0) n = atoi(e);
  if (n < 1) n = 1;
  g_threads = n;
  return n;
}
int osim_t_threads(void) { return oracle_threads(); }

/* =====================================================================================
 * Counter-based PRNG and the per-tick fan-out permutation (DESIGN.md SIMSPEC §2).  The reference
 * draws from OS entropy (base.rs:629, query.rs:399); the simulator replaces that with a keyed
 * hash of (seed, stream, tick, node) so the CPU and the GPU see identical sequences.
 * ===================================================================================== */
static inline uint64_t mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
enum { STREAM_PERM = 1, STREAM_OFF = 2, STREAM_ROT = 3, STREAM_LOSS = 4, STREAM_PROBE = 5 };
static inline uint64_t rng_base(uint64_t seed, uint64_t stream, uint64_t a) {
  return mix64(mix64(seed ^ (stream * 0xD6E8FEB86659FD93ull)) ^ a);
}
static inline uint64_t rng4(uint64_t seed, uint64_t stream, uint64_t a, uint64_t b) {
  return mix64(rng_base(seed, stream, a) ^ b);
}

typedef struct tickp {
  uint64_t tick;
  uint32_t M, nbits, mask, shift, feff, V, blk;
  uint32_t mul[3], add[3], imul[3];
  uint32_t off[SIM_MAX_FANOUT], rot[SIM_MAX_FANOUT];
  uint64_t loss_base;
  uint32_t loss_u32;
} tickp;

static uint32_t modinv32(uint32_t a) { /* a odd: Newton iteration mod 2^32 */
  uint32_t x = a;
  for (int i = 0; i < 5; ++i) x *= 2u - a * x;
  return x;
}
static uint32_t ceil_log2_u32(uint32_t m) {
  uint32_t b = 0;
  while (b < 32 && (1ull << b) < m) ++b;
  return b;
}

static void tickp_make(tickp* p, const sim_config* c, uint64_t tick) {
  memset(p, 0, sizeof *p);
  p->tick = tick;
  p->V = c->vshards;
  p->M = c->n_nodes / c->vshards;
  p->blk = p->M / p->V;
  p->nbits = ceil_log2_u32(p->M);
  if (p->nbits < 1) p->nbits = 1;
  p->mask = p->nbits >= 32 ? 0xFFFFFFFFu : ((1u << p->nbits) - 1u);
  p->shift = (p->nbits + 1) / 2;
  p->feff = c->fanout;
  if (p->M - 1 < p->feff) p->feff = p->M - 1;
  for (int r = 0; r < 3; ++r) {
    uint64_t w = rng4(c->seed, STREAM_PERM, tick, (uint64_t)r);
    p->mul[r] = (uint32_t)w | 1u;
    p->add[r] = (uint32_t)(w >> 32);
    p->imul[r] = modinv32(p->mul[r]);
  }
  for (uint32_t k = 0; k < p->feff; ++k) {
    uint64_t u = rng4(c->seed, STREAM_OFF, tick, k);
    uint32_t ck = 1u + (uint32_t)(u % (uint64_t)(p->M - 1));
    for (;;) {
      int clash = 0;
      for (uint32_t j = 0; j < k; ++j) clash |= (p->off[j] == ck);
      if (!clash) break;
      ck = ck % (p->M - 1) + 1u;
    }
    p->off[k] = ck;
    p->rot[k] = (uint32_t)(rng4(c->seed, STREAM_ROT, tick, k) % (uint64_t)p->V);
  }
  p->loss_base = rng_base(c->seed, STREAM_LOSS, tick);
  p->loss_u32 = c->loss_u32;
}

static inline uint32_t perm_f(const tickp* p, uint32_t x) {
  x = (x * p->mul[0] + p->add[0]) & p->mask;
  x ^= x >> p->shift;
  x = (x * p->mul[1] + p->add[1]) & p->mask;
  x ^= x >> p->shift;
  x = (x * p->mul[2] + p->add[2]) & p->mask;
  return x;
}
static inline uint32_t perm_fi(const tickp* p, uint32_t y) {
  y = ((y - p->add[2]) * p->imul[2]) & p->mask;
  y ^= y >> p->shift;
  y = ((y - p->add[1]) * p->imul[1]) & p->mask;
  y ^= y >> p->shift;
  y = ((y - p->add[0]) * p->imul[0]) & p->mask;
  return y;
}
static inline uint32_t sigma(const tickp* p, uint32_t x) { /* cycle-walking bijection on [0,M) */
  do x = perm_f(p, x); while (x >= p->M);
  return x;
}
static inline uint32_t sigma_inv(const tickp* p, uint32_t y) {
  do y = perm_fi(p, y); while (y >= p->M);
  return y;
}
/* k-th gossip target of the node whose sigma-image is sx, living in shard g. */
static inline void fan_target(const tickp* p, uint32_t g, uint32_t sx, uint32_t k, uint32_t* h,
                              uint32_t* lp) {
  uint32_t y = sx + p->off[k];
  if (y >= p->M) y -= p->M;
  uint32_t t = sigma_inv(p, y);
  uint32_t b = t / p->blk;
  *lp = t;
  *h = (g + p->V - ((b + p->rot[k]) % p->V)) % p->V;
}
static inline int pkt_lost(const tickp* p, uint32_t gid, uint32_t k) {
  if (!p->loss_u32) return 0;
  return (uint32_t)(mix64(p->loss_base ^ ((uint64_t)gid * 4u + k)) >> 32) < p->loss_u32;
}

/* =====================================================================================
 * State
 * ===================================================================================== */
#define NOSLOT 0xFFFFFFFFu
#define STAMP_MASK 0x1FFFFFu
#define MAX_PEND (2 * SIM_MAX_FANOUT * SIM_P + 4)

typedef struct sim_opent {
  uint64_t tick;
  uint32_t op, node, a, b;
} sim_opent;

struct sim_handle {
  sim_config cfg;
  uint32_t N, V, M, Nl, A, Bev, Bq, f, dense, shard0; /* Nl local nodes; shard0 = first global id */
  uint64_t tick;
  tickp prev; /* parameters of the tick that produced the current inbox (sharded reads) */
  sim_row* rows;        /* [Nl]            */
  sim_record* queue;    /* [Nl][Q], sorted */
  sim_packet* inbox[2]; /* local mode: [f][Nl]; current = tick & 1 */
  sim_packet *xsend, *xrecv; /* sharded mode: [V][f][blk]          */
  int own_x;
  sim_view* view;       /* [A][Nl]   */
  sim_bucket* ering;    /* [Bev][Nl] */
  sim_bucket* qring;    /* [Bq][Nl]  */
  uint32_t* slot_of;    /* [N] global subject -> slot */
  uint32_t* subject_of; /* [A] */
  uint32_t n_slots;
  sim_view* base;       /* [N] baseline entry per subject (non-dense) */
  sim_opent* ops;
  size_t n_ops, cap_ops, op_cursor;
  sim_event* events;
  size_t n_events, cap_events;
  uint32_t n_watched;
};
typedef struct sim_handle osim;

static inline uint32_t digits10(uint32_t n) { /* = ceil(log10(n+1)), App. B.1 retransmit limit */
  uint32_t d = 0;
  while (n) { ++d; n /= 10; }
  return d;
}

/* ---- Lamport clock: types/clock.rs:142-172 ---- */
static inline void lc_witness(uint64_t* c, uint64_t t) { /* clock.rs:155-172 */
  if (t < *c) return;
  *c = t + 1;
}

/* ---- view bits helpers ---- */
static inline uint32_t vb_make(uint32_t known, uint32_t status, uint32_t swim, uint32_t intent,
                               uint32_t nconf, uint32_t stamp) {
  return (known & 1u) | ((status & 7u) << 1) | ((swim & 3u) << 4) | ((intent & 3u) << 6) |
         ((nconf & 7u) << 8) | (stamp << 11);
}
static inline uint32_t vb_set_status(uint32_t b, uint32_t s) { return (b & ~(7u << 1)) | ((s & 7u) << 1); }
static inline uint32_t vb_set_intent(uint32_t b, uint32_t t) { return (b & ~(3u << 6)) | ((t & 3u) << 6); }
static inline uint32_t vb_set_stamp(uint32_t b, uint32_t st) { return (b & 0x7FFu) | (st << 11); }

/* per-node processing context */
typedef struct nctx {
  osim* s;
  uint32_t l;   /* local node index */
  uint32_t gid; /* global node id   */
  sim_row* row;
  sim_record pend[MAX_PEND];
  uint32_t n_pend;
} nctx;

static inline sim_view* view_at(osim* s, uint32_t l, uint32_t subject) {
  if (subject >= s->N) return NULL;
  uint32_t a = s->slot_of[subject];
  if (a == NOSLOT) return NULL;
  return &s->view[(size_t)a * s->Nl + l];
}
static void emit_event(nctx* c, uint32_t type, uint32_t key, uint64_t ltime) {
  osim* s = c->s;
  if (!(c->row->flags & SIM_RF_WATCHED)) return; /* watchers force the serial tick loop */
  if (s->n_events == s->cap_events) {
    s->cap_events = s->cap_events ? s->cap_events * 2 : 256;
    s->events = (sim_event*)realloc(s->events, s->cap_events * sizeof(sim_event));
  }
  sim_event* e = &s->events[s->n_events++];
  e->tick = (uint32_t)s->tick;
  e->observer = c->gid;
  e->type = type;
  e->key = key;
  e->ltime = ltime;
}

static inline uint32_t kind_class(uint32_t kind) {
  switch (kind) {
    case SIM_K_JOIN:
    case SIM_K_LEAVE: return 1; /* serf `broadcasts`   delegate.rs:328     */
    case SIM_K_QUERY: return 2; /* `query_broadcasts`  delegate.rs:346-350 */
    case SIM_K_EVENT: return 3; /* `event_broadcasts`  delegate.rs:365-369 */
    default: return 0;          /* memberlist's own broadcasts go first (App. B.2) */
  }
}
static inline uint32_t wire_meta(uint32_t kind, uint32_t flags, uint32_t len_bytes) {
  uint32_t len64 = (len_bytes + 15u) / 16u;
  if (len64 > 63u) len64 = 63u;
  return ((63u - len64) << 18) | ((kind & 15u) << 4) | (flags & 15u);
}
static inline void pend_push(nctx* c, uint32_t key, uint32_t wmeta, uint64_t val) {
  if (c->n_pend >= MAX_PEND) return;
  sim_record* r = &c->pend[c->n_pend++];
  r->key = key;
  r->meta = wmeta & SIM_META_WIRE_MASK;
  r->val = val;
}

/* ---- intent buffer: base.rs:1820-1866 ---- */
static int upsert_intent(osim* s, sim_view* e, uint32_t ty, uint64_t ltime) {
  uint32_t stamp = (uint32_t)s->tick & STAMP_MASK;
  if (SIM_VB_INTENT(e->bits)) { /* Entry::Occupied  base.rs:1847-1857 */
    if (ltime > e->ltime) {
      e->bits = vb_set_stamp(vb_set_intent(e->bits, ty), stamp);
      e->ltime = ltime;
      return 1;
    }
    return 0;
  }
  e->bits = vb_set_stamp(vb_set_intent(e->bits, ty), stamp); /* Entry::Vacant base.rs:1858-1865 */
  e->ltime = ltime;
  return 1;
}
static int recent_intent(const sim_view* e, uint32_t ty, uint64_t* ltime) { /* base.rs:1824-1833 */
  if (!(e->bits & SIM_VB_KNOWN) && SIM_VB_INTENT(e->bits) == ty) {
    *ltime = e->ltime;
    return 1;
  }
  return 0;
}

/* erase_node!: base.rs:499-518 (and the list bookkeeping of its callers) */
static void erase_member(nctx* c, sim_view* e, uint32_t subject) {
  uint32_t st = SIM_VB_STATUS(e->bits);
  if (st == SIM_STATUS_FAILED && c->row->n_failed) c->row->n_failed--;
  if (st == SIM_STATUS_LEFT && c->row->n_left) c->row->n_left--;
  memset(e, 0, sizeof *e);
  if (c->row->n_known) c->row->n_known--;
  emit_event(c, SIM_EV_REAP, subject, 0);
}

/* handle_node_join_intent: base.rs:1338-1373 */
static int handle_join_intent(nctx* c, uint32_t subject, uint64_t ltime) {
  lc_witness(&c->row->clock, ltime); /* base.rs:1340 */
  sim_view* e = view_at(c->s, c->l, subject);
  if (!e) return 0; /* model bound: subject without a view slot */
  if (e->bits & SIM_VB_KNOWN) {
    if (ltime <= e->ltime) return 0; /* base.rs:1346 */
    e->ltime = ltime;                /* base.rs:1351 */
    if (SIM_VB_STATUS(e->bits) == SIM_STATUS_LEAVING) /* base.rs:1356 */
      e->bits = vb_set_status(e->bits, SIM_STATUS_ALIVE);
    return 1;
  }
  return upsert_intent(c->s, e, 1, ltime); /* base.rs:1362-1369 */
}

/* broadcast_join: base.rs:381-397 */
static void broadcast_join(nctx* c, uint64_t ltime) {
  lc_witness(&c->row->clock, ltime);                         /* base.rs:384 */
  handle_join_intent(c, c->gid, ltime);                      /* base.rs:387 */
  pend_push(c, c->gid, wire_meta(SIM_K_JOIN, 0, 16), ltime); /* base.rs:389-391 */
}

/* handle_prune: base.rs:1628-1653.  The Leaving-state sleep (broadcast_timeout +
 * leave_propagate_delay) is not modelled: the erase happens in the same tick. */
static void handle_prune(nctx* c, sim_view* e, uint32_t subject) { erase_member(c, e, subject); }

/* handle_node_leave_intent: base.rs:1442-1572 */
static int handle_leave_intent(nctx* c, uint32_t subject, uint64_t ltime, int prune) {
  uint32_t state = SIM_RF_STATE(c->row->flags); /* base.rs:1443 */
  lc_witness(&c->row->clock, ltime);            /* base.rs:1446 */
  sim_view* e = view_at(c->s, c->l, subject);
  if (!e) return 0;
  if (!(e->bits & SIM_VB_KNOWN)) return upsert_intent(c->s, e, 2, ltime); /* base.rs:1450-1458 */
  if (ltime <= e->ltime) return 0;                                        /* base.rs:1464 */
  if (subject == c->gid && state == SIM_SERF_ALIVE) {                     /* base.rs:1470-1480 */
    broadcast_join(c, c->row->clock); /* refute with clock.time(); spawned task => same tick */
    return 0;
  }
  e->ltime = ltime; /* base.rs:1497 */
  switch (SIM_VB_STATUS(e->bits)) {
    case SIM_STATUS_NONE: return 0; /* base.rs:1501 */
    case SIM_STATUS_ALIVE:          /* base.rs:1502-1511 */
      e->bits = vb_set_status(e->bits, SIM_STATUS_LEAVING);
      if (prune) handle_prune(c, e, subject);
      return 1;
    case SIM_STATUS_LEAVING:
    case SIM_STATUS_LEFT: /* base.rs:1512-1519 */
      if (prune) handle_prune(c, e, subject);
      return 1;
    case SIM_STATUS_FAILED: /* base.rs:1520-1557 */
      e->bits = vb_set_status(e->bits, SIM_STATUS_LEFT);
      if (c->row->n_failed) c->row->n_failed--;
      c->row->n_left++;
      emit_event(c, SIM_EV_LEAVE, subject, 0);
      if (prune) handle_prune(c, e, subject);
      return 1;
    default: /* base.rs:1558-1569 */
      e->bits = vb_set_status(e->bits, SIM_STATUS_LEAVING);
      if (prune) handle_prune(c, e, subject);
      return 1;
  }
}

/* handle_node_join (memberlist notify_join): base.rs:1206-1334 */
static void handle_node_join(nctx* c, uint32_t subject) {
  sim_view* e = view_at(c->s, c->l, subject);
  if (!e) return;
  if (e->bits & SIM_VB_KNOWN) { /* base.rs:1234-1274 */
    uint32_t old = SIM_VB_STATUS(e->bits);
    e->bits = vb_set_stamp(vb_set_status(e->bits, SIM_STATUS_ALIVE), 0); /* leave_time = None */
    if (old == SIM_STATUS_FAILED && c->row->n_failed) c->row->n_failed--; /* base.rs:1317-1320 */
    if (old == SIM_STATUS_LEFT && c->row->n_left) c->row->n_left--;
  } else { /* base.rs:1275-1315 */
    uint32_t status = SIM_STATUS_ALIVE;
    uint64_t lt = 0, t;
    if (recent_intent(e, 1, &t)) lt = t;                          /* base.rs:1281 */
    if (recent_intent(e, 2, &t)) { lt = t; status = SIM_STATUS_LEAVING; } /* base.rs:1285 */
    e->ltime = lt;
    e->bits = vb_make(1, status, SIM_VB_SWIM(e->bits), 0, 0, 0);
    c->row->n_known++;
  }
  emit_event(c, SIM_EV_JOIN, subject, 0);
}

/* handle_node_leave (memberlist notify_leave): base.rs:1375-1440 */
static void handle_node_leave(nctx* c, uint32_t subject) {
  sim_view* e = view_at(c->s, c->l, subject);
  if (!e || !(e->bits & SIM_VB_KNOWN)) return; /* base.rs:1378-1380 */
  uint32_t stamp = (uint32_t)c->s->tick & STAMP_MASK;
  switch (SIM_VB_STATUS(e->bits)) {
    case SIM_STATUS_LEAVING: /* base.rs:1384-1393 */
      e->bits = vb_set_stamp(vb_set_status(e->bits, SIM_STATUS_LEFT), stamp);
      c->row->n_left++;
      emit_event(c, SIM_EV_LEAVE, subject, 0);
      break;
    case SIM_STATUS_ALIVE: /* base.rs:1394-1402 */
      e->bits = vb_set_stamp(vb_set_status(e->bits, SIM_STATUS_FAILED), stamp);
      c->row->n_failed++;
      emit_event(c, SIM_EV_FAILED, subject, 0);
      break;
    default: return; /* base.rs:1403-1406 */
  }
}

/* ring bucket of node l */
static inline sim_bucket* ring_at(sim_bucket* ring, uint32_t Nl, uint32_t idx, uint32_t l) {
  return &ring[(size_t)idx * Nl + l];
}

/* handle_user_event: base.rs:750-837.  (name,payload) identity is the 32-bit event key.
 * Quirk U1 is reproduced: an existing bucket's ltime is not compared (base.rs:801-807). */
static int handle_user_event(nctx* c, uint32_t key, uint64_t ltime) {
  osim* s = c->s;
  lc_witness(&c->row->event_clock, ltime);      /* base.rs:760 */
  if (ltime < c->row->event_min) return 0;      /* base.rs:765 */
  uint64_t B = s->Bev, cur = c->row->event_clock; /* base.rs:770-771 */
  if (cur > B && ltime < cur - B) return 0;     /* base.rs:772 */
  uint32_t idx = (uint32_t)(ltime % B);         /* base.rs:783 */
  sim_bucket* b = ring_at(s->ering, s->Nl, idx, c->l);
  if (b->keys[0]) { /* Some(seen)  base.rs:801-807 */
    uint32_t n = 0;
    for (; n < SIM_C && b->keys[n]; ++n)
      if (b->keys[n] == key) return 0;
    if (n == SIM_C) { /* model bound: bucket full => treated as seen */
      c->row->overflow++;
      return 0;
    }
    b->keys[n] = key;
  } else { /* base.rs:808-813 */
    b->ltime = ltime;
    b->keys[0] = key;
  }
  emit_event(c, SIM_EV_USER, key, ltime); /* base.rs:832 */
  return 1;
}

/* handle_query (de-dup + rebroadcast decision): base.rs:972-1073.
 * Quirk Q1 (age test uses the ring length, base.rs:1012-1014) and quirk Q2 (bucket ltime not
 * updated, base.rs:1027-1036) are reproduced. */
static int handle_query(nctx* c, uint32_t id, uint64_t ltime, uint32_t flags) {
  osim* s = c->s;
  lc_witness(&c->row->query_clock, ltime);   /* base.rs:1002 */
  if (ltime < c->row->query_min) return 0;   /* base.rs:1007 */
  uint64_t cur = c->row->query_clock, qt = s->Bq; /* base.rs:1012-1013 */
  if (cur > qt && qt < cur - qt) return 0;   /* base.rs:1014 (sic) */
  uint32_t idx = (uint32_t)(ltime % qt);     /* base.rs:1025 */
  sim_bucket* b = ring_at(s->qring, s->Nl, idx, c->l);
  if (b->keys[0]) {
    uint32_t n = 0;
    for (; n < SIM_C && b->keys[n]; ++n)
      if (b->ltime == ltime && b->keys[n] == id) return 0; /* base.rs:1028-1035 */
    if (n == SIM_C) {
      c->row->overflow++;
      return 0;
    }
    b->keys[n] = id; /* base.rs:1036 */
  } else {           /* base.rs:1038-1042 */
    b->ltime = ltime;
    b->keys[0] = id;
  }
  emit_event(c, SIM_EV_QUERY, id, ltime);
  return (flags & SIM_F_NO_BROADCAST) ? 0 : 1; /* base.rs:1062-1073 */
}

/* SerfDelegate::notify_message dispatch: delegate.rs:183-300 */
static void dispatch_record(nctx* c, const sim_record* r) {
  uint32_t kind = SIM_META_KIND(r->meta), flags = SIM_META_FLAGS(r->meta);
  int rb = 0;
  switch (kind) {
    case SIM_K_LEAVE: rb = handle_leave_intent(c, r->key, r->val, flags & SIM_F_PRUNE); break; /* delegate.rs:193-204 */
    case SIM_K_JOIN: rb = handle_join_intent(c, r->key, r->val); break;                         /* delegate.rs:205-216 */
    case SIM_K_EVENT: rb = handle_user_event(c, r->key, r->val); break;                         /* delegate.rs:217-228 */
    case SIM_K_QUERY: rb = handle_query(c, r->key, r->val, flags); break;                       /* delegate.rs:229-256 */
    default: return;
  }
  if (rb) { /* delegate.rs:294-300: re-queue the ORIGINAL message unchanged */
    /* the refute join (if any) was pushed by the handler before we get here; the original
     * message is appended after it — but a refuted leave is never rebroadcast, so order is moot */
    pend_push(c, r->key, r->meta, r->val);
  }
}

/* =====================================================================================
 * TransmitLimitedQueue (memberlist-core, App. B.1) in its bounded, pooled form.
 * The Q slots of a node are kept sorted by `meta` (= drain order); empties last.
 * ===================================================================================== */
static int rec_cmp(const void* a, const void* b) {
  uint32_t x = ((const sim_record*)a)->meta, y = ((const sim_record*)b)->meta;
  return x < y ? -1 : x > y;
}
static inline void rec_clear(sim_record* r) {
  r->key = 0;
  r->meta = SIM_META_EMPTY;
  r->val = 0;
}
static void queue_renorm(sim_row* row, sim_record* q) {
  /* seq := rank by age (older = smaller); next_seq := count */
  uint32_t seqs[SIM_Q], n = 0;
  for (uint32_t i = 0; i < SIM_Q; ++i)
    if (q[i].meta != SIM_META_EMPTY) seqs[n++] = SIM_META_SEQ(q[i].meta);
  for (uint32_t i = 0; i < SIM_Q; ++i) {
    if (q[i].meta == SIM_META_EMPTY) continue;
    uint32_t sq = SIM_META_SEQ(q[i].meta), rank = 0;
    for (uint32_t j = 0; j < n; ++j) rank += (seqs[j] < sq);
    q[i].meta = (q[i].meta & ~(0x3FFu << 8)) | ((1023u - rank) << 8);
  }
  row->next_seq = n;
  qsort(q, SIM_Q, sizeof(sim_record), rec_cmp);
}
/* queue_broadcast for every pending record, in arrival order (B.1): pool = old ∪ new; a
 * memberlist (class 0) broadcast invalidates older class-0 broadcasts about the same node;
 * keep the best Q by drain order, count the rest as overflow. */
static void queue_enqueue(nctx* c, sim_record* q) {
  sim_row* row = c->row;
  if (!c->n_pend) return;
  sim_record pool[SIM_Q + MAX_PEND];
  uint32_t n = 0;
  for (uint32_t i = 0; i < SIM_Q; ++i)
    if (q[i].meta != SIM_META_EMPTY) pool[n++] = q[i];
  for (uint32_t i = 0; i < c->n_pend; ++i) {
    sim_record r = c->pend[i];
    uint32_t kind = SIM_META_KIND(r.meta);
    uint32_t seq = row->next_seq++;
    r.meta = (kind_class(kind) << 30) | (r.meta & SIM_META_WIRE_MASK) | ((1023u - seq) << 8);
    pool[n++] = r;
  }
  /* invalidation among class-0 entries with the same key: keep the newest */
  for (uint32_t i = 0; i < n; ++i) {
    if ((pool[i].meta >> 30) != 0 || pool[i].meta == SIM_META_EMPTY) continue;
    for (uint32_t j = 0; j < n; ++j) {
      if (j == i || pool[j].meta == SIM_META_EMPTY || (pool[j].meta >> 30) != 0) continue;
      if (pool[j].key == pool[i].key && SIM_META_SEQ(pool[j].meta) > SIM_META_SEQ(pool[i].meta)) {
        rec_clear(&pool[i]);
        break;
      }
    }
  }
  qsort(pool, n, sizeof(sim_record), rec_cmp);
  uint32_t valid = 0;
  while (valid < n && pool[valid].meta != SIM_META_EMPTY) ++valid;
  for (uint32_t i = 0; i < SIM_Q; ++i) {
    if (i < valid) q[i] = pool[i];
    else rec_clear(&q[i]);
  }
  if (valid > SIM_Q) row->overflow += valid - SIM_Q;
  c->n_pend = 0;
}
/* get_broadcasts for one packet (B.1 with a record-count budget of SIM_P): the first P entries
 * in drain order; transmits+1; drop at the retransmit limit; re-insert. */
static void queue_emit(sim_row* row, sim_record* q, uint32_t limit, sim_packet* out) {
  (void)row;
  memset(out, 0, sizeof *out);
  for (uint32_t p = 0; p < SIM_P; ++p) {
    sim_record* r = &q[p];
    if (r->meta == SIM_META_EMPTY) break;
    out->rec[p].key = r->key;
    out->rec[p].meta = r->meta & SIM_META_WIRE_MASK;
    out->rec[p].val = r->val;
    uint32_t t = SIM_META_TRANSMITS(r->meta) + 1;
    if (t >= limit) rec_clear(r);
    else r->meta = (r->meta & ~(0x3Fu << 24)) | (t << 24);
  }
  qsort(q, SIM_Q, sizeof(sim_record), rec_cmp);
}

/* =====================================================================================
 * Operations (the user-facing API acting on one node): api.rs / base.rs
 * ===================================================================================== */
static void nctx_init(nctx* c, osim* s, uint32_t l) {
  c->s = s;
  c->l = l;
  c->gid = s->shard0 + l;
  c->row = &s->rows[l];
  c->n_pend = 0;
}
static int has_alive_members(const osim* s) { return s->N > 1; } /* base.rs:346-359, bulk form */

static void apply_op(osim* s, const sim_opent* op) {
  if (op->node < s->shard0 || op->node >= s->shard0 + s->Nl) return; /* another shard's node */
  uint32_t l = op->node - s->shard0;
  nctx c;
  nctx_init(&c, s, l);
  sim_row* row = c.row;
  sim_record* q = &s->queue[(size_t)l * SIM_Q];
  if (row->next_seq > 1023u - 64u) queue_renorm(row, q);
  switch (op->op) {
    case SIM_OP_USER_EVENT: { /* api.rs:241-299 */
      if (!(row->flags & SIM_RF_UP)) break;
      uint64_t lt = row->event_clock;               /* api.rs:264 */
      row->event_clock++;                           /* api.rs:285 */
      handle_user_event(&c, op->a, lt);             /* api.rs:288 */
      pend_push(&c, op->a, wire_meta(SIM_K_EVENT, 0, op->b), lt); /* api.rs:290-297 */
      break;
    }
    case SIM_OP_QUERY: { /* base.rs:875-940 */
      if (!(row->flags & SIM_RF_UP)) break;
      uint64_t lt = row->query_clock;               /* base.rs:904 */
      handle_query(&c, op->a, lt, op->b);           /* base.rs:932 */
      pend_push(&c, op->a, wire_meta(SIM_K_QUERY, op->b, 32), lt); /* base.rs:935-942 */
      break;
    }
    case SIM_OP_LEAVE: { /* api.rs:422-460 */
      if (!(row->flags & SIM_RF_UP)) break;
      uint32_t st = SIM_RF_STATE(row->flags);
      if (st != SIM_SERF_ALIVE) break;              /* api.rs:426-435 */
      row->flags = (row->flags & ~(3u << 1)) | (SIM_SERF_LEAVING << 1);
      uint64_t lt = row->clock;                     /* api.rs:444 */
      row->clock++;                                 /* api.rs:449 */
      handle_leave_intent(&c, c.gid, lt, 0);        /* api.rs:452 */
      if (has_alive_members(s)) pend_push(&c, c.gid, wire_meta(SIM_K_LEAVE, 0, 16), lt); /* api.rs:456-460 */
      break;
    }
    case SIM_OP_LEAVE_FINISH: { /* api.rs:474-497: memberlist.leave, then state = Left */
      uint32_t st = SIM_RF_STATE(row->flags);
      if (st != SIM_SERF_LEAVING) break;
      row->flags = (row->flags & ~(3u << 1)) | (SIM_SERF_LEFT << 1);
      row->flags &= ~SIM_RF_UP;
      break;
    }
    case SIM_OP_JOIN: { /* api.rs:318-364: (memberlist.join,) broadcast_join(clock.time()) */
      row->flags |= SIM_RF_UP;
      row->flags = (row->flags & ~(3u << 1)) | (SIM_SERF_ALIVE << 1);
      broadcast_join(&c, row->clock);               /* api.rs:342 */
      break;
    }
    case SIM_OP_FORCE_LEAVE: { /* base.rs:452-480 */
      if (!(row->flags & SIM_RF_UP)) break;
      uint64_t lt = row->clock;                     /* base.rs:456-460 */
      handle_leave_intent(&c, op->a, lt, (int)op->b); /* base.rs:463 */
      if (has_alive_members(s))                     /* base.rs:466 */
        pend_push(&c, op->a, wire_meta(SIM_K_LEAVE, op->b ? SIM_F_PRUNE : 0, 16), lt);
      break;
    }
    case SIM_OP_CRASH: row->flags &= ~SIM_RF_UP; break;
    case SIM_OP_REVIVE: row->flags |= SIM_RF_UP; break;
    default: break;
  }
  queue_enqueue(&c, q);
}

/* =====================================================================================
 * The tick (DESIGN.md SIMSPEC §4)
 * ===================================================================================== */
static inline const sim_packet* inbox_cell(const osim* s, uint32_t k, uint32_t l) {
  if (s->cfg.shard_count > 1) { /* sharded: [src shard][k][blk] written by the previous tick */
    const tickp* pp = &s->prev;
    uint32_t b = l / pp->blk;
    uint32_t g = (s->cfg.shard_rank + b + pp->rot[k]) % pp->V;
    return &s->xrecv[((size_t)g * s->f + k) * pp->blk + (l % pp->blk)];
  }
  return &s->inbox[s->tick & 1][(size_t)k * s->Nl + l];
}

static void tick_node(osim* s, const tickp* p, uint32_t l) {
  nctx c;
  nctx_init(&c, s, l);
  sim_row* row = c.row;
  sim_record* q = &s->queue[(size_t)l * SIM_Q];
  uint32_t g = c.gid / p->M, ll = c.gid % p->M;
  sim_packet out[SIM_MAX_FANOUT];
  memset(out, 0, sizeof out);
  int up = (row->flags & SIM_RF_UP) != 0;
  if (up) {
    if (row->next_seq > 1023u - 64u) queue_renorm(row, q);
    if (s->tick > 0) {
      for (uint32_t k = 0; k < s->f; ++k) {
        const sim_packet* pk = inbox_cell(s, k, l);
        for (uint32_t r = 0; r < SIM_P; ++r)
          if (SIM_META_KIND(pk->rec[r].meta) != SIM_K_EMPTY) dispatch_record(&c, &pk->rec[r]);
      }
    }
    queue_enqueue(&c, q);
    uint32_t limit = s->cfg.retransmit_mult * digits10(row->n_known); /* B.1, serf.rs:123-131 */
    for (uint32_t k = 0; k < p->feff; ++k) queue_emit(row, q, limit, &out[k]);
  }
  /* push the f packets (empty ones too: every inbox cell is rewritten every tick) */
  uint32_t sx = p->feff ? sigma(p, ll) : 0;
  for (uint32_t k = 0; k < p->feff; ++k) {
    uint32_t h, lp;
    fan_target(p, g, sx, k, &h, &lp);
    if (up && pkt_lost(p, c.gid, k)) memset(&out[k], 0, sizeof out[k]);
    if (s->cfg.shard_count > 1)
      s->xsend[((size_t)h * s->f + k) * p->blk + (lp % p->blk)] = out[k];
    else
      s->inbox[(s->tick + 1) & 1][(size_t)k * s->Nl + (size_t)h * p->M + lp] = out[k];
  }
}

static void step_one(osim* s) {
  tickp p;
  tickp_make(&p, &s->cfg, s->tick);
  while (s->op_cursor < s->n_ops && s->ops[s->op_cursor].tick <= s->tick) {
    apply_op(s, &s->ops[s->op_cursor]);
    s->op_cursor++;
  }
  if (s->n_watched) {
    for (uint32_t l = 0; l < s->Nl; ++l) tick_node(s, &p, l);
  } else {
    int nt = oracle_threads();
#pragma omp parallel for schedule(static) num_threads(nt) if (s->Nl >= 4096)
    for (uint32_t l = 0; l < s->Nl; ++l) tick_node(s, &p, l);
  }
  s->prev = p;
  s->tick++;
}

/* =====================================================================================
 * C ABI (same shape as include/serf_sim.h, prefix osim_)
 * ===================================================================================== */
uint32_t API(abi_version)(void) { return SIM_ABI_VERSION; }
const char* API(backend_name)(void) { return "cpu-oracle"; }

static int cfg_check(const sim_config* c) {
  if (!c || c->struct_size != sizeof(sim_config)) return SIM_EINVAL;
  if (c->n_nodes < 1 || c->vshards < 1 || c->n_nodes % c->vshards) return SIM_EINVAL;
  uint32_t M = c->n_nodes / c->vshards;
  if (c->vshards > 1 && (M % c->vshards || M <= SIM_MAX_FANOUT)) return SIM_EINVAL;
  if (c->shard_count != 1 && c->shard_count != c->vshards) return SIM_EINVAL;
  if (c->shard_rank >= c->shard_count) return SIM_EINVAL;
  if (c->fanout < 1 || c->fanout > SIM_MAX_FANOUT) return SIM_EINVAL;
  if (c->event_ring < 1 || c->query_ring < 1) return SIM_EINVAL;
  if (c->retransmit_mult * digits10(c->n_nodes) > 63u) return SIM_EINVAL;
  return SIM_OK;
}

int API(destroy)(osim* s) {
  if (!s) return SIM_EINVAL;
  free(s->rows); free(s->queue); free(s->inbox[0]); free(s->inbox[1]);
  if (s->own_x) { free(s->xsend); free(s->xrecv); }
  free(s->view); free(s->ering); free(s->qring); free(s->slot_of); free(s->subject_of);
  free(s->base); free(s->ops); free(s->events); free(s);
  return SIM_OK;
}

int API(create)(const sim_config* cfg, osim** out) {
  int rc = cfg_check(cfg);
  if (rc) return rc;
  if (!out) return SIM_EINVAL;
  osim* s = (osim*)calloc(1, sizeof *s);
  if (!s) return SIM_ENOMEM;
  s->cfg = *cfg;
  s->N = cfg->n_nodes; s->V = cfg->vshards; s->M = s->N / s->V;
  s->Nl = cfg->shard_count > 1 ? s->M : s->N;
  s->shard0 = cfg->shard_count > 1 ? cfg->shard_rank * s->M : 0;
  s->dense = (cfg->view_slots == 0 || cfg->view_slots >= s->N);
  s->A = s->dense ? s->N : cfg->view_slots;
  s->Bev = cfg->event_ring; s->Bq = cfg->query_ring; s->f = cfg->fanout;
  size_t Nl = s->Nl;
  s->rows = (sim_row*)calloc(Nl, sizeof(sim_row));
  s->queue = (sim_record*)malloc(Nl * SIM_Q * sizeof(sim_record));
  if (cfg->shard_count > 1) {
    size_t cells = (size_t)s->f * s->M;
    s->xsend = (sim_packet*)calloc(cells, sizeof(sim_packet));
    s->xrecv = (sim_packet*)calloc(cells, sizeof(sim_packet));
    s->own_x = 1;
  } else {
    s->inbox[0] = (sim_packet*)calloc((size_t)s->f * Nl, sizeof(sim_packet));
    s->inbox[1] = (sim_packet*)calloc((size_t)s->f * Nl, sizeof(sim_packet));
  }
  s->view = (sim_view*)calloc((size_t)s->A * Nl, sizeof(sim_view));
  s->ering = (sim_bucket*)calloc((size_t)s->Bev * Nl, sizeof(sim_bucket));
  s->qring = (sim_bucket*)calloc((size_t)s->Bq * Nl, sizeof(sim_bucket));
  s->slot_of = (uint32_t*)malloc((size_t)s->N * sizeof(uint32_t));
  s->subject_of = (uint32_t*)malloc((size_t)s->A * sizeof(uint32_t));
  s->base = (sim_view*)calloc(s->N, sizeof(sim_view));
  if (!s->rows || !s->queue || !s->view || !s->ering || !s->qring || !s->slot_of ||
      !s->subject_of || !s->base || (cfg->shard_count > 1 ? (!s->xsend || !s->xrecv)
                                                          : (!s->inbox[0] || !s->inbox[1]))) {
    API(destroy)(s);
    return SIM_ENOMEM;
  }
  int joined = (cfg->flags & SIM_CF_BASELINE_JOINED) != 0;
  for (size_t i = 0; i < Nl * SIM_Q; ++i) rec_clear(&s->queue[i]);
  for (uint32_t a = 0; a < s->A; ++a) s->subject_of[a] = NOSLOT;
  for (uint32_t i = 0; i < s->N; ++i) {
    s->slot_of[i] = s->dense ? i : NOSLOT;
    if (joined) { s->base[i].ltime = 1; s->base[i].bits = vb_make(1, SIM_STATUS_ALIVE, 0, 0, 0, 0); }
  }
  if (s->dense) {
    s->n_slots = s->N;
    for (uint32_t a = 0; a < s->A; ++a) {
      s->subject_of[a] = a;
      for (size_t l = 0; l < Nl; ++l) s->view[(size_t)a * Nl + l] = s->base[a];
    }
  }
  for (size_t l = 0; l < Nl; ++l) {
    sim_row* r = &s->rows[l];
    /* base.rs:196-205: each clock increment()ed once => every clock starts at 1 */
    r->clock = r->event_clock = r->query_clock = 1;
    r->flags = SIM_RF_UP | (SIM_SERF_ALIVE << 1);
    if (joined) {
      r->clock = 2; /* own join at ltime 1 witnessed (base.rs:381-385) */
      r->n_known = s->N;
    } else {
      /* the synthetic notify_join(local) of new_in (base.rs:266-272): self is known, Alive @ 0 */
      r->n_known = 1;
      if (s->dense) {
        sim_view* e = &s->view[(size_t)(s->shard0 + l) * Nl + l];
        e->ltime = 0;
        e->bits = vb_make(1, SIM_STATUS_ALIVE, 0, 0, 0, 0);
      }
    }
  }
  *out = s;
  return SIM_OK;
}

int API(set_stream)(osim* s, void* st) { (void)s; (void)st; return SIM_OK; }

/* active-subject slots (non-dense): allocate at injection time, column := baseline */
static int ensure_slot(osim* s, uint32_t subject) {
  if (subject >= s->N) return SIM_EINVAL;
  if (s->slot_of[subject] != NOSLOT) return SIM_OK;
  if (s->n_slots >= s->A) return SIM_ENOSLOT;
  uint32_t a = s->n_slots++;
  s->slot_of[subject] = a;
  s->subject_of[a] = subject;
  for (size_t l = 0; l < s->Nl; ++l) s->view[(size_t)a * s->Nl + l] = s->base[subject];
  return SIM_OK;
}

int API(inject)(osim* s, uint64_t tick, uint32_t op, uint32_t node, uint32_t a, uint32_t b) {
  if (!s || node >= s->N) return SIM_EINVAL;
  if (tick < s->tick) tick = s->tick;
  int rc = SIM_OK;
  switch (op) {
    case SIM_OP_USER_EVENT: if (!a) return SIM_EINVAL; if (b > 9 * 1024) return SIM_ETOOBIG; break;
    case SIM_OP_QUERY: if (!a) return SIM_EINVAL; break;
    case SIM_OP_LEAVE: case SIM_OP_JOIN: case SIM_OP_LEAVE_FINISH: rc = ensure_slot(s, node); break;
    case SIM_OP_FORCE_LEAVE: rc = ensure_slot(s, a); break;
    case SIM_OP_CRASH: case SIM_OP_REVIVE: break;
    default: return SIM_EINVAL;
  }
  if (rc) return rc;
  if (s->n_ops == s->cap_ops) {
    s->cap_ops = s->cap_ops ? s->cap_ops * 2 : 64;
    s->ops = (sim_opent*)realloc(s->ops, s->cap_ops * sizeof(sim_opent));
    if (!s->ops) return SIM_ENOMEM;
  }
  /* stable insertion by tick (ops already consumed stay in front) */
  size_t pos = s->n_ops;
  while (pos > s->op_cursor && s->ops[pos - 1].tick > tick) { s->ops[pos] = s->ops[pos - 1]; --pos; }
  s->ops[pos].tick = tick; s->ops[pos].op = op; s->ops[pos].node = node; s->ops[pos].a = a; s->ops[pos].b = b;
  s->n_ops++;
  return SIM_OK;
}

int API(join)(osim* s, uint32_t node, uint32_t peer) { return API(inject)(s, s ? s->tick : 0, SIM_OP_JOIN, node, peer, 0); }
int API(leave)(osim* s, uint32_t node) {
  if (!s) return SIM_EINVAL;
  int rc = API(inject)(s, s->tick, SIM_OP_LEAVE, node, 0, 0);
  if (rc) return rc;
  return API(inject)(s, s->tick + s->cfg.leave_delay + 1, SIM_OP_LEAVE_FINISH, node, 0, 0);
}
int API(force_leave)(osim* s, uint32_t node, uint32_t subject, int prune) {
  return API(inject)(s, s ? s->tick : 0, SIM_OP_FORCE_LEAVE, node, subject, prune ? 1u : 0u);
}
int API(user_event)(osim* s, uint32_t node, uint32_t key, uint32_t len, int cc) {
  (void)cc;
  return API(inject)(s, s ? s->tick : 0, SIM_OP_USER_EVENT, node, key, len);
}
int API(query)(osim* s, uint32_t node, uint32_t id, uint32_t flags) {
  return API(inject)(s, s ? s->tick : 0, SIM_OP_QUERY, node, id, flags);
}

int API(step)(osim* s, uint32_t n) {
  if (!s) return SIM_EINVAL;
  for (uint32_t i = 0; i < n; ++i) step_one(s);
  return SIM_OK;
}
int API(sync)(osim* s) { return s ? SIM_OK : SIM_EINVAL; }
int API(tick)(const osim* s, uint64_t* t) { if (!s || !t) return SIM_EINVAL; *t = s->tick; return SIM_OK; }

static const sim_view* view_or_base(const osim* s, uint32_t l, uint32_t subject) {
  uint32_t a = s->slot_of[subject];
  return a == NOSLOT ? &s->base[subject] : &s->view[(size_t)a * s->Nl + l];
}
int API(members)(osim* s, uint32_t obs, uint8_t* st, uint64_t* lt, uint32_t cap) {
  if (!s || obs < s->shard0 || obs >= s->shard0 + s->Nl || cap < s->N) return s && cap < s->N ? SIM_ERANGE : SIM_EINVAL;
  for (uint32_t i = 0; i < s->N; ++i) {
    const sim_view* e = view_or_base(s, obs - s->shard0, i);
    int known = e->bits & SIM_VB_KNOWN;
    if (st) st[i] = known ? (uint8_t)SIM_VB_STATUS(e->bits) : SIM_STATUS_NONE;
    if (lt) lt[i] = known ? e->ltime : 0;
  }
  return SIM_OK;
}
int API(stats_get)(osim* s, uint32_t node, sim_stats* o) {
  if (!s || !o || node < s->shard0 || node >= s->shard0 + s->Nl) return SIM_EINVAL;
  const sim_row* r = &s->rows[node - s->shard0];
  memset(o, 0, sizeof *o);
  o->members = r->n_known; o->failed = r->n_failed; o->left = r->n_left;
  o->health_score = r->awareness;
  o->member_time = r->clock; o->event_time = r->event_clock; o->query_time = r->query_clock;
  const sim_record* q = &s->queue[(size_t)(node - s->shard0) * SIM_Q];
  for (uint32_t i = 0; i < SIM_Q; ++i) {
    if (q[i].meta == SIM_META_EMPTY) continue;
    switch (q[i].meta >> 30) { case 0: o->swim_queue++; break; case 1: o->intent_queue++; break;
                               case 2: o->query_queue++; break; default: o->event_queue++; }
  }
  o->serf_state = SIM_RF_STATE(r->flags); o->up = r->flags & SIM_RF_UP; o->incarnation = r->inc;
  o->queue_overflow = r->overflow;
  return SIM_OK;
}
int API(watch)(osim* s, uint32_t obs) {
  if (!s || obs < s->shard0 || obs >= s->shard0 + s->Nl) return SIM_EINVAL;
  if (!(s->rows[obs - s->shard0].flags & SIM_RF_WATCHED)) s->n_watched++;
  s->rows[obs - s->shard0].flags |= SIM_RF_WATCHED;
  return SIM_OK;
}
int API(drain_events)(osim* s, sim_event* out, uint32_t cap, uint32_t* n) {
  if (!s || !n) return SIM_EINVAL;
  uint32_t m = (uint32_t)(s->n_events < cap ? s->n_events : cap);
  if (out) memcpy(out, s->events, m * sizeof(sim_event));
  memmove(s->events, s->events + m, (s->n_events - m) * sizeof(sim_event));
  s->n_events -= m;
  *n = m;
  return SIM_OK;
}

/* ---- digest / dump ---- */
static inline uint64_t dig(uint64_t w, uint64_t idx) { return mix64(w ^ (idx * 0xD1342543DE82EF95ull)); }
static uint64_t dig_words(const void* p, size_t n_words) {
  const uint64_t* w = (const uint64_t*)p;
  uint64_t acc = 0;
  int nt = oracle_threads();
#pragma omp parallel for reduction(+ : acc) schedule(static) num_threads(nt) if (n_words >= (1u << 16))
  for (size_t i = 0; i < n_words; ++i) acc += dig(w[i], (uint64_t)i);
  return acc;
}
static const sim_packet* cur_inbox(const osim* s) {
  return s->cfg.shard_count > 1 ? s->xrecv : s->inbox[s->tick & 1];
}
int API(state_digest)(osim* s, uint64_t out[8]) {
  if (!s || !out) return SIM_EINVAL;
  memset(out, 0, 8 * sizeof(uint64_t));
  out[0] = dig_words(s->rows, (size_t)s->Nl * sizeof(sim_row) / 8);
  out[1] = dig_words(s->queue, (size_t)s->Nl * SIM_Q * 2);
  out[2] = dig_words(cur_inbox(s), (size_t)s->f * s->Nl * 8);
  out[3] = dig_words(s->view, (size_t)s->A * s->Nl * 4);
  out[4] = dig_words(s->ering, (size_t)s->Bev * s->Nl * 4);
  out[5] = dig_words(s->qring, (size_t)s->Bq * s->Nl * 4);
  {
    uint64_t acc = 0;
    for (uint32_t i = 0; i < s->N; ++i) acc += dig((uint64_t)s->slot_of[i], (uint64_t)i);
    out[6] = acc;
  }
  return SIM_OK;
}
int API(dump_state)(osim* s, uint32_t which, void* buf, size_t cap, size_t* bytes) {
  if (!s || !bytes) return SIM_EINVAL;
  const void* src; size_t n;
  switch (which) {
    case SIM_ARR_ROWS: src = s->rows; n = (size_t)s->Nl * sizeof(sim_row); break;
    case SIM_ARR_QUEUE: src = s->queue; n = (size_t)s->Nl * SIM_Q * sizeof(sim_record); break;
    case SIM_ARR_INBOX: src = cur_inbox(s); n = (size_t)s->f * s->Nl * sizeof(sim_packet); break;
    case SIM_ARR_VIEW: src = s->view; n = (size_t)s->A * s->Nl * sizeof(sim_view); break;
    case SIM_ARR_ERING: src = s->ering; n = (size_t)s->Bev * s->Nl * sizeof(sim_bucket); break;
    case SIM_ARR_QRING: src = s->qring; n = (size_t)s->Bq * s->Nl * sizeof(sim_bucket); break;
    case SIM_ARR_SLOTMAP: src = s->slot_of; n = (size_t)s->N * sizeof(uint32_t); break;
    default: return SIM_EINVAL;
  }
  *bytes = n;
  if (!buf) return SIM_OK;
  if (cap < n) return SIM_ERANGE;
  memcpy(buf, src, n);
  return SIM_OK;
}

int API(convergence)(osim* s, uint32_t kind, uint32_t key, uint64_t ltime, uint64_t* seen, uint64_t* up) {
  if (!s || !seen || !up) return SIM_EINVAL;
  uint64_t ns = 0, nu = 0;
  for (uint32_t l = 0; l < s->Nl; ++l) {
    if (!(s->rows[l].flags & SIM_RF_UP)) continue;
    nu++;
    switch (kind) {
      case SIM_K_JOIN: case SIM_K_LEAVE: {
        if (key >= s->N) return SIM_EINVAL;
        const sim_view* e = view_or_base(s, l, key);
        ns += ((e->bits & SIM_VB_KNOWN) && e->ltime >= ltime);
        break;
      }
      case SIM_K_EVENT: case SIM_K_QUERY: {
        const sim_bucket* ring = kind == SIM_K_EVENT ? s->ering : s->qring;
        uint32_t B = kind == SIM_K_EVENT ? s->Bev : s->Bq;
        const sim_bucket* b = &ring[(size_t)(ltime % B) * s->Nl + l];
        int hit = 0;
        for (uint32_t i = 0; i < SIM_C; ++i) hit |= (b->keys[i] == key && key != 0);
        ns += hit;
        break;
      }
      default: return SIM_EINVAL;
    }
  }
  *seen = ns; *up = nu;
  return SIM_OK;
}

int API(exchange_bytes)(const osim* s, size_t* bytes) {
  if (!s || !bytes) return SIM_EINVAL;
  *bytes = s->cfg.shard_count > 1 ? (size_t)s->f * s->M * sizeof(sim_packet) : 0;
  return SIM_OK;
}
int API(bind_exchange)(osim* s, void* send, void* recv) {
  if (!s || s->cfg.shard_count <= 1 || !send || !recv) return SIM_EINVAL;
  if (s->own_x) { free(s->xsend); free(s->xrecv); s->own_x = 0; }
  s->xsend = (sim_packet*)send;
  s->xrecv = (sim_packet*)recv;
  memset(send, 0, (size_t)s->f * s->M * sizeof(sim_packet));
  memset(recv, 0, (size_t)s->f * s->M * sizeof(sim_packet));
  return SIM_OK;
}

/* =====================================================================================
 * Handler-level test hooks (oracle only): let tests/test_oracle_kat.py replay the reference's
 * known-answer tests against the very handler functions the tick loop uses.
 * ===================================================================================== */
#define TCTX(s, node)                                                              \
  if (!(s) || (node) < (s)->shard0 || (node) >= (s)->shard0 + (s)->Nl) return SIM_EINVAL; \
  nctx c;                                                                          \
  nctx_init(&c, (s), (node) - (s)->shard0)
static int t_finish(nctx* c) { /* queue what the handler asked to rebroadcast */
  queue_enqueue(c, &c->s->queue[(size_t)c->l * SIM_Q]);
  return 0;
}
int osim_t_clock_get(osim* s, uint32_t node, uint32_t which, uint64_t* t) {
  TCTX(s, node);
  *t = which == 0 ? c.row->clock : which == 1 ? c.row->event_clock : c.row->query_clock;
  return SIM_OK;
}
int osim_t_clock_set(osim* s, uint32_t node, uint32_t which, uint64_t t) {
  TCTX(s, node);
  *(which == 0 ? &c.row->clock : which == 1 ? &c.row->event_clock : &c.row->query_clock) = t;
  return SIM_OK;
}
int osim_t_clock_witness(osim* s, uint32_t node, uint32_t which, uint64_t t) {
  TCTX(s, node);
  lc_witness(which == 0 ? &c.row->clock : which == 1 ? &c.row->event_clock : &c.row->query_clock, t);
  return SIM_OK;
}
int osim_t_clock_increment(osim* s, uint32_t node, uint32_t which, uint64_t* t) { /* clock.rs:148 */
  TCTX(s, node);
  uint64_t* p = which == 0 ? &c.row->clock : which == 1 ? &c.row->event_clock : &c.row->query_clock;
  *t = ++*p;
  return SIM_OK;
}
int osim_t_set_member(osim* s, uint32_t node, uint32_t subject, uint32_t status, uint64_t ltime, uint32_t stamp) {
  TCTX(s, node);
  sim_view* e = view_at(s, c.l, subject);
  if (!e) return SIM_ENOSLOT;
  if (!(e->bits & SIM_VB_KNOWN)) c.row->n_known++;
  else { if (SIM_VB_STATUS(e->bits) == SIM_STATUS_FAILED) c.row->n_failed--; if (SIM_VB_STATUS(e->bits) == SIM_STATUS_LEFT) c.row->n_left--; }
  e->ltime = ltime;
  e->bits = vb_make(1, status, 0, 0, 0, stamp & STAMP_MASK);
  if (status == SIM_STATUS_FAILED) c.row->n_failed++;
  if (status == SIM_STATUS_LEFT) c.row->n_left++;
  return SIM_OK;
}
int osim_t_set_tick(osim* s, uint64_t t) { if (!s) return SIM_EINVAL; s->tick = t; return SIM_OK; }
int osim_t_set_serf_state(osim* s, uint32_t node, uint32_t st) {
  TCTX(s, node);
  c.row->flags = (c.row->flags & ~(3u << 1)) | ((st & 3u) << 1);
  return SIM_OK;
}
int osim_t_set_min_time(osim* s, uint32_t node, uint32_t which, uint64_t t) {
  TCTX(s, node);
  if (which == 1) c.row->event_min = t; else c.row->query_min = t;
  return SIM_OK;
}
int osim_t_recent_intent(osim* s, uint32_t node, uint32_t subject, uint32_t ty, uint64_t* ltime) {
  TCTX(s, node);
  sim_view* e = view_at(s, c.l, subject);
  if (!e) return 0;
  return recent_intent(e, ty, ltime);
}
int osim_t_upsert_intent(osim* s, uint32_t node, uint32_t subject, uint32_t ty, uint64_t ltime) {
  TCTX(s, node);
  sim_view* e = view_at(s, c.l, subject);
  if (!e) return SIM_ENOSLOT;
  return upsert_intent(s, e, ty, ltime);
}
int osim_t_join_intent(osim* s, uint32_t node, uint32_t subject, uint64_t ltime) {
  TCTX(s, node);
  int rb = handle_join_intent(&c, subject, ltime);
  if (rb) pend_push(&c, subject, wire_meta(SIM_K_JOIN, 0, 16), ltime);
  t_finish(&c);
  return rb;
}
int osim_t_leave_intent(osim* s, uint32_t node, uint32_t subject, uint64_t ltime, int prune) {
  TCTX(s, node);
  int rb = handle_leave_intent(&c, subject, ltime, prune);
  if (rb) pend_push(&c, subject, wire_meta(SIM_K_LEAVE, prune ? SIM_F_PRUNE : 0, 16), ltime);
  t_finish(&c);
  return rb;
}
int osim_t_user_event(osim* s, uint32_t node, uint32_t key, uint64_t ltime) {
  TCTX(s, node);
  int rb = handle_user_event(&c, key, ltime);
  if (rb) pend_push(&c, key, wire_meta(SIM_K_EVENT, 0, 32), ltime);
  t_finish(&c);
  return rb;
}
int osim_t_query(osim* s, uint32_t node, uint32_t id, uint64_t ltime, uint32_t flags) {
  TCTX(s, node);
  int rb = handle_query(&c, id, ltime, flags);
  if (rb) pend_push(&c, id, wire_meta(SIM_K_QUERY, flags, 32), ltime);
  t_finish(&c);
  return rb;
}
int osim_t_notify_join(osim* s, uint32_t node, uint32_t subject) { TCTX(s, node); handle_node_join(&c, subject); return SIM_OK; }
int osim_t_notify_leave(osim* s, uint32_t node, uint32_t subject) { TCTX(s, node); handle_node_leave(&c, subject); return SIM_OK; }

/* Reaper::run body: base.rs:521-581 (reap! on failed with reconnect_timeout, on left with
 * tombstone_timeout, then reap_intents base.rs:1820-1822); `now` and timeouts in ticks. */
int osim_t_reap(osim* s, uint32_t node, uint64_t now, uint64_t reconnect_timeout,
                uint64_t tombstone_timeout, uint64_t intent_timeout) {
  TCTX(s, node);
  for (uint32_t subj = 0; subj < s->N; ++subj) {
    sim_view* e = view_at(s, c.l, subj);
    if (!e) continue;
    uint64_t age = (now - SIM_VB_STAMP(e->bits)) & STAMP_MASK;
    if (e->bits & SIM_VB_KNOWN) {
      uint32_t st = SIM_VB_STATUS(e->bits);
      if (st == SIM_STATUS_FAILED && age > reconnect_timeout) erase_member(&c, e, subj);     /* base.rs:535-552 */
      else if (st == SIM_STATUS_LEFT && age > tombstone_timeout) erase_member(&c, e, subj);
    } else if (SIM_VB_INTENT(e->bits) && age > intent_timeout) { /* base.rs:1820-1822 */
      memset(e, 0, sizeof *e);
    }
  }
  return SIM_OK;
}

/* get_queue_max: base.rs:437-450 / QueueChecker base.rs:728-739 */
uint64_t osim_t_queue_max(uint64_t n_members, uint64_t max_queue_depth, uint64_t min_queue_depth) {
  uint64_t max = max_queue_depth;
  if (min_queue_depth > 0) {
    max = n_members * 2;
    if (max < min_queue_depth) max = min_queue_depth;
  }
  return max;
}

/* SerfDelegate::merge_remote_state: delegate.rs:427-554.  status_ltimes as parallel arrays;
 * left[] lists subjects in left_members; events as (ltime,key) pairs. */
int osim_t_merge_remote_state(osim* s, uint32_t node, uint64_t ltime, uint64_t event_ltime,
                              uint64_t query_ltime, const uint32_t* subj, const uint64_t* st_ltime,
                              uint32_t n_status, const uint32_t* left, uint32_t n_left,
                              const uint64_t* ev_ltime, const uint32_t* ev_key, uint32_t n_ev,
                              int is_join, int event_join_ignore) {
  TCTX(s, node);
  if (ltime > 0) lc_witness(&c.row->clock, ltime - 1);              /* delegate.rs:466-468 */
  if (event_ltime > 0) lc_witness(&c.row->event_clock, event_ltime - 1); /* delegate.rs:469-474 */
  if (query_ltime > 0) lc_witness(&c.row->query_clock, query_ltime - 1); /* delegate.rs:475-480 */
  for (uint32_t i = 0; i < n_left; ++i) {                           /* delegate.rs:495-512 */
    for (uint32_t j = 0; j < n_status; ++j)
      if (subj[j] == left[i]) { handle_leave_intent(&c, left[i], st_ltime[j] + 1, 0); break; }
  }
  for (uint32_t j = 0; j < n_status; ++j) {                         /* delegate.rs:515-526 */
    int is_left = 0;
    for (uint32_t i = 0; i < n_left; ++i) is_left |= (left[i] == subj[j]);
    if (is_left) continue;
    handle_join_intent(&c, subj[j], st_ltime[j]);
  }
  if (is_join && event_join_ignore && event_ltime > c.row->event_min) c.row->event_min = event_ltime; /* delegate.rs:531-537 */
  for (uint32_t i = 0; i < n_ev; ++i) handle_user_event(&c, ev_key[i], ev_ltime[i]); /* delegate.rs:540-552 */
  c.n_pend = 0; /* merge does not rebroadcast */
  return SIM_OK;
}

/* fan-out inspection for tests: targets of global node `gid` at `tick` (returns feff) */
int osim_t_targets(osim* s, uint64_t tick, uint32_t gid, uint32_t* out_targets) {
  if (!s || gid >= s->N) return SIM_EINVAL;
  tickp p;
  tickp_make(&p, &s->cfg, tick);
  uint32_t g = gid / p.M, l = gid % p.M;
  uint32_t sx = p.feff ? sigma(&p, l) : 0;
  for (uint32_t k = 0; k < p.feff; ++k) {
    uint32_t h, lp;
    fan_target(&p, g, sx, k, &h, &lp);
    out_targets[k] = h * p.M + lp;
  }
  return (int)p.feff;
}
