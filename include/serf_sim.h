/*
 * serf_sim.h — C ABI of the MI355X bulk SWIM/Serf gossip simulator.
 *
 * This is the drop-in boundary for the hot path of al8n/serf (serf-core 0.5.1):
 * Lamport-clocked broadcast / rebroadcast, the join/leave-intent state machine, the user-event
 * and query de-duplication rings and the retransmit-limited piggyback queues — executed for all N
 * simulated nodes per gossip tick on the GPU.  One call acts on every simulated node (bulk form).
 *
 * The reference has no FFI today (it is 100 % Rust, `#![forbid(unsafe_code)]`,
 * serf-core/src/lib.rs:3).  The entry points below are what a Rust `extern "C"` block for this path
 * would bind (see INTEGRATION.md for the binding).  Each entry point cites the reference interface
 * it replaces.  All pointers are plain host pointers unless the name says `dev`; sizes are element
 * counts unless they say bytes.  Return value: 0 = ok, < 0 = SIM_E* code.  A handle is
 * single-threaded (the host serialises calls; parallelism is the GPU's).
 *
 * Both the HIP product library (serf_amd/csrc, exported with the prefix `sim_`) and the CPU oracle
 * (oracle/, exported with the prefix `osim_`, TEST INFRASTRUCTURE ONLY) implement exactly this
 * interface; only POD layouts and constants are shared through this header, never code.
 */
#ifndef SERF_SIM_H
#define SERF_SIM_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ constants */

#define SIM_ABI_VERSION 15u

#define SIM_P 4u  /* piggyback records per packet PAGE (48-byte cell: 4 records x 12 wire bytes) */
#define SIM_PKT_BYTES 1400u /* byte budget of a gossip packet: memberlist's UDP payload limit (lan() and wan()); a packet
                             * takes records in drain order while they fit it (delegate.rs:317-384 `limit`, App. B.1
                             * get_broadcasts) — and at most SIM_P of them.  Lengths count in 16-byte units.            */
#define SIM_PKT_UNITS (SIM_PKT_BYTES / 16u)
#define SIM_PAGES_MAX 4u        /* pages of SIM_P records in one packet                            */
#define SIM_PKT_RECORDS_MAX (SIM_P * SIM_PAGES_MAX)
/* The capacity bounds of the model.  The product is built with exactly these values; the oracle can ALSO be
 * built with far larger ones (oracle/Makefile: liboracle_unbounded.so) so that a test can show that a bounded run
 * which never hit a bound (overflow == 0) is identical to the run without the bounds
 * (tests/test_oracle_unbounded.py; reference: queues of up to 4096 entries options.rs:513, a Vec per ring bucket
 * base.rs:783-813, one suspicion timer per suspected member).
 * (ABI 14) SIM_Q went from 16 to 64 and a ring bucket that holds SIM_C keys continues in the ring's OVERFLOW ROWS
 * (sim_config.ring_overflow, below) instead of treating a seventh key as seen. */
#ifndef SIM_Q
#define SIM_Q 64u /* retransmit-queue slots per node (all four queues share the pool).  The HIP library keeps the SIM_Q_HOT
                   * entries that drain first in registers inside the tick kernel; a node whose queue is (or would get)
                   * deeper is finished by a second kernel behind it (deep_queue_kernel) — same results */
#endif
#define SIM_Q_HOT 16u
#ifndef SIM_C
#define SIM_C 6u  /* keys per de-dup ring bucket (32-byte bucket); further keys: the ring's overflow rows */
#endif
#define SIM_MAX_FANOUT 4u
#define SIM_MAX_CONF 4u /* conf[0] = node that started the suspicion, conf[1..3] = confirmers (k <= 3) */
#ifndef SIM_S
#define SIM_S 16u       /* suspicion timers a node can track at once (16-bit view slots).  At 1 Mi nodes under 1 % packet
                         * loss there are always two or three FALSE suspicions in flight (0.26 failed probes per tick,
                         * each alive for the ~8 ticks its refutation takes) next to the real ones: 8 was too tight   */
#endif
#define SIM_MAX_AWARENESS 7u /* memberlist awareness_max_multiplier - 1 (lan: 8)      */

/* error codes (serf-core/src/error.rs:64-81 maps its enum onto these for the bulk path) */
#define SIM_OK 0
#define SIM_EINVAL -1    /* bad argument / bad config                                  */
#define SIM_ENOMEM -2    /* host or device allocation failed                           */
#define SIM_EDEVICE -3   /* HIP runtime error (no GPU, launch failure)                 */
#define SIM_ENOSLOT -4   /* no free view slot for a new active subject                 */
#define SIM_ESTATE -5    /* call not valid in the node's SerfState (api.rs:422-437)    */
#define SIM_ERANGE -6    /* output buffer too small                                    */
#define SIM_ETOOBIG -7   /* user event / query larger than the configured limit        */

/* MemberStatus — serf-core/src/types/member.rs:54-87 (same numeric values). */
enum sim_member_status {
  SIM_STATUS_NONE = 0,
  SIM_STATUS_ALIVE = 1,
  SIM_STATUS_LEAVING = 2,
  SIM_STATUS_LEFT = 3,
  SIM_STATUS_FAILED = 4
};

/* SerfState — serf-core/src/serf.rs:80-89. */
enum sim_serf_state { SIM_SERF_ALIVE = 0, SIM_SERF_LEAVING = 1, SIM_SERF_LEFT = 2, SIM_SERF_SHUTDOWN = 3 };

/* memberlist node state (memberlist-core 0.8.1, not vendored). */
enum sim_swim_state { SIM_SWIM_ALIVE = 0, SIM_SWIM_SUSPECT = 1, SIM_SWIM_DEAD = 2, SIM_SWIM_LEFT = 3 };

/* MemberEventType — serf-core/src/event.rs:263-279, plus user/query events (event.rs:367-378). */
enum sim_event_type {
  SIM_EV_JOIN = 0,
  SIM_EV_LEAVE = 1,
  SIM_EV_FAILED = 2,
  SIM_EV_UPDATE = 3, /* key = node, ltime = its new incarnation: handle_node_update (base.rs:1576-1624) after a
                        SIM_OP_SET_TAGS elsewhere.  Fires for alive messages only; a push-pull that delivers the
                        same news does not carry the meta in this model and stays silent */
  SIM_EV_REAP = 4,
  SIM_EV_USER = 5,
  SIM_EV_QUERY = 6
};

/* Piggyback record kinds.  1-4 are serf messages (serf-core/src/types/message.rs:17-28),
 * 5-7 are memberlist's own state broadcasts (App. B.4). */
enum sim_kind {
  SIM_K_EMPTY = 0,
  SIM_K_JOIN = 1,    /* JoinMessage{ltime,id}         types/join.rs:18-38            */
  SIM_K_LEAVE = 2,   /* LeaveMessage{ltime,id,prune}  types/leave.rs:21-44           */
  SIM_K_EVENT = 3,   /* UserEventMessage              types/user_event/message.rs:15 */
  SIM_K_QUERY = 4,   /* QueryMessage (dedup fields)   types/query.rs:17-28           */
  SIM_K_ALIVE = 5,
  SIM_K_SUSPECT = 6,
  SIM_K_DEAD = 7
};

/* record flags (4 bits) */
#define SIM_F_PRUNE 1u        /* LeaveMessage.prune                       */
#define SIM_F_NO_BROADCAST 1u /* QueryFlag::NO_BROADCAST (types/query.rs) */
#define SIM_F_ACK 2u          /* QueryFlag::ACK                           */
#define SIM_F_RESPOND 4u      /* simulation: every node that processes the query also calls respond() (query.rs:117-149) */
#define SIM_QT 256u           /* running-query table, direct-mapped by query_id % SIM_QT (a newer query with the
                               * same residue takes the entry over: model bound)                 */
#define SIM_F_CC 1u           /* UserEventMessage.cc (coalesce)           */
#define SIM_F_META 1u         /* ALIVE: the node's meta (tags) differs from what its previous incarnation carried:
                               * a receiver that already knew the node alive gets notify_update (SIM_EV_UPDATE)     */
/* QueryParam.filters (types/filter.rs; should_process_query, query.rs:439-521): per running query one entry next to
 * the tracker, {query id, number of ids, tag-class mask, sealed, ids[SIM_QF_IDS]} (16 words; `sealed` = 1 once the
 * entry's SIM_OP_QUERY has been consumed: a later filter or query operation under the same id starts a fresh entry).  Filter::Id lists up to
 * SIM_QF_IDS node ids.  Filter::Tag is a regular expression over a tag's value: string work the host does once per
 * query, not per node — every node carries a TAG CLASS (0..31; class 0 = "no tags", which no tag filter matches,
 * query.rs:475-477/509-511), the host evaluates each tag filter against the (at most 31) distinct tag sets and
 * hands over the AND of the masks of matching classes.  A node processes the query iff its id is in the id list (when
 * there is one) and its class is in the mask (when there is one). */
#define SIM_QF_IDS 12u
#define SIM_QF_WORDS 16u
#define SIM_TAG_CLASSES 32u

/*
 * 16-byte piggyback record — the unit carried in packets and held in queues.
 *   key : subject node id (JOIN/LEAVE/ALIVE/SUSPECT/DEAD), event key (EVENT) or query id (QUERY)
 *   val : Lamport time (JOIN/LEAVE/EVENT/QUERY), incarnation (ALIVE),
 *         incarnation | from << 32 (SUSPECT/DEAD)
 *   meta: [31:30] class  [29:24] transmits  [23:18] 63-len64  [17:8] 1023-seq  [7:4] kind  [3:0] flags
 * Comparing `meta` as an unsigned integer yields TransmitLimitedQueue's drain order
 * (class asc, transmits asc, length desc, id desc — App. B.1); an empty slot is meta = 0xFFFFFFFF.
 * On the wire class/transmits/seq are zero (SIM_META_WIRE_MASK).
 */
typedef struct sim_record {
  uint32_t key;
  uint32_t meta;
  uint64_t val;
} sim_record;

#define SIM_META_EMPTY 0xFFFFFFFFu
#define SIM_META_WIRE_MASK 0x00FC00FFu /* len, kind, flags */
#define SIM_META_KIND(m) (((m) >> 4) & 0xFu)
#define SIM_META_FLAGS(m) ((m)&0xFu)
#define SIM_META_TRANSMITS(m) (((m) >> 24) & 0x3Fu)
#define SIM_META_SEQ(m) (1023u - (((m) >> 8) & 0x3FFu))
#define SIM_META_LEN64(m) (63u - (((m) >> 18) & 0x3Fu))

/* 48-byte gossip packet PAGE: SIM_P records in their 12-byte WIRE form, stored field by field (a packet is
 * sim_config.pkt_records / SIM_P of them, records in order, unused positions and pages zero).  A record on the wire needs
 * its key (32 bits), 48 bits of value — a Lamport time or an incarnation; for SUSPECT / DEAD the incarnation (24 bits)
 * and the accuser `from` (24 bits) — and 14 bits of meta (len64, kind, flags: SIM_META_WIRE_MASK squeezed together);
 * class, transmits and queue id never travel.  Packets are half of the tick's HBM traffic: 48 instead
 * of 64 bytes is an eighth of all bytes moved.  Model bounds: Lamport times below 2^48, incarnations and node ids
 * below 2^24.  An empty record is all zero (kind 0). */
typedef struct sim_packet {
  uint32_t key[SIM_P];
  uint32_t val_lo[SIM_P];   /* value bits 31..0                                    */
  uint32_t hi_meta[SIM_P];  /* [31:16] value bits 47..32   [13:8] 63-len64   [7:4] kind   [3:0] flags */
} sim_packet;
#define SIM_WIRE_TWO_PART(kind) ((kind) == SIM_K_SUSPECT || (kind) == SIM_K_DEAD)
/* sim_record.val -> the 48 bits that travel, and back */
#define SIM_WIRE_VAL48(kind, val) \
  (SIM_WIRE_TWO_PART(kind) ? (((val) & 0xFFFFFFull) | ((((val) >> 32) & 0xFFFFFFull) << 24)) : ((val) & 0xFFFFFFFFFFFFull))
#define SIM_WIRE_VAL(kind, v48) (SIM_WIRE_TWO_PART(kind) ? (((v48) & 0xFFFFFFull) | (((v48) >> 24) << 32)) : (v48))
/* sim_record.meta (wire bits) -> the 14 bits that travel, and back */
#define SIM_WIRE_META14(meta) (((((meta) >> 18) & 0x3Fu) << 8) | ((meta) & 0xFFu))
#define SIM_WIRE_META(m14) (((((m14) >> 8) & 0x3Fu) << 18) | ((m14) & 0xFFu))

/*
 * 32-byte per-(observer, subject-slot) view entry.
 *   bits: [0] known  [3:1] MemberStatus  [5:4] swim state  [7:6] buffered intent (0 none,1 join,2 leave)
 *         [10:8] nconf  [31:11] stamp (tick, 21 bits: suspicion start / intent wall time / leave time)
 *   ltime: status_time when known, buffered intent ltime otherwise (base.rs:1835-1866)
 */
typedef struct sim_view {
  uint64_t ltime;
  uint32_t inc;
  uint32_t bits;
  uint32_t conf[SIM_MAX_CONF];
} sim_view;

#define SIM_VB_KNOWN 1u
#define SIM_VB_STATUS(b) (((b) >> 1) & 7u)
#define SIM_VB_SWIM(b) (((b) >> 4) & 3u)
#define SIM_VB_INTENT(b) (((b) >> 6) & 3u)
#define SIM_VB_NCONF(b) (((b) >> 8) & 7u)
#define SIM_VB_STAMP(b) ((b) >> 11)

/* 32-byte de-dup ring bucket (event ring: base.rs:783-813, query ring: base.rs:1025-1042).
 * keys[0] == 0 means "bucket absent" (None).
 * (ABI 14) A ring array is [X + B][Nl] buckets, X = sim_config.ring_overflow: rows 0 .. X-1 are the node's OVERFLOW ROWS,
 * the bucket of Lamport time t is row X + t mod B.  The reference keeps a Vec per bucket (base.rs:801-813 push, 1027-1042);
 * here the keys beyond SIM_C of bucket i continue in overflow rows whose `ltime` field holds i + 1 (0: the row is free) —
 * rows are handed out in ascending order, one owner each, never given back (the reference never shrinks the Vec
 * either); a bucket's keys in push order = its own SIM_C, then its overflow rows in ascending row order.  A key that
 * finds every row taken is treated as seen and counted in sim_row.overflow (the model bound that is left). */
typedef struct sim_bucket {
  uint64_t ltime;
  uint32_t keys[SIM_C];
} sim_bucket;

/* 112-byte per-node row: the node's own (non-view) state.  serf.rs:133-169 (`SerfCore`): three
 * Lamport clocks, EventCore/QueryCore min_time, SerfState; plus memberlist's incarnation,
 * awareness and the suspicion timers it is running (memberlist-core). */
typedef struct sim_row {
  uint64_t clock, event_clock, query_clock; /* types/clock.rs:124; all start at 1     */
  uint64_t event_min, query_min;            /* EventCore.min_time / QueryCore.min_time */
  uint32_t flags;                           /* SIM_RF_*                                */
  uint32_t inc;                             /* own incarnation (memberlist)            */
  uint32_t n_known, n_failed, n_left;       /* |members.states|, |failed|, |left|      */
  uint32_t next_seq;                        /* next TransmitLimitedQueue id            */
  uint32_t overflow;                        /* records dropped by the Q bound          */
  uint32_t susp_next;                       /* earliest suspicion deadline (tick), 0 = none */
  uint32_t awareness;                       /* memberlist health score                 */
  uint32_t reap_next;                       /* earliest tick the Reaper has work for this node, 0 = none */
  uint16_t susp[SIM_S];                     /* view slot + 1 of each running suspicion timer, 0 = free
                                             * (the SWIM layer therefore needs view_slots <= 65534)    */
} sim_row;

#define SIM_RF_UP 1u
#define SIM_RF_STATE(f) (((f) >> 1) & 3u) /* enum sim_serf_state */
#define SIM_RF_WATCHED 8u
#define SIM_RF_MINTIME 16u /* event_min or query_min is non-zero (snapshot restore / join-ignore) */

/* ------------------------------------------------------------------ configuration */

typedef struct sim_config {
  uint32_t struct_size;       /* sizeof(sim_config), for ABI checking                           */
  uint32_t n_nodes;           /* N: simulated nodes, ids 0..N-1 (cluster-wide)                  */
  uint32_t vshards;           /* V >= 1 virtual shards (N % V == 0, (N/V) % V == 0)             */
  uint32_t shard_rank;        /* this process's shard when shard_count > 1                      */
  uint32_t shard_count;       /* 1 (all V shards local) or V (one shard per process/GPU)        */
  uint32_t fanout;            /* gossip_nodes: 3 = lan(), 4 = wan()   (1..SIM_MAX_FANOUT)       */
  uint32_t view_slots;        /* A: active-subject slots; 0 or >= N => dense (slot == subject)  */
  uint32_t event_ring;        /* event_buffer_size (options.rs:516), default 512                */
  uint32_t query_ring;        /* query_buffer_size (options.rs:517), default 512                */
  uint32_t retransmit_mult;   /* memberlist retransmit_mult, lan() = 4                          */
  uint32_t probe_interval;    /* ticks per probe (lan: 1 s / 200 ms = 5); 0 = SWIM layer off    */
  uint32_t suspicion_mult;    /* lan 4                                                          */
  uint32_t suspicion_max_mult;/* lan 6                                                          */
  uint32_t indirect_checks;   /* lan 3                                                          */
  uint32_t loss_u32;          /* packet loss probability * 2^32 (0 = lossless)                  */
  uint32_t intent_timeout;    /* recent_intent_timeout in ticks (options.rs:515), 0 = never     */
  uint32_t leave_delay;       /* broadcast_timeout + leave_propagate_delay in ticks             */
  uint32_t reap_interval;     /* Reaper period in ticks (options.rs:506, 15 s), 0 = reaper off  */
  uint32_t reconnect_timeout; /* failed members are reaped after this many ticks (24 h)         */
  uint32_t tombstone_timeout; /* left members are reaped after this many ticks (24 h)           */
  uint32_t queue_check_interval; /* QueueChecker period in ticks (30 s), 0 = off               */
  uint32_t max_queue_depth;   /* options.rs:513 (4096)                                          */
  uint32_t min_queue_depth;   /* options.rs:514: > 0 => cap = max(2 * members, min)             */
  uint32_t push_pull_interval;/* memberlist push_pull_interval in ticks (lan: 30 s = 150), before its
                               * log2(N) scaling; 0 = no anti-entropy                             */
  uint32_t chunks;            /* C sender chunks per shard for the chunk-wise exchange (0 or 1: one); must divide
                               * (N / V) / V.  Chunk c = the nodes whose offset inside their N/V/V-node block lies in
                               * [c, c+1) * (N/V/V/C): their packets for one (destination, slot) form ONE dense slab     */
  uint32_t recycle_interval;  /* view-slot recycling pass every this many ticks (0 = never): a subject whose entry
                               * is the same at every running node and that nothing in flight mentions gives its
                               * slot back (SIMSPEC §2.6; reference analogue: erase_node! base.rs:499-518)        */
  uint32_t pkt_records;       /* records a gossip packet can carry: 0 (= 4), 4, 8, 12 or 16 — a packet is up to
                               * SIM_PAGES_MAX pages of SIM_P records, filled in drain order under the byte budget
                               * SIM_PKT_BYTES (delegate.rs:317-384; App. B.1 get_broadcasts).  With 4 the record
                               * budget binds first for small messages; with 16 = SIM_Q a packet can carry the whole
                               * queue and only the byte budget is left                           */
  uint32_t flags;             /* SIM_CF_*                                                       */
  uint32_t gossip_to_the_dead;/* memberlist gossip_to_the_dead_time in ticks (lan: 30 s = 150; App. B.2): a node whose
                               * view says the target of one of its packets has been dead / left for longer does not
                               * gossip to it (kRandomNodes would not have picked it) — here: that packet is not sent,
                               * the transmits are spent all the same.  0 = every node stays a gossip target          */
  uint32_t reconnect_interval;/* Reconnector period in ticks (options.rs reconnect_interval, 30 s = 150), 0 = off: a node
                               * with failed members attempts, with probability failed / alive, a memberlist.join — a
                               * push-pull — with one of them (base.rs:612-681; SIMSPEC §2.9).  Needs the SWIM layer
                               * (members only fail there).  (ABI <= 9: this word was reserved, 0)                   */
  uint32_t ring_overflow;     /* X: overflow rows per de-dup ring and node (sim_bucket above), each SIM_C further keys
                               * for ONE bucket that ran full; 0 = none (a full bucket then treats a new key as seen,
                               * counted — the behaviour of ABI <= 13).  (ABI 14)                                     */
  uint32_t reserved0;         /* 0                                                                                    */
  uint64_t seed;              /* master seed; default 0x5EEDC0DE5E4F0001                        */
} sim_config;

#define SIM_CF_BASELINE_JOINED 1u /* all nodes known+Alive at status_time 1, clock 2 (config 2-5) */
#define SIM_CF_RANDOM_FANOUT 2u   /* gossip targets are memberlist's literal kRandomNodes (uniform over the other nodes, no
                                   * replacement, skip self — App. B.2) instead of the per-tick bijection; in-degree is then
                                   * Poisson-like and a node's packets are handed over in (sender, slot) order.  Sender chunks
                                   * (sim_config.chunks > 1) only on a shard, where they are the exchange's schedule (SIM_EINVAL
                                   * on a handle that holds every node); packets of 1 - 4 pages; checkpoints like every other run
                                   * (the targets of the packets in flight are a function of (seed, tick, sender): drawn again
                                   * on restore).  Canonical form of the packets in flight in this mode (SIM_ARR_INBOX, digests,
                                   * images): inbox[k * PG + pg][SENDER] — on a shard: its own senders.
                                   * (r4) The packets stay in their senders' cells and every receiver pulls what the tick's graph
                                   * (a CSR the HIP library builds two ticks ahead with its own two-level bucket sort) addresses
                                   * to it.  On SHARDS (r5): every shard draws the targets of its OWN senders,
                                   * sorts the (target, sender, slot) triples, and PACKS the packets bound for shard h into one
                                   * dense slab in that order, with one count byte per target of h; the round's exchange is ONE
                                   * equal-split all-to-all of those slabs (sim_exchange_layout: SIM_XCHG_PACKED) — f * 64 B *
                                   * M * (V - 1) / V bytes leave a GPU per round (+ 2 % of room, + 1 byte per target) —, and the
                                   * receiver's row is V sorted runs, one per source shard, in ascending source = ascending
                                   * sender order.  No index and no count travels ahead; nobody draws anybody else's targets.
                                   * With sim_config.chunks = C the senders are cut into C ranges, each sorted, packed and
                                   * exchanged behind its own launch (slab (c, h) of the send buffer; a row is V * C runs): chunk
                                   * c travels while chunk c + 1 computes, like the bijection's chunks. */
/* random fan-out only: broadcast requests one node can park in ONE tick beyond f * pkt_records + SIM_S + 1 (the bijection's
 * maximum; with a random in-degree there is none).  A counted model bound, the same in the oracle. */
#define SIM_RF_PEND_EXTRA 32u
#define SIM_CF_AWARENESS_PROBE 4u  /* memberlist scales its probe interval by the node's health score (awareness, state.go
                                   * probeNode: ScaleTimeout): a node with score s probes in every (s + 1)-th round of its
                                   * group's probe phase instead of every round (App. B.3)                              */
#define SIM_CF_JOIN_SYNC 8u        /* Serf::join is memberlist.join: a push-pull with the peer before anything else.  With this
                                   * flag SIM_OP_JOIN first lets the joining node ADOPT the view of a running node of its own
                                   * shard (the first one at or after `peer` mod shard size): every allocated view entry, the
                                   * member counters, the clocks (witnessed), the suspicion timers of the adopted entries.
                                   * The re-broadcasts a real merge would queue tell the cluster nothing it does not have
                                   * and are not modelled.  Without it a re-joining node keeps the view it went down with
                                   * until a push-pull batch reaches it (DESIGN.md SIMSPEC §2.8)                        */
#define SIM_CF_TCP_FALLBACK 16u     /* memberlist probeNode's fallback (state.go; `disable_tcp_pings` = false is memberlist's default):
                                   * next to the indirect pings the prober pings the target over TCP, and a probe whose UDP legs
                                   * were all lost still SUCCEEDS when that stream ping gets through — reliable in this model, so
                                   * with the flag a probe fails only when the target's process is down: packet loss produces no
                                   * false suspicions (at 1 Mi nodes and 1 % loss they were 0.26 per tick, each a suspicion and a
                                   * refutation for everybody to carry)                                                          */
#define SIM_CF_NACKS 32u           /* memberlist's nack accounting (protocol >= 4): a probe that fails raises the prober's health
                                   * score by the number of relays that were asked and did NOT answer with a nack (relay down,
                                   * or the request / the nack lost) instead of by one — a prober whose own network is fine
                                   * does not degrade itself for a peer that is really dead; no relay asked: + 1 as before     */
#define SIM_CF_FORCE_SHARDED 64u     /* run the handle as ONE SHARD of a sharded cluster although shard_count == 1 (vshards == 1): exchange
                                   * buffers, the sharded instantiation of the tick kernel, the host-driven push-pull / recycling /
                                   * suspicion hand-over — the N > 1 path with a single rank, so that it (and its collective) can be
                                   * rehearsed on one GPU.  State and digests are those of the plain single-handle run.            */
#define SIM_CF_PRUNE_DELAY 128u      /* (ABI 15) handle_prune's wait (serf/base.rs:1628-1653): a pruning leave intent about a member that is — or
                                   * thereby becomes — Leaving erases it `leave_delay` ticks (broadcast_timeout + leave_propagate_delay) AFTER the
                                   * intent was handled, not in the same tick: the node notes (node, subject) on the tick's request list (the list
                                   * of the slot-less suspicions and the reconnect attempts), the library replays it as SIM_OP_PRUNE at tick
                                   * t + max(2, leave_delay) — the list is read two ticks after it was written — and the member is erased then,
                                   * whatever it has become, with its Reap event (erase_node!).  Left and Failed members are erased at once, as the
                                   * reference does.  Needs the SWIM layer (probe_interval > 0: the request lists are its machinery); sim_create
                                   * returns SIM_EINVAL otherwise.  Model bound: the request list's (SIM_SUSPECT_REQ_MAX per tick; cluster-wide,
                                   * sharded or not) — a tick that overflows it loses its requests, counted in ops_dropped.  NOT modelled: the
                                   * reference sleeps with the member lock held, so every member handler of that node stalls for the duration; here
                                   * the node keeps handling messages.  Off (the default): the erase happens in the tick of the intent. */
#define SIM_DEFAULT_SEED 0x5EEDC0DE5E4F0001ull

/* Stats — mirrors serf-core/src/serf/api.rs:586-602 (`Stats`) for one simulated node. */
typedef struct sim_stats {
  uint32_t members, failed, left;
  uint32_t health_score;
  uint64_t member_time, event_time, query_time; /* the three Lamport clocks          */
  uint32_t intent_queue, event_queue, query_queue, swim_queue; /* queue depths       */
  uint32_t serf_state;                          /* enum sim_serf_state               */
  uint32_t up;                                  /* ground truth: process running     */
  uint32_t incarnation;
  uint32_t queue_overflow;                      /* records dropped by the Q bound    */
} sim_stats;

/* Event record surfaced for watched nodes (event.rs:325-378). */
typedef struct sim_event {
  uint32_t tick;
  uint32_t observer;
  uint32_t type;  /* enum sim_event_type                          */
  uint32_t key;   /* subject id / event key / query id            */
  uint64_t ltime; /* Lamport time of the message, 0 for member events from SWIM */
} sim_event;

/* Cluster-wide load figures of the local shard (no reference counterpart: `Stats` summed over nodes).
 * `overflow` > 0 means the run hit one of the simulator's model bounds (SIM_Q pooled queue slots, SIM_C
 * keys per ring bucket, SIM_S suspicion timers, a probe target without a view slot) and is no longer a
 * run of the unbounded protocol; benchmarks report it and refuse a non-zero value. */
typedef struct sim_cluster_stats {
  uint64_t up;            /* nodes whose process is running                                     */
  uint64_t queued[4];     /* queue entries by class: memberlist, intents, queries, events       */
  uint64_t overflow;      /* sum of sim_row.overflow                                            */
  uint64_t inbox_records; /* non-empty records in the packets in flight                         */
  uint64_t failed, left;  /* sum of n_failed / n_left                                           */
  uint64_t max_queue;     /* deepest queue of any node                                          */
  uint64_t ops_dropped;   /* scheduled operations skipped because no view slot was free (model bound; handle-wide) */
  uint64_t slots_in_use, slots_recycled; /* view slots handed out now / given back so far (handle-wide)  */
  uint64_t events_lost;   /* watched nodes' events that did not fit the device log (it holds 2^20 events between two
                             sim_drain_events calls; the CPU oracle's log grows and never loses any)             */
} sim_cluster_stats;

/* One candidate of a view-slot recycling pass (sharded runs: every shard scans its own nodes, the host combines). */
typedef struct sim_recycle_cand {
  uint32_t subject, slot;
  uint32_t flags;          /* 1: not recyclable on this shard; 2: `ref` is valid (the shard has a running node) */
  uint32_t pad;
  sim_view ref;            /* the entry the shard's running nodes agree on (conf zeroed)                        */
} sim_recycle_cand;
#define SIM_RECYCLE_BATCH 64u

typedef struct sim_handle sim_handle;

/* Scheduled operation kinds for sim_inject. */
enum sim_op {
  SIM_OP_USER_EVENT = 1, /* a = event key (!= 0), b = encoded length in bytes     api.rs:241  */
  SIM_OP_QUERY = 2,      /* a = query id (!= 0),  b = flags                       api.rs:304  */
  SIM_OP_LEAVE = 3,      /* graceful leave of `node`                              api.rs:422  */
  SIM_OP_JOIN = 4,       /* (re)join: a = peer                                    api.rs:318  */
  SIM_OP_FORCE_LEAVE = 5,/* a = subject, b = prune                                api.rs:505  */
  SIM_OP_CRASH = 6,      /* ground truth: process stops (tests `shutdown()` a node, event.rs:112) */
  SIM_OP_REVIVE = 7,     /* ground truth: process resumes with its old state                   */
  SIM_OP_LEAVE_FINISH = 8,/* internal: memberlist.leave + state = Left (api.rs:462-497)        */
  SIM_OP_SET_TAGS = 9,   /* a = tag class (< SIM_TAG_CLASSES): Serf::set_tags, api.rs:219 — stores the tags and
                            triggers memberlist.update_node (incarnation + 1, an alive broadcast)      */
  SIM_OP_QUERY_FILTER_ID = 10,  /* a = query id, b = a node id to add to the query's Filter::Id (`node` is not used);
                                   scheduled BEFORE the SIM_OP_QUERY it belongs to.  A 13th id does not fit: the
                                   operation is dropped and counted in ops_dropped (model bound)                  */
  SIM_OP_QUERY_FILTER_TAGS = 11,/* a = query id, b = mask of the tag classes one Filter::Tag matches (ANDed into
                                   the query's mask); BEFORE the SIM_OP_QUERY, like the ids                       */
  SIM_OP_SUSPECT = 13,          /* internal: `node` suspects a = target (from = node) — a probe that failed on a target without
                                   a view slot in the previous tick (SIMSPEC §2.7: the suspicion is taken up one tick late,
                                   once the target has its slot); scheduled by the library / by sim_suspect_requests' caller */
  SIM_OP_RECONNECT = 14,        /* internal: `node`'s Reconnector attempts memberlist.join(a = one of its failed members) — a push-pull
                                   between the two, run in this tick if both processes are up, no push-pull batch falls on the tick
                                   and no earlier attempt of the tick involves either node (otherwise: the next tick; an attempt on
                                   a process that is down fails and is forgotten).  Scheduled by the library from the tick's request
                                   list, two ticks after the attempt was drawn (like SIM_OP_SUSPECT)                              */
  SIM_OP_DELIVER = 12,          /* internal (sim_inject_record / sim_deliver_message): `node` receives one record from
                                   outside the simulated cluster — SerfDelegate::notify_message (delegate.rs:157-315) for
                                   the serf kinds, memberlist's alive / suspect / dead handling for its own.  b = the wire bits
                                   of meta; b | SIM_DELIVER_MUTE: the record comes out of a PushPull message —
                                   merge_remote_state (delegate.rs:427-554) runs the handler and re-queues nothing but a
                                   refutation                                                                              */
  SIM_OP_QRESP = 15,            /* internal (sim_deliver_message of a QueryResponseMessage, or of a Relay that wraps one): `node`,
                                   the origin of running query a, receives the ack (b bit 31 set) or the response (clear) of
                                   node b & 0xFFFFFF — counted when the query is still inside its deadline and names `node`
                                   as its origin (handle_query_response base.rs:1158-1204, query.rs:240-303; one entry per
                                   responder), exactly like one that arrived over the simulated network                      */
  SIM_OP_WITNESS = 16,          /* internal (a PushPull message's clocks, delegate.rs:466-480): `node` witnesses Lamport time
                                   `val` on clock a (0 member, 1 event, 2 query) — the caller passes remote - 1              */
  SIM_OP_PRUNE = 17             /* internal (SIM_CF_PRUNE_DELAY): `node` erases member a — the end of handle_prune's wait (base.rs:1636-1652);
                                   scheduled by the library from the tick's request list, like SIM_OP_SUSPECT: an entry (node, a | 1 << 30)  */
};
#define SIM_DELIVER_MUTE 0x80000000u

/* ------------------------------------------------------------------ entry points */

/* Serf::new / new_in (api.rs:25, base.rs:62-344): allocate all N nodes' state in HBM.
 * Clocks start at 1 (base.rs:198-205).  Fails with SIM_EDEVICE when no HIP device is usable. */
int sim_create(const sim_config* cfg, sim_handle** out);
/* Serf::shutdown (api.rs:525). */
int sim_destroy(sim_handle* h);
/* Use an existing HIP stream (e.g. torch's current stream) for all launches; NULL = default. */
int sim_set_stream(sim_handle* h, void* hip_stream);

/* Serf::join (api.rs:318) / Serf::leave (api.rs:422) / remove_failed_node(_prune) (api.rs:505,
 * base.rs:452-480) / user_event (api.rs:241) / query (api.rs:304, base.rs:875) — applied at the
 * start of the next tick, in call order. */
int sim_join(sim_handle* h, uint32_t node, uint32_t peer);
int sim_leave(sim_handle* h, uint32_t node);
int sim_force_leave(sim_handle* h, uint32_t node, uint32_t subject, int prune);
int sim_user_event(sim_handle* h, uint32_t node, uint32_t event_key, uint32_t encoded_len, int coalesce);
/* flags: SIM_F_NO_BROADCAST | SIM_F_ACK | SIM_F_RESPOND, and QueryParam.relay_factor (0..7) in bits [10:8]. */
int sim_query(sim_handle* h, uint32_t node, uint32_t query_id, uint32_t flags);
/* The same with QueryParam.filters (query.rs:439-521): ids[n_ids] is the Filter::Id list (n_ids <= SIM_QF_IDS, else
 * SIM_ETOOBIG; NULL / 0 = none), tag_mask the AND of the Filter::Tag class masks (0xFFFFFFFF = no tag filter).  Nodes the
 * filters exclude still rebroadcast the query the first time they see it (base.rs:1063-1073) but neither ack, respond
 * nor see the event.  Shorthand for SIM_OP_QUERY_FILTER_* + SIM_OP_QUERY at the next tick. */
int sim_query_filtered(sim_handle* h, uint32_t node, uint32_t query_id, uint32_t flags, const uint32_t* ids, uint32_t n_ids,
                       uint32_t tag_mask);
/* Serf::set_tags (api.rs:219) in the tag-class model above, applied at the start of the next tick. */
int sim_set_tags(sim_handle* h, uint32_t node, uint32_t tag_class);
/* The tags the nodes [first, first + count) were STARTED with (Options::with_tags, options.rs; read by the filter at
 * query.rs:465/499): classes[i] < SIM_TAG_CLASSES, written straight into the table — no update_node, no gossip, no view
 * slot.  Every shard of a sharded run makes the same call. */
int sim_init_tags(sim_handle* h, uint32_t first, uint32_t count, const uint8_t* classes);
/* Churn / packet-loss / kill / revive schedule: run `op` on `node` at the start of tick `tick`
 * (reference analogue: MessageDropper delegate.rs:42-45 and tests that shutdown() a node). */
int sim_inject(sim_handle* h, uint64_t tick, uint32_t op, uint32_t node, uint32_t a, uint32_t b);

/* ---- the byte boundary of the delegate: SerfDelegate::notify_message(buf) (delegate.rs:157-163) and
 * SerfDelegate::broadcast_messages(..) -> Bytes (delegate.rs:317-384) are byte interfaces.  A message is the reference's
 * own encoding — one type byte, the body length as a varint, the body (types/message.rs:397-428; join.rs:123-158,
 * leave.rs:138-195, user_event/message.rs:205-272, query.rs:404-527); node ids are decimal strings; the bits of
 * memberlist-proto's tag bytes are an UPSTREAM-RECALL assumption (serf_amd/wire.py).
 *
 * sim_inject_record: at the start of tick `tick`, `node` receives `rec` (key, wire bits of meta, val) as if it had come
 *   in a packet: the handler runs, a rebroadcast is queued.  A member record whose subject finds no view slot is
 *   dropped and counted like an operation (ops_dropped).
 * sim_deliver_message: decode ONE framed serf message from buf[0 .. len) and schedule what it means for the next tick;
 *   *consumed (may be NULL) = bytes used, so a caller can walk a packet of several messages.  Join, Leave, UserEvent, Query:
 *   sim_inject_record.  (r4) QueryResponse (types/query/response.rs): the origin `node` counts the ack / response of the
 *   node it names (SIM_OP_QRESP).  Relay (types/message.rs:431-470: a destination Node and a framed message): `node`
 *   forwards the wrapped message to the destination as it is (delegate.rs:262-313: memberlist.send) — it is delivered to the
 *   destination exactly as if handed to it directly, provided `node` is running as of the last tick; a relay that is down
 *   forwards nothing (SIM_OK, the bytes are consumed).  A Relay or a PushPull inside a Relay is refused (SIM_EINVAL).
 *   ConflictResponse: taken and ignored — notify_message has no arm for it (delegate.rs:286-288).
 *   PushPull (types/push_pull.rs; merge_remote_state delegate.rs:427-554): the three clocks are witnessed at remote - 1
 *   (SIM_OP_WITNESS), every left member becomes a leave intent at its status_ltime + 1, every other member a join intent at
 *   its status_ltime, every buffered user event is replayed — all with SIM_DELIVER_MUTE: nothing is rebroadcast but a
 *   refutation.  A user event is identified by the 32-bit FNV-1a key of (name, payload) (serf_amd/host/wire.hpp
 *   event_key) and its content is remembered for sim_peek_packet; QueryFlag bits are mapped (ACK 1 -> SIM_F_ACK,
 *   NO_BROADCAST 2 -> SIM_F_NO_BROADCAST); Filter::Id lists are installed for the query, a Filter::Tag is refused with
 *   SIM_EINVAL (tag expressions are evaluated by the host: sim_query_filtered).  SIM_EINVAL for anything malformed.
 * sim_user_event_bytes: Serf::user_event(name, payload, coalesce) (api.rs:241-299) with the bytes themselves: checks
 *   name + payload <= 512 (SIM_ETOOBIG), derives key and encoded length with the codec, remembers the content.
 * sim_peek_packet: the serf messages of the packet `node` sent in fan-out slot k during the LAST tick — what
 *   broadcast_messages handed memberlist for that packet — encoded back to back, in packet order (memberlist's own
 *   alive / suspect / dead records, which travel in the same simulated packet, are not serf messages and are left out;
 *   memberlist's compound framing is memberlist-proto and is not emitted).  A user event whose content the library
 *   was never told (sim_user_event with a bare key) is encoded with the name "#<key in hex>" and no payload; a query with
 *   the name "#q" and the tracker's origin and relay factor when it is still running.  buf == NULL returns the size. */
int sim_inject_record(sim_handle* h, uint64_t tick, uint32_t node, const sim_record* rec);
int sim_deliver_message(sim_handle* h, uint32_t node, const uint8_t* buf, size_t len, size_t* consumed);
int sim_user_event_bytes(sim_handle* h, uint32_t node, const uint8_t* name, size_t name_len, const uint8_t* payload,
                         size_t payload_len, int coalesce);
int sim_peek_packet(sim_handle* h, uint32_t node, uint32_t k, uint8_t* buf, size_t cap, size_t* len);

/* The hot loop: n_ticks gossip intervals for every node.  Replaces, per node and tick:
 * SerfDelegate::notify_message (delegate.rs:157-315), the handlers it dispatches to
 * (base.rs:750-837, 972-1073, 1338-1373, 1442-1572), SerfDelegate::broadcast_messages
 * (delegate.rs:317-384) and memberlist's TransmitLimitedQueue / gossip fan-out (App. B.1-B.2).
 * Asynchronous on the handle's stream; sim_sync waits. */
int sim_step(sim_handle* h, uint32_t n_ticks);
int sim_sync(sim_handle* h);
int sim_tick(const sim_handle* h, uint64_t* tick);

/* Serf::members (api.rs:136) as seen by `observer`: out_status[s] = MemberStatus of subject s,
 * out_ltime[s] = status_time.  cap must be >= N. */
int sim_members(sim_handle* h, uint32_t observer, uint8_t* out_status, uint64_t* out_ltime, uint32_t cap);
/* Serf::stats (api.rs:150-183). */
int sim_stats_get(sim_handle* h, uint32_t node, sim_stats* out);
/* Event stream (EventSubscriber, event.rs:430-491) for watched observers.  sim_drain_events hands over at most `cap`
 * events in (tick, observer) order and keeps the rest for the next call.  The device log is bounded (2^20 events
 * between two drains): what does not fit is dropped and counted in sim_cluster_stats.events_lost — drain often
 * enough, or watch fewer nodes. */
int sim_watch(sim_handle* h, uint32_t observer);
int sim_drain_events(sim_handle* h, sim_event* out, uint32_t cap, uint32_t* n);

/* Bit-exact comparison support.  The digest is 8 x u64: one order-independent 64-bit sum
 * (sum over elements of mix64(index, value)) per state array:
 * [0] rows [1] queues [2] inbox [3] view [4] event ring [5] query ring [6] ops/aux [7] reserved */
int sim_state_digest(sim_handle* h, uint64_t out[8]);
/* Raw dump of one state array of the local shard (host buffer).  which = enum sim_array.
 * Call with buf == NULL to get the size in *bytes.  SIM_ARR_INBOX is the canonical form of the packets in flight:
 * inbox[k * PG + pg][node] = page pg of the packet `node` is about to receive in fan-out slot k ([fanout * PG][local
 * nodes] sim_packet, PG = pkt_records / SIM_P pages per packet).  An
 * implementation is free to keep them otherwise — the HIP library keeps one copy of each distinct packet at its SENDER
 * (DESIGN.md sections 2.3, 3) — as long as the dump, the digest and the image are this form. */
enum sim_array { SIM_ARR_ROWS = 0, SIM_ARR_QUEUE = 1, SIM_ARR_INBOX = 2, SIM_ARR_VIEW = 3,
                 SIM_ARR_ERING = 4, SIM_ARR_QRING = 5, SIM_ARR_SLOTMAP = 6 };
int sim_dump_state(sim_handle* h, uint32_t which, void* buf, size_t cap_bytes, size_t* bytes);

/* Rumor convergence: number of up nodes that have applied the message (kind,key,ltime) and
 * number of up nodes (rounds-to-99 % = first tick with seen >= 0.99 * up). */
int sim_convergence(sim_handle* h, uint32_t kind, uint32_t key, uint64_t ltime,
                    uint64_t* seen, uint64_t* up);

/* The same for n <= SIM_CONV_MAX rumours in ONE pass over the nodes (bench.py follows every user event of its workload
 * through a window of ticks: one launch per tick instead of one per rumour and tick): seen[i] for (kinds[i], keys[i],
 * ltimes[i]); `up` is the common denominator. */
#define SIM_CONV_MAX 64u
int sim_convergence_many(sim_handle* h, uint32_t n, const uint32_t* kinds, const uint32_t* keys, const uint64_t* ltimes,
                         uint64_t* seen, uint64_t* up);

/* Sharded mode (shard_count == V > 1): the tick kernel writes outgoing packets into `send` and
 * reads incoming ones from `recv`, both DEVICE buffers of sim_exchange_bytes() bytes laid out as
 * [V destinations][fanout][N/V/V packets].  The caller (serf_amd.shard, RCCL all_to_all_single)
 * moves send -> recv between sim_step(h,1) calls. */
int sim_exchange_bytes(const sim_handle* h, size_t* bytes);
int sim_bind_exchange(sim_handle* h, void* send_dev, void* recv_dev);
/* Chunk-wise exchange (sim_config.chunks = C > 1): the buffers are [C sender chunks][V peers][fanout][N/V/V/C packets],
 * so chunk c is one contiguous region of bytes_per_chunk bytes with an equal split per peer, complete as soon as the
 * launch of chunk c has finished — its all-to-all can travel while chunk c + 1 computes.  Because the receive buffer
 * of round t is still being read by the later chunks of round t + 1 while the first chunks of round t + 1 are
 * already arriving, the receive side is double-buffered: packets sent during tick t land in recv[t & 1].
 * A tick is then driven as  sim_step_begin;  for c in 0..C-1: sim_step_chunk(c), <exchange chunk c into recv[t & 1]>;
 * sim_step_end  — and every exchange of tick t must have completed before sim_step_chunk of tick t + 1. */
int sim_bind_exchange2(sim_handle* h, void* send_dev, void* recv0_dev, void* recv1_dev);
int sim_exchange_chunks(const sim_handle* h, uint32_t* chunks, size_t* bytes_per_chunk);
/* (r5) The same with the sizes of the caller's buffers: SIM_EINVAL when `send_bytes` is less than sim_exchange_bytes() or
 * `recv_bytes` (of recv0 and of recv1 each) less than sim_exchange_layout()'s recv_bytes — the unsized calls above trust the caller. */
int sim_bind_exchange3(sim_handle* h, void* send_dev, size_t send_bytes, void* recv0_dev, void* recv1_dev, size_t recv_bytes);
/* (r4, r5) What the round's exchange of this handle IS.  Every kind is an EQUAL-SPLIT all-to-all of the send buffer: V slabs of
 * send_plane_bytes / V bytes, slab p to rank p, slab g of the receive buffer from rank g (torch.distributed.all_to_all_single,
 * or sim_exchange_chunk); `planes` is 1 and recv_bytes = send_plane_bytes.
 *   SIM_XCHG_ALL_TO_ALL  the bijection's slabs [C][V][fp][M / V / C], written by the tick kernel itself.
 *   SIM_XCHG_PACKED      SIM_CF_RANDOM_FANOUT on a shard (r5).  memberlist's kRandomNodes sends a packet to ANY node, so there is no
 *                        dense slab the tick kernel could write into: the packets stay in their senders' cells (as on one GPU) and
 *                        sim_step_chunk PACKS, behind the tick's launch, the packets bound for shard h into slab h in (target,
 *                        sender, slot) order — the order of a sort of the shard's own f * M (target, sender, slot) triples —
 *                        together with one count byte per target of h (with C sender chunks: C x V slabs, chunk-major, each
 *                        chunk's V slabs one region of sim_exchange_chunks' bytes_per_chunk, exchanged on its own).  A slab holds
 *                        serf_rf_slab_cap(f, M / C, V) packets (mean + 12
 *                        sigma of the binomial: a slab that would overflow makes the step fail with SIM_ERANGE).  sim_step_begin of
 *                        the next tick turns the V slabs it received into the tick's rows.  The slab format is the
 *                        implementation's own (HIP: 64-byte cells; oracle: 48-byte packets with explicit targets): ranks of one
 *                        run use one implementation.  AFTER sim_restore the handle has packed the restored packets in flight into
 *                        the send buffer again and the host has to run the exchange once more before the next tick.
 *                        Limits of the HIP library in this mode (sim_create returns SIM_EINVAL beyond them; the oracle, whose slabs
 *                        are plain arrays, has none): at most 64 shards (V <= 64: one partial sum per destination in a 64-entry
 *                        table), a receive buffer of fewer than 2^30 16-byte units (C * V slabs), and (4 + V * C) << LB bytes of
 *                        LDS for the receiver's index pass, LB = the graph build's bucket width (11 at 1 Mi nodes per shard): V * C <= 28.
 * (SIM_XCHG_ALL_GATHER, ABI 12's O(N)-per-shard form of the random fan-out, is retired: no handle reports it.) */
#define SIM_XCHG_ALL_TO_ALL 0u
#define SIM_XCHG_ALL_GATHER 1u
#define SIM_XCHG_PACKED 2u
/* packets one (source shard, destination shard) slab of SIM_XCHG_PACKED holds: the mean f * M / V of the binomial, 12 sigma and
 * some — or everything the source can send, if that is less (tiny clusters).  Integer arithmetic: the same on every side. */
static inline uint32_t serf_rf_slab_cap(uint32_t fanout, uint32_t M, uint32_t V) {
  uint64_t all = (uint64_t)fanout * M, mean = (all + V - 1) / V, r = 0;
  while ((r + 1) * (r + 1) <= mean) ++r; /* floor(sqrt(mean)) */
  uint64_t cap = mean + 12 * (r + 1) + 64;
  return (uint32_t)(cap < all ? cap : all);
}
int sim_exchange_layout(const sim_handle* h, uint32_t* kind, uint32_t* planes, size_t* send_plane_bytes, size_t* recv_bytes);
/* The round's all-to-all ISSUED BY THE LIBRARY over RCCL (grouped ncclSend / ncclRecv pairs over xGMI) — for a
 * host that has no collective library of its own (the north star's Rust host) and to take the per-chunk host cost out of the
 * tick: sim_exchange_chunk is one call, no tensor bookkeeping, and the ordering lives on streams, not in the host.
 *   sim_exchange_unique_id(id)            rank 0 makes the communicator's id (ncclGetUniqueId) and hands the SIM_EXCHANGE_ID_BYTES
 *                                         bytes to every rank over whatever side channel the host has
 *   sim_exchange_init(h, id, rank, world) ncclCommInitRank; world == the handle's shard count, rank == its shard rank; the
 *                                         exchange buffers must be bound (sim_bind_exchange2).  Collective: every rank calls it.
 *   sim_exchange_chunk(h, c)              after sim_step_chunk(h, c): the slabs of sender chunk c — [V peers] equal splits of
 *                                         bytes_per_chunk / V — go out and come in as ONE group of V ncclSend + V ncclRecv on the
 *                                         library's exchange stream, which waits for that chunk's launch only: chunk c travels
 *                                         while chunk c + 1 computes.  Packets sent during tick t land in recv[t & 1].
 *                                         (r6) With ONE chunk per tick there is no later launch to travel beside: the group is issued
 *                                         on the handle's own stream, behind the pack — nothing hops between two streams.  The
 *                                         slab a shard addresses to itself never travels (it is packed in place).
 *                                         With the SWIM layer on, the LAST chunk's group also carries the head of the tick's list of
 *                                         slot-less suspicions to every peer (sim_suspect_import with heads == NULL reads them).
 *   sim_exchange_wait(h)                  the handle's stream waits (on the device, no host wait) for every exchange issued so
 *                                         far; sim_step_begin, sim_sync and every call that reads the packets in flight (digest,
 *                                         dump, checkpoint, recycling scan) do it by themselves.
 *   sim_exchange_library(buf, cap)        "RCCL <major>.<minor>.<patch>" of the library the calls above go to.
 * SIM_ESTATE before sim_exchange_init; SIM_EDEVICE when RCCL reports an error.  (The CPU oracle has no collective library:
 * its entry points return SIM_EDEVICE — a test moves its buffers by hand.) */
#define SIM_EXCHANGE_ID_BYTES 128u
int sim_exchange_unique_id(uint8_t* id_out);
int sim_exchange_init(sim_handle* h, const uint8_t* id, uint32_t rank, uint32_t world);
int sim_exchange_chunk(sim_handle* h, uint32_t chunk);
int sim_exchange_wait(sim_handle* h);
int sim_exchange_library(char* buf, size_t cap);
/* View-slot recycling in sharded runs.  A single-process handle runs the pass inside sim_step_begin.  With one shard
 * per process the verdict needs every shard: when sim_recycle_due() the host calls sim_recycle_scan on every shard
 * (same candidates everywhere: the slot bookkeeping is replicated), keeps the candidates for which NO shard set flag 1
 * and all shards that set flag 2 report the same `ref`, and hands that list to sim_recycle_apply on every shard —
 * before sim_step_begin, which otherwise refuses with SIM_ESTATE. */
/* Cross-shard push-pull.  memberlist's pushPull picks ANY peer, so the pairs of a batch come
 * from a matching over all N nodes and most of them span two shards.  A single-process handle merges them inside
 * sim_step_begin; with one shard per process, on a batch tick the host runs, AFTER sim_step_begin (the tick's
 * operations come first) and before the first sim_step_chunk (which refuses with SIM_ESTATE while sim_pp_due):
 *   sim_pp_plan(h, send1, recv1, &record_bytes)  records this shard sends to / receives from each of the V peers in
 *                                                round 1; round 2 uses the same counts swapped
 *   sim_pp_export(h, 1, send); <all-to-all-v>; sim_pp_merge(h, 1, recv)
 *   sim_pp_export(h, 2, send); <all-to-all-v>; sim_pp_merge(h, 2, recv)
 * Round 1: the odd-sigma node `b` of every cross pair ships its state (clocks, view heads, event ring: what
 * SerfDelegate::local_state and memberlist's node list carry, delegate.rs:386-425) to the shard of the even one `a`,
 * which merges it; in-shard pairs merge both ways.  Round 2: `a` ships its UPDATED state back, `b` merges.  Buffers
 * are DEVICE memory for the HIP library, grouped by peer shard, records in ascending pair order. */
int sim_pp_due(const sim_handle* h);
int sim_pp_plan(sim_handle* h, uint32_t* send1, uint32_t* recv1, size_t* record_bytes);
int sim_pp_export(sim_handle* h, int round, void* send_dev);
int sim_pp_merge(sim_handle* h, int round, const void* recv_dev);
/* Probes that failed on a target WITHOUT a view slot during the tick that just ended (only with packet loss: a target
 * that is really down got its slot when it crashed).  The prober cannot hold that suspicion yet: the pair goes on the
 * tick's request list and is replayed at the next tick as SIM_OP_SUSPECT (after that tick's scheduled operations, in
 * ascending prober order), which gives the target its slot first.  A single-process handle does this by itself in
 * sim_step_end.  With one shard per process the host calls sim_suspect_requests on every shard after sim_step_end —
 * out[2 i] = prober, out[2 i + 1] = target, sorted by prober; the call empties the list —, gathers the lists, and injects
 * sim_inject(h, tick, SIM_OP_SUSPECT, prober, target, 0) for the merged list, ascending by prober, on EVERY shard
 * (serf_amd/shard.py).  More than SIM_SUSPECT_REQ_MAX requests in one tick on one shard: all of them are dropped and
 * counted in ops_dropped (model bound).
 * The list also carries the tick's Reconnector attempts (sim_config.reconnect_interval): out[2 i] = node, out[2 i + 1] =
 * target | 1 << 31; sorted by (node, second word).  Injected the same way — sim_inject(h, tick, SIM_OP_SUSPECT, node, word) —
 * an entry with bit 31 set becomes SIM_OP_RECONNECT.  A reconnect attempt is a push-pull pair of its tick: with one shard per
 * process sim_pp_due is then also true on a tick that has such pairs and no batch, and sim_pp_plan / _export / _merge run them. */
#define SIM_SUSPECT_REQ_MAX 4096u
int sim_suspect_requests(sim_handle* h, uint32_t* out, uint32_t cap_pairs, uint32_t* n_pairs);
/* The same hand-over WITHOUT a host round trip per tick (what serf_amd/shard.py uses: a collective whose result the host
 * has to read before the next tick would let no rank run ahead of its GPU).  After sim_step_end(t), sim_suspect_export
 * enqueues a copy of the HEAD of tick t's list — SIM_SREQ_HEAD_WORDS 32-bit words: the count, then the first
 * SIM_SREQ_HEAD_PAIRS (prober, target) pairs, unsorted — into `out` (DEVICE memory for the HIP library).  The host
 * all-gathers the heads of all shards with an asynchronous device-side collective, copies the result to host memory behind
 * it, and any time before sim_step_begin of tick t + 2 hands it to sim_suspect_import(h, t, heads, world) on every shard,
 * which merges the lists in ascending prober order and schedules them as SIM_OP_SUSPECT for tick t + 2 — the tick a
 * single-process handle replays them in.  (ABI 15) The bound is the single-process handle's: more than SIM_SUSPECT_REQ_MAX
 * requests in one tick over ALL shards together and every one of them is dropped and counted in ops_dropped, on every shard.
 * (r6) A handle whose round's exchange the LIBRARY issues (sim_exchange_init) needs neither the export nor a collective of the
 * host's: sim_exchange_chunk of a tick's last chunk sends the head to every peer in the same group as the slabs and writes what
 * the V heads hold into pinned host memory behind it (an event marks it); sim_suspect_import(h, t, NULL, world) — heads == NULL —
 * waits for that event and imports them.  SIM_EINVAL without sim_exchange_init (and in the CPU oracle, which has no collective
 * library), SIM_ESTATE when no exchange of tick t was issued or its heads were imported already. */
#define SIM_SREQ_HEAD_PAIRS 4096u   /* (ABI 15: = SIM_SUSPECT_REQ_MAX, was 255 — the notes of SIM_CF_PRUNE_DELAY come a node a tick in a rumour's wavefront) */
#define SIM_SREQ_HEAD_WORDS 8193u
int sim_suspect_export(sim_handle* h, void* out);
int sim_suspect_import(sim_handle* h, uint64_t of_tick, const uint32_t* heads, uint32_t world);
int sim_recycle_due(const sim_handle* h);
int sim_recycle_scan(sim_handle* h, sim_recycle_cand* out, uint32_t cap, uint32_t* n);
int sim_recycle_apply(sim_handle* h, const sim_recycle_cand* agreed, uint32_t n);
int sim_step_begin(sim_handle* h);
int sim_step_chunk(sim_handle* h, uint32_t chunk);
int sim_step_end(sim_handle* h);

/* Checkpoint / resume of the whole simulated cluster (the reference checkpoints one node's members and
 * clocks, serf-core/src/snapshot.rs:117-126,228-347; here the unit is the simulation).  The image is the
 * CANONICAL state — the arrays of sim_dump_state plus slot maps, liveness, running queries, the pending
 * operation schedule and the tick — so an image written by one implementation of this ABI restores into
 * another.  sim_snapshot with buf == NULL returns the size.  sim_restore needs a handle created with the
 * same sim_config (checked) that has not been stepped yet; un-drained events are not part of the image. */
int sim_snapshot(sim_handle* h, void* buf, size_t cap_bytes, size_t* bytes);
int sim_restore(sim_handle* h, const void* buf, size_t bytes);

/* Query acks and responses (serf-core/src/serf/base.rs:1075-1154 sender side, 1158-1204 and
 * serf/query.rs:240-303 origin side): number of distinct nodes whose ack (QueryFlag::ACK) / response
 * (SIM_F_RESPOND) reached the origin of the running query `query_id` before its deadline
 * (query.rs:421-427: gossip_interval * query_timeout_mult (16) * ceil(log10(N+1))); `open` = still
 * inside the deadline.  SIM_EINVAL if the id does not own its entry of the running-query table. */
int sim_query_status(sim_handle* h, uint32_t query_id, uint64_t* acks, uint64_t* responses, int* open);
/* WHO they are — what the reference hands the caller of Serf::query over QueryResponse::ack_rx / response_rx
 * (serf/query.rs:201-212; handle_query_response :240-303 keeps one entry per responder: the `acks` / `responses` sets): the
 * ids of the nodes of THIS shard whose ack (which = 0) or response (which = 1) reached the origin, ascending, as many as fit
 * `cap`; *n = how many there are.  The response's payload is what the responding node's user passes to respond(): not
 * simulated data, so the host supplies it per responder (serf_amd/host/serf.hpp QueryResponse, NodeResponse{from,
 * payload}).  SIM_EINVAL like sim_query_status. */
int sim_query_responders(sim_handle* h, uint32_t query_id, int which, uint32_t* out_ids, uint32_t cap, uint32_t* n);

/* Measurement: with profiling on, launches of the tick kernel are bracketed by HIP events on the
 * handle's stream — every launch for enable == 1, every n-th for enable == n > 1 (an event pair
 * costs several microseconds of stream time); sim_profile_read waits for the stream and returns
 * the summed kernel time and the number of timed launches since the last read (bench.py's
 * roofline figure).  No reference counterpart. */
int sim_profile(sim_handle* h, int enable);
int sim_profile_read(sim_handle* h, double* tick_kernel_ms, uint64_t* launches);
/* The same read with the spread: out_ms[0] = sum, [1] = min, [2] = max over the timed launches. */
int sim_profile_read_stats(sim_handle* h, double out_ms[3], uint64_t* launches);
/* Sums over the local shard's nodes (one reduction kernel; see sim_cluster_stats). */
int sim_cluster_stats_get(sim_handle* h, sim_cluster_stats* out);
/* How much of the handle's big arrays has memory behind it.  A view entry of slot a lives in plane a of the view array, a ring
 * bucket of Lamport time t in plane t mod ring size; the HIP library reserves the address range of all planes and gives a plane
 * physical memory when the host first hands out its slot / admits a Lamport time that reaches it (HIP virtual-memory API; a
 * plane must be a multiple of the mapping granularity — 128 Ki nodes per shard —, the view must be sparse; the rings of a
 * shard are whole; SERF_SIM_EAGER=1 in the environment turns it off).  Digests, dumps and images do not change: a plane without
 * memory is the zeros it stands for.  out = { view planes with memory, view planes, event-ring planes with memory, event-ring
 * planes, query-ring planes with memory, query-ring planes }; bytes_per_plane = 32 x the shard's nodes.  (The oracle keeps
 * whole arrays: resident == total.)  No reference counterpart. */
int sim_resident_planes(const sim_handle* h, uint32_t out[6], uint64_t* bytes_per_plane);

uint32_t sim_abi_version(void);
const char* sim_backend_name(void);

#ifdef __cplusplus
}
#endif
#endif /* SERF_SIM_H */
