"""A THIRD, independent model of serf-core's per-message handlers — test infrastructure, pure Python, dictionary state like the
reference's `HashMap`s, no capacity bounds — written from the Rust sources alone (not from oracle/serf_oracle.c, not from the
HIP kernel), so that a misreading the oracle and the kernel share (they have one author and one reading of base.rs) has
somewhere to show: tests/test_third_model.py feeds it the very record stream the oracle (or the HIP library) delivers at
N <= 64 and compares clocks, member table, intents, de-dup rings and every rebroadcast decision, node by node, tick by tick.

Reference (relative to /root/reference/serf-core/src/):
  LamportClock            types/clock.rs:125-172   (new = 0, `Serf::new` increments every clock once: serf/base.rs:196-205)
  handle_user_event       serf/base.rs:750-837
  handle_query (de-dup)   serf/base.rs:972-1073
  handle_node_join_intent serf/base.rs:1338-1373
  handle_node_leave_intent serf/base.rs:1442-1572
  upsert_intent           serf/base.rs:1835-1866
  broadcast_join          serf/base.rs:381-397
  Serf::user_event        serf/api.rs:241-299     Serf::query  serf/base.rs:875-942
  Serf::leave             serf/api.rs:422-460     force_leave  serf/base.rs:452-480     Serf::join  serf/api.rs:318-364
"""
NONE, ALIVE, LEAVING, LEFT, FAILED = 0, 1, 2, 3, 4          # MemberStatus, types/member.rs:54-87
S_ALIVE, S_LEAVING, S_LEFT, S_SHUTDOWN = 0, 1, 2, 3         # SerfState, serf.rs:80-89
JOIN, LEAVE, EVENT, QUERY = 1, 2, 3, 4                      # the record kinds of the simulated packets
NO_BROADCAST = 1                                            # QueryFlag::NO_BROADCAST as the simulator carries it


M64 = (1 << 64) - 1


def mix64(z):
    """splitmix64's finaliser (DESIGN.md SIMSPEC §2.2: the simulator's only source of randomness)"""
    z = (z + 0x9E3779B97F4A7C15) & M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def k_random_nodes(seed, tick, node, n, fanout):
    """memberlist's kRandomNodes as the simulator specifies it (SIM_CF_RANDOM_FANOUT): up to 3 n uniform
    draws over all n nodes from stream 7 of (seed, tick), skipping the node itself and nodes already chosen, until `fanout` are
    found — written from the specification, not from the oracle's rf_draw."""
    base = mix64(mix64(seed ^ ((7 * 0xD6E8FEB86659FD93) & M64)) ^ tick)
    chosen = []
    for i in range(3 * n):
        if len(chosen) == fanout:
            break
        t = ((mix64(base ^ ((node * 4096 + i) & M64)) >> 32) * n) >> 32
        if t != node and t not in chosen:
            chosen.append(t)
    return chosen


def push_pull_batch_tick(tick, n, interval, groups=8):
    """is `tick` one of the ticks a class of the push-pull matching synchronises in (see push_pull_pairs)"""
    if not interval:
        return False
    mult = 1
    if n > 32:
        import math
        mult = math.ceil(math.log2(n) - 5.0) + 1
    step = max(1, interval * mult // groups)
    return tick > 0 and tick % step == 0


def push_pull_pairs(seed, tick, n, interval, groups=8):
    """The pairs of the tick's push-pull batch as the simulator specifies them (DESIGN.md SIMSPEC §2.10; memberlist's pushPull picks
    any peer, App. B.6): the interval is scaled by memberlist's pushPullScale (x (ceil(log2 n - 5) + 1) above 32 nodes) and cut into
    `groups` classes; every step = interval / groups ticks one class of the tick's perfect matching {sigma^-1(2 p), sigma^-1(2 p + 1)}
    synchronises, p = class, class + groups ... — sigma the tick's cycle-walking three-round multiply-xorshift bijection of the node ids,
    its multipliers and offsets drawn from stream 1 of (seed, tick).  `a` merges first, then `b` merges a's updated state.
    Written from the specification, not from the oracle's pp_pair_at."""
    if not interval:
        return []
    mult = 1
    if n > 32:
        import math
        mult = math.ceil(math.log2(n) - 5.0) + 1
    step = max(1, interval * mult // groups)
    if tick == 0 or tick % step:
        return []
    cls = (tick // step) % groups
    bits = max(1, (n - 1).bit_length())
    mask, shift = (1 << bits) - 1, (bits + 1) // 2
    base = mix64(mix64(seed ^ ((1 * 0xD6E8FEB86659FD93) & M64)) ^ tick)
    w = [mix64(base ^ r) for r in range(3)]
    mul = [(x & 0xFFFFFFFF) | 1 for x in w]
    add = [x >> 32 for x in w]
    imul = [pow(m, -1, 1 << 32) for m in mul]

    def inv(y):
        while True:
            y = (((y - add[2]) & 0xFFFFFFFF) * imul[2]) & mask
            y ^= y >> shift
            y = (((y - add[1]) & 0xFFFFFFFF) * imul[1]) & mask
            y ^= y >> shift
            y = (((y - add[0]) & 0xFFFFFFFF) * imul[0]) & mask
            if y < n:
                return y

    pairs, p = [], cls
    while 2 * p + 1 < n:
        pairs.append((inv(2 * p), inv(2 * p + 1)))
        p += groups
    return pairs


class QueryTrackers:
    """The origin's side of a query (serf/base.rs:875-942 registers the QueryResponse before the query is sent; handle_query_response
    base.rs:1158-1204 and QueryResponse::handle_query_response query.rs:240-303 count what comes back: nothing after the deadline or
    when the origin is not running, one ack and one response per sender) and the responder's (base.rs:1075-1154: an ack when the query
    asks for one, the response when the user code calls respond(); both straight to the origin over memberlist.send, subject to packet
    loss; relay_response query.rs:523-601: up to relay_factor more tries through a random live member, two more legs each) — as the
    simulator specifies the draws (DESIGN.md SIMSPEC: stream 6 of (seed, tick), keyed by the query id; lane = node x 64 + 32 for the
    response; draw 0 = the direct leg, 1 + 3 r = relay r's choice, 2 + 3 r and 3 + 3 r = its two legs).  Written from the Rust sources and
    the specification, not from the oracle's query_respond."""
    ACK, RESPOND = 2, 4

    def __init__(self, seed, n, loss):
        self.seed, self.n = seed, n
        self.loss_u32 = min(0xFFFFFFFF, int(round(loss * 2 ** 32)))
        self.timeout = 16 * len(str(n))                      # query.rs:421-427: gossip_interval x query_timeout_mult (16) x ceil(log10(n + 1)) ticks
        self.running = {}                                   # query id -> [origin, deadline, flags, ackers, responders]

    def register(self, qid, origin, flags, tick):
        self.running[qid] = [origin, tick + self.timeout, flags, set(), set()]

    def respond(self, node, qid, flags, tick, up):
        tr = self.running.get(qid)
        if tr is None or not flags & (self.ACK | self.RESPOND):
            return
        origin, deadline, qflags = tr[0], tr[1], tr[2]
        if tick > deadline or not up[origin]:
            return
        base = mix64(mix64(mix64(self.seed ^ ((6 * 0xD6E8FEB86659FD93) & M64)) ^ tick) ^ ((qid << 32) & M64))
        relay = (qflags >> 8) & 7
        if self.n < relay + 1:                              # "members.states.len() < relay_factor + 1": no relays
            relay = 0
        for which, bit in ((0, self.ACK), (1, self.RESPOND)):
            if not flags & bit:
                continue
            lane = node * 64 + which * 32

            def lost(i):
                return bool(self.loss_u32) and (mix64(base ^ (lane + i)) >> 32) < self.loss_u32

            ok = not lost(0)
            for r in range(relay):
                if ok:
                    break
                via = ((mix64(base ^ (lane + 1 + 3 * r)) >> 32) * self.n) >> 32
                if via != node and up[via]:
                    ok = not lost(2 + 3 * r) and not lost(3 + 3 * r)
            if ok:
                tr[3 + which].add(node)

    def status(self, qid, tick_now):
        tr = self.running[qid]
        return len(tr[3]), len(tr[4]), tick_now <= tr[1]


class Clock:
    """types/clock.rs:125-172"""

    def __init__(self):
        self.v = 0

    def time(self):
        return self.v

    def increment(self):
        self.v += 1
        return self.v

    def witness(self, t):
        if t < self.v:
            return
        self.v = t + 1


class Node:
    def __init__(self, me, n, ring_ev, ring_q, everybody_joined):
        self.me, self.up, self.state = me, True, S_ALIVE
        self.clock, self.event_clock, self.query_clock = Clock(), Clock(), Clock()
        for c in (self.clock, self.event_clock, self.query_clock):
            c.increment()                                   # serf/base.rs:196-205
        self.members = {}                                   # id -> [status, status_time]   (Members.states)
        self.intents = {}                                   # id -> [type, ltime]           (Members.recent_intents)
        self.event_buf = [None] * ring_ev                   # EventCore.buffer: Option<UserEvents{ltime, events}>
        self.query_buf = [None] * ring_q                    # QueryCore.buffer: Option<Queries{ltime, query_ids}>
        self.event_min = self.query_min = 0
        self.rebroadcast = []                               # (kind, key, ltime) the delegate re-queues, in order
        self.prune_wait = None                              # a list: handle_prune's sleep is modelled (SIM_CF_PRUNE_DELAY) — the subjects whose erase waits, in order
        self.on_query = None                                # the responder's half (QueryTrackers.respond), when the harness models it
        if everybody_joined:                                # the simulator's pre-joined baseline: every member Alive at status_time 1,
            for s in range(n):                              # the own join at ltime 1 witnessed
                self.members[s] = [ALIVE, 1]
            self.clock.witness(1)
        else:
            self.members[me] = [ALIVE, 0]                   # new_in's synthetic notify_join(local)

    # ---- serf/base.rs:1835-1866
    def upsert_intent(self, node, ty, ltime):
        cur = self.intents.get(node)
        if cur is not None:
            if ltime > cur[1]:
                cur[0], cur[1] = ty, ltime
                return True
            return False
        self.intents[node] = [ty, ltime]
        return True

    # ---- serf/base.rs:1338-1373
    def handle_node_join_intent(self, node, ltime):
        self.clock.witness(ltime)
        member = self.members.get(node)
        if member is None:
            return self.upsert_intent(node, JOIN, ltime)
        if ltime <= member[1]:
            return False
        member[1] = ltime
        if member[0] == LEAVING:
            member[0] = ALIVE
        return True

    # ---- serf/base.rs:381-397
    def broadcast_join(self, ltime):
        self.clock.witness(ltime)
        self.handle_node_join_intent(self.me, ltime)
        self.rebroadcast.append((JOIN, self.me, ltime))

    # ---- serf/base.rs:1442-1572
    def handle_node_leave_intent(self, node, ltime, prune=False):
        state = self.state
        self.clock.witness(ltime)
        if node not in self.members:
            return self.upsert_intent(node, LEAVE, ltime)
        member = self.members[node]
        if ltime <= member[1]:
            return False
        if node == self.me and state == S_ALIVE:            # refute: a join at the current clock, nothing rebroadcast
            self.broadcast_join(self.clock.time())
            return False
        member[1] = ltime
        st = member[0]
        if st == NONE:
            return False
        if st == ALIVE:
            member[0] = LEAVING
        elif st == FAILED:
            member[0] = LEFT
        elif st not in (LEAVING, LEFT):
            member[0] = LEAVING
        if prune:                                           # handle_prune (serf/base.rs:1628-1653): a Leaving member after a sleep of broadcast_timeout + leave_propagate_delay,
            if self.prune_wait is not None and member[0] == LEAVING:    # anybody else at once
                self.prune_wait.append(node)
            else:
                del self.members[node]
                self.intents.pop(node, None)
        return True

    # ---- serf/base.rs:750-837
    def handle_user_event(self, key, ltime):
        self.event_clock.witness(ltime)
        if ltime < self.event_min:
            return False
        b = len(self.event_buf)
        cur = self.event_clock.time()
        if cur > b and ltime < cur - b:
            return False
        idx = ltime % b
        seen = self.event_buf[idx]
        if seen is not None:
            if key in seen[1]:                              # (the bucket's ltime is NOT compared: base.rs:801-806)
                return False
            seen[1].append(key)
        else:
            self.event_buf[idx] = [ltime, [key]]
        return True

    # ---- serf/base.rs:972-1073 (the filters, acks and responses behind it are not part of this model)
    def handle_query(self, qid, ltime, flags):
        self.query_clock.witness(ltime)
        if ltime < self.query_min:
            return False
        cur = self.query_clock.time()
        q_time = len(self.query_buf)
        if cur > q_time and q_time < cur - q_time:          # (sic: base.rs:1013)
            return False
        idx = ltime % q_time
        seen = self.query_buf[idx]
        if seen is not None:
            if seen[0] == ltime and qid in seen[1]:
                return False
            seen[1].append(qid)                             # (the bucket keeps its old ltime: base.rs:1035)
        else:
            self.query_buf[idx] = [ltime, [qid]]
        if self.on_query is not None:                       # base.rs:1075-1154: ack / respond (no filters in this model: every node processes)
            self.on_query(self.me, qid, flags)
        return not (flags & NO_BROADCAST)

    # ---- SerfDelegate::notify_message, serf/delegate.rs:183-300: dispatch, re-queue the original message when told to
    def notify(self, kind, key, ltime, flags):
        if kind == LEAVE:
            rb = self.handle_node_leave_intent(key, ltime, bool(flags & 1))
        elif kind == JOIN:
            rb = self.handle_node_join_intent(key, ltime)
        elif kind == EVENT:
            rb = self.handle_user_event(key, ltime)
        elif kind == QUERY:
            rb = self.handle_query(key, ltime, flags)
        else:
            return
        if rb:
            self.rebroadcast.append((kind, key, ltime))

    # ---- SerfDelegate::local_state, serf/delegate.rs:386-425: what a push-pull ships
    def local_state(self):
        return {"ltime": self.clock.time(), "event_ltime": self.event_clock.time(), "query_ltime": self.query_clock.time(),
                "status_ltimes": {m: st[1] for m, st in self.members.items()},
                "left_members": [m for m, st in self.members.items() if st[0] == LEFT],
                "events": [None if b is None else (b[0], list(b[1])) for b in self.event_buf]}

    # ---- SerfDelegate::merge_remote_state(is_join = false), serf/delegate.rs:427-554
    def merge_remote_state(self, pp):
        for clock, t in ((self.clock, pp["ltime"]), (self.event_clock, pp["event_ltime"]), (self.query_clock, pp["query_ltime"])):
            if t > 0:
                clock.witness(t - 1)                        # "no message with that clock has been sent yet"
        for node in pp["left_members"]:                     # the left nodes first, one past their status time (:488-512)
            if node in pp["status_ltimes"]:
                self.handle_node_leave_intent(node, pp["status_ltimes"][node] + 1)
        for node, ltime in pp["status_ltimes"].items():     # every other status time as a join intent (:515-526)
            if node not in pp["left_members"]:
                self.handle_node_join_intent(node, ltime)
        for bucket in pp["events"]:                         # the event buffer, replayed (:540-552); nothing of this is rebroadcast
            if bucket is not None:
                for key in bucket[1]:
                    self.handle_user_event(key, bucket[0])

    # ---- the user-facing calls
    def user_event(self, key):                              # serf/api.rs:241-299
        ltime = self.event_clock.time()
        self.event_clock.increment()
        self.handle_user_event(key, ltime)
        self.rebroadcast.append((EVENT, key, ltime))

    def query(self, qid, flags):                            # serf/base.rs:875-942
        ltime = self.query_clock.time()
        self.handle_query(qid, ltime, flags)
        self.rebroadcast.append((QUERY, qid, ltime))

    def leave(self, others_alive=True):                     # serf/api.rs:422-460
        if self.state != S_ALIVE:
            return
        self.state = S_LEAVING
        ltime = self.clock.time()
        self.clock.increment()
        self.handle_node_leave_intent(self.me, ltime)
        if others_alive:
            self.rebroadcast.append((LEAVE, self.me, ltime))

    def force_leave(self, subject, prune, others_alive=True):   # serf/base.rs:452-480
        ltime = self.clock.time()
        self.handle_node_leave_intent(subject, ltime, prune)
        if others_alive:
            self.rebroadcast.append((LEAVE, subject, ltime))

    def join(self):                                         # serf/api.rs:318-364 behind memberlist.join
        self.up, self.state = True, S_ALIVE
        self.broadcast_join(self.clock.time())
