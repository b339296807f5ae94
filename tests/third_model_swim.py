"""The memberlist half of the THIRD model — test infrastructure, pure Python, dictionary state, no capacity
bounds — written from SURVEY.md Appendix B (B.1 TransmitLimitedQueue, B.3 probe, B.4 alive / suspect / dead, B.5 suspicion timer) and
the simulator's SIMSPEC (DESIGN.md §2.1 tick order, §2.2 PRNG streams, §2.4 record lengths, §2.5 queue pooling, §2.7 probe phases,
§2.8 operations), NOT from oracle/serf_oracle.c's swim_* / queue functions and not from the HIP kernel.  memberlist-core's source is
not in /root/reference, so this cannot pin rows a13 / a16 to memberlist itself; what it removes is the single-author common mode:
the oracle and the kernel share one reading of Appendix B, this is a second one, and it runs CLOSED LOOP — it makes its own packets
out of its own queues (get_broadcasts: drain order, byte budget, retransmit limit), delivers them by memberlist's kRandomNodes
(third_model.k_random_nodes), loses them by the specified loss draw, runs its own suspicion timers and probes — and is compared with
the implementation under test after EVERY tick: the packets in flight, every queue in drain order (class, transmits, length, kind,
flags, key, value), clocks, SerfState, member tables, memberlist states, incarnations, confirmers, awareness, both de-dup rings.

The serf handlers are third_model.Node's (written from the Rust sources); handle_node_join / handle_node_leave below are restated
from serf/base.rs:1206-1334 and 1375-1440."""
import math

from tests import third_model as tm
from tests.third_model import mix64, M64, JOIN, LEAVE, EVENT, QUERY, NONE, ALIVE, LEAVING, LEFT, FAILED, S_ALIVE, S_LEAVING, S_LEFT

K_ALIVE, K_SUSPECT, K_DEAD = 5, 6, 7                        # memberlist's own broadcasts as the simulated packets carry them
ML_ALIVE, ML_SUSPECT, ML_DEAD, ML_LEFT = 0, 1, 2, 3         # memberlist node states (B.4)
PKT_UNITS = 1400 // 16                                      # the UDP payload budget in 16-byte units (SIMSPEC §2.4)
MAX_AWARENESS = 8                                           # awareness_max_multiplier (App. B defaults)
STREAM_LOSS, STREAM_PROBE = 4, 5                            # PRNG streams (SIMSPEC §2.2)
F_PRUNE = 1
F_META = 1                                                  # ALIVE: the node's tags changed with this incarnation


def units(nbytes):
    return min(63, (nbytes + 15) // 16)


def rng_base(seed, stream, a):
    return mix64(mix64(seed ^ ((stream * 0xD6E8FEB86659FD93) & M64)) ^ a)


def digits10(n):
    return len(str(n)) if n > 0 else 0


class Params:
    """what Appendix B derives from the configuration"""

    def __init__(self, n, fanout, probe_interval, suspicion_mult=4, suspicion_max_mult=6, indirect_checks=3, retransmit_mult=4,
                 loss=0.0, pkt_records=4, leave_delay=30, seed=None, push_pull_interval=0, reap_interval=0, reconnect_timeout=432000,
                 tombstone_timeout=432000, intent_timeout=0, queue_check_interval=0, max_queue_depth=4096, reconnect_interval=0,
                 awareness_probe=False, tcp_fallback=False, nacks=False, gossip_to_the_dead=0, join_sync=False, prune_delay=False):
        from serf_amd import _ffi
        self.prune_delay = prune_delay                      # handle_prune's sleep while the member is Leaving (serf/base.rs:1634-1639), leave_delay ticks
        self.pp_interval = push_pull_interval
        self.reap_interval, self.reconnect_timeout, self.tombstone_timeout, self.intent_timeout = reap_interval, reconnect_timeout, tombstone_timeout, intent_timeout
        self.queue_check_interval, self.max_queue_depth = queue_check_interval, max_queue_depth
        self.reconnect_interval = reconnect_interval if probe_interval else 0
        # memberlist behaviours behind switches (App. B.3, UPSTREAM-RECALL of state.go): the probe interval scaled by the health score, the fallback
        # ping over the stream transport, nacks from the relays, gossip_to_the_dead_time
        self.awareness_probe, self.tcp_fallback, self.nacks, self.gttd = awareness_probe, tcp_fallback, nacks, gossip_to_the_dead
        self.join_sync = join_sync                          # memberlist.join = a push-pull with the peer: the simulator lets the joiner ADOPT a running node's view (SIMSPEC §2.8)
        self.n, self.fanout, self.pi, self.ic, self.rmult = n, fanout, probe_interval, indirect_checks, retransmit_mult
        self.loss_u32 = min(0xFFFFFFFF, int(round(loss * 2 ** 32)))
        self.P, self.leave_delay = pkt_records, leave_delay
        self.seed = _ffi.DEFAULT_SEED if seed is None else seed
        # B.5: k confirmations shrink the timeout from max to min
        k = max(0, suspicion_mult - 2)
        self.k = 0 if n < 2 or n - 2 < k else min(k, 3)
        scale = max(1.0, math.log10(max(1, n)))
        mn = max(1, suspicion_mult * math.floor(scale * 1000.0) * probe_interval // 1000)
        mx = max(mn, suspicion_max_mult * mn)
        self.T = []
        for c in range(4):
            t = float(mn)
            if self.k >= 1 and c <= self.k:
                t = max(float(mn), math.floor(mx - math.log(c + 1.0) / math.log(self.k + 1.0) * (mx - mn)))
            self.T.append(int(t))


class SwimNode(tm.Node):
    def __init__(self, me, par: Params, ring_ev, ring_q, joined):
        super().__init__(me, par.n, ring_ev, ring_q, joined)
        self.par = par
        self.inc, self.awareness = 0, 0
        # memberlist's node map: subject -> [state, incarnation]; a suspicion: subject -> [start tick, confirmers (who started it first)]
        self.ml = {s: [ML_ALIVE, 0] for s in range(par.n)} if joined else {me: [ML_ALIVE, 0]}
        self.susp = {}
        self.slots = []                                     # the timers in the order they are looked at: first free place (SIMSPEC §2.7)
        self.queue, self.next_id = [], 0                    # TransmitLimitedQueue: [class, transmits, length, id, kind, flags, key, val]
        self.tick = 0
        self.left_at, self.intent_at = {}, {}               # MemberState.leave_time of the failed / left members, NodeIntent.wall_time (ticks)

    # ---- B.1 TransmitLimitedQueue, pooled (SIMSPEC §2.5: memberlist's queue < intents < queries < events) --------------------
    @staticmethod
    def _cls(kind):
        return 1 if kind in (JOIN, LEAVE) else 2 if kind == QUERY else 3 if kind == EVENT else 0

    def queue_broadcast(self, kind, flags, nbytes, key, val):
        cls = self._cls(kind)
        if cls == 0:                                        # a memberlist broadcast invalidates the queued one about the same node
            self.queue = [e for e in self.queue if not (e[0] == 0 and e[6] == key)]
        self.queue.append([cls, 0, units(nbytes), self.next_id, kind, flags, key, val])
        self.next_id += 1

    def drain_order(self):
        return sorted(self.queue, key=lambda e: (e[0], e[1], -e[2], -e[3]))

    def get_broadcasts(self):
        """one packet: walk the queue in drain order, take what still fits the byte budget, at most P records; transmits + 1,
        finished at the retransmit limit"""
        limit = self.par.rmult * digits10(len(self.members))
        free, out = PKT_UNITS, []
        for e in self.drain_order():
            if len(out) == self.par.P:
                break
            if e[2] > free:
                continue
            free -= e[2]
            out.append((e[4], e[5], e[2], e[6], e[7]))
            e[1] += 1
            if e[1] >= limit:
                self.queue.remove(e)
        return out

    # ---- serf's side of memberlist's notifications -----------------------------------------------------------------------------
    def upsert_intent(self, node, ty, ltime):               # serf/base.rs:1835-1866: a new or a newer intent is stamped with the time it came
        changed = super().upsert_intent(node, ty, ltime)
        if changed:
            self.intent_at[node] = self.tick
        return changed

    def handle_node_join(self, s):                          # serf/base.rs:1206-1334
        m = self.members.get(s)
        if m is not None:
            m[0] = ALIVE                                    # status_time stays
            self.left_at.pop(s, None)                       # (out of failed_members / left_members)
            return
        status, lt = ALIVE, 0
        it = self.intents.get(s)
        if it is not None and it[0] == JOIN:
            lt = it[1]
        if it is not None and it[0] == LEAVE:
            status, lt = LEAVING, it[1]
        self.members[s] = [status, lt]
        self.intents.pop(s, None)                           # (one entry per subject in the simulator: a member has no buffered intent)
        self.intent_at.pop(s, None)

    def handle_node_leave(self, s):                         # serf/base.rs:1375-1440
        m = self.members.get(s)
        if m is None:
            return
        if m[0] == LEAVING:
            m[0] = LEFT
            self.left_at[s] = self.tick                     # leave_time (base.rs:1384-1402): what the Reaper measures against
        elif m[0] == ALIVE:
            m[0] = FAILED
            self.left_at[s] = self.tick

    # ---- Reconnector::run, serf/base.rs:612-681: every reconnect_interval, when there are failed members: with probability failed / alive
    # pick one of them uniformly and memberlist.join it — a push-pull with that node, if it answers.  The simulator (SIMSPEC §2.8): the draws
    # are numbers 30 and 31 of the node's probe stream; the attempt goes on the tick's request list and runs two ticks later as a push-pull pair
    def reconnect(self):
        par = self.par
        if not par.reconnect_interval or (self.tick + (self.me >> 6)) % par.reconnect_interval:
            return None
        failed = [s for s in sorted(self.members) if self.members[s][0] == FAILED]
        if not failed:
            return None                                     # base.rs:640-642
        n_left = sum(1 for m in self.members.values() if m[0] == LEFT)
        alive = max(1, len(self.members) - len(failed) - n_left)   # base.rs:645-651
        base = rng_base(par.seed, STREAM_PROBE, self.tick)
        r = mix64(base ^ ((self.me * 32 + 30) & M64)) >> 32
        if r * alive > (len(failed) << 32):                 # "forgoing reconnect for random throttling"
            return None
        target = failed[((mix64(base ^ ((self.me * 32 + 31) & M64)) >> 32) * len(failed)) >> 32]
        return None if target == self.me else target

    # ---- QueueChecker::run, serf/base.rs:683-740: `if numq >= max { queue.prune(max) }` for each of serf's three queues (intents, queries,
    # events: classes 1 - 3 of the pooled queue); memberlist's TransmitLimitedQueue::prune keeps the `max` entries that drain first
    def queue_check(self):
        for cls in (1, 2, 3):
            mine = [e for e in self.drain_order() if e[0] == cls]
            for e in mine[self.par.max_queue_depth:]:
                self.queue.remove(e)

    # ---- Reaper::run, serf/base.rs:483-610 (reap! 521-553; reap_intents 1817-1822): every reap_interval ticks ----------------------------
    def reap(self):
        par = self.par
        for s in sorted(self.members):
            st = self.members[s][0]
            if st == FAILED:                                # a failed member (a leave intent may have made it LEFT since: then the tombstone timeout counts, from the same leave_time)
                timeout = par.reconnect_timeout
            elif st == LEFT:
                timeout = par.tombstone_timeout
            else:
                continue
            if self.tick - self.left_at[s] > timeout:       # erase_node!: out of the member table altogether
                del self.members[s]
                del self.left_at[s]
        if par.intent_timeout:
            for s in [s for s, at in self.intent_at.items() if s in self.intents and self.tick - at > par.intent_timeout]:
                del self.intents[s]
                del self.intent_at[s]
        self.prune_sync()

    # ---- B.4 ------------------------------------------------------------------------------------------------------------------
    def refute(self, accused_inc, flags=0):
        self.inc = max(self.inc + 1, accused_inc + 1)
        if self.me in self.ml:
            self.ml[self.me][1] = self.inc
        self.awareness = min(MAX_AWARENESS, self.awareness + 1)
        self.queue_broadcast(K_ALIVE, flags, 64, self.me, self.inc)

    def _cancel(self, s):
        if s in self.susp:
            del self.susp[s]
            self.slots[self.slots.index(s)] = None

    def alive(self, s, inc, rec):
        if s == self.me:
            if inc > self.inc:
                self.refute(inc)
            return
        cur = self.ml.get(s)
        if cur is None:
            self.ml[s] = [ML_ALIVE, inc]
            self.handle_node_join(s)
            self.queue_broadcast(*rec)
            return
        if inc <= cur[1]:
            return
        old = cur[0]
        self._cancel(s)
        cur[0], cur[1] = ML_ALIVE, inc
        self.queue_broadcast(*rec)
        if old in (ML_DEAD, ML_LEFT):
            self.handle_node_join(s)

    def suspect(self, s, inc, frm, rec):
        cur = self.ml.get(s)
        if cur is None or inc < cur[1]:
            return
        if cur[0] == ML_SUSPECT:                            # a timer exists: confirm(from); rebroadcast iff it was a new confirmation
            t = self.susp[s]
            if len(t[1]) - 1 >= self.par.k or frm in t[1]:
                return
            t[1].append(frm)
            self.queue_broadcast(*rec)
            return
        if cur[0] != ML_ALIVE:
            return
        if s == self.me:
            self.refute(inc)
            return
        self.queue_broadcast(*rec)
        cur[0], cur[1] = ML_SUSPECT, inc
        self.susp[s] = [self.tick, [frm]]
        if None in self.slots:
            self.slots[self.slots.index(None)] = s
        else:
            self.slots.append(s)

    def dead(self, s, inc, frm, rec):
        cur = self.ml.get(s)
        if cur is None or inc < cur[1]:
            return
        self._cancel(s)
        if cur[0] in (ML_DEAD, ML_LEFT):
            return
        if cur[0] == ML_SUSPECT:
            cur[0] = ML_ALIVE                               # (the timer is gone; what follows decides the state)
        if s == self.me and self.state not in (S_LEAVING, S_LEFT):
            self.refute(inc)
            return
        self.queue_broadcast(*rec)
        cur[0], cur[1] = (ML_LEFT if frm == s else ML_DEAD), inc
        self.handle_node_leave(s)

    # ---- B.5: the timers, in slot order; a timer that has run out declares the node dead, from = self -------------------------------
    def run_timers(self):
        for s in list(self.slots):
            if s is None or s not in self.susp:
                continue
            start, conf = self.susp[s]
            if self.tick - start >= self.par.T[len(conf) - 1]:
                self.dead(s, self.ml[s][1], self.me, (K_DEAD, 0, 32, s, self.ml[s][1] | (self.me << 32)))

    # ---- B.3: one probe per probe interval, the 64 nodes of an id-aligned group in the same phase (SIMSPEC §2.7) ---------------
    def probe(self, up):
        par = self.par
        if par.n < 2 or (self.tick + (self.me >> 6)) % par.pi:
            return
        if par.awareness_probe and ((self.tick + (self.me >> 6)) // par.pi) % (self.awareness + 1):
            return                                          # probeNode: the interval is (score + 1) x the configured one
        base = rng_base(par.seed, STREAM_PROBE, self.tick)

        def draw(j):
            return mix64(base ^ ((self.me * 32 + j) & M64))

        def below(d, n):
            return ((d >> 32) * n) >> 32

        def lost(j):
            return bool(par.loss_u32) and (draw(j) >> 32) < par.loss_u32

        t = below(draw(0), par.n - 1)
        if t >= self.me:
            t += 1
        cur = self.ml.get(t)
        if cur is None or cur[0] in (ML_DEAD, ML_LEFT):
            return
        ok = False
        if up[t]:
            ok = not lost(1) and not lost(2)
            for j in range(min(par.ic, 4)):
                if ok:
                    break
                r = below(draw(3 + 5 * j), par.n)
                if r == self.me or r == t or not up[r]:
                    continue
                ok = not lost(3 + 5 * j + 1) and not lost(3 + 5 * j + 2) and not lost(3 + 5 * j + 3) and not lost(3 + 5 * j + 4)
            if not ok and par.tcp_fallback:                 # the fallback ping over the stream transport reaches a running node: "didContact"
                ok = True
        if ok:
            self.awareness = max(0, self.awareness - 1)
            return
        delta = 1
        if par.nacks:                                       # awarenessDelta = nacks expected - nacks received (no relay asked: + 1)
            expected = got = 0
            for j in range(min(par.ic, 4)):
                r = below(draw(3 + 5 * j), par.n)
                if r == self.me or r == t:
                    continue
                expected += 1
                if up[r] and not lost(3 + 5 * j + 1) and not lost(3 + 5 * j + 4):
                    got += 1
            delta = expected - got if expected else 1
        self.awareness = max(0, min(MAX_AWARENESS, self.awareness + delta))
        self.suspect(t, cur[1], self.me, (K_SUSPECT, 0, 32, t, cur[1] | (self.me << 32)))

    # ---- SerfDelegate::notify_message with the original message re-queued unchanged (delegate.rs:294-300) ----------------------------
    def receive(self, kind, flags, length, key, val):
        rec = (kind, flags, length * 16, key, val)
        if kind == K_ALIVE:
            self.alive(key, val & 0xFFFFFFFF, rec)
        elif kind == K_SUSPECT:
            self.suspect(key, val & 0xFFFFFFFF, val >> 32, rec)
        elif kind == K_DEAD:
            self.dead(key, val & 0xFFFFFFFF, val >> 32, rec)
        else:
            self.rebroadcast = []
            self.notify(kind, key, val, flags)              # third_model.Node: the serf handlers
            self._flush(default=rec)

    def _flush(self, default=None, lens=None):
        """what the serf handlers asked to (re)broadcast, into the queue: the received message as it came, or a message of this
        node's own making with the length the codec gives it (SIMSPEC §2.4)"""
        for kind, key, lt in self.rebroadcast:
            if default is not None and (kind, key, lt) == (default[0], default[3], default[4]):
                self.queue_broadcast(*default)
            else:                                           # a refutation (broadcast_join) the handler queued on the way
                fl, nb = (lens or {}).get(kind, (0, 16))
                self.queue_broadcast(kind, fl, nb, key, lt)
        self.rebroadcast = []

    # ---- B.6 push-pull: memberlist's mergeState, then SerfDelegate::merge_remote_state (third_model.Node) ----------------------------------
    def push_pull_merge(self, remote):
        """local <- remote.  memberlist hands every node state of the remote to the state machine (SIMSPEC §2.10): an alive node as an
        alive message, a node that left as dead{from = the node}, a suspect or dead one as a suspicion raised by the merging node itself;
        then the serf delegate merges the remote's local_state.  What the handlers queue is queued (a merge tells the cluster what it
        learnt); of the serf half only a refutation is."""
        for subj in sorted(remote.members):
            st, inc = remote.ml[subj]
            if st == ML_ALIVE:
                self.alive(subj, inc, (K_ALIVE, 0, 64, subj, inc))
            elif st == ML_LEFT:
                self.dead(subj, inc, subj, (K_DEAD, 0, 32, subj, inc | (subj << 32)))
            else:
                self.suspect(subj, inc, self.me, (K_SUSPECT, 0, 32, subj, inc | (self.me << 32)))
        self.rebroadcast = []
        self.merge_remote_state(remote.local_state())
        self._flush()

    def prune_sync(self):
        """erase_node! also forgets memberlist's node state in the simulator (waived / simplified)"""
        for s in list(self.ml):
            if s != self.me and s not in self.members:
                self._cancel(s)
                del self.ml[s]


class Cluster:
    """the tick of SIMSPEC §2.1 over SwimNodes: operations, deliveries, timers, probe, drain"""

    def __init__(self, par: Params, ring_ev, ring_q, joined):
        self.par = par
        self.nodes = [SwimNode(i, par, ring_ev, ring_q, joined) for i in range(par.n)]
        self.up = [True] * par.n
        self.tick = 0
        self.flight = None                                  # packets sent during the last tick: [sender][slot] -> list of records (None: not sent)
        self.rc_made = {}                                   # tick -> the reconnect attempts (initiator, target) made in it, by initiator
        self.rc_postponed = []                              # attempts that found a partner busy (or a batch tick): next tick, first
        self.prune_due = {}                                 # tick -> the (node, subject) whose forced erase ends its wait then
        if par.prune_delay:
            for x in self.nodes:
                x.prune_wait = []

    def apply(self, op, node, a, b):
        from serf_amd import _ffi
        x = self.nodes[node]
        if op == _ffi.OP_CRASH:
            x.up = self.up[node] = False
        elif op == _ffi.OP_REVIVE:
            x.up = self.up[node] = True
        elif op == _ffi.OP_JOIN:                            # memberlist.join + Serf::join (api.rs:318-364): a fresh incarnation, then broadcast_join
            self.up[node] = True
            if self.par.join_sync:
                self._adopt(x, a)
            x.up, x.state = True, S_ALIVE
            me = x.ml.get(node)
            old = me[0] if me else ML_ALIVE
            if me:
                x._cancel(node)
                me[0] = ML_ALIVE
            x.refute(me[1] if me else x.inc)
            x.awareness = max(0, x.awareness - 1)           # not an accusation
            if old in (ML_DEAD, ML_LEFT):
                x.handle_node_join(node)
            x.rebroadcast = []
            x.broadcast_join(x.clock.time())
            x._flush()
        elif op == _ffi.OP_LEAVE_FINISH:                    # memberlist.leave: dead{self, from = self}, then SerfState::Left (api.rs:474-497)
            if x.up and x.state == S_LEAVING:
                x.dead(node, x.inc, node, (K_DEAD, 0, 32, node, x.inc | (node << 32)))
                x.state = S_LEFT
        elif not x.up:
            return
        elif op == _ffi.OP_SET_TAGS:                        # Serf::set_tags (api.rs:219-235): memberlist.update_node — the next incarnation and an
            if self.par.pi:                                 # alive broadcast whose meta differs (the receivers' notify_update, base.rs:1574-1625)
                x.refute(x.inc, F_META)
                x.awareness = max(0, x.awareness - 1)       # not an accusation
        elif op == _ffi.OP_USER_EVENT:
            x.rebroadcast = []
            x.user_event(a)
            x._flush(lens={EVENT: (0, b & 0x7FFFFFFF)})
        elif op == _ffi.OP_QUERY:
            x.rebroadcast = []
            x.query(a, b & 15)
            x._flush(lens={QUERY: (b & 15, 48)})
        elif op == _ffi.OP_LEAVE:
            x.rebroadcast = []
            x.leave(others_alive=self.par.n > 1)
            x._flush()
        elif op == _ffi.OP_FORCE_LEAVE:
            x.rebroadcast = []
            x.force_leave(a, bool(b), others_alive=self.par.n > 1)
            x._flush(lens={LEAVE: (F_PRUNE if b else 0, 16)})
            x.prune_sync()

    def _adopt(self, x, peer):
        """SIM_CF_JOIN_SYNC: the joining node takes over the view of the first running node at or after `peer` that is not itself — every
        entry but its own (of that it keeps the higher incarnation), the suspicions with their timers, the buffered intents —, and witnesses
        the partner's clocks one below their value (delegate.rs:466-480)"""
        n = self.par.n
        partner = next((c for c in ((peer % n + i) % n for i in range(n)) if c != x.me and self.up[c]), None)
        if partner is None:
            return
        p = self.nodes[partner]
        own = {"m": x.members.get(x.me), "ml": x.ml.get(x.me), "left": x.left_at.get(x.me), "it": x.intents.get(x.me), "ia": x.intent_at.get(x.me)}
        seen_inc = p.ml[x.me][1] if x.me in p.ml else None
        x.members = {s: list(v) for s, v in p.members.items() if s != x.me}
        x.ml = {s: list(v) for s, v in p.ml.items() if s != x.me}
        x.left_at = {s: v for s, v in p.left_at.items() if s != x.me}
        x.intents = {s: list(v) for s, v in p.intents.items() if s != x.me}
        x.intent_at = {s: v for s, v in p.intent_at.items() if s != x.me}
        x.susp = {s: [v[0], list(v[1])] for s, v in p.susp.items() if s != x.me}
        for key, d in (("m", x.members), ("ml", x.ml), ("left", x.left_at), ("it", x.intents), ("ia", x.intent_at)):
            if own[key] is not None:
                d[x.me] = own[key]
        if seen_inc is not None and x.me in x.ml and seen_inc > x.ml[x.me][1]:
            x.ml[x.me][1] = seen_inc
        x.slots = sorted(x.susp)                            # the adopted suspicions keep running, looked at in subject order
        for mine, theirs in ((x.clock, p.clock), (x.event_clock, p.event_clock), (x.query_clock, p.query_clock)):
            if theirs.time() > 0:
                mine.witness(theirs.time() - 1)

    def step(self, ops):
        par, t = self.par, self.tick
        for x in self.nodes:
            x.tick = t
        for op, node, a, b in ops:
            self.apply(op, node, a, b)
        # (0') the forced erases whose wait ends now (handle_prune, serf/base.rs:1636-1652: the member goes, whatever it has become), behind the tick's
        # operations, by (node, subject); the sleeps begun during this tick's operations are noted for their tick
        for node, subject in sorted(self.prune_due.pop(t, [])):
            x = self.nodes[node]
            if x.up and subject in x.members:
                del x.members[subject]
                x.left_at.pop(subject, None)
                x.intents.pop(subject, None)
                x.intent_at.pop(subject, None)
                x.prune_sync()
        # (0a) the reconnect attempts due now — postponed ones first, then the ones made two ticks ago — run as push-pull pairs of their own,
        # the initiator merging first; on a batch tick, or when one of the two is already in a pair of this tick, an attempt waits a tick
        due = self.rc_postponed + sorted(self.rc_made.pop(t - 2, []))
        self.rc_postponed = []
        batch = bool(par.pp_interval) and tm.push_pull_batch_tick(t, par.n, par.pp_interval)
        taken, pairs = set(), []
        for a, b in due:
            if a == b or not self.up[a] or not self.up[b]:
                continue
            if batch or a in taken or b in taken:
                self.rc_postponed.append((a, b))
                continue
            taken.update((a, b))
            pairs.append((a, b))
        for a, b in pairs:
            self.nodes[a].push_pull_merge(self.nodes[b])
            self.nodes[b].push_pull_merge(self.nodes[a])
        # (0b) the tick's push-pull batch: both processes running; `a` merges first, then `b` merges a's updated state
        for a, b in tm.push_pull_pairs(par.seed, t, par.n, par.pp_interval):
            if self.up[a] and self.up[b]:
                self.nodes[a].push_pull_merge(self.nodes[b])
                self.nodes[b].push_pull_merge(self.nodes[a])
        # gossip_to_the_dead_time: whom a node does not gossip to this tick — a member it has believed dead (or gone) for longer than that,
        # by its view as the tick's deliveries begin
        skip = []
        for i, x in enumerate(self.nodes):
            sk = set()
            if par.gttd:
                for k, tgt in enumerate(tm.k_random_nodes(par.seed, t, i, par.n, min(par.fanout, par.n - 1))):
                    st = x.ml.get(tgt)
                    if tgt in x.members and st is not None and st[0] in (ML_DEAD, ML_LEFT) and t - x.left_at.get(tgt, t) > par.gttd:
                        sk.add(k)
            skip.append(sk)
        # (1) deliveries: every packet addressed to the node, (sender, slot) order, records in packet order
        if self.flight is not None:
            rows = [[] for _ in range(par.n)]
            for snd in range(par.n):
                for k, tgt in enumerate(tm.k_random_nodes(par.seed, t - 1, snd, par.n, min(par.fanout, par.n - 1))):
                    rows[tgt].append((snd, k))
            for i, x in enumerate(self.nodes):
                if not x.up:
                    continue
                for snd, k in rows[i]:
                    for rec in self.flight[snd][k] or ():
                        x.receive(*rec)
                        x.prune_sync()
        # (2) timers, (3) probe, (5) drain
        loss_base = rng_base(par.seed, STREAM_LOSS, t)
        flight = []
        for i, x in enumerate(self.nodes):
            pk = [None] * par.fanout
            if x.up:
                if par.pi:
                    x.run_timers()
                    x.probe(self.up)
                if par.reap_interval and (t + (i >> 6)) % par.reap_interval == 0:   # (the phase is shared by a group of 64 nodes, like the probe's)
                    x.reap()
                target = x.reconnect()
                if target is not None:
                    self.rc_made.setdefault(t, []).append((i, target))
                if par.queue_check_interval and (t + (i >> 6)) % par.queue_check_interval == 0:
                    x.queue_check()
                targets = tm.k_random_nodes(par.seed, t, i, par.n, min(par.fanout, par.n - 1))
                for k in range(min(par.fanout, par.n - 1)):
                    recs = x.get_broadcasts()
                    lost = bool(par.loss_u32) and (mix64(loss_base ^ ((i * 4 + k) & M64)) >> 32) < par.loss_u32
                    if k < len(targets) and not lost and k not in skip[i]:
                        pk[k] = recs
            flight.append(pk)
        self.flight = flight
        if par.prune_delay:                                 # the sleeps begun in this tick (its operations and deliveries) end max(2, leave_delay) ticks from it
            for i, x in enumerate(self.nodes):
                for subject in x.prune_wait:
                    self.prune_due.setdefault(t + max(2, par.leave_delay), []).append((i, subject))
                x.prune_wait = []
        self.tick += 1
