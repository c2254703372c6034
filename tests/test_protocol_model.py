"""CPU model of the direct peer exchange's hand-off protocol (csrc/prcg_kernels.h: PeerDev,
csrc/prcg_device.hpp: peer_send_slot / peer_collect, csrc/prcg_win.hip: communication wave, k_peer_push / k_peer_collect).

No hardware has run this path on more than one GPU, so the ORDERING argument is checked here on a model: R ranks, each a
stream of kernels; a kernel is a set of concurrent actors (communication wave, tile waves); every remote store is its own
event; a randomised scheduler interleaves the ranks' events with one rank far slower than the others.  What must hold for
every interleaving:
  * a rank that has seen all R counters of iteration k reads exactly the R partial-sum payloads of iteration k of THIS
    session (payload before counter; two parities are enough; an earlier session's slots never satisfy a wait: epoch);
  * a tile that reads its ghost rows after the publication of iteration k-1 reads every neighbour's rows of iteration k-1
    (no neighbour can have overwritten that parity yet: its stores of iteration k+1 wait for MY slot of iteration k);
  * a session's first stores overwrite the previous session's last slot and ghost parity: they may only leave once every
    rank has finished that session.  prcg_solve_begin guarantees it: its set-up inner products are all-reduced over RCCL
    (every rank joins the collective only after its last prcg_iterate call has returned) BEFORE the rank's first push.
    (The model found this: without that barrier a rank that has seen a counter of the old session's last collect can
    read the new session's payload -- a window of microseconds against a host round trip, but a hole in the argument.)
The same model with the protocol BROKEN (one parity; rows stored before the publication; counter before payload; no
barrier between sessions) must fail: the test has teeth."""
import random

import pytest


class Stale(Exception):
    pass


class Model:
    def __init__(self, R, neighbours, parities=2, wait_before_stores=True, payload_first=True, session_barrier=True, seed=0, slow_rank=0, slow_weight=0.03):
        self.R, self.nb = R, neighbours
        self.P = parities
        self.wait_before_stores, self.payload_first, self.session_barrier = wait_before_stores, payload_first, session_barrier
        self.arrived = {}                      # session -> ranks that reached its start (the set-up all-reduce of prcg_solve_begin)
        self.rng = random.Random(seed)
        self.weight = [slow_weight if q == slow_rank else 1.0 for q in range(R)]
        # exchange buffer of every rank: per parity R slots [payload, counter] and one ghost entry per neighbour
        self.slot = [[[[None, 0] for _ in range(R)] for _ in range(self.P)] for _ in range(R)]
        self.ghost = [[{q: None for q in neighbours[r]} for _ in range(self.P)] for r in range(R)]
        self.pub = [(-1, None)] * R            # per rank: (iteration published to the rank's tile waves, session)
        self.checked = 0

    # ---- what the kernels do, as generators that yield before every globally visible step ----
    def send_slot(self, me, E, k):
        steps = [(0, ('sum', E, me, k)), (1, (E << 32) | (k + 1))]     # payload into EVERY rank's buffer, drained, then the counters
        for field, value in (steps if self.payload_first else steps[::-1]):
            for q in range(self.R):
                yield
                self.slot[q][k % self.P][me][field] = value

    def collect(self, me, E, k):
        want = (E << 32) | (k + 1)
        for q in range(self.R):
            while self.slot[me][k % self.P][q][1] < want:              # bounded spin in the kernel; here: until it arrives
                yield
        for q in range(self.R):                                        # counters seen -> payload loads (acquire between)
            yield
            got = self.slot[me][k % self.P][q][0]
            if got != ('sum', E, q, k):
                raise Stale(f'rank {me} session {E}: slot of rank {q} for iteration {k} holds {got}')
            self.checked += 1
        self.pub[me] = (k, E)

    def push_rows(self, me, E, k):
        for q in self.nb[me]:
            yield
            self.ghost[q][k % self.P][me] = ('rows', E, me, k)

    def read_ghosts(self, me, E, k):
        for q in self.nb[me]:
            yield
            got = self.ghost[me][k % self.P][q]
            if got != ('rows', E, q, k):
                raise Stale(f'rank {me} session {E}: ghost rows of rank {q} for iteration {k} hold {got}')
            self.checked += 1

    def tiles(self, me, E, k):
        # interior tiles first (nothing to wait for), then: wait for the publication of iteration k-1 -> deferred updates and all
        # stores (rows of iteration k to the neighbours) -> boundary tiles read the ghost rows of iteration k-1
        if self.wait_before_stores:
            while self.pub[me] != (k - 1, E):
                yield
        yield from self.push_rows(me, E, k)
        while self.pub[me] != (k - 1, E):
            yield
        yield from self.read_ghosts(me, E, k - 1)

    def comm_wave(self, me, E, k, pending):
        if pending:                                                    # iteration k-1's sums are still this rank's block partials
            yield from self.send_slot(me, E, k - 1)
            yield from self.collect(me, E, k - 1)

    def rank_program(self, me, sessions):
        """the rank's stream: kernels run one after the other; the actors of one kernel run concurrently"""
        for E, calls in enumerate(sessions, start=1):
            # prcg_solve_begin: the set-up products' inner products are all-reduced (RCCL) before anything is pushed
            self.arrived.setdefault(E, set()).add(me)
            while self.session_barrier and len(self.arrived[E]) < self.R:
                yield
            # ... then the rows of state 0 and slot 0 are pushed, collected, published
            yield from self.run_kernel([self.push_then_slot(me, E, 0)])
            yield from self.run_kernel([self.collect(me, E, 0)])
            k, pending = 0, False
            for iters in calls:
                for _ in range(iters):
                    k += 1
                    yield from self.run_kernel([self.comm_wave(me, E, k, pending), self.tiles(me, E, k)])
                    pending = True
                if pending:                                            # end of prcg_iterate: the last iteration's sums are exchanged
                    yield from self.run_kernel([self.send_then_collect(me, E, k)])
                    pending = False

    def push_then_slot(self, me, E, k):
        yield from self.push_rows(me, E, k)
        yield from self.send_slot(me, E, k)

    def send_then_collect(self, me, E, k):
        yield from self.send_slot(me, E, k)
        yield from self.collect(me, E, k)

    def run_kernel(self, actors):
        live = list(actors)
        while live:
            a = self.rng.choice(live)
            try:
                next(a)
                yield
            except StopIteration:
                live.remove(a)

    def run(self, sessions, max_events=2_000_000):
        progs = {r: self.rank_program(r, sessions) for r in range(self.R)}
        events = 0
        while progs:
            ranks = list(progs)
            r = self.rng.choices(ranks, weights=[self.weight[q] for q in ranks])[0]
            try:
                next(progs[r])
            except StopIteration:
                del progs[r]
            events += 1
            if events > max_events:
                raise RuntimeError('model did not finish (deadlock or livelock)')
        return self.checked


def ring(R):
    return {r: sorted({(r - 1) % R, (r + 1) % R} - {r}) for r in range(R)}


def chain(R):
    return {r: [q for q in (r - 1, r + 1) if 0 <= q < R] for r in range(R)}


SESSIONS = [[3, 1, 4], [2, 5], [1]]        # three sessions of several prcg_iterate calls each (odd and even iteration counts)


@pytest.mark.parametrize('R', [2, 3, 4, 5, 8])
def test_peer_exchange_protocol_never_reads_another_iterations_data(R):
    total = 0
    for topo in (chain, ring):
        for seed in range(6):
            m = Model(R, topo(R), seed=seed, slow_rank=seed % R, slow_weight=(0.02, 0.3, 1.0)[seed % 3])
            total += m.run(SESSIONS)
    # every rank checked R payloads per reduction and its neighbours' rows per launch
    assert total > 12 * R * R * sum(sum(s) for s in SESSIONS)


@pytest.mark.parametrize('broken', ['one_parity', 'stores_before_publication', 'counter_before_payload', 'no_session_barrier'])
def test_the_model_detects_a_broken_protocol(broken):
    """Each weakening is caught by some interleaving (the checks are not vacuous)."""
    kw = {'one_parity': dict(parities=1), 'stores_before_publication': dict(wait_before_stores=False),
          'counter_before_payload': dict(payload_first=False), 'no_session_barrier': dict(session_barrier=False)}[broken]
    caught = 0
    for R in (2, 3, 4):
        for seed in range(40):
            m = Model(R, ring(R) if R > 2 else chain(R), seed=seed, slow_rank=seed % R, slow_weight=0.02, **kw)
            try:
                m.run(SESSIONS, max_events=400_000)
            except (Stale, RuntimeError):
                caught += 1
    assert caught > 0, broken
