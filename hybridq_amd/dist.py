"""Multi-GPU state-vector evolution: high-qubit sharding + RCCL all-to-all qubit exchange.

No reference counterpart: ``optimize='evolution'`` "does not support MPI"
(hybridq/circuit/simulation/simulation.py:379-380).  This is the extension named by
BASELINE.json's north_star.

Layout.  G = 2^g ranks (one process per GPU).  Rank r holds the 2^m amplitudes
(m = n - g) whose top g index bits equal r, as split re/im planes -- i.e. index bit
m+i IS bit i of the rank.  A gate whose targets all sit at positions < m runs the
single-GPU kernel on every shard with no communication.  When a gate needs a qubit that
currently sits at a global position the planner

  1. picks g local qubits to evict (Belady: the ones whose next use is farthest away,
     never one the blocked gates need),
  2. if they are not already at the top g local positions, moves them there with ONE
     out-of-place bit-permutation pass (``hq_permute_bits``; 'P' op),
  3. exchanges the top g local bits with the g global bits: ``all_to_all_single`` on the
     plane viewed as [G, 2^(m-g)] -- chunk j of rank r becomes chunk r of rank j
     ('X' op).  Every GPU talks to all 7 peers at once, which is what the xGMI
     point-to-point mesh wants (per-link bound), unlike a ring.

Gates are list-scheduled over their dependency DAG so that everything executable
locally runs before an exchange is paid.  The schedule is a pure function of the gate
list, so every rank computes the same one without communication.

The numerical work goes through ``hybridq_amd.core`` (HIP library) by default.  The
``backend`` argument exists so that the CPU test-suite can drive the SAME planner and
exchange logic over gloo with a host backend; the product never selects one itself.
"""
from collections import deque

import numpy as np

_FLOAT_OF = {np.dtype('complex64'): np.dtype('float32'), np.dtype('complex128'): np.dtype('float64')}


# ------------------------------------------------------------------------------------
# planner (pure Python, deterministic)
# ------------------------------------------------------------------------------------
def plan_schedule(gate_qubits, qubits, g):
    """Schedule gates (given as tuples of qubit labels) on n = len(qubits) qubits sharded
    over 2^g ranks.

    Returns (ops, final_pos): ops is a list of
        ('G', gate_index, local_positions)   positions are LSB-first like simulation.py:633
        ('P', perm)                          local bit permutation, dst bit i <- src bit perm[i]
        ('X',)                               exchange top-g local bits with the rank bits
    and final_pos maps qubit label -> physical position after the last op."""
    n = len(qubits)
    m = n - g
    pos = {q: n - 1 - i for i, q in enumerate(qubits)}  # simulation.py:512
    at = {p: q for q, p in pos.items()}
    queues = {q: deque() for q in qubits}
    for gi, qs in enumerate(gate_qubits):
        for q in qs:
            queues[q].append(gi)
    done = [False] * len(gate_qubits)
    n_done = 0
    ops = []

    def ready(gi):
        return all(queues[q][0] == gi for q in gate_qubits[gi])

    while n_done < len(gate_qubits):
        progress = True
        while progress:
            progress = False
            heads = sorted({queues[q][0] for q in qubits if queues[q]})
            for gi in heads:
                if done[gi] or not ready(gi):
                    continue
                qs = gate_qubits[gi]
                if all(pos[q] < m for q in qs):
                    ops.append(('G', gi, [pos[q] for q in reversed(qs)]))
                    for q in qs:
                        queues[q].popleft()
                    done[gi] = True
                    n_done += 1
                    progress = True
        if n_done == len(gate_qubits):
            break
        if g == 0:
            raise RuntimeError('planner stalled without global qubits')
        # blocked frontier: ready gates that touch a global position
        heads = sorted({queues[q][0] for q in qubits if queues[q]})
        # qubits the exchange must not evict: those of the ready gates, in program order, as long as g evictable
        # local qubits remain (small shards with wide fused gates: protecting EVERY ready gate can leave none)
        needed = set()
        for gi in heads:
            if ready(gi):
                more = needed | set(gate_qubits[gi])
                if sum(1 for q in qubits if pos[q] < m and q not in more) < g:
                    if not needed:
                        needed = more  # even the first gate alone leaves too few: reported below
                    break
                needed = more
        local = [q for q in qubits if pos[q] < m and q not in needed]
        if len(local) < g:
            raise RuntimeError('not enough evictable local qubits for an exchange')
        inf = len(gate_qubits) + 1

        def key(q):
            nxt = queues[q][0] if queues[q] else inf
            in_top = pos[q] >= m - g
            return (-nxt, 0 if in_top else 1, -pos[q])

        evict = sorted(local, key=key)[:g]
        # bring the evictees to the top-g local positions (keep those already there)
        top = list(range(m - g, m))
        stay = [q for q in evict if pos[q] in top]
        free_slots = [p for p in top if at[p] not in stay]
        movers = [q for q in evict if q not in stay]
        if movers:
            perm = list(range(m))
            for slot, q in zip(free_slots, movers):
                a, b = slot, pos[q]
                perm[a], perm[b] = perm[b], perm[a]
                qa, qb = at[a], at[b]
                at[a], at[b] = qb, qa
                pos[qa], pos[qb] = b, a
            ops.append(('P', perm))
        ops.append(('X',))
        for i in range(g):
            a, b = m - g + i, m + i
            qa, qb = at[a], at[b]
            at[a], at[b] = qb, qa
            pos[qa], pos[qb] = b, a
    return ops, dict(pos)


def plan_restore(pos, qubits, g):
    """Ops that bring the placement `pos` (label -> position) back to the canonical one
    (label #x at position n-1-x): at most three exchanges and four permutation passes.
    Returns (ops, canonical_pos)."""
    n = len(qubits)
    m = n - g
    pos = dict(pos)
    at = {p: q for q, p in pos.items()}
    want = {q: n - 1 - i for i, q in enumerate(qubits)}
    ops = []

    def permute_to(target):  # target: {qubit: local position} for a subset
        # every other local qubit stays where it is unless a target claims its position; only the
        # displaced ones move, into the positions the targets vacate (few moved bits per pass)
        taken = set(target.values())
        rest = [at[p] for p in range(m) if at[p] not in target]
        new_local = dict(target)
        displaced = []
        for q in rest:
            if pos[q] in taken:
                displaced.append(q)
            else:
                new_local[q] = pos[q]
                taken.add(pos[q])
        free = [p for p in range(m) if p not in taken]
        new_local.update({q: p for q, p in zip(displaced, free)})
        perm = [0] * m
        for q, p_new in new_local.items():
            perm[p_new] = pos[q]  # dst bit p_new <- src bit pos[q]
        if perm != list(range(m)):
            ops.append(('P', perm))
            for q, p_new in new_local.items():
                pos[q] = p_new
            for q in new_local:
                at[pos[q]] = q

    def exchange():
        ops.append(('X',))
        for i in range(g):
            a, b = m - g + i, m + i
            qa, qb = at[a], at[b]
            at[a], at[b] = qb, qa
            pos[qa], pos[qb] = b, a

    if g:
        W = [q for q in qubits if want[q] >= m]  # must end up global
        if any(pos[q] >= m and pos[q] != want[q] for q in W):
            # some wanted-global qubit is global but in the wrong slot: bring every global in,
            # sending out g local qubits that are not wanted-global
            evict = [at[p] for p in range(m - 1, -1, -1) if at[p] not in W][:g]
            permute_to({q: m - g + i for i, q in enumerate(evict)})
            exchange()
        if any(pos[q] != want[q] for q in W):
            movers = {q: want[q] - g for q in W if pos[q] < m}  # slot m-g+i feeds global position m+i
            # slots whose global partner is already correct must keep a non-W qubit: handled because
            # after the step above either every W qubit is local or the global ones are already right
            if any(pos[q] >= m for q in W):
                # the correct globals would be swapped out by a full exchange: park them first
                evict = [at[p] for p in range(m - 1, -1, -1) if at[p] not in W][:g]
                permute_to({q: m - g + i for i, q in enumerate(evict)})
                exchange()
                movers = {q: want[q] - g for q in W}
            permute_to(movers)
            exchange()
    permute_to({q: want[q] for q in qubits if want[q] < m})
    assert all(pos[q] == want[q] for q in qubits), 'restore planning failed'
    return ops, dict(pos)


def fuse_evictions(schedule):
    """('P', perm) directly followed by ('X',) -> ('XP', perm): the eviction permutation is applied
    by the exchange's own pack pass (hq_exchange_* `perm`) instead of a pass of its own."""
    out = []
    for op in schedule:
        if op[0] == 'X' and out and out[-1][0] == 'P':
            out[-1] = ('XP', out[-1][1])
        else:
            out.append(op)
    return out


#: sub-chunks per peer of an overlapped exchange (2^OVERLAP_SUB_BITS): enough pieces for the arithmetic on piece s to hide
#: behind the transfer of piece s + 1, few enough that a piece (n = 33 on 8 GPUs: 256 MiB per peer) still fills a link
OVERLAP_SUB_BITS = 2
#: a sub-state of fewer qubits than this is not worth separate launches
OVERLAP_MIN_SUB_QUBITS = 12


def overlap_exchanges(schedule, m, g, sub_bits=OVERLAP_SUB_BITS, itemsize=4, budget=1.25):
    """Exchange / compute overlap.  The exchange swaps the top g local bits with the rank bits, so
    a gate that touches NONE of the top g + sub_bits local positions acts inside every 2^(m-g-sub_bits)-amplitude piece
    of every chunk on its own -- on the sender before the exchange or on the receiver after it, the same arithmetic.
    The list scheduler has run everything runnable BEFORE it pays for an exchange, so the candidates are the local gates
    in front of an exchange: scanning back from it, a gate is taken along while it qualifies (in the layout the exchange
    sees, i.e. behind the eviction permutation), nothing between it and the exchange that stays shares a position with it,
    and the work taken does not exceed `budget` x the modelled transfer time (one chunk per link at 153 GB/s against
    6.3 TB/s gate passes: more would only add launches).  They are attached to the exchange,

        ('XO', perm or None, sub_bits, [('G', U, positions in the exchange's layout), ...])

    and the executor moves the shard in 2^sub_bits rounds of one piece per peer (all links busy in every round), applying
    the attached gates to the pieces of round s while round s + 1 is on the wire.  Everything else keeps its place."""
    limit = m - g - sub_bits
    if g == 0 or limit < OVERLAP_MIN_SUB_QUBITS:
        return list(schedule)
    shard_bytes = 2.0 * (1 << m) * itemsize
    transfer_s = shard_bytes / (1 << g) / 153e9
    gate_s = 2.0 * shard_bytes / 6.3e12
    out = []
    for op in schedule:
        if op[0] not in ('X', 'XP'):
            out.append(op)
            continue
        perm = None if op[0] == 'X' else [int(p) for p in op[1]]
        where = {p: p for p in range(m)} if perm is None else {src: dst for dst, src in enumerate(perm)}  # dst bit i <- src bit perm[i]
        taken, held, spent = [], set(), 0.0
        k = len(out)
        while k > 0 and out[k - 1][0] == 'G' and spent + gate_s <= budget * transfer_s + 1e-12:
            o = out[k - 1]
            pos = [where[int(p)] for p in o[2]]
            if max(pos) < limit and not (set(int(p) for p in o[2]) & held):
                taken.append((k - 1, ('G', o[1], np.asarray(pos, dtype=np.uint32))))
                spent += gate_s
            else:
                held |= set(int(p) for p in o[2])  # whatever it shares a position with, further back, must stay in front of it
            k -= 1
        if not taken:
            out.append(op)
            continue
        for idx, _ in taken:  # indices descend: deleting from the back keeps the others valid
            del out[idx]
        out.append(('XO', None if perm is None else np.asarray(perm, dtype=np.uint32), sub_bits, [o for _, o in reversed(taken)]))
    return out


def exchange_in_rounds(be, dist, src, dst, m, g, sub_bits, ops, group, rank, apply_ops):
    """The executor of 'XO' for backends WITHOUT an exchange of their own (the numpy backend of the CPU tests over gloo;
    HipBackend on its torch.distributed fallback transport) on top of torch.distributed point-to-point operations: round s posts, for every peer j, the send of piece s of chunk j and the receive of piece s of the
    peer's chunk for this rank; all rounds are posted at once, then the attached ops run on the pieces of round s as soon
    as that round has landed -- on the compute stream, while the later rounds are still moving.  The result is in `dst`."""
    G, S = 1 << g, 1 << sub_bits
    sub = 1 << (m - g - sub_bits)
    sv = [src[pl].view(G, S, sub) for pl in (0, 1)]
    dv = [dst[pl].view(G, S, sub) for pl in (0, 1)]
    peer = (lambda j: j) if group is None else (lambda j: dist.get_global_rank(group, j))
    rounds = []
    for s_ in range(S):
        p2p = []
        for j in range(G):
            for pl in (0, 1):
                if j == rank:
                    dv[pl][j, s_].copy_(sv[pl][j, s_])  # this rank's own piece
                else:
                    p2p.append(dist.P2POp(dist.isend, sv[pl][j, s_], peer(j), group))
                    p2p.append(dist.P2POp(dist.irecv, dv[pl][j, s_], peer(j), group))
        rounds.append(dist.batch_isend_irecv(p2p) if p2p else [])
    for s_ in range(S):
        for w in rounds[s_]:
            w.wait()  # device backends: the compute stream waits for the transfer, the host does not
        if ops:
            for j in range(G):
                apply_ops((dv[0][j, s_], dv[1][j, s_]), ops, m - g - sub_bits)


# ------------------------------------------------------------------------------------
# backends
# ------------------------------------------------------------------------------------
def exchange_selftest(torch, tdt, device, world, rank, exchange, buffers=None):
    """Run ``exchange(src, dst, perm, m) -> result_in_src`` (collective) once without and once with a local
    permutation on rank-tagged data and check every received chunk: chunk j of the result must be chunk `rank` of
    rank j's (permuted) source.  Raises RuntimeError on a mismatch.  Transport-agnostic: the CPU test-suite runs it
    over gloo with the host backend, HipBackend runs it on a fresh RCCL communicator.

    ``buffers`` = the rank's two REAL shard buffers ((2, 2^m) plane pairs, contents are overwritten): the test then
    moves the very memory the run will move -- same allocator, same size class (a self-test on
    torch.empty memory proves nothing about library-mapped planes).  Without them two small torch buffers are used."""
    g = int(np.log2(world))
    if buffers is not None:
        src, dst = buffers
        m = int(src.shape[1]).bit_length() - 1
    else:
        m = max(2 * g + 2, 12)
        src = torch.empty((2, 1 << m), dtype=tdt, device=device)
        dst = torch.zeros_like(src)
    chunk = (1 << m) >> g
    step = min(1 << m, 1 << 24)  # fill and check in slices: no index array of the shard's size

    def tag(lo, hi, r):  # exactly representable, rank-tagged
        return (torch.arange(lo, hi, device=device) % 4096).to(tdt) + 4096.0 * r

    for perm in (None, np.concatenate([[1, 0], np.arange(2, m)]).astype(np.uint32)):
        for lo in range(0, 1 << m, step):
            src[0, lo:lo + step] = tag(lo, lo + step, rank)
            src[1, lo:lo + step] = -src[0, lo:lo + step]
        got = src if exchange(src, dst, perm, m) else dst
        cstep = min(chunk, step)
        for j in range(world):
            for lo in range(0, chunk, cstep):
                x = torch.arange(rank * chunk + lo, rank * chunk + lo + cstep, device=device)
                if perm is not None:  # bits 0 and 1 swapped (its own inverse: independent of the direction convention)
                    x = (x & ~3) | ((x & 1) << 1) | ((x >> 1) & 1)
                want = (x % 4096).to(tdt) + 4096.0 * j
                sl = slice(j * chunk + lo, j * chunk + lo + cstep)
                if not torch.equal(got[0, sl], want) or not torch.equal(got[1, sl], -want):
                    raise RuntimeError(f'exchange self-test: wrong data in chunk {j}' + (' (with permutation)' if perm is not None else ''))


def _call_with_timeout(fn, seconds, what):
    """Run fn() in a helper thread and give up after `seconds`: a collective whose peers never arrive (one rank failed
    before entering it) must not hang this rank.  The abandoned thread is a daemon."""
    import threading
    box = {}

    def run():
        try:
            box['value'] = fn()
        except BaseException as e:  # noqa: BLE001
            box['error'] = e

    t = threading.Thread(target=run, daemon=True, name=f'hq-{what}')
    t.start()
    t.join(seconds)
    if t.is_alive():
        raise TimeoutError(f'{what} did not finish within {seconds:.0f} s')
    if 'error' in box:
        raise box['error']
    return box.get('value')


class HipBackend:
    """Shard planes in HBM (torch tensors), kernels and the qubit exchange from libhq_hip.so.

    Exchange transports (``transport=`` or env HQ_SHARD_TRANSPORT; default 'auto'):
      'rccl'   hq_exchange_* over the library's own RCCL communicator (ncclSend/ncclRecv to the G-1
               peers in one group, both planes, self chunk by an own kernel, eviction permutation
               folded into the pack pass) -- the default whenever the process group runs on nccl;
      'p2p'    hq_exchange_* with peer-to-peer stores into the other ranks' planes mapped through
               HIP IPC: ONE pass, bracketed by host barriers -- the default for ranks that share one
               GPU (gloo process group: RCCL refuses two ranks per device), also usable over xGMI;
      'torch'  permute_bits + torch.distributed.all_to_all_single per plane (the round-1 path; kept
               as a cross-check and as the fallback when the library's transports cannot start)."""

    _ipc_mappings = {}  # (peer rank, exported handle) -> base address of the mapping in this process

    def __init__(self, float_type, device=None, transport=None, placement='tuned'):
        import os
        import torch
        import torch.distributed as dist
        from . import core
        if not torch.cuda.is_available():
            raise RuntimeError('HipBackend needs a HIP device')
        self.torch, self.dist, self.core = torch, dist, core
        self.float_type = np.dtype(float_type)
        self.tdt = {np.dtype('float32'): torch.float32, np.dtype('float64'): torch.float64}[self.float_type]
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        self.transport = transport or os.environ.get('HQ_SHARD_TRANSPORT', 'auto')
        self.transport_note = ''
        self.placement = placement  # 'tuned': draw-and-probe VMM placement of the shard buffers (seconds); 'plain': torch's allocator
        core.use_torch_stream()

    def _wanted_transport(self):
        dist = self.dist
        if self.transport != 'auto' or not dist.is_initialized() or dist.get_world_size() == 1:
            return self.transport
        return 'rccl' if dist.get_backend() == 'nccl' else 'p2p'

    def empty_planes(self, m):
        from .simulation import alloc_planes
        # re/im rows offset by PLANE_PAD_BYTES.  Planes that other ranks map through HIP IPC (p2p transport)
        # must come from hipMalloc: hipIpcGetMemHandle does not export the library's VMM mappings.
        return alloc_planes(m, self.tdt, self.device, vmm=self.placement == 'tuned' and self._wanted_transport() != 'p2p')

    # -- exchange transport ---------------------------------------------------------------
    #: seconds a rank waits inside the collective parts of the RCCL start-up (communicator creation, self-test) before
    #: it gives up and votes for the fallback (env HQ_SHARD_TIMEOUT)
    SETUP_TIMEOUT = 120.0

    def _agree(self, group, error):
        """Collective over the torch process group: every rank reports its own failure ('' = fine); returns the first
        failure anywhere, so that all ranks take the same branch."""
        reports = [None] * self.world
        self.dist.all_gather_object(reports, '' if error is None else repr(error), group=group)
        bad = [(r, e) for r, e in enumerate(reports) if e]
        return None if not bad else f'rank {bad[0][0]}: {bad[0][1]}'

    def setup_exchange(self, group, buffers):
        """Collective.  `buffers`: the rank's two shard buffers (each a (2, 2^m) plane pair).

        Failure-safe start-up of the RCCL transport: (1) everything a rank can check ALONE
        (librccl loads, the unique id exists) is checked and agreed on over the torch process group BEFORE anybody
        enters ncclCommInitRank; (2) the communicator is created and (3) a real exchange -- with and without a
        permutation, every chunk checked -- is run on the REAL shard buffers, each in a helper thread with a timeout, so
        that a rank whose peers never arrive raises instead of waiting forever; after each phase the ranks agree again.
        Any failure anywhere sends ALL ranks to the torch.distributed fallback together."""
        import os
        dist = self.dist
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.group, self.world, self.rank = group, world, rank
        if world == 1:
            self.transport = 'none'
            return
        want = self.transport
        if want == 'auto':
            want = 'rccl' if dist.get_backend(group) == 'nccl' else 'p2p'
        if want not in ('rccl', 'p2p', 'torch'):
            raise ValueError(f'unknown exchange transport {want!r}')
        timeout = float(os.environ.get('HQ_SHARD_TIMEOUT', self.SETUP_TIMEOUT))
        failure = None
        if want == 'rccl':
            # phase 1 (local): the library binds librccl, rank 0 draws the unique id
            err, uid = None, [None]
            try:
                self.core.shard_load_rccl()
                if rank == 0:
                    uid[0] = self.core.shard_unique_id()
            except Exception as e:  # noqa: BLE001
                err = e
            failure = self._agree(group, err)
            if failure is None:
                dist.broadcast_object_list(uid, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
                # phase 2 (collective, may block on a missing peer): the communicator
                err = None
                try:
                    dev = self.torch.cuda.current_device()

                    def init():
                        self.torch.cuda.set_device(dev)  # the HIP device is per thread
                        self.core.shard_init_rccl(world, rank, uid[0])
                    _call_with_timeout(init, timeout, 'ncclCommInitRank')
                except BaseException as e:  # noqa: BLE001
                    err = e
                failure = self._agree(group, err)
            if failure is None:
                # phase 3: a real exchange on the real shard buffers
                err = None
                try:
                    self._rccl_selftest(world, rank, buffers, timeout)
                except BaseException as e:  # noqa: BLE001
                    err = e
                failure = self._agree(group, err)
            if failure is not None:
                try:
                    self.core.shard_free()  # also cancels a communicator creation that is still pending
                except Exception:  # noqa: BLE001
                    pass
        elif want == 'p2p':
            # the same discipline: local step (export the planes), agree, collective step (gather the handles), local
            # step (map the peers' planes), agree
            err, mine = None, None
            try:
                mine = self._p2p_export(buffers)
            except Exception as e:  # noqa: BLE001
                err = e
            failure = self._agree(group, err)
            if failure is None:
                everyone = [None] * world
                dist.all_gather_object(everyone, [(h, o) for h, o, _ in mine], group=group)
                err = None
                try:
                    self._p2p_map(buffers, everyone)
                except Exception as e:  # noqa: BLE001
                    err = e
                failure = self._agree(group, err)
        if failure is not None:
            self.transport_note = f'{want} transport unavailable ({failure}); using torch.distributed collectives'
            want = 'torch'
        self.transport = want

    def _rccl_selftest(self, world, rank, buffers=None, timeout=None):
        """One real exchange through hq_exchange_* right after the communicator exists (exchange_selftest), on the
        shard buffers themselves: a transport that errors, delivers the wrong chunks or never completes is detected
        HERE, where every rank can still agree on the torch fallback."""
        torch = self.torch

        def ex(src, dst, perm, m):
            in_src = self.core.exchange(src[0], src[1], dst[0], dst[1], perm, m)
            done = torch.cuda.Event()
            done.record()  # the library runs on torch's current stream
            import time
            t0 = time.monotonic()
            while not done.query():  # poll: hipStreamSynchronize on a stuck transfer would never return
                if timeout is not None and time.monotonic() - t0 > timeout:
                    raise TimeoutError(f'the exchange self-test did not complete within {timeout:.0f} s')
                time.sleep(0.002)
            return in_src
        use = None
        if buffers is not None and len(buffers) == 2 and buffers[0] is not None and buffers[1] is not None:
            use = (buffers[0], buffers[1])
        exchange_selftest(torch, self.tdt, self.device, world, rank, ex, buffers=use)

    def _p2p_export(self, buffers):
        planes = [b[p] for b in buffers for p in (0, 1)]
        return [self.core.ipc_export(t) + (t.data_ptr(),) for t in planes]  # (handle, offset, local address)

    def _p2p_map(self, buffers, everyone):
        core = self.core
        planes = [b[p] for b in buffers for p in (0, 1)]
        core.shard_init_p2p(self.world, self.rank)
        opened = HipBackend._ipc_mappings  # process-wide: an exported allocation is mapped once
        for i, t in enumerate(planes):
            addrs = []
            for r in range(self.world):
                if r == self.rank:
                    addrs.append(t.data_ptr())
                    continue
                h, o = everyone[r][i]
                if (r, h) not in opened:  # one mapping per exported allocation
                    opened[(r, h)] = core.ipc_open(h, 0)
                addrs.append(opened[(r, h)] + o)
            core.shard_p2p_register(t, addrs)

    def exchange(self, src, dst, perm, m, group):
        """Exchange the top-g local bits with the rank bits, applying the local permutation `perm`
        (or None) on the way.  Returns True if the result is in `src` (else in `dst`)."""
        core = self.core
        if self.transport in ('rccl', 'none'):
            return core.exchange(src[0], src[1], dst[0], dst[1], perm, m)
        if self.transport == 'p2p':
            # nobody may write into a peer's dst planes before that peer is done with them, and nobody
            # may read its own dst planes before every peer's stores have landed
            core.sync()
            self.dist.barrier(group=group)
            where = core.exchange(src[0], src[1], dst[0], dst[1], perm, m)
            core.sync()
            self.dist.barrier(group=group)
            return where
        # 'torch': two passes (three with a permutation)
        if perm is not None:
            self.permute(src, dst, perm, m)
            src, dst = dst, src
        self.all_to_all(dst, src, group)
        return perm is not None

    def exchange_rounds(self, src, dst, perm, m, sub_bits, group):
        """The exchange in 2^sub_bits rounds (hq_exchange_rounds_*): returns (result in `src`?, number of rounds) without
        having waited for the transfers; exchange_round_wait(r) makes the gate stream wait for round r.  The RCCL
        transport runs all rounds on the library's communication stream; the peer-to-peer transport (one pass of
        stores into the peers' planes between two host barriers) and the torch fallback complete before they return
        and report ONE round."""
        core = self.core
        if self.transport in ('rccl', 'none'):
            return core.exchange_rounds(src[0], src[1], dst[0], dst[1], perm, m, sub_bits)
        return self.exchange(src, dst, perm, m, group), 1

    def exchange_round_wait(self, r):
        if self.transport in ('rccl', 'none'):
            self.core.exchange_round_wait(r)

    def exchange_in_real_rounds(self):
        """True where exchange_rounds really moves the shard in several rounds the gate stream can work behind (RCCL)."""
        return self.transport == 'rccl'

    def fill_zero(self, planes):
        planes.zero_()

    def fill_basis(self, planes, local_index):
        self.core.init_state(planes[0], planes[1], 'basis', local_index)

    def fill_const(self, planes, value):
        planes[0].fill_(value)
        planes[1].zero_()

    def fill_product(self, planes, chars_by_position, hi_bits):
        self.core.init_product_state(planes[0], planes[1], chars_by_position, hi_bits)

    def apply(self, planes, U, pos, m):
        self.core.apply_U(planes[0], planes[1], U, pos, m)

    def apply_blocked(self, planes, tile_pos, gates, m):
        """gates: [(U, local positions)] all inside tile_pos: one LDS-tile pass (blocking.py)."""
        self.core.apply_blocked(planes[0], planes[1], tile_pos, gates, m)

    def permute(self, src, dst, perm, m):
        self.core.permute_bits(src[0], dst[0], perm, m)
        self.core.permute_bits(src[1], dst[1], perm, m)

    def all_to_all(self, dst, src, group):
        self.dist.all_to_all_single(dst[0], src[0], group=group)
        self.dist.all_to_all_single(dst[1], src[1], group=group)

    def interleave(self, planes, out):
        self.core.to_complex(planes[0], planes[1], out)

    def sync(self):
        self.torch.cuda.synchronize()

    def to_numpy(self, planes):
        return planes.cpu().numpy()

    def norm2(self, planes):
        return self.core.norm2(planes[0], planes[1])


# ------------------------------------------------------------------------------------
# runtime
# ------------------------------------------------------------------------------------
class ShardedEvolution:
    """n-qubit state sharded by the top g = log2(world) index bits over the process group."""

    def __init__(self, n, complex_type='complex64', initial_state=None, qubits=None, backend=None,
                 group=None, overlap=None):
        """``overlap``: overlap every qubit exchange with the local gates that do not touch the moving bits
        (:func:`overlap_exchanges`; default: the environment variable HQ_SHARD_OVERLAP, off -- the round-trip on xGMI has
        not been measured yet).  Overlapped exchanges run behind the C ABI (hq_exchange_rounds_*) on
        HipBackend, as torch.distributed send / recv rounds on minimal backends.  ``True`` takes effect only where the
        exchange really proceeds in rounds (HipBackend on the RCCL transport, minimal backends): on a transport that
        completes in one blocking round (peer-to-peer stores, the torch fallback) attaching gates to an exchange would only
        split them into per-piece launches, so the plan keeps plain exchanges there; ``'force'`` attaches them regardless
        (tests of the one-round executor path)."""
        import os
        import torch.distributed as dist
        self.overlap = (os.environ.get('HQ_SHARD_OVERLAP', '0') == '1') if overlap is None else (overlap if overlap == 'force' else bool(overlap))
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.g = int(self.world).bit_length() - 1
        if 1 << self.g != self.world:
            raise ValueError('the number of ranks must be a power of two')
        self.n = n
        self.m = n - self.g
        if self.m < 2 * self.g:
            raise ValueError('need at least 2*log2(world) local qubits')
        self.qubits = list(range(n)) if qubits is None else list(qubits)
        self.complex_type = np.dtype(complex_type)
        self.float_type = _FLOAT_OF[self.complex_type]
        self.backend = HipBackend(self.float_type) if backend is None else backend
        # second buffer (receive side of the exchange / destination of the permutation pass)
        # is allocated on first use when there is a single rank
        self.bufs = [self.backend.empty_planes(self.m), self.backend.empty_planes(self.m) if self.g else None]
        if hasattr(self.backend, 'setup_exchange'):
            self.backend.setup_exchange(group, [b for b in self.bufs if b is not None])
        self.cur = 0
        self.pos = {q: n - 1 - i for i, q in enumerate(self.qubits)}
        self._gates = None
        self.set_state('0' * n if initial_state is None else initial_state)

    # -- state ------------------------------------------------------------------
    @property
    def planes(self):
        return self.bufs[self.cur]

    def set_state(self, initial_state):
        """'01+-' strings (hybridq/circuit/simulation/utils.py:99-153), one character per qubit in
        canonical order; every rank writes its own shard on the device."""
        s = initial_state
        if len(s) == 1:
            s = s * self.n
        if len(s) != self.n:
            raise ValueError("'initial_state' has the wrong number of qubits.")
        if any(c not in '01+-' for c in s):
            raise ValueError("'initial_state' may contain only '0', '1', '+', '-'.")
        self.pos = {q: self.n - 1 - i for i, q in enumerate(self.qubits)}
        if all(c in '01' for c in s):
            b = int(s, 2)
            if (b >> self.m) == self.rank:
                self.backend.fill_basis(self.planes, b & ((1 << self.m) - 1))
            else:
                self.backend.fill_zero(self.planes)
        elif all(c == '+' for c in s):
            self.backend.fill_const(self.planes, 2.0**(-0.5 * self.n))
        else:
            self.backend.fill_product(self.planes, {self.n - 1 - i: c for i, c in enumerate(s)}, self.rank << self.m)

    # -- planning ----------------------------------------------------------------
    def plan(self, gates, compress=0, blocked=False):
        """Schedule `gates` ([(U, qubits), ...]) from the CURRENT qubit placement.  Returns
        the op list for run(); the matrices are cast once here.  ``compress`` > 0 first fuses
        the circuit into <= compress-qubit gates (hybridq_amd.fusion, the reference's default
        is 4): fewer local passes, same exchanges.  ``blocked`` (True or a dict of
        ``blocking.plan_blocked`` options) re-schedules every run of local gates between two
        exchanges as cache-blocked passes (many gates per HBM pass, 'B' ops)."""
        if compress:
            from .fusion import fuse
            gates = fuse(gates, compress, complex_type=self.complex_type)
        gq = [tuple(qs) for _, qs in gates]
        order = sorted(self.qubits, key=lambda q: -self.pos[q])  # label at position n-1, n-2, ...
        ops, final_pos = plan_schedule(gq, order, self.g)
        mats = [np.ascontiguousarray(U, dtype=self.complex_type) for U, _ in gates]
        sched = []
        for op in ops:
            if op[0] == 'G':
                sched.append(('G', mats[op[1]], np.asarray(op[2], dtype=np.uint32)))
            elif op[0] == 'P':
                sched.append(('P', np.asarray(op[1], dtype=np.uint32)))
            else:
                sched.append(op)
        self._planned_final_pos = final_pos
        sched = fuse_evictions(sched)
        if self.overlap == 'force' or (self.overlap and getattr(self.backend, 'exchange_in_real_rounds', lambda: True)()):
            sched = overlap_exchanges(sched, self.m, self.g, itemsize=self.float_type.itemsize)
        if blocked and self.m >= 14:
            from .blocking import plan_blocked
            opts = dict(blocked) if isinstance(blocked, dict) else {}
            c64 = self.complex_type == np.dtype('complex64')
            opts.setdefault('tile_bits', 13 if c64 else 12)
            opts.setdefault('low_bits', 5 if c64 else 4)
            opts.setdefault('complex_type', self.complex_type)
            out, run = [], []

            def replan(run, n_sub, out):
                ident = {p: p for p in range(n_sub)}  # "qubits" of the sub-plan are local positions
                for op in plan_blocked([(U, tuple(int(p) for p in reversed(pos))) for U, pos in run], ident, n_sub, **opts):
                    if op[0] == 'B':
                        out.append(('B', op[1], [(np.ascontiguousarray(U, dtype=self.complex_type),
                                                  np.asarray(p, dtype=np.uint32)) for U, p in op[2]]))
                    else:
                        out.append(('G', np.ascontiguousarray(op[1], dtype=self.complex_type),
                                    np.asarray(op[2], dtype=np.uint32)))

            def flush():
                if run:
                    replan(run, self.m, out)
                    run.clear()

            for op in sched:
                if op[0] == 'G':
                    run.append((op[1], op[2]))
                else:
                    flush()
                    if op[0] == 'XO' and self.m - self.g - op[2] >= 14:  # the attached gates as passes over the pieces
                        inner = []
                        replan([(o[1], o[2]) for o in op[3]], self.m - self.g - op[2], inner)
                        op = ('XO', op[1], op[2], inner)
                    out.append(op)
            flush()
            sched = out
        return sched

    def run(self, schedule, update_map=True, timer=None):
        """Execute a schedule.  `timer` (optional, used by bench.py): an object with
        ``start(op) -> token`` / ``stop(token)`` called around every op on the issuing stream."""
        be = self.backend
        for op in schedule:
            if op[0] in ('P', 'X', 'XP', 'XO') and self.bufs[1 - self.cur] is None:
                self.bufs[1 - self.cur] = be.empty_planes(self.m)
            tok = timer.start(op) if timer is not None else None
            if op[0] == 'G':
                be.apply(self.bufs[self.cur], op[1], op[2], self.m)
            elif op[0] == 'B':
                be.apply_blocked(self.bufs[self.cur], op[1], op[2], self.m)
            elif op[0] == 'P':
                be.permute(self.bufs[self.cur], self.bufs[1 - self.cur], op[1], self.m)
                self.cur = 1 - self.cur
            elif op[0] == 'XO':  # exchange in rounds, the attached local ops applied to the pieces as they land
                if hasattr(be, 'exchange_rounds'):
                    # behind the C ABI (hq_exchange_rounds_*): eviction permutation folded into the pack pass, the library's
                    # own transports, one completion event per round that the gate stream waits for
                    where, n_rounds = be.exchange_rounds(self.bufs[self.cur], self.bufs[1 - self.cur], op[1], self.m, op[2], self.group)
                    res = self.bufs[self.cur] if where else self.bufs[1 - self.cur]
                    G, S, n_sub = 1 << self.g, 1 << op[2], self.m - self.g - op[2]
                    pieces = [res[pl].view(G, S, 1 << n_sub) for pl in (0, 1)]
                    for r in range(n_rounds):
                        be.exchange_round_wait(r)
                        for s_ in range(r * S // n_rounds, (r + 1) * S // n_rounds):  # (one round: every piece)
                            for j in range(G):
                                self._apply_local_ops((pieces[0][j, s_], pieces[1][j, s_]), op[3], n_sub)
                    if not where:
                        self.cur = 1 - self.cur
                else:
                    if op[1] is not None:  # the eviction permutation as a pass of its own (the rounds move contiguous pieces)
                        be.permute(self.bufs[self.cur], self.bufs[1 - self.cur], op[1], self.m)
                        self.cur = 1 - self.cur
                    exchange_in_rounds(be, self.dist, self.bufs[self.cur], self.bufs[1 - self.cur], self.m, self.g, op[2], op[3],
                                       self.group, self.rank, self._apply_local_ops)
                    self.cur = 1 - self.cur
            else:  # 'X' / 'XP': the exchange, with the eviction permutation folded in for 'XP'
                perm = op[1] if op[0] == 'XP' else None
                if hasattr(be, 'exchange'):
                    in_src = be.exchange(self.bufs[self.cur], self.bufs[1 - self.cur], perm, self.m, self.group)
                else:  # minimal backends (tests): the two primitive passes
                    if perm is not None:
                        be.permute(self.bufs[self.cur], self.bufs[1 - self.cur], perm, self.m)
                        self.cur = 1 - self.cur
                    be.all_to_all(self.bufs[1 - self.cur], self.bufs[self.cur], self.group)
                    in_src = False
                if not in_src:
                    self.cur = 1 - self.cur
            if timer is not None:
                timer.stop(tok)
        if update_map:
            self.pos = dict(self._planned_final_pos)

    def _apply_local_ops(self, planes, ops, n_sub):
        be = self.backend
        for o in ops:
            if o[0] == 'G':
                be.apply(planes, o[1], o[2], n_sub)
            else:
                be.apply_blocked(planes, o[1], o[2], n_sub)

    def simulate(self, gates, compress=0, blocked=False):
        self.run(self.plan(gates, compress=compress, blocked=blocked))
        return self

    def restore_order(self):
        """Bring every qubit back to its canonical position (label #x at index bit n-1-x), the
        counterpart of the reference's final un-permute (simulation.py:655-663)."""
        order = list(self.qubits)
        ops, final = plan_restore(self.pos, order, self.g)
        sched = fuse_evictions([('P', np.asarray(op[1], dtype=np.uint32)) if op[0] == 'P' else op for op in ops])
        self._planned_final_pos = final
        self.run(sched)
        return self

    # -- results -----------------------------------------------------------------
    def to_complex(self):
        """This rank's shard as ONE interleaved complex tensor on the device (the sharded
        counterpart of to_complex64/128, python_U.cpp:114-123): after restore_order() rank r's
        tensor is the slice [r * 2^m, (r+1) * 2^m) of the canonical state."""
        be = self.backend
        torch = be.torch
        cdt = {np.dtype('complex64'): torch.complex64, np.dtype('complex128'): torch.complex128}[self.complex_type]
        out = torch.empty(1 << self.m, dtype=cdt, device=be.device)
        be.interleave(self.planes, out)
        return out

    def norm2(self):
        """Global squared norm (all-reduce of the local ones)."""
        import torch
        v = torch.tensor([self.backend.norm2(self.planes)], dtype=torch.float64)
        if self.world > 1:
            if self.dist.get_backend(self.group) == 'nccl':
                v = v.cuda()
            self.dist.all_reduce(v, group=self.group)
        return float(v.item())

    def state_numpy(self):
        """Full state in CANONICAL qubit order on every rank (tests / small n only)."""
        import torch
        self.backend.sync()
        local = np.ascontiguousarray(self.backend.to_numpy(self.planes))
        if self.world > 1:
            t = torch.from_numpy(local)
            nccl = self.dist.get_backend(self.group) == 'nccl'
            if nccl:
                t = t.cuda()
            parts = [torch.empty_like(t) for _ in range(self.world)]
            self.dist.all_gather(parts, t, group=self.group)
            full = np.stack([p.cpu().numpy() for p in parts], axis=1)  # [plane, rank, local]
        else:
            full = local[:, None, :]
        psi = (full[0] + 1j * full[1]).reshape(-1)  # physical order: index bit p <-> position p
        # physical position p holds qubit at[p]; canonical wants label #x at position n-1-x
        n = self.n
        at = {p: q for q, p in self.pos.items()}
        axes_now = [at[n - 1 - a] for a in range(n)]  # qubit label on numpy axis a
        want = self.qubits
        perm = [axes_now.index(q) for q in want]
        return np.ascontiguousarray(np.transpose(psi.reshape((2,) * n), perm)).reshape(-1)
