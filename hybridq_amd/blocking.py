"""Cache-blocked scheduling: many gates per HBM pass (driver side of ``hq_apply_blocked_*``).

Every k <= 4 gate kernel already runs at the memory system's ceiling (one pass over the
state, ~3 ms at n = 30), so the remaining lever is FEWER passes.  A blocked pass stages
tiles of 2^tile_bits amplitudes in LDS -- the index bits of a tile are the low
``low_bits`` positions (coalescing) plus ``tile_bits - low_bits`` freely chosen ones --
and applies every gate of a list whose targets lie inside those bits before the tile goes
back to HBM.  This module decides which bits and which gates:

  * gates are list-scheduled over their dependency DAG (per-qubit program order), so a pass
    may run arbitrarily far ahead on the qubits it holds;
  * the free tile positions are chosen greedily: absorb whatever is ready inside the tile,
    spend spare capacity on the ready gate that needs the fewest new positions, repeat;
    several visiting orders are tried per pass and the one absorbing most gates wins;
  * the gates of a pass are fused (``fusion.fuse``) up to ``inner_max`` qubits, because an
    inner gate costs matrix-core time only (k <= 3: ~0.76 ms, k = 4: ~1.2 ms at n = 30); the default
    ``'auto'`` fuses to 3 qubits and widens to 4 where the second round saves more gates than it costs;
  * passes that would hold fewer than ``min_gates`` gates are emitted as plain gates.

There is no reference counterpart (the reference applies one fused gate per pass,
hybridq/circuit/simulation/simulation.py:522-646); results are identical up to rounding.
"""
import numpy as np

from .fusion import fuse


#: relative cost of an inner gate by width (measured, n = 30 complex64: 0.63 ms for k <= 3, 1.20 ms for k = 4)
INNER_COST = {1: 1.0, 2: 1.0, 3: 1.0, 4: 1.9}


#: LDS next to a 64 KiB tile with two workgroups per CU (hq_apply.hip: a_budget), and what a gate needs of it:
#: its MFMA A-operand table (k = 2, 3: 256 elements, k = 4: 1024; k = 1 runs on the VALU from scalar registers)
#: plus 136 32-bit words of slot-address tables
LDS_TABLE_BUDGET = 15 * 1024


def _big_tile(tile_bits, complex_type):
    """128 KiB tiles (2^14 complex64 / 2^13 complex128 amplitudes): one 1024-thread workgroup per CU (HQ_BLOCKED_BIG=1),
    800-byte address tables per gate, 31 KiB of LDS beside the tile."""
    return tile_bits is not None and (2 << tile_bits) * (4 if np.dtype(complex_type) == np.dtype('complex64') else 8) == 128 * 1024


def lds_bytes(gate_list, complex_type='complex64', tile_bits=None):
    elem = 4 if np.dtype(complex_type) == np.dtype('complex64') else 8
    tab = 800 if _big_tile(tile_bits, complex_type) else 544
    return sum({1: 0, 2: 256, 3: 256, 4: 1024}[len(qs)] * elem + tab for _, qs in gate_list)


def lds_budget(tile_bits, complex_type='complex64'):
    return 31 * 1024 if _big_tile(tile_bits, complex_type) else LDS_TABLE_BUDGET


def _dry_layers(qsets, kmax):
    """The grouping fusion.fuse would produce for gates arriving in this order, on qubit sets alone (sliding left
    through disjoint layers only: the matrix-commutation test needs matrices).  An entry ``('F', qubits or None)`` is a gate
    without a matrix (fusion.Opaque): a layer of its own that nothing merges into, that gates on other qubits slide
    across and that stops the rest (one without qubits stops everything); returned as such."""
    layers = []
    for q in qsets:
        if isinstance(q, tuple):
            layers.append(q)
            continue
        merge_to = len(layers)
        for i in range(len(layers) - 1, -1, -1):
            cq = layers[i]
            if isinstance(cq, tuple):
                if cq[1] is None or (q & cq[1]):
                    break
                continue
            if len(q | cq) <= max(kmax, len(cq), len(q)):
                merge_to = i
            if not (q & cq):
                continue
            break
        if merge_to < len(layers):
            layers[merge_to] = layers[merge_to] | q
        else:
            layers.append(q)
    return layers


def _best_fusion_order(chosen, gq, n_orders, rnd):
    """Among `n_orders` topological orders of the pass's gates (the given one first), the one whose dry fusion
    (to 3 qubits, optionally widened to 4) is cheapest by INNER_COST."""
    if n_orders <= 1 or len(chosen) < 3:
        return chosen
    qsets = {g: frozenset(gq[g]) for g in chosen}
    qlist = {}
    for g in chosen:
        for q in gq[g]:
            qlist.setdefault(q, []).append(g)

    def random_order():
        ptr = {q: 0 for q in qlist}
        heads = {}
        for q, l in qlist.items():
            heads[l[0]] = heads.get(l[0], 0) + 1
        ready = sorted(g for g, c in heads.items() if c == len(gq[g]))
        out = []
        while ready:
            g = ready.pop(rnd.randrange(len(ready)))
            out.append(g)
            for q in gq[g]:
                ptr[q] += 1
                if ptr[q] < len(qlist[q]):
                    h = qlist[q][ptr[q]]
                    heads[h] = heads.get(h, 0) + 1
                    if heads[h] == len(gq[h]):
                        ready.append(h)
        return out

    def score(order):
        l3 = _dry_layers([qsets[g] for g in order], 3)
        l4 = _dry_layers(l3, 4)
        return min(sum(INNER_COST[len(x)] for x in l3), sum(INNER_COST[len(x)] for x in l4))

    best, best_cost = chosen, score(chosen)
    for _ in range(n_orders - 1):
        cand = random_order()
        c = score(cand)
        if c < best_cost - 1e-9:
            best, best_cost = cand, c
    return best


def plan_blocked(gates, pos_of, n, tile_bits=13, low_bits=5, inner_max='auto', min_gates=3, tries=32, seed=0,
                 complex_type='complex64', fusion_orders=16, native=True, seeds=1):
    """Schedule `gates` ([(U, qubits), ...]) given the placement ``pos_of[qubit] = position``.

    ``seeds`` > 1: plan with that many consecutive seeds and keep the plan with the smallest modelled device time
    (simulation.estimate_ms).  The growth of a pass is a randomised greedy search: over seeds the modelled time of the n = 30
    benchmark plan spreads by +-3 % (140.7 ... 149.3 ms, 28-30 passes), each seed costs ~9 ms of host time with the native
    planner -- worth it for a circuit that is planned once and run often (the plan cache), or whose loop is long.

    ``native=True`` (default): the planner behind the C ABI (``hq_plan_blocked``, csrc/hq_plan.hip) -- the algorithm
    below in C++, 20-40x faster (the caller of simulate() waits for the plan); ``native=False`` runs this Python
    statement of it.  Both are deterministic for a seed; they draw different random numbers, so their plans may differ
    by a pass.

    Returns a list of ops:
        ('B', tile_pos uint32[tile_bits] ascending, [(U, positions LSB-first), ...])
        ('G', U, positions LSB-first)
    ``tries`` > 1: each pass is grown from several (seeded, deterministic) visiting orders of
    the ready gates and the one that absorbs the most gates is kept (n=30 depth-40 circuit:
    34 -> 28 passes for 32 tries, ~0.1 s of host time for the growth + ~0.15 s for the inner fusion)."""
    import random
    if seeds > 1:
        from .simulation import estimate_ms
        best = None
        for s_ in range(seed, seed + int(seeds)):
            ops = plan_blocked(gates, pos_of, n, tile_bits, low_bits, inner_max, min_gates, tries, s_, complex_type, fusion_orders, native)
            ms = estimate_ms(ops, n, np.dtype(complex_type))
            if best is None or ms < best[0] - 1e-12:
                best = (ms, ops)
        return best[1]
    tile_bits = min(tile_bits, n)
    low_bits = min(low_bits, tile_bits)
    if native:
        return _plan_blocked_native(gates, pos_of, n, tile_bits, low_bits, inner_max, min_gates, tries, seed, complex_type,
                                    fusion_orders)
    gq = [tuple(qs) for _, qs in gates]
    gp = [frozenset(pos_of[q] for q in qs) for qs in gq]
    gk = [len(qs) for qs in gq]
    qubits = sorted(pos_of, key=lambda q: pos_of[q])
    qlist = {q: [] for q in qubits}  # per-qubit program order
    for gi, qs in enumerate(gq):
        for q in qs:
            qlist[q].append(gi)
    ptr = {q: 0 for q in qubits}  # next unscheduled gate of every qubit
    done = 0
    ops = []
    low = frozenset(range(low_bits))
    rnd = random.Random(seed)

    def grow(ptr, order_key):
        """One candidate pass: returns (chosen gate indices in execution order, tile set, pointers after).
        List scheduling with head counters: a gate is ready when it heads the queue of every qubit it acts on."""
        ptr = dict(ptr)
        S = set(low)
        chosen = []
        at_head = {}
        for q in qubits:
            if ptr[q] < len(qlist[q]):
                g = qlist[q][ptr[q]]
                at_head[g] = at_head.get(g, 0) + 1
        ready = {g for g, c in at_head.items() if c == gk[g] and gk[g] <= 4}

        def take(g):
            chosen.append(g)
            ready.discard(g)
            for q in gq[g]:
                ptr[q] += 1
                if ptr[q] < len(qlist[q]):
                    h = qlist[q][ptr[q]]
                    c = at_head[h] = at_head.get(h, 0) + 1
                    if c == gk[h] and gk[h] <= 4:
                        ready.add(h)

        while ready:
            fit = [g for g in ready if gp[g] <= S]  # everything ready that already fits
            if fit:
                for g in sorted(fit, key=order_key):
                    take(g)
                continue
            best, best_new = None, None  # spend spare capacity on the cheapest ready gate
            for g in sorted(ready, key=order_key):
                new = gp[g] - S
                if len(S) + len(new) <= tile_bits and (best is None or len(new) < len(best_new)):
                    best, best_new = g, new
            if best is None:
                break
            S |= best_new
            take(best)
        return chosen, S, ptr

    while done < len(gates):
        best = None
        for t in range(max(1, tries)):
            if t == 0:
                key = lambda g: g  # noqa: E731  (program order)
            else:
                r = {}
                key = lambda g, r=r: r.setdefault(g, rnd.random())  # noqa: E731
            cand = grow(ptr, key)
            if best is None or len(cand[0]) > len(best[0]):
                best = cand
        chosen, S, new_ptr = best
        if not chosen:  # a gate that fits no tile (k > 4): run it on its own
            heads = {qlist[q][ptr[q]] for q in qubits if ptr[q] < len(qlist[q])}
            gi = min(g for g in heads if all(qlist[x][ptr[x]] == g for x in gq[g]))
            for q in gq[gi]:
                ptr[q] += 1
            ops.append(('G', np.asarray(gates[gi][0]), [pos_of[q] for q in reversed(gq[gi])]))
            done += 1
            continue
        ptr = new_ptr
        done += len(chosen)
        if len(chosen) < min_gates:
            for gi in chosen:
                ops.append(('G', np.asarray(gates[gi][0]), [pos_of[q] for q in reversed(gq[gi])]))
            continue
        p = 0  # pad the tile with the lowest unused positions (any bits do; low ones coalesce best)
        while len(S) < tile_bits:
            if p not in S:
                S.add(p)
            p += 1
        if inner_max == 'auto':
            # The greedy fusion depends on the order the gates arrive in, and inside a pass any topological order
            # of the dependency DAG is allowed: a few random ones are scored on qubit sets alone (no matrices) and
            # the cheapest is fused for real (benchmark circuit: -7 % inner-gate cost).
            chosen = _best_fusion_order(chosen, gq, fusion_orders, random.Random(seed * 7919 + len(ops)))
            # an inner gate costs about the same for k <= 3 and 1.9x that for k = 4 (INNER_COST): fuse to 3 qubits,
            # then let a second round merge neighbours into 4-qubit gates and keep it where that is cheaper
            inner = fuse([gates[gi] for gi in chosen], 3, complex_type=complex_type, exact_commutation=True)
            wider = fuse(inner, 4, complex_type=complex_type, exact_commutation=True)

            def cost(gl):  # a pass whose operand + address tables overflow the LDS left beside the tile runs the
                c = sum(INNER_COST[len(qs)] for _, qs in gl)  # slower global-memory variant of every gate
                return c * (1.0 if lds_bytes(gl, complex_type, tile_bits) <= lds_budget(tile_bits, complex_type) else 1.25)
            if cost(wider) < cost(inner):
                inner = wider
        elif inner_max:
            inner = fuse([gates[gi] for gi in chosen], inner_max, complex_type=complex_type, exact_commutation=True)
        else:
            inner = [(np.asarray(gates[gi][0]), gq[gi]) for gi in chosen]
        ops.append(('B', np.asarray(sorted(S), dtype=np.uint32),
                    [(U, [pos_of[q] for q in reversed(qs)]) for U, qs in inner]))
    return ops


def _plan_blocked_native(gates, pos_of, n, tile_bits, low_bits, inner_max, min_gates, tries, seed, complex_type, fusion_orders):
    from . import core
    from .fusion import exact_tolerance
    ctype = np.dtype(complex_type)
    single = ctype == np.dtype('complex64')
    # positions stand for the qubits: the planner sorts a fused gate's qubits by them (most significant first), this
    # module by label -- the same operator either way
    as_pos = [(U, [pos_of[q] for q in qs]) for U, qs in gates]
    tol = exact_tolerance(gates, ctype)
    kind, first, tile, gk, gpos, mats = core.plan_blocked(n, as_pos, tile_bits, low_bits, inner_max, min_gates, max(1, tries),
                                                           max(1, fusion_orders), 4 if single else 8, seed, tol)
    ops = []
    po = np.concatenate([[0], np.cumsum(gk)]).astype(np.int64)
    mo = np.concatenate([[0], np.cumsum(np.int64(1) << (2 * gk.astype(np.int64)))])
    mats = mats.astype(ctype)
    for i in range(len(kind)):
        inner = []
        for g in range(first[i], first[i + 1]):
            d = 1 << int(gk[g])
            inner.append((mats[mo[g]:mo[g + 1]].reshape(d, d), [int(p) for p in gpos[po[g]:po[g + 1]][::-1]]))  # LSB first
        if kind[i]:
            ops.append(('B', tile[i].copy(), inner))
        else:
            ops.append(('G', inner[0][0], inner[0][1]))
    return ops


def blocked_stats(ops):
    nb = sum(1 for op in ops if op[0] == 'B')
    ng = sum(1 for op in ops if op[0] == 'G')
    inner = [len(op[2]) for op in ops if op[0] == 'B']
    ks = {}
    for op in ops:
        if op[0] == 'B':
            for _, p in op[2]:
                ks[len(p)] = ks.get(len(p), 0) + 1
    return {'blocked_passes': nb, 'plain_gates': ng, 'inner_gates': int(sum(inner)), 'inner_k_histogram': ks}
