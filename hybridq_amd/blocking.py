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
from collections import deque

import numpy as np

from .fusion import fuse


#: relative cost of an inner gate by width (measured, n = 30 complex64: 0.63 ms for k <= 3, 1.20 ms for k = 4)
INNER_COST = {1: 1.0, 2: 1.0, 3: 1.0, 4: 1.9}


#: LDS next to a 64 KiB tile with two workgroups per CU (hq_hip.hip: a_budget), and what a gate needs of it:
#: its MFMA A-operand table (k = 2, 3: 256 elements, k = 4: 1024; k = 1 runs on the VALU from scalar registers)
#: plus 136 32-bit words of slot-address tables
LDS_TABLE_BUDGET = 15 * 1024


def lds_bytes(gate_list, complex_type='complex64'):
    elem = 4 if np.dtype(complex_type) == np.dtype('complex64') else 8
    return sum({1: 0, 2: 256, 3: 256, 4: 1024}[len(qs)] * elem + 544 for _, qs in gate_list)


def plan_blocked(gates, pos_of, n, tile_bits=13, low_bits=5, inner_max='auto', min_gates=3, tries=32, seed=0,
                 complex_type='complex64'):
    """Schedule `gates` ([(U, qubits), ...]) given the placement ``pos_of[qubit] = position``.

    Returns a list of ops:
        ('B', tile_pos uint32[tile_bits] ascending, [(U, positions LSB-first), ...])
        ('G', U, positions LSB-first)
    ``tries`` > 1: each pass is grown from several (seeded, deterministic) visiting orders of
    the ready gates and the one that absorbs the most gates is kept (n=30 depth-40 circuit:
    34 -> 30 passes for 16 tries, ~0.2 s of host time)."""
    import random
    tile_bits = min(tile_bits, n)
    low_bits = min(low_bits, tile_bits)
    gq = [tuple(qs) for _, qs in gates]
    gp = [frozenset(pos_of[q] for q in qs) for qs in gq]
    qubits = sorted(pos_of, key=lambda q: pos_of[q])
    queues = {q: deque() for q in qubits}
    for gi, qs in enumerate(gq):
        for q in qs:
            queues[q].append(gi)
    done = 0
    ops = []
    low = frozenset(range(low_bits))
    rnd = random.Random(seed)

    def grow(queues, order_key):
        """One candidate pass: returns (chosen gate indices in execution order, tile set, queues after)."""
        queues = {q: deque(v) for q, v in queues.items()}
        S = set(low)
        chosen = []

        def heads():
            return sorted({queues[q][0] for q in qubits if queues[q]}, key=order_key)

        def ready(gi):
            return all(queues[q][0] == gi for q in gq[gi])

        def take(gi):
            chosen.append(gi)
            for q in gq[gi]:
                queues[q].popleft()

        progress = True
        while progress:
            progress = False
            for gi in heads():  # everything ready that already fits
                if len(gq[gi]) <= 4 and ready(gi) and gp[gi] <= S:
                    take(gi)
                    progress = True
            if progress:
                continue
            best, best_new = None, None  # spend spare capacity on the cheapest ready gate
            for gi in heads():
                if not ready(gi) or len(gq[gi]) > 4:
                    continue
                new = gp[gi] - S
                if len(S) + len(new) <= tile_bits and (best is None or len(new) < len(best_new)):
                    best, best_new = gi, new
            if best is not None:
                S |= best_new
                take(best)
                progress = True
        return chosen, S, queues

    while done < len(gates):
        best = None
        for t in range(max(1, tries)):
            if t == 0:
                key = lambda g: g  # noqa: E731  (program order)
            else:
                r = {}
                key = lambda g, r=r: r.setdefault(g, rnd.random())  # noqa: E731
            cand = grow(queues, key)
            if best is None or len(cand[0]) > len(best[0]):
                best = cand
        chosen, S, new_queues = best
        if not chosen:  # a gate that fits no tile (k > 4): run it on its own
            gi = min(queues[q][0] for q in qubits if queues[q] and all(queues[x][0] == queues[q][0] for x in gq[queues[q][0]]))
            for q in gq[gi]:
                queues[q].popleft()
            ops.append(('G', np.asarray(gates[gi][0]), [pos_of[q] for q in reversed(gq[gi])]))
            done += 1
            continue
        queues = new_queues
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
            # an inner gate costs about the same for k <= 3 and 1.6x that for k = 4 (INNER_COST): fuse to 3 qubits,
            # then let a second round merge neighbours into 4-qubit gates and keep it where that is cheaper
            inner = fuse([gates[gi] for gi in chosen], 3, complex_type=complex_type)
            wider = fuse(inner, 4, complex_type=complex_type)

            def cost(gl):  # a pass whose operand + address tables overflow the LDS left beside the tile runs the
                c = sum(INNER_COST[len(qs)] for _, qs in gl)  # slower global-memory variant of every gate
                return c * (1.0 if lds_bytes(gl, complex_type) <= LDS_TABLE_BUDGET else 1.25)
            if cost(wider) < cost(inner):
                inner = wider
        elif inner_max:
            inner = fuse([gates[gi] for gi in chosen], inner_max, complex_type=complex_type)
        else:
            inner = [(np.asarray(gates[gi][0]), gq[gi]) for gi in chosen]
        ops.append(('B', np.asarray(sorted(S), dtype=np.uint32),
                    [(U, [pos_of[q] for q in reversed(qs)]) for U, qs in inner]))
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
