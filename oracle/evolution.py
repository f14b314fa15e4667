"""numpy restatement of the reference's evolution driver -- TEST INFRASTRUCTURE ONLY.

``evolve_reference_protocol`` follows hybridq/circuit/simulation/simulation.py:491-675
(the ``optimize='evolution-hybridq'`` branch of ``_simulate_evolution``): split
planes, logical->physical maps, the "swap the lowest 8 positions so that no target
sits below log2_pack_size" policy, ``pos`` built from ``reversed(gate.qubits)``,
final un-permute and ``to_complex``.  It drives any :class:`oracle.binding.OracleLib`
(the C port or the compiled reference) and can record the exact C-ABI call trace.

``evolve_tensordot`` is an independent float64 evolution (numpy tensordot /
moveaxis) used to cross-check both.

A circuit here is a list of ``(U, qubits)``: ``U`` a dense 2^k x 2^k complex matrix
whose row/column index has ``qubits[0]`` as its MOST significant bit (the
convention of ``gate.matrix()``, cf. simulation.py:633 which reverses the qubits
to get LSB-first positions), ``qubits`` a tuple of k distinct labels.  Labels are
sorted to fix the qubit->axis map exactly like ``Circuit.all_qubits()``
(hybridq/circuit/circuit.py:406-451).
"""
import time

import numpy as np

from .binding import aligned_empty


def all_qubits(gates, qubits=None):
    if qubits is not None:
        return list(qubits)
    return sorted({q for _, qs in gates for q in qs})


def apply_gate_numpy(psi, U, pos):
    """out[idx(b,t)] = sum_s U[t][s] in[idx(b,s)], idx(b,t) = b | sum_j t_j << pos[j]
    (include/U.h:77-99) on a flat complex array of length 2^n.  Returns a new array."""
    n = int(np.log2(psi.size))
    k = len(pos)
    U = np.asarray(U).reshape((2,) * (2 * k))
    # flat index bit p <-> axis n-1-p ; matrix index bit j <-> U axis k-1-j
    axes = [n - 1 - int(pos[k - 1 - a]) for a in range(k)]  # U axis a (MSB first) -> psi axis
    t = psi.reshape((2,) * n)
    out = np.tensordot(U, t, axes=(list(range(k, 2 * k)), axes))
    out = np.moveaxis(out, list(range(k)), axes)
    return np.ascontiguousarray(out).reshape(-1)


def swap_numpy(a, pos):
    """new[x] = old[(x & ~(2^s-1)) | sum_i ((x>>i)&1) << pos[i]] (include/swap.h:28-95)."""
    s = len(pos)
    x = np.arange(1 << s, dtype=np.int64)
    src = np.zeros_like(x)
    for i, p in enumerate(pos):
        src |= ((x >> i) & 1) << int(p)
    return np.ascontiguousarray(a.reshape(-1, 1 << s)[:, src]).reshape(a.shape)


def evolve_tensordot(gates, n=None, initial_state=None, qubits=None, dtype=np.complex128):
    """Independent evolution: psi <- moveaxis(tensordot(U, psi)) per gate, in `dtype`."""
    qubits = all_qubits(gates, qubits)
    n = len(qubits) if n is None else n
    if len(qubits) != n:
        raise ValueError(f'{n} qubits requested but the gates act on {len(qubits)}: pass qubits=[...] explicitly')
    index = {q: i for i, q in enumerate(qubits)}
    psi = _initial(initial_state, n, dtype).reshape((2,) * n)
    for U, qs in gates:
        k = len(qs)
        axes = [index[q] for q in qs]
        Ut = np.asarray(U, dtype=dtype).reshape((2,) * (2 * k))
        psi = np.moveaxis(np.tensordot(Ut, psi, axes=(list(range(k, 2 * k)), axes)), list(range(k)), axes)
    return np.ascontiguousarray(psi).reshape(-1)


def _initial(initial_state, n, dtype):
    if initial_state is None:
        psi = np.zeros(1 << n, dtype=dtype)
        psi[0] = 1
        return psi
    if isinstance(initial_state, str):
        # '01+-' product state, first character = first (most significant) qubit; the value
        # hybridq's prepare_state builds (circuit/simulation/utils.py:99-153: basis state,
        # uniform superposition, or kron of ones and parity signs over sqrt(#'+' * #'-')).
        s = initial_state * n if len(initial_state) == 1 else initial_state
        if len(s) != n:
            raise ValueError('initial_state has the wrong number of qubits')
        single = {'0': (1.0, 0.0), '1': (0.0, 1.0), '+': (1.0, 1.0), '-': (1.0, -1.0)}
        psi = np.ones(1, dtype=np.float64)
        for ch in s:
            psi = np.kron(psi, np.array(single[ch]))
        psi /= np.sqrt(2.0 ** sum(ch in '+-' for ch in s))
        return psi.astype(dtype)
    return np.array(initial_state, dtype=dtype).reshape(-1)


def evolve_reference_protocol(lib, gates, n=None, initial_state=None, qubits=None,
                              complex_type='complex64', trace=None, log2_pack_size=None,
                              planes=None, max_seconds=None, warmup_gates=0, to_complex=True,
                              checkpoints=None):
    """Replay the reference driver loop (simulation.py:491-675) on `lib`.

    Returns (psi_complex, info) with info['runtime (s)'] measured like the
    reference (loop + final restore, simulation.py:519,666).  If `trace` is a
    list, every C-ABI call is appended as ('S'|'U', pos list[, k]).

    Baseline-timing options (bench.py's cpu_baseline leg): `planes` = preallocated
    aligned (2, 2^n) float array already holding the initial state (avoids the
    complex temporary at n = 30); `warmup_gates` = leading gates executed before the
    clock starts (page faults, thread pool); `max_seconds` = stop after the first gate
    that crosses this budget (info['n_gates'] says how many were timed; the final
    restore is then skipped and the state is NOT the circuit's final state);
    `to_complex=False` skips the interleave and returns the planes.

    `checkpoints`: iterable of gate counts; info['checkpoints'][c] is the state (canonical qubit
    order, complex) after the first c gates -- taken on a COPY of the planes brought back to
    canonical order with the same final un-permute the loop ends with (:655-663)."""
    complex_type = np.dtype(complex_type)
    ft = np.dtype('float32') if complex_type == np.dtype('complex64') else np.dtype('float64')
    qubits = all_qubits(gates, qubits)
    n = len(qubits) if n is None else n
    if len(qubits) != n:
        raise ValueError(f'{n} qubits requested but the gates act on {len(qubits)}: pass qubits=[...] explicitly')
    L = lib.log2_pack_size if log2_pack_size is None else log2_pack_size

    if planes is None:
        psi0 = _initial(initial_state, n, complex_type)
        planes = aligned_empty((2, 1 << n), ft)  # simulation.py:491-494
        planes[0] = psi0.real
        planes[1] = psi0.imag
    re, im = planes[0], planes[1]

    _map = {q: n - x - 1 for x, q in enumerate(qubits)}  # simulation.py:512
    _inv = [q for q, _ in sorted(_map.items(), key=lambda kv: kv[1])]  # :513
    max_swap = 0
    t0 = time.perf_counter()
    n_timed = 0
    truncated = False
    want_cp = set(int(c) for c in checkpoints) if checkpoints is not None else set()
    snapshots = {}

    def snapshot(count):
        order = [_inv.index(q) for q in reversed(qubits)][:max_swap]
        cre, cim = aligned_empty(re.shape, ft), aligned_empty(im.shape, ft)
        cre[...] = re
        cim[...] = im
        if order:
            lib.swap(cre, order, n)
            lib.swap(cim, order, n)
        snapshots[count] = lib.to_complex(cre, cim)

    for gi, (U, qs) in enumerate(gates):
        if gi in want_cp:
            snapshot(gi)
        if gi == warmup_gates and warmup_gates:
            t0 = time.perf_counter()
            n_timed = 0
        if max_seconds is not None and gi > warmup_gates and time.perf_counter() - t0 > max_seconds:
            truncated = True
            break
        n_timed += 1
        k = len(qs)
        if any(q in _inv[:L] for q in qs):  # :559
            if k <= 4:  # :584-594
                order = [x for x, q in enumerate(_inv[:8]) if q not in qs]
                order += [x for x, q in enumerate(_inv[:8]) if q in qs]
            else:  # :596-605
                gidx = [_inv.index(q) for q in qs]
                order = [x for x in range(n) if x not in gidx][:L]
                order += [x for x in gidx if x < max(order)]
            s = len(order)
            max_swap = max(max_swap, s)
            _inv[:s] = [_inv[:s][x] for x in order]  # :615-619
            _map.update({q: x for x, q in enumerate(_inv[:s])})
            if lib.swap(re, order, n) or lib.swap(im, order, n):  # :623-630
                raise RuntimeError('swap failed')
            if trace is not None:
                trace.append(('S', list(order)))
        pos = [_map[q] for q in reversed(qs)]  # :633
        if lib.apply_U(re, im, U, pos, n):  # :640-646
            raise RuntimeError('something went wrong')
        if trace is not None:
            trace.append(('U', list(pos), k))
    order = [_inv.index(q) for q in reversed(qubits)][:max_swap]  # :655-657
    if order and not truncated:
        lib.swap(re, order, n)
        lib.swap(im, order, n)
        if trace is not None:
            trace.append(('S', list(order)))
    t1 = time.perf_counter()
    info = {'runtime (s)': t1 - t0, 'n_gates': n_timed, 'truncated': truncated}
    if checkpoints is not None:
        info['checkpoints'] = snapshots
    if not to_complex:
        return planes, info
    psi = lib.to_complex(re, im)  # :669-675
    return psi, info
