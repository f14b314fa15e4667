#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ FROM THE REFERENCE ITSELF.

Run in the build container only (needs /root/reference and oracle/_ref built by
`make -C oracle ref`); the outputs (.npz, data only) are committed, the reference is not.

    cd /tmp && LD_LIBRARY_PATH=/root/repo/oracle/_ref PYTHONDONTWRITEBYTECODE=1 \
        python /root/repo/tests/golden/make_golden.py

What is recorded
  calls_apply_U.npz   per-call vectors of apply_U_float32/64 produced by the reference's
                      compiled C++ core (include/python_U.cpp -> oracle/_ref/hybridq.so):
                      k = 1..6, sorted/unsorted positions (>= 3, U.h:48-54), unitary and
                      non-unitary U, inputs and outputs.
  calls_swap.npz      swap_* of arange(2^n) by the reference core, n_pos = 1..12.
  e2e_simple_qasm.npz examples/circuit_simple.qasm (24 qubits, 99 gates) through the
                      reference's simulate(optimize='evolution-hybridq'): gate list as
                      (name, qubits), the C-ABI call trace with the fused 16x16 matrices,
                      and a strided sample / head / norm of the final state.
  e2e_rqc.npz         hybridq.extras.random.get_rqc circuits (n = 12) through simulate():
                      gate list as dense (U, qubits), final states for complex64
                      (compress=4) and complex128 (compress=0).
  e2e_dm.npz          a 6-qubit circuit with depolarizing noise through
                      hybridq.dm.circuit.simulation.simulate (-> 12-qubit state vector):
                      the C-ABI call trace (swaps + fused NON-unitary gates) and the final rho.
  e2e_dm_circuit.npz  the same noisy circuit UNFUSED, as data: unitaries, Kraus operators,
                      weights and the reference's superoperator matrices (python
                      make_golden.py dm_circuit).
  e2e_api.npz         host-level API around the path (python make_golden.py api): prepare_state,
                      simulate() with mixed initial states / compress 4 and 8, Projection and
                      Measure gates, expectation_value(), utils.dot(), utils.transpose().
  e2e_containers.npz  TupleGate / StochasticGate (allow_sampling, sampling_seed) / zero-qubit MessageGate circuits through
                      the reference's simulate() (python make_golden.py containers).
  e2e_matrix.npz      utils.matrix / compress / to_matrix_gate on small circuits (python make_golden.py matrix).
  e2e_fn_streams.npz  circuits with Projection / Measure gates through the reference's simplify / compress / simulate: order of the
                      resulting gate lists, matrices as probe products, projection-run states (python make_golden.py fn_streams).
The import of the reference Python needs stand-ins for three absent third-party modules
(opt_einsum, more_itertools, numba); none of them is on the evolution-hybridq path except
numba.vectorize for '+-' initial states.
"""
import ctypes
import itertools
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, ROOT)


def install_stubs():
    oe = types.ModuleType('opt_einsum')
    oe.contract = lambda path, *ops, **kw: np.einsum(path, *ops)
    oe.get_symbol = lambda i: chr(ord('a') + i) if i < 26 else chr(ord('A') + i - 26)
    oec = types.ModuleType('opt_einsum.contract')
    oec.PathInfo = type('PathInfo', (), {})
    sys.modules['opt_einsum'] = oe
    sys.modules['opt_einsum.contract'] = oec
    mi = types.ModuleType('more_itertools')
    mi.flatten = lambda it: itertools.chain.from_iterable(it)

    def chunked(it, n):
        it = iter(it)
        while True:
            c = list(itertools.islice(it, n))
            if not c:
                return
            yield c

    mi.chunked = chunked
    mi.ichunked = chunked
    mi.distribute = lambda n, it: [list(it)[i::n] for i in range(n)]
    sys.modules['more_itertools'] = mi
    nb = types.ModuleType('numba')

    def vectorize(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return np.vectorize(a[0])
        return lambda f: np.vectorize(f)

    def njit(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f

    nb.vectorize, nb.njit, nb.jit, nb.prange = vectorize, njit, njit, range
    sys.modules['numba'] = nb


def per_call_vectors():
    import oracle
    from oracle.binding import aligned_empty
    ref = oracle.load_ref()
    rng = np.random.default_rng(20260928)
    n = 10
    out = {}
    idx = 0
    for ft in (np.float32, np.float64):
        for k in range(1, 7):
            for variant in range(3):
                pos = rng.permutation(np.arange(3, n))[:k]
                if variant == 0:
                    pos = np.sort(pos)
                d = 1 << k
                U = rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d))
                if variant == 2:  # unitary
                    U = np.linalg.qr(U)[0]
                U = U.astype(np.complex64 if ft == np.float32 else np.complex128)
                pl = aligned_empty((2, 1 << n), ft)
                pl[:] = rng.standard_normal((2, 1 << n))
                inp = pl.copy()
                assert ref.apply_U(pl[0], pl[1], U, pos) == 0
                out[f'c{idx}_in'] = inp
                out[f'c{idx}_out'] = pl.copy()
                out[f'c{idx}_U'] = U
                out[f'c{idx}_pos'] = pos.astype(np.uint32)
                idx += 1
    out['n_cases'] = idx
    out['n_qubits'] = n
    np.savez_compressed(os.path.join(HERE, 'calls_apply_U.npz'), **out)
    print('calls_apply_U.npz:', idx, 'cases')

    out = {}
    idx = 0
    n = 12
    for s in range(1, 13):
        pos = rng.permutation(s).astype(np.uint32)
        a = aligned_empty(1 << n, np.uint32, alignment=4096)
        a[:] = np.arange(1 << n)
        assert ref.swap(a, pos) == 0
        out[f's{idx}_pos'] = pos
        out[f's{idx}_out'] = a.astype(np.uint16 if n <= 16 else np.uint32)
        idx += 1
    out['n_cases'] = idx
    out['n_qubits'] = n
    np.savez_compressed(os.path.join(HERE, 'calls_swap.npz'), **out)
    print('calls_swap.npz:', idx, 'cases')


class Tracer:
    """Wrap the reference's ctypes function tables (looked up at call time,
    simulation.py:482-485) to record the C-ABI call sequence."""

    def __init__(self, sim):
        self.sim = sim
        self.calls = []
        self._dot = dict(sim._dot_core)
        self._swap = dict(sim._swap_core)

    def __enter__(self):
        calls = self.calls

        def wrap_dot(f, ft):
            def g(re, im, U, pos, n, k):
                d = 1 << k
                Uarr = np.ctypeslib.as_array(U, shape=(2 * d * d,)).copy()
                parr = np.ctypeslib.as_array(pos, shape=(k,)).copy()
                calls.append(('U', parr, Uarr.view(np.complex64 if ft == np.float32 else np.complex128).reshape(d, d)))
                return f(re, im, U, pos, n, k)
            return g

        def wrap_swap(f):
            def g(a, pos, n, s):
                calls.append(('S', np.ctypeslib.as_array(pos, shape=(s,)).copy(), None))
                return f(a, pos, n, s)
            return g

        for dt, f in self._dot.items():
            self.sim._dot_core[dt] = wrap_dot(f, dt.type)
        for dt, f in self._swap.items():
            self.sim._swap_core[dt] = wrap_swap(f)
        return self

    def __exit__(self, *a):
        self.sim._dot_core.update(self._dot)
        self.sim._swap_core.update(self._swap)


def pack_trace(calls, prefix, out):
    kinds = ''.join(c[0] for c in calls)
    out[prefix + 'kinds'] = np.frombuffer(kinds.encode(), dtype=np.uint8)
    for i, (kind, pos, U) in enumerate(calls):
        out[f'{prefix}{i}_pos'] = np.asarray(pos, dtype=np.uint32)
        if U is not None:
            out[f'{prefix}{i}_U'] = U


def end_to_end():
    install_stubs()
    sys.path.insert(0, REF)
    import hybridq.circuit.simulation.simulation as sim
    from hybridq.circuit.simulation import simulate
    from hybridq.extras.io.qasm import from_qasm
    from hybridq.extras.random import get_rqc
    from hybridq.circuit import utils
    assert sim._log2_pack_size == 3, 'reference core not found: set LD_LIBRARY_PATH=oracle/_ref'

    # ---- examples/circuit_simple.qasm -------------------------------------------------
    c = from_qasm(open(os.path.join(REF, 'examples', 'circuit_simple.qasm')).read())
    qubits = c.all_qubits()
    n = len(qubits)
    out = {}
    names = [g.name for g in c]
    out['gate_names'] = np.array(names)
    out['gate_qubits'] = np.array([list(g.qubits) + [-1] * (2 - len(g.qubits)) for g in c], dtype=np.int32)
    uniq = sorted(set(names))
    out['matrix_names'] = np.array(uniq)
    for nm in uniq:
        out['matrix_' + nm] = np.asarray(next(g for g in c if g.name == nm).matrix(), dtype=np.complex128)
    with Tracer(sim) as tr:
        psi, info = simulate(c, initial_state='0', optimize='evolution-hybridq', complex_type='complex64',
                             return_info=True, verbose=False)
    psi = psi.reshape(-1)
    pack_trace(tr.calls, 'trace_', out)
    out['n_qubits'] = n
    out['sample_stride'] = (1 << n) // 4096
    out['psi_sample'] = psi[::(1 << n) // 4096].copy()
    out['psi_head'] = psi[:8].copy()
    out['norm2'] = float(np.vdot(psi, psi).real)
    np.savez_compressed(os.path.join(HERE, 'e2e_simple_qasm.npz'), **out)
    print('e2e_simple_qasm.npz: n =', n, 'gates =', len(c), 'calls =', ''.join(k for k, _, _ in tr.calls),
          'norm2 =', out['norm2'], 'psi[0:2] =', psi[:2])

    # ---- seeded reference RQCs ---------------------------------------------------------
    out = {}
    np.random.seed(1234)
    n = 12
    for tag, ct, compress in (('a', 'complex64', 4), ('b', 'complex128', 0)):
        c = get_rqc(n, 60, use_random_indexes=False)
        qubits = c.all_qubits()
        c = utils.flatten(c) if hasattr(utils, 'flatten') else c
        mats, qs = [], []
        for g in c:
            mats.append(np.asarray(g.matrix(), dtype=np.complex128))
            qs.append([qubits.index(q) for q in g.qubits])
        with Tracer(sim) as tr:
            psi = simulate(c, initial_state='0', optimize='evolution-hybridq', complex_type=ct,
                           compress=compress, simplify=False, remove_id_gates=False, verbose=False)
        out[f'{tag}_n_gates'] = len(mats)
        for i, (U, q) in enumerate(zip(mats, qs)):
            out[f'{tag}_U{i}'] = U
            out[f'{tag}_q{i}'] = np.asarray(q, dtype=np.int32)
        out[f'{tag}_psi'] = psi.reshape(-1)
        pack_trace(tr.calls, f'{tag}_trace_', out)
        out[f'{tag}_calls'] = np.frombuffer(''.join(k for k, _, _ in tr.calls).encode(), dtype=np.uint8)
        print(f'e2e_rqc {tag}: {ct} compress={compress} gates={len(mats)} calls={"".join(k for k, _, _ in tr.calls)}')
    out['n_qubits'] = n
    np.savez_compressed(os.path.join(HERE, 'e2e_rqc.npz'), **out)

    # ---- density matrix with noise -> 2n-qubit state vector ----------------------------
    import hybridq.dm.circuit.simulation as dmsim
    from hybridq.noise.utils import add_depolarizing_noise
    from hybridq.gate import Gate
    np.random.seed(4321)
    nq = 6
    circ = get_rqc(nq, 20, use_random_indexes=False)
    noisy = add_depolarizing_noise(circ, probs=(0.01, 0.02))
    with Tracer(sim) as tr:
        rho = dmsim.simulate(noisy, initial_state='0', optimize='evolution-hybridq', verbose=False)
    # what the evolution driver was asked to do, at the C ABI: swaps + fused (non-unitary) gates
    out = {'n_qubits': 2 * nq}
    pack_trace(tr.calls, 'trace_', out)
    out['rho'] = np.asarray(rho).reshape(-1)
    np.savez_compressed(os.path.join(HERE, 'e2e_dm.npz'), **out)
    r = np.asarray(rho).reshape(1 << nq, 1 << nq)
    print('e2e_dm.npz: 2n =', 2 * nq, 'trace(rho) =', np.trace(r).real,
          'calls =', ''.join(k for k, _, _ in tr.calls))


def dm_circuit():
    """e2e_dm_circuit.npz: the UNFUSED noisy circuit of e2e_dm.npz as data -- unitary gates
    (U, qubits) and channels (Kraus operators L_i = R_i, weights s_i, qubits, and the
    superoperator matrix gate.map() the reference builds from them) -- plus the final rho."""
    install_stubs()
    sys.path.insert(0, REF)
    import hybridq.dm.circuit.simulation as dmsim
    from hybridq.extras.random import get_rqc
    from hybridq.noise.utils import add_depolarizing_noise
    np.random.seed(4321)
    nq = 6
    circ = get_rqc(nq, 20, use_random_indexes=False)
    noisy = add_depolarizing_noise(circ, probs=(0.01, 0.02))
    out = {'n_qubits': nq, 'n_items': len(noisy)}
    kinds = ''
    for i, g in enumerate(noisy):
        out[f'q{i}'] = np.asarray([int(q) for q in g.qubits], dtype=np.int32)
        if hasattr(g, 'Kraus'):
            kinds += 'K'
            L, R = g.Kraus.gates
            out[f'L{i}'] = np.stack([np.asarray(x.matrix(), dtype=np.complex128) for x in L])
            out[f'R{i}'] = np.stack([np.asarray(x.matrix(), dtype=np.complex128) for x in R])
            out[f's{i}'] = np.asarray(g.Kraus.s, dtype=np.float64)
            out[f'M{i}'] = np.asarray(g.map(), dtype=np.complex128)
        else:
            kinds += 'U'
            out[f'U{i}'] = np.asarray(g.matrix(), dtype=np.complex128)
    out['kinds'] = np.frombuffer(kinds.encode(), dtype=np.uint8)
    rho = dmsim.simulate(noisy, initial_state='0', optimize='evolution-hybridq', verbose=False)
    out['rho'] = np.asarray(rho).reshape(-1)
    ref = np.load(os.path.join(HERE, 'e2e_dm.npz'))['rho']
    assert np.abs(out['rho'] - ref).max() < 1e-6, 'not the same circuit as e2e_dm.npz'
    np.savez_compressed(os.path.join(HERE, 'e2e_dm_circuit.npz'), **out)
    print('e2e_dm_circuit.npz:', kinds)


def api_vectors():
    """e2e_api.npz: outputs of the reference's host-level API around the hot path, as data:
    prepare_state strings, simulate() with mixed initial states (complex64 compress=4 /
    complex128 compress=8, non-unitary gates, like tests.py:2335-2369), Projection and Measure
    functional gates inside and outside simulate(), expectation_value(), utils.dot() through
    the compiled core (split planes, complex input, swap_back=False) and utils.transpose()."""
    install_stubs()
    sys.path.insert(0, REF)
    import hybridq.circuit.simulation.simulation as sim
    from hybridq.circuit import Circuit
    from hybridq.circuit.simulation import expectation_value, prepare_state, simulate
    from hybridq.extras.random import get_rqc
    from hybridq.gate import Gate, Measure, Projection
    from hybridq.utils.dot import dot
    from hybridq.utils.transpose import transpose
    assert sim._log2_pack_size == 3, 'reference core not found: set LD_LIBRARY_PATH=oracle/_ref'
    out = {}

    # ---- prepare_state -----------------------------------------------------------------
    strings = ['0', '1', '+', '-', '00101', '+++++', '01+-0', '-+10+-', '1-', '--+0110+']
    out['ps_strings'] = np.array(strings)
    for i, st in enumerate(strings):
        out[f'ps_{i}'] = np.asarray(prepare_state(st, complex_type='complex128')).reshape(-1)

    def dump_circuit(c, qubits, tag):
        mats = [np.asarray(g.matrix(), dtype=np.complex128) for g in c]
        out[f'{tag}_n_gates'] = len(mats)
        for i, (U, g) in enumerate(zip(mats, c)):
            out[f'{tag}_U{i}'] = U
            out[f'{tag}_q{i}'] = np.asarray([qubits.index(q) for q in g.qubits], dtype=np.int32)

    # ---- simulate: mixed initial state, non-unitary gates, compress 4 / 8 -------------------
    np.random.seed(2024)
    n = 12
    c = get_rqc(n, 150, use_random_indexes=False, use_unitary_only=False)
    qubits = c.all_qubits()
    assert qubits == list(range(n))
    init = ''.join(np.random.choice(list('01'), size=3)) + ''.join(np.random.choice(list('01+-'), size=n - 3))
    dump_circuit(c, qubits, 'sim')
    out['sim_init'] = np.array(init)
    out['sim_psi64'] = simulate(c, initial_state=init, optimize='evolution-hybridq', complex_type='complex64',
                                compress=4, simplify=False, remove_id_gates=False, verbose=False).reshape(-1)
    out['sim_psi128'] = simulate(c, initial_state=init, optimize='evolution-hybridq', complex_type='complex128',
                                 compress=8, simplify=False, remove_id_gates=False, verbose=False).reshape(-1)

    # ---- Projection / Measure on a raw state (tests.py:948-1008, :819-945) -------------------
    np.random.seed(77)
    n = 10
    r = (np.random.random((2,) * n) + 1).astype('complex64') * np.exp(1j * np.random.random((2,) * n)).astype('complex64')
    order = tuple(int(x) for x in np.random.permutation(n) * 7 - 20)  # arbitrary integer labels
    out['fg_state'] = r.reshape(-1)
    out['fg_order'] = np.asarray(order, dtype=np.int64)
    pq = [order[4], order[0], order[7]]
    out['proj_qubits'] = np.asarray(pq, dtype=np.int64)
    out['proj_string'] = np.array('101')
    P = Projection(state='101', qubits=pq)
    out['proj_raw'] = np.asarray(P(np.array(r), order, renormalize=False)[0]).reshape(-1)
    out['proj_norm'] = np.asarray(P(np.array(r), order, renormalize=True)[0]).reshape(-1)
    mq = [order[2], order[9], order[5], order[1]]
    out['meas_qubits'] = np.asarray(mq, dtype=np.int64)
    M = Measure(qubits=mq)
    out['meas_probs'] = np.asarray(M(np.array(r / np.linalg.norm(r.reshape(-1))), order, get_probs_only=True))
    for j, seed in enumerate((5, 6, 7)):
        np.random.seed(seed)
        psi_m, _ = M(np.array(r / np.linalg.norm(r.reshape(-1))), order)
        out[f'meas_seed{j}'] = seed
        out[f'meas_state{j}'] = np.asarray(psi_m).reshape(-1)

    # ---- functional gates inside simulate(): projection in the middle of a circuit -----------
    np.random.seed(99)
    n = 12
    c1 = get_rqc(n, 40, use_random_indexes=False)
    c2 = get_rqc(n, 40, use_random_indexes=False)
    qubits = list(range(n))
    dump_circuit(c1, qubits, 'fs1')
    dump_circuit(c2, qubits, 'fs2')
    out['fs_proj_qubits'] = np.asarray([3, 8], dtype=np.int64)
    out['fs_proj_string'] = np.array('01')
    circ = Circuit(list(c1) + [Projection(state='01', qubits=[3, 8])] + list(c2))
    out['fs_psi'] = simulate(circ, initial_state='0' * n, optimize='evolution-hybridq', complex_type='complex64',
                             compress=4, simplify=False, remove_id_gates=False, verbose=False).reshape(-1)

    # ---- expectation_value (tests.py:2374-2456) ------------------------------------------------
    np.random.seed(314)
    n = 12  # n <= 10 silently falls back to einsum (simulation.py:400)
    c = get_rqc(n, 60, use_random_indexes=False)
    qubits = c.all_qubits()
    op = get_rqc(2, 3, indexes=qubits[3:5], use_random_indexes=False)
    dump_circuit(c, qubits, 'ev')
    dump_circuit(op, qubits, 'evop')
    psi = simulate(c, initial_state='+' * n, optimize='evolution-hybridq', complex_type='complex64', simplify=False,
                   remove_id_gates=False, verbose=False)
    out['ev_state'] = psi.reshape(-1)
    out['ev_value'] = np.asarray(expectation_value(state=psi, op=op, qubits_order=qubits, remove_id_gates=False,
                                                   simplify=False, verbose=False), dtype=np.complex128)

    # ---- utils.dot through the compiled core (tests.py:299-391) -------------------------------
    np.random.seed(2718)
    n = 12
    for j, (k, t) in enumerate(((1, 'float32'), (3, 'float32'), (4, 'float64'), (6, 'float32'), (7, 'float64'))):
        psi = np.random.random((2, 2**n)).astype(t) - 0.5
        ct = (1j * psi[0][:1]).dtype
        U = (np.random.random((2**k, 2**k)) + 1j * np.random.random((2**k, 2**k)) - 0.5 - 0.5j).astype(ct)
        axes_b = np.random.choice(n, size=k, replace=False)
        res = dot(U, np.reshape(np.array(psi), (2,) * (n + 1)), axes_b=axes_b, b_as_complex_array=True,
                  raise_if_hcore_fails=True)
        res_c = dot(U, np.reshape(psi[0] + 1j * psi[1], (2,) * n), axes_b=axes_b, raise_if_hcore_fails=True)
        nsb, tr = dot(U, np.reshape(np.array(psi), (2,) * (n + 1)), axes_b=axes_b, b_as_complex_array=True,
                      swap_back=False, raise_if_hcore_fails=True)
        out[f'dot{j}_psi'] = psi
        out[f'dot{j}_U'] = U
        out[f'dot{j}_axes'] = np.asarray(axes_b, dtype=np.int64)
        out[f'dot{j}_res'] = np.asarray(res).reshape(2, -1)
        out[f'dot{j}_res_complex'] = np.asarray(res_c).reshape(-1)
        out[f'dot{j}_noswap'] = np.asarray(nsb).reshape(2, -1)
        out[f'dot{j}_tr'] = np.asarray([-1] if tr is None else tr, dtype=np.int64)
    out['dot_n'] = n
    out['dot_cases'] = 5

    # ---- utils.transpose (tests.py:256-296) -----------------------------------------------------
    np.random.seed(1618)
    for j, (n, t) in enumerate(((12, 'float32'), (14, 'int64'), (13, 'uint32'), (12, 'float64'))):
        a = (np.random.random((2,) * n) * 1000).astype(t)
        axes = np.arange(n)
        m = int(np.random.randint(3, 9))
        axes[-m:] = np.random.permutation(axes[-m:])
        b = transpose(np.array(a), axes, raise_if_hcore_fails=True)
        assert np.array_equal(b, np.transpose(a, axes))
        out[f'tr{j}_a'] = a.reshape(-1)
        out[f'tr{j}_axes'] = axes.astype(np.int64)
        out[f'tr{j}_res'] = np.asarray(b).reshape(-1)
        out[f'tr{j}_n'] = n
    out['tr_cases'] = 4
    np.savez_compressed(os.path.join(HERE, 'e2e_api.npz'), **out)
    print('e2e_api.npz:', len(out), 'arrays;', 'sim init', init, '; ev', out['ev_value'], '; dot tr',
          [out[f'dot{j}_tr'].tolist() for j in range(5)])


def matrix_vectors():
    """e2e_matrix.npz: hybridq.circuit.utils.matrix (circuit -> dense unitary, the routine behind
    to_matrix_gate, circuit/utils.py:688-807) on small random circuits, default order and a
    permuted order, plus the layers of utils.compress(max_n_qubits=3) as (qubits, matrix)."""
    install_stubs()
    sys.path.insert(0, REF)
    from hybridq.circuit import utils
    from hybridq.extras.random import get_rqc
    out = {}
    np.random.seed(808)
    for tag, n, ng in (('a', 4, 12), ('b', 6, 25)):
        c = get_rqc(n, ng, use_random_indexes=False)
        qubits = c.all_qubits()
        out[f'{tag}_n_gates'] = len(c)
        for i, g in enumerate(c):
            out[f'{tag}_U{i}'] = np.asarray(g.matrix(), dtype=np.complex128)
            out[f'{tag}_q{i}'] = np.asarray([int(q) for q in g.qubits], dtype=np.int32)
        out[f'{tag}_qubits'] = np.asarray(qubits, dtype=np.int32)
        out[f'{tag}_matrix'] = np.asarray(utils.matrix(c, complex_type='complex128', max_compress=0))
        order = [int(q) for q in np.random.permutation(qubits)]
        out[f'{tag}_order'] = np.asarray(order, dtype=np.int32)
        out[f'{tag}_matrix_order'] = np.asarray(utils.matrix(c, order=order, complex_type='complex128'))
        layers = utils.compress(c, max_n_qubits=3)
        out[f'{tag}_n_layers'] = len(layers)
        for j, layer in enumerate(layers):
            mg = utils.to_matrix_gate(layer, complex_type='complex128')
            out[f'{tag}_layer{j}_qubits'] = np.asarray([int(q) for q in mg.qubits], dtype=np.int32)
            out[f'{tag}_layer{j}_matrix'] = np.asarray(mg.matrix(), dtype=np.complex128)
    # utils.simplify on a circuit with planted identities, inverse pairs and commuting gates
    from hybridq.circuit import Circuit
    from hybridq.gate import Gate
    np.random.seed(909)
    base = list(get_rqc(5, 30, use_random_indexes=False))
    planted = []
    for i, g in enumerate(base):
        planted.append(g)
        if i % 4 == 1:
            planted.append(Gate('I', qubits=[int(np.random.randint(5))]))
        if i % 5 == 2:  # an inverse pair separated by a gate on other qubits
            h = base[(i * 7) % len(base)]
            others = [q for q in range(5) if q not in h.qubits]
            planted += [h, Gate('Z', qubits=[others[0]]) if others else Gate('I', qubits=[0]), h.inv()]
    c = Circuit(planted)
    out['s_n_gates'] = len(c)
    for i, g in enumerate(c):
        out[f's_U{i}'] = np.asarray(g.matrix(), dtype=np.complex128)
        out[f's_q{i}'] = np.asarray([int(q) for q in g.qubits], dtype=np.int32)
    sc = utils.simplify(c, remove_id_gates=True, verbose=False)
    out['s_simplified_n'] = len(sc)
    for i, g in enumerate(sc):
        out[f's_simplified_U{i}'] = np.asarray(g.matrix(), dtype=np.complex128)
        out[f's_simplified_q{i}'] = np.asarray([int(q) for q in g.qubits], dtype=np.int32)
    print('simplify:', len(c), '->', len(sc))
    np.savez_compressed(os.path.join(HERE, 'e2e_matrix.npz'), **out)
    print('e2e_matrix.npz:', {k: v.shape for k, v in out.items() if k.endswith('matrix')})


def container_vectors():
    """e2e_containers.npz (python make_golden.py containers): the reference's simulate() on circuits holding container
    gates -- TupleGate chunks (tests.py:1942-1977), a StochasticGate drawn with allow_sampling / sampling_seed
    (tests.py:2111-2197, simulation.py:241-256) and zero-qubit MessageGates (tests.py:1980-2034) -- as data: every
    gate's matrix and qubits, the probabilities, the seeds, and the final states the reference returned."""
    install_stubs()
    sys.path.insert(0, REF)
    import io
    import hybridq.circuit.simulation.simulation as sim
    from hybridq.circuit import Circuit
    from hybridq.circuit.simulation import simulate
    from hybridq.extras.gate import Gate as ExtraGate
    from hybridq.extras.random import get_rqc
    from hybridq.gate import Gate
    assert sim._log2_pack_size == 3, 'reference core not found: set LD_LIBRARY_PATH=oracle/_ref'
    out = {}
    np.random.seed(77)
    n = 12

    def dump(c, tag):
        out[f'{tag}_n_gates'] = len(c)
        for i, g in enumerate(c):
            out[f'{tag}_U{i}'] = np.asarray(g.matrix(), dtype=np.complex128)
            out[f'{tag}_q{i}'] = np.asarray([int(q) for q in g.qubits], dtype=np.int32)

    c = get_rqc(n, 80, use_random_indexes=False, use_unitary_only=False)
    assert c.all_qubits() == list(range(n))
    init = ''.join(np.random.choice(list('01+-'), size=n))
    out['init'] = np.array(init)
    dump(c, 'tup')
    gs = list(c)
    chunks = Circuit(Gate('TUPLE', gates=gs[i:i + 4]) for i in range(0, len(gs), 4))
    whole = Circuit([Gate('TUPLE', gates=gs)])
    psi = simulate(c, initial_state=init, optimize='evolution', complex_type='complex128', compress=0, simplify=False)
    psi_chunks = simulate(chunks, initial_state=init, optimize='evolution', complex_type='complex128', compress=0, simplify=False)
    psi_whole = simulate(whole, initial_state=init, optimize='evolution', complex_type='complex128', compress=0, simplify=False)
    assert np.array_equal(psi, psi_chunks) and np.array_equal(psi, psi_whole)
    out['tup_psi'] = np.asarray(psi).reshape(-1)
    # stochastic gate between the two halves of the circuit
    cand = get_rqc(n, 8, use_random_indexes=False, use_unitary_only=False)
    prob = np.random.random(len(cand))
    prob /= prob.sum()
    dump(cand, 'stoc')
    out['stoc_p'] = prob
    stoc = Gate('STOC', gates=list(cand), p=prob)
    seeds = [3, 11, 12345]
    out['stoc_seeds'] = np.asarray(seeds)
    for s_ in seeds:
        r = simulate(Circuit(gs[:40]) + [stoc] + Circuit(gs[40:]), initial_state=init, optimize='evolution', complex_type='complex128',
                     compress=0, simplify=False, allow_sampling=True, sampling_seed=s_)
        out[f'stoc_psi_{s_}'] = np.asarray(r).reshape(-1)
    # zero-qubit MessageGates after every gate
    buf = io.StringIO()
    msg = Circuit(x for i, g in enumerate(gs) for x in (g, ExtraGate('MESSAGE', qubits=tuple(), message=f'{i}', file=buf)))
    r = simulate(msg, initial_state=init, optimize='evolution', complex_type='complex128')
    out['msg_psi'] = np.asarray(r).reshape(-1)
    buf.seek(0)
    out['msg_lines'] = np.asarray([int(x.strip()) for x in buf.readlines()])
    np.savez_compressed(os.path.join(HERE, 'e2e_containers.npz'), **out)
    print('tuple / stochastic / message vectors written;', len(out['msg_lines']), 'messages')


def live_cases(path, seed):
    """python make_golden.py live OUT.npz SEED: random circuits through the reference's simulate() under random options,
    for the live differential test (tests/test_reference_live_host.py runs this in a subprocess when /root/reference is
    there and compares this package's host side, on the numpy test double, case by case)."""
    install_stubs()
    sys.path.insert(0, REF)
    import hybridq.circuit.simulation.simulation as sim
    from hybridq.circuit.simulation import simulate
    from hybridq.extras.random import get_rqc
    assert sim._log2_pack_size == 3, 'reference core not found: set LD_LIBRARY_PATH=oracle/_ref'
    rng = np.random.default_rng(seed)
    np.random.seed(seed)
    out = {'n_cases': 0}
    for i in range(10):
        n = int(rng.integers(11, 14))
        unitary = bool(rng.integers(0, 2))
        c = get_rqc(n, int(rng.integers(40, 120)), use_random_indexes=False, use_unitary_only=unitary)
        qubits = c.all_qubits()
        init = ''.join(rng.choice(list('01+-'), size=len(qubits)))
        compress = int(rng.choice([0, 2, 4, 6]))
        simplify = bool(rng.integers(0, 2))
        ctype = 'complex128' if i % 3 else 'complex64'
        try:
            psi = simulate(c, initial_state=init, optimize='evolution-hybridq', complex_type=ctype, compress=compress, simplify=simplify)
        except ValueError as e:  # "Active qubits have changed after simplification": a qubit lost its only gates; next circuit
            print('case skipped:', e)
            continue
        out[f'c{i}_n_gates'] = len(c)
        for j, g in enumerate(c):
            out[f'c{i}_U{j}'] = np.asarray(g.matrix(), dtype=np.complex128)
            out[f'c{i}_q{j}'] = np.asarray([qubits.index(q) for q in g.qubits], dtype=np.int32)
        out[f'c{i}_n'] = len(qubits)
        out[f'c{i}_names'] = np.array([g.name for g in c])
        out[f'c{i}_unitary'] = unitary
        out[f'c{i}_init'] = np.array(init)
        out[f'c{i}_compress'] = compress
        out[f'c{i}_simplify'] = simplify
        out[f'c{i}_ctype'] = np.array(ctype)
        out[f'c{i}_psi'] = np.asarray(psi).reshape(-1)
        # a Projection in the middle of the same circuit, and an expectation value on the final state
        from hybridq.circuit import Circuit as _Circuit
        from hybridq.circuit.simulation import expectation_value
        from hybridq.gate import Projection
        k = int(rng.integers(1, 4))
        pq = [qubits[int(x)] for x in rng.permutation(len(qubits))[:k]]
        bits = ''.join(rng.choice(list('01'), size=k))
        cut = len(c) // 2
        gl = list(c)
        try:
            proj = simulate(_Circuit(gl[:cut] + [Projection(state=bits, qubits=pq)] + gl[cut:]), initial_state=init,
                            optimize='evolution-hybridq', complex_type='complex128', compress=compress, simplify=False)
            out[f'c{i}_proj_psi'] = np.asarray(proj).reshape(-1)
        except Exception as e:  # noqa: BLE001 -- e.g. nothing survives the projection: recorded as absent
            print('projection case skipped:', repr(e))
        out[f'c{i}_proj_q'] = np.asarray([qubits.index(q) for q in pq], dtype=np.int32)
        out[f'c{i}_proj_bits'] = np.array(bits)
        out[f'c{i}_proj_cut'] = cut
        # the gate stream of the projection run (compress slides gates on other qubits ACROSS the FunctionalGate,
        # circuit/utils.py:630-648) and, with a Measure and a second Projection added, of simplify + compress: which
        # element sits where, matrices of the fused ones.  An entry with k = -1 is functional gate number fF{j}.
        from hybridq.circuit import utils as _utils
        from hybridq.gate import Measure
        from hybridq.gate import property as _pr

        def stream(gates_, tag, do_simplify, fns):
            # (simplify deep-copies what it inserts: the functional gates are told apart by name + qubits)
            fkey = [(f.name, tuple(f.qubits)) for f in fns]
            assert len(set(fkey)) == len(fkey)

            def which(g):
                return fkey.index((g.name, tuple(g.qubits)))
            cc_ = _Circuit(g for g in gates_ if g.name != 'I')
            if do_simplify:
                cc_ = _utils.simplify(cc_, remove_id_gates=True, atol=1e-8, verbose=False)
                out[f'{tag}_s_n'] = len(cc_)
                for j, g in enumerate(cc_):  # the simplified list: functional gates by number, matrix gates by qubits + matrix
                    fn = isinstance(g, _pr.FunctionalGate)
                    out[f'{tag}_sF{j}'] = which(g) if fn else -1
                    if not fn:
                        out[f'{tag}_sq{j}'] = np.asarray([qubits.index(q) for q in g.qubits], dtype=np.int32)
                        out[f'{tag}_sU{j}'] = np.asarray(g.matrix(), dtype=np.complex128)
            layers_ = _utils.compress(cc_, compress, verbose=False, skip_compression=[_pr.FunctionalGate])
            out[f'{tag}_f_n'] = len(layers_)
            for j, layer in enumerate(layers_):
                fn = any(isinstance(g, _pr.FunctionalGate) for g in layer)
                assert not fn or len(layer) == 1
                out[f'{tag}_fF{j}'] = which(layer[0]) if fn else -1
                if not fn:
                    mg = _utils.to_matrix_gate(layer, complex_type='complex128')
                    out[f'{tag}_fU{j}'] = np.asarray(mg.matrix())
                    out[f'{tag}_fq{j}'] = np.asarray([qubits.index(q) for q in mg.qubits], dtype=np.int32)

        P1 = Projection(state=bits, qubits=pq)
        stream(gl[:cut] + [P1] + gl[cut:], f'c{i}_pj', False, [P1])
        mq = [qubits[int(x)] for x in rng.permutation(len(qubits))[:2]]
        p2q = [[q for q in (qubits[int(x)] for x in rng.permutation(len(qubits))) if [q] != list(pq)][0]]
        out[f'c{i}_fn_mq'] = np.asarray([qubits.index(q) for q in mq], dtype=np.int32)
        out[f'c{i}_fn_p2q'] = np.asarray([qubits.index(q) for q in p2q], dtype=np.int32)
        c3 = len(gl) // 3
        M1, P2 = Measure(qubits=mq), Projection(state='1', qubits=p2q)
        stream(gl[:c3] + [P1] + gl[c3:2 * c3] + [M1] + gl[2 * c3:] + [P2], f'c{i}_fn', True, [P1, M1, P2])
        op = get_rqc(2, 3, indexes=[qubits[1], qubits[len(qubits) - 2]], use_random_indexes=False, use_unitary_only=False)
        for j, g in enumerate(op):
            out[f'c{i}_opU{j}'] = np.asarray(g.matrix(), dtype=np.complex128)
            out[f'c{i}_opq{j}'] = np.asarray([qubits.index(q) for q in g.qubits], dtype=np.int32)
        out[f'c{i}_op_n'] = len(op)
        psi128 = np.asarray(psi).astype('complex128')
        out[f'c{i}_ev'] = np.asarray(expectation_value(state=psi128, op=op, qubits_order=qubits, complex_type='complex128',
                                                       verbose=False), dtype=np.complex128)
        # the gate stream the reference's driver hands to its core for this call (simulation.py:289-305, :436-454)
        from hybridq.circuit import Circuit, utils
        from hybridq.gate import property as pr
        cc = Circuit(g for g in c if g.name != 'I')
        if simplify:
            cc = utils.simplify(cc, remove_id_gates=True, atol=1e-8, verbose=False)
        layers = utils.compress(cc, compress, verbose=False, skip_compression=[pr.FunctionalGate])
        fused = [utils.to_matrix_gate(layer, complex_type=ctype) for layer in layers]
        out[f'c{i}_f_n'] = len(fused)
        for j, g in enumerate(fused):
            out[f'c{i}_fU{j}'] = np.asarray(g.matrix())
            out[f'c{i}_fq{j}'] = np.asarray([qubits.index(q) for q in g.qubits], dtype=np.int32)
        out['n_cases'] = i + 1
    np.savez_compressed(path, **out)


def fn_stream_fixture():
    """python make_golden.py fn_streams: tests/golden/e2e_fn_streams.npz -- circuits with FunctionalGates (a Projection in the
    middle; a Projection, a Measure and a closing Projection) through the REFERENCE's simulate / simplify / compress, recorded
    by live_cases() and reduced to what travels: the input gates, the functional gates' qubits, the order of the resulting gate
    lists, and every resulting matrix as its product with a fixed probe vector (tests/fn_stream_checks.py: probe) -- 2^k numbers
    instead of 4^k -- plus the final state of the projection run for the smaller cases.  Cases: seed 4 (a width-6 layer whose
    gates commute only to 1e-5: to_matrix_gate's inner regrouping shows) and seed 5 (non-unitary gates around the projection)."""
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from fn_stream_checks import probe
    out = {}
    n_out = 0
    for seed in (4, 5):
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, 'live.npz')
            live_cases(path, seed)
            z = np.load(path, allow_pickle=False)
            # per seed: the (up to three) compress = 6 cases -- fused gates wider than 4 are where to_matrix_gate regroups -- and
            # one case at width 2 or 4
            cand = [i for i in range(int(z['n_cases'])) if f'c{i}_pj_f_n' in z.files and f'c{i}_proj_psi' in z.files]
            wide = [i for i in cand if int(z[f'c{i}_compress']) == 6][:3]
            narrow = [i for i in cand if int(z[f'c{i}_compress']) in (2, 4)][:1]
            picked = 0
            for i in sorted(wide + narrow):
                pre, dst = f'c{i}_', f'c{n_out}_'
                for key in ('n', 'n_gates', 'names', 'unitary', 'init', 'compress', 'proj_q', 'proj_bits', 'proj_cut', 'fn_mq', 'fn_p2q'):
                    out[dst + key] = z[pre + key]
                for j in range(int(z[pre + 'n_gates'])):
                    out[f'{dst}U{j}'], out[f'{dst}q{j}'] = z[f'{pre}U{j}'], z[f'{pre}q{j}']
                if int(z[pre + 'n']) <= 12:
                    out[dst + 'proj_psi'] = z[pre + 'proj_psi']
                for stag, kind in (('pj', 'f'), ('fn', 's'), ('fn', 'f')):
                    cnt = int(z[f'{pre}{stag}_{kind}_n'])
                    out[f'{dst}{stag}_{kind}_n'] = cnt
                    for j in range(cnt):
                        out[f'{dst}{stag}_{kind}F{j}'] = z[f'{pre}{stag}_{kind}F{j}']
                        if int(z[f'{pre}{stag}_{kind}F{j}']) < 0:
                            U = z[f'{pre}{stag}_{kind}U{j}']
                            out[f'{dst}{stag}_{kind}q{j}'] = z[f'{pre}{stag}_{kind}q{j}']
                            out[f'{dst}{stag}_{kind}Uv{j}'] = U @ probe(U.shape[0])
                out[dst + 'seed'] = seed
                n_out += 1
                picked += 1
    out['n_cases'] = n_out
    np.savez_compressed(os.path.join(HERE, 'e2e_fn_streams.npz'), **out)
    print('functional-gate stream fixture written:', n_out, 'cases')


def live_dm_cases(path, seed):
    """python make_golden.py live_dm OUT.npz SEED: random noisy circuits (depolarizing / dephasing / amplitude-damping noise
    from hybridq.noise.utils) through the reference's dm simulate(), as data like dm_circuit() records them."""
    install_stubs()
    sys.path.insert(0, REF)
    import hybridq.dm.circuit.simulation as dmsim
    from hybridq.extras.random import get_rqc
    from hybridq.noise.utils import add_amplitude_damping_noise, add_dephasing_noise, add_depolarizing_noise
    rng = np.random.default_rng(seed)
    np.random.seed(seed)
    out = {'n_cases': 0}
    for i in range(4):
        nq = int(rng.integers(6, 8))  # 2 nq > 10 qubits: below that the reference leaves its C++ core for einsum (:401-404)
        circ = get_rqc(nq, int(rng.integers(16, 28)), use_random_indexes=False)
        while len(circ.all_qubits()) != nq:  # every qubit in use
            circ = get_rqc(nq, int(rng.integers(16, 28)), use_random_indexes=False)
        kind = ['depolarizing', 'dephasing', 'damping', 'depolarizing'][i]
        if kind == 'depolarizing':
            noisy = add_depolarizing_noise(circ, probs=(float(rng.uniform(0.005, 0.05)), float(rng.uniform(0.01, 0.08))))
        elif kind == 'dephasing':
            noisy = add_dephasing_noise(circ, probs=float(rng.uniform(0.01, 0.1)), pauli_indexes=int(rng.integers(1, 4)))
        else:
            noisy = add_amplitude_damping_noise(circ, gammas=float(rng.uniform(0.01, 0.1)), probs=float(rng.uniform(0.0, 0.3)))
        init = ''.join(rng.choice(list('01+-'), size=len(noisy.all_qubits()[0])))
        kinds = ''
        for j, g in enumerate(noisy):
            out[f'c{i}_q{j}'] = np.asarray([int(q) for q in g.qubits], dtype=np.int32)
            if hasattr(g, 'Kraus'):
                kinds += 'K'
                L, R = g.Kraus.gates
                out[f'c{i}_L{j}'] = np.stack([np.asarray(x.matrix(), dtype=np.complex128) for x in L])
                out[f'c{i}_R{j}'] = np.stack([np.asarray(x.matrix(), dtype=np.complex128) for x in R])
                out[f'c{i}_s{j}'] = np.asarray(g.Kraus.s, dtype=np.float64)
            else:
                kinds += 'U'
                out[f'c{i}_U{j}'] = np.asarray(g.matrix(), dtype=np.complex128)
        out[f'c{i}_kinds'] = np.frombuffer(kinds.encode(), dtype=np.uint8)
        out[f'c{i}_init'] = np.array(init)
        out[f'c{i}_kind'] = np.array(kind)
        rho = dmsim.simulate(noisy, initial_state=init, optimize='evolution-hybridq', complex_type='complex128', verbose=False)
        out[f'c{i}_rho'] = np.asarray(rho).reshape(-1)
        out['n_cases'] = i + 1
    np.savez_compressed(path, **out)


def live_qasm_write(path, seed):
    """python make_golden.py live_qasm_write OUT.npz SEED: a random circuit over EVERY named gate of the reference
    (hybridq/gate/gate.py:127-350), with parameters, powers, conj / T and MATRIX gates, written by the reference's to_qasm;
    the text and every gate's matrix and qubits."""
    install_stubs()
    sys.path.insert(0, REF)
    from hybridq.circuit import Circuit
    from hybridq.extras.io.qasm import to_qasm
    from hybridq.gate import Gate
    rng = np.random.default_rng(seed)
    one = ['I', 'H', 'X', 'Y', 'Z', 'P', 'T', 'SQRT_X', 'SQRT_Y']
    two = ['ZZ', 'CZ', 'CX', 'SWAP', 'ISWAP', 'SQRT_SWAP', 'SQRT_ISWAP']
    par = {'RX': (1, 1), 'RY': (1, 1), 'RZ': (1, 1), 'R_PI_2': (1, 1), 'U3': (1, 3), 'CPHASE': (2, 1), 'FSIM': (2, 2)}
    labels = [3, 17, 42, 5, 8, 100]
    gates = []
    for rep in range(3):
        for name in one + two + list(par):
            k, npar = (1, 0) if name in one else (2, 0) if name in two else par[name]
            qs = [labels[int(x)] for x in rng.permutation(len(labels))[:k]]
            g = Gate(name, qubits=qs, params=[float(x) for x in rng.uniform(-3, 3, size=npar)]) if npar else Gate(name, qubits=qs)
            what = int(rng.integers(0, 5)) if rep else 0
            if what == 1:
                g = g**float(rng.choice([2, 3, -1, 0.5, 1.23, -0.7]))
            elif what == 2:
                g = g.conj()
            elif what == 3:
                g = g.T()
            elif what == 4:
                g = (g**float(rng.choice([2, 0.37]))).conj().T()
            gates.append(g)
    for k in (1, 2, 3):
        M = rng.standard_normal((1 << k, 1 << k)) + 1j * rng.standard_normal((1 << k, 1 << k))
        gates.append(Gate('MATRIX', qubits=[labels[int(x)] for x in rng.permutation(len(labels))[:k]], U=M))
    c = Circuit(gates)
    text = to_qasm(c)
    out = {'text': np.frombuffer(text.encode(), dtype=np.uint8), 'n_gates': len(c)}
    for i, g in enumerate(c):
        out[f'U{i}'] = np.asarray(g.matrix(), dtype=np.complex128)
        out[f'q{i}'] = np.asarray([int(q) for q in g.qubits], dtype=np.int64)
        out[f'name{i}'] = np.array(g.name)
    np.savez_compressed(path, **out)


def live_qasm_read(text_path, path):
    """python make_golden.py live_qasm_read IN.txt OUT.npz: the reference's from_qasm on a text THIS package wrote."""
    install_stubs()
    sys.path.insert(0, REF)
    from hybridq.extras.io.qasm import from_qasm
    c = from_qasm(open(text_path).read())
    out = {'n_gates': len(c)}
    for i, g in enumerate(c):
        out[f'U{i}'] = np.asarray(g.matrix(), dtype=np.complex128)
        out[f'q{i}'] = np.array([str(q) for q in g.qubits])
    np.savez_compressed(path, **out)


def live_named_cases(path, seed):
    """python make_golden.py live_named OUT.npz SEED: circuits of NAMED gates -- diagonal ones that commute (Z, T, P, CZ,
    RZ, CPHASE, ZZ), planted inverse pairs and identities -- through the reference's simplify / compress under random
    option dictionaries (use_matrix_commutation, max_n_qubits_matrix, exclude_qubits), and through its simulate()."""
    install_stubs()
    sys.path.insert(0, REF)
    from hybridq.circuit import Circuit, utils
    from hybridq.circuit.simulation import simulate
    from hybridq.gate import Gate
    from hybridq.gate import property as pr
    rng = np.random.default_rng(seed)
    out = {'n_cases': 0}
    n = 12
    one = ['H', 'X', 'Z', 'T', 'P', 'SQRT_X', 'I']
    two = ['CZ', 'CX', 'ZZ', 'ISWAP']
    for i in range(6):
        gates = []
        while len(gates) < 90:
            r = rng.random()
            if r < 0.45:
                gates.append(Gate(str(rng.choice(one)), qubits=[int(rng.integers(0, n))]))
            elif r < 0.7:
                a, b = (int(x) for x in rng.permutation(n)[:2])
                gates.append(Gate(str(rng.choice(two)), qubits=[a, b]))
            elif r < 0.8:
                gates.append(Gate('RZ', qubits=[int(rng.integers(0, n))], params=[float(rng.uniform(-3, 3))]))
            elif r < 0.88:
                a, b = (int(x) for x in rng.permutation(n)[:2])
                gates.append(Gate('CPHASE', qubits=[a, b], params=[float(rng.uniform(-3, 3))]))
            elif r < 0.92:  # a random diagonal gate on three qubits: commutes with every other diagonal gate
                qs = [int(x) for x in rng.permutation(n)[:3]]
                gates.append(Gate('MATRIX', qubits=qs, U=np.diag(np.exp(1j * rng.uniform(-3, 3, size=8)))))
            else:  # an inverse pair with something commuting (or not) in between
                a, b = (int(x) for x in rng.permutation(n)[:2])
                g = Gate(str(rng.choice(['CX', 'ISWAP', 'CZ'])), qubits=[a, b])
                mid = Gate(str(rng.choice(['Z', 'T', 'H'])), qubits=[int(rng.integers(0, n))])
                gates += [g, mid, g.inv()]
        for q in range(n):  # every qubit in use
            gates.append(Gate('H', qubits=[q]))
        c = Circuit(gates)
        qubits = c.all_qubits()
        assert qubits == list(range(n))
        simp = {'use_matrix_commutation': bool(rng.integers(0, 2))}
        if rng.random() < 0.5:
            simp['max_n_qubits_matrix'] = int(rng.choice([1, 2, 10]))
        comp = {'max_n_qubits': int(rng.choice([2, 3, 4, 5, 6])), 'use_matrix_commutation': bool(rng.integers(0, 2)),
                'max_n_qubits_matrix': int(rng.choice([1, 2, 3, 4, 10]))}
        if rng.random() < 0.5:
            comp['exclude_qubits'] = [int(x) for x in rng.permutation(n)[:2]]
        init = ''.join(rng.choice(list('01+-'), size=n))
        out[f'c{i}_n_gates'] = len(c)
        for j, g in enumerate(c):
            out[f'c{i}_U{j}'] = np.asarray(g.matrix(), dtype=np.complex128)
            out[f'c{i}_q{j}'] = np.asarray([int(q) for q in g.qubits], dtype=np.int32)
        out[f'c{i}_names'] = np.array([g.name for g in c])
        out[f'c{i}_init'] = np.array(init)
        out[f'c{i}_simp_umc'] = simp['use_matrix_commutation']
        out[f'c{i}_simp_mnm'] = simp.get('max_n_qubits_matrix', -1)
        out[f'c{i}_comp_n'] = comp['max_n_qubits']
        out[f'c{i}_comp_umc'] = comp['use_matrix_commutation']
        out[f'c{i}_comp_mnm'] = comp['max_n_qubits_matrix']
        out[f'c{i}_comp_excl'] = np.asarray(comp.get('exclude_qubits', []), dtype=np.int32)
        cc = Circuit(g for g in c if g.name != 'I')
        cc = utils.simplify(cc, remove_id_gates=True, atol=1e-8, verbose=False, **simp)
        out[f'c{i}_s_n'] = len(cc)
        for j, g in enumerate(cc):
            out[f'c{i}_sU{j}'] = np.asarray(g.matrix(), dtype=np.complex128)
            out[f'c{i}_sq{j}'] = np.asarray([int(q) for q in g.qubits], dtype=np.int32)
        layers = utils.compress(cc, comp['max_n_qubits'], verbose=False, skip_compression=[pr.FunctionalGate],
                                **{k: v for k, v in comp.items() if k != 'max_n_qubits'})
        fused = [utils.to_matrix_gate(layer, complex_type='complex128') for layer in layers]
        out[f'c{i}_f_n'] = len(fused)
        for j, g in enumerate(fused):
            out[f'c{i}_fU{j}'] = np.asarray(g.matrix())
            out[f'c{i}_fq{j}'] = np.asarray([int(q) for q in g.qubits], dtype=np.int32)
        psi = simulate(c, initial_state=init, optimize='evolution-hybridq', complex_type='complex128', compress=comp, simplify=simp)
        out[f'c{i}_psi'] = np.asarray(psi).reshape(-1)
        out['n_cases'] = i + 1
    np.savez_compressed(path, **out)


def qasm_vectors():
    """e2e_qasm_ext.npz: a circuit with string / tuple-free labels, powers, conj / T and a MATRIX gate written
    by the reference's to_qasm (hybridq/extras/io/qasm.py:160) -- the text it produced (output data) and every
    gate's matrix and qubits, for the reader's ``#@`` extension blocks."""
    install_stubs()
    sys.path.insert(0, REF)
    from hybridq.circuit import Circuit
    from hybridq.extras.io.qasm import to_qasm
    from hybridq.gate import Gate
    np.random.seed(1234)
    rnd = np.random.standard_normal((4, 4)) + 1j * np.random.standard_normal((4, 4))
    U, _ = np.linalg.qr(rnd)
    c = Circuit([
        Gate('H', qubits=[42]),
        Gate('RZ', qubits=[7], params=[0.5])**1.23,
        Gate('CX', qubits=[42, 7]).conj(),
        Gate('ISWAP', qubits=[7, 3]).T(),
        Gate('SQRT_X', qubits=[3], tags={'a': 1})**2,
        Gate('MATRIX', qubits=[3, 42], U=U),
        Gate('MATRIX', qubits=[7], U=U[:2, :2] * 0.5)**0.5,
        Gate('CPHASE', qubits=[3, 7], params=[0.25]).conj().T(),
        Gate('U3', qubits=[42], params=[0.1, 0.2, 0.3]),
    ])
    text = to_qasm(c)
    out = {'text': np.frombuffer(text.encode(), dtype=np.uint8), 'n_gates': len(c)}
    for i, g in enumerate(c):
        out[f'U{i}'] = np.asarray(g.matrix(), dtype=np.complex128)
        out[f'q{i}'] = np.asarray([int(q) for q in g.qubits], dtype=np.int32)
    np.savez_compressed(os.path.join(HERE, 'e2e_qasm_ext.npz'), **out)
    print(text)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'qasm':
        qasm_vectors()
        raise SystemExit(0)
    if len(sys.argv) > 3 and sys.argv[1] == 'live_qasm_write':
        live_qasm_write(sys.argv[2], int(sys.argv[3]))
        raise SystemExit(0)
    if len(sys.argv) > 3 and sys.argv[1] == 'live_qasm_read':
        live_qasm_read(sys.argv[2], sys.argv[3])
        raise SystemExit(0)
    if len(sys.argv) > 3 and sys.argv[1] == 'live_named':
        live_named_cases(sys.argv[2], int(sys.argv[3]))
        raise SystemExit(0)
    if len(sys.argv) > 3 and sys.argv[1] == 'live_dm':
        live_dm_cases(sys.argv[2], int(sys.argv[3]))
        raise SystemExit(0)
    if len(sys.argv) > 3 and sys.argv[1] == 'live':
        live_cases(sys.argv[2], int(sys.argv[3]))
        raise SystemExit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'fn_streams':
        fn_stream_fixture()
        raise SystemExit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'containers':
        container_vectors()
        raise SystemExit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'matrix':
        matrix_vectors()
        raise SystemExit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'dm_circuit':
        dm_circuit()
        raise SystemExit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'api':
        api_vectors()
        raise SystemExit(0)
    if os.path.exists('hybridq.so'):
        raise SystemExit('run from a directory that does not contain hybridq.so')
    per_call_vectors()
    end_to_end()
