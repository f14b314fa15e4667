"""Pin the CPU oracle on the golden vectors recorded from the reference (compiled C++
core for per-call vectors; reference Python driver for the end-to-end traces)."""
import numpy as np
import pytest

import oracle
from oracle.binding import aligned_empty

import golden_util as gu


def _tol(dt):
    return 2e-6 if np.dtype(dt) in (np.dtype('float32'), np.dtype('complex64')) else 1e-13


def test_port_apply_U_vs_reference_vectors(oracle_port):
    n_cases = 0
    for inp, out, U, pos in gu.apply_cases():
        pl = aligned_empty(inp.shape, inp.dtype)
        pl[:] = inp
        assert oracle_port.apply_U(pl[0], pl[1], U, pos) == 0
        err = np.abs(pl - out).max() / np.abs(out).max()
        assert err < _tol(inp.dtype) * len(U), (n_cases, err)
        n_cases += 1
    assert n_cases == 36


def test_port_swap_vs_reference_vectors(oracle_port):
    for n, pos, out in gu.swap_cases():
        for dt in (np.float32, np.float64, np.int32, np.int64, np.uint32, np.uint64):
            a = np.arange(1 << n).astype(dt)
            assert oracle_port.swap(a, pos) == 0
            assert (a == out.astype(dt)).all(), (dt, list(pos))
        assert (oracle.swap_numpy(np.arange(1 << n), pos) == out).all()


def _replay_on(lib, z, prefix, n, ft):
    pl = aligned_empty((2, 1 << n), ft)
    pl[:] = 0
    pl[0, 0] = 1
    gu.replay(z, prefix, n,
              lambda U, pos: lib.apply_U(pl[0], pl[1], U, pos) and pytest.fail('apply_U failed'),
              lambda pos: (lib.swap(pl[0], pos), lib.swap(pl[1], pos)))
    return pl[0] + 1j * pl[1]


def test_simple_qasm_trace_and_circuit(oracle_port):
    """BASELINE cfg1: examples/circuit_simple.qasm, n=24, initial '0'*24, complex64."""
    z = gu.load('e2e_simple_qasm.npz')
    n = int(z['n_qubits'])
    assert n == 24 and bytes(z['trace_kinds']).decode().count('U') == 13
    stride = int(z['sample_stride'])
    # (a) replay the reference's own C-ABI call sequence (fused k=4 gates + swaps)
    psi = _replay_on(oracle_port, z, 'trace_', n, np.float32)
    scale = np.abs(z['psi_sample']).max()
    assert np.abs(psi[::stride] - z['psi_sample']).max() / scale < 5e-6
    assert np.abs(psi[:8] - z['psi_head']).max() / scale < 5e-6
    # (b) the 99 unfused gates through the reference driver protocol
    gates = gu.simple_qasm_gates(z)
    assert len(gates) == 99
    psi2, _ = oracle.evolve_reference_protocol(oracle_port, gates, n, complex_type='complex64')
    assert np.abs(psi2[::stride] - z['psi_sample']).max() / scale < 1e-5
    assert abs(float((np.abs(psi2.astype(np.complex128))**2).sum()) - 1.0) < 1e-5  # f64 accumulation
    # the matrices this repo transcribed for the named gates agree with the reference's
    from hybridq_amd.circuits import GATE_MATRICES
    for nm in z['matrix_names']:
        key = str(nm).lower()
        key = {'sqrt_x': 'x_1_2', 'sqrt_y': 'y_1_2'}.get(key, key)
        assert np.allclose(GATE_MATRICES[key], z['matrix_' + str(nm)], atol=1e-12), nm


@pytest.mark.parametrize('tag,ct', [('a', 'complex64'), ('b', 'complex128')])
def test_reference_rqc(oracle_port, tag, ct):
    z = gu.load('e2e_rqc.npz')
    n = int(z['n_qubits'])
    gates = gu.rqc_gates(z, tag)
    exp = z[f'{tag}_psi']
    tol = 5e-6 if ct == 'complex64' else 1e-12
    psi, _ = oracle.evolve_reference_protocol(oracle_port, gates, n, complex_type=ct)
    assert np.abs(psi - exp).max() / np.abs(exp).max() < tol
    psi_t = oracle.evolve_tensordot(gates, n)
    assert np.abs(psi_t - exp).max() / np.abs(exp).max() < tol
    ft = np.float32 if ct == 'complex64' else np.float64
    psi_r = _replay_on(oracle_port, z, f'{tag}_trace_', n, ft)
    assert np.abs(psi_r - exp).max() / np.abs(exp).max() < tol


def test_dm_trace(oracle_port):
    """BASELINE cfg5 shape: noisy 6-qubit circuit = 12-qubit state vector, non-unitary U."""
    z = gu.load('e2e_dm.npz')
    n = int(z['n_qubits'])
    rho = _replay_on(oracle_port, z, 'trace_', n, np.float32)
    exp = z['rho']
    assert np.abs(rho - exp).max() / np.abs(exp).max() < 5e-6
    r = rho.reshape(1 << (n // 2), 1 << (n // 2))
    assert abs(np.trace(r).real - 1) < 1e-5 and np.abs(r - r.conj().T).max() < 1e-6


def test_fusion_reproduces_reference_compress():
    """hybridq_amd.fusion.fuse == utils.compress + to_matrix_gate of the reference on the same
    gate order: the fused matrices equal the ones the reference handed to apply_U (recorded
    trace of get_rqc with compress=4, simplify=False), one for one."""
    from hybridq_amd.fusion import compress, fuse, to_matrix_gate
    z = gu.load('e2e_rqc.npz')
    gates = gu.rqc_gates(z, 'a')
    fused = fuse(gates, 4)
    rec = [U for kind, pos, U in gu.trace(z, 'a_trace_') if kind == 'U']
    assert len(fused) == len(rec) == 11
    for (U, qs), Ur in zip(fused, rec):
        assert U.shape == Ur.shape and np.abs(U - Ur).max() < 5e-7
        assert list(qs) == sorted(qs)
    # compress() layers + to_matrix_gate() agree with fuse(); compress=0 splits every gate
    layers = compress(gates, 4)
    assert sum(len(L) for L in layers) == len(gates)
    for L, (U, qs) in zip(layers, fused):
        U2, q2 = to_matrix_gate(L)
        assert q2 == qs and np.abs(U2 - U).max() < 1e-6
    assert len(fuse(gates, 0)) == len(gates)
    # fused circuit == unfused circuit
    n = int(z['n_qubits'])
    a = oracle.evolve_tensordot(gates, n)
    b = oracle.evolve_tensordot([(U.astype(np.complex128), qs) for U, qs in fuse(gates, 4, complex_type='complex128')], n)
    assert np.abs(a - b).max() < 1e-12
    for kmax in (2, 3, 5):
        f = fuse(gates, kmax, complex_type='complex128')
        assert max(len(qs) for _, qs in f) <= max(kmax, max(len(qs) for _, qs in gates))
        b = oracle.evolve_tensordot(f, n, qubits=list(range(n)))
        assert np.abs(a - b).max() < 1e-12


def test_dm_front_end_builds_reference_superoperators():
    """hybridq_amd.dm: Kraus(...).map() equals the matrix the reference built for every
    channel of the recorded noisy circuit, and the 2n-qubit rewrite reproduces rho."""
    from hybridq_amd.dm import Kraus, depolarizing, to_statevector_circuit
    z = gu.load('e2e_dm_circuit.npz')
    kinds = bytes(z['kinds']).decode()
    n = int(z['n_qubits'])
    circuit = []
    for i, kind in enumerate(kinds):
        qs = tuple(int(q) for q in z[f'q{i}'])
        if kind == 'K':
            ch = Kraus(list(z[f'L{i}']), qs, s=z[f's{i}'], right_ops=list(z[f'R{i}']))
            assert np.abs(ch.map() - z[f'M{i}']).max() < 1e-14
            p = 0.01 if len(qs) == 1 else 0.02
            assert np.abs(depolarizing(qs, p).map() - z[f'M{i}']).max() < 1e-12
            circuit.append(ch)
        else:
            circuit.append((z[f'U{i}'], qs))
    sv = to_statevector_circuit(circuit)
    labels = [(0, q) for q in range(n)] + [(1, q) for q in range(n)]
    gates = [(U, tuple(labels.index(q) for q in qs)) for U, qs in sv]
    rho = oracle.evolve_tensordot(gates, 2 * n)
    assert np.abs(rho - z['rho']).max() / np.abs(z['rho']).max() < 5e-6


def test_qasm_reader_matches_reference_circuit():
    """hybridq_amd.qasm on the text of the reference example == the gate list the reference's
    own parser produced (names, qubits, matrices recorded in e2e_simple_qasm.npz).  The QASM
    text is regenerated from that record (the example file itself is not in this repo)."""
    from hybridq_amd.qasm import from_qasm
    z = gu.load('e2e_simple_qasm.npz')
    exp = gu.simple_qasm_gates(z)
    back = {'SQRT_X': 'x_1_2', 'SQRT_Y': 'y_1_2'}
    lines = ['# regenerated', '']
    for nm, qs in zip(z['gate_names'], z['gate_qubits']):
        nm = back.get(str(nm), str(nm).lower())
        lines.append(nm + ' ' + ' '.join(str(int(q)) for q in qs if q >= 0))
    got = from_qasm('\n'.join(lines))
    assert len(got) == len(exp) == 99
    for (U, qs), (Ue, qe) in zip(got, exp):
        assert qs == qe and np.abs(U - Ue).max() < 1e-12
    g = from_qasm('3\nrx 0 0.3\ncphase 0 2 1.1\nid 1\ncnot 1 2\ns 2\n')
    assert [len(q) for _, q in g] == [1, 2, 1, 2, 1] and np.allclose(g[4][0], np.diag([1, 1j]))
    with pytest.raises(ValueError):
        from_qasm('foo 1')


def test_api_golden_oracle_side(oracle_port):
    """e2e_api.npz (reference host-level API outputs): the oracle's '01+-' initial states are the
    reference's prepare_state, and the oracle's evolution reproduces the reference's simulate()
    with a mixed initial state and non-unitary gates (complex128 through compress=8)."""
    from oracle.evolution import _initial
    z = gu.load('e2e_api.npz')
    for i, st in enumerate(str(x) for x in z['ps_strings']):
        exp = z[f'ps_{i}']
        n = int(np.log2(exp.size))
        got = _initial(st, n, np.complex128)
        assert np.abs(got - exp).max() < 1e-15, st
    gates = gu.rqc_gates(z, 'sim')
    init = str(z['sim_init'])
    n = len(init)
    psi = oracle.evolve_tensordot(gates, n, initial_state=init)
    assert np.abs(psi - z['sim_psi128']).max() / np.abs(psi).max() < 1e-12
    got, _ = oracle.evolve_reference_protocol(oracle_port, gates, n, initial_state=init, complex_type='complex64')
    assert np.abs(got - z['sim_psi64']).max() / np.abs(psi).max() < 5e-6
    # dot(): the recorded results are the plain tensor contraction; swap_back=False + tr is consistent
    for j in range(int(z['dot_cases'])):
        n = int(z['dot_n'])
        psi, U, axes = z[f'dot{j}_psi'], z[f'dot{j}_U'], [int(a) for a in z[f'dot{j}_axes']]
        k = len(axes)
        c = (psi[0] + 1j * psi[1]).astype(np.complex128).reshape((2,) * n)
        exp = np.moveaxis(np.tensordot(U.astype(np.complex128).reshape((2,) * (2 * k)), c,
                                       axes=(list(range(k, 2 * k)), axes)), list(range(k)), axes).reshape(-1)
        tol = 1e-5 if psi.dtype == np.float32 else 1e-13
        scale = np.abs(exp).max()
        res = z[f'dot{j}_res']
        assert np.abs(res[0] + 1j * res[1] - exp).max() / scale < tol * 2**k
        assert np.abs(z[f'dot{j}_res_complex'] - exp).max() / scale < tol * 2**k
        tr = [int(a) for a in z[f'dot{j}_tr']]
        ns = z[f'dot{j}_noswap']
        ns = (ns[0] + 1j * ns[1]).reshape((2,) * n)
        if tr != [-1]:
            ns = np.transpose(ns, tr)
        assert np.abs(ns.reshape(-1) - (res[0] + 1j * res[1])).max() == 0


@pytest.mark.parametrize('ct', ['complex64', 'complex128'])
def test_blocked_planner_preserves_the_circuit(oracle_port, ct):
    """hybridq_amd.blocking.plan_blocked (host side of the cache-blocked path) on the CPU: executing
    its ops in order -- every inner gate of a 'B' pass, every plain 'G' gate -- with the oracle's
    apply_U gives the state of the original circuit; tiles contain the low bits and every target of
    their gates; options (tile size, inner fusion width, no fusion) keep that true."""
    from hybridq_amd.blocking import blocked_stats, plan_blocked
    from hybridq_amd.circuits import random_dense, rqc_1q2q
    ft = np.dtype('float32') if ct == 'complex64' else np.dtype('float64')
    for n, gates in ((14, rqc_1q2q(14, depth=10, seed=1)), (16, rqc_1q2q(16, depth=6, seed=2) + random_dense(16, 30, kmax=6, seed=3))):
        exp = oracle.evolve_tensordot(gates, n, qubits=list(range(n)))
        pos_of = {q: n - 1 - q for q in range(n)}
        for opts in (dict(), dict(tile_bits=12, low_bits=4, inner_max=4), dict(tile_bits=10, inner_max=0), dict(tries=1, min_gates=1)):
            ops = plan_blocked(gates, pos_of, n, complex_type=ct, **opts)
            pl = aligned_empty((2, 1 << n), ft)
            pl[:] = 0
            pl[0, 0] = 1
            n_inner = 0
            for op in ops:
                if op[0] == 'B':
                    tile = set(int(p) for p in op[1])
                    assert {0, 1} <= tile and len(tile) == min(opts.get('tile_bits', 13), n)
                    for U, pos in op[2]:
                        assert set(int(p) for p in pos) <= tile and len(pos) <= max(4, opts.get('inner_max', 3))
                        assert oracle_port.apply_U(pl[0], pl[1], np.ascontiguousarray(U, dtype=ct), pos) == 0
                        n_inner += 1
                else:
                    assert oracle_port.apply_U(pl[0], pl[1], np.ascontiguousarray(op[1], dtype=ct), op[2]) == 0
            psi = pl[0] + 1j * pl[1]
            tol = 5e-6 if ct == 'complex64' else 1e-12
            assert np.abs(psi - exp).max() / np.abs(exp).max() < tol, (n, opts)
            st = blocked_stats(ops)
            assert st['inner_gates'] == n_inner
            if opts.get('inner_max', 3) == 0:  # no algebraic fusion: every gate of the circuit appears once
                assert st['inner_gates'] + st['plain_gates'] == len(gates)


def test_driver_planning_host_logic():
    """simulation._plan_ops (pure host code of the driver): fusion width, blocking and the cut at
    FunctionalGates (never fused: simulation.py:441 skip_compression=[FunctionalGate])."""
    from hybridq_amd.circuits import rqc_1q2q
    from hybridq_amd.fusion import fuse
    from hybridq_amd.simulation import FunctionalGate, _plan_ops, all_qubits
    n = 16
    g = rqc_1q2q(n, depth=6, seed=5)
    qubits = all_qubits(g)
    assert qubits == list(range(n))
    ct = np.dtype('complex64')
    plain = _plan_ops(g, qubits, n, ct, 0, False)
    assert len(plain) == len(g) and all(len(qs) <= 2 for qs, _ in plain)
    f4 = _plan_ops(g, qubits, n, ct, 4, False)
    assert len(f4) == len(fuse(g, 4)) < len(g) / 3 and max(len(qs) for qs, _ in f4) <= 4
    f5 = _plan_ops(g, qubits, n, ct, {'max_n_qubits': 5}, False)
    assert len(f5) <= len(f4) and max(len(qs) for qs, _ in f5) == 5
    fg = FunctionalGate((3,), lambda psi, order: (psi, order))
    cut = _plan_ops(g[:40] + [fg] + g[40:], qubits, n, ct, 4, False)
    k = [i for i, op in enumerate(cut) if op is fg]
    assert len(k) == 1
    # gates on other qubits slide ACROSS the functional gate into earlier fused gates, as in the reference's walk
    # (circuit/utils.py:636-637); gates on its qubit stay on their side of it
    assert len(fuse(g, 4)) <= len(cut) - 1 <= len(fuse(g[:40], 4)) + len(fuse(g[40:], 4))
    import oracle
    full = _plan_ops(g[:40] + g[40:], qubits, n, ct, 0, False)
    a = oracle.evolve_tensordot([(U, qs) for qs, U in cut[:k[0]]] + [(U, qs) for qs, U in cut[k[0] + 1:]], n, qubits=qubits)
    b = oracle.evolve_tensordot([(U, qs) for qs, U in full], n, qubits=qubits)
    assert np.abs(a - b).max() / np.abs(b).max() < 1e-5  # (an identity functional gate: the circuit's action is unchanged)
    c0 = _plan_ops(g[:40] + [fg] + g[40:], qubits, n, ct, 0, False)  # compress = 0: nothing moves
    assert c0.index(fg) == 40 and len(c0) == len(g) + 1
    blk = _plan_ops(g[:40] + [fg] + g[40:], qubits, n, ct, 4, True)
    kinds = [op[0] if isinstance(op, tuple) and isinstance(op[0], str) else 'F' for op in blk]
    assert kinds.count('F') == 1 and 'B' in kinds[:kinds.index('F')] and 'B' in kinds[kinds.index('F') + 1:]
    # blocking needs n >= 14: below that the fused stream is used
    small = rqc_1q2q(12, depth=4, seed=6)
    assert all(not isinstance(op[0], str) for op in _plan_ops(small, list(range(12)), 12, ct, 4, True))


def test_matrix_and_layers_match_reference():
    """e2e_matrix.npz: fusion.matrix == utils.matrix (default and permuted order), and
    compress(max_n_qubits=3) + to_matrix_gate give the reference's layers one for one."""
    from hybridq_amd.fusion import compress, matrix, to_matrix_gate
    z = gu.load('e2e_matrix.npz')
    for tag in ('a', 'b'):
        gates = gu.rqc_gates(z, tag)
        assert np.abs(matrix(gates, complex_type='complex128') - z[f'{tag}_matrix']).max() < 1e-12
        order = [int(q) for q in z[f'{tag}_order']]
        assert np.abs(matrix(gates, order=order, complex_type='complex128') - z[f'{tag}_matrix_order']).max() < 1e-12
        with pytest.raises(ValueError):
            matrix(gates, order=order[:-1] + [99])
        layers = compress(gates, max_n_qubits=3)
        assert len(layers) == int(z[f'{tag}_n_layers'])
        for j, layer in enumerate(layers):
            U, qs = to_matrix_gate(layer, complex_type='complex128')
            assert list(qs) == [int(q) for q in z[f'{tag}_layer{j}_qubits']]
            assert np.abs(U - z[f'{tag}_layer{j}_matrix']).max() < 1e-12


def test_simplify_matches_reference():
    """fusion.simplify == utils.simplify on a circuit with planted identity gates, inverse pairs
    behind commuting gates and ordinary gates: same gate list (qubits and matrices, in order),
    and the circuit's unitary is unchanged."""
    from hybridq_amd.fusion import matrix, simplify
    z = gu.load('e2e_matrix.npz')
    gates = gu.rqc_gates(z, 's')
    exp = [(z[f's_simplified_U{i}'], tuple(int(q) for q in z[f's_simplified_q{i}'])) for i in range(int(z['s_simplified_n']))]
    got = simplify(gates)
    assert len(got) == len(exp) < len(gates)
    for (U, qs), (V, vs) in zip(got, exp):
        assert tuple(qs) == tuple(vs)
        assert np.abs(np.asarray(U) - V).max() < 1e-12
    order = list(range(5))
    assert np.abs(matrix(got, order=order, complex_type='complex128') - matrix(gates, order=order, complex_type='complex128')).max() < 1e-10
    # without identity removal the identities stay; without matrix commutation fewer pairs cancel
    assert len(simplify(gates, remove_id_gates=False)) > len(got)
    assert len(simplify(gates, use_matrix_commutation=False)) >= len(got)
