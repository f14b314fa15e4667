"""The host side of ``simulate()`` end to end on a box without a GPU: argument handling, container / stochastic gates,
simplification, schedule choice, fusion, the cache-blocked planner and the gate loop's position arithmetic run for real;
only the device is replaced -- by a TEST DOUBLE that keeps the state in numpy and applies every ``apply_U`` /
``apply_blocked`` call with the oracle's index arithmetic (oracle.evolution.apply_gate_numpy).  What is checked is that
the calls the driver ISSUES evolve the state like an independent float64 evolution of the circuit as given.  The product
itself has no such path: without the HIP library's device ``simulate()`` raises (tests/test_abi.py)."""
import os
import sys

import numpy as np
import pytest

# the `numpy_device` fixture (the test double of the device) lives in conftest.py

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _rel(a, b):
    return np.abs(a - b).max() / np.abs(b).max()


@pytest.mark.parametrize('kw', [dict(compress=0), dict(compress=4), dict(compress=6), dict(blocked=True),
                                dict(blocked={'tile_bits': 12, 'low_bits': 4, 'inner_max': 4}), dict(optimize='evolution-hip'),
                                dict(optimize='evolution-hybridq'), dict(compress={'max_n_qubits': 3, 'exclude_qubits': [0, 5]})],
                         ids=lambda kw: '-'.join(f'{k}={v}' for k, v in kw.items())[:40])
def test_issued_calls_evolve_the_state(numpy_device, kw):
    log, oracle = numpy_device
    from hybridq_amd.circuits import random_dense, rqc_1q2q
    from hybridq_amd.simulation import simulate
    n = 16
    for gates, init in ((rqc_1q2q(n, depth=12, seed=3), '0' * n), (random_dense(n, 60, kmax=4, seed=4, unitary=True), '01+-' * 4)):
        psi, info = simulate(gates, initial_state=init, complex_type='complex128', qubits=list(range(n)), return_info=True, **kw)
        exp = oracle.evolve_tensordot(gates, n, initial_state=init, qubits=list(range(n)))
        assert psi.shape == (2,) * n and _rel(psi.reshape(-1), exp) < 1e-12, kw
        assert info['n_gates_given'] == len(gates) and info['n_qubits'] == n and info['n_passes'] <= len(gates)
        if kw.get('blocked'):
            assert log['apply_blocked'] > 0 and info['n_passes'] < len(gates) / 2
        if kw.get('compress') == 0:
            assert info['n_passes'] == len(gates)


def test_labels_functional_and_container_gates(numpy_device):
    """Heterogeneous qubit labels (sorted like Circuit.all_qubits), a FunctionalGate between fused runs, container gates,
    an array as initial state."""
    log, oracle = numpy_device
    from hybridq_amd.circuits import random_dense
    from hybridq_amd.simulation import FunctionalGate, simulate
    n = 10
    labels = ['q%d' % i for i in range(5)] + [(0, i) for i in range(5)]
    base = random_dense(n, 50, kmax=3, seed=9, unitary=True)
    gates = [(U, tuple(labels[q] for q in qs)) for U, qs in base]
    from hybridq_amd.simulation import all_qubits
    order = all_qubits(gates)
    assert sorted(map(str, order)) == sorted(map(str, labels))
    index = {q: i for i, q in enumerate(order)}
    as_int = [(U, tuple(index[q] for q in qs)) for U, qs in gates]
    rng = np.random.default_rng(1)
    psi0 = rng.standard_normal((2,) * n) + 1j * rng.standard_normal((2,) * n)
    psi0 /= np.linalg.norm(psi0.ravel())
    seen = []

    def negate_where_first_is_one(psi, order):
        seen.append(order)
        out = psi.copy()
        ax = order.index(order[0]) + 1
        idx = [slice(None)] * psi.ndim
        idx[ax] = 1
        out[tuple(idx)] *= -1
        return out, order

    class Tup:  # the reference's TupleGate duck-typed (a list subclass would read as a (U, qubits) pair)
        def __init__(self, gates):
            self.gates = list(gates)

        def __iter__(self):
            return iter(self.gates)

        def flatten(self):
            return self

    fn = FunctionalGate((order[0],), negate_where_first_is_one)
    circuit = [Tup(gates[:10]), Tup([Tup(gates[10:20]), gates[20]])] + gates[21:30] + [fn] + gates[30:]
    psi = simulate(circuit, initial_state=psi0, complex_type='complex128', compress=4)
    assert seen == [tuple(order)]
    a = oracle.evolve_tensordot(as_int[:30], n, initial_state=psi0.reshape(-1), qubits=list(range(n))).reshape((2,) * n).copy()
    a[1] *= -1
    exp = oracle.evolve_tensordot(as_int[30:], n, initial_state=a.reshape(-1), qubits=list(range(n)))
    assert _rel(psi.reshape(-1), exp) < 1e-12


def test_density_matrix_front_end(numpy_device):
    """dm.simulate: unitary gates act as U (x) conj(U) on the (0, q) / (1, q) copies, channels as one superoperator gate;
    against rho -> sum_i s_i K_i rho K_i^dagger carried out on the 2^n x 2^n matrix in numpy."""
    log, oracle = numpy_device
    from hybridq_amd import dm
    from hybridq_amd.circuits import random_dense
    n = 5
    rng = np.random.default_rng(2)
    gates = random_dense(n, 24, kmax=2, seed=21, unitary=True)
    circuit, rho_ops = [], []
    for i, (U, qs) in enumerate(gates):
        circuit.append((U, qs))
        rho_ops.append(('U', U, qs))
        if i % 6 == 5:
            qs2 = tuple(int(q) for q in rng.permutation(n)[:2])
            ch = dm.depolarizing(qs2, 0.1 + 0.05 * (i // 6))
            circuit.append(ch)
            rho_ops.append(('K', ch, qs2))
    init = '01+-0'

    psi0 = oracle.evolution._initial(init, n, np.complex128)
    rho = np.outer(psi0, psi0.conj())
    for kind, op, qs in rho_ops:
        if kind == 'U':
            F = _full(op, qs, n)
            rho = F @ rho @ F.conj().T
        else:
            rho = sum(s * (_full(L, qs, n) @ rho @ _full(L, qs, n).conj().T) for s, L in zip(op.s, op.left))
    for kw in (dict(compress=4), dict(compress=0), dict(blocked=True), {}):
        got = dm.simulate(circuit, initial_state=init, complex_type='complex128', **kw)
        assert got.shape == (2,) * (2 * n)
        assert _rel(got.reshape(1 << n, 1 << n), rho) < 1e-12, kw
    assert abs(np.trace(rho) - 1) < 1e-12


def _full(M, qs, n):
    """Dense 2^n x 2^n matrix of gate (M, qs): columns = images of the basis states under an independent evolution."""
    k = len(qs)
    Mt = np.asarray(M, dtype=np.complex128).reshape((2,) * (2 * k))
    eye = np.eye(1 << n, dtype=np.complex128).reshape((2,) * n + (1 << n,))
    out = np.moveaxis(np.tensordot(Mt, eye, axes=(list(range(k, 2 * k)), list(qs))), list(range(k)), list(qs))
    return out.reshape(1 << n, 1 << n)


def test_expectation_value(numpy_device):
    """expectation_value(state, op) = <state| op |state> with the qubits mapped to the axes of `state` in sorted order."""
    log, oracle = numpy_device
    from hybridq_amd.circuits import random_dense
    from hybridq_amd.simulation import expectation_value
    n = 8
    rng = np.random.default_rng(5)
    state = rng.standard_normal((2,) * n) + 1j * rng.standard_normal((2,) * n)
    state /= np.linalg.norm(state.ravel())
    op = random_dense(n, 6, kmax=2, seed=6)  # non-unitary: the value is complex
    exp = np.vdot(state.reshape(-1), oracle.evolve_tensordot(op, n, initial_state=state.reshape(-1), qubits=list(range(n))))
    got = expectation_value(state, op, qubits_order=list(range(n)), complex_type='complex128')
    assert abs(got - exp) < 1e-12
    herm = [(np.diag([1.0, -1.0]), (3,))]
    val = expectation_value(state, herm, qubits_order=list(range(n)), complex_type='complex128')
    assert isinstance(val, float) or abs(complex(val).imag) == 0
    z = state.reshape((2,) * n)
    assert abs(val - (np.sum(np.abs(np.take(z, 0, axis=3))**2) - np.sum(np.abs(np.take(z, 1, axis=3))**2))) < 1e-12
    with pytest.raises(ValueError):
        expectation_value(state, [(np.eye(2), (n + 3,))], qubits_order=list(range(n)))


def test_device_functional_gates_in_the_loop(numpy_device):
    """functional.Projection / Measure inside simulate(): bit conventions (qubits[0] = most significant outcome bit),
    renormalisation, the "nothing survives" case, and a measurement of more qubits than the marginal kernel tabulates
    (chunk by chunk: the joint law of one draw)."""
    log, oracle = numpy_device
    from hybridq_amd.circuits import random_dense
    from hybridq_amd.functional import Measure, Projection
    from hybridq_amd.simulation import simulate
    n = 14
    g1, g2 = random_dense(n, 40, kmax=2, seed=31, unitary=True), random_dense(n, 20, kmax=2, seed=32, unitary=True)
    q = list(range(n))
    mid = oracle.evolve_tensordot(g1, n, qubits=q).reshape((2,) * n)

    def collapse(psi, qubits, bits, renorm=True):
        out = np.zeros_like(psi)
        idx = [slice(None)] * n
        for qq, b in zip(qubits, bits):
            idx[qq] = int(b)
        out[tuple(idx)] = psi[tuple(idx)]
        return out / np.linalg.norm(out.ravel()) if renorm else out

    for qs, bits, renorm in (((3, 0, 9), '101', True), ((5,), '0', False), (tuple(range(12)), '011010011101', True)):
        psi = simulate(g1 + [Projection(bits, qs, renormalize=renorm)] + g2, initial_state='0' * n, complex_type='complex128',
                       qubits=q, compress=4)
        exp = oracle.evolve_tensordot(g2, n, initial_state=collapse(mid, qs, bits, renorm).reshape(-1), qubits=q)
        assert _rel(psi.reshape(-1), exp) < 1e-12, (qs, bits)
    # nothing survives: '1' on a qubit that is still |0>
    only_q0 = [(U, qs) for U, qs in g1 if 7 not in qs][:10]
    psi = simulate(only_q0 + [Projection('1', (7,))], initial_state='0' * n, complex_type='complex128', qubits=q, compress=0)
    assert not psi.any()
    for qs in ((2, 11, 4), tuple(range(1, 13))):  # 3 qubits: one draw; 12 qubits: two chunks
        m = Measure(qs, rng=np.random.default_rng(len(qs)))
        psi = simulate(g1 + [m] + g2, initial_state='0' * n, complex_type='complex128', qubits=q, compress=4)
        bits = format(m.outcome, '0%db' % len(qs))  # qubits[0] = most significant bit of the outcome
        exp = oracle.evolve_tensordot(g2, n, initial_state=collapse(mid, qs, bits).reshape(-1), qubits=q)
        assert _rel(psi.reshape(-1), exp) < 1e-12, qs
        assert abs(np.linalg.norm(psi.ravel()) - 1) < 1e-12
    # the sampled law: outcome frequencies of a 2-qubit measurement follow the marginal
    counts = np.zeros(4)
    rng = np.random.default_rng(0)
    marg = np.array([np.sum(np.abs(np.take(np.take(mid, a, axis=6), b, axis=1))**2) for a in (0, 1) for b in (0, 1)])  # qubits (6, 1)
    for _ in range(300):
        m = Measure((6, 1), rng=rng)
        simulate(g1 + [m], initial_state='0' * n, complex_type='complex128', qubits=q, compress=0, return_numpy_array=False)
        counts[m.outcome] += 1
    assert np.abs(counts / 300 - marg).max() < 0.1


def test_plans_are_cached_by_circuit_content(numpy_device):
    """simulate() keeps the schedules of the last circuits by content: the same circuit again skips planning
    (info['schedule']['from_cache']), a circuit that differs in one matrix entry does not, FunctionalGates switch it off."""
    from hybridq_amd import simulation as sim
    from hybridq_amd.circuits import rqc_1q2q
    sim._PLAN_CACHE.clear()
    n = 12
    gates = rqc_1q2q(n, depth=6, seed=3)
    kw = dict(initial_state='0' * n, optimize='evolution', return_info=True, qubits=list(range(n)))
    psi1, info1 = sim.simulate(gates, **kw)
    psi2, info2 = sim.simulate(gates, **kw)
    assert not info1['schedule'].get('from_cache') and info2['schedule'].get('from_cache') is True
    assert np.array_equal(psi1, psi2) and info2['host_ms']['plan'] <= info1['host_ms']['plan']
    other = list(gates)
    other[3] = (other[3][0] * np.exp(0.25j), other[3][1])
    _, info3 = sim.simulate(other, **kw)
    assert not info3['schedule'].get('from_cache') and len(sim._PLAN_CACHE) == 2
    fn = sim.FunctionalGate((0,), lambda psi, order: (psi, order))
    sim.simulate(gates[:5] + [fn] + gates[5:], **kw)
    assert len(sim._PLAN_CACHE) == 2
    _, info5 = sim.simulate(gates, compress=4, **{k: v for k, v in kw.items() if k != 'optimize'}, optimize='evolution-hybridq')
    assert 'schedule' not in info5 and len(sim._PLAN_CACHE) == 3


@pytest.mark.parametrize('kw', [dict(compress=0), dict(optimize='evolution'), dict(blocked=True)])
def test_cached_plans_do_not_alias_the_callers_matrices(numpy_device, kw):
    """A parameter scan that updates its matrices IN PLACE between calls: the plan kept under the first content digest
    must keep the first values (compress=0 and the gate-by-gate plan used to store the caller's own arrays,
    so a later circuit equal to the first one came back with the second one's amplitudes)."""
    from hybridq_amd import simulation as sim
    from hybridq_amd.circuits import rqc_1q2q
    sim._PLAN_CACHE.clear()
    n = 14
    gates = [(np.array(U, dtype='complex128'), q) for U, q in rqc_1q2q(n, depth=4, seed=5)]
    v1 = [(U.copy(), q) for U, q in gates]
    run = dict(dict(initial_state='0' * n, complex_type='complex128', qubits=list(range(n)), simplify=False), **kw)
    psi1 = sim.simulate(gates, **run)
    rng = np.random.default_rng(1)
    for U, _ in gates:  # in place: same objects, new values
        U *= np.exp(2j * np.pi * rng.random())
        U[0, :] *= -1
    psi2 = sim.simulate(gates, **run)
    assert np.abs(psi2 - psi1).max() > 1e-3
    psi3 = sim.simulate(v1, **run)  # fresh arrays equal to the first circuit: a cache hit on the first key
    assert np.array_equal(psi3, psi1)
    for plan, _ in sim._PLAN_CACHE.values():
        for op in plan:
            mats = [U for U, _ in op[2]] if op[0] == 'B' else [op[1]]
            assert not any(np.shares_memory(M, U) for M in mats for U, _ in gates)
