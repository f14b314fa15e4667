"""The reference's own tests of the evolution path (/root/reference/tests/tests.py), restated one by one
against this package under the reference's names.  Each test follows the reference test's procedure -- same sizes, same
keyword arguments, same assertions at the reference's tolerance (rtol = atol = 1e-3, tests.py:57-62) -- with the
reference's gate classes duck-typed where the path needs them (TupleGate, StochasticGate, MessageGate), and adds the
north-star parity check against the float64 oracle on top.  Tests of this path mirrored elsewhere under other names:
test_simulation_2__fn (test_gpu_round2.py), test_simulation_4__simulation_large (test_gpu_parity.py),
test_utils__dot / __transpose (test_gpu_golden.py, test_gpu_parity.py), test_gates__measure / __projection and
test_simulation_5__expectation_value (test_gpu_golden.py, test_gpu_parity.py), test_dm_2__simulation_2
(test_gpu_golden.py::test_dm_front_end on the reference's recorded circuit)."""
import io
import os
import sys
from functools import partial

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from tolerances import circuit_tol  # noqa: E402

assert_allclose = partial(np.testing.assert_allclose, rtol=1e-3, atol=1e-3)  # tests.py:57-62


class TupleGate:
    """Gate('TUPLE', gates=...) duck-typed: iterable, provides flatten (base/property.py:445-451)."""

    def __init__(self, gates):
        self.gates = list(gates)

    def __iter__(self):
        return iter(self.gates)

    def flatten(self):
        return TupleGate(g for x in self.gates for g in (x.flatten() if isinstance(x, TupleGate) else [x]))

    @property
    def qubits(self):
        return tuple(sorted({q for g in self.flatten() for q in g[1]}))


class StochasticGate:
    """Gate('STOC', gates=..., p=...) duck-typed: sample() draws one gate with numpy's global generator."""

    def __init__(self, gates, p):
        self.gates, self.p = list(gates), np.asarray(p)

    def sample(self):
        return self.gates[np.random.choice(len(self.gates), p=self.p)]


class MessageGate:
    """extras.gate.MessageGate duck-typed: a FunctionalGate on no qubits that prints and leaves the state alone."""
    qubits = ()
    name = 'MESSAGE'

    def __init__(self, message, file):
        self.message, self.file = message, file

    def apply(self, psi, order):
        print(self.message, file=self.file)
        return psi, order


class IdentityGate:
    """Gate('I', [q])"""
    name = 'I'

    def __init__(self, q):
        self.qubits = (q,)

    def matrix(self):
        return np.eye(2)


def _prepare_state(initial_state):
    import oracle
    from oracle.evolution import _initial
    n = len(initial_state)
    return _initial(initial_state, n, np.complex64).reshape((2,) * n)


@pytest.mark.parametrize('t', [t for t in ['float32', 'float64', 'float128'] for _ in range(8)])
def test_utils__to_complex(torch_cuda, t):
    """tests.py:122-149: random 8-D shapes; float128 takes the numpy fallback like in the reference (dot.py:113-124)."""
    from hybridq_amd.dot import to_complex, to_complex_array
    shape = np.random.randint(2, 6, size=8)
    a = np.random.random(shape).astype(t)
    b = np.random.random(shape).astype(t)
    c = to_complex(a, b)
    _a, _b = to_complex_array(c)
    assert c.dtype == (a[:1] + 1j * b[:1]).dtype and _a.dtype == a.dtype and _b.dtype == b.dtype
    assert c.shape == a.shape and _a.shape == a.shape and _b.shape == a.shape
    assert np.array_equal(c, a + 1j * b) and np.array_equal(_a, a) and np.array_equal(_b, b)  # exact, not just close


@pytest.mark.parametrize('n_qubits', [n for n in range(16, 25, 4) for _ in range(2)])
@pytest.mark.parametrize('alphabet', ['01', '0+', '01+-'], ids=['1a', '1b', '2'])
def test_simulation_1__initialize_state(torch_cuda, n_qubits, alphabet):
    """tests.py:1871-1939 (1a: random '01' strings; 1b: all '0' and all '+'; 2: random '01+-'): a circuit of identity
    gates with remove_id_gates=False through the path returns prepare_state's array."""
    from hybridq_amd.simulation import simulate
    if alphabet == '0+':
        states = ['0' * n_qubits, '+' * n_qubits]
    else:
        states = [''.join(np.random.choice(list(alphabet), size=n_qubits))]
    for initial_state in states:
        _s1 = _prepare_state(initial_state)
        _s2 = simulate(circuit=[IdentityGate(q) for q in range(n_qubits)], initial_state=initial_state,
                       remove_id_gates=False, optimize='evolution', verbose=False)
        assert _s2.shape == (2,) * n_qubits and _s2.dtype == np.complex64
        assert_allclose(_s1, _s2)
        assert np.abs(_s1 - _s2).max() <= 1e-6 * np.abs(_s1).max()  # north-star bar


@pytest.mark.parametrize('seed', range(3))
def test_simulation_2__tuple(torch_cuda, seed):
    """tests.py:1942-1977: n = 12, 200 random non-unitary gates; the circuit, the circuit cut into TupleGates of 4 / 5
    gates and ONE TupleGate holding everything give the same state (simulate flattens containers, simulation.py:239)."""
    import oracle
    from hybridq_amd.circuits import random_dense
    from hybridq_amd.simulation import simulate
    n_qubits, depth = 12, 200
    circuit = random_dense(n_qubits, depth, kmax=2, seed=100 + seed)
    initial_state = ''.join(np.random.default_rng(seed).choice(list('01+-'), size=n_qubits))
    c1 = [TupleGate(circuit[i:i + 4]) for i in range(0, depth, 4)]
    c2 = [TupleGate([TupleGate(circuit[i:i + 3]), TupleGate(circuit[i + 3:i + 5])]) for i in range(0, depth, 5)]  # nested
    g = TupleGate(circuit)
    kw = dict(initial_state=initial_state, qubits=list(range(n_qubits)))
    psi1, psi2, psi3, psi4 = (simulate(c, **kw) for c in (circuit, c1, c2, [g]))
    assert g.qubits == tuple(range(n_qubits))
    assert_allclose(psi1, psi2)
    assert_allclose(psi1, psi3)
    assert np.array_equal(psi1, psi2) and np.array_equal(psi1, psi3) and np.array_equal(psi1, psi4)  # same calls issued
    exp = oracle.evolve_tensordot(circuit, n_qubits, initial_state=initial_state, qubits=list(range(n_qubits)))
    assert np.abs(psi1.reshape(-1) - exp).max() / np.abs(exp).max() < circuit_tol(circuit)


@pytest.mark.parametrize('seed', range(2))
def test_simulation_2__message(torch_cuda, seed):
    """tests.py:1980-2034: a MessageGate (FunctionalGate on no qubits) after every gate neither changes the state nor
    joins a fused gate, and every message is printed exactly once."""
    from hybridq_amd.circuits import random_dense
    from hybridq_amd.simulation import _is_functional, _plan_ops, simulate
    n_qubits, depth = 12, 200
    file = io.StringIO()
    circuit = random_dense(n_qubits, depth, kmax=2, seed=200 + seed)
    circuit_msg = [x for i, g in enumerate(circuit) for x in (g, MessageGate(f'{i}', file))]
    # compression: the MessageGates stay isolated (never part of a fused gate), and -- acting on no qubit -- let every gate
    # slide across them exactly as the reference's walk does (circuit/utils.py:636-637): the matrix gates fuse as if the
    # messages were not there
    from hybridq_amd.fusion import fuse
    ops = _plan_ops(circuit_msg, list(range(n_qubits)), n_qubits, np.dtype('complex64'), 4, False)
    assert sum(1 for o in ops if _is_functional(o)) == depth and len(ops) == depth + len(fuse(circuit, 4))
    psi = simulate(circuit, initial_state='0', qubits=list(range(n_qubits)), compress=0, simplify=False)
    psi_msg = simulate(circuit_msg, initial_state='0', qubits=list(range(n_qubits)))
    assert_allclose(psi, psi_msg)  # (the gates now fuse across the messages: equal to rounding, as upstream asserts)
    file.seek(0)
    # every message exactly once; like the reference's, the simplify pass leaves the zero-qubit gates in another order
    # (they commute with everything) -- tests.py:2030-2034 sorts before comparing, and so does this
    assert sorted(int(x.strip()) for x in file.readlines()) == list(range(depth))


def test_simulation_2__stochastic(torch_cuda):
    """tests.py:2111-2197: n = 12; circuit_1 + STOC(20 gates, random p) + circuit_2 sampled 200 times with
    allow_sampling=True averages to the p-weighted sum of the 20 exact evolutions within 1 / sqrt(n_samples);
    `sampling_seed` makes a draw reproducible and leaves numpy's global generator where it was."""
    from hybridq_amd.circuits import random_dense
    from hybridq_amd.fusion import fuse, simplify
    from hybridq_amd.simulation import simulate
    n_qubits, depth, n_samples = 12, 100, 200
    qubits = list(range(n_qubits))
    for attempt in range(3):  # the reference repeats as well: sampling may fail by chance
        rng = np.random.default_rng(300 + attempt)
        circuit_1, circuit_2 = ([(U, qs) for U, qs in fuse(simplify(random_dense(n_qubits, depth // 2, kmax=2, seed=310 + 2 * attempt + i)), 4)]
                                for i in (0, 1))
        initial_state = ''.join(rng.choice(list('01'), size=3)) + ''.join(rng.choice(list('01+-'), size=n_qubits - 3))
        p = rng.random(20)
        p /= p.sum()
        stoc = StochasticGate(random_dense(n_qubits, 20, kmax=2, seed=330 + attempt), p)
        kw = dict(initial_state=initial_state, optimize='evolution', simplify=False, compress=0, qubits=qubits)
        exact = np.zeros((2,) * n_qubits, dtype='complex64')
        for gate, pr in zip(stoc.gates, stoc.p):
            exact += pr * simulate(circuit_1 + [gate] + circuit_2, **kw)
        sample = np.zeros((2,) * n_qubits, dtype='complex64')
        for _ in range(n_samples):
            sample += simulate(circuit_1 + [stoc] + circuit_2, allow_sampling=True, **kw)
        sample /= n_samples
        # reproducible draws, global generator untouched
        state = np.random.get_state()[1].copy()
        a = simulate(circuit_1 + [stoc] + circuit_2, allow_sampling=True, sampling_seed=7, **kw)
        b = simulate(circuit_1 + [stoc] + circuit_2, allow_sampling=True, sampling_seed=7, **kw)
        assert np.array_equal(a, b) and np.array_equal(np.random.get_state()[1], state)
        # without allow_sampling a stochastic gate is not something the loop can apply (simulation.py:648-649)
        with pytest.raises(RuntimeError):
            simulate(circuit_1 + [stoc] + circuit_2, **kw)
        try:
            np.testing.assert_allclose(exact, sample, atol=1 / np.sqrt(n_samples))
            return
        except AssertionError as e:
            print(e, file=sys.stderr)
    raise RuntimeError('All tests have failed')


def test_dm_1__simulation_1(torch_cuda):
    """tests.py:2632-2670: n = 12 unitary circuit of 200 gates -> the 24-qubit state vector of rho through the path;
    rho is Hermitian, idempotent, of unit purity, positive semi-definite and equals psi (x) psi*."""
    from scipy.linalg import eigvalsh
    from hybridq_amd import dm
    from hybridq_amd.circuits import random_dense
    from hybridq_amd.simulation import simulate
    n_qubits, n_gates = 12, 200
    circuit = random_dense(n_qubits, n_gates, kmax=2, seed=12, unitary=True)
    initial_state = ''.join(np.random.default_rng(12).choice(list('01+-'), size=n_qubits))
    psi_1 = simulate(circuit, initial_state=initial_state, qubits=list(range(n_qubits)))
    rho_1 = dm.simulate(circuit, initial_state=initial_state)
    assert rho_1.shape == (2,) * (2 * n_qubits)
    _rho_1 = np.reshape(rho_1, (2**n_qubits, 2**n_qubits))
    assert_allclose(_rho_1, _rho_1.conj().T)
    sq = _rho_1 @ _rho_1
    assert_allclose(_rho_1, sq)
    assert np.isclose(np.trace(sq), 1, atol=1e-3)
    assert np.all(np.round(eigvalsh(_rho_1.astype(np.complex128)), 5) >= 0)
    expected = np.kron(psi_1.ravel(), psi_1.ravel().conj())
    assert_allclose(expected, rho_1.ravel())
    sv = dm.to_statevector_circuit(circuit)
    # rho's own rounding (400 one-sided gates) against psi (x) psi*, which carries psi's rounding twice
    assert np.abs(expected - rho_1.ravel()).max() / np.abs(expected).max() < 2 * circuit_tol(sv, circuit)


def test_return_numpy_array_false_is_the_split_state(torch_cuda):
    """simulation.py:669-675: without `return_numpy_array` the reference hands back its working array, real and imaginary
    parts as a (2,) + (2,)*n real array; here that is np.asarray() of the device-resident state."""
    from hybridq_amd.circuits import rqc_1q2q
    from hybridq_amd.simulation import EvolutionState, simulate
    n = 14
    gates = rqc_1q2q(n, depth=6, seed=5)
    psi = simulate(gates, initial_state='+' * n, qubits=list(range(n)))
    state = simulate(gates, initial_state='+' * n, qubits=list(range(n)), return_numpy_array=False)
    assert isinstance(state, EvolutionState)
    split = np.asarray(state)
    assert split.shape == (2,) + (2,) * n and split.dtype == np.float32
    assert np.array_equal(split[0], psi.real) and np.array_equal(split[1], psi.imag)
    assert np.asarray(state, dtype=np.float64).dtype == np.float64


def test_container_gates_against_the_reference_itself(torch_cuda):
    """tests/golden/e2e_containers.npz: final states the REFERENCE returned (make_golden.py containers) for a circuit cut
    into TupleGates, for a StochasticGate drawn under `sampling_seed` (the draw is numpy's global generator seeded the way
    simulation.py:241-256 does it: same seed, same gate) and for zero-qubit MessageGates after every gate."""
    import golden_util as gu
    from hybridq_amd.simulation import simulate
    z = gu.load('e2e_containers.npz')
    gs = gu.rqc_gates(z, 'tup')
    n = 12
    init = str(z['init'])
    q = list(range(n))
    kw = dict(initial_state=init, complex_type='complex128', compress=0, simplify=False, qubits=q)
    exp = z['tup_psi']
    scale = np.abs(exp).max()
    for circuit in (gs, [TupleGate(gs[i:i + 4]) for i in range(0, len(gs), 4)], [TupleGate(gs)]):
        assert np.abs(simulate(circuit, **kw).reshape(-1) - exp).max() / scale < 1e-12
    stoc = StochasticGate(gu.rqc_gates(z, 'stoc'), z['stoc_p'])
    picked = set()
    for seed in z['stoc_seeds']:
        psi = simulate(gs[:40] + [stoc] + gs[40:], allow_sampling=True, sampling_seed=int(seed), **kw).reshape(-1)
        ref = z[f'stoc_psi_{int(seed)}']
        assert np.abs(psi - ref).max() / np.abs(ref).max() < 1e-12, int(seed)
        picked.add(ref.tobytes())
    assert len(picked) > 1  # the seeds do select different gates
    file = io.StringIO()
    msg = [x for i, g in enumerate(gs) for x in (g, MessageGate(f'{i}', file))]
    psi = simulate(msg, initial_state=init, complex_type='complex128', qubits=q).reshape(-1)
    assert np.abs(psi - z['msg_psi']).max() / np.abs(z['msg_psi']).max() < 1e-12
    file.seek(0)
    # every message exactly once, in the order the REFERENCE printed them: its simplify pass slides the zero-qubit gates past
    # everything (they commute with everything: 79, 78, ... here), and this driver's walk is the same (fusion.Opaque)
    ours = [int(x.strip()) for x in file.readlines()]
    assert ours == [int(x) for x in z['msg_lines']] and sorted(ours) == list(range(len(gs)))


def test_prepare_state_api(torch_cuda):
    """hybridq.circuit.simulation.prepare_state (utils.py:41-156) against the arrays the reference returned for the same
    strings (tests/golden/e2e_api.npz), and its argument errors."""
    import golden_util as gu
    from hybridq_amd.simulation import prepare_state
    z = gu.load('e2e_api.npz')
    for i, st in enumerate(z['ps_strings']):
        st = str(st)
        for ct, tol in (('complex64', 1e-7), ('complex128', 1e-15)):
            got = prepare_state(st, complex_type=ct)
            assert got.shape == (2,) * len(st) and got.dtype == np.dtype(ct)
            assert np.abs(got.reshape(-1) - z[f'ps_{i}']).max() <= tol, (st, ct)
    for bad, kw in (('01a', {}), ('01', dict(d=3)), ('01', dict(d=[2, 2, 2])), ('0', dict(d=0))):
        with pytest.raises(ValueError):
            prepare_state(bad, **kw)
