"""CPU tests of the planner behind the C ABI (hq_plan_blocked, csrc/hq_plan.hip) against the Python statement of the
same algorithm (hybridq_amd/blocking.py, native=False) and against an independent evolution: every op the native plan
issues is legal for hq_apply_blocked_* (k <= 4, targets inside an ascending tile of the right size, component bits
included) and the plan is the same operator as the circuit it was given."""
import time

import numpy as np
import pytest

from hybridq_amd.blocking import blocked_stats, plan_blocked
from hybridq_amd.circuits import random_dense, rqc_1q2q
from oracle.evolution import apply_gate_numpy


def _evolve(ops_or_gates, n, psi, pos_of=None):
    for op in ops_or_gates:
        if isinstance(op[0], str):
            if op[0] == 'B':
                tile = [int(p) for p in op[1]]
                assert tile == sorted(set(tile)) and len(tile) == min(13, n) or len(tile) == len(op[1])
                for U, pos in op[2]:
                    assert 1 <= len(pos) <= 4 and set(pos) <= set(tile)
                    psi = apply_gate_numpy(psi, np.asarray(U, dtype=np.complex128), list(pos))
            else:
                psi = apply_gate_numpy(psi, np.asarray(op[1], dtype=np.complex128), list(op[2]))
        else:
            U, qs = op
            psi = apply_gate_numpy(psi, np.asarray(U, dtype=np.complex128), [pos_of[q] for q in reversed(qs)])
    return psi


@pytest.mark.parametrize('case', ['rqc', 'dense_k34', 'wide', 'options'])
def test_native_plan_is_the_circuit(case):
    n = 15
    rng = np.random.default_rng(5)
    if case == 'rqc':
        gates, kw = rqc_1q2q(n, depth=12, seed=3), {}
    elif case == 'dense_k34':
        gates, kw = random_dense(n, 60, kmax=4, seed=4, unitary=True), {}
    elif case == 'wide':  # gates that fit no tile run on their own, in order
        gates, kw = random_dense(n, 40, kmax=6, seed=6, unitary=True), {}
    else:
        gates, kw = rqc_1q2q(n, depth=8, seed=7), dict(tile_bits=10, low_bits=4, inner_max=3, min_gates=4, tries=4, fusion_orders=1)
    # a placement that is not the identity: label q sits at position perm[q]
    perm = rng.permutation(n)
    pos_of = {q: int(perm[q]) for q in range(n)}
    psi0 = rng.standard_normal(1 << n) + 1j * rng.standard_normal(1 << n)
    psi0 /= np.linalg.norm(psi0)
    want = _evolve(gates, n, psi0.copy(), pos_of)
    for native in (True, False):
        ops = plan_blocked(gates, pos_of, n, complex_type='complex128', native=native, **kw)
        got = _evolve(ops, n, psi0.copy())
        assert np.abs(got - want).max() < 1e-12, (case, native)
        st = blocked_stats(ops)
        if case != 'wide':
            assert st['blocked_passes'] >= 1
        for op in ops:
            if op[0] == 'B':
                tb = kw.get('tile_bits', 13)
                assert len(op[1]) == min(tb, n) and op[1].dtype == np.uint32
                assert list(op[1][:2]) == [0, 1]  # the vector-component index bits are always in the tile


def test_native_plan_is_deterministic_and_no_worse_than_the_python_planner():
    n = 24
    gates = rqc_1q2q(n, depth=20, seed=1)
    pos_of = {q: n - 1 - q for q in range(n)}
    a = plan_blocked(gates, pos_of, n, native=True)
    b = plan_blocked(gates, pos_of, n, native=True)
    assert len(a) == len(b) and all(x[0] == y[0] and np.array_equal(x[1], y[1]) for x, y in zip(a, b))
    c = plan_blocked(gates, pos_of, n, native=False)
    sa, sc = blocked_stats(a), blocked_stats(c)
    assert sa['blocked_passes'] + sa['plain_gates'] <= sc['blocked_passes'] + sc['plain_gates'] + 2, (sa, sc)
    assert sa['inner_gates'] <= 1.15 * sc['inner_gates'], (sa, sc)


def test_native_planner_is_fast_enough_to_wait_for():
    """planning the n = 30 benchmark circuit must cost far less than its 137 ms loop."""
    n = 30
    gates = rqc_1q2q(n, depth=40, seed=1)
    pos_of = {q: n - 1 - q for q in range(n)}
    plan_blocked(gates[:50], pos_of, n)  # load / warm
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        ops = plan_blocked(gates, pos_of, n)
        best = min(best, time.perf_counter() - t0)
    assert blocked_stats(ops)['blocked_passes'] <= 31
    assert best < 0.060, f'{best * 1e3:.1f} ms'  # measured 17 ms on a busy 8-core container; the Python planner: 110-370 ms


def test_plan_abi_rejects_bad_input():
    from hybridq_amd import core
    U = np.eye(2, dtype=np.complex128)
    with pytest.raises(core.HQError, match='invalid positions'):
        core.plan_blocked(4, [(U, [7])], 4, 2, 'auto', 3, 1, 1, 4, 0, 1e-12)
    with pytest.raises(core.HQError, match='commute_tol'):
        core.plan_blocked(4, [(U, [1])], 4, 2, 'auto', 3, 1, 1, 4, 0, 0.0)
    kind, first, tile, gk, gpos, mats = core.plan_blocked(4, [], 4, 2, 'auto', 3, 1, 1, 4, 0, 1e-12)
    assert len(kind) == 0 and list(first) == [0]


def test_native_simplify_equals_the_python_statement():
    """hq_plan_simplify against fusion.simplify(native=False) -- the statement the live tests hold against the reference
    itself -- on circuits with planted identities, inverse pairs and commuting diagonal gates, under every option."""
    from hybridq_amd import fusion
    rng = np.random.default_rng(0)

    def same(a, b):
        return len(a) == len(b) and all(x[1] == y[1] and np.array_equal(x[0], y[0]) for x, y in zip(a, b))

    for trial in range(30):
        n = int(rng.integers(4, 9))
        g = list(rqc_1q2q(n, depth=int(rng.integers(2, 6)), seed=trial))
        for _ in range(6):
            q, i, kind = int(rng.integers(n)), int(rng.integers(len(g) + 1)), int(rng.integers(4))
            if kind == 0:
                g.insert(i, (np.eye(2), (q,)))
            elif kind == 1:
                U = g[int(rng.integers(len(g)))][0]
                qs = tuple(int(x) for x in rng.permutation(n)[:int(np.log2(U.shape[0]))])
                g.insert(i, (U, qs))
                g.insert(i + 1, (np.linalg.inv(U), qs))
            elif kind == 2:
                g.insert(i, (np.diag(np.exp(1j * rng.standard_normal(4))), tuple(int(x) for x in rng.permutation(n)[:2])))
            else:
                g.insert(i, (np.diag(np.exp(1j * rng.standard_normal(2))), (q,)))
        labelled = [(U, tuple(('q', x) if x % 2 else f's{x}' for x in qs)) for U, qs in g]  # labels need not be integers
        for kw in (dict(), dict(use_matrix_commutation=False), dict(max_n_qubits_matrix=1), dict(remove_id_gates=False), dict(atol=1e-3)):
            assert same(fusion.simplify(g, native=False, **kw), fusion.simplify(g, native=True, **kw)), (trial, kw)
            assert same(fusion.simplify(labelled, native=False, **kw), fusion.simplify(labelled, **kw)), (trial, kw)
    assert len(fusion.simplify([(np.eye(2), (0,))])) == 0 and fusion.simplify([]) == []


def test_native_fuse_equals_the_python_statement():
    """hq_plan_fuse against fusion.fuse(native=False) (the statement the live tests hold against the reference's
    compress + to_matrix_gate) under the reference's options and the planner's exact commutation."""
    from hybridq_amd import fusion
    rng = np.random.default_rng(1)

    def same(a, b):
        return len(a) == len(b) and all(x[1] == y[1] and x[0].dtype == y[0].dtype and np.abs(x[0] - y[0]).max() < 1e-12
                                        for x, y in zip(a, b))

    for trial in range(25):
        n = int(rng.integers(4, 10))
        g = list(rqc_1q2q(n, depth=int(rng.integers(2, 8)), seed=trial)) + random_dense(n, int(rng.integers(0, 10)), kmax=3, seed=trial)
        for _ in range(5):  # commuting diagonal gates: the matrix-commutation branch
            i = int(rng.integers(len(g) + 1))
            qs = tuple(int(x) for x in rng.permutation(n)[:int(rng.integers(1, 4))])
            g.insert(i, (np.diag(np.exp(1j * rng.standard_normal(1 << len(qs)))), qs))
        labelled = [(U, tuple(f'q{x:02d}' for x in qs)) for U, qs in g]
        for kw in (dict(max_n_qubits=2), dict(max_n_qubits=4), dict(max_n_qubits=5, complex_type='complex128'),
                   dict(max_n_qubits=4, use_matrix_commutation=False), dict(max_n_qubits=4, max_n_qubits_matrix=2),
                   dict(max_n_qubits=3, exclude_qubits=[0, 2]), dict(max_n_qubits=3, exact_commutation=True),
                   dict(max_n_qubits=6, max_n_qubits_matrix=3)):
            assert same(fusion.fuse(g, native=False, **kw), fusion.fuse(g, native=True, **kw)), (trial, kw)
        assert same(fusion.fuse(labelled, 4, native=False), fusion.fuse(labelled, 4)), trial


def test_planner_limits_come_back_as_error_codes():
    """a C caller's oversized limits must not abort the process (std::bad_alloc across the ABI): max_n_qubits
    above 10 and gates wider than 10 qubits are refused with a message; a commutator over a union wider than 12 qubits is
    not evaluated (both planners treat the pair as not commuting)."""
    import ctypes
    from hybridq_amd import core, fusion
    U32P = ctypes.POINTER(ctypes.c_uint32)
    k = np.array([1, 1], dtype=np.uint32)
    q = np.array([0, 1], dtype=np.uint32)
    U = np.tile(np.eye(2, dtype=np.complex128).view(np.float64).reshape(-1), 2).copy()
    plan = ctypes.c_void_p()
    rc = core._lib.hq_plan_fuse(ctypes.c_uint(2), ctypes.c_uint(2), k.ctypes.data_as(U32P), q.ctypes.data_as(U32P), ctypes.c_void_p(U.ctypes.data),
                                ctypes.c_uint(40), ctypes.c_int(1), ctypes.c_uint(40), ctypes.c_uint64(0), ctypes.c_double(1e-5), ctypes.byref(plan))
    assert rc == 1 and 'limited to 10 qubits' in core.last_error() and not plan.value
    k11 = np.array([11], dtype=np.uint32)
    q11 = np.arange(11, dtype=np.uint32)
    rc = core._lib.hq_plan_fuse(ctypes.c_uint(11), ctypes.c_uint(1), k11.ctypes.data_as(U32P), q11.ctypes.data_as(U32P), ctypes.c_void_p(U.ctypes.data),
                                ctypes.c_uint(4), ctypes.c_int(1), ctypes.c_uint(10), ctypes.c_uint64(0), ctypes.c_double(1e-5), ctypes.byref(plan))
    assert rc == 1 and '1..10 qubits' in core.last_error()
    wide = np.eye(1 << 7, dtype=np.complex128)  # identities commute -- but the union of the two has 13 qubits
    assert not fusion.commute(wide, tuple(range(7)), wide, tuple(range(6, 13)))
    assert fusion.commute(wide, tuple(range(7)), wide, tuple(range(5, 12)))
