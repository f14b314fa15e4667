"""CPU tests of the oracle itself: the C restatement against the index formula in numpy,
against the compiled reference (oracle/_ref, when it was built) and the reference-protocol
driver against an independent float64 tensordot evolution."""
import numpy as np
import pytest

import oracle
from oracle.binding import aligned_empty

TOL = {np.dtype('float32'): 2e-6, np.dtype('float64'): 1e-13}


def _apply(lib, psi, U, pos, ft):
    pl = aligned_empty((2, psi.size), ft)
    pl[0], pl[1] = psi.real, psi.imag
    assert lib.apply_U(pl[0], pl[1], U, pos) == 0
    return pl[0] + 1j * pl[1]


@pytest.mark.parametrize('ft', ['float32', 'float64'])
def test_port_apply_U_any_position(oracle_port, ft):
    ft = np.dtype(ft)
    rng = np.random.default_rng(0)
    n = 12
    for k in range(1, 7):
        for _ in range(4):
            pos = rng.permutation(n)[:k]
            U = rng.standard_normal((2**k, 2**k)) + 1j * rng.standard_normal((2**k, 2**k))
            psi = rng.standard_normal(2**n) + 1j * rng.standard_normal(2**n)
            got = _apply(oracle_port, psi, U, pos, ft)
            exp = oracle.apply_gate_numpy(psi, U, pos)
            assert np.abs(got - exp).max() / np.abs(exp).max() < TOL[ft] * 2**k


def test_port_edge_cases(oracle_port):
    rng = np.random.default_rng(1)
    # k == n (whole-state matrix), n = 1, k = 0 no-op, invalid positions
    for n in (1, 2, 3, 5):
        U = rng.standard_normal((2**n, 2**n)) + 1j * rng.standard_normal((2**n, 2**n))
        psi = rng.standard_normal(2**n) + 1j * rng.standard_normal(2**n)
        pos = rng.permutation(n)
        got = _apply(oracle_port, psi, U, pos, np.float64)
        assert np.allclose(got, oracle.apply_gate_numpy(psi, U, pos))
    re = np.ones(8)
    im = np.zeros(8)
    assert oracle_port.apply_U(re, im, np.ones((1, 1)), []) == 0 and (re == 1).all()
    assert oracle_port.apply_U(re, im, np.eye(2), [3]) != 0
    assert oracle_port.apply_U(re, im, np.eye(4), [1, 1]) != 0


@pytest.mark.parametrize('ft', ['float32', 'float64'])
def test_port_matches_compiled_reference(oracle_port, oracle_ref, ft):
    """Pin the restatement on the reference's own binary (positions >= 3 as the reference
    requires, U.h:48-54; planes 256-byte aligned)."""
    ft = np.dtype(ft)
    rng = np.random.default_rng(2)
    n = 14
    assert oracle_ref.log2_pack_size == 3
    for k in range(1, 8):
        for _ in range(3):
            pos = rng.permutation(np.arange(3, n))[:k]
            U = rng.standard_normal((2**k, 2**k)) + 1j * rng.standard_normal((2**k, 2**k))
            psi = rng.standard_normal(2**n) + 1j * rng.standard_normal(2**n)
            a = _apply(oracle_port, psi, U, pos, ft)
            b = _apply(oracle_ref, psi, U, pos, ft)
            assert np.abs(a - b).max() / np.abs(b).max() < TOL[ft] * 2**k
    # the reference refuses positions below its pack size; the port does not
    re = aligned_empty(1 << n, ft)
    im = aligned_empty(1 << n, ft)
    re[:] = 1
    im[:] = 0
    assert oracle_ref.apply_U(re, im, np.eye(2), [1]) != 0


@pytest.mark.parametrize('dt', ['float32', 'float64', 'int32', 'int64', 'uint32', 'uint64'])
def test_swap_port_ref_numpy(oracle_port, dt):
    dt = np.dtype(dt)
    rng = np.random.default_rng(3)
    n = 14
    ref = oracle.load_ref() if oracle.have_ref() else None
    for s in (1, 2, 4, 6, 8, 9, 11, 14):
        a = rng.integers(0, 1000, 2**n).astype(dt)
        pos = rng.permutation(s)
        exp = oracle.swap_numpy(a, pos)
        tr = np.transpose(a.reshape((2,) * n),
                          list(range(n - s)) + [n - 1 - int(pos[i]) for i in reversed(range(s))])
        assert (tr.reshape(-1) == exp).all()  # swap == this transpose
        b = a.copy()
        assert oracle_port.swap(b, pos) == 0
        assert (b == exp).all()
        if ref is not None:
            c = aligned_empty(a.shape, dt, alignment=4096)  # reference needs vector alignment
            c[:] = a
            assert ref.swap(c, pos) == 0
            assert (c == exp).all()
    assert oracle_port.swap(a, [0, 0]) != 0


def test_to_complex(oracle_port):
    rng = np.random.default_rng(4)
    for ft in (np.float32, np.float64):
        re = rng.standard_normal(1000).astype(ft)
        im = rng.standard_normal(1000).astype(ft)
        assert (oracle_port.to_complex(re, im) == re + 1j * im).all()


@pytest.mark.parametrize('ct', ['complex64', 'complex128'])
def test_reference_protocol_vs_tensordot(oracle_port, ct):
    """The driver restatement (swap policy + pos construction, simulation.py:491-675)
    against an independent float64 evolution, incl. k > 4 gates (the :596-605 branch)."""
    rng = np.random.default_rng(5)
    n = 12
    gates = []
    for _ in range(60):
        k = int(rng.integers(1, 7))
        qs = tuple(int(x) for x in rng.permutation(n)[:k])
        U = (rng.standard_normal((2**k, 2**k)) + 1j * rng.standard_normal((2**k, 2**k))) / 2**(k / 2)
        gates.append((U, qs))
    exp = oracle.evolve_tensordot(gates, n)
    trace = []
    psi, info = oracle.evolve_reference_protocol(oracle_port, gates, n, complex_type=ct, trace=trace)
    tol = 2e-6 if ct == 'complex64' else 1e-12
    assert np.abs(psi - exp).max() / np.abs(exp).max() < tol
    assert any(t[0] == 'S' for t in trace) and all(min(t[1]) >= 3 for t in trace if t[0] == 'U')
    if oracle.have_ref():
        psi2, _ = oracle.evolve_reference_protocol(oracle.load_ref(), gates, n, complex_type=ct)
        assert np.abs(psi2 - exp).max() / np.abs(exp).max() < tol
