"""Seeded synthetic circuits for the benchmark configurations.

A circuit is a plain list of ``(U, qubits)``: ``U`` a dense 2^k x 2^k complex128
matrix whose index has ``qubits[0]`` as MOST significant bit (the convention of the
reference's ``gate.matrix()``), ``qubits`` a tuple of integer labels.  No reference
code is used: Haar unitaries come from a QR decomposition with phase fix.
"""
import numpy as np


def haar_unitary(dim, rng):
    """Haar-distributed dim x dim unitary (QR of a Ginibre matrix, phases fixed)."""
    z = (rng.standard_normal((dim, dim)) + 1j * rng.standard_normal((dim, dim))) / np.sqrt(2.0)
    q, r = np.linalg.qr(z)
    d = np.diagonal(r)
    return q * (d / np.abs(d))


def rqc_1q2q(n, depth=40, seed=None):
    """BASELINE cfg2/cfg3: `depth` layers alternating (a) Haar U(2) on every qubit and
    (b) Haar U(4) on a random perfect matching of the qubits.  depth=40, n=30 gives
    20*(30+15) = 900 gate applications, k in {1,2}."""
    rng = np.random.default_rng(n if seed is None else seed)
    gates = []
    for layer in range(depth):
        if layer % 2 == 0:
            for q in range(n):
                gates.append((haar_unitary(2, rng), (q,)))
        else:
            perm = rng.permutation(n)
            for i in range(0, n - 1, 2):
                gates.append((haar_unitary(4, rng), (int(perm[i]), int(perm[i + 1]))))
    return gates


def dense_kq(n, n_gates=200, ks=(3, 4), seed=34):
    """BASELINE cfg4: Haar U(2^k) on random distinct qubit k-tuples, k cycling over `ks`."""
    rng = np.random.default_rng(seed)
    gates = []
    for i in range(n_gates):
        k = ks[i % len(ks)]
        qs = tuple(int(q) for q in rng.permutation(n)[:k])
        gates.append((haar_unitary(1 << k, rng), qs))
    return gates


def random_dense(n, n_gates, kmax=4, seed=0, unitary=False):
    """Random (by default NON-unitary, like the reference's own tests, tests.py:299-391)
    dense gates on random qubits, k uniform in 1..kmax."""
    rng = np.random.default_rng(seed)
    gates = []
    for _ in range(n_gates):
        k = int(rng.integers(1, kmax + 1))
        qs = tuple(int(q) for q in rng.permutation(n)[:k])
        if unitary:
            U = haar_unitary(1 << k, rng)
        else:
            U = (rng.standard_normal((1 << k, 1 << k)) + 1j * rng.standard_normal((1 << k, 1 << k)))
            U /= np.sqrt(2.0 * (1 << k))  # keeps the state norm O(1)
        gates.append((U, qs))
    return gates


# ---------------------------------------------------------------------------------
# BASELINE cfg1: the 99-gate, 24-qubit circuit of examples/circuit_simple.qasm,
# transcribed as DATA (gate name, qubits); matrices below follow the definitions of
# hybridq/gate/gate.py:141-147,213-220,297-329 (H, CZ, T, sqrt-X, sqrt-Y).
# ---------------------------------------------------------------------------------
_S2 = 1.0 / np.sqrt(2.0)
GATE_MATRICES = {
    'h': np.array([[1, 1], [1, -1]], dtype=np.complex128) * _S2,
    't': np.array([[1, 0], [0, np.exp(0.25j * np.pi)]], dtype=np.complex128),
    'cz': np.diag([1, 1, 1, -1]).astype(np.complex128),
    # X^(1/2), Y^(1/2) up to the global phase the reference uses (gate.py: sqrt_x/sqrt_y)
    'x_1_2': np.array([[1 + 1j, 1 - 1j], [1 - 1j, 1 + 1j]], dtype=np.complex128) / 2,
    'y_1_2': np.array([[1 + 1j, -1 - 1j], [1 + 1j, 1 + 1j]], dtype=np.complex128) / 2,
}


def from_named(gate_list):
    """[(name, qubits), ...] -> [(U, qubits), ...] using GATE_MATRICES."""
    return [(GATE_MATRICES[name], tuple(qs)) for name, qs in gate_list]
