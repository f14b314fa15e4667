"""Density-matrix front-end: the counterpart of ``hybridq.dm.circuit.simulation.simulate``
(hybridq/dm/circuit/simulation.py:118-274) for the evolution path (BASELINE config 5).

An n-qubit density matrix is evolved as a 2n-qubit state vector rho ~ psi (x) psi*:
  * a unitary gate ``(U, qubits)`` becomes ``U`` on the left copies ``(0, q)`` and
    ``conj(U)`` on the right copies ``(1, q)`` (``__transform``, simulation.py:24-51);
  * a channel becomes ONE dense, generally non-unitary 2k-qubit matrix gate on
    ``[(0, q)...] + [(1, q)...]`` whose matrix is the superoperator
    ``sum_i s_i  L_i (x) conj(R_i)`` (``Map.map``, hybridq/dm/gate/property.py:73-137 and
    KrausSuperGate);
  * labels are tuples, ``(0, q) < (1, q)``, so every left qubit sits in the high half of
    the index (``Circuit.all_qubits()`` sorts) and the result reshapes to rho[left, right].
The 2n-qubit circuit then runs through :func:`hybridq_amd.simulation.simulate` unchanged
(fusion included: the reference fuses the super-circuit the same way).
"""
import numpy as np

from .simulation import simulate as _simulate


class Kraus:
    """Channel rho -> sum_i s_i L_i rho R_i^dagger on `qubits` (R = L for a Kraus map)."""

    def __init__(self, ops, qubits, s=None, right_ops=None):
        self.left = [np.asarray(K, dtype=np.complex128) for K in ops]
        self.right = self.left if right_ops is None else [np.asarray(K, dtype=np.complex128) for K in right_ops]
        self.qubits = tuple(qubits)
        self.s = np.ones(len(self.left)) if s is None else np.asarray(s)
        d = 1 << len(self.qubits)
        if any(K.shape != (d, d) for K in self.left + self.right):
            raise ValueError('Kraus operators must be 2^k x 2^k for k = len(qubits)')

    def map(self):
        """Superoperator matrix on (left qubits..., right qubits...)."""
        s = self.s
        if s.ndim == 1:
            return sum(s[i] * np.kron(self.left[i], self.right[i].conj()) for i in range(len(s)))
        return sum(s[i, j] * np.kron(self.left[i], self.right[j].conj())
                   for i in range(s.shape[0]) for j in range(s.shape[1]))


def depolarizing(qubits, p):
    """Global depolarizing channel (1-p) rho + p I/d tr(rho) written with Pauli Kraus operators,
    weights (1 - p (d^2-1)/d^2, p/d^2, ...) like hybridq.noise's GlobalDepolarizingChannel."""
    k = len(qubits)
    paulis = [np.eye(2), np.array([[0, 1], [1, 0]]), np.array([[0, -1j], [1j, 0]]), np.diag([1, -1])]
    ops = [np.eye(1)]
    for _ in range(k):
        ops = [np.kron(a, b) for a in ops for b in paulis]
    d2 = 4**k
    s = np.full(d2, p / d2)
    s[0] = 1 - p * (d2 - 1) / d2
    return Kraus(ops, qubits, s)


def to_statevector_circuit(circuit):
    """SuperCircuit -> 2n-qubit circuit of (U, qubits) with (0, q)/(1, q) labels."""
    out = []
    for g in circuit:
        if isinstance(g, Kraus):
            out.append((g.map(), tuple((0, q) for q in g.qubits) + tuple((1, q) for q in g.qubits)))
        elif hasattr(g, 'map') and hasattr(g, 'qubits') and not isinstance(g, (tuple, list)):
            out.append((np.asarray(g.map()), tuple((0, q) for q in g.qubits) + tuple((1, q) for q in g.qubits)))
        else:
            U, qs = (np.asarray(g[0]), tuple(g[1])) if isinstance(g, (tuple, list)) else (np.asarray(g.matrix()), tuple(g.qubits))
            out.append((U, tuple((0, q) for q in qs)))
            out.append((U.conj(), tuple((1, q) for q in qs)))
    return out


def simulate(circuit, initial_state=None, **kwargs):
    """rho after `circuit` as an array of shape (2,)*2n (left indices first), through the
    evolution core.  `initial_state`: '01+-' string for the n qubits (or 2n), or psi as an
    array of 2^n amplitudes (rho0 = psi (x) psi, simulation.py:259-261).  ``devices=N`` shards the
    2n-qubit state over the N ranks of the torch.distributed job (BASELINE config 5)."""
    circuit = list(circuit)
    sv = to_statevector_circuit(circuit)
    lq = sorted({q for _, qs in sv for (side, q) in qs if side == 0})
    rq = sorted({q for _, qs in sv for (side, q) in qs if side == 1})
    nl, nr = len(lq), len(rq)
    if isinstance(initial_state, str):
        s = initial_state * (nl + nr) if len(initial_state) == 1 else initial_state
        if not (len(s) == nl + nr or (lq == rq and len(s) == nl)):
            raise ValueError("'initial_state' has the wrong number of qubits.")
        initial_state = s + s if len(s) == nl else s
    elif initial_state is not None:
        st = np.asarray(initial_state)
        if st.size == 1 << nl and lq == rq:
            st = np.kron(st.ravel(), st.ravel())
        if st.size != 1 << (nl + nr):
            raise ValueError("'initial_state' has the wrong number of qubits.")
        initial_state = st
    qubits = [(0, q) for q in lq] + [(1, q) for q in rq]
    return _simulate(sv, initial_state=initial_state, qubits=qubits, **kwargs)
