"""Minimal reader for HybridQ-flavoured QASM (the gate-per-line format of
``examples/circuit_simple.qasm`` / ``examples/circuit.qasm``; full parser:
hybridq/extras/io/qasm.py:240): ``name q0 [q1 ...] [param ...]`` per line, ``#`` comments,
an optional leading line with the number of qubits.  Returns a plain ``[(U, qubits)]``
circuit for :func:`hybridq_amd.simulation.simulate`.

Gate matrices follow hybridq/gate/gate.py:127-348 (fixed gates; SQRT_* and P/T through
scipy's sqrtm / fractional power like the reference; rotations exp(-i r P / 2),
property.py:676) and the aliases of gate.py:351-365.  ``#@`` extension blocks (qubit maps,
powers, tags, explicit matrices) are not supported and raise."""
import numpy as np
from scipy.linalg import fractional_matrix_power, sqrtm

_X = np.array([[0, 1], [1, 0]], dtype=np.complex128)
_Y = np.array([[0, -1j], [1j, 0]], dtype=np.complex128)
_Z = np.array([[1, 0], [0, -1]], dtype=np.complex128)
_SWAP = np.array([[1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], dtype=np.complex128)
_ISWAP = np.array([[1, 0, 0, 0], [0, 0, 1j, 0], [0, 1j, 0, 0], [0, 0, 0, 1]], dtype=np.complex128)

FIXED = {
    'H': (1, np.array([[1, 1], [1, -1]], dtype=np.complex128) / np.sqrt(2)),
    'X': (1, _X), 'Y': (1, _Y), 'Z': (1, _Z),
    'P': (1, sqrtm(_Z)), 'T': (1, fractional_matrix_power(_Z, 0.25)),
    'SQRT_X': (1, sqrtm(_X)), 'SQRT_Y': (1, sqrtm(_Y)),
    'ZZ': (2, np.diag([1, -1, -1, 1]).astype(np.complex128)),
    'CZ': (2, np.diag([1, 1, 1, -1]).astype(np.complex128)),
    'CX': (2, np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 0, 1], [0, 0, 1, 0]], dtype=np.complex128)),
    'SWAP': (2, _SWAP), 'ISWAP': (2, _ISWAP),
    'SQRT_SWAP': (2, sqrtm(_SWAP)), 'SQRT_ISWAP': (2, sqrtm(_ISWAP)),
}
ALIASES = {'ID': 'I', 'S': 'P', 'Z_1_2': 'P', 'SQRT_Z': 'P', 'CNOT': 'CX', 'X_1_2': 'SQRT_X',
           'Y_1_2': 'SQRT_Y', 'FS': 'FSIM'}


def _rot(P):
    from scipy.linalg import expm
    return lambda r: expm(-1j * float(r) * P / 2)


PARAM = {
    'RX': (1, 1, _rot(_X)), 'RY': (1, 1, _rot(_Y)), 'RZ': (1, 1, _rot(_Z)),
    'R_PI_2': (1, 1, lambda phi: np.array([[1, -1j * np.exp(-1j * float(phi))],
                                            [-1j * np.exp(1j * float(phi)), 1]]) / np.sqrt(2)),
    'U3': (1, 3, lambda t, p, l: np.array(
        [[np.cos(float(t) / 2), -np.exp(1j * float(l)) * np.sin(float(t) / 2)],
         [np.exp(1j * float(p)) * np.sin(float(t) / 2), np.exp(1j * (float(l) + float(p))) * np.cos(float(t) / 2)]])),
    'CPHASE': (2, 1, lambda p: np.diag([1, 1, 1, np.exp(1j * float(p))])),
    'FSIM': (2, 2, lambda t, p: np.array([[1, 0, 0, 0], [0, np.cos(float(t)), -1j * np.sin(float(t)), 0],
                                          [0, -1j * np.sin(float(t)), np.cos(float(t)), 0],
                                          [0, 0, 0, np.exp(-1j * float(p))]])),
}


def from_qasm(text):
    """Parse `text` into ``[(U, qubits), ...]`` (identity gates are kept as explicit matrices)."""
    gates = []
    first = True
    for ln, raw in enumerate(text.splitlines(), 1):
        line = raw.strip()
        if line.startswith('#@'):
            raise NotImplementedError(f'line {ln}: #@ extension blocks are not supported by this reader')
        if not line or line.startswith('#'):
            continue
        tok = line.split()
        if first and len(tok) == 1 and tok[0].isdigit():
            first = False
            continue  # number of qubits
        first = False
        name = tok[0].upper()
        name = ALIASES.get(name, name)
        args = tok[1:]
        if name == 'I':
            qs = tuple(int(q) for q in args)
            gates.append((np.eye(1 << len(qs), dtype=np.complex128), qs))
        elif name in FIXED:
            k, U = FIXED[name]
            if len(args) != k:
                raise ValueError(f'line {ln}: {name} takes {k} qubit(s)')
            gates.append((U, tuple(int(q) for q in args)))
        elif name in PARAM:
            k, npar, gen = PARAM[name]
            if len(args) != k + npar:
                raise ValueError(f'line {ln}: {name} takes {k} qubit(s) and {npar} parameter(s)')
            gates.append((np.asarray(gen(*args[k:]), dtype=np.complex128), tuple(int(q) for q in args[:k])))
        else:
            raise ValueError(f"line {ln}: gate '{tok[0]}' is not supported")
    return gates
