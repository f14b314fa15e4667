"""Minimal reader for HybridQ-flavoured QASM (the gate-per-line format of
``examples/circuit_simple.qasm`` / ``examples/circuit.qasm``; full parser:
hybridq/extras/io/qasm.py:240): ``name q0 [q1 ...] [param ...]`` per line, ``#`` comments,
an optional leading line with the number of qubits.  Returns a plain ``[(U, qubits)]``
circuit for :func:`hybridq_amd.simulation.simulate`.

Gate matrices follow hybridq/gate/gate.py:127-348 (fixed gates; SQRT_* and P/T through
scipy's sqrtm / fractional power like the reference; rotations exp(-i r P / 2),
property.py:676) and the aliases of gate.py:351-365, plus the ``#@`` extension blocks of the reference's writer
(qubit maps, powers, conj / T, explicit matrices; tags are dropped)."""
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


def _parse_extensions(lines):
    """Split the text into (extensions, gate line) records.  HybridQ's QASM extensions are comment lines that
    start with ``#@`` (hybridq/extras/io/qasm.py:57-75, written by to_qasm :160-233): ``#@ key =`` followed by a
    JSON value spread over ``#@`` lines (``qubits``: QASM index -> label, ``tags``, ``U``: the matrix of a ``matrix``
    gate), ``#@ power = p``, ``#@ conj``, ``#@ T``.  Extensions apply to the next gate line; ``qubits`` is global."""
    import json
    pending, records, qmap = {}, [], None
    i = 0
    while i < len(lines):
        raw = lines[i].strip()
        i += 1
        if not raw.startswith('#@'):
            if raw and not raw.startswith('#'):
                records.append((pending, raw, i))
                pending = {}
            continue
        body = raw[2:].strip()
        if body in ('conj', 'T'):
            pending[body] = True
            continue
        if '=' not in body:
            raise ValueError(f'line {i}: unknown extension {body!r}')
        key, val = (x.strip() for x in body.split('=', 1))
        if not val:  # JSON value on the following #@ lines: read until it parses
            buf = ''
            while i < len(lines) and lines[i].strip().startswith('#@'):
                buf += lines[i].strip()[2:] + '\n'
                i += 1
                try:
                    val = json.loads(buf)
                    break
                except json.JSONDecodeError:
                    val = None
            if val is None:
                raise ValueError(f'extension {key!r}: malformed JSON value')
        else:
            try:
                val = json.loads(val)
            except json.JSONDecodeError as e:
                raise ValueError(f'line {i}: extension {key!r}: malformed JSON value ({e.msg})') from None
        if key == 'qubits':
            qmap = {int(k): _label_from_text(v) for k, v in val.items()}
        else:
            pending[key] = val
    return records, qmap


def _label_from_text(v):
    """Qubit label as the writer spelled it: digit strings are ints, '(…)' spellings of tuples (the density-matrix
    front-end labels qubits (side, q)) are read back with ast.literal_eval, everything else stays a string."""
    v = str(v)
    if v.lstrip('-').isdigit():
        return int(v)
    if v.startswith('(') and v.endswith(')'):
        import ast
        try:
            lab = ast.literal_eval(v)
            if isinstance(lab, tuple):
                return lab
        except (ValueError, SyntaxError):
            pass
    return v


def from_qasm(text):
    """Parse `text` into ``[(U, qubits), ...]`` (identity gates are kept as explicit matrices).  Supports the
    ``#@`` extensions of HybridQ's writer: the qubits map (labels instead of QASM indices), ``power`` (matrix
    power through scipy like gate/property.py:433), ``conj``, ``T``, ``U`` for ``matrix`` gates; ``tags`` are
    read and dropped.  A gate without qubits (``.``) cannot be simulated and raises."""
    records, qmap = _parse_extensions(text.splitlines())
    gates = []
    first = True
    for ext, line, ln in records:
        tok = line.split()
        if first and len(tok) == 1 and tok[0].isdigit():
            first = False
            continue  # number of qubits
        first = False
        name = tok[0].upper()
        name = ALIASES.get(name, name)
        args = tok[1:]
        if args and args[0] == '.':
            raise ValueError(f"line {ln}: gate '{tok[0]}' has no qubits ('.'): nothing to simulate")

        def labels(idx):
            out = tuple(int(q) for q in idx)
            return tuple(qmap.get(q, q) for q in out) if qmap else out

        if name == 'I':
            qs = labels(args)
            U = np.eye(1 << len(qs), dtype=np.complex128)
        elif name == 'MATRIX':
            if 'U' not in ext:
                raise ValueError(f"line {ln}: 'matrix' needs a '#@ U =' block")
            U = np.asarray([[complex(str(x).replace(' ', '')) for x in row] for row in ext['U']], dtype=np.complex128)
            qs = labels(args)
            if U.shape != (1 << len(qs),) * 2:
                raise ValueError(f'line {ln}: matrix shape {U.shape} does not fit {len(qs)} qubit(s)')
        elif name in FIXED:
            k, U = FIXED[name]
            if len(args) != k:
                raise ValueError(f'line {ln}: {name} takes {k} qubit(s)')
            qs = labels(args)
        elif name in PARAM:
            k, npar, gen = PARAM[name]
            if len(args) != k + npar:
                raise ValueError(f'line {ln}: {name} takes {k} qubit(s) and {npar} parameter(s)')
            U = np.asarray(gen(*args[k:]), dtype=np.complex128)
            qs = labels(args[:k])
        else:
            raise ValueError(f"line {ln}: gate '{tok[0]}' is not supported")
        U = np.asarray(U, dtype=np.complex128)
        if 'power' in ext and float(ext['power']) != 1:
            p = float(ext['power'])
            U = np.linalg.matrix_power(U, int(p)) if p == int(p) else fractional_matrix_power(U, p)
        if ext.get('conj'):
            U = U.conj()
        if ext.get('T'):
            U = U.T
        gates.append((np.ascontiguousarray(U), qs))
    return gates


def to_qasm(gates, qubits_map=None):
    """Write ``[(U, qubits), ...]`` in the reference's extended QASM (hybridq/extras/io/qasm.py:160-233): the number
    of qubits, a ``#@ qubits =`` map from QASM index to label, and every gate as a ``matrix`` line preceded by its
    ``#@ U =`` block (a plain (U, qubits) pair carries no gate name).  ``from_qasm(to_qasm(c))`` returns the same
    matrices and labels; the reference's ``from_qasm`` reads the text as MATRIX gates."""
    import json
    gates = [(np.asarray(U), tuple(qs)) for U, qs in gates]
    labels = sorted({q for _, qs in gates for q in qs}, key=lambda q: (type(q).__name__, q))
    # the text form of a label must identify it: '3' and 3 would both come back as the int 3, and a string spelled like a
    # tuple would come back as one
    spelled = {}
    for q in labels:
        if isinstance(q, str) and _label_from_text(q) != q:
            raise ValueError(f'qubit label {q!r} would not survive the round trip (it reads back as {_label_from_text(q)!r})')
        if spelled.setdefault(str(q), q) != q:
            raise ValueError(f'qubit labels {spelled[str(q)]!r} and {q!r} have the same spelling')
    if qubits_map is None:
        qubits_map = {q: i for i, q in enumerate(labels)}
    lines = [str(len(labels)), '#@ qubits = ']
    lines += ['#@ ' + x for x in json.dumps({str(i): str(q) for q, i in sorted(qubits_map.items(), key=lambda kv: kv[1])},
                                             indent=2).split('\n')]
    for U, qs in gates:
        if U.shape != (1 << len(qs),) * 2:
            raise ValueError(f'matrix shape {U.shape} does not fit {len(qs)} qubit(s)')
        rows = [[str(complex(x)) for x in row] for row in U]
        lines.append('#@ U = ')
        lines += ['#@ ' + x for x in json.dumps(rows, indent=2).split('\n')]
        lines.append('matrix ' + ' '.join(str(qubits_map[q]) for q in qs))
    return '\n'.join(lines) + '\n'
