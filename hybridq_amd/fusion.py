"""Gate fusion for the evolution driver: the counterpart of ``utils.compress`` +
``utils.to_matrix_gate`` (hybridq/circuit/utils.py:467-685, :419-464) as used by
``_simulate_evolution`` (hybridq/circuit/simulation/simulation.py:436-454, default
``compress=4``, :314).

Same greedy rule as the reference, restated on plain ``(U, qubits)`` pairs:
for every gate, walk the already-built layers from the newest to the oldest;
remember a layer as merge target when ``|q ∪ cq| <= max(max_n_qubits, |cq|, |q|)``
(utils.py:626-630); keep walking while the gate shares no qubit with the layer or
commutes with the layer's matrix (:633-646); stop at the first layer it cannot pass.
Merge into the oldest remembered layer, else open a new one.  A layer becomes ONE dense
gate on the sorted union of its qubits (``Circuit.all_qubits()`` sorts,
circuit/circuit.py:406-451), matrix = product of its gates in order with ``qubits[0]`` as
most significant index bit.

Why it matters on the GPU: every k <= 4 gate costs the same HBM pass (~3 ms at n=30),
so fusing 900 one- and two-qubit gates into ~110 four-qubit gates is an ~8x end-to-end
win; the matrix-core kernel absorbs the 16x larger matrices for free.
"""
import numpy as np


_EMBED_INDEX = {}  # (k, gate axes) -> (rows, cols) of the embedded matrix, cached
_BLAS = []


def single_thread_blas():
    """Context manager: numpy's BLAS on ONE thread while a planner runs.  The planners multiply thousands of 4x4 ... 64x64
    matrices with Python in between; a threaded BLAS wakes its pool for each of them (measured here: 0.7 ms per call once
    the workers have gone to sleep, planning the n = 30 benchmark circuit 0.7-1.3 s instead of 0.05 s; worse on a GPU
    box with 256 hardware threads).  Needs threadpoolctl; without it nothing changes."""
    if not _BLAS:
        try:
            from threadpoolctl import ThreadpoolController
            _BLAS.append(ThreadpoolController())
        except Exception:  # noqa: BLE001 -- not installed, or a BLAS it cannot introspect
            _BLAS.append(None)
    if _BLAS[0] is None:
        import contextlib
        return contextlib.nullcontext()
    return _BLAS[0].limit(limits=1, user_api='blas')


def _embed(U, qs, Q):
    """Matrix of gate (U, qs) on the ordered qubit list Q (Q[0] = most significant bit).  One fancy assignment
    through cached index arrays: the planners call this tens of thousands of times on tiny matrices, where
    numpy's tensordot / moveaxis bookkeeping cost more than the arithmetic."""
    k = len(Q)
    kg = len(qs)
    U = np.asarray(U, dtype=np.complex128).reshape(1 << kg, 1 << kg)
    if kg == k and list(qs) == list(Q):
        return U.copy()
    axes = tuple(Q.index(q) for q in qs)
    key = (k, axes)
    idx = _EMBED_INDEX.get(key)
    if idx is None:
        gate_bits = [k - 1 - a for a in axes]  # index bit of the embedded matrix carrying gate bit kg-1-j
        rest_bits = [b for b in range(k) if b not in gate_bits]
        g = np.arange(1 << kg)
        gpart = np.zeros(1 << kg, dtype=np.int64)
        for j, b in enumerate(gate_bits):  # gate index bit kg-1-j (qs[0] most significant) -> matrix bit b
            gpart |= ((g >> (kg - 1 - j)) & 1) << b
        r = np.arange(1 << (k - kg))
        rpart = np.zeros(1 << (k - kg), dtype=np.int64)
        for j, b in enumerate(rest_bits):
            rpart |= ((r >> j) & 1) << b
        rows = rpart[:, None, None] | gpart[None, :, None]
        cols = rpart[:, None, None] | gpart[None, None, :]
        idx = _EMBED_INDEX[key] = (rows, cols)
    M = np.zeros((1 << k, 1 << k), dtype=np.complex128)
    M[idx[0], idx[1]] = U[None, :, :]
    return M


def _sorted_union(a, b):
    s = set(a) | set(b)
    try:
        return sorted(s)
    except TypeError:
        return sorted(s, key=lambda q: (type(q).__name__, q))


#: the tolerance of ``PowerMatrixGate.commutes_with``: its `atol` argument is not what decides -- the comparison is
#: ``np.allclose(P1, P2, atol=1e-5)`` whatever the caller passes (hybridq/gate/property.py:573), for compress and simplify alike
_COMMUTE_ATOL = 1e-5


#: widest union of two gates whose commutator is still evaluated (three 2^(2 |Q|) matrices)
MAX_COMMUTE_QUBITS = 12


def commute(U1, q1, U2, q2, atol=None, exact=False):
    """True if the two gates commute (trivially when they share no qubit); mirrors
    ``PowerMatrixGate.commutes_with`` (hybridq/gate/property.py:498-580), whose tolerance is fixed (`atol` is accepted and,
    as there, not used).  ``exact``: only gates that commute to rounding count (``True``: 1e-12; a float: that absolute
    and relative tolerance -- see :func:`exact_tolerance`) -- for this package's own schedules, which have no reference
    behaviour to reproduce: reordering gates that commute only within 1e-5 moves the final state by as much (seen live:
    6e-6 on a noisy circuit, in the reference's compress=0 run just as here)."""
    if not set(q1) & set(q2):
        return True
    Q = _sorted_union(q1, q2)
    if len(Q) > MAX_COMMUTE_QUBITS:  # not tested = no reordering (csrc/hq_plan.hip: kMaxCommuteQubits)
        return False
    A, B = _embed(U1, q1, Q), _embed(U2, q2, Q)
    # one row of the commutator first: generic gates that share a qubit fail right here (the verdict is the full
    # test's: it needs EVERY entry to pass), for O(D^2) instead of two D^3 products -- which the planners otherwise pay
    # a thousand times per circuit, each a threaded BLAS call on a matrix too small for it
    if exact:
        a_tol = r_tol = 1e-12 if exact is True else float(exact)
    else:
        a_tol, r_tol = _COMMUTE_ATOL, 1e-5
    r_ab, r_ba = A[0] @ B, B[0] @ A
    if not (np.abs(r_ab - r_ba) <= a_tol + r_tol * np.abs(r_ba)).all():
        return False
    AB, BA = A @ B, B @ A
    return bool((np.abs(AB - BA) <= a_tol + r_tol * np.abs(BA)).all())  # np.allclose's test without its bookkeeping


def exact_tolerance(gates, complex_type=None):
    """The "commutes to rounding" tolerance for a gate list: 1e-12 for double-precision matrices; 8 eps(float32) = 9.5e-7
    as soon as one matrix is given in single precision -- the commutator of two such gates that commute mathematically is
    ~sqrt(D) eps = 2-5e-7, and the stricter bound would silently stop the planner from sliding them, while
    anything above stays below the 1e-6 bar of complex64 parity.  A complex128 evolution (`complex_type`) keeps 1e-12
    whatever the precision the matrices arrive in: its bar is 1e-12, and reordering a pair whose commutator is ~1e-6 would
    move the state by as much."""
    if complex_type is not None and np.dtype(complex_type) == np.dtype('complex128'):
        return 1e-12
    single = any(np.asarray(U).dtype in (np.dtype('complex64'), np.dtype('float32'), np.dtype('float16')) for U, _ in gates)
    return 8 * float(np.finfo(np.float32).eps) if single else 1e-12


class Opaque:
    """A gate WITHOUT a matrix inside a gate list (the reference's FunctionalGate family: Projection, Measure, user
    functions).  ``qubits``: the labels it acts on, or None when it declares none (``Gate('fn', n_qubits=...)``).  The
    planners never merge it and never look inside; what they need is how the reference's walks treat such an element:
      compress (circuit/utils.py:583-669, called with skip_compression=[FunctionalGate], simulation.py:441): it always
        opens a layer of its own; a later matrix gate can never be merged into that layer, slides ACROSS it when they
        share no qubit (:636-637; matrix commutation switched on) and stops at it otherwise (``commutes_with(None)`` raises,
        :640-646); a layer whose gate has no qubits stops every gate (``all_qubits()`` raises, :618-622);
      simplify (insert_from_left, :166-208): it slides to the right past gates of at most max_n_qubits_matrix qubits that
        share no qubit with it (``inv`` / ``commutes_with`` raise: no cancellation, no matrix test), matrix gates slide past
        it under the same condition, and one without qubits is inserted at the far left and stops everything."""
    __slots__ = ('obj', 'qubits')

    def __init__(self, obj, qubits):
        self.obj = obj
        self.qubits = None if qubits is None else tuple(qubits)


class _Layer:
    __slots__ = ('gates', 'qubits', 'U', 'compress', 'has_matrix', 'opaque')

    def __init__(self, U, qs, compress=True, has_matrix=True):
        self.gates = [(U, qs)]
        self.qubits = _sorted_union(qs, ())
        self.U = _embed(U, qs, self.qubits)
        self.opaque = None
        self.compress = compress  # False: nothing may be merged into this layer (utils.py:615-617)
        # the reference keeps a layer's matrix for its commutation tests only while the layer spans at most
        # max_n_qubits_matrix qubits, and never gets it back afterwards (utils.py:660-669); the fused matrix `U` itself is
        # always kept here -- it is the output
        self.has_matrix = has_matrix

    @classmethod
    def of_opaque(cls, item):
        """The layer an Opaque element opens: never merged into, no matrix."""
        L = cls.__new__(cls)
        L.gates, L.qubits, L.U, L.compress, L.has_matrix, L.opaque = [item], item.qubits, None, False, False, item
        return L

    def merge(self, U, qs, has_matrix=True):
        self.gates.append((U, qs))
        self.has_matrix = self.has_matrix and has_matrix
        Q = _sorted_union(self.qubits, qs)
        # the new gate acts AFTER everything already in the layer
        self.U = _embed(U, qs, Q) @ _embed(self.U, self.qubits, Q)
        self.qubits = Q


def _build_layers(gates, max_n_qubits, use_matrix_commutation, max_n_qubits_matrix, atol, exclude_qubits=(), exact_commutation=False):
    layers = []
    exclude = set(exclude_qubits or ())
    for item in gates:
        if isinstance(item, Opaque):  # never merged anywhere: a layer of its own at the end (see Opaque)
            layers.append(_Layer.of_opaque(item))
            continue
        U, qs = item
        q = set(qs)
        can = not (q & exclude)  # gates on `exclude_qubits` are never compressed (utils.py:615-617)
        merge_to = len(layers)
        # the gate's own matrix takes part in commutation tests only if the gate is small enough (utils.py:606-611)
        gate_matrix = use_matrix_commutation and len(q) <= max_n_qubits_matrix
        for i in range(len(layers) - 1, -1, -1):
            L = layers[i]
            if L.opaque is not None:
                if L.qubits is None or not use_matrix_commutation or (q & set(L.qubits)):
                    break
                continue  # disjoint from a functional gate: slide across it (utils.py:636-637)
            cq = set(L.qubits)
            if can and L.compress and len(q | cq) <= max(max_n_qubits, len(cq), len(q)):  # utils.py:626-630
                merge_to = i
            if use_matrix_commutation:  # utils.py:633-646
                if not (q & cq):
                    continue
                # both matrices must exist; the size of their UNION is not limited (commutes_with builds the product)
                if gate_matrix and L.has_matrix and commute(U, qs, L.U, L.qubits, atol, exact=exact_commutation):
                    continue
            break
        if merge_to < len(layers):
            L = layers[merge_to]
            L.merge(U, qs, has_matrix=gate_matrix and len(q | set(L.qubits)) <= max_n_qubits_matrix)  # utils.py:660-669
        else:
            layers.append(_Layer(U, qs, can, has_matrix=gate_matrix))
    return layers


def compress(gates, max_n_qubits=4, use_matrix_commutation=True, max_n_qubits_matrix=10, atol=1e-7,
             exclude_qubits=None):
    """Group `gates` ([(U, qubits), ...]) into layers like hybridq's ``utils.compress``.
    Returns a list of layers, each a list of the original gates in application order."""
    gates = [g if isinstance(g, Opaque) else (np.asarray(g[0]), tuple(g[1])) for g in gates]
    if max_n_qubits is None or max_n_qubits <= 0:  # utils.py:565-566
        return [[g] for g in gates]
    return [L.gates for L in _build_layers(gates, max_n_qubits, use_matrix_commutation, max_n_qubits_matrix, atol,
                                           exclude_qubits)]


def to_matrix_gate(layer, complex_type='complex64'):
    """One dense gate for a layer: (U, qubits) with qubits = sorted union
    (utils.to_matrix_gate, hybridq/circuit/utils.py:419-464)."""
    Q = []
    for _, qs in layer:
        Q = _sorted_union(Q, qs)
    M = np.eye(1 << len(Q), dtype=np.complex128)
    for U, qs in layer:
        M = _embed(U, qs, Q) @ M
    return M.astype(complex_type), tuple(Q)


def _layer_matrix_like_reference(layer_gates, qubits):
    """The matrix the REFERENCE computes for a layer wider than 4 qubits: ``to_matrix_gate`` -> ``utils.matrix`` first
    compresses the layer's own gates to width 4 with the default options (circuit/utils.py:751-759: max_compress = 4), i.e.
    it regroups them with the 1e-5 commutation test, and multiplies the groups in that order.  Where every pair that is
    swapped commutes exactly this is the plain product; where a pair commutes only to 1e-5 (named gates raised to tiny
    powers) it differs from it by as much -- reproduced here so that reference-schedule runs agree with the reference as
    vectors, not to 1e-5."""
    inner = _build_layers(layer_gates, 4, True, 10, _COMMUTE_ATOL)
    M = np.eye(1 << len(qubits), dtype=np.complex128)
    for L in inner:
        M = _embed(L.U, L.qubits, qubits) @ M
    return M


def fuse(gates, max_n_qubits=4, complex_type='complex64', use_matrix_commutation=True,
         max_n_qubits_matrix=10, atol=1e-7, exclude_qubits=None, exact_commutation=False, native=None, reference_matrices=False):
    """compress + to_matrix_gate in one go: the fused gate stream ``_simulate_evolution``
    hands to the core (simulation.py:436-454).  Layer matrices are accumulated in
    complex128 and cast once.  ``native`` (default: whenever the circuit has at most 62 distinct qubits and fused gates stay
    within 10): the same rule behind the C ABI (``hq_plan_fuse``, csrc/hq_plan.hip; 36 -> 4 ms for the 900-gate benchmark
    circuit at width 4); ``native=False``: the Python statement (``_build_layers``).  ``reference_matrices``:
    layers wider than 4 qubits get the matrix the reference's ``to_matrix_gate`` computes for them
    (:func:`_layer_matrix_like_reference`; Python statement only)."""
    gates = [g if isinstance(g, Opaque) else (np.asarray(g[0]), tuple(g[1])) for g in gates]
    mixed = any(isinstance(g, Opaque) for g in gates)  # Opaque entries come back as they are, in their place among the fused gates
    if max_n_qubits is None or max_n_qubits <= 0:
        return [g if isinstance(g, Opaque) else (g[0].astype(complex_type), g[1]) for g in gates]
    if exact_commutation is True:
        exact_commutation = exact_tolerance([g for g in gates if not isinstance(g, Opaque)], complex_type)
    if mixed and native:
        raise ValueError('hq_plan_fuse has no notion of gates without a matrix')
    reference_matrices = bool(reference_matrices) and max_n_qubits > 4
    if (native is None or native) and gates and not mixed and not reference_matrices:
        labels = _sorted_union([q for _, qs in gates for q in qs], ())  # ids follow the order of the labels
        ok = (len(labels) <= 62 and max_n_qubits <= 10 and
              all(len(qs) <= 10 and len(set(qs)) == len(qs) and U.size == 4 ** len(qs) for U, qs in gates))
        if ok:
            from . import core
            ident = {q: i for i, q in enumerate(labels)}
            excl = 0
            for q in (exclude_qubits or ()):
                if q in ident:
                    excl |= 1 << ident[q]
            gk, gids, mats = core.plan_fuse(len(labels), [(U, [ident[q] for q in qs]) for U, qs in gates], max_n_qubits,
                                            use_matrix_commutation, max_n_qubits_matrix, excl,
                                            float(exact_commutation) if exact_commutation else _COMMUTE_ATOL)
            out, po, mo = [], 0, 0
            for kk in gk:
                kk = int(kk)
                d = 1 << kk
                out.append((mats[mo:mo + d * d].reshape(d, d).astype(complex_type), tuple(labels[int(i)] for i in gids[po:po + kk])))
                po += kk
                mo += d * d
            return out
        if native:
            raise ValueError('hq_plan_fuse takes at most 62 distinct qubits and gates of at most 10 qubits')
    layers = _build_layers(gates, max_n_qubits, use_matrix_commutation, max_n_qubits_matrix, atol, exclude_qubits,
                           exact_commutation)
    if reference_matrices:
        for L in layers:
            if L.opaque is None and len(L.qubits) > 4 and len(L.gates) > 1:
                L.U = _layer_matrix_like_reference(L.gates, L.qubits)
    return [L.opaque if L.opaque is not None else (L.U.astype(complex_type), tuple(L.qubits)) for L in layers]


def matrix(gates, order=None, complex_type='complex64'):
    """Dense matrix of a whole circuit: the counterpart of ``hybridq.circuit.utils.matrix``
    (circuit/utils.py:688-807).  Rows/columns are indexed with ``order[0]`` as the most
    significant bit; ``order`` defaults to the sorted qubits of the circuit and must be a
    permutation of them."""
    gates = [(np.asarray(U), tuple(qs)) for U, qs in gates]
    Q = []
    for _, qs in gates:
        Q = _sorted_union(Q, qs)
    if order is not None:
        order = list(order)
        if set(order).difference(Q) or len(order) != len(Q):  # utils.py:748-751
            raise ValueError("'order' must be a valid permutation of indexes in 'Circuit'.")
    else:
        order = Q
    M = np.eye(1 << len(order), dtype=np.complex128)
    for U, qs in gates:
        M = _embed(U, qs, order) @ M
    return np.ascontiguousarray(M.astype(complex_type))


def _inverse_of(U1, q1, U2, q2, atol):
    """True if gate 2 is the inverse of gate 1 (same qubit set, any order):
    ``gate.inv().isclose(_g)`` of hybridq/gate/property.py:447-500."""
    if sorted(map(str, q1)) != sorted(map(str, q2)) or set(q1) != set(q2):
        return False
    Q = _sorted_union(q1, q2)
    A, B = _embed(U1, q1, Q), _embed(U2, q2, Q)
    try:
        return bool(np.allclose(np.linalg.inv(A), B, atol=atol))
    except np.linalg.LinAlgError:
        return False


def simplify(gates, atol=1e-8, use_matrix_commutation=True, max_n_qubits_matrix=10, remove_id_gates=True, native=None):
    """``native`` (default: whenever the circuit has at most 62 distinct qubits and no gate wider than 10): the same
    algorithm behind the C ABI (``hq_plan_simplify``, csrc/hq_plan.hip; 24 -> 1.5 ms for the 900-gate benchmark circuit);
    ``native=False``: the Python statement below, which the live tests hold against the reference itself.

    Counterpart of ``hybridq.circuit.utils.simplify`` (circuit/utils.py:825-866 with
    ``insert_from_left``, :122-208) on ``(U, qubits)`` pairs: drop identity gates, then rebuild the
    circuit from its LAST gate backwards, sliding every gate to the right through the gates it
    commutes with (no shared qubit, or commuting matrices) and cancelling it against the first
    gate that is its inverse.  The circuit's action is unchanged; what changes is the gate
    list the fusion sees (the reference's ``simulate`` runs this by default, simulation.py:304)."""
    gates = [g if isinstance(g, Opaque) else (np.asarray(g[0]), tuple(g[1])) for g in gates]
    if any(isinstance(g, Opaque) for g in gates):
        if native:
            raise ValueError('hq_plan_simplify has no notion of gates without a matrix')
        return _simplify_mixed(gates, atol, use_matrix_commutation, max_n_qubits_matrix, remove_id_gates)
    if native is None or native:
        labels = {}
        for _, qs in gates:
            for q in qs:
                labels.setdefault(q, len(labels))
        fits = len(labels) <= 62 and all(len(qs) <= 10 and len(set(qs)) == len(qs) and np.asarray(U).size == 4 ** len(qs)
                                         for U, qs in gates)
        if fits and gates:
            from . import core
            order = core.plan_simplify(max(1, len(labels)), [(U, [labels[q] for q in qs]) for U, qs in gates], atol,
                                       use_matrix_commutation, max_n_qubits_matrix, remove_id_gates)
            return [gates[int(i)] for i in order]
        if native:
            raise ValueError('hq_plan_simplify takes at most 62 distinct qubits and gates of at most 10 qubits')
    if remove_id_gates:  # utils.py:841-845
        eyes = {}

        def is_identity(U):  # np.allclose(U, eye, atol=atol) without its bookkeeping (900 tiny calls per circuit)
            eye = eyes.get(U.shape[0])
            if eye is None:
                eye = eyes[U.shape[0]] = np.eye(U.shape[0])
            return U.shape == eye.shape and bool((np.abs(U - eye) <= atol + 1e-5 * eye).all())

        gates = [(U, qs) for U, qs in gates if len(qs) > max_n_qubits_matrix or not is_identity(U)]
    new = []  # in circuit order; gates are inserted from the left.  Entries carry their qubit set: most of the time goes
    #           into sliding past gates on other qubits, which needs nothing else
    for U, qs in reversed(gates):
        q = frozenset(qs)
        placed = False
        for p, (V, vs, vset) in enumerate(new):
            if vset == q and _inverse_of(U, qs, V, vs, atol):  # :182-184 (an inverse acts on the same qubits)
                del new[p]
                placed = True
                break
            ok = False
            if len(vs) <= max_n_qubits_matrix:  # :192-195
                ok = not (q & vset)
                if not ok and use_matrix_commutation:
                    ok = commute(U, qs, V, vs, atol)
            if not ok:  # :199-201
                new.insert(p, (U, qs, q))
                placed = True
                break
        if not placed:
            new.append((U, qs, q))
    return [(U, qs) for U, qs, _ in new]


def _simplify_mixed(gates, atol, use_matrix_commutation, max_n_qubits_matrix, remove_id_gates):
    """simplify() on a list that holds Opaque entries (gates without a matrix): the same walk, with the reference's
    treatment of a FunctionalGate on either side of a comparison (see Opaque; insert_from_left, circuit/utils.py:166-208)."""
    def is_identity(U):
        eye = np.eye(U.shape[0])
        return U.shape == eye.shape and bool((np.abs(U - eye) <= atol + 1e-5 * eye).all())

    if remove_id_gates:  # utils.py:841-845: gates that provide no matrix are kept
        gates = [g for g in gates if isinstance(g, Opaque) or len(g[1]) > max_n_qubits_matrix or not is_identity(g[0])]
    new = []  # entries: (U or None, qubits or None, frozenset or None, Opaque or None)
    for g in reversed(gates):
        if isinstance(g, Opaque):
            U, qs, op = None, g.qubits, g
        else:
            (U, qs), op = g, None
        if qs is None:  # no qubits declared: "just append to the left" (:170-173)
            new.insert(0, (None, None, None, op))
            continue
        q = frozenset(qs)
        placed = False
        for p, (V, vs, vset, vop) in enumerate(new):
            if op is None and vop is None and vset == q and _inverse_of(U, qs, V, vs, atol):  # :182-184
                del new[p]
                placed = True
                break
            ok = False
            if vs is not None and len(vs) <= max_n_qubits_matrix:  # :192-195 (an entry without qubits: the test itself raises)
                ok = not (q & vset)
                if not ok and use_matrix_commutation and op is None and vop is None:
                    ok = commute(U, qs, V, vs, atol)
            if not ok:  # :199-201
                new.insert(p, (U, qs, q, op))
                placed = True
                break
        if not placed:
            new.append((U, qs, q, op))
    return [op if op is not None else (U, qs) for U, qs, _, op in new]
