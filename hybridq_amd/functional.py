"""Device-side functional gates: ``Projection`` and ``Measure`` acting on the state in HBM
(counterparts of hybridq/gate/projection.py:25-119 and hybridq/gate/measure.py:25-125, which
are numpy code on the host).  Inside :func:`hybridq_amd.simulation.simulate` they are applied
without the D2H/H2D round trip that generic FunctionalGates need (simulation.py:525-554).

Conventions kept from the reference: ``state`` is a '01' string with one character per
qubit of the gate; probabilities / sampled outcomes are indexed with ``qubits[0]`` as the
MOST significant bit (``_Measure`` transposes the measured axes to the front in the order
given, measure.py:44-50); ``renormalize=True`` rescales the surviving amplitudes to norm 1
(projection: only if the norm exceeds ``atol``, projection.py:58-66)."""
import numpy as np

from . import core

MAX_MARGINAL_QUBITS = 10  # hq_probabilities_* bins 2^k outcomes in LDS, k <= 10


class Projection:
    """Project `qubits` onto the computational-basis `state` ('0'/'1' per qubit)."""

    def __init__(self, state, qubits, renormalize=True, atol=1e-6):
        self.qubits = tuple(qubits)
        self.state = ''.join(str(s) for s in state)
        if len(self.state) != len(self.qubits):
            raise ValueError("'state' is not consistent with 'qubits'.")
        if any(c not in '01' for c in self.state):
            raise ValueError("Only projections to the z-basis are supported at the moment.")
        self.renormalize = renormalize
        self.atol = atol
        self.name = 'PROJECTION'

    def apply_device(self, st):
        """`st`: hybridq_amd.simulation.EvolutionState."""
        pos = [st.map[q] for q in reversed(self.qubits)]  # bit j of the outcome <-> qubits[k-1-j]
        want = int(self.state, 2)
        scale = 1.0
        if self.renormalize:
            if len(pos) <= MAX_MARGINAL_QUBITS:
                p = core.probabilities(st.planes[0], st.planes[1], pos, st.n)[want]
            else:  # wider than the marginal kernel: project first, the surviving weight is the norm
                core.project(st.planes[0], st.planes[1], pos, want, 1.0, st.n)
                p = core.norm2(st.planes[0], st.planes[1])
            norm = np.sqrt(p)
            if norm <= self.atol:  # projection.py:58-66: nothing survives -> all zeros
                core.project(st.planes[0], st.planes[1], pos, want, 0.0, st.n)
                return
            scale = 1.0 / norm
        core.project(st.planes[0], st.planes[1], pos, want, scale, st.n)


class Measure:
    """Sample an outcome of measuring `qubits` and collapse the state onto it."""

    def __init__(self, qubits, renormalize=True, rng=None):
        self.qubits = tuple(qubits)
        self.renormalize = renormalize
        self.rng = rng  # numpy Generator / RandomState; None -> numpy's global state like the reference
        self.outcome = None
        self.name = 'MEASURE'

    def probabilities(self, st):
        pos = [st.map[q] for q in reversed(self.qubits)]
        if len(pos) > MAX_MARGINAL_QUBITS:
            raise ValueError(f'probabilities of more than {MAX_MARGINAL_QUBITS} qubits are not tabulated; '
                             'apply_device() measures wider sets chunk by chunk')
        return core.probabilities(st.planes[0], st.planes[1], pos, st.n)

    def sample(self, probs):
        p = probs / probs.sum()
        if self.rng is None:
            return int(np.random.choice(len(p), p=p))
        return int(self.rng.choice(len(p), p=p))

    def apply_device(self, st):
        pos = [st.map[q] for q in reversed(self.qubits)]
        if len(pos) <= MAX_MARGINAL_QUBITS:
            probs = core.probabilities(st.planes[0], st.planes[1], pos, st.n)
            self.outcome = self.sample(probs)
            scale = 1.0 / np.sqrt(probs[self.outcome]) if self.renormalize else 1.0
            core.project(st.planes[0], st.planes[1], pos, self.outcome, scale, st.n)
            return
        # More qubits than the marginal kernel bins (2^10): measure them chunk by chunk.  The
        # marginal of the next chunk on the state already projected (NOT renormalised) onto the
        # earlier outcomes is the joint probability, so sampling from it normalised is sampling the
        # conditional distribution: the same joint law as one draw over 2^k outcomes.
        outcome, joint = 0, 1.0
        for lo in range(0, len(pos), MAX_MARGINAL_QUBITS):
            chunk = pos[lo:lo + MAX_MARGINAL_QUBITS]
            probs = core.probabilities(st.planes[0], st.planes[1], chunk, st.n)
            o = self.sample(probs)
            joint = probs[o]
            core.project(st.planes[0], st.planes[1], chunk, o, 1.0, st.n)
            outcome |= o << lo
        self.outcome = outcome
        if self.renormalize:
            core.project(st.planes[0], st.planes[1], pos, outcome, 1.0 / np.sqrt(joint), st.n)
