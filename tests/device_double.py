"""A TEST DOUBLE of the device for the host-side tests: the state lives in numpy and every call the driver issues -- apply_U,
apply_blocked, probabilities, project, vdot, norm2 -- is applied with the oracle's index arithmetic
(oracle.evolution.apply_gate_numpy).  Everything above the C ABI runs for real.  TEST INFRASTRUCTURE ONLY: the product has no
such path (without the HIP library's device simulate() raises)."""
import numpy as np


class _Planes:
    """Stands for one plane tensor; both planes of a state share the owner."""

    def __init__(self, owner):
        self.owner = owner


def install(setattr_fn):
    """Replace the device under hybridq_amd.simulation by the numpy double; `setattr_fn(obj, name, value)` does the patching
    (pytest's monkeypatch.setattr in tests, plain setattr in a worker process).  Returns (call log, oracle module)."""
    import numpy as np
    import oracle
    from oracle.evolution import _initial, apply_gate_numpy
    import hybridq_amd.simulation as sim
    log = {'apply_U': 0, 'apply_blocked': 0, 'states': 0}

    class State:
        def __init__(self, qubits, complex_type='complex64', initial_state=None, device=None, placement='plain'):
            self.qubits, self.n = list(qubits), len(qubits)
            self.complex_type = np.dtype(complex_type)
            self.map = {q: self.n - x - 1 for x, q in enumerate(self.qubits)}
            self.psi = _initial(initial_state, self.n, np.complex128)  # the double keeps float64: only the calls are on trial
            self.re = self.im = _Planes(self)
            self.planes = [self.re, self.im]
            self.device = None
            log['states'] += 1

        def apply_functional(self, gate):
            if callable(getattr(gate, 'apply_device', None)):  # as EvolutionState.apply_functional does
                gate.apply_device(self)
                return
            order = tuple(self.qubits)
            shape = (2,) + (2,) * self.n

            def fetch():
                return np.stack([self.psi.real, self.psi.imag]).reshape(shape)

            def store(new_psi):
                new_psi = np.asarray(new_psi).reshape(2, -1)
                self.psi = new_psi[0] + 1j * new_psi[1]
            sim._apply_host_functional(gate, order, fetch, store, shape, np.float64)  # the product's own host branch

        def to_numpy(self):
            return self.psi.astype(self.complex_type)

        def to_complex(self):  # EvolutionState.to_complex returns a device tensor: .cpu().numpy() gives the amplitudes
            arr = self.psi.astype(self.complex_type)
            from types import SimpleNamespace
            return SimpleNamespace(cpu=lambda: SimpleNamespace(numpy=lambda: arr))

    def apply_U(re, im, U, pos, n):
        st = re.owner
        assert im.owner is st and n == st.n and len(set(int(p) for p in pos)) == len(pos) and all(0 <= int(p) < n for p in pos)
        st.psi = apply_gate_numpy(st.psi, np.asarray(U, dtype=np.complex128), [int(p) for p in pos])
        log['apply_U'] += 1

    def apply_blocked(re, im, tile_pos, gates, n):
        tile = set(int(p) for p in tile_pos)
        assert len(tile) == len(tile_pos) and list(tile_pos) == sorted(tile)
        for U, pos in gates:
            assert set(int(p) for p in pos) <= tile and 1 <= len(pos) <= 4  # what hq_apply_blocked_* demands
            apply_U(re, im, U, pos, n)
            log['apply_U'] -= 1
        log['apply_blocked'] += 1

    setattr_fn(sim, 'EvolutionState', State)
    setattr_fn(sim, '_torch', lambda: None)
    setattr_fn(sim.core, 'apply_U', apply_U)
    setattr_fn(sim.core, 'apply_blocked', apply_blocked)
    setattr_fn(sim.core, 'use_torch_stream', lambda: None)
    setattr_fn(sim.core, 'sync', lambda: None)
    setattr_fn(sim.core, 'vdot', lambda are, aim, bre, bim: complex(np.vdot(are.owner.psi, bre.owner.psi)))

    def _outcome_index(st, pos):  # outcome bit j <-> index bit pos[j]
        idx = np.arange(1 << st.n)
        out = np.zeros_like(idx)
        for j, p in enumerate(pos):
            out |= ((idx >> int(p)) & 1) << j
        return out

    def probabilities(re, im, pos, n):
        st = re.owner
        assert len(pos) <= 10  # the marginal kernel's limit
        return np.bincount(_outcome_index(st, pos), weights=np.abs(st.psi)**2, minlength=1 << len(pos))

    def project(re, im, pos, state, scale=1.0, n=None):
        st = re.owner
        st.psi = np.where(_outcome_index(st, pos) == int(state), st.psi * scale, 0)

    setattr_fn(sim.core, 'probabilities', probabilities)
    setattr_fn(sim.core, 'project', project)
    setattr_fn(sim.core, 'norm2', lambda re, im: float(np.sum(np.abs(re.owner.psi)**2)))
    return log, oracle


