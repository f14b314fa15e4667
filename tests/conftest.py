import os
import sys


def _usable_cpus():
    """Affinity mask capped by the cgroup CPU quota (GPU boxes: 256 hardware threads, 16-CPU quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


# The CPU oracle (OpenMP) and numpy's BLAS would spin one thread per hardware thread; under a CFS
# quota that freezes the whole process for most of every 100 ms period.  Must precede numpy's import.
for _var in ('OMP_NUM_THREADS', 'OPENBLAS_NUM_THREADS'):
    os.environ.setdefault(_var, str(_usable_cpus()))

import pytest  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


def pytest_sessionstart(session):
    """Build artefacts are git-ignored: (re)build the HIP library and the CPU oracle when they are
    missing or stale, so that a fresh checkout can run the suite directly (hipcc cross-compiles
    gfx950 without a GPU; on the GPU box the prebuilt files arrive with the snapshot)."""
    import subprocess
    from hybridq_amd import build as hq_build
    hq_build.build(force=False)
    if not os.path.exists(os.path.join(ROOT, 'oracle', 'libhq_oracle.so')):
        subprocess.check_call(['make', '-s', '-C', os.path.join(ROOT, 'oracle'), 'port'])
    if not os.path.exists(os.path.join(ROOT, 'oracle', '_ref', 'hybridq.so')) and os.path.exists('/root/reference/include'):
        subprocess.call(['make', '-s', '-C', os.path.join(ROOT, 'oracle'), 'ref'])


@pytest.fixture(scope='session')
def oracle_port():
    import subprocess
    import oracle
    if not os.path.exists(os.path.join(ROOT, 'oracle', 'libhq_oracle.so')):
        subprocess.check_call(['make', '-C', os.path.join(ROOT, 'oracle'), 'port'])
    return oracle.load_port()


@pytest.fixture(scope='session')
def oracle_ref():
    import oracle
    if not oracle.have_ref():
        pytest.skip('oracle/_ref not built (needs /root/reference at build time)')
    return oracle.load_ref()


@pytest.fixture(scope='session')
def torch_cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail('GPU test selected but no HIP device is visible')
    return torch


# ---------------------------------------------------------------------------------------------------------------------
# A TEST DOUBLE of the device for the host-side tests (tests/test_host_driver_loop.py, test_upstream_mirrors_host.py): the
# state lives in numpy and every call the driver issues -- apply_U, apply_blocked, probabilities, project, ... -- is
# applied with the oracle's index arithmetic.  Everything above the C ABI runs for real.  The product has no such path.
# ---------------------------------------------------------------------------------------------------------------------
class _Planes:
    """Stands for one plane tensor; both planes of a state share the owner."""

    def __init__(self, owner):
        self.owner = owner


@pytest.fixture
def numpy_device(monkeypatch):
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
            host = np.stack([self.psi.real, self.psi.imag]).reshape((2,) + (2,) * self.n)
            new_psi, new_order = gate.apply(psi=host, order=order)
            assert tuple(new_order) == order
            new_psi = np.asarray(new_psi).reshape(2, -1)
            self.psi = new_psi[0] + 1j * new_psi[1]

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

    monkeypatch.setattr(sim, 'EvolutionState', State)
    monkeypatch.setattr(sim, '_torch', lambda: None)
    monkeypatch.setattr(sim.core, 'apply_U', apply_U)
    monkeypatch.setattr(sim.core, 'apply_blocked', apply_blocked)
    monkeypatch.setattr(sim.core, 'use_torch_stream', lambda: None)
    monkeypatch.setattr(sim.core, 'sync', lambda: None)
    monkeypatch.setattr(sim.core, 'vdot', lambda are, aim, bre, bim: complex(np.vdot(are.owner.psi, bre.owner.psi)))

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

    monkeypatch.setattr(sim.core, 'probabilities', probabilities)
    monkeypatch.setattr(sim.core, 'project', project)
    monkeypatch.setattr(sim.core, 'norm2', lambda re, im: float(np.sum(np.abs(re.owner.psi)**2)))
    return log, oracle


