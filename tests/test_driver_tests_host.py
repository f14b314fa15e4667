"""Driver-level `-m gpu` tests (tests/test_gpu_parity.py, test_gpu_round2.py) run a second time without a GPU on the numpy
test double of the device (conftest.numpy_device).  The bodies are the GPU tests' own: reference protocol vs simulate(),
initial states, the FunctionalGate branch, expectation values, cache-blocked schedules, simplification, wide
measurements, the schedule choice.  On the double they check the host side (what is planned and issued); on the device
the same statements check the kernels."""
import inspect
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import test_gpu_parity as P  # noqa: E402
import test_gpu_round2 as R2  # noqa: E402

TWINS = [(P, 'test_simulate_matches_reference_protocol', dict(ct='complex64')),
         (P, 'test_simulate_matches_reference_protocol', dict(ct='complex128')),
         (P, 'test_simulate_initial_states', {}), (P, 'test_simulate_functional_gate_branch', {}),
         (P, 'test_expectation_value', dict(ct='complex128')), (P, 'test_simulate_blocked_matches_oracle', {}),
         (P, 'test_simulate_simplify_like_reference', {}), (R2, 'test_measure_and_projection_wide', {}),
         (R2, 'test_evolution_hip_chooses_a_schedule', {})]


@pytest.mark.parametrize('mod,name,kw', TWINS, ids=[t[1] + ('-' + t[2]['ct'] if t[2] else '') for t in TWINS])
def test_on_the_double(numpy_device, oracle_port, monkeypatch, mod, name, kw):
    fn = getattr(mod, name)
    args = {}
    for p in inspect.signature(fn).parameters:
        args[p] = {'torch_cuda': None, 'oracle_port': oracle_port, 'monkeypatch': monkeypatch}[p] if p not in kw else kw[p]
    fn(**args)
