"""The reference's own tests of the path (tests/test_gpu_upstream_mirrors.py) run a second time WITHOUT a GPU, on the numpy
test double of the device (conftest.numpy_device): same test bodies, so what is checked here is everything above the C ABI
-- flattening of container gates, identity circuits and initial states, stochastic sampling, MessageGate isolation --
while the `-m gpu` run checks the same statements on the device."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import test_gpu_upstream_mirrors as mirrors  # noqa: E402


@pytest.fixture
def host(numpy_device, monkeypatch):
    import hybridq_amd.simulation as sim

    def to_complex(a, b, out):  # the library's host-pointer path needs a device; the double interleaves in numpy
        out[...] = np.asarray(a) + 1j * np.asarray(b)
        return out
    monkeypatch.setattr(sim.core, 'to_complex', to_complex)
    return numpy_device


@pytest.mark.parametrize('t', ['float32', 'float64', 'float128'])
def test_utils__to_complex(host, t):
    mirrors.test_utils__to_complex(None, t)


@pytest.mark.parametrize('n_qubits', [16, 20])
@pytest.mark.parametrize('alphabet', ['01', '0+', '01+-'], ids=['1a', '1b', '2'])
def test_simulation_1__initialize_state(host, n_qubits, alphabet):
    mirrors.test_simulation_1__initialize_state(None, n_qubits, alphabet)


def test_simulation_2__tuple(host):
    mirrors.test_simulation_2__tuple(None, 0)


def test_simulation_2__message(host):
    mirrors.test_simulation_2__message(None, 0)


def test_simulation_2__stochastic(host):
    mirrors.test_simulation_2__stochastic(None)


def test_container_gates_against_the_reference_itself(host):
    mirrors.test_container_gates_against_the_reference_itself(None)


def test_prepare_state_api(host):
    mirrors.test_prepare_state_api(None)
