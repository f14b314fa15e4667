"""The API-level golden tests (tests/test_gpu_golden.py: outputs recorded from the reference's own Python API by
tests/golden/make_golden.py) run a second time without a GPU on the numpy test double of the device: the host side of
this package -- labels and axis order, '01+-' initial states, fusion, the dm front-end's super-circuit, Projection /
Measure conventions and RNG use, expectation_value -- must reproduce what the REFERENCE returned."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import test_gpu_golden as golden  # noqa: E402


@pytest.mark.parametrize('name', ['test_dm_front_end', 'test_api_simulate_mixed_initial_state', 'test_api_projection_and_measure',
                                  'test_api_expectation_value', 'test_functional_gate_streams_against_the_reference_record'])
def test_reference_recorded_outputs_on_the_double(numpy_device, name):
    getattr(golden, name)(None)
