"""The `-m gpu` tests on a box WITHOUT a GPU, against the host emulation of the HIP library (tests/emu).

What this is: the five translation units of libhq_hip.so compiled as plain C++ for the host against a shim of the HIP
programming model -- the same host planners, the same kernel bodies, threads as fibers, wave-level operations (MFMA,
shuffles, readfirstlane) exchanged with the gfx950 lane layouts, LDS addressed as the kernels address it -- and CPU torch
tensors standing for device tensors.  It executes the real dispatch, the real index arithmetic and the real arithmetic
order of every kernel the tests reach; it says nothing about timing, occupancy or the memory system.
What this is not: a GPU run, and not a product path -- `hybridq_amd` has no CPU fallback and never loads the emulation.

The quick subset below runs inside the CPU suite; HQ_EMU_FULL=1 runs everything that fits (n <= 18), a few minutes on 8
cores: `HQ_EMU_FULL=1 python -m pytest tests/test_emu_gpu_suite.py -s`."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _workers():
    from conftest import _usable_cpus
    return max(2, min(_usable_cpus(), 16))


def test_gpu_tests_pass_against_the_emulated_library():
    full = os.environ.get('HQ_EMU_FULL') == '1'
    env = dict(os.environ, HQ_EMU_GPU_SUITE='1', HQ_EMU_QUICK='0' if full else '1', PYTHONPATH=ROOT)
    env.pop('HQ_HIP_LIBRARY', None)
    cmd = [sys.executable, '-m', 'pytest', os.path.join(ROOT, 'tests'), '-m', 'gpu', '-q', '-n', str(_workers()), '-p', 'no:cacheprovider',
           '--timeout', '900', '-x']
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=3600, cwd=ROOT)
    tail = out.stdout[-3000:]
    assert out.returncode == 0, tail + out.stderr[-2000:]
    m = re.search(r'(\d+) passed', tail)
    assert m and int(m.group(1)) >= (150 if full else 95), tail
    if full or os.environ.get('HQ_EMU_SHOW') == '1':
        print(tail)
