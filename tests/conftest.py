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


@pytest.fixture
def numpy_device(monkeypatch):
    """The numpy test double of the device (tests/device_double.py) under hybridq_amd.simulation for one test."""
    import device_double
    return device_double.install(monkeypatch.setattr)
