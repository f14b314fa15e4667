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

#: HQ_EMU_GPU_SUITE=1: run the `-m gpu` tests on a box WITHOUT a GPU against the host emulation of the HIP library
#: (tests/emu: the same planners and kernel bodies compiled for the host, lane-exact wave operations) with CPU torch
#: tensors standing for device tensors (tests/emu/fake_cuda.py).  tests/test_emu_gpu_suite.py does this in a subprocess
#: as part of the CPU suite; it is evidence that the code paths and index arithmetic are right, not a GPU run.
EMU_SUITE = os.environ.get('HQ_EMU_GPU_SUITE') == '1'
if EMU_SUITE:
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import emu_util  # noqa: E402
    os.environ['HQ_HIP_LIBRARY'] = emu_util.emu_library()
    os.environ['HQ_EMU_HOST_IS_DEVICE'] = '1'
    os.environ['HQ_RCCL_LIBRARY'] = emu_util.emu_rccl_library()  # tests/emu/rccl_emu.cpp: grouped send / recv between processes
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'emu'))
    import fake_cuda  # noqa: E402
    fake_cuda.install()


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


#: -m gpu tests that cannot (or need not) run against the host emulation, with the reason; everything else does
EMU_SKIP = (
    ('test_gpu_fullsize.py', 'BASELINE-size states (n = 30...) are hours of emulation'),
    ('c_abi_demo_without_python', 'a C program linked against the real HIP runtime'),
    ('c_abi_state_demo_without_python', 'a C program linked against the real HIP runtime'),
    ('many_tiles_per_workgroup[complex', 'n = 23...25: minutes of emulation per case'),
    ('swap_many_tiles_per_workgroup', 'n = 25, 26'),
    ('one_pass_tiles[', 'n = 25, 26'),
    ('one_pass_in_place[', 'n = 25, 26'),
    ('simulation_large_like_reference[2', 'n = 20, 22 with 600 gates (n = 16 runs)'),
    ('test_gpu_depth_parity.py', 'n = 24 depth-40 circuits (the same statements at n = 16: smoke(), test_emu_kernels.py)'),
    ('simple_qasm_trace_and_circuit', 'n = 24'),
    ('large_state_comes_back', 'n >= 25'),
    ('large_array_initial_state', 'n >= 25'),
    ('host_functional_gate_on_a_large_state', 'n >= 25'),
    ('to_numpy_never_holds', 'n >= 25 and the real allocator statistics'),
    ('test_dm_1__simulation_1', '12-qubit density matrices = 24-qubit states'),
    ('prepare_state_mixed_large', 'n = 26'),
    ('auto_dispatch_follows_the_measured_rule', 'n = 30 planes'),
    ('qasm_text_to_gpu', 'n = 24'),
    ('stream_switch_is_ordered', 'real streams'),
)


#: HQ_EMU_QUICK=1 (what tests/test_emu_gpu_suite.py runs inside the CPU suite): also leave out what takes more than ~20 s
#: of emulation; the full emulated run is a few minutes on 8 cores (profiles/r04_emulated_gpu_suite.txt)
EMU_SLOW = ('apply_U_mfma_kernels', 'randomized_differential', 'apply_U_gemm_kernel', 'simulate_blocked_matches_oracle',
            'evolution_hip_chooses', 'compiled_program', 'exchange_pack_one_pass', 'simulation_large_like_reference',
            'state_allocator_behind', 'restore_order_hip_backend', 'simulate_matches_reference_protocol',
            'permute_bits_many_moved_bits', 'initialize_state[', 'guard_bands', 'two_ranks_one_gpu', 'eight_ranks_sharing',
            'sharded_api', 'test_gpu_determinism')


#: order of the `-m gpu` files under `-x`: the per-call oracle tests (which pin U::apply and swap_array) first, then the
#: reference-recorded vectors, the circuit-level parity files, the full-size and sharded runs, and last the files whose
#: subject is the newest kernel code, so that a fault there cannot hide the oracle tests behind it
GPU_FILE_ORDER = ('test_gpu_parity.py', 'test_gpu_golden.py', 'test_gpu_round2.py', 'test_gpu_round3.py',
                  'test_gpu_depth_parity.py', 'test_gpu_fullsize.py', 'test_gpu_dist.py', 'test_gpu_determinism.py',
                  'test_gpu_round4.py', 'test_gpu_upstream_mirrors.py', 'test_gpu_zz_guard_bands.py')


def _file_rank(item):
    name = os.path.basename(str(item.fspath))
    return GPU_FILE_ORDER.index(name) if name in GPU_FILE_ORDER else -1   # CPU files keep their place in front


def pytest_collection_modifyitems(config, items):
    items.sort(key=_file_rank)   # stable: the order inside a file is untouched
    if not EMU_SUITE:
        return
    quick = os.environ.get('HQ_EMU_QUICK') == '1'
    for item in items:
        if quick and any(pat in item.nodeid for pat in EMU_SLOW):
            item.add_marker(pytest.mark.skip(reason='HQ_EMU_QUICK: more than ~20 s of emulation (runs in the full emulated suite)'))
            continue
        for pat, why in EMU_SKIP:
            if pat in item.nodeid:
                item.add_marker(pytest.mark.skip(reason=f'not under emulation: {why}'))
                break


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
