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

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _workers():
    from conftest import _usable_cpus
    return max(2, min(_usable_cpus(), 16))


@pytest.mark.timeout(4200)  # (HQ_EMU_FULL=1: a quarter of an hour; the inner run has its own per-test timeout)
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


CONTRACT_KEYS = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
                 'dtype', 'data', 'config', 'roofline', 'cpu_baseline')


def _check_contract(line):
    for key in CONTRACT_KEYS + ('roofline_plain_placement',):
        assert key in line, key
    assert line['n_gpus'] == 1 and line['dtype'] == 'f32' and line['scaling'] == 'weak' and line['vs_baseline'] is None
    assert 'workload' in line['config'] and 'model' not in line['config']
    r = line['roofline']
    assert r['bound'] == 'hbm' and r['unit'] == 'GB/s' and r['peak'] == 8000.0 and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-12
    rp = line['roofline_plain_placement']  # the same kernels on a state from torch's allocator
    assert rp['kernel'] == r['kernel'] and rp['frac'] > 0 and rp['launches'] >= 1, rp
    assert line['cpu_baseline']['kind'] in ('reference', 'port') and line['cpu_baseline']['cores'] >= 1


def test_bench_line_end_to_end_against_the_emulated_library():
    """bench.py itself -- the file the driver runs for the round's record -- on a small state against the emulation.  The line
    is printed TWICE: right after the timed region + roofline + cpu_baseline (before any extra leg runs) and, complete, as the
    last line; both carry the contract fields.  Every block of the complete line is produced (no `*_error` keys); the
    config-4 / config-5 legs and the parity block carry what DESIGN section 6 says; the cache-blocked leg and the A/B of the
    opt-in kernel variants ran in processes of their own.  The numbers mean nothing; the code paths are the real ones."""
    import json
    env = dict(os.environ, PYTHONPATH=ROOT)
    for var in ('HQ_HIP_LIBRARY', 'HQ_BLOCKED_PIPE', 'HQ_BLOCKED_GROUPS', 'HQ_BLOCKED_DIRECT', 'HQ_BLOCKED_BIG', 'HQ_GEMM_PIPE', 'HQ_BIG_TWOBASE'):
        env.pop(var, None)
    cmd = [sys.executable, os.path.join(ROOT, 'tests', 'emu', 'run_emulated.py'), 'bench.py', '--qubits', '14', '--depth', '3',
           '--steps', '1', '--warmup', '1', '--parity-qubits', '10', '--leg-parity-qubits', '10', '--cpu-seconds', '0.5', '--variants-min-qubits', '14']
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1800, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [json.loads(ln) for ln in out.stdout.strip().splitlines() if ln.startswith('{')]
    assert len(lines) == 2, len(lines)
    head, line = lines
    _check_contract(head)
    _check_contract(line)
    assert head['line'].startswith('headline') and line['line'] == 'complete'
    for key in CONTRACT_KEYS:  # the complete line repeats the headline's fields unchanged
        assert head[key] == line[key], key
    assert not [k for k in head if k in ('blocked', 'fused', 'per_k', 'aux', 'parity_check', 'blocked_variants')]
    assert not [k for k in line if k.endswith('_error')], [k for k in line if k.endswith('_error')]
    for key in ('parity_check', 'cfg4_dense_k34', 'cfg5_noisy_dm', 'blocked', 'blocked_no_fusion', 'fused', 'per_k', 'aux', 'extras'):
        assert key in line, key
    assert line['extras']['skipped_for_budget'] == [] and line['extras']['used_seconds'] <= line['extras']['budget_seconds']
    for leg in ('cfg4_dense_k34', 'cfg5_noisy_dm'):
        assert line[leg]['roofline']['kernel'] and line[leg]['gate_apps_per_s'] > 0 and line[leg]['parity_small_n']['pass'] is True
    # the library's defaults are the kernels hardware has run: the blocked leg launched the round-2 loops, no self-check needed
    assert line['blocked']['kernel'].endswith('pipe=0') and line['blocked']['selfcheck'] == {'runs': 0, 'failures': 0, 'pipe': False, 'groups': False, 'direct': False, 'big': False}
    assert line['blocked']['process'].startswith('subprocess') and len(line['blocked']['ms_per_step_runs']) == 3 and line['blocked']['blocked_passes'] >= 1
    bv = line['blocked_variants']  # the opt-in kernel switches of rounds 4-5, one subprocess each
    assert set(bv) == {'pipe', 'groups', 'pipe_groups', 'direct', 'direct_groups', 'big_tiles', 'big_tiles_direct', 'low_bits_minus_1'} and not [k for k, v in bv.items() if 'error' in v], bv
    assert all(len(v['ms_per_step']) == 3 and v['passes'] >= 1 and v['selfcheck']['failures'] == 0 for v in bv.values()), bv
    assert bv['pipe']['selfcheck']['runs'] >= 1 and bv['pipe']['kernel'].endswith('pipe=1') and bv['direct']['selfcheck']['pipe'] is True  # (direct implies the pipelined gates)
    # the LAST key of the complete line is a compact summary (records keep the tail of stdout): headline, both placements, the
    # cache-blocked step and every variant, the parity verdict
    assert list(line)[-1] == 'summary' and line['summary']['errors'] == [] and line['summary']['parity_check']['pass'] is True
    assert set(line['summary']['blocked_variants_ms_per_step']) == set(bv) and line['summary']['roofline_frac_plain_placement'] == line['roofline']['plain_placement_frac']
    pc = line['parity_check']
    assert pc['pass'] is True and pc['literal_bar_depth'] == pc['literal_bar_depth_of'] and len(pc['prefixes']) >= 8
    assert 'l2_rel_diff_per_gate' in pc and 'reference_vs_f64_leaves_bar_after' in pc


def test_bench_line_survives_a_kill_after_the_timed_region():
    """`kill -9` of the whole process group the moment the first line is out (= any time after the timed region): what was
    printed is one parseable JSON line with every contract field, roofline and cpu_baseline included (VERDICT r05 next #1a).
    A zero budget (--extras-seconds 0) skips every extra and still prints both lines."""
    import json
    import signal
    env = dict(os.environ, PYTHONPATH=ROOT)
    env.pop('HQ_HIP_LIBRARY', None)
    base = [sys.executable, os.path.join(ROOT, 'tests', 'emu', 'run_emulated.py'), 'bench.py', '--qubits', '13', '--depth', '2',
            '--steps', '1', '--warmup', '1', '--cpu-seconds', '0.3', '--parity-qubits', '10', '--leg-parity-qubits', '10']
    proc = subprocess.Popen(base, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, cwd=ROOT, start_new_session=True)
    try:
        first = ''
        while not first.startswith('{'):
            first = proc.stdout.readline()
            assert first or proc.poll() is None, 'bench.py ended without a line'
        os.killpg(proc.pid, signal.SIGKILL)
    finally:
        proc.wait(timeout=60)
    assert proc.returncode == -signal.SIGKILL
    _check_contract(json.loads(first))
    out = subprocess.run(base + ['--extras-seconds', '0'], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [json.loads(ln) for ln in out.stdout.strip().splitlines() if ln.startswith('{')]
    assert len(lines) == 2 and lines[1]['line'] == 'complete'
    _check_contract(lines[1])
    assert set(lines[1]['extras']['skipped_for_budget']) >= {'fused', 'blocked', 'per_k', 'aux', 'parity_check'} and 'blocked' not in lines[1]



@pytest.mark.timeout(4200)
@pytest.mark.parametrize('transport', ['p2p', 'rccl'])
def test_multi_rank_gpu_tests_against_the_emulated_library(transport):
    """The multi-rank `-m gpu` tests -- 2 and 4 ranks on the real HipBackend, the sharded simulate() / dm.simulate() API,
    `bench.py --gpus 8` for both sharded workloads -- as real PROCESSES against the emulation, once per transport of the
    library's own exchange (hq_exchange_*):
      p2p   the pack kernel stores straight into the other ranks' planes, mapped through HIP IPC (emulated over POSIX
            shared memory: tests/emu/hip_emu.cpp);
      rccl  communicator from a unique id, grouped ncclSend / ncclRecv of one chunk per peer and plane around the pack
            (tests/emu/rccl_emu.cpp stands in for librccl: unix sockets between the processes).
    The tests assert the transport that ran, states against the oracle, and bit-identity with the host-staged path.  On the
    RCCL transport this includes test_sharded_one_rank_per_gpu_over_rccl (2 / 4 / 8 processes): the test an 8-GPU box would run
    first, here against the emulated librccl."""
    env = dict(os.environ, HQ_EMU_GPU_SUITE='1', HQ_EMU_QUICK='0', HQ_SHARD_TRANSPORT=transport, PYTHONPATH=ROOT)
    env.pop('HQ_HIP_LIBRARY', None)
    env.pop('HQ_RCCL_LIBRARY', None)
    cmd = [sys.executable, '-m', 'pytest', os.path.join(ROOT, 'tests', 'test_gpu_dist.py'), '-m', 'gpu', '-q', '-p', 'no:cacheprovider',
           '--timeout', '900', '-x']
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=3600, cwd=ROOT)
    tail = out.stdout[-3000:]
    assert out.returncode == 0, tail + out.stderr[-2000:]
    # 11 tests on ranks that share the device + the one-rank-per-GPU RCCL test at 2 / 4 / 8 processes (RCCL transport only)
    assert re.search(r'\b14 passed' if transport == 'rccl' else r'\b11 passed, 3 skipped', tail), tail
