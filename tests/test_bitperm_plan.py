"""CPU test of the one-pass bit permutation: the REAL host planner (csrc/hq_bitperm.h) drives a host emulation of
bitperm_tile_kernel's index arithmetic (tools/bitperm_emul.hip: every tile / thread / iteration, LDS slots, swizzle)
for random permutations, bit reversals, evictions, rotations and in-place low-bit swaps, 4- and 8-byte elements."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which('hipcc') is None and not os.path.exists('/opt/rocm/bin/hipcc'), reason='needs hipcc')
def test_bitperm_planner_and_index_arithmetic(tmp_path):
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    exe = str(tmp_path / 'bitperm_emul')
    res = subprocess.run([hipcc, '-std=c++17', '-O1', '-Wno-unused-value', os.path.join(ROOT, 'tools', 'bitperm_emul.hip'), '-o', exe],
                         capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-2000:]
    run = subprocess.run([exe], capture_output=True, text=True)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-2000:]
    assert ' 0 failed' in run.stdout and ' 0 skipped' in run.stdout, run.stdout
    import re
    m = re.search(r'split mode: (\d+) cases ran', run.stdout)  # 16 / 15 moved bits in place through the register-held quarter
    assert m and int(m.group(1)) >= 8, run.stdout
