"""Round-4 device code that no earlier GPU test reaches: apply_blocked_direct_kernel (HQ_BLOCKED_DIRECT=1, the tile
movement folded into the first gate of a cache-blocked pass).  The staged kernel is the default until the two have been
timed against each other (tools/ab_round4.sh); this test makes sure the opt-in path is RIGHT on the device."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(**env):
    e = dict(os.environ, **env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'direct_gpu_worker.py')], env=e, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_blocked_direct_pass_on_the_device(torch_cuda, capsys):
    """The cache-blocked schedule of a depth-16 circuit (n = 24 complex64 / 23 complex128) with the direct first gate,
    one tile and eight tiles per workgroup (HQ_BLOCKED_GRID), against the per-gate kernels within the rounding model;
    two runs bit-identical; most passes eligible."""
    staged = _worker(HQ_BLOCKED_DIRECT='0')
    for grid in ('0', '256'):
        res = _worker(HQ_BLOCKED_DIRECT='1', HQ_BLOCKED_GRID=grid)
        with capsys.disabled():
            print(f'\n  HQ_BLOCKED_GRID={grid}: {res}')
        for ct, r in res.items():
            assert r['err_vs_per_gate'] <= r['tol'], (ct, grid, r)
            assert r['repeatable'], (ct, grid)
            assert staged[ct]['direct_passes'] == 0 and staged[ct]['err_vs_per_gate'] <= staged[ct]['tol']
        assert res['complex64 inner_max=3']['direct_passes'] >= res['complex64 inner_max=3']['passes'] // 2, res
