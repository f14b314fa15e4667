"""Device code of rounds 4-5 that no earlier GPU test reaches, in every switch setting, against the ORACLE: the staged
cache-blocked kernel with / without pipelined inner gates and barrier-free wave groups (HQ_BLOCKED_PIPE, HQ_BLOCKED_GROUPS),
apply_blocked_direct_kernel (HQ_BLOCKED_DIRECT=1, the tile movement folded into the first gate of a pass -- k <= 4 first
gates) and the 1024-thread kernels for 128 KiB tiles (HQ_BLOCKED_BIG=1).  Every worker also reports the library's own
bit-for-bit cross-check of these variants against the round-2 kernels (hq_blocked_selfcheck): it must have run and found
nothing.  Timings: tools/r6_second.sh, bench.py's blocked_variants."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


_ROUND2_LOOPS = {}


def _worker(**env):
    """{case: result} of tests/direct_gpu_worker.py under `env`; the library's own cross-check of the kernel variants (the
    first passes of the process, bit for bit against the round-2 kernels) must have run and found nothing."""
    e = dict(os.environ, **env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'direct_gpu_worker.py')], env=e, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    res = json.loads(r.stdout.strip().splitlines()[-1])
    chk = res.pop('selfcheck')
    assert 'disagree' not in r.stderr, r.stderr[-2000:]
    if any(env.get(k, d) != '0' for k, d in (('HQ_BLOCKED_PIPE', '0'), ('HQ_BLOCKED_GROUPS', '0'), ('HQ_BLOCKED_DIRECT', '0'), ('HQ_BLOCKED_BIG', '0'))):
        assert chk['runs'] >= 1 and chk['failures'] == 0, chk
    for ct, r_ in res.items():  # the parity statement: every setting against the oracle
        assert r_['err_vs_oracle'] <= r_['tol'], (ct, env, r_)
    return res


@pytest.mark.parametrize('env', [dict(HQ_BLOCKED_PIPE='0', HQ_BLOCKED_GROUPS='0'), dict(HQ_BLOCKED_PIPE='1', HQ_BLOCKED_GROUPS='0'),
                                 dict(HQ_BLOCKED_PIPE='0', HQ_BLOCKED_GROUPS='1'), dict(HQ_BLOCKED_PIPE='1', HQ_BLOCKED_GROUPS='1')],
                         ids=lambda e: 'pipe%s_groups%s' % (e['HQ_BLOCKED_PIPE'], e['HQ_BLOCKED_GROUPS']))
def test_blocked_staged_variants_against_the_oracle(torch_cuda, capsys, env):
    """The staged cache-blocked kernel in its four settings -- the loops of round 2 (HQ_BLOCKED_PIPE=0 HQ_BLOCKED_GROUPS=0:
    the ones GPUTEST_r02 saw = the default), pipelined inner gates, barrier-free wave groups, both -- each against the
    oracle on a depth-16 circuit, several tiles per workgroup; all four agree bit for bit (same arithmetic, same order)."""
    res = _worker(HQ_BLOCKED_GRID='256', **env)
    with capsys.disabled():
        print(f'\n  {env}: {res}')
    if env == dict(HQ_BLOCKED_PIPE='0', HQ_BLOCKED_GROUPS='0'):  # (first parameter set)
        _ROUND2_LOOPS.update(res)
    base = _ROUND2_LOOPS or _worker(HQ_BLOCKED_GRID='256', HQ_BLOCKED_PIPE='0', HQ_BLOCKED_GROUPS='0')
    for ct, r in res.items():
        assert r['repeatable'] and r['direct_passes'] == 0, (ct, r)
        assert r['sha'] == base[ct]['sha'], (ct, env)


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


def test_blocked_128k_tiles_on_the_device(torch_cuda, capsys):
    """128 KiB tiles (2^14 complex64 / 2^13 complex128 amplitudes) on one 1024-thread workgroup per CU, staged and with the
    direct first gate, four and sixteen tiles per workgroup, against the per-gate kernels; two runs bit-identical."""
    for direct in ('0', '1'):
        for grid in ('0', '64'):
            res = _worker(HQ_BLOCKED_BIG='1', HQ_BLOCKED_DIRECT=direct, HQ_BLOCKED_GRID=grid)
            with capsys.disabled():
                print(f'\n  HQ_BLOCKED_DIRECT={direct} HQ_BLOCKED_GRID={grid}: {res}')
            for ct, r in res.items():
                assert r['err_vs_per_gate'] <= r['tol'] and r['repeatable'], (ct, direct, grid, r)
            assert res['complex64 inner_max=3']['passes_1024_threads'] >= 1, res
            if direct == '1':
                assert res['complex64 inner_max=3']['direct_passes'] >= 1, res


def test_wide_kernels_both_loop_forms_against_the_oracle(torch_cuda, capsys):
    """k = 5..10 (role kernel with the operand table in LDS: apply_mfma_big_kernel; tile GEMM: apply_gemm_kernel), both
    precisions, several position patterns, PER CALL against the oracle (north_star's literal bar; k >= 7 in float32: the
    rounding model of one 2^(k+1)-term accumulation, as in test_gpu_parity.py: wide_tol) -- once with the library's
    defaults (the K loop / operand reads hardware has run: HQ_GEMM_PIPE=0, HQ_BIG_TWOBASE=0) and once with the operand-ahead
    forms of rounds 4-5 (HQ_GEMM_PIPE=1, HQ_BIG_TWOBASE=1: compiled, ISA-checked and emulated only until a GPU runs this
    test).  The two forms move requests, not arithmetic: bit-identical results.  Reference: /root/reference/include/U.h:123-202."""
    res = {}
    for name, env in (('defaults', dict(HQ_GEMM_PIPE='0', HQ_BIG_TWOBASE='0')), ('operand_ahead', dict(HQ_GEMM_PIPE='1', HQ_BIG_TWOBASE='1'))):
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'wide_gpu_worker.py')], env=dict(os.environ, **env), capture_output=True,
                           text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        res[name] = json.loads(r.stdout.strip().splitlines()[-1])
        for case, v in res[name].items():
            assert v['err_vs_oracle'] <= v['tol'], (name, case, v)
            assert v['literal_bar_met'] or ('complex64' in case and int(case.split('k=')[1].split()[0]) >= 7), (name, case, v)
    assert res['defaults'].keys() == res['operand_ahead'].keys() and len(res['defaults']) >= 22
    assert not any('twobase=1' in v['kernel'] for v in res['defaults'].values())
    assert any('twobase=1' in v['kernel'] for v in res['operand_ahead'].values()) and any('twobase=0' in v['kernel'] for v in res['defaults'].values())
    for case, v in res['defaults'].items():
        assert v['sha'] == res['operand_ahead'][case]['sha'], case
    with capsys.disabled():
        worst = max(res['operand_ahead'].items(), key=lambda kv: kv[1]['err_vs_oracle'] / kv[1]['tol'])
        print(f"\n  {len(res['defaults'])} cases x 2 loop forms against the {worst[1]['oracle']} oracle, bit-identical between the forms; worst: {worst[0]} {worst[1]['err_vs_oracle']:.2e}")
