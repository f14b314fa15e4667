"""Static checks on the ORDER of instructions in the gfx950 assembly of the cache-blocked kernels (no GPU needed: hipcc -S).

Instruction counts and register budgets (tests/test_kernel_resources.py) did not show what cost the cache-blocked pass a
quarter of its matrix-core time for three rounds: the compiler sinks LDS reads to their consumers, and the wait in between
puts an LDS round trip in front of every 4-8 MFMAs.  The source now pins the requests ahead of the MFMAs
(blocked_inner_gate_tab); this test keeps it that way -- a compiler or source change that serialises the
loops again shows up here as "reads awaited in front of MFMAs" per MFMA going back up (old loop: 0.13, pipelined: 0.03) --
and keeps the direct first gate free of vector-memory waits behind its stores."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _kernels_of(unit, workdir):
    """{demangled kernel name: assembly text} of one translation unit of csrc/, compiled for gfx950 (no GPU needed)."""
    asm = os.path.join(str(workdir), unit + '.s')
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '--cuda-device-only', '-S',
                           os.path.join(ROOT, 'hybridq_amd', 'csrc', unit + '.hip'), '-o', asm], stderr=subprocess.DEVNULL)
    funcs = re.split(r'\n(?=\s*\.globl\s)', open(asm).read())
    named = [(m.group(1), f) for f in funcs for m in [re.search(r'\.globl\s+(\S+)', f)] if m]
    dem = subprocess.run(['c++filt'], input='\n'.join(n for n, _ in named), capture_output=True, text=True).stdout.splitlines()
    return {re.sub(r'\(.*$', '', d).replace('void hq::', ''): f for (_, f), d in zip(named, dem)}


@pytest.fixture(scope='module')
def kernels(tmp_path_factory):
    return _kernels_of('hq_apply', tmp_path_factory.mktemp('isa'))


def _instructions(body):
    return [ln.strip() for ln in body.splitlines() if re.match(r'^\s+[a-z]', ln)]


def _awaited_reads(body):
    """LDS vector reads waited for with lgkmcnt(0) before any MFMA was issued behind them, with MFMAs right behind the wait."""
    ins, n = _instructions(body), 0
    for i, ln in enumerate(ins):
        if ln.startswith(('ds_read_b128', 'ds_read_b64 ')):
            j = i + 1
            while j < len(ins) and j < i + 6 and not ins[j].startswith(('v_mfma', 's_waitcnt', 's_cbranch', 's_barrier')):
                j += 1
            if j < len(ins) and ins[j].startswith('s_waitcnt') and 'lgkmcnt(0)' in ins[j]:
                k = j + 1
                while k < len(ins) and k < j + 4 and not ins[k].startswith(('v_mfma', 'ds_', 's_cbranch')):
                    k += 1
                n += k < len(ins) and ins[k].startswith('v_mfma')
    return n


@pytest.mark.parametrize('name', ['apply_blocked_kernel<float, 512, true, true, true>', 'apply_blocked_kernel<double, 512, true, true, true>',
                                  'apply_blocked_kernel<float, 1024, true, true, true>', 'apply_blocked_direct_kernel<float, 512>',
                                  'apply_blocked_direct_kernel<double, 512>'])
def test_lds_reads_stay_ahead_of_the_matrix_cores(kernels, name):
    body = kernels[name]
    mfma = sum(ln.startswith('v_mfma') for ln in _instructions(body))
    awaited = _awaited_reads(body)
    assert mfma >= 100 and awaited / mfma <= 0.06, (name, awaited, mfma)


def test_the_old_loops_are_still_what_pipe_off_selects(kernels):
    """PIPE = false (HQ_BLOCKED_PIPE=0) must stay the loop of rounds 2-4a -- it is the reference of the library's self-check
    and of the A/B: reads awaited in front of the MFMAs as the compiler places them (0.13 per MFMA)."""
    body = kernels['apply_blocked_kernel<float, 512, true, true, false>']
    mfma = sum(ln.startswith('v_mfma') for ln in _instructions(body))
    assert mfma >= 100 and _awaited_reads(body) / mfma >= 0.10


@pytest.mark.parametrize('name,min_mfma', [('apply_mfma_big_kernel<double, 7, 0, true, 512, true, true>', 512),
                                           ('apply_mfma_big_kernel<double, 7, 1, true, 512, true, true>', 256),
                                           ('apply_mfma_big_kernel<float, 7, 0, true, 512, true, true>', 512)])
def test_role_kernel_operand_reads_are_base_plus_immediate(kernels, name, min_mfma):
    """k = 6 role kernels: every operand read of the MFMA phase is `ds_read_b128 dst, base offset:imm` on one of at most two
    base registers (tables above 64 KiB: a second base 64 KiB up), one pair-group ahead of its MFMAs (lgkmcnt(2) / (3),
    never 0 inside the phase), and nothing is spilled -- the complex128 instantiation kept per-read address registers alive
    across its column blocks until round 5 (52 B/lane of scratch, no operand pipelining)."""
    ins = _instructions(kernels[name])
    assert sum(ln.startswith('v_mfma') for ln in ins) >= min_mfma
    assert not [ln for ln in ins if ln.startswith('scratch_')]
    reads = [ln for ln in ins if ln.startswith('ds_read_b128')]
    bases = {re.split(r'[,\s]+', ln)[2] for ln in reads}
    assert len(reads) >= min_mfma // 4 and len(bases) <= 2, (len(reads), bases)
    first, last = [i for i, ln in enumerate(ins) if ln.startswith('v_mfma')][0], [i for i, ln in enumerate(ins) if ln.startswith('v_mfma')][-1]
    drained = [ln for ln in ins[first + 8:last] if ln.startswith('s_waitcnt') and 'lgkmcnt(0)' in ln]
    assert len(drained) <= 2, drained


@pytest.mark.parametrize('name', ['apply_blocked_direct_kernel<float, 512>', 'apply_blocked_direct_kernel<double, 512>',
                                  'apply_blocked_direct_kernel<float, 1024>'])
def test_direct_first_gate_never_drains_its_stores(kernels, name):
    """One s_waitcnt vmcnt(0) per tile, in front of the first gate (the prefetch is a tile old there); none between the
    gate's global stores and the next prefetch; the rest are the kernel preamble and the linear store of the last tile."""
    ins = _instructions(kernels[name])
    assert sum(ln.startswith('s_waitcnt') and 'vmcnt(0)' in ln for ln in ins) <= 6, name
    stores = [i for i, ln in enumerate(ins) if ln.startswith('global_store')]
    loads = [i for i, ln in enumerate(ins) if ln.startswith('global_load_dwordx4')]
    gate_stores = [i for i in stores if any(j > i for j in loads)]  # stores with a prefetch behind them = the first gate's
    assert gate_stores
    nxt = min(j for j in loads if j > gate_stores[-1])
    between = ins[gate_stores[-1]:nxt]
    assert not [ln for ln in between if ln.startswith('s_waitcnt') and 'vmcnt' in ln], [ln for ln in between if 'vmcnt' in ln]


def test_default_kernels_are_the_binaries_hardware_has_run(kernels, tmp_path):
    """Every kernel of hq_apply.hip that a DEFAULT run can launch -- per-gate kernels of every width (VALU, matrix-core role
    kernels incl. complex128 k = 6 in its TWOB = false form, tile GEMM with PIPE = false, generic / naive / tile) and the
    cache-blocked family of hq_kernels_blocked_r3.h, and every kernel of hq_swap.hip (swaps, one-pass bit permutations,
    to_complex) -- compiles, instruction for instruction, to what the last commit whose
    device code ran on a GPU compiled to (tests/golden/isa_digests_round3.json, written by `tools/isa_vs_round.py af36621
    --digests`; the whole library: profiles/r06_isa_vs_round3.txt).  A change to shared device code that alters one of these
    binaries fails here: it has to go behind a switch (like PIPE / TWOB / HQ_BLOCKED_R3) until hardware has run it."""
    import hashlib
    import json
    ref = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'isa_digests_round3.json')))['kernels']

    def digest(body):
        ins = []
        for ln in body.splitlines():
            if re.match(r'^\s+[a-z]', ln) and not ln.strip().startswith('.'):
                ins.append(re.sub(r'\.LBB\d+_\d+', '.L', re.sub(r';.*$', '', ln).strip()))
        return hashlib.sha256('\n'.join(ins).encode()).hexdigest()[:32]

    def old_name(n):  # template parameters added since, at the value a default run uses
        if n.startswith('r3::'):
            return n[4:]
        m = re.match(r'apply_gemm_kernel<(.*), false>$', n)
        if m:
            return 'apply_gemm_kernel<%s>' % m.group(1)
        m = re.match(r'apply_mfma_big_kernel<(.*), (true|false)>$', n)
        if m and (m.group(2) == 'false' or not m.group(1).startswith('double, 7')):
            return 'apply_mfma_big_kernel<%s>' % m.group(1)
        return n

    # the swap / permutation / to_complex kernels as well (rows a7, a11-a13 of SURVEY section 8 and the pack pass of the exchange)
    swap_kernels = _kernels_of('hq_swap', tmp_path)
    checked, bad = {}, []
    for name, body in swap_kernels.items():
        fam = name.split('<')[0]
        if fam.startswith('__hip') or fam == 'hq::upload_kernel':
            continue
        m = re.match(r'bitperm_tile_kernel<(.*), (\d+)>$', name)  # (the register-prefetch parameter, always false in use, left in round 5)
        old = 'bitperm_tile_kernel<%s, false, %s>' % m.groups() if m else name
        assert old in ref, (name, old)
        checked[fam] = checked.get(fam, 0) + 1
        if digest(body) != ref[old]:
            bad.append(name)
    assert not bad, bad
    assert checked.get('bitperm_tile_kernel') == 36 and checked.get('swap_lds_kernel') == 8 and checked.get('interleave4_kernel') == 2 and len(checked) >= 6, checked
    checked = {}
    for name, body in kernels.items():
        fam = name.split('<')[0]
        if fam not in ('apply_direct_kernel', 'apply_mfma_kernel', 'apply_mfma_big_kernel', 'apply_gemm_kernel', 'apply_generic_kernel',
                       'apply_naive_kernel', 'apply_mfma_tile_kernel', 'r3::apply_blocked_kernel'):
            continue
        if fam == 'apply_gemm_kernel' and not name.endswith(', false>'):
            continue  # (PIPE = true: opt-in)
        if fam == 'apply_mfma_big_kernel' and name.startswith('apply_mfma_big_kernel<double, 7') and name.endswith(', true>'):
            continue  # (TWOB = true: opt-in)
        old = old_name(name)
        assert old in ref, (name, old)
        checked[fam] = checked.get(fam, 0) + 1
        if digest(body) != ref[old]:
            bad.append(name)
    assert not bad, bad
    assert checked == {'apply_direct_kernel': 34, 'apply_mfma_kernel': 24, 'apply_mfma_big_kernel': 48, 'apply_gemm_kernel': 21, 'apply_generic_kernel': 2,
                       'apply_naive_kernel': 2, 'apply_mfma_tile_kernel': 4, 'r3::apply_blocked_kernel': 7}, checked

