"""Which kernels of HEAD are, instruction for instruction, the kernels an earlier round compiled -- i.e. the code a GPU has run?
Compiles the translation units of csrc/ of a given commit and of the working tree to gfx950 assembly (no GPU needed) and compares the
instruction streams of every kernel both have (labels normalised, comments and directives dropped; template parameters that
were added since with a default are mapped: PIPE = false, TWOB = false / true).
    python tools/isa_vs_round.py af36621 > profiles/r06_isa_vs_round3.txt        (af36621 = "round 3: VERDICT", the last hardware contact)"""
import difflib
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
commit = sys.argv[1] if len(sys.argv) > 1 else 'af36621'


UNITS = ('hq_apply', 'hq_swap', 'hq_state', 'hq_shard', 'hq_core')


def asm_of(src_root, out):
    text = ''
    for unit in UNITS:
        src = os.path.join(src_root, 'hybridq_amd', 'csrc', unit + '.hip')
        if not os.path.exists(src):
            continue
        subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '--cuda-device-only', '-S', src, '-o', out],
                              stderr=subprocess.DEVNULL, cwd=src_root)
        text += '\n' + open(out).read()
    funcs = re.split(r'\n(?=\s*\.globl\s)', text)
    named = [(m.group(1), f) for f in funcs for m in [re.search(r'\.globl\s+(\S+)', f)] if m]
    dem = subprocess.run(['c++filt'], input='\n'.join(n for n, _ in named), capture_output=True, text=True).stdout.splitlines()
    out = {}
    for (_, f), d in zip(named, dem):
        ins = []
        for ln in f.splitlines():
            if re.match(r'^\s+[a-z]', ln) and not ln.strip().startswith('.'):
                ins.append(re.sub(r'\.LBB\d+_\d+', '.L', re.sub(r';.*$', '', ln).strip()))
        out[re.sub(r'\(.*$', '', d).replace('void hq::', '')] = ins
    return out


with tempfile.TemporaryDirectory() as td:
    old_root = os.path.join(td, 'old')
    os.makedirs(old_root)
    tar = subprocess.run(['git', 'archive', commit, 'hybridq_amd/csrc', 'include'], cwd=ROOT, capture_output=True, check=True).stdout
    subprocess.run(['tar', '-x', '-C', old_root], input=tar, check=True)
    old = asm_of(old_root, os.path.join(td, 'old.s'))
    new = asm_of(ROOT, os.path.join(td, 'new.s'))


def old_names(n):
    if re.match(r'apply_blocked_kernel<.*, true>$', n):  # (the fifth parameter was another switch then)
        return []
    m = re.match(r'r3::apply_blocked_kernel<(.*)>$', n)   # hq_kernels_blocked_r3.h: that commit's family under its own namespace
    if m:
        return ['apply_blocked_kernel<%s, false>' % m.group(1), 'apply_blocked_kernel<%s>' % m.group(1)]
    cand = [n]
    m = re.match(r'bitperm_tile_kernel<(.*), (\d+)>$', n)  # (the register-prefetch parameter, always false in use, left in round 5)
    if m:
        cand.append('bitperm_tile_kernel<%s, false, %s>' % m.groups())
    m = re.match(r'apply_gemm_kernel<(\w+), (\d+), (\d+), (\d+), false>', n)
    if m:
        cand.append('apply_gemm_kernel<%s, %s, %s, %s>' % m.groups())
    m = re.match(r'apply_mfma_big_kernel<(.*), (true|false)>$', n)
    if m and (m.group(2) == 'false' or not m.group(1).startswith('double, 7')):  # TWOB matters for complex128 k = 6 only
        cand.append('apply_mfma_big_kernel<%s>' % m.group(1))
    return cand


rows = {'identical': [], 'renamed registers / a few scalar instructions': [], 'different': []}
for n, ins in sorted(new.items()):
    o = next((old[c] for c in old_names(n) if c in old), None)
    if o is None:
        continue
    if o == ins:
        rows['identical'].append(n)
        continue
    strip = lambda seq: [re.sub(r'\b[sv]\d+\b|\b[sv]\[\d+:\d+\]', 'R', x) for x in seq]  # noqa: E731
    ratio = difflib.SequenceMatcher(None, strip(o), strip(ins), autojunk=False).ratio()
    key = 'renamed registers / a few scalar instructions' if ratio >= 0.97 else 'different'
    rows[key].append('%-62s %5d -> %5d instructions, %3d -> %3d MFMA, %d -> %d barriers, similarity %.2f' % (
        n, len(o), len(ins), sum(x.startswith('v_mfma') for x in o), sum(x.startswith('v_mfma') for x in ins),
        sum(x.startswith('s_barrier') for x in o), sum(x.startswith('s_barrier') for x in ins), ratio))
if len(sys.argv) > 3 and sys.argv[2] == '--digests':  # sha256 of every kernel's instruction stream at `commit`, for tests/test_kernel_schedule.py
    import hashlib
    import json
    json.dump({'commit': commit, 'what': 'sha256 of the normalised gfx950 instruction stream (labels -> .L, comments and directives dropped) of every kernel '
               'of csrc/ at this commit: the device code of the last round-3 hardware leases', 'kernels': {n: hashlib.sha256('\n'.join(i).encode()).hexdigest()[:32] for n, i in sorted(old.items())}},
              open(sys.argv[3], 'w'), indent=0)
    print(len(old), 'digests ->', sys.argv[3])
    raise SystemExit(0)
print(__doc__)
print(f'commit {commit} vs working tree: {sum(len(v) for v in rows.values())} kernels present in both')
for key, v in rows.items():
    print(f'\n== {key}: {len(v)}')
    if key == 'identical':
        fam = {}
        for n in v:
            fam.setdefault(n.split('<')[0], []).append(n)
        for f, names in sorted(fam.items()):
            print(f'   {f}: {len(names)} instantiations')
    else:
        for r in v:
            print('   ' + r)
