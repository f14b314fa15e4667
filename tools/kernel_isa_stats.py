"""Static instruction statistics of selected kernels from the gfx950 assembly hipcc emits for hq_apply.hip (no GPU needed):
barriers, wait counts, LDS / vector-memory / matrix-core instruction counts, registers and scratch.  Evidence for
changes made without a GPU at hand.
    python tools/kernel_isa_stats.py [regex of demangled kernel names]  > profiles/r04_blocked_kernel_isa.txt"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pat = re.compile(sys.argv[1] if len(sys.argv) > 1 else r'apply_blocked_kernel<float, 512, true, true, (true|false)>')
with tempfile.TemporaryDirectory() as td:
    asm = os.path.join(td, 'hq_apply.s')
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '--cuda-device-only', '-S',
                           os.path.join(ROOT, 'hybridq_amd', 'csrc', 'hq_apply.hip'), '-o', asm])
    text = open(asm).read()
funcs = re.split(r'\n(?=\s*\.globl\s)', text)
names = []
for f in funcs:
    m = re.search(r'\.globl\s+(\S+)', f)
    if m:
        names.append((m.group(1), f))
dem = subprocess.run(['c++filt'], input='\n'.join(n for n, _ in names), capture_output=True, text=True).stdout.splitlines()
keys = [('s_barrier', r'^\s*s_barrier'), ('s_waitcnt (any)', r'^\s*s_waitcnt'), ('  of which vmcnt(0)', r'^\s*s_waitcnt.*vmcnt\(0\)'),
        ('  of which lgkmcnt(0)', r'^\s*s_waitcnt.*lgkmcnt\(0\)'), ('v_mfma', r'^\s*v_mfma'), ('ds_read / ds_load', r'^\s*ds_(read|load)'),
        ('ds_write / ds_store', r'^\s*ds_(write|store)'), ('global_load', r'^\s*global_load'), ('global_store', r'^\s*global_store'),
        ('scratch_ / buffer_ (spills)', r'^\s*(scratch_|buffer_(load|store))'), ('v_mov_b32', r'^\s*v_mov_b32'), ('s_load', r'^\s*s_load'),
        ('s_cbranch', r'^\s*s_cbranch'), ('all instructions', r'^\s+[sv]_|^\s+ds_|^\s+global_|^\s+buffer_|^\s+scratch_')]
for (mangled, body), name in zip(names, dem):
    name = re.sub(r'^void ', '', name)
    if not pat.search(name):
        continue
    print(re.sub(r'\(.*$', '', name).replace('hq::', ''))
    lines = body.splitlines()
    for label, rx in keys:
        print(f'    {label:28s} {sum(1 for ln in lines if re.search(rx, ln))}')
    for key in ('.vgpr_count', '.sgpr_count', '.private_segment_fixed_size', '.agpr_count'):
        m = re.search(re.escape(key) + r':\s*(\d+)', text[text.find(mangled + '.kd'):] if (mangled + '.kd') in text else '')
        if m:
            print(f'    {key:28s} {m.group(1)}')
