"""Build libhq_hip.so (gfx950) in-tree with hipcc.  No JIT cache: the .so lives next
to its sources so that it travels with the repo snapshot to the GPU box."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(CSRC, 'libhq_hip.so')
SOURCES = [os.path.join(CSRC, 'hq_hip.hip')]
HEADERS = [os.path.join(CSRC, 'hq_kernels.h'), os.path.join(HERE, '..', 'include', 'hq_hip.h')]
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC', '-Wall',
         '-Wno-unused-function']


def _hipcc():
    for cand in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found (set HIPCC)')


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > t for p in SOURCES + HEADERS if os.path.exists(p))


def build(force=False, verbose=False, extra_flags=()):
    """Compile the HIP library if it is missing or older than its sources."""
    if not force and not needs_build():
        return LIB
    cmd = [_hipcc()] + FLAGS + list(extra_flags) + SOURCES + ['-o', LIB + '.tmp']
    if verbose:
        print(' '.join(cmd), file=sys.stderr)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError('hipcc failed:\n' + res.stdout + res.stderr)
    os.replace(LIB + '.tmp', LIB)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
