"""Build libhq_hip.so (gfx950) in-tree with hipcc.  No JIT cache: the .so lives next
to its sources so that it travels with the repo snapshot to the GPU box.

The library is six translation units (csrc/hq_{core,apply,swap,shard,state,plan}.hip) compiled in parallel
and linked into one shared object; an object is rebuilt when its source or any header is newer."""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(CSRC, 'libhq_hip.so')
UNITS = ['hq_core', 'hq_apply', 'hq_swap', 'hq_shard', 'hq_state', 'hq_plan']
SOURCES = [os.path.join(CSRC, u + '.hip') for u in UNITS]
HEADERS = [os.path.join(CSRC, 'libhq_hip.map')] + [os.path.join(CSRC, h) for h in ('hq_common.h', 'hq_kernels_common.h', 'hq_kernels_apply.h', 'hq_kernels_blocked.h', 'hq_kernels_blocked_r3.h', 'hq_kernels_gemm.h', 'hq_kernels_swap.h',
                                           'hq_kernels_aux.h', 'hq_bitperm.h')] + [os.path.join(HERE, '..', 'include', 'hq_hip.h')]
ARCH = '--offload-arch=gfx950'
# -fvisibility=hidden: the dynamic symbol table is what include/hq_hip.h declares (its visibility pragma) and nothing else
CFLAGS = [ARCH, '-O3', '-std=c++17', '-fPIC', '-fvisibility=hidden', '-fvisibility-inlines-hidden', '-Wall', '-Wno-unused-function']
LDFLAGS = [ARCH, '-shared', '-fPIC', '-Wl,--version-script=' + os.path.join(CSRC, 'libhq_hip.map')]
OBJDIR = os.path.join(CSRC, 'build')


def _hipcc():
    for cand in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found (set HIPCC)')


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(p) > t for p in deps if os.path.exists(p))


def needs_build():
    return _newer(LIB, SOURCES + HEADERS)


def _run(cmd, verbose):
    if verbose:
        print(' '.join(cmd), file=sys.stderr)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError('hipcc failed:\n' + ' '.join(cmd) + '\n' + res.stdout + res.stderr)
    return res.stderr


def build(force=False, verbose=False, extra_flags=(), lib=LIB, objdir=OBJDIR, log=None):
    """Compile the HIP library if it is missing or older than its sources.  `extra_flags` go to every compile step
    (pass a distinct `lib` / `objdir` with them so that experiment builds do not replace the product); `log`: a list
    that receives the compiler's stderr of every step (remarks such as -Rpass-analysis=kernel-resource-usage)."""
    if not force and not _newer(lib, SOURCES + HEADERS):
        return lib
    hipcc = _hipcc()
    os.makedirs(objdir, exist_ok=True)
    objs, jobs = [], []
    for src in SOURCES:
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + '.o')
        objs.append(obj)
        if force or _newer(obj, [src] + HEADERS):
            jobs.append([hipcc] + CFLAGS + list(extra_flags) + ['-c', src, '-o', obj])
    with ThreadPoolExecutor(max(1, min(len(jobs), os.cpu_count() or 1))) as pool:
        errs = list(pool.map(lambda cmd: _run(cmd, verbose), jobs))
    if log is not None:
        log.extend(errs)
    _run([hipcc] + LDFLAGS + objs + ['-o', lib + '.tmp'], verbose)
    os.replace(lib + '.tmp', lib)
    return lib


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
