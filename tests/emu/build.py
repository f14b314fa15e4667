"""TEST INFRASTRUCTURE: builds tests/emu/_build/libhq_emu.so -- the five translation units of libhq_hip.so compiled as
plain C++ for the host against the emulation shim (tests/emu/shim), plus the emulator runtime (hip_emu.cpp).  Same
sources, same planners, same kernel bodies; the HIP programming model is provided by the shim.  Only tests load the
result (tests/emu_util.py); nothing under hybridq_amd/ knows about it."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, 'hybridq_amd', 'csrc')
#: HQ_EMU_ASAN=1: the same build under AddressSanitizer (shared runtime; load with LD_PRELOAD=asan_runtime()): every global
#: memory access of every kernel body is bounds-checked against the emulated device allocations (malloc) -- the sanitizer
#: run the GPU boxes could not do (their instrumented code object needs XNACK)
ASAN = os.environ.get('HQ_EMU_ASAN') == '1'
#: HQ_EMU_EXTRA_FLAGS='-D...': an A/B build of the emulated library in its own directory (bit-identity checks
#: between loop variants of a kernel: tests/test_emu_kernels.py::test_pipelined_loops_are_bit_identical)
EXTRA = os.environ.get('HQ_EMU_EXTRA_FLAGS', '').split()
OUT = os.path.join(HERE, '_build_asan' if ASAN else ('_build_ab' if EXTRA else '_build'))
LIB = os.path.join(OUT, 'libhq_emu.so')
RCCL = os.path.join(OUT, 'librccl_emu.so')


def asan_runtime():
    return subprocess.run([_cxx(), '-print-file-name=libclang_rt.asan-x86_64.so'], capture_output=True, text=True).stdout.strip()
UNITS = ['hq_core', 'hq_apply', 'hq_swap', 'hq_shard', 'hq_state', 'hq_plan']


def _cxx():
    for cand in (os.environ.get('HQ_EMU_CXX'), '/opt/rocm/lib/llvm/bin/clang++', '/usr/bin/clang++'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('the emulation build needs clang++ (ext_vector_type, address_space, __builtin_nontemporal_*)')


def _deps():
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(('.hip', '.h'))]
    deps += [os.path.join(ROOT, 'include', 'hq_hip.h'), os.path.join(HERE, 'hip_emu.cpp'), os.path.join(HERE, 'emu_selftest.cpp'), os.path.join(HERE, 'rccl_emu.cpp'), os.path.abspath(__file__),
             os.path.join(HERE, 'shim', 'hip', 'hip_runtime.h'), os.path.join(HERE, 'shim', 'rccl', 'rccl.h')]
    return deps


def build(force=False, verbose=False):
    if not force and os.path.exists(LIB) and all(os.path.getmtime(d) <= os.path.getmtime(LIB) for d in _deps()):
        return LIB
    os.makedirs(OUT, exist_ok=True)
    cxx = _cxx()
    flags = ['-std=c++17', '-O2', '-g1', '-fPIC', '-DHQ_EMU=1', '-ffp-contract=off', '-Wno-unused-function', '-Wno-unused-value',
             '-Wno-unknown-attributes', '-Wno-ignored-attributes', '-I', os.path.join(HERE, 'shim')]
    flags += EXTRA
    link = []
    if ASAN:
        flags += ['-fsanitize=address', '-shared-libasan', '-fno-omit-frame-pointer']
        link = ['-fsanitize=address', '-shared-libasan']
    jobs, objs = [], []
    for u in UNITS:
        obj = os.path.join(OUT, u + '.o')
        objs.append(obj)
        jobs.append([cxx, '-x', 'c++'] + flags + ['-c', os.path.join(CSRC, u + '.hip'), '-o', obj])
    for extra in ('hip_emu', 'emu_selftest'):
        obj = os.path.join(OUT, extra + '.o')
        objs.append(obj)
        jobs.append([cxx] + flags + ['-c', os.path.join(HERE, extra + '.cpp'), '-o', obj])

    def run(cmd):
        if verbose:
            print(' '.join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError('emulation build failed:\n' + ' '.join(cmd) + '\n' + r.stdout + r.stderr[-6000:])

    with ThreadPoolExecutor(len(jobs)) as pool:
        list(pool.map(run, jobs))
    run([cxx, '-shared', '-fPIC'] + link + objs + ['-o', LIB + '.tmp', '-ldl', '-lpthread', '-lrt'])
    os.replace(LIB + '.tmp', LIB)
    # the stand-in for librccl (HQ_RCCL_LIBRARY): its own shared object, like the real one
    run([cxx, '-shared', '-fPIC', '-std=c++17', '-O1', os.path.join(HERE, 'rccl_emu.cpp'), '-o', RCCL + '.tmp'])
    os.replace(RCCL + '.tmp', RCCL)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
