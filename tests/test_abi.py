"""CPU tests of the drop-in boundary: libhq_hip.so loads without a GPU and exports every
symbol include/hq_hip.h declares; argument validation that needs no device works."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    src = open(os.path.join(ROOT, 'include', 'hq_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    src = re.sub(r'^\s*#.*$', '', src, flags=re.M)  # preprocessor lines (the visibility pragma)
    return sorted(set(re.findall(r'\b([A-Za-z_][A-Za-z0-9_]*)\s*\(', src)) - {'defined'})


def test_library_exports_every_declared_symbol():
    from hybridq_amd import core
    syms = _header_symbols()
    assert len(syms) >= 24
    for s in syms:
        assert hasattr(core._lib, s), s
    assert sorted(core.EXPORTED) == syms


def test_dynamic_symbol_table_is_exactly_the_header():
    """`nm -D` of libhq_hip.so == the prototypes of include/hq_hip.h, in both directions: no mangled hq:: internals, no kernel
    handles, no undeclared C entry points (the reference's .so files export their C symbols and nothing else,
    python_U.cpp:127-154, python_swap.cpp:68-99).  -fvisibility=hidden + the header's visibility pragma + libhq_hip.map."""
    import subprocess
    from hybridq_amd import build
    out = subprocess.run(['nm', '-D', '--defined-only', build.LIB], capture_output=True, text=True, check=True).stdout
    rows = [ln.split() for ln in out.splitlines() if ln.strip()]
    exported = sorted(r[-1] for r in rows)
    assert all(r[-2] == 'T' for r in rows), [r for r in rows if r[-2] != 'T'][:5]
    assert not [s for s in exported if s.startswith('_Z')], [s for s in exported if s.startswith('_Z')][:5]
    assert exported == _header_symbols(), (set(exported) ^ set(_header_symbols()))


def test_reference_boundary_names_present():
    """Exactly what hybridq/utils/dot.py:49-71 and transpose.py:52-58 bind."""
    from hybridq_amd import core
    for s in ['get_log2_pack_size', 'apply_U_float32', 'apply_U_float64', 'to_complex64', 'to_complex128'
              ] + [f'swap_{t}{b}' for t in ('float', 'int', 'uint') for b in (32, 64)]:
        getattr(core._lib, s)
    # truthy, else the reference driver silently falls back to einsum (simulation.py:393-397)
    assert core.log2_pack_size() >= 1
    assert set(core._dot_core) == {np.dtype('float32'), np.dtype('float64')}
    assert set(core._to_complex_core) == {np.dtype('complex64'), np.dtype('complex128')}
    assert len(core._swap_core) == 6


def test_noop_and_validation_without_gpu():
    from hybridq_amd import core
    a = np.zeros(16, dtype=np.float32)
    pos = np.zeros(1, dtype=np.uint32)
    P = ctypes.POINTER(ctypes.c_uint32)
    # n_pos == 0 is a no-op returning 0 (python_U.cpp:38-39, python_swap.cpp:35-36)
    assert core._dot_core[np.dtype('float32')](a.ctypes.data, a.ctypes.data, a.ctypes.data,
                                               pos.ctypes.data_as(P), 4, 0) == 0
    assert core._swap_core[np.dtype('float32')](a.ctypes.data, pos.ctypes.data_as(P), 4, 0) == 0
    # invalid arguments are rejected before any device work
    bad = np.asarray([7], dtype=np.uint32)
    assert core._dot_core[np.dtype('float32')](a.ctypes.data, a.ctypes.data, a.ctypes.data,
                                               bad.ctypes.data_as(P), 4, 1) != 0
    assert 'position' in core.last_error()
    bad = np.asarray([1, 1], dtype=np.uint32)
    assert core._swap_core[np.dtype('int64')](a.ctypes.data, bad.ctypes.data_as(P), 4, 2) != 0
    assert core.set_apply_mode('auto') is None
    try:
        core.set_apply_mode('bogus')
        assert False
    except core.HQError:
        pass


def test_product_never_imports_oracle():
    """The product path must not route through the CPU oracle (tier rule 3)."""
    pkg = os.path.join(ROOT, 'hybridq_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                txt = open(os.path.join(dirpath, f)).read()
                assert 'import oracle' not in txt and 'from oracle' not in txt, f
                assert 'libhq_oracle' not in txt and '_ref/' not in txt, f


def test_apply_U_argument_marshalling_without_a_device():
    """core.apply_U hands `pos` over as a ctypes array built from the integers; the library validates the positions before
    it touches a device, so the marshalling can be checked on a box without a GPU: bad positions are named as such, good
    ones get past that check (and then fail on something else here -- alignment or the missing device)."""
    import numpy as np
    from hybridq_amd import core
    U = np.eye(4, dtype=np.complex64)
    pre, pim = np.zeros(1024, np.float32), np.zeros(1024, np.float32)
    for pos in ([3, 50], [3, 3], np.array([10, 2]), (np.uint32(1), np.int64(99))):
        with pytest.raises(core.HQError, match='invalid positions'):
            core.apply_U(pre, pim, U, pos, 10)
    for pos in ([3, 5], np.array([0, 9], dtype=np.uint32), (np.int64(4), 7)):
        try:
            core.apply_U(pre, pim, U, pos, 10)
        except core.HQError as e:
            assert 'invalid positions' not in str(e)
    with pytest.raises(ValueError, match='incompatible'):
        core.apply_U(pre, pim, U, [1, 2, 3], 10)
    with pytest.raises(TypeError):
        core.apply_U(pre, pim, U, [1.5, 2], 10)
    with pytest.raises(ValueError, match='non-negative'):  # ctypes would wrap -1 to 4294967295 silently
        core.apply_U(pre, pim, U, [-1, 2], 10)


def test_rccl_that_cannot_be_loaded_is_an_error_not_a_crash():
    """the message was built from two dlerror() calls (the second returns NULL).  A rank without a loadable
    librccl must come back with rc = 1 and a message, so that it can vote for the fallback instead of dying while its peers
    wait.  HQ_RCCL_LIBRARY names the only candidate; a fresh process because a loaded RCCL stays loaded."""
    import subprocess
    import sys
    code = ("from hybridq_amd import core\n"
            "rc = core._lib.hq_shard_load_rccl()\n"
            "print('RC', rc, core.last_error())\n")
    env = dict(os.environ, HQ_RCCL_LIBRARY='/nonexistent/librccl.so', PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert 'RC 1 cannot load librccl' in out.stdout and '/nonexistent/librccl.so' in out.stdout, out.stdout


def test_exact_commutation_tolerance_follows_the_precision_of_the_gates():
    """gates given in complex64 that commute mathematically have commutators ~1e-7; the blocked planner's
    "commutes to rounding" test must still slide them, and must stay at 1e-12 for double-precision gates."""
    from hybridq_amd import fusion

    def rx(t):
        return np.array([[np.cos(t / 2), -1j * np.sin(t / 2)], [-1j * np.sin(t / 2), np.cos(t / 2)]])

    a64, b64 = rx(0.3).astype(np.complex64), rx(1.1).astype(np.complex64)
    assert fusion.exact_tolerance([(rx(0.3), (0,))]) == 1e-12
    tol = fusion.exact_tolerance([(a64, (0,)), (rx(0.2), (1,))])
    assert 1e-7 < tol < 1e-6
    # ... of the gates AND of the evolution: a complex128 run keeps its 1e-12 whatever precision the matrices arrive in
    # (ADVICE r04), in both planners
    assert fusion.exact_tolerance([(a64, (0,)), (rx(0.2), (1,))], 'complex128') == 1e-12
    assert fusion.exact_tolerance([(a64, (0,)), (rx(0.2), (1,))], 'complex64') == tol
    from hybridq_amd.blocking import plan_blocked
    # a pair that commutes to ~1e-7 only (single-precision rounding of the matrices) around a gate that blocks one of them
    c = np.array([[0, 1], [1, 0]], dtype=np.complex64)
    circ = [(a64, (0,)), (np.kron(c, c).astype(np.complex64), (0, 1)), (b64, (0,))]
    for native in (True, False):
        kw = dict(tile_bits=10, low_bits=4, native=native)
        ops64 = plan_blocked(circ, {q: q for q in range(14)}, 14, complex_type='complex64', **kw)
        ops128 = plan_blocked(circ, {q: q for q in range(14)}, 14, complex_type='complex128', **kw)
        for ops, ct in ((ops64, np.complex64), (ops128, np.complex128)):
            assert all(U.dtype == ct for op in ops for U in ([u for u, _ in op[2]] if op[0] == 'B' else [op[1]]))
    assert not fusion.commute(a64, (0,), b64, (0,), exact=True) or np.abs(a64 @ b64 - b64 @ a64).max() <= 1e-12
    assert fusion.commute(a64, (0,), b64, (0,), exact=tol)
    z = np.diag([1, 1j]).astype(np.complex64)
    assert not fusion.commute(a64, (0,), z, (0,), exact=tol)
