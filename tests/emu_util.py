"""TEST INFRASTRUCTURE: loads the host emulation build of the HIP library (tests/emu) behind a second instance of the
ctypes binding module, so that tests can drive the REAL planners and kernel bodies of libhq_hip.so on a box without a GPU:

    core = emu_util.emu_core()          # hybridq_amd/core.py bound to tests/emu/_build/libhq_emu.so
    core.apply_U(re, im, U, pos)         # numpy planes: the host-pointer path stages them into emulated device memory
    re, im, free = emu_util.device_planes(core, n, np.float32)   # or planes that ARE emulated device memory

The product never loads this library: `hybridq_amd.core` (the module the package imports) binds csrc/libhq_hip.so and
raises if that is missing."""
import ctypes
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_CORE = None


def emu_library():
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'emu'))
    try:
        import build as emu_build
        return emu_build.build()
    finally:
        sys.path.pop(0)
        sys.modules.pop('build', None)


def emu_rccl_library():
    """The stand-in for librccl (tests/emu/rccl_emu.cpp), built next to the emulated library."""
    return os.path.join(os.path.dirname(emu_library()), 'librccl_emu.so')


def emu_core():
    global _CORE
    if _CORE is None:
        lib = emu_library()
        spec = importlib.util.spec_from_file_location('hq_core_emu', os.path.join(ROOT, 'hybridq_amd', 'core.py'))
        mod = importlib.util.module_from_spec(spec)
        old = os.environ.get('HQ_HIP_LIBRARY')
        os.environ['HQ_HIP_LIBRARY'] = lib
        try:
            spec.loader.exec_module(mod)
        finally:
            if old is None:
                os.environ.pop('HQ_HIP_LIBRARY', None)
            else:
                os.environ['HQ_HIP_LIBRARY'] = old
        assert mod._lib._name == lib
        _CORE = mod
    return _CORE


def device_planes(core, n, float_dtype):
    """Two numpy views (re, im) of an emulated DEVICE allocation made by hq_alloc_state (plain placement), and a
    function that frees it: calls on them take the library's device-pointer path."""
    float_dtype = np.dtype(float_dtype)
    re, im = ctypes.c_void_p(), ctypes.c_void_p()
    rc = core._lib.hq_alloc_state(ctypes.c_uint(n), ctypes.c_int(8 * float_dtype.itemsize), ctypes.c_int(1),
                                  ctypes.byref(re), ctypes.byref(im))
    assert rc == 0, core.last_error()
    ct = ctypes.c_float if float_dtype.itemsize == 4 else ctypes.c_double
    size = 1 << n
    a_re = np.ctypeslib.as_array(ctypes.cast(re, ctypes.POINTER(ct)), shape=(size,))
    a_im = np.ctypeslib.as_array(ctypes.cast(im, ctypes.POINTER(ct)), shape=(size,))

    def free():
        assert core._lib.hq_free_state(re) == 0

    return a_re, a_im, free
