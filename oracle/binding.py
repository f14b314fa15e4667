"""ctypes loader for the CPU oracle libraries -- TEST INFRASTRUCTURE ONLY.

Mirrors the reference's bindings (hybridq/utils/dot.py:49-71,
hybridq/utils/transpose.py:52-58): same symbols, argtypes and restype, so the
port (``libhq_oracle.so``) and the compiled reference (``_ref/hybridq.so`` +
``_ref/hybridq_swap.so``) are interchangeable behind :class:`OracleLib`.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

_C = {
    np.dtype('float32'): ctypes.c_float,
    np.dtype('float64'): ctypes.c_double,
    np.dtype('int32'): ctypes.c_int32,
    np.dtype('int64'): ctypes.c_int64,
    np.dtype('uint32'): ctypes.c_uint32,
    np.dtype('uint64'): ctypes.c_uint64,
}


def aligned_empty(shape, dtype, alignment=256):
    """numpy array whose data pointer is `alignment`-byte aligned (the reference
    requires 32 B, U.h:34-36; 256 B keeps even an AVX-512 build safe)."""
    dtype = np.dtype(dtype)
    n = int(np.prod(shape))
    raw = np.empty(n * dtype.itemsize + alignment, dtype=np.uint8)
    off = (-raw.ctypes.data) % alignment
    return raw[off:off + n * dtype.itemsize].view(dtype).reshape(shape)


class OracleLib:
    """apply_U / swap / to_complex on host numpy arrays through one C library
    (or a pair, for the reference build which splits them in two .so files)."""

    def __init__(self, path_u, path_swap=None, kind='port'):
        self.kind = kind
        self._u = ctypes.CDLL(path_u)
        self._s = ctypes.CDLL(path_swap) if path_swap else self._u
        self._u.get_log2_pack_size.restype = ctypes.c_uint
        self._u.get_log2_pack_size.argtypes = []
        for b, ct in ((32, ctypes.c_float), (64, ctypes.c_double)):
            f = getattr(self._u, f'apply_U_float{b}')
            f.restype = ctypes.c_int
            f.argtypes = [ctypes.POINTER(ct)] * 3 + [
                ctypes.POINTER(ctypes.c_uint32), ctypes.c_uint, ctypes.c_uint
            ]
            g = getattr(self._u, f'to_complex{2 * b}')
            g.restype = ctypes.c_int
            g.argtypes = [ctypes.POINTER(ct)] * 3 + [ctypes.c_uint]
        for dt, ct in _C.items():
            f = getattr(self._s, f'swap_{dt.name}')
            f.restype = ctypes.c_int
            f.argtypes = [
                ctypes.POINTER(ct),
                ctypes.POINTER(ctypes.c_uint32), ctypes.c_uint, ctypes.c_uint
            ]

    @property
    def log2_pack_size(self):
        return int(self._u.get_log2_pack_size())

    # --- raw calls on numpy arrays (in place) ---------------------------------
    def apply_U(self, re, im, U, pos, n=None):
        ft = re.dtype
        assert im.dtype == ft and ft in (np.dtype('float32'), np.dtype('float64'))
        assert re.flags.c_contiguous and im.flags.c_contiguous
        ct = _C[ft]
        ctype = np.dtype('complex64') if ft == np.dtype('float32') else np.dtype('complex128')
        U = np.ascontiguousarray(U, dtype=ctype)
        pos = np.ascontiguousarray(pos, dtype=np.uint32)
        n = int(np.log2(re.size)) if n is None else n
        return int(
            getattr(self._u, f'apply_U_{ft.name}')(
                re.ctypes.data_as(ctypes.POINTER(ct)), im.ctypes.data_as(ctypes.POINTER(ct)),
                U.ctypes.data_as(ctypes.POINTER(ct)),
                pos.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), n, len(pos)))

    def swap(self, a, pos, n=None):
        assert a.flags.c_contiguous
        ct = _C[a.dtype]
        pos = np.ascontiguousarray(pos, dtype=np.uint32)
        n = int(np.log2(a.size)) if n is None else n
        return int(
            getattr(self._s, f'swap_{a.dtype.name}')(
                a.ctypes.data_as(ctypes.POINTER(ct)),
                pos.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), n, len(pos)))

    def to_complex(self, re, im):
        ft = re.dtype
        ct = _C[ft]
        ctype = np.dtype('complex64') if ft == np.dtype('float32') else np.dtype('complex128')
        out = np.empty(re.shape, dtype=ctype)
        rc = getattr(self._u, f'to_complex{ctype.itemsize * 8}')(
            re.ctypes.data_as(ctypes.POINTER(ct)), im.ctypes.data_as(ctypes.POINTER(ct)),
            out.ctypes.data_as(ctypes.POINTER(ct)), re.size)
        assert rc == 0
        return out


def _port_path():
    return os.path.join(_HERE, 'libhq_oracle.so')


def _ref_paths():
    return (os.path.join(_HERE, '_ref', 'hybridq.so'), os.path.join(_HERE, '_ref', 'hybridq_swap.so'))


def load_port():
    p = _port_path()
    if not os.path.exists(p):
        raise FileNotFoundError(f'{p} missing: run `make -C oracle port` (or __graft_entry__.build())')
    return OracleLib(p, kind='port')


def have_ref():
    return all(os.path.exists(p) for p in _ref_paths())


def load_ref():
    u, s = _ref_paths()
    if not have_ref():
        raise FileNotFoundError(f'{u} missing: run `make -C oracle ref` where /root/reference exists')
    return OracleLib(u, s, kind='reference')
