"""Aligned host arrays: the counterpart of ``hybridq.utils.aligned`` (hybridq/utils/aligned/aligned_array.py:23-420), which
the reference's driver uses for its split planes (simulation.py:490-497: 32-byte alignment, U.h:34-36).  Here the state
lives in HBM; these helpers serve callers of the host-pointer compatibility path (``apply_U_float32`` on
host planes wants them 32-byte aligned) and code written against the reference's module.  Plain numpy, nothing on the hot
path."""
import numpy as np


def isaligned(a, alignment):
    """True if the data of `a` starts at a multiple of `alignment` bytes."""
    return a.ctypes.data % int(alignment) == 0


def get_alignment(a, max_alignment=128):
    """Largest power of two <= `max_alignment` (itself a power of two) that divides the address of `a`'s data."""
    max_alignment = int(max_alignment)
    if max_alignment <= 0 or max_alignment & (max_alignment - 1):
        raise ValueError("'max_alignment' must be a power of 2.")
    addr, al = a.ctypes.data, max_alignment
    while al > 1 and addr % al:
        al >>= 1
    return al


def _check(alignment):
    alignment = int(alignment)
    if alignment <= 0 or alignment & (alignment - 1):
        raise ValueError("'alignment' must be a power of 2.")
    return alignment


def empty(shape, dtype=float, order='C', *, alignment=16):
    """Uninitialised array of `shape` / `dtype` / `order` ('C' or 'F') whose data is `alignment`-byte aligned."""
    alignment = _check(alignment)
    if order not in ('C', 'F'):
        raise ValueError("'order' must be either 'C' or 'F'.")
    dtype = np.dtype(dtype)
    shape = (int(shape),) if np.ndim(shape) == 0 else tuple(int(x) for x in shape)
    nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
    raw = np.empty(nbytes + alignment, dtype=np.uint8)
    off = (-raw.ctypes.data) % alignment
    return np.ndarray(shape, dtype=dtype, buffer=raw.data, offset=off, order=order)


def zeros(shape, dtype=float, order='C', *, alignment=16):
    a = empty(shape, dtype, order, alignment=alignment)
    a.fill(0)
    return a


def ones(shape, dtype=float, order='C', *, alignment=16):
    a = empty(shape, dtype, order, alignment=alignment)
    a.fill(1)
    return a


def _like(gen, a):
    return gen(a.shape, a.dtype, 'C' if a.flags.c_contiguous else 'F', alignment=get_alignment(a))


def empty_like(a):
    return _like(empty, a)


def zeros_like(a):
    return _like(zeros, a)


def ones_like(a):
    return _like(ones, a)


def _target_order(a, order):
    if order in ('C', 'F'):
        return order
    if order in ('A', 'K'):
        return 'F' if (a.flags.f_contiguous and not a.flags.c_contiguous) else 'C'
    raise ValueError("'order' must be one of 'C', 'F', 'A', 'K'.")


def array(a, dtype=None, order='K', *, alignment=16, copy=True):
    """An aligned array with the contents of `a`; ``copy=False`` hands `a` itself back when it already has the dtype, the
    layout and the alignment asked for."""
    a = np.asarray(a)
    dtype = a.dtype if dtype is None else np.dtype(dtype)
    want = _target_order(a, order)
    fits = a.dtype == dtype and isaligned(a, _check(alignment)) and (a.flags.c_contiguous if want == 'C' else a.flags.f_contiguous)
    if order in ('A', 'K') and a.flags.c_contiguous and a.flags.f_contiguous:
        fits = a.dtype == dtype and isaligned(a, alignment)
    if fits and not copy:
        return a
    out = empty(a.shape, dtype, want, alignment=alignment)
    out[...] = a
    return out


def asarray(a, dtype=None, order='K', *, alignment=16):
    """`a` itself if it is an aligned array of the requested dtype and layout, an aligned copy otherwise."""
    return array(a, dtype, order, alignment=alignment, copy=False)
