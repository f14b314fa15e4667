"""Single-gate convenience API: the counterpart of ``hybridq.utils.dot`` / ``to_complex`` /
``to_complex_array`` (hybridq/utils/dot.py:81-356), same arguments and return
conventions, running on ``libhq_hip.so``.

Differences that follow from the GPU kernels accepting any target position: no pre/post
low-bit swaps are ever issued (dot.py:217-221,281-317), so ``swap_back=False`` always
returns ``(result, None)`` (the reference returns ``None`` for the transposition whenever
no swap was needed, dot.py:323-329).  Host numpy inputs go through the library's
host-pointer path (staged over PCIe); split-plane torch CUDA tensors are updated in place
in HBM.  ``force_numpy=True`` is the explicit numpy path of the reference API (its tests
use it as the cross-check).  There is NO implicit CPU fallback: inputs outside the core's
domain (non-binary axes, > 10 target axes, unsupported dtypes) raise NotImplementedError
where the reference would warn and fall back (dot.py:332-335).
"""
from warnings import warn

import numpy as np

from . import core

_FLOAT_TYPES = (np.dtype('float32'), np.dtype('float64'))


def aligned_empty(shape, dtype, alignment=32):
    """numpy array whose data pointer is `alignment`-byte aligned (hybridq/utils/aligned)."""
    dtype = np.dtype(dtype)
    n = int(np.prod(shape))
    raw = np.empty(n * dtype.itemsize + alignment, dtype=np.uint8)
    off = (-raw.ctypes.data) % alignment
    return raw[off:off + n * dtype.itemsize].view(dtype).reshape(shape)


def to_complex(a, b):
    """a + 1j*b through ``to_complex64/128`` (dot.py:81-124)."""
    if a.shape != b.shape:
        raise ValueError("'a' and 'b' must have the same shape.")
    if np.iscomplexobj(a) or np.iscomplexobj(b):
        raise ValueError("Both 'a' and 'b' must be real valued.")
    if a.dtype == b.dtype and a.dtype in _FLOAT_TYPES:
        a = np.ascontiguousarray(a)
        b = np.ascontiguousarray(b)
        ctype = np.dtype('complex64') if a.dtype == np.dtype('float32') else np.dtype('complex128')
        c = np.empty(a.shape, dtype=ctype)
        core.to_complex(a, b, c)
        return c
    return a + 1j * b


def to_complex_array(a):
    """(2,)+a.shape real array [re, im] of a complex array (dot.py:127-136)."""
    if not np.iscomplexobj(a):
        raise ValueError("'a' must be an array of complex numbers.")
    ft = np.real(np.array([0], dtype=a.dtype)).dtype
    return np.reshape(np.reshape(np.asarray(a, order='C').view(ft), (int(np.prod(a.shape)), 2)).T,
                      (2,) + a.shape)


def _is_cuda_tensor(x):
    return hasattr(x, 'data_ptr') and getattr(x, 'is_cuda', False)


def dot(a, b, axes_b=None, b_as_complex_array=False, inplace=False, backend='numpy', **kwargs):
    """Apply the square matrix `a` to the axes `axes_b` of `b` (all of dimension 2).

    `b` is either a complex array of shape (2,)*n or, with ``b_as_complex_array=True``, a
    real array of shape (2,)+(2,)*n holding [re, im].  ``a``'s index has ``axes_b[0]`` as
    most significant bit (dot.py:214: ``pos = b_ndim - axes_b[::-1] - 1``)."""
    if backend != 'numpy':
        raise ValueError(f"Backend {backend} is not supported.")
    kwargs.setdefault('out', None)
    kwargs.setdefault('force_numpy', False)
    kwargs.setdefault('raise_if_hcore_fails', False)
    kwargs.setdefault('swap_back', True)
    kwargs.setdefault('alignment', 32)
    if axes_b is None:
        return np.dot(a, b, out=kwargs['out'])

    a = np.asarray(a, order='C')
    axes_b = np.asarray(axes_b)

    # device-resident split planes: in-place update in HBM
    if _is_cuda_tensor(b):
        if not b_as_complex_array or b.shape[0] != 2:
            raise ValueError("CUDA tensors must be split planes: shape (2,)+(2,)*n, b_as_complex_array=True")
        b_ndim = b.dim() - 1
        if any(axes_b >= b_ndim):
            raise IndexError("Index not in 'b'")
        pos = (b_ndim - axes_b[::-1] - 1).astype('uint32')
        planes = b if inplace else b.clone()
        flat = planes.reshape(2, -1)
        core.apply_U(flat[0], flat[1], a, pos, b_ndim)
        return planes if kwargs['swap_back'] else (planes, None)

    _b_orig = b
    b = np.asarray(b, order='C')
    _new_b = b is not _b_orig
    a_ndim = a.ndim
    b_ndim = b.ndim - (1 if b_as_complex_array else 0)
    a_shape = np.asarray(a.shape)
    b_shape = np.asarray(b.shape[:len(b.shape) - (1 if b_as_complex_array else 0)]) if not b_as_complex_array \
        else np.asarray(b.shape[1:])
    real_type = b.dtype if b_as_complex_array else np.real(np.array([1], dtype=b.dtype)).dtype
    complex_type = (1j * np.array([1], dtype=real_type)).dtype
    if b_as_complex_array:
        if b.shape[0] != 2:
            raise ValueError("'b' is in the wrong format.")
        if np.iscomplexobj(b):
            raise ValueError("'b' is expected to be real.")
    if any(axes_b >= b_ndim):
        raise IndexError("Index not in 'b'")
    if a_shape[-1] != np.prod(b_shape[axes_b]):
        raise ValueError("'a' and 'b' are incompatible.")
    pos = (b_ndim - axes_b[::-1] - 1).astype('uint32')

    use_core = not kwargs['force_numpy']
    use_core &= np.dtype(real_type) in _FLOAT_TYPES
    use_core &= a_ndim == 2 and a_shape[0] == a_shape[1]
    use_core &= bool(all(x == 2 for x in b_shape))
    use_core &= len(axes_b) <= 10 and len(set(axes_b.tolist())) == len(axes_b)
    if not use_core and not kwargs['force_numpy'] and kwargs['raise_if_hcore_fails']:
        raise AssertionError("Cannot use HybridQ core.")

    if use_core:
        if a.dtype != complex_type:
            warn(f"'a' is recast to '{complex_type}' to match 'b'.")
            a = a.astype(complex_type)
        n_amp = 1 << b_ndim
        if b_as_complex_array:
            aligned = b.ctypes.data % 32 == 0 and (b.ctypes.data + n_amp * b.itemsize) % 32 == 0
            if (inplace or _new_b) and aligned and b.flags.c_contiguous and b.flags.writeable:
                planes = b
            else:
                planes = aligned_empty(b.shape, real_type, kwargs['alignment'])
                planes[...] = b
        else:
            planes = aligned_empty((2,) + b.shape, real_type, kwargs['alignment'])
            planes[0] = np.real(b)
            planes[1] = np.imag(b)
        flat = planes.reshape(2, -1)
        core.apply_U(flat[0], flat[1], a, pos, b_ndim)
        res = planes if b_as_complex_array else to_complex(planes[0], planes[1])
        if b_as_complex_array and inplace and planes is not b:
            _b_orig[...] = planes  # honour inplace for unaligned / non-contiguous inputs
            res = _b_orig
        return res if kwargs['swap_back'] is True else (res, None)

    if not kwargs['force_numpy']:
        # The reference warns and falls back to numpy here (dot.py:332-335).  This package has no
        # implicit CPU path: inputs outside the HIP core's domain are an error unless the caller
        # asks for numpy explicitly.
        raise NotImplementedError(
            "dot: input outside the HIP core's domain (needs all axes of dimension 2, a square matrix, "
            "float32/float64 planes, <= 10 target axes); pass force_numpy=True for the numpy path")
    if b_as_complex_array:
        b = np.reshape(b[0] + 1j * b[1], b_shape)
    perm = axes_b.tolist() + [x for x in range(b_ndim) if x not in axes_b]
    bb = np.reshape(np.transpose(b, perm), (int(np.prod(b_shape[axes_b])), -1))
    inv = [perm.index(x) for x in range(len(perm))]
    bb = np.transpose(np.reshape(np.dot(a, bb), b_shape), inv)
    return np.array([np.real(bb), np.imag(bb)]) if b_as_complex_array else bb
