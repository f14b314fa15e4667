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


_DOT_DEFAULTS = dict(out=None, force_numpy=False, raise_if_hcore_fails=False, swap_back=True, alignment=32)


class _HostOperand:
    """Shape / dtype bookkeeping of a host `b` operand: either a complex (2,)*n array or, in split
    form, a real (2,)+(2,)*n array [re, im].  Validation errors match the reference's wording
    (dot.py:197-214) because its tests look for them."""

    def __init__(self, b, split):
        self.given = b
        self.arr = np.asarray(b, order='C')
        self.copied = self.arr is not b
        self.split = bool(split)
        if self.split:
            if self.arr.shape[0] != 2:
                raise ValueError("'b' is in the wrong format.")
            if np.iscomplexobj(self.arr):
                raise ValueError("'b' is expected to be real.")
        self.shape = tuple(self.arr.shape[1:] if self.split else self.arr.shape)
        self.n = len(self.shape)
        self.real_type = np.dtype(self.arr.dtype if self.split else np.empty(0, self.arr.dtype).real.dtype)
        self.complex_type = np.result_type(self.real_type, np.complex64)

    def binary(self):
        return all(d == 2 for d in self.shape)

    def planes(self, inplace, alignment):
        """(2, 2^n)-shaped view of 32-byte aligned planes holding the operand, and whether that is
        the caller's own memory."""
        if self.split:
            b = self.arr
            second = b.ctypes.data + (1 << self.n) * b.itemsize
            usable = (inplace or self.copied) and b.flags.c_contiguous and b.flags.writeable
            if usable and b.ctypes.data % 32 == 0 and second % 32 == 0:
                return b, True
            mine = aligned_empty(b.shape, self.real_type, alignment)
            mine[...] = b
            return mine, False
        mine = aligned_empty((2,) + self.shape, self.real_type, alignment)
        mine[0], mine[1] = self.arr.real, self.arr.imag
        return mine, False


def _numpy_dot(a, op, axes):
    """The explicit ``force_numpy=True`` route: the target axes are brought to the front, flattened
    into the row index of a matrix product, and put back."""
    b = op.arr[0] + 1j * op.arr[1] if op.split else op.arr
    order = list(axes) + [i for i in range(op.n) if i not in axes]
    rows = np.transpose(b, order).reshape(a.shape[-1], -1)
    out = np.dot(a, rows).reshape([op.shape[i] for i in order])
    out = np.transpose(out, np.argsort(order))
    return np.array([out.real, out.imag]) if op.split else out


def dot(a, b, axes_b=None, b_as_complex_array=False, inplace=False, backend='numpy', **kwargs):
    """Apply the square matrix `a` to the axes `axes_b` of `b` (all of dimension 2).

    `b` is either a complex array of shape (2,)*n or, with ``b_as_complex_array=True``, a
    real array of shape (2,)+(2,)*n holding [re, im] -- or a torch CUDA tensor in that split
    form, which is updated in HBM.  ``a``'s index has ``axes_b[0]`` as most significant bit
    (dot.py:214: ``pos = b_ndim - axes_b[::-1] - 1``)."""
    if backend != 'numpy':
        raise ValueError(f"Backend {backend} is not supported.")
    opt = {**_DOT_DEFAULTS, **kwargs}
    if axes_b is None:
        return np.dot(a, b, out=opt['out'])
    a = np.asarray(a, order='C')
    axes = [int(x) for x in np.asarray(axes_b).reshape(-1)]
    wrap = (lambda r: r) if opt['swap_back'] is True else (lambda r: (r, None))  # no swaps are ever pending

    if _is_cuda_tensor(b):  # device-resident split planes: in-place update in HBM
        if not b_as_complex_array or b.shape[0] != 2:
            raise ValueError("CUDA tensors must be split planes: shape (2,)+(2,)*n, b_as_complex_array=True")
        n = b.dim() - 1
        if any(x >= n for x in axes):
            raise IndexError("Index not in 'b'")
        planes = b if inplace else b.clone()
        flat = planes.reshape(2, -1)
        core.apply_U(flat[0], flat[1], a, [n - 1 - x for x in reversed(axes)], n)
        return wrap(planes)

    op = _HostOperand(b, b_as_complex_array)
    if any(x >= op.n for x in axes):
        raise IndexError("Index not in 'b'")
    if a.shape[-1] != int(np.prod([op.shape[x] for x in axes])):
        raise ValueError("'a' and 'b' are incompatible.")

    in_domain = (op.real_type in _FLOAT_TYPES and a.ndim == 2 and a.shape[0] == a.shape[1] and op.binary()
                 and len(axes) <= 10 and len(set(axes)) == len(axes))
    if opt['force_numpy']:
        return _numpy_dot(a, op, axes)
    if not in_domain:
        if opt['raise_if_hcore_fails']:
            raise AssertionError("Cannot use HybridQ core.")
        # The reference warns and falls back to numpy here (dot.py:332-335).  This package has no
        # implicit CPU path: inputs outside the HIP core's domain are an error unless the caller
        # asks for numpy explicitly.
        raise NotImplementedError(
            "dot: input outside the HIP core's domain (needs all axes of dimension 2, a square matrix, "
            "float32/float64 planes, <= 10 target axes); pass force_numpy=True for the numpy path")

    if a.dtype != op.complex_type:
        warn(f"'a' is recast to '{op.complex_type}' to match 'b'.")
        a = a.astype(op.complex_type)
    planes, own = op.planes(inplace, opt['alignment'])
    flat = planes.reshape(2, -1)
    core.apply_U(flat[0], flat[1], a, [op.n - 1 - x for x in reversed(axes)], op.n)
    if not op.split:
        return wrap(to_complex(planes[0], planes[1]))
    if inplace and not own:  # honour inplace for unaligned / non-contiguous inputs
        op.given[...] = planes
        return wrap(op.given)
    return wrap(planes)
