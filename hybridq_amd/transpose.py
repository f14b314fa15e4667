"""``transpose(a, axes)`` for arrays whose dimensions are all 2: the counterpart of
``hybridq.utils.transpose`` (hybridq/utils/transpose.py:61-169) on ``swap_<dtype>`` of
libhq_hip.so.  Same rule as the reference: the leading already-ordered axes are skipped
and the remaining (trailing) axes are permuted by the core when their number is in
(3, 16] (transpose.py:131).  Anything else raises NotImplementedError unless
``force_numpy=True`` asks for numpy.transpose explicitly (the reference warns and falls
back; this package has no implicit CPU path)."""
import numpy as np

from . import core

_TYPES = tuple(np.dtype(t) for t in ('float32', 'float64', 'int32', 'int64', 'uint32', 'uint64'))


def _unordered_tail(axes):
    """Index of the first axis that is not already in place (len(axes) if all are)."""
    for i, x in enumerate(axes):
        if i != x:
            return i
    return len(axes)


def transpose(a, axes=None, inplace=False, backend='numpy', **kwargs):
    if backend != 'numpy':
        raise ValueError(f"Backend {backend} is not supported.")
    force_numpy = kwargs.get('force_numpy', False)
    if axes is None:
        return np.transpose(a)
    axes = [int(x) for x in np.asarray(axes).reshape(-1)]
    arr = np.asarray(a, order='C')
    if sorted(axes) != list(range(arr.ndim)):
        raise ValueError("axes don't match array")
    if force_numpy:
        return np.transpose(arr, axes)
    first = _unordered_tail(axes)
    tail = len(axes) - first
    in_domain = arr.dtype in _TYPES and arr.shape == (2,) * arr.ndim and (tail == 0 or 3 < tail <= 16)
    if not in_domain:
        if kwargs.get('raise_if_hcore_fails', False):
            raise AssertionError("Cannot use HybridQ core.")
        # reference: warn + numpy.transpose (transpose.py:155-166).  No implicit CPU path here.
        raise NotImplementedError(
            "transpose: outside the HIP core's domain (needs all dimensions 2, a 4/8-byte real or integer "
            "dtype and 4..16 unordered trailing axes); pass force_numpy=True for numpy.transpose")
    if tail == 0:
        return arr
    work = arr if (inplace or arr is not a) else arr.copy()
    # axis x is index bit ndim-1-x; swap() wants, for each low bit i, the bit that moves there
    pos = np.asarray([arr.ndim - 1 - x for x in reversed(axes[first:])], dtype=np.uint32)  # transpose.py:139-142
    core.swap(work.reshape(-1), pos, arr.ndim)
    return work
