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


def transpose(a, axes=None, inplace=False, backend='numpy', **kwargs):
    if backend != 'numpy':
        raise ValueError(f"Backend {backend} is not supported.")
    kwargs.setdefault('force_numpy', False)
    kwargs.setdefault('raise_if_hcore_fails', False)
    if axes is None:
        return np.transpose(a)
    axes = np.asarray(axes)
    _orig = a
    a = np.asarray(a, order='C')
    _new = a is not _orig
    if sorted(axes.tolist()) != list(range(a.ndim)):
        raise ValueError("axes don't match array")
    use_core = not kwargs['force_numpy'] and a.dtype in _TYPES and a.shape == (2,) * a.ndim
    if not use_core and not kwargs['force_numpy'] and kwargs['raise_if_hcore_fails']:
        raise AssertionError("Cannot use HybridQ core.")
    if use_core:
        n_ord = next((i for i, x in enumerate(axes) if i != x), len(axes))
        if n_ord == len(axes):
            return a
        if 3 < len(axes) - n_ord <= 16:
            if not inplace and not _new:
                a = np.array(a)
            sub = axes[n_ord:]
            pos = (a.ndim - sub[::-1] - 1).astype('uint32')  # transpose.py:139-142
            core.swap(a.reshape(-1), pos, a.ndim)
            return a
    if not kwargs['force_numpy']:
        # reference: warn + numpy.transpose (transpose.py:155-166).  No implicit CPU path here.
        raise NotImplementedError(
            "transpose: outside the HIP core's domain (needs all dimensions 2, a 4/8-byte real or integer "
            "dtype and 4..16 unordered trailing axes); pass force_numpy=True for numpy.transpose")
    return np.transpose(a, axes)
