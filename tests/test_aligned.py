"""hybridq_amd.aligned against the statements of the reference's test_utils__aligned_array (tests.py:152-253): every
generator at every alignment / order / dtype, `array` copies, `asarray` returns its argument when nothing has to change."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

DTYPES = ['int8', 'int16', 'int32', 'int64', 'uint8', 'uint16', 'uint32', 'uint64', 'float32', 'float64', 'float128',
          'complex64', 'complex128']


def _order(a):
    return ('C' if a.flags.c_contiguous else '') + ('F' if a.flags.f_contiguous else '')


@pytest.mark.filterwarnings('ignore')  # complex -> real and float -> int casts of random data, as in the reference's test
@pytest.mark.parametrize('order', 'CF')
@pytest.mark.parametrize('alignment', [16, 32, 64, 128])
def test_utils__aligned_array(order, alignment):
    from hybridq_amd.aligned import array, asarray, empty, empty_like, get_alignment, isaligned, ones, ones_like, zeros, zeros_like
    rng = np.random.default_rng(alignment + ord(order))
    for _ in range(25):
        shape = tuple(int(x) + 1 for x in rng.integers(0, 2**4, size=1 + int(rng.integers(0, 5))))
        dtype = np.dtype(str(rng.choice(DTYPES)))
        for gen in (empty, ones, zeros, array):
            if gen is array:
                r = np.asarray(rng.random(shape), dtype=dtype, order=order)
                _a = array(r, alignment=alignment)
                assert np.array_equal(r, _a) and r.shape == _a.shape and _order(r) == _order(_a)
            else:
                _a = gen(shape=shape, dtype=dtype, order=order, alignment=alignment)
            a = array(_a, alignment=alignment)
            assert not np.may_share_memory(a, _a)
            for x in (_a, a):
                assert x.shape == shape and x.dtype == dtype and order in _order(x) and isaligned(x, alignment)
                assert get_alignment(x) >= min(alignment, 128)
                if gen is zeros:
                    assert not x.any()
                elif gen is ones:
                    assert (x == 1).all()
            assert asarray(a, dtype=dtype, order=order, alignment=alignment) is a and asarray(a, alignment=alignment) is a
            other = next(t for t in DTYPES if np.dtype(t) != a.dtype)
            c1 = asarray(a, dtype=other, alignment=alignment)
            assert c1.shape == a.shape and c1.dtype == np.dtype(other) and isaligned(c1, alignment) and order in _order(c1)
            assert not np.may_share_memory(c1, a)
            flip = 'C' if order == 'F' else 'F'
            c2 = asarray(a, order=flip, alignment=alignment)
            if _order(a) == 'CF':
                assert c2 is a
            else:
                assert c2.shape == a.shape and c2.dtype == a.dtype and isaligned(c2, alignment) and flip in _order(c2)
                assert not np.may_share_memory(c2, a) and (gen is empty or np.array_equal(c2, a))
            for like, val in ((empty_like, None), (zeros_like, 0), (ones_like, 1)):
                b = like(a)
                assert b.shape == a.shape and b.dtype == a.dtype and isaligned(b, min(alignment, 128)) and order in _order(b)
                assert val is None or (b == val).all()
    with pytest.raises(ValueError):
        empty((2, 2), alignment=24)
    with pytest.raises(ValueError):
        get_alignment(np.zeros(3), max_alignment=48)


def test_aligned_planes_pass_the_library_alignment_check():
    """What the host-pointer path asks of its planes (U.h:34-36: 32 bytes) is what aligned.empty(alignment=32) delivers: the
    call gets past the library's alignment check (and fails later here, for want of a device)."""
    from hybridq_amd import aligned, core
    n = 10
    psi = aligned.zeros((2, 1 << n), dtype='float32', alignment=32)
    assert aligned.isaligned(psi[0], 32) and aligned.isaligned(psi[1], 32)
    U = np.eye(2, dtype=np.complex64)
    try:
        core.apply_U(psi[0], psi[1], U, [3], n)
    except core.HQError as e:
        assert 'aligned' not in str(e) and 'invalid positions' not in str(e)
    raw = np.zeros((1 << n) + 1, dtype=np.float32)[1:]  # 4 bytes off an aligned address
    if not aligned.isaligned(raw, 32):
        with pytest.raises(core.HQError, match='32-byte aligned'):
            core.apply_U(raw, psi[1], U, [3], n)
