"""Guard bands around every buffer a kernel writes ("memory errors"): the substitute for the part of
AddressSanitizer that can run on these boxes (the instrumented build needs XNACK, which they switch off).  Each operand of
every kernel family lives in the middle of a larger allocation whose margins hold a fixed bit pattern; after the call the
margins must be bit-identical and the result must equal the one computed on an ordinary tensor.  An out-of-range global
STORE lands in a margin (or faults); out-of-range LDS accesses are not visible to this test -- the exactness and
determinism tests are what covers those.

Written after GPU access had been closed for the round: sorts last on purpose, has not run on a GPU yet."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GUARD = 1 << 16  # elements on either side (a multiple of every alignment the library asks for)


class Guarded:
    """`shape` elements of `dtype` in the middle of a larger device allocation with patterned margins."""

    def __init__(self, torch, shape, dtype, fill=None):
        self.torch = torch
        numel = int(np.prod(shape))
        self.raw = torch.empty(numel + 2 * GUARD, dtype=dtype, device='cuda')
        esz = self.raw.element_size()
        self.lo, self.hi = GUARD * esz, (GUARD + numel) * esz  # the operand's bytes inside the allocation
        self.pattern = (torch.arange(2 * GUARD * esz, device='cuda', dtype=torch.int64) * 2654435761 % 251).to(torch.uint8)
        margins = self.raw.view(torch.uint8)
        margins[:self.lo] = self.pattern[:self.lo]
        margins[self.hi:] = self.pattern[self.lo:]
        self.t = self.raw[GUARD:GUARD + numel].view(shape)
        assert self.t.data_ptr() % 32 == 0
        if fill is not None:
            self.t.copy_(fill)

    def intact(self):
        m = self.raw.view(self.torch.uint8)
        return bool(self.torch.equal(m[:self.lo], self.pattern[:self.lo]) and self.torch.equal(m[self.hi:], self.pattern[self.lo:]))


def _state(torch, n, ft, seed):
    g = torch.Generator(device='cuda').manual_seed(seed)
    x = torch.randn((2, 1 << n), generator=g, device='cuda', dtype=ft)
    return x / x.norm()


@pytest.mark.parametrize('ct,n', [('complex64', 20), ('complex128', 19)])
def test_apply_kernels_stay_inside_their_planes(torch_cuda, ct, n):
    """apply_U for k = 1..8 in low / high / scattered positions (role, VALU, big, tile-GEMM kernels by the library's own
    dispatch and with the kernel forced) and cache-blocked passes: planes held in separately guarded allocations."""
    from hybridq_amd import core
    from hybridq_amd.blocking import plan_blocked
    from hybridq_amd.circuits import haar_unitary, rqc_1q2q
    torch = torch_cuda
    ft = torch.float32 if ct == 'complex64' else torch.float64
    rng = np.random.default_rng(n)
    base = _state(torch, n, ft, 7)
    re, im = Guarded(torch, (1 << n,), ft), Guarded(torch, (1 << n,), ft)
    plain = torch.empty_like(base)
    try:
        for mode in ('auto', 'direct', 'generic', 'tile', 'gemm'):
            core.set_apply_mode(mode)
            for k in range(1, 9):
                for pos in (list(range(k)), list(range(n - k, n)), sorted(int(p) for p in rng.permutation(n)[:k])):
                    U = np.ascontiguousarray(haar_unitary(1 << k, rng), dtype=ct)
                    re.t.copy_(base[0])
                    im.t.copy_(base[1])
                    plain.copy_(base)
                    core.apply_U(re.t, im.t, U, pos, n)
                    core.apply_U(plain[0], plain[1], U, pos, n)
                    core.sync()
                    what = (ct, mode, k, pos, core.last_kernel())
                    assert re.intact() and im.intact(), what
                    assert torch.equal(re.t, plain[0]) and torch.equal(im.t, plain[1]), what
    finally:
        core.set_apply_mode('auto')
    gates = rqc_1q2q(n, depth=10, seed=n)
    ident = {q: n - 1 - q for q in range(n)}
    for tb in ((13, 12) if ct == 'complex64' else (12, 11)):
        ops = plan_blocked(gates, ident, n, tile_bits=tb, low_bits=tb - 8, complex_type=ct)
        assert any(op[0] == 'B' for op in ops)
        re.t.copy_(base[0])
        im.t.copy_(base[1])
        plain.copy_(base)
        for op in ops:
            for a, b in ((re.t, im.t), (plain[0], plain[1])):
                if op[0] == 'B':
                    core.apply_blocked(a, b, op[1], op[2], n)
                else:
                    core.apply_U(a, b, op[1], op[2], n)
        core.sync()
        assert re.intact() and im.intact(), (ct, 'blocked', tb)
        assert torch.equal(re.t, plain[0]) and torch.equal(im.t, plain[1]), (ct, 'blocked', tb)


@pytest.mark.parametrize('dt,n', [('float32', 20), ('float64', 19), ('int32', 17), ('int64', 16)])
def test_data_movement_kernels_stay_inside_their_buffers(torch_cuda, dt, n):
    """swap_* for 3..16 moved bits (table kernel, one-pass tiles in place, SPLIT mode, two-pass), hq_permute_bits_* and the
    pack pass of hq_exchange_* (one rank) with source and destination in guarded allocations."""
    from hybridq_amd import core
    torch = torch_cuda
    tdt = getattr(torch, dt)
    rng = np.random.default_rng(n)
    data = (torch.arange(1 << n, device='cuda', dtype=torch.int64) % 1000003).to(tdt)
    g = Guarded(torch, (1 << n,), tdt)
    plain = torch.empty_like(data)
    for s in (3, 8, 12, 13, 14, 15, 16):
        if s > n:
            continue
        pos = np.roll(np.arange(s), 3) if s == 16 else rng.permutation(s)
        g.t.copy_(data)
        plain.copy_(data)
        core.swap(g.t, pos, n)
        core.swap(plain, pos, n)
        core.sync()
        assert g.intact(), (dt, 'swap', s, core.last_kernel())
        assert torch.equal(g.t, plain), (dt, 'swap', s)
    if dt not in ('float32', 'float64'):
        return
    dst = Guarded(torch, (1 << n,), tdt)
    for name, perm in (('random', rng.permutation(n)), ('reversal', np.arange(n)[::-1].copy()),
                       ('rotation', np.roll(np.arange(n), 7)), ('low 4 fixed', np.concatenate([np.arange(4), 4 + rng.permutation(n - 4)]))):
        g.t.copy_(data)
        dst.t.zero_()
        plain.zero_()
        core.permute_bits(g.t, dst.t, perm, n)
        core.permute_bits(data, plain, perm, n)
        core.sync()
        assert g.intact() and dst.intact(), (dt, 'permute_bits', name)
        assert torch.equal(dst.t, plain) and torch.equal(g.t, data), (dt, 'permute_bits', name)
    # exchange on one rank = the pack pass: (re, im) -> (re', im') under a permutation of the n - 1 local bits
    core.shard_free()
    m = n - 1
    half = data[:1 << m]
    src = [Guarded(torch, (1 << m,), tdt, half), Guarded(torch, (1 << m,), tdt, -half)]
    out = [Guarded(torch, (1 << m,), tdt), Guarded(torch, (1 << m,), tdt)]
    ref_src = torch.stack([half, -half]).contiguous()
    ref_dst = torch.empty_like(ref_src)
    perm = rng.permutation(m)
    in_src = core.exchange(src[0].t, src[1].t, out[0].t, out[1].t, perm, m)
    in_src_ref = core.exchange(ref_src[0], ref_src[1], ref_dst[0], ref_dst[1], perm, m)
    core.sync()
    assert in_src == in_src_ref
    assert all(x.intact() for x in src + out), (dt, 'exchange pack')
    got, exp = (src if in_src else out), (ref_src if in_src_ref else ref_dst)
    assert torch.equal(got[0].t, exp[0]) and torch.equal(got[1].t, exp[1]), (dt, 'exchange pack')


@pytest.mark.parametrize('ct,n', [('complex64', 20), ('complex128', 19)])
def test_auxiliary_kernels_stay_inside_their_buffers(torch_cuda, ct, n):
    """to_complex, the initial states, projection and the reductions on guarded planes / outputs."""
    from hybridq_amd import core
    torch = torch_cuda
    ft = torch.float32 if ct == 'complex64' else torch.float64
    cdt = torch.complex64 if ct == 'complex64' else torch.complex128
    base = _state(torch, n, ft, 11)
    re, im = Guarded(torch, (1 << n,), ft, base[0]), Guarded(torch, (1 << n,), ft, base[1])
    out = Guarded(torch, (1 << n,), cdt)
    core.to_complex(re.t, im.t, out.t)
    core.sync()
    assert re.intact() and im.intact() and out.intact()
    assert torch.equal(torch.view_as_real(out.t)[:, 0], base[0]) and torch.equal(torch.view_as_real(out.t)[:, 1], base[1])
    assert abs(core.norm2(re.t, im.t) - 1.0) < 1e-4  # (the state was normalised by torch in its own precision)
    p = core.probabilities(re.t, im.t, [2, n // 2, n - 1], n)
    assert abs(float(np.sum(p)) - 1.0) < 1e-4
    assert abs(core.vdot(re.t, im.t, re.t, im.t) - 1.0) < 1e-4
    core.project(re.t, im.t, [2, n // 2, n - 1], 5, 1.0, n)
    core.sync()
    assert re.intact() and im.intact()
    for kind, basis in (('basis', 5), ('plus', 0)):
        core.init_state(re.t, im.t, kind, basis)
        core.sync()
        assert re.intact() and im.intact(), kind
        assert abs(core.norm2(re.t, im.t) - 1.0) < 1e-5
    chars = {b: '01+-'[b % 4] for b in range(n)}
    core.init_product_state(re.t, im.t, chars)
    core.sync()
    assert re.intact() and im.intact()
    assert abs(core.norm2(re.t, im.t) - 1.0) < 1e-5
