"""GPU parity tests: the HIP library (through its C ABI) against the CPU oracle on the
same seeded inputs.  Bars (BASELINE.json north_star): max|d|/max|psi| <= 1e-6 for
complex64, <= 1e-12 for complex128; swaps and to_complex are bit-exact.

Mirrors the reference's own differential tests: tests.py:299-391 (dot: k=2..6, random
NON-unitary U, random axes), :256-296 (transpose/swap, exact equality, six dtypes),
:122-149 (to_complex)."""
import os

import numpy as np
import pytest

from tolerances import circuit_tol

pytestmark = pytest.mark.gpu

TOL = {np.dtype('float32'): 1e-6, np.dtype('float64'): 1e-12}
CT_OF = {np.dtype('float32'): 'complex64', np.dtype('float64'): 'complex128'}


def wide_tol(ft, k):
    """One k-qubit call, HIP vs the oracle in the SAME precision: the bar, or the rounding model of
    two 2^(k+1)-term accumulations (tests/tolerances.py) once that exceeds it (k >= 7 in float32).  ONE wide call is
    not a random walk over many gates: the figure is the maximum over 2^n amplitudes of a single 2^(k+1)-term sum with
    Gaussian matrix entries, i.e. the tail of that distribution -- measured constant 0.78 at k = 7 (1.05e-6 between the
    matrix-core kernel and the C oracle, n = 12) -- so this bound keeps c = 1 while the end-to-end tests use C_MODEL."""
    return circuit_tol([k], [k], complex_type=CT_OF[np.dtype(ft)], c=1.0)


def _rand_state(rng, n, ft):
    re = rng.standard_normal(1 << n).astype(ft)
    im = rng.standard_normal(1 << n).astype(ft)
    return re, im


def _rand_U(rng, k):
    d = 1 << k
    return (rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d))) / np.sqrt(2.0 * d)


def _oracle_apply(lib, re, im, U, pos):
    from oracle.binding import aligned_empty
    pl = aligned_empty((2, re.size), re.dtype)
    pl[0], pl[1] = re, im
    assert lib.apply_U(pl[0], pl[1], U, pos) == 0
    return pl[0].copy(), pl[1].copy()


def _gpu_apply(torch, re, im, U, pos, mode='auto'):
    from hybridq_amd import core
    core.set_stream(torch.cuda.current_stream().cuda_stream)
    core.set_apply_mode(mode)
    try:
        # separate allocations: for tiny n a stacked (2, 2^n) tensor would leave the
        # imaginary plane short of the 32-byte alignment the ABI requires (U.h:34-36)
        dre, dim_ = torch.from_numpy(re).cuda(), torch.from_numpy(im).cuda()
        core.apply_U(dre, dim_, U, pos)
        core.sync()
        kern = core.last_kernel()
        out = dre.cpu().numpy(), dim_.cpu().numpy()
    finally:
        core.set_apply_mode('auto')
    return out[0], out[1], kern


def _relerr(gr, gi, orr, oi):
    scale = max(np.abs(orr).max(), np.abs(oi).max())
    return max(np.abs(gr - orr).max(), np.abs(gi - oi).max()) / scale


# positions chosen to hit: in-vector targets (0,1), wave-lane bits (2..7), block bits,
# high bits, unsorted orders, adjacent and spread targets
POS_CASES = {
    1: [[0], [1], [2], [3], [5], [7], [8], [11], [17]],
    2: [[0, 1], [1, 0], [0, 5], [9, 1], [2, 3], [4, 12], [12, 4], [16, 17], [17, 8], [6, 7]],
    3: [[0, 1, 2], [2, 1, 0], [0, 7, 13], [1, 2, 17], [3, 4, 5], [15, 9, 11], [17, 16, 15], [1, 0, 9]],
    4: [[0, 1, 2, 3], [5, 6, 7, 8], [7, 6, 8, 11], [0, 9, 13, 17], [14, 15, 16, 17], [1, 12, 3, 16]],
    5: [[0, 1, 2, 3, 4], [3, 8, 9, 12, 17], [17, 2, 11, 5, 9]],
    6: [[1, 3, 5, 7, 9, 11], [12, 13, 14, 15, 16, 17]],
}


@pytest.mark.parametrize('ft', ['float32', 'float64'])
@pytest.mark.parametrize('k', [1, 2, 3, 4, 5, 6])
def test_apply_U_matches_oracle(torch_cuda, oracle_port, ft, k):
    ft = np.dtype(ft)
    n = 18
    rng = np.random.default_rng(1000 + k)
    for pos in POS_CASES[k]:
        re, im = _rand_state(rng, n, ft)
        U = _rand_U(rng, k)
        orr, oi = _oracle_apply(oracle_port, re, im, U, pos)
        gr, gi, kern = _gpu_apply(torch_cuda, re, im, U, pos)
        err = _relerr(gr, gi, orr, oi)
        assert err <= TOL[ft], (ft, k, pos, kern, err)
        assert kern == 'mfma', kern  # matrix-core role kernels (f32 and f64), any target position, k <= 6
        if k >= 5:  # the LDS-staged tile GEMM (second implementation for k = 5, 6)
            gr, gi, kern = _gpu_apply(torch_cuda, re, im, U, pos, mode='tile')
            assert kern == 'mfma_tile', kern
            assert _relerr(gr, gi, orr, oi) <= TOL[ft], (ft, k, pos, kern)


@pytest.mark.parametrize('ft', ['float32', 'float64'])
def test_apply_U_mfma_kernels(torch_cuda, oracle_port, ft):
    """k=1..4 on the matrix cores (real-embedded f32 / f64 MFMA, role-assigned index
    digits): every role combination -- targets in the vector components (bits 0,1 / bit 0),
    in the permuted lane range, high, unsorted -- with non-unitary U."""
    ft = np.dtype(ft)
    n = 18
    rng = np.random.default_rng(77)
    cases = {
        1: [[p] for p in range(n)],
        2: [[0, 1], [1, 0], [0, 2], [1, 5], [2, 3], [3, 2], [4, 5], [2, 17], [17, 3], [0, 17], [6, 7], [9, 13], [16, 17]],
        3: [[0, 1, 2], [2, 1, 0], [0, 1, 17], [0, 2, 3], [1, 4, 9], [2, 3, 4], [4, 3, 2], [2, 9, 17], [17, 5, 11],
            [7, 8, 9], [15, 16, 17], [10, 2, 6], [0, 7, 13], [3, 4, 5]],
        4: [[0, 1, 2, 3], [3, 2, 1, 0], [0, 1, 9, 17], [0, 2, 3, 4], [1, 5, 6, 12], [2, 3, 4, 5], [5, 4, 3, 2],
            [7, 6, 8, 11], [2, 9, 13, 17], [14, 15, 16, 17], [12, 3, 16, 8], [0, 9, 13, 17], [1, 12, 3, 16]],
    }
    for k, plist in cases.items():
        for pos in plist:
            re, im = _rand_state(rng, n, ft)
            U = _rand_U(rng, k)
            orr, oi = _oracle_apply(oracle_port, re, im, U, pos)
            gr, gi, kern = _gpu_apply(torch_cuda, re, im, U, pos, mode='mfma')
            assert kern == 'mfma', (kern, k, pos)
            assert _relerr(gr, gi, orr, oi) <= TOL[ft], (k, pos)
    # k = 5, 6 (apply_mfma_big_kernel: A operands through LDS, grid-stride): every component /
    # lane / high-bit role combination, unsorted targets, n large enough for several workgroups
    big = {
        5: [[0, 1, 2, 3, 4], [4, 3, 2, 1, 0], [0, 1, 9, 13, 17], [1, 2, 3, 4, 5], [0, 5, 6, 7, 8], [2, 3, 4, 5, 6],
            [6, 2, 5, 3, 4], [7, 8, 9, 10, 11], [13, 14, 15, 16, 17], [2, 9, 13, 16, 17], [17, 0, 8, 3, 12], [1, 6, 11, 16, 4]],
        6: [[0, 1, 2, 3, 4, 5], [5, 4, 3, 2, 1, 0], [0, 1, 8, 11, 14, 17], [1, 2, 3, 4, 5, 6], [0, 3, 6, 9, 12, 15],
            [2, 3, 4, 5, 6, 7], [7, 3, 6, 2, 5, 4], [8, 9, 10, 11, 12, 13], [12, 13, 14, 15, 16, 17], [2, 5, 9, 13, 16, 17],
            [17, 0, 8, 3, 12, 1], [1, 6, 11, 16, 4, 9]],
    }
    for nn in (18, 21):
        for k, plist in big.items():
            for pos in plist:
                pos = [p if p < 12 else p + (nn - 18) for p in pos]
                re, im = _rand_state(rng, nn, ft)
                U = _rand_U(rng, k)
                orr, oi = _oracle_apply(oracle_port, re, im, U, pos)
                gr, gi, kern = _gpu_apply(torch_cuda, re, im, U, pos, mode='mfma')
                assert kern == 'mfma', (kern, k, pos)
                assert _relerr(gr, gi, orr, oi) <= TOL[ft], (nn, k, pos)
    # smallest states the matrix-core path accepts, and the fallback below that
    for nn in (8, 9, 10, 11, 12, 13):
        for k in (1, 2, 3, 4, 5, 6):
            if k > nn - 2:
                continue
            pos = rng.permutation(nn)[:k]
            re, im = _rand_state(rng, nn, ft)
            U = _rand_U(rng, k)
            orr, oi = _oracle_apply(oracle_port, re, im, U, pos)
            gr, gi, kern = _gpu_apply(torch_cuda, re, im, U, pos)
            assert _relerr(gr, gi, orr, oi) <= TOL[ft], (nn, k, list(pos), kern)


@pytest.mark.parametrize('mode', ['generic', 'naive'])
@pytest.mark.parametrize('ft', ['float32', 'float64'])
def test_apply_U_alternative_kernels(torch_cuda, oracle_port, ft, mode):
    ft = np.dtype(ft)
    n = 14
    rng = np.random.default_rng(7)
    for k in (1, 2, 3, 4, 5, 7):
        for trial in range(3):
            pos = rng.permutation(n)[:k]
            re, im = _rand_state(rng, n, ft)
            U = _rand_U(rng, k)
            orr, oi = _oracle_apply(oracle_port, re, im, U, pos)
            gr, gi, kern = _gpu_apply(torch_cuda, re, im, U, pos, mode=mode)
            assert kern == mode
            assert _relerr(gr, gi, orr, oi) <= TOL[ft], (k, list(pos))


@pytest.mark.parametrize('ft', ['float32', 'float64'])
def test_apply_U_small_and_edge_sizes(torch_cuda, oracle_port, ft):
    """Ragged / tiny states: every n from 1, k up to n (full-state matrix), k = 0 no-op."""
    ft = np.dtype(ft)
    rng = np.random.default_rng(11)
    for n in range(1, 13):
        for k in range(1, min(n, 6) + 1):
            pos = rng.permutation(n)[:k]
            re, im = _rand_state(rng, n, ft)
            U = _rand_U(rng, k)
            orr, oi = _oracle_apply(oracle_port, re, im, U, pos)
            gr, gi, kern = _gpu_apply(torch_cuda, re, im, U, pos)
            assert _relerr(gr, gi, orr, oi) <= TOL[ft], (n, k, list(pos), kern)
    # k = 0: no-op, returns 0 (python_U.cpp:38-39)
    re, im = _rand_state(rng, 8, ft)
    gr, gi, _ = _gpu_apply(torch_cuda, re, im, np.ones((1, 1)), [])
    assert (gr == re).all() and (gi == im).all()


def test_apply_U_large_k(torch_cuda, oracle_port):
    """k = 8 (simulate(compress=8), tests.py:2355) and k = 10 (dot.py:236 cap)."""
    rng = np.random.default_rng(5)
    for ft, n, k in (('float32', 16, 8), ('float64', 14, 8), ('float32', 14, 10)):
        ft = np.dtype(ft)
        pos = rng.permutation(n)[:k]
        re, im = _rand_state(rng, n, ft)
        U = _rand_U(rng, k)
        orr, oi = _oracle_apply(oracle_port, re, im, U, pos)
        gr, gi, kern = _gpu_apply(torch_cuda, re, im, U, pos)
        assert _relerr(gr, gi, orr, oi) <= wide_tol(ft, k), (k, kern)


@pytest.mark.parametrize('ft', ['float32', 'float64'])
def test_apply_U_gemm_kernel(torch_cuda, oracle_port, ft):
    """k = 7..10 on the matrix cores (apply_gemm_kernel: LDS tile as B operand, A operands from a
    host-built table, per-gate LDS swizzle): low / high / scattered / unsorted targets, targets
    on index bits 0 and 1, several tiles per workgroup, non-unitary U; generic kernel as the
    second implementation."""
    ft = np.dtype(ft)
    rng = np.random.default_rng(707)
    kmax = 10 if ft == np.dtype('float32') else 9
    tb = 14 if ft == np.dtype('float32') else 13  # largest tile; k = 7, 8 use smaller ones (more workgroups per CU)
    for n in (tb - 2, tb, tb + 1, tb + 4):
        for k in range(7, kmax + 1):
            cases = [list(range(k)), list(range(n - k, n)), list(range(2, 2 + k)),
                     [int(p) for p in rng.permutation(n)[:k]], [int(p) for p in rng.permutation(n)[:k]],
                     [1] + list(range(5, 4 + k)), [0] + [int(p) for p in 2 + rng.permutation(n - 2)[:k - 1]]]
            if n > tb + 1:
                cases = cases[:1] + cases[3:5]
            for pos in cases:
                if max(pos) >= n or len(set(pos)) != k:
                    continue
                re, im = _rand_state(rng, n, ft)
                U = _rand_U(rng, k)
                orr, oi = _oracle_apply(oracle_port, re, im, U, pos)
                gr, gi, kern = _gpu_apply(torch_cuda, re, im, U, pos)
                min_n = min(tb, k + (5 if ft == np.dtype('float32') else 4))
                assert kern == ('gemm' if n >= min_n else 'generic'), (kern, n, k, pos)
                assert _relerr(gr, gi, orr, oi) <= wide_tol(ft, k), (n, k, pos)
    # below the tile size the LDS-tile VALU kernel takes over; 'generic' can always be forced
    re, im = _rand_state(rng, 10, ft)
    U = _rand_U(rng, 7)
    pos = [int(p) for p in rng.permutation(10)[:7]]
    orr, oi = _oracle_apply(oracle_port, re, im, U, pos)
    gr, gi, kern = _gpu_apply(torch_cuda, re, im, U, pos)
    assert kern == 'generic' and _relerr(gr, gi, orr, oi) <= wide_tol(ft, 7)


def test_apply_U_error_codes(torch_cuda):
    import ctypes
    from hybridq_amd import core
    torch = torch_cuda
    planes = torch.zeros((2, 1 << 10), dtype=torch.float32, device='cuda')
    U = np.eye(2, dtype=np.complex64)

    def call(pos, n=10, re=planes[0], im=planes[1]):
        pos = np.asarray(pos, dtype=np.uint32)
        return core._dot_core[np.dtype('float32')](core._ptr(re), core._ptr(im), U.ctypes.data,
                                                   pos.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)),
                                                   n, len(pos))

    assert call([3]) == 0
    assert call([10]) != 0  # position out of range
    assert call([3, 3]) != 0  # duplicate positions
    assert call([3], re=planes[0][1:]) != 0  # misaligned plane (U.h:34-36)
    assert 'aligned' in core.last_error()
    with pytest.raises(core.HQError):
        core.apply_U(planes[0], planes[1], np.eye(2), [12])


@pytest.mark.parametrize('ft', ['float32', 'float64'])
def test_apply_U_host_pointer_path(torch_cuda, oracle_port, ft):
    """Host numpy planes (what the unmodified reference Python passes) are staged."""
    from hybridq_amd import core
    from oracle.binding import aligned_empty
    ft = np.dtype(ft)
    rng = np.random.default_rng(3)
    n = 15
    for k in (1, 2, 4, 5):
        pos = rng.permutation(n)[:k]
        re, im = _rand_state(rng, n, ft)
        U = _rand_U(rng, k)
        orr, oi = _oracle_apply(oracle_port, re, im, U, pos)
        pl = aligned_empty((2, 1 << n), ft, alignment=64)
        pl[0], pl[1] = re, im
        core.apply_U(pl[0], pl[1], U, pos)
        assert _relerr(pl[0], pl[1], orr, oi) <= TOL[ft]


@pytest.mark.parametrize('dt', ['float32', 'float64', 'int32', 'int64', 'uint32', 'uint64'])
def test_swap_exact(torch_cuda, oracle_port, dt):
    """tests.py:256-296: exact equality for all six dtypes; s from 1 to n (LDS path and the
    out-of-place gather path), device and host pointers."""
    from hybridq_amd import core
    import oracle
    torch = torch_cuda
    dt = np.dtype(dt)
    rng = np.random.default_rng(17)
    n = 16
    for s in (1, 2, 3, 5, 6, 8, 9, 11, 12, 13, 14, 16):
        a = rng.integers(0, 2**31 - 1, 1 << n).astype(dt)
        pos = rng.permutation(s)
        exp = oracle.swap_numpy(a, pos)
        tdt = getattr(torch, dt.name) if dt.name not in ('uint32', 'uint64') else None
        if tdt is not None:
            t = torch.from_numpy(a.copy()).cuda()
            core.swap(t, pos)
            core.sync()
            assert (t.cpu().numpy() == exp).all(), (dt, s)
        h = a.copy()
        core.swap(h, pos)  # host path
        assert (h == exp).all(), (dt, s, 'host')
    # against numpy.transpose, the reference's own check
    s = 6
    a = rng.integers(0, 1000, 1 << n).astype(dt)
    pos = rng.permutation(s)
    h = a.copy()
    core.swap(h, pos)
    tr = np.transpose(a.reshape((2,) * n),
                      list(range(n - s)) + [n - 1 - int(pos[i]) for i in reversed(range(s))])
    assert (h == tr.reshape(-1)).all()


def test_swap_rejects_non_permutation(torch_cuda):
    import ctypes
    from hybridq_amd import core
    a = np.zeros(1 << 8, dtype=np.float32)
    pos = np.asarray([0, 0, 1], dtype=np.uint32)
    rc = core._swap_core[np.dtype('float32')](a.ctypes.data, pos.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), 8, 3)
    assert rc != 0


@pytest.mark.parametrize('ft', ['float32', 'float64'])
def test_to_complex_exact(torch_cuda, ft):
    from hybridq_amd import core
    torch = torch_cuda
    ft = np.dtype(ft)
    ct = np.dtype('complex64') if ft == np.dtype('float32') else np.dtype('complex128')
    rng = np.random.default_rng(2)
    for size in (1 << 12, 1 << 17, 1000, 7):
        re = rng.standard_normal(size).astype(ft)
        im = rng.standard_normal(size).astype(ft)
        out = np.empty(size, dtype=ct)
        core.to_complex(re, im, out)  # host path
        assert (out == re + 1j * im).all()
        dre, dim_ = torch.from_numpy(re).cuda(), torch.from_numpy(im).cuda()
        dout = torch.empty(size, dtype=getattr(torch, ct.name), device='cuda')
        core.to_complex(dre, dim_, dout)
        core.sync()
        assert (dout.cpu().numpy() == re + 1j * im).all()


@pytest.mark.parametrize('ct', ['complex64', 'complex128'])
def test_simulate_matches_reference_protocol(torch_cuda, oracle_port, ct):
    """End to end: hybridq_amd.simulate (no swaps, state in HBM) == the reference driver
    protocol (swap policy + apply_U, simulation.py:491-675) replayed on the CPU oracle."""
    import oracle
    from hybridq_amd.simulation import simulate
    from hybridq_amd.circuits import random_dense, rqc_1q2q
    n = 16
    for gates in (random_dense(n, 80, kmax=4, seed=1), rqc_1q2q(n, depth=8, seed=2)):
        exp, _ = oracle.evolve_reference_protocol(oracle_port, gates, n, complex_type=ct)
        tol = 1e-6 if ct == 'complex64' else 1e-12
        for compress in (0, 4, 6):  # as given / the reference's default fusion / k>4 generic path
            psi, info = simulate(gates, initial_state='0' * n, complex_type=ct, return_info=True,
                                 qubits=list(range(n)), compress=compress)
            psi = psi.reshape(-1)
            assert np.abs(psi - exp).max() / np.abs(exp).max() <= tol * (1 if compress == 0 else 2)
            assert info['runtime (s)'] > 0 and info['n_gates_given'] == len(gates)
            assert (info['n_gates'] == len(gates)) == (compress == 0)


def test_simulate_initial_states(torch_cuda):
    from hybridq_amd.simulation import simulate
    n = 12
    ident = [(np.eye(2), (q,)) for q in range(n)]
    # like the reference (simulation.py:301-305): simplification removes the identities, the active
    # qubits change and the run is stopped
    with pytest.raises(ValueError):
        simulate(ident, initial_state='+' * n)
    psi = simulate(ident, initial_state='+' * n, simplify=False)
    assert np.allclose(psi, 2**(-n / 2))
    s = '0110' * 3
    psi = simulate(ident, initial_state=s, simplify=False).reshape(-1)
    assert psi[int(s, 2)] == 1 and np.abs(psi).sum() == 1
    psi = simulate(ident, initial_state='+-01' * 3, qubits=list(range(n))).reshape(-1)  # explicit qubits: nothing to apply
    exp = np.ones(1)
    single = {'0': [1, 0], '1': [0, 1], '+': [2**-0.5, 2**-0.5], '-': [2**-0.5, -2**-0.5]}
    for c in '+-01' * 3:
        exp = np.kron(exp, single[c])
    assert np.allclose(psi, exp, atol=1e-6)


def test_permute_bits_exact(torch_cuda):
    """hq_permute_bits: dst[x] = src[pi(x)] for arbitrary bit permutations (vector and
    scalar paths, 4- and 8-byte elements), bit-exact against numpy."""
    from hybridq_amd import core
    torch = torch_cuda
    rng = np.random.default_rng(9)
    n = 16
    x = np.arange(1 << n, dtype=np.int64)
    for dt in (torch.float32, torch.float64):
        for trial in range(6):
            perm = np.arange(n)
            moved = rng.permutation(n if trial % 2 else np.arange(2, n))[:int(rng.integers(2, 7))]
            perm[np.sort(moved)] = moved
            if sorted(perm) != list(range(n)):
                continue
            src = torch.randn(1 << n, dtype=dt, device='cuda')
            dst = torch.empty_like(src)
            core.permute_bits(src, dst, perm)
            core.sync()
            y = np.zeros_like(x)
            for i, p in enumerate(perm):
                y |= ((x >> i) & 1) << int(p)
            assert (dst.cpu().numpy() == src.cpu().numpy()[y]).all(), (dt, list(perm))
    with pytest.raises(core.HQError):
        core.permute_bits(src, src, np.arange(n))  # must be out of place
    with pytest.raises(core.HQError):
        core.permute_bits(src, dst, np.zeros(n))  # not a permutation


def test_sharded_single_rank_matches_oracle(torch_cuda, oracle_port):
    """hybridq_amd.dist with the HIP backend on one rank (g = 0): same planner/runtime code
    path as the multi-GPU run minus the exchanges, plus a forced P op."""
    import oracle
    from hybridq_amd.circuits import random_dense
    from hybridq_amd.dist import ShardedEvolution
    n = 14
    gates = random_dense(n, 60, kmax=4, seed=3)
    sh = ShardedEvolution(n, complex_type='complex64', initial_state='0' * n)
    sched = sh.plan(gates)
    assert all(op[0] == 'G' for op in sched)
    sh.run(sched)
    exp = oracle.evolve_tensordot(gates, n)
    psi = sh.state_numpy()
    assert np.abs(psi - exp).max() / np.abs(exp).max() < 1e-6
    assert abs(sh.norm2() - float(np.vdot(exp, exp).real)) < 1e-4 * float(np.vdot(exp, exp).real)


def test_simulate_functional_gate_branch(torch_cuda, oracle_port):
    """simulation.py:525-554: a FunctionalGate receives the raw (2,)+(2,)*n split-plane state
    and the qubit order; matrix gates around it are fused separately (never across it)."""
    import oracle
    from hybridq_amd.circuits import random_dense
    from hybridq_amd.simulation import FunctionalGate, simulate
    n = 12
    seen = {}

    def project_q3_to_1(psi, order):  # numpy code in the style of gate/projection.py:72-119
        seen['shape'], seen['order'] = psi.shape, order
        ax = order.index(3)
        new = np.zeros_like(psi)
        idx = [slice(None)] * psi.ndim
        idx[ax + 1] = 1
        new[tuple(idx)] = psi[tuple(idx)]
        new /= np.linalg.norm(new.ravel())
        return new, order

    g1, g2 = random_dense(n, 30, kmax=3, seed=1, unitary=True), random_dense(n, 30, kmax=3, seed=2, unitary=True)
    circuit = list(g1) + [FunctionalGate((3,), project_q3_to_1)] + list(g2)
    psi, info = simulate(circuit, initial_state='+' * n, complex_type='complex128', return_info=True, compress=4)
    assert seen['shape'] == (2,) + (2,) * n and seen['order'] == tuple(range(n))
    a = oracle.evolve_tensordot(g1, n, initial_state=np.full(1 << n, 2.0**(-n / 2)), qubits=list(range(n)))
    a = a.reshape((2,) * n).copy()
    idx = [slice(None)] * n
    idx[3] = 0
    a[tuple(idx)] = 0
    a /= np.linalg.norm(a.ravel())
    exp = oracle.evolve_tensordot(g2, n, initial_state=a.reshape(-1), qubits=list(range(n)))
    assert np.abs(psi.reshape(-1) - exp).max() / np.abs(exp).max() < 1e-12
    assert info['n_gates'] < len(circuit)  # fused on both sides of the functional gate


@pytest.mark.parametrize('t', ['float32', 'float64'])
@pytest.mark.parametrize('k', [2, 3, 4, 5, 6])
def test_dot_api(torch_cuda, t, k):
    """Reference tests.py:299-391 (test_utils__dot) against hybridq_amd.dot: split-plane and
    complex inputs, inplace, swap_back=False, vs the explicit numpy path."""
    from hybridq_amd.dot import aligned_empty, dot
    from hybridq_amd.transpose import transpose
    rng = np.random.default_rng(100 + k)
    n = 14
    tol = dict(rtol=1e-3, atol=1e-3)  # the reference's own bar (tests.py:56-62) ...
    # ... and ours: one k-qubit call on each side (numpy's BLAS in the same precision sums in another order)
    tight = wide_tol(t, k)
    for _ in range(3):
        psi = rng.random((2, 2**n)).astype(t)
        psi = (psi.T / np.linalg.norm(psi, axis=1)).T
        psi1 = np.array(psi)
        psi2 = aligned_empty(psi.shape, t)
        psi2[...] = psi
        U = (rng.random((2**k, 2**k)) + 1j * rng.random((2**k, 2**k))).astype((1j * psi[0][:1]).dtype)
        axes_b = rng.choice(n, size=k, replace=False)
        shp = (2,) * (n + 1)
        b1 = dot(U, np.reshape(psi1, shp), axes_b=axes_b, b_as_complex_array=True, force_numpy=True)
        b1h = dot(U, np.reshape(psi1, shp), axes_b=axes_b, b_as_complex_array=True, raise_if_hcore_fails=True)
        p2 = dot(U, np.reshape(psi2, shp), axes_b=axes_b, b_as_complex_array=True, inplace=True,
                 raise_if_hcore_fails=True)
        np.testing.assert_allclose(psi, psi1)  # not modified unless inplace
        np.testing.assert_allclose(b1, b1h, **tol)
        assert np.abs(np.asarray(b1) - b1h).max() < tight * np.abs(b1h).max()
        np.testing.assert_allclose(p2, b1h, **tol)
        assert np.shares_memory(p2, psi2)
        no_tr, tr1 = dot(U, np.reshape(psi, shp), axes_b=axes_b, b_as_complex_array=True, swap_back=False,
                         raise_if_hcore_fails=True)
        assert tr1 is None
        np.testing.assert_allclose(b1, no_tr, **tol)
        c = np.reshape(psi[0] + 1j * psi[1], (2,) * n)
        b2 = dot(U, c, axes_b=axes_b, force_numpy=True)
        b2h = dot(U, c, axes_b=axes_b, raise_if_hcore_fails=True)
        np.testing.assert_allclose(b2, b2h, **tol)
        assert np.abs(b2 - b2h).max() < tight * np.abs(b2h).max()
    # device-resident planes
    torch = torch_cuda
    d = torch.from_numpy(np.reshape(psi, shp)).cuda()
    out = dot(U, d, axes_b=axes_b, b_as_complex_array=True, inplace=True)
    assert out.data_ptr() == d.data_ptr()
    np.testing.assert_allclose(d.cpu().numpy(), b1h, **tol)


@pytest.mark.parametrize('t', ['float32', 'float64', 'int32', 'int64', 'uint32', 'uint64'])
def test_transpose_api(torch_cuda, t):
    """Reference tests.py:256-296 (test_utils__transpose): exact equality with np.transpose."""
    from hybridq_amd.transpose import transpose
    rng = np.random.default_rng(5)
    n = 14
    v = np.reshape(rng.integers(0, 2**31 - 1, 2**n).astype(t), (2,) * n)
    v0 = np.array(v)
    axes = rng.permutation(n)
    v1 = transpose(v, axes, force_numpy=True)
    v1h = transpose(v, axes, raise_if_hcore_fails=True)
    assert (v == v0).all() and (v1 == v1h).all()
    v2 = np.array(v)
    v2h = transpose(v2, axes, inplace=True, raise_if_hcore_fails=True)
    assert (v1 == v2h).all() and np.shares_memory(v2, v2h)
    axes = np.concatenate([np.arange(n - 6), n - 6 + rng.permutation(6)])
    assert (transpose(v, axes, force_numpy=True) == transpose(v, axes, raise_if_hcore_fails=True)).all()


@pytest.mark.parametrize('ct', ['complex64', 'complex128'])
def test_device_projection_and_measure(torch_cuda, ct):
    """Device-side Projection / Measure (gate/projection.py, gate/measure.py semantics) vs
    numpy on the gathered state: probabilities, collapse, renormalisation, inside simulate."""
    import oracle
    from hybridq_amd import core
    from hybridq_amd.circuits import random_dense
    from hybridq_amd.functional import Measure, Projection
    from hybridq_amd.simulation import EvolutionState, simulate
    n = 12
    g1 = random_dense(n, 25, kmax=3, seed=4, unitary=True)
    g2 = random_dense(n, 25, kmax=3, seed=5, unitary=True)
    tol = circuit_tol(g1 + g2, complex_type=ct)  # vs a complex128 evolution; derived quantities carry their own factor
    psi1 = oracle.evolve_tensordot(g1, n, initial_state=np.full(1 << n, 2.0**(-n / 2)), qubits=list(range(n)))
    t = psi1.reshape((2,) * n)
    # marginal probabilities, qubits[0] = most significant outcome bit, arbitrary order
    st = EvolutionState(list(range(n)), complex_type=ct, initial_state='+' * n)
    for U, qs in g1:
        st.apply(U.astype(ct), qs)
    for qs in ((3,), (7, 2), (0, 11, 5), (9, 1, 4, 10)):
        m = Measure(qs)
        probs = m.probabilities(st)
        ax = list(qs) + [a for a in range(n) if a not in qs]
        exp = (np.abs(np.transpose(t, ax).reshape(1 << len(qs), -1))**2).sum(1)
        assert np.abs(probs - exp).max() < 2 * tol  # d|a|^2 = 2 |a| d|a|
    # projection inside a circuit
    circuit = list(g1) + [Projection('10', (7, 2))] + list(g2)
    psi = simulate(circuit, initial_state='+' * n, complex_type=ct).reshape(-1)
    a = t.copy()
    keep = np.zeros((2,) * n, dtype=bool)
    idx = [slice(None)] * n
    idx[7], idx[2] = 1, 0
    keep[tuple(idx)] = True
    a = np.where(keep, a, 0)
    a /= np.linalg.norm(a.ravel())
    exp = oracle.evolve_tensordot(g2, n, initial_state=a.reshape(-1), qubits=list(range(n)))
    assert np.abs(psi - exp).max() / np.abs(exp).max() < 2 * tol
    # measurement: outcome is recorded, state collapses onto it with norm 1
    m = Measure((4, 8), rng=np.random.default_rng(3))
    psi = simulate(list(g1) + [m], initial_state='+' * n, complex_type=ct).reshape((2,) * n)
    o = m.outcome
    assert 0 <= o < 4 and abs(np.linalg.norm(psi.ravel()) - 1) < 10 * tol
    b4, b8 = (o >> 1) & 1, o & 1
    idx = [slice(None)] * n
    idx[4], idx[8] = 1 - b4, slice(None)
    assert np.abs(psi[tuple(idx)]).max() == 0
    idx[4], idx[8] = b4, b8
    ref = t[tuple(idx)] / np.linalg.norm(t[tuple(idx)].ravel())
    assert np.abs(psi[tuple(idx)] - ref).max() < 10 * tol
    # sampling statistics follow the probabilities
    st2 = EvolutionState(list(range(n)), complex_type=ct, initial_state='+' * n)
    for U, qs in g1:
        st2.apply(U.astype(ct), qs)
    mm = Measure((6,), rng=np.random.default_rng(0))
    p = mm.probabilities(st2)
    draws = np.array([mm.sample(p) for _ in range(4000)])
    assert abs(draws.mean() - p[1]) < 0.05


@pytest.mark.parametrize('ct', ['complex64', 'complex128'])
def test_expectation_value(torch_cuda, ct):
    """simulation.py:1125-1216: sum(conj(state) * (op state)), reduced on the device."""
    import oracle
    from hybridq_amd.circuits import random_dense
    from hybridq_amd.simulation import expectation_value
    rng = np.random.default_rng(8)
    n = 12
    psi = rng.standard_normal(1 << n) + 1j * rng.standard_normal(1 << n)
    psi /= np.linalg.norm(psi)
    op = random_dense(n, 12, kmax=3, seed=9)
    used = sorted({q for _, qs in op for q in qs})
    val = expectation_value(psi.reshape((2,) * n), op, qubits_order=list(range(n)), complex_type=ct)
    exp = np.vdot(psi, oracle.evolve_tensordot(op, n, initial_state=psi, qubits=list(range(n))))
    tol = 2e-6 if ct == 'complex64' else 1e-12
    assert abs(val - exp) < tol * max(1, abs(exp)), (val, exp, used)
    # Hermitian operator -> real expectation value (np.real_if_close in the reference)
    z = np.diag([1.0, -1.0])
    v = expectation_value(psi.reshape((2,) * n), [(z, (3,)), (z, (7,))], qubits_order=list(range(n)),
                          complex_type='complex128')
    t = np.abs(psi.reshape((2,) * n))**2
    sign = np.ones((2,) * n)
    idx = [slice(None)] * n
    for q in (3, 7):
        sl = list(idx)
        sl[q] = 1
        sign[tuple(sl)] *= -1
    assert isinstance(v, float) and abs(v - (t * sign).sum()) < 1e-12
    with pytest.raises(ValueError):
        expectation_value(psi.reshape((2,) * n), [(z, (99,))], qubits_order=list(range(n)))


@pytest.mark.parametrize('ft', ['float32', 'float64'])
def test_apply_blocked_matches_gate_by_gate(torch_cuda, oracle_port, ft):
    """hq_apply_blocked_float32: a list of k<=4 gates inside one LDS tile in ONE pass == the
    same gates applied one by one by the oracle (non-unitary U, targets in the vector
    components, in the low bits and among the high tile bits; several tile sizes)."""
    from hybridq_amd import core
    from oracle.binding import aligned_empty
    torch = torch_cuda
    core.set_stream(torch.cuda.current_stream().cuda_stream)
    rng = np.random.default_rng(21)
    ft = np.dtype(ft)
    shapes = ((18, 13, 14), (17, 14, 9), (16, 13, 30), (14, 10, 6), (13, 13, 5), (20, 12, 8))
    if ft == np.dtype('float64'):
        shapes = ((18, 12, 14), (17, 13, 9), (16, 12, 30), (14, 10, 6), (20, 11, 8))
    for n, tb, ngates in shapes:
        high = np.sort(rng.permutation(np.arange(5, n))[:tb - 5]) if n > tb else np.arange(5, n)
        tile = np.concatenate([np.arange(5), high]).astype(np.uint32)
        assert len(tile) == tb
        gates = []
        for _ in range(ngates):
            k = int(rng.integers(1, 5))
            pos = rng.permutation(tile)[:k]
            gates.append((_rand_U(rng, k) * (1.0 if k > 1 else 1.3), pos))
        re, im = _rand_state(rng, n, ft)
        pl = aligned_empty((2, 1 << n), ft)
        pl[0], pl[1] = re, im
        for U, pos in gates:
            assert oracle_port.apply_U(pl[0], pl[1], U, pos) == 0
        dre, dim_ = torch.from_numpy(re).cuda(), torch.from_numpy(im).cuda()
        core.apply_blocked(dre, dim_, tile, gates)
        core.sync()
        assert core.last_kernel() == 'blocked'
        err = _relerr(dre.cpu().numpy(), dim_.cpu().numpy(), pl[0], pl[1])
        assert err <= circuit_tol(gates, gates, complex_type=CT_OF[ft]), (n, tb, ngates, err)
    # argument validation
    with pytest.raises(core.HQError):
        core.apply_blocked(dre, dim_, np.arange(2, 15), gates[:1])  # tile without bits 0, 1
    with pytest.raises(core.HQError):
        core.apply_blocked(dre, dim_, np.arange(12), [(np.eye(2), [15])])  # target outside the tile


def test_simulate_blocked_matches_oracle(torch_cuda, oracle_port, monkeypatch):
    """simulate(blocked=True): the cache-blocked schedule (many gates per LDS-tile pass, inner
    fusion) gives the same state as gate-by-gate evolution; far fewer passes than gates."""
    import oracle
    from hybridq_amd.blocking import blocked_stats, plan_blocked
    from hybridq_amd.circuits import random_dense, rqc_1q2q
    from hybridq_amd.simulation import FunctionalGate, simulate
    for ct, tol in (('complex128', 1e-12),):
        g = rqc_1q2q(18, depth=12, seed=3)
        psi, info = simulate(g, initial_state='0' * 18, complex_type=ct, blocked=True, return_info=True,
                             qubits=list(range(18)))
        exp = oracle.evolve_tensordot(g, 18)
        assert np.abs(psi.reshape(-1) - exp).max() / np.abs(exp).max() < tol and info['n_passes'] < len(g) / 3
    for n, gates in ((18, rqc_1q2q(18, depth=12, seed=3)), (16, random_dense(16, 120, kmax=4, seed=4)),
                     (20, rqc_1q2q(20, depth=10, seed=5)), (16, rqc_1q2q(16, depth=8, seed=8) + random_dense(16, 12, kmax=6, seed=9) * 3)):
        exp = oracle.evolve_tensordot(gates, n)
        for opts in (True, {'tile_bits': 12, 'low_bits': 4, 'inner_max': 4}, {'tile_bits': 14, 'inner_max': 0}):
            psi, info = simulate(gates, initial_state='0' * n, complex_type='complex64', blocked=opts,
                                 return_info=True, qubits=list(range(n)))
            err = np.abs(psi.reshape(-1) - exp).max() / np.abs(exp).max()
            assert err < circuit_tol(gates), (n, opts, err)
            assert info['n_passes'] < len(gates) / 2.5
    # optimize='evolution-hip' = the cost model's own choice; with planning priced at nothing (a loop this short would not
    # win any planning time back and run gate by gate) it takes the fused / cache-blocked schedules
    from hybridq_amd import simulation
    monkeypatch.setattr(simulation, 'PLAN_HOST_MS_PER_GATE', dict.fromkeys(simulation.PLAN_HOST_MS_PER_GATE, 0.0))
    for n in (12, 18):
        g = rqc_1q2q(n, depth=10, seed=11) + random_dense(n, 6, kmax=5, seed=12)
        psi, info = simulate(g, initial_state='0' * n, optimize='evolution-hip', return_info=True, qubits=list(range(n)))
        exp = oracle.evolve_tensordot(g, n)
        assert np.abs(psi.reshape(-1) - exp).max() / np.abs(exp).max() < circuit_tol(g)
        assert info['n_passes'] < len(g) / 2
    # every gate is scheduled exactly once and dependencies are kept (pure planner check)
    n = 22
    gates = rqc_1q2q(n, depth=16, seed=6)
    ops = plan_blocked(gates, {q: n - 1 - q for q in range(n)}, n, inner_max=0)
    st = blocked_stats(ops)
    assert st['inner_gates'] + st['plain_gates'] == len(gates)
    for op in ops:
        if op[0] == 'B':
            tile = set(int(p) for p in op[1])
            assert len(tile) == 13 and {0, 1} <= tile
            assert all(set(pos) <= tile for _, pos in op[2])
    # a functional gate cuts the schedule
    seen = []
    fg = FunctionalGate((0,), lambda psi, order: (seen.append(1) or psi, order))
    g = rqc_1q2q(16, depth=6, seed=7)
    psi = simulate(g[:50] + [fg] + g[50:], initial_state='0' * 16, blocked=True, qubits=list(range(16)))
    exp = oracle.evolve_tensordot(g, 16)
    assert seen and np.abs(psi.reshape(-1) - exp).max() / np.abs(exp).max() < circuit_tol(g)


@pytest.mark.parametrize('graph', ['1', '0'])
def test_compiled_program_matches_oracle(torch_cuda, oracle_port, graph, monkeypatch):
    """hq_program_*: record a circuit once (per-gate, fused, blocked, swap + permute included),
    replay it several times (loop / hipGraph capture / graph replay) -- each replay equals one
    more application of the circuit by the oracle.  Non-recordable calls fail while recording
    and leave the library usable."""
    import oracle
    from hybridq_amd import core
    from hybridq_amd.circuits import random_dense, rqc_1q2q
    from hybridq_amd.simulation import EvolutionState
    torch = torch_cuda
    monkeypatch.setenv('HQ_PROGRAM_GRAPH', graph)
    core.set_stream(torch.cuda.current_stream().cuda_stream)
    n = 16
    gates = rqc_1q2q(n, depth=6, seed=41) + random_dense(n, 10, kmax=6, seed=42) + \
        [g for g in random_dense(n, 60, kmax=8, seed=43) if len(g[1]) >= 7][:2]  # incl. the LDS-table and GEMM kernels
    one = oracle.evolve_tensordot(gates, n)
    for ct, kw in (('complex64', dict(compress=0)), ('complex64', dict(compress=4)), ('complex64', dict(blocked=True)),
                   ('complex128', dict(compress=4)), ('complex128', dict(blocked=True))):
        tol = circuit_tol(gates, complex_type=ct)
        st = EvolutionState(list(range(n)), complex_type=ct, initial_state='0' * n)
        prog = st.compile(gates, **kw)
        assert len(prog) > 0
        core.sync()
        # recording must not have touched the state
        psi0 = st.to_complex().cpu().numpy()
        assert psi0[0] == 1 and np.count_nonzero(psi0) == 1
        prog.run()
        core.sync()
        psi = st.to_complex().cpu().numpy()
        assert np.abs(psi - one).max() / np.abs(one).max() < tol, (ct, kw)
        # replays (second run = captured graph): circuit applied 3 times in total
        prog.run()
        prog.run()
        core.sync()
        three = oracle.evolve_tensordot(gates * 3, n)
        psi = st.to_complex().cpu().numpy()
        assert np.abs(psi - three).max() / np.abs(three).max() < circuit_tol(gates * 3, complex_type=ct), (ct, kw)
        # the library still runs eagerly after a program was recorded, and the program is
        # unaffected by eager calls in between
        st.apply(np.array([[0, 1], [1, 0]]), (3,))
        st.apply(np.array([[0, 1], [1, 0]]), (3,))
        prog.free()
        with pytest.raises(core.HQError):
            prog.run()
    # swap and permute_bits record too (exact)
    rng = np.random.default_rng(5)
    a = torch.from_numpy(rng.standard_normal(1 << n).astype(np.float32)).cuda()
    b = torch.empty_like(a)
    ref = a.clone()
    perm = [int(x) for x in rng.permutation(n)]
    with core.Program() as prog:
        core.swap(a, [2, 0, 1], n)
        core.permute_bits(a, b, perm, n)
    core.sync()
    assert bool((a == ref).all())
    prog.run()
    core.sync()
    a2, b2 = ref.clone(), torch.empty_like(ref)
    core.swap(a2, [2, 0, 1], n)
    core.permute_bits(a2, b2, perm, n)
    core.sync()
    assert bool((a == a2).all()) and bool((b == b2).all()) and not bool((a == ref).all())
    # non-recordable calls: fail loudly, recording can be closed, nothing half-recorded runs
    re = torch.zeros(1 << n, dtype=torch.float32, device='cuda')
    im = torch.zeros_like(re)
    prog = core.Program()
    with prog:
        with pytest.raises(core.HQError):
            core.norm2(re, im)
        if os.environ.get('HQ_EMU_HOST_IS_DEVICE') != '1':  # (under the host emulation every pointer is device memory)
            with pytest.raises(core.HQError):
                core.apply_U(np.zeros(1 << 10, np.float32), np.zeros(1 << 10, np.float32), np.eye(2), [3], 10)
    assert len(prog) == 0
    with pytest.raises(core.HQError):
        with core.Program():
            with core.Program():  # nested recording is an error
                pass
    core.init_state(re, im, 'plus')
    assert abs(core.norm2(re, im) - 1.0) < 1e-12  # 2^-n/2 is exact in float32, the sum runs in double


@pytest.mark.parametrize('n', [16, 20, 22])
def test_simulation_large_like_reference(torch_cuda, oracle_port, n):
    """Mirror of the reference's test_simulation_4__simulation_large (tests.py:2335-2369):
    600 random NON-unitary 1-/2-qubit gates on random qubits, initial state = 3 basis characters
    + random '01+-', complex64 with compress=4 and complex128 with compress=8 (k up to 8 through
    the generic kernel), compared with an independent evolution.  The reference asserts 1e-3;
    here: 2e-6-level for c64 (600 roundings) and 1e-12 for c128 against the f64 oracle run of
    the reference's driver protocol."""
    import oracle
    from hybridq_amd.circuits import random_dense
    from hybridq_amd.simulation import simulate
    rng = np.random.default_rng(100 + n)
    init = ''.join(rng.choice(list('01'), size=3)) + ''.join(rng.choice(list('01+-'), size=n - 3))
    gates = random_dense(n, 600, kmax=2, seed=200 + n)
    exp, _ = oracle.evolve_reference_protocol(oracle_port, gates, n, initial_state=init, complex_type='complex128')
    scale = np.abs(exp).max()
    p64 = simulate(gates, initial_state=init, complex_type='complex64', compress=4, qubits=list(range(n)))
    p128, info = simulate(gates, initial_state=init, complex_type='complex128', compress=8, qubits=list(range(n)),
                          return_info=True)
    assert p64.dtype == np.complex64 and p128.dtype == np.complex128 and p64.shape == (2,) * n
    assert info['n_gates'] < 600 / 4
    assert np.abs(p128.reshape(-1) - exp).max() / scale < 1e-12
    assert np.abs(p64.reshape(-1) - exp).max() / scale < circuit_tol(gates)
    np.testing.assert_allclose(p64.reshape(-1), exp, rtol=1e-3, atol=1e-3 * scale)  # the reference's own bar


@pytest.mark.parametrize('ft', ['float32', 'float64'])
def test_apply_U_randomized_differential(torch_cuda, oracle_port, ft):
    """Randomized differential test: random (n, k, positions, kernel family) against the oracle.
    Every family either runs the call or falls back to the automatic choice, so each case must
    match whatever kernel ends up running (seeded: failures are reproducible from the message)."""
    ft = np.dtype(ft)
    rng = np.random.default_rng(20240928 if ft == np.dtype('float32') else 20240929)
    modes = ['auto', 'auto', 'auto', 'mfma', 'direct', 'tile', 'gemm', 'generic']
    seen = {}
    for case in range(260):
        n = int(rng.integers(6, 21))
        kcap = min(10, n - 2)
        # bias towards small k (the common case) but cover every k
        k = int(min(kcap, rng.choice([1, 1, 2, 2, 3, 3, 4, 4, 5, 6, 7, 8, 9, 10])))
        if n >= 19 and k >= 9:
            k = 8  # keep the oracle's share of the run time small
        pos = [int(p) for p in rng.permutation(n)[:k]]
        if rng.random() < 0.3:
            pos = sorted(pos)
        mode = str(rng.choice(modes))
        re, im = _rand_state(rng, n, ft)
        U = _rand_U(rng, k)
        orr, oi = _oracle_apply(oracle_port, re, im, U, pos)
        gr, gi, kern = _gpu_apply(torch_cuda, re, im, U, pos, mode=mode)
        seen[kern] = seen.get(kern, 0) + 1
        err = _relerr(gr, gi, orr, oi)
        assert err <= wide_tol(ft, k), (case, n, k, pos, mode, kern, err)
    # the sample exercised every kernel family
    assert {'mfma', 'direct', 'mfma_tile', 'gemm', 'generic'} <= set(seen), seen


@pytest.mark.parametrize('ft', ['float32', 'float64'])
def test_probabilities_stream_kernel(torch_cuda, ft):
    """hq_probabilities_* at n >= 16 (streaming kernel: per-thread register sums, measured bits
    in the component / thread / iteration / chunk fields of the index) vs numpy."""
    from hybridq_amd import core
    torch = torch_cuda
    core.set_stream(torch.cuda.current_stream().cuda_stream)
    rng = np.random.default_rng(9)
    ft = np.dtype(ft)
    for n in (16, 17, 20):
        re, im = _rand_state(rng, n, ft)
        p = re.astype(np.float64)**2 + im.astype(np.float64)**2
        dre, dim_ = torch.from_numpy(re).cuda(), torch.from_numpy(im).cuda()
        cases = [[0], [1], [0, 1], [1, 0], [2], [9], [10], [15], [n - 1], [5, 3], [12, 11, 14], [10, 11, 12, 13, 14, 15],
                 [0, 4, 12, n - 1], [n - 1, 1, 13, 6], [15, 0, 9, 10, 1], list(range(8)), [n - 1, n - 2, 0],
                 [int(x) for x in rng.permutation(n)[:7]], [int(x) for x in rng.permutation(n)[:10]]]
        for pos in cases:
            got = core.probabilities(dre, dim_, pos, n)
            x = np.arange(1 << n, dtype=np.int64)
            t = np.zeros(1 << n, dtype=np.int64)
            for j, q in enumerate(pos):
                t |= ((x >> q) & 1) << j
            exp = np.bincount(t, weights=p, minlength=1 << len(pos))
            assert np.abs(got - exp).max() / exp.max() < (1e-6 if ft == np.dtype('float32') else 1e-13), (n, pos)


def test_simulate_simplify_like_reference(torch_cuda, oracle_port):
    """simulate(simplify=True) (the reference's default, simulation.py:293-305): planted identity
    gates and inverse pairs disappear before fusion, the state is the one of the full circuit."""
    import oracle
    from hybridq_amd.circuits import haar_unitary, rqc_1q2q
    from hybridq_amd.simulation import simulate
    n = 14
    rng = np.random.default_rng(12)
    base = rqc_1q2q(n, depth=6, seed=13)
    planted = []
    for i, g in enumerate(base):
        planted.append(g)
        if i % 5 == 0:
            planted.append((np.eye(2), (int(rng.integers(n)),)))
        if i % 7 == 3:
            U = haar_unitary(4, rng)
            a, b = (int(x) for x in rng.permutation(n)[:2])
            other = next(q for q in range(n) if q not in (a, b))
            planted += [(U, (a, b)), (np.diag([1, 1j]), (other,)), (U.conj().T, (a, b))]
    exp = oracle.evolve_tensordot(planted, n, qubits=list(range(n)))
    psi, info = simulate(planted, initial_state='0' * n, compress=0, return_info=True)
    assert np.abs(psi.reshape(-1) - exp).max() / np.abs(exp).max() < circuit_tol(planted)
    assert info['n_gates'] < len(planted) - len(base) // 5  # identities and inverse pairs are gone
    psi2, info2 = simulate(planted, initial_state='0' * n, compress=0, simplify=False, remove_id_gates=False,
                           return_info=True)
    assert info2['n_gates'] == len(planted)
    assert np.abs(psi2.reshape(-1) - exp).max() / np.abs(exp).max() < circuit_tol(planted)
