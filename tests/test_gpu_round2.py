"""GPU tests of the round-2 additions: tiny states through simulate() (plane alignment), device
prepare_state for mixed '01+-' strings vs the reference's vectors, bit permutations with many
moved bits, restore_order on the HIP backend, Measure / Projection on more than 10 qubits,
QASM text -> simulate, stream switching, per-call parity directly against oracle/_ref."""
import numpy as np
import pytest

import golden_util as gu
from tolerances import BAR, C_STRUCTURED, circuit_tol

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('ct', ['complex64', 'complex128'])
def test_simulate_tiny_states(torch_cuda, ct):
    """n = 1..4 (a Bell pair is n = 2): the im plane of the single allocation must stay 32-byte
    aligned (alloc_planes put it at base + 2^n * itemsize)."""
    import oracle
    from hybridq_amd.circuits import random_dense
    from hybridq_amd.dm import simulate as dm_simulate
    from hybridq_amd.simulation import EvolutionState, simulate
    h = np.array([[1, 1], [1, -1]]) / np.sqrt(2)
    cx = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 0, 1], [0, 0, 1, 0]], dtype=complex)
    bell = simulate([(h, (0,)), (cx, (0, 1))], initial_state='00', complex_type=ct, compress=0).reshape(-1)
    assert np.abs(bell - np.array([1, 0, 0, 1]) / np.sqrt(2)).max() < BAR[np.dtype(ct)]
    for n in (1, 2, 3, 4):
        st = EvolutionState(list(range(n)), complex_type=ct, initial_state='0' * n)
        assert st.planes[0].data_ptr() % 32 == 0 and st.planes[1].data_ptr() % 32 == 0
        gates = random_dense(n, 12, kmax=min(n, 3), seed=n, unitary=True)
        exp = oracle.evolve_tensordot(gates, n, qubits=list(range(n)))
        for kw in (dict(compress=0), dict(compress=4)):
            psi = simulate(gates, initial_state='0' * n, complex_type=ct, qubits=list(range(n)), **kw).reshape(-1)
            assert np.abs(psi - exp).max() / np.abs(exp).max() < circuit_tol(gates, complex_type=ct), (n, kw)
    # a 1-qubit density matrix = 2-qubit state vector
    rho = dm_simulate([(h, (0,))], initial_state='0', complex_type=ct).reshape(2, 2)
    assert np.abs(rho - 0.5).max() < BAR[np.dtype(ct)]


def test_prepare_state_matches_reference_vectors(torch_cuda):
    """prepare_state strings recorded from the reference (e2e_api.npz: '0', '1', '+', '-' mixes):
    every amplitude, written by the device kernel."""
    from hybridq_amd.simulation import EvolutionState
    z = gu.load('e2e_api.npz')
    strings = [str(s) for s in z['ps_strings']]
    assert any(set(s) - set('01') and set(s) != {'+'} for s in strings)  # the mixed branch is covered
    for i, s in enumerate(strings):
        exp = z[f'ps_{i}']
        for ct in ('complex64', 'complex128'):
            st = EvolutionState(list(range(len(s))), complex_type=ct, initial_state=s)
            got = st.to_complex().cpu().numpy()
            assert np.abs(got - exp).max() <= (1e-7 if ct == 'complex64' else 1e-15) * max(1.0, np.abs(exp).max()), (s, ct)


def test_prepare_state_mixed_large(torch_cuda):
    """n = 28 mixed string: norm 1, zero outside the '0'/'1' pattern, signs from the '-' characters
    (sampled), and the same state reached by applying H / X gates to |0...0> with the core."""
    from hybridq_amd import core
    from hybridq_amd.simulation import EvolutionState
    n = 28
    rng = np.random.default_rng(7)
    s = ''.join(rng.choice(list('01+-'), size=n))
    st = EvolutionState(list(range(n)), complex_type='complex64', initial_state=s)
    assert abs(st.norm2() - 1.0) < 1e-6
    ref = EvolutionState(list(range(n)), complex_type='complex64', initial_state='0' * n)
    h = np.array([[1, 1], [1, -1]]) / np.sqrt(2)
    x = np.array([[0, 1], [1, 0]])
    for q, ch in enumerate(s):
        if ch in '1-':
            ref.apply(x, (q,))
        if ch in '+-':
            ref.apply(h, (q,))
    ov = core.vdot(ref.planes[0], ref.planes[1], st.planes[0], st.planes[1])
    assert abs(ov - 1.0) < 1e-5  # float32 H gates on 2^28 amplitudes


def test_permute_bits_many_moved_bits(torch_cuda):
    """hq_permute_bits with more than 16 moved bits (the cap made restore_order fail
    at the target scale): full random permutations of 20..24 index bits, 4- and 8-byte elements."""
    from hybridq_amd import core
    torch = torch_cuda
    rng = np.random.default_rng(3)
    for n, dt in ((20, torch.float32), (22, torch.float64), (24, torch.int32)):
        for trial in range(3):
            perm = rng.permutation(n) if trial else np.roll(np.arange(n), 7)  # trial 0: one long rotated field
            src = torch.arange(1 << n, device='cuda').to(dt)
            dst = torch.empty_like(src)
            core.permute_bits(src, dst, perm, n)
            core.sync()
            x = np.arange(1 << n, dtype=np.int64)
            y = np.zeros_like(x)
            for i in range(n):
                y |= ((x >> i) & 1) << int(perm[i])
            assert (dst.cpu().numpy().astype(np.int64) == y).all(), (n, list(perm))


def test_restore_order_hip_backend(torch_cuda):
    """ShardedEvolution.restore_order() on the HIP backend with m = 21 local qubits (world = 1: the
    permutation passes are the whole story) after a circuit that leaves a scrambled placement."""
    import oracle
    from hybridq_amd.circuits import rqc_1q2q
    from hybridq_amd.dist import ShardedEvolution
    n = 21
    gates = rqc_1q2q(n, depth=4, seed=2)
    sh = ShardedEvolution(n, complex_type='complex64', initial_state='0' * n)
    sh.simulate(gates)
    rng = np.random.default_rng(0)  # scramble the placement the way evictions would
    perm = rng.permutation(n)
    sh.run([('P', np.asarray(perm, dtype=np.uint32))], update_map=False)
    at = {p: q for q, p in sh.pos.items()}
    sh.pos = {at[int(perm[i])]: i for i in range(n)}  # dst bit i <- src bit perm[i]
    assert sum(sh.pos[q] != n - 1 - q for q in range(n)) > 16
    sh.restore_order()
    assert all(sh.pos[q] == n - 1 - q for q in range(n))
    raw = sh.backend.to_numpy(sh.planes)
    exp = oracle.evolve_tensordot(gates, n)
    assert np.abs(raw[0] + 1j * raw[1] - exp).max() / np.abs(exp).max() < circuit_tol(gates)


def test_measure_and_projection_wide(torch_cuda):
    """More than 10 measured / projected qubits (hq_probabilities bins at most 2^10
    outcomes): chunked measurement collapses onto ONE basis pattern with norm 1; a wide
    Projection equals the numpy slice."""
    import oracle
    from hybridq_amd.circuits import rqc_1q2q
    from hybridq_amd.functional import Measure, Projection
    from hybridq_amd.simulation import simulate
    n = 16
    gates = rqc_1q2q(n, depth=4, seed=9)
    exp = oracle.evolve_tensordot(gates, n).reshape((2,) * n)
    qs = tuple(int(q) for q in np.random.default_rng(1).permutation(n)[:13])
    m = Measure(qs, rng=np.random.default_rng(5))
    psi = simulate(gates + [m], initial_state='0' * n, complex_type='complex64', qubits=list(range(n)))
    bits = [(m.outcome >> (len(qs) - 1 - i)) & 1 for i in range(len(qs))]  # qubits[0] = most significant bit
    idx = [slice(None)] * n
    for q, b in zip(qs, bits):
        idx[q] = b
    kept = exp[tuple(idx)]
    assert np.linalg.norm(kept.ravel()) > 0
    got = psi[tuple(idx)]
    assert abs(np.linalg.norm(psi.ravel()) - 1) < 1e-5
    assert abs(np.linalg.norm(got.ravel()) - 1) < 1e-5  # nothing survives outside the pattern
    ref = kept / np.linalg.norm(kept.ravel())
    assert np.abs(got - ref).max() / np.abs(ref).max() < 10 * circuit_tol(gates)
    state = ''.join(str(b) for b in bits)
    for renorm in (True, False):
        psi = simulate(gates + [Projection(state, qs, renormalize=renorm)], initial_state='0' * n,
                       complex_type='complex64', qubits=list(range(n)))
        want = np.zeros_like(exp)
        want[tuple(idx)] = ref if renorm else kept
        assert np.abs(psi - want).max() / np.abs(want).max() < 10 * circuit_tol(gates), renorm


def test_qasm_text_to_gpu(torch_cuda):
    """BASELINE cfg1 from TEXT: the gate-per-line QASM of examples/circuit_simple.qasm rebuilt from
    the golden fixture's gate list -> hybridq_amd.qasm.from_qasm -> simulate, against the
    reference's recorded amplitudes."""
    from hybridq_amd.qasm import from_qasm
    from hybridq_amd.simulation import simulate
    z = gu.load('e2e_simple_qasm.npz')
    n = int(z['n_qubits'])
    lines = [str(n)]
    for nm, qs in zip(z['gate_names'], z['gate_qubits']):
        lines.append(' '.join([str(nm).lower()] + [str(int(q)) for q in qs if q >= 0]))
    gates = from_qasm('\n'.join(['# rebuilt from e2e_simple_qasm.npz'] + lines))
    assert len(gates) == len(z['gate_names'])
    psi = simulate(gates, initial_state='0' * n, complex_type='complex64').reshape(-1)
    stride = int(z['sample_stride'])
    scale = np.abs(z['psi_sample']).max()
    calls = [len(pos) for kind, pos, _ in gu.trace(z, 'trace_') if kind == 'U']
    assert np.abs(psi[::stride] - z['psi_sample']).max() / scale < circuit_tol(calls, calls, c=C_STRUCTURED)
    assert np.abs(psi[:8] - z['psi_head']).max() / scale < circuit_tol(calls, calls, c=C_STRUCTURED)


def test_stream_switch_is_ordered(torch_cuda):
    """Work issued under `with torch.cuda.stream(s)` after work on the default stream sees its
    results (hq_set_stream makes the new stream wait on the device) and vice versa."""
    import oracle
    from hybridq_amd.circuits import rqc_1q2q
    from hybridq_amd.simulation import EvolutionState
    torch = torch_cuda
    n = 22
    gates = rqc_1q2q(n, depth=6, seed=4)
    st = EvolutionState(list(range(n)), complex_type='complex64', initial_state='0' * n)
    side = torch.cuda.Stream()
    for i, (U, qs) in enumerate(gates):
        if (i // 7) % 2:
            with torch.cuda.stream(side):
                st.apply(U, qs)
        else:
            st.apply(U, qs)
    psi = st.to_complex()
    torch.cuda.synchronize()
    exp = oracle.evolve_tensordot(gates, n)
    assert np.abs(psi.cpu().numpy() - exp).max() / np.abs(exp).max() < circuit_tol(gates)


def test_apply_U_directly_against_reference_core(torch_cuda, oracle_ref):
    """Per-call parity of the HIP library against the reference's OWN compiled core (oracle/_ref;
    the other per-call tests use the C port, itself pinned to _ref on CPU): k = 1..6, positions
    >= 3 as the reference build requires (LOG2_PACK_SIZE = 3), both precisions."""
    from hybridq_amd import core
    from oracle.binding import aligned_empty
    torch = torch_cuda
    rng = np.random.default_rng(17)
    for ft, n in (('float32', 18), ('float64', 16)):
        ft = np.dtype(ft)
        for k in range(1, 7):
            for trial in range(3):
                pos = 3 + rng.permutation(n - 3)[:k]
                re = rng.standard_normal(1 << n).astype(ft)
                im = rng.standard_normal(1 << n).astype(ft)
                U = (rng.standard_normal((1 << k, 1 << k)) + 1j * rng.standard_normal((1 << k, 1 << k))) / np.sqrt(2.0 * (1 << k))
                pl = aligned_empty((2, 1 << n), ft)
                pl[0], pl[1] = re, im
                assert oracle_ref.apply_U(pl[0], pl[1], U, pos, n) == 0
                dre, dim_ = torch.from_numpy(re).cuda(), torch.from_numpy(im).cuda()
                core.apply_U(dre, dim_, U, pos, n)
                core.sync()
                scale = max(np.abs(pl[0]).max(), np.abs(pl[1]).max())
                err = max(np.abs(dre.cpu().numpy() - pl[0]).max(), np.abs(dim_.cpu().numpy() - pl[1]).max()) / scale
                assert err <= BAR[np.dtype('complex64' if ft == np.dtype('float32') else 'complex128')], (ft, k, list(pos), err)


@pytest.mark.parametrize('dt', ['float32', 'float64', 'int32', 'int64'])
def test_swap_large_s_two_pass(torch_cuda, dt):
    """swap_* beyond one LDS tile (s = 14..18 for 4-byte, 13..17 for 8-byte elements): two in-place tile
    passes (plan_two_pass_swap) instead of the gather + copy fallback; exact for random, rotated and
    nearly-sorted permutations, n = s and n > s (several chunks), and still exact past the two-pass range
    (s = 19/20: fallback)."""
    import oracle
    from hybridq_amd import core
    torch = torch_cuda
    dt = np.dtype(dt)
    rng = np.random.default_rng(23)
    first = 14 if dt.itemsize == 4 else 13
    for s in range(first, first + 7):
        n = s if s >= 18 else s + int(rng.integers(0, 3))
        perms = [rng.permutation(s), np.roll(np.arange(s), 5), np.concatenate([np.arange(s - 3), rng.permutation(3) + s - 3]),
                 np.concatenate([rng.permutation(4), np.arange(4, s)])[::1], np.arange(s)[::-1].copy()]
        for pos in perms:
            a = rng.integers(0, 2**31 - 1, 1 << n).astype(dt)
            exp = oracle.swap_numpy(a, pos)
            t = torch.from_numpy(a.copy()).cuda()
            core.swap(t, pos, n)
            core.sync()
            assert (t.cpu().numpy() == exp).all(), (dt, s, n, list(pos))


def test_functional_gates_with_dot_inplace_on_device(torch_cuda):
    """Mirror of the reference's test_simulation_2__fn (tests.py:2037-2110): n = 14, 400 random NON-unitary
    gates, about half of them wrapped as FunctionalGates whose body is ``dot(U, psi, axes_b=axes,
    b_as_complex_array=True, inplace=True)`` on the raw split-plane state; mixed '01+-' initial state.  Here
    the functional gates see a DEVICE view of the planes (on_device=True), so every inner dot() is one
    kernel in HBM; the same gates written against numpy (host round trip per gate) must agree, and both
    must equal the plain circuit and an independent complex128 evolution."""
    import oracle
    from hybridq_amd.circuits import random_dense
    from hybridq_amd.dot import dot
    from hybridq_amd.simulation import FunctionalGate, simulate
    n, depth = 14, 400
    rng = np.random.default_rng(77)
    gates = random_dense(n, depth, kmax=2, seed=78)
    calls = {'device': 0, 'host': 0}

    def as_fn(U, qubits, on_device):
        def f(psi, order):
            if psi.ndim - len(order) != 1:
                raise ValueError("'psi' is not consistent with order")
            axes = [next(i for i, y in enumerate(order) if y == x) for x in qubits]
            calls['device' if on_device else 'host'] += 1
            if on_device:
                assert psi.is_cuda
            else:
                assert isinstance(psi, np.ndarray)
            return dot(a=U, b=psi, axes_b=axes, b_as_complex_array=True, inplace=True), order
        return FunctionalGate(qubits, f, on_device=on_device)

    pick = rng.random(len(gates)) < 0.5
    circ_dev = [as_fn(U, qs, True) if p else (U, qs) for (U, qs), p in zip(gates, pick)]
    init = ''.join(rng.choice(list('01+-'), size=n))
    exp = oracle.evolve_tensordot(gates, n, initial_state=init, qubits=list(range(n)))
    scale = np.abs(exp).max()
    tol = circuit_tol(gates)
    psi = simulate(gates, initial_state=init, complex_type='complex64', qubits=list(range(n))).reshape(-1)
    psi_fn = simulate(circ_dev, initial_state=init, complex_type='complex64', qubits=list(range(n))).reshape(-1)
    assert calls['device'] == int(pick.sum()) and calls['host'] == 0
    assert np.abs(psi - exp).max() / scale < tol
    assert np.abs(psi_fn - exp).max() / scale < tol
    # complex128 and the host variant of the same functional gates (a shorter circuit: every host gate is a
    # D2H + H2D round trip of the state)
    short = gates[:60]
    circ_host = [as_fn(U, qs, False) if p else (U, qs) for (U, qs), p in zip(short, pick)]
    circ_dev2 = [as_fn(U, qs, True) if p else (U, qs) for (U, qs), p in zip(short, pick)]
    exp2 = oracle.evolve_tensordot(short, n, initial_state=init, qubits=list(range(n)))
    a = simulate(circ_host, initial_state=init, complex_type='complex128', qubits=list(range(n))).reshape(-1)
    b = simulate(circ_dev2, initial_state=init, complex_type='complex128', qubits=list(range(n))).reshape(-1)
    assert calls['host'] == int(pick[:60].sum())
    assert np.abs(a - exp2).max() / np.abs(exp2).max() < 1e-12
    assert np.abs(b - exp2).max() / np.abs(exp2).max() < 1e-12


def test_evolution_hip_chooses_a_schedule(torch_cuda, monkeypatch):
    """optimize='evolution-hip': the cost model picks between gate-by-gate / fused 4 / fused 5 / cache-blocked
    and records the choice; whatever it picks, the state is the circuit's."""
    import oracle
    from hybridq_amd.circuits import random_dense, rqc_1q2q
    from hybridq_amd.simulation import simulate
    for n, gates in ((12, rqc_1q2q(12, depth=10, seed=1)), (20, rqc_1q2q(20, depth=12, seed=2)),
                     (18, random_dense(18, 24, kmax=6, seed=3, unitary=True))):
        psi, info = simulate(gates, initial_state='0' * n, optimize='evolution-hip', return_info=True, qubits=list(range(n)))
        sch = info['schedule']
        assert sch['chosen'] in sch['modelled_ms'] and sch['modelled_ms'][sch['chosen']] == min(sch['modelled_ms'].values())
        assert ('blocked' in list(sch['modelled_ms']) + sch['not_planned']) == (n >= 14)
        exp = oracle.evolve_tensordot(gates, n, qubits=list(range(n)))
        assert np.abs(psi.reshape(-1) - exp).max() / np.abs(exp).max() < circuit_tol(gates), (n, sch)
    # an explicit setting overrides the model
    _, info = simulate(rqc_1q2q(16, depth=6, seed=4), initial_state='0' * 16, optimize='evolution-hip', compress=3, return_info=True)
    assert 'schedule' not in info
    # 'evolution' (the reference's "best engine" alias) chooses too; 'evolution-hybridq' keeps the reference schedule
    g20 = rqc_1q2q(20, depth=12, seed=2)
    psi_a, info_a = simulate(g20, initial_state='0' * 20, optimize='evolution', return_info=True, qubits=list(range(20)))
    psi_h, info_h = simulate(g20, initial_state='0' * 20, optimize='evolution-hybridq', return_info=True, qubits=list(range(20)))
    # a short loop: fusing to 4 pays for its (native, 0.006 ms per gate) planning, the cache-blocked plan does not
    assert info_a['schedule']['chosen'] == 'fused_4' and 'blocked' in info_a['schedule']['not_planned'] and 'schedule' not in info_h
    assert np.abs(psi_a - psi_h).max() / np.abs(psi_h).max() < 2 * circuit_tol(g20)
    # with planning priced at nothing the schedule predicted to be fastest is planned first: the cache-blocked one, as at n = 30
    from hybridq_amd import simulation
    monkeypatch.setattr(simulation, 'PLAN_HOST_MS_PER_GATE', dict.fromkeys(simulation.PLAN_HOST_MS_PER_GATE, 0.0))
    psi_b, info_b = simulate(g20, initial_state='0' * 20, optimize='evolution', return_info=True, qubits=list(range(20)))
    assert info_b['schedule']['chosen'] == 'blocked'
    assert np.abs(psi_b - psi_h).max() / np.abs(psi_h).max() < 2 * circuit_tol(g20)


@pytest.mark.parametrize('ct,n,tb', [('complex64', 25, 13), ('complex128', 24, 12), ('complex64', 23, 13)])
def test_apply_blocked_many_tiles_per_workgroup(torch_cuda, ct, n, tb):
    """The cache-blocked kernel with several tiles per (persistent) workgroup -- register prefetch of the next tile,
    incremental tile base, table-driven gates -- and with more gates than the LDS tables hold (the variant that
    computes its addresses and reads A operands from global memory): same state as the gates applied one by one
    on the device (the per-gate kernels are pinned to the oracle elsewhere)."""
    from hybridq_amd import core
    from hybridq_amd.circuits import haar_unitary
    torch = torch_cuda
    core.use_torch_stream()
    rng = np.random.default_rng(n)
    ft = torch.float32 if ct == 'complex64' else torch.float64
    low = 5 if ct == 'complex64' else 4
    for n_gates in (3, 7, 24):
        high = np.sort(rng.permutation(np.arange(low, n))[:tb - low])
        tile = np.concatenate([np.arange(low), high]).astype(np.uint32)
        gates = []
        for _ in range(n_gates):
            k = int(rng.integers(1, 5))
            gates.append((haar_unitary(1 << k, rng).astype(ct), [int(p) for p in rng.permutation(tile)[:k]]))
        a = torch.from_numpy(rng.standard_normal((2, 1 << n))).to(ft).cuda()
        a /= torch.linalg.norm(a)
        b = a.clone()
        core.apply_blocked(a[0], a[1], tile, gates, n_qubits=n)
        assert core.last_kernel() == 'blocked'
        for U, pos in gates:
            core.apply_U(b[0], b[1], U, pos, n)
        core.sync()
        err = float((a - b).abs().max() / b.abs().max())
        as_circuit = [(U, tuple(pos)) for U, pos in gates]
        assert err <= circuit_tol(as_circuit, as_circuit, complex_type=ct), (ct, n, n_gates, err)


@pytest.mark.parametrize('dt,n', [('float32', 26), ('float64', 25)])
def test_swap_many_tiles_per_workgroup(torch_cuda, dt, n):
    """swap_* with far more LDS tiles than workgroups (the persistent loop with the register prefetch of the next
    tile for 32 KiB float32 tiles): s = 8, 12, 13, 15 on 2^n elements, exact against a gather computed with torch
    index arithmetic on the device."""
    from hybridq_amd import core
    torch = torch_cuda
    rng = np.random.default_rng(n)
    tdt = getattr(torch, dt)
    a = torch.arange(1 << n, device='cuda', dtype=torch.int64)
    for s in (8, 12, 13, 15):
        pos = rng.permutation(s)
        x = a & ((1 << s) - 1)
        y = torch.zeros_like(x)
        for i in range(s):  # new[x] = old[(x & ~(2^s - 1)) | sum_i x_i << pos[i]]   (python_swap.cpp:68-99)
            y |= ((x >> i) & 1) << int(pos[i])
        src = (a & ~((1 << s) - 1)) | y
        data = (a % 1000003).to(tdt)  # exactly representable, all chunks different
        exp = data[src]
        core.swap(data, pos, n)
        core.sync()
        assert torch.equal(data, exp), (dt, n, s, list(pos))


@pytest.mark.parametrize('ct,n,ks', [('complex64', 25, (7, 8)), ('complex128', 24, (7, 8, 9))])
def test_gemm_kernel_many_tiles_per_workgroup(torch_cuda, ct, n, ks):
    """k >= 7 on more tiles than workgroups (the persistent loop of apply_gemm_kernel with the register prefetch
    of the next tile, incremental tile base): same result as the VALU kernel (`generic` mode: an independent
    implementation pinned to the oracle at small n), and U^dagger undoes U."""
    from hybridq_amd import core
    from hybridq_amd.circuits import haar_unitary
    torch = torch_cuda
    core.use_torch_stream()
    rng = np.random.default_rng(5 * n)
    ft = torch.float32 if ct == 'complex64' else torch.float64
    tol = 4 * BAR[np.dtype(ct)]
    for k in ks:
        pos = sorted(int(p) for p in rng.permutation(n)[:k])
        U = haar_unitary(1 << k, rng).astype(ct)
        a = torch.from_numpy(rng.standard_normal((2, 1 << n))).to(ft).cuda()
        a /= torch.linalg.norm(a)
        b, orig = a.clone(), a.clone()
        core.apply_U(a[0], a[1], U, pos, n)
        assert core.last_kernel() == 'gemm', core.last_kernel_desc()
        core.set_apply_mode('generic')
        try:
            core.apply_U(b[0], b[1], U, pos, n)
            assert core.last_kernel() != 'gemm'
        finally:
            core.set_apply_mode('auto')
        core.sync()
        scale = float(b.abs().max())
        assert float((a - b).abs().max()) / scale < tol, (ct, n, k, pos)
        core.apply_U(a[0], a[1], np.ascontiguousarray(U.conj().T), pos, n)
        core.sync()
        assert float((a - orig).abs().max()) / float(orig.abs().max()) < 2 * tol, (ct, n, k, pos)


def test_large_state_comes_back_through_the_chunked_copy(torch_cuda):
    """simulate(return_numpy_array=True) on a state above CHUNKED_RETURN_MIN_BYTES: the chunked, threaded device -> host
    copy returns exactly what the plain tensor.cpu() gives."""
    from hybridq_amd.circuits import rqc_1q2q
    from hybridq_amd.simulation import CHUNKED_RETURN_MIN_BYTES, _to_host, simulate
    torch = torch_cuda
    n = 26
    assert (8 << n) >= CHUNKED_RETURN_MIN_BYTES
    g = rqc_1q2q(n, depth=4, seed=3)
    psi = simulate(g, initial_state='+' * n, qubits=list(range(n)))
    st = simulate(g, initial_state='+' * n, qubits=list(range(n)), return_numpy_array=False)
    ref = st.to_complex().cpu().numpy()
    assert psi.shape == (2,) * n and psi.dtype == np.complex64
    assert np.array_equal(psi.reshape(-1), ref)
    x = torch.randn(1 << 25, dtype=torch.complex128, device='cuda')
    assert np.array_equal(_to_host(x), x.cpu().numpy())


def test_host_functional_gate_on_a_large_state(torch_cuda):
    """A reference-style (host numpy) FunctionalGate on a state large enough for the chunked, threaded D2H / H2D
    copies: same result as the equivalent matrix gate."""
    from hybridq_amd.circuits import haar_unitary, rqc_1q2q
    from hybridq_amd.simulation import FunctionalGate, simulate
    n = 26
    rng = np.random.default_rng(9)
    U = haar_unitary(2, rng).astype(np.complex64)
    q = 7

    def apply(psi, order):  # psi: (2,) + (2,)*n real array (re, im); gate on qubit q, in place
        ax = order.index(q) + 1
        re, im = np.moveaxis(psi[0], ax - 1, 0), np.moveaxis(psi[1], ax - 1, 0)
        r0, r1, i0, i1 = re[0].copy(), re[1].copy(), im[0].copy(), im[1].copy()
        for row, (a, b) in enumerate(U):
            re[row] = a.real * r0 - a.imag * i0 + b.real * r1 - b.imag * i1
            im[row] = a.real * i0 + a.imag * r0 + b.real * i1 + b.imag * r1
        return psi, order
    pre = rqc_1q2q(n, depth=3, seed=4)
    circ_f = pre + [FunctionalGate([q], apply)] + pre[:20]
    circ_m = pre + [(U, (q,))] + pre[:20]
    a = simulate(circ_f, initial_state='0' * n, qubits=list(range(n)), compress=0)
    b = simulate(circ_m, initial_state='0' * n, qubits=list(range(n)), compress=0)
    assert np.abs(a - b).max() / np.abs(b).max() < 2 * BAR[np.dtype('complex64')]


def test_large_array_initial_state(torch_cuda):
    """initial_state given as a 2^n array above the chunked-copy threshold (uploaded as complex, split on the device):
    the state comes back unchanged through an empty circuit's worth of identity, and a circuit on it equals the same
    circuit started from the equivalent string."""
    from hybridq_amd.circuits import rqc_1q2q
    from hybridq_amd.simulation import simulate
    n = 25
    rng = np.random.default_rng(6)
    psi0 = (rng.standard_normal(1 << n) + 1j * rng.standard_normal(1 << n)).astype(np.complex64)
    psi0 /= np.linalg.norm(psi0)
    ident = [(np.eye(2, dtype=np.complex64), (q,)) for q in range(n)]
    out = simulate(ident, initial_state=psi0, qubits=list(range(n)), compress=0, simplify=False, remove_id_gates=False)
    assert np.array_equal(out.reshape(-1), psi0)
    g = rqc_1q2q(n, depth=3, seed=8)
    plus = np.full(1 << n, 2.0 ** (-n / 2), dtype=np.complex64)
    a = simulate(g, initial_state=plus, qubits=list(range(n)))
    b = simulate(g, initial_state='+' * n, qubits=list(range(n)))
    assert np.abs(a - b).max() / np.abs(b).max() < 2 * BAR[np.dtype('complex64')]
