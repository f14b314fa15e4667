"""World-size 2 and 4 tests of the sharded evolution over gloo on CPU.

The planner, the exchange logic and the map bookkeeping are the product's
(hybridq_amd.dist); only the per-shard numerical backend is swapped for a host one built
on the CPU oracle, because there is no GPU here.  The result must equal the single-process
reference evolution in canonical qubit order."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


class CpuBackend:
    """Host stand-in for hybridq_amd.dist.HipBackend (same interface), TEST ONLY."""

    def __init__(self, float_type):
        import torch
        import torch.distributed as dist
        import oracle
        self.torch, self.dist = torch, dist
        self.lib = oracle.load_port()
        self.float_type = np.dtype(float_type)
        self.tdt = {np.dtype('float32'): torch.float32, np.dtype('float64'): torch.float64}[self.float_type]

    def empty_planes(self, m):
        return self.torch.zeros((2, 1 << m), dtype=self.tdt)

    def fill_zero(self, planes):
        planes.zero_()

    def fill_basis(self, planes, local_index):
        planes.zero_()
        planes[0, local_index] = 1

    def fill_const(self, planes, value):
        planes[0].fill_(value)
        planes[1].zero_()

    def fill_product(self, planes, chars_by_position, hi_bits):
        m = int(planes.shape[1]).bit_length() - 1
        x = np.arange(1 << m, dtype=np.int64) | int(hi_bits)
        v = np.ones(1 << m)
        for p, ch in chars_by_position.items():
            bit = (x >> p) & 1
            v *= {'0': 1 - bit, '1': bit, '+': np.full(1 << m, 2**-0.5), '-': (1 - 2 * bit) * 2**-0.5}[ch]
        planes[0].copy_(self.torch.from_numpy(v.astype(self.float_type)))
        planes[1].zero_()

    def apply(self, planes, U, pos, m):
        assert self.lib.apply_U(planes[0].numpy(), planes[1].numpy(), U, pos, m) == 0

    def apply_blocked(self, planes, tile_pos, gates, m):
        tile = set(int(p) for p in tile_pos)
        for U, pos in gates:  # same result as one LDS-tile pass: the gates one by one
            assert set(int(p) for p in pos) <= tile
            self.apply(planes, U, pos, m)

    def permute(self, src, dst, perm, m):
        x = np.arange(1 << m, dtype=np.int64)
        y = np.zeros_like(x)
        for i, p in enumerate(perm):
            y |= ((x >> i) & 1) << int(p)
        dst[0].copy_(src[0][self.torch.from_numpy(y)])
        dst[1].copy_(src[1][self.torch.from_numpy(y)])

    def all_to_all(self, dst, src, group):
        self.dist.all_to_all_single(dst[0], src[0], group=group)
        self.dist.all_to_all_single(dst[1], src[1], group=group)

    def sync(self):
        pass

    def to_numpy(self, planes):
        return planes.numpy()

    def norm2(self, planes):
        return float((planes.double()**2).sum())


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, seed, ct, out_dir):
    import torch
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        # the exchange self-test of the product (run on a fresh RCCL communicator there): its expectation must hold
        # for the reference two-step exchange (permute, then all_to_all_single) on every rank count
        from hybridq_amd.dist import exchange_selftest
        _be = CpuBackend(np.float64)

        def _ex(src, dst, perm, m):
            if perm is not None:
                _be.permute(src, dst, perm, m)
                src, dst = dst, src
            _be.all_to_all(dst, src, None)
            return perm is not None
        exchange_selftest(torch, torch.float64, 'cpu', world, rank, _ex)
        from hybridq_amd.circuits import random_dense, rqc_1q2q
        from hybridq_amd.dist import ShardedEvolution
        ft = np.float32 if ct == 'complex64' else np.float64
        gates = rqc_1q2q(n, depth=6, seed=seed) + random_dense(n, 30, kmax=4, seed=seed + 1)
        sh = ShardedEvolution(n, complex_type=ct, initial_state='0' * n, backend=CpuBackend(ft))
        sched = sh.plan(gates)
        sh.run(sched)
        psi = sh.state_numpy()
        n_x = sum(1 for op in sched if op[0] in ('X', 'XP'))
        n_p = sum(1 for op in sched if op[0] in ('P', 'XP'))
        # second circuit from the permuted placement + a mixed '01+-' initial state
        sh2 = ShardedEvolution(n, complex_type=ct, initial_state=('-+10+' * n)[:n], backend=CpuBackend(ft))
        g2 = random_dense(n, 25, kmax=3, seed=seed + 2)
        sh2.simulate(g2[:12])
        sh2.simulate(g2[12:])  # re-planned from a non-identity map
        psi2 = sh2.state_numpy()
        # fused schedule + restore to the canonical placement: the raw shards, concatenated in
        # rank order, must then BE the canonical state (no host-side un-permutation)
        sh3 = ShardedEvolution(n, complex_type=ct, initial_state='0' * n, backend=CpuBackend(ft))
        sh3.simulate(gates, compress=5 if n >= 11 else 4)  # width 5: what the k = 5 kernel makes worthwhile
        moved = any(sh3.pos[q] != n - 1 - q for q in range(n))
        sh3.restore_order()
        assert all(sh3.pos[q] == n - 1 - q for q in range(n))
        import torch
        loc = torch.from_numpy(np.ascontiguousarray(sh3.backend.to_numpy(sh3.planes)))
        parts = [torch.empty_like(loc) for _ in range(world)]
        dist.all_gather(parts, loc)
        raw = np.concatenate([p[0].numpy() + 1j * p[1].numpy() for p in parts])
        nrm = sh.norm2()
        nb = -1
        psi4 = psi
        if sh.m >= 14:  # cache-blocked local passes between the exchanges (blocking.py)
            sh4 = ShardedEvolution(n, complex_type=ct, initial_state='0' * n, backend=CpuBackend(ft))
            sched4 = sh4.plan(gates, blocked=True)
            nb = sum(1 for op in sched4 if op[0] == 'B')
            sh4.run(sched4)
            psi4 = sh4.state_numpy()
        # exchange / compute overlap (round 4): exchanges in rounds of one piece per peer, the local gates that touch none
        # of the moving bits applied to the pieces as they land -- same state, plain and with cache-blocked passes
        import hybridq_amd.dist as dist_mod
        dist_mod.OVERLAP_MIN_SUB_QUBITS = 3  # (the product wants >= 12-qubit pieces; the shards here are tiny)
        sh5 = ShardedEvolution(n, complex_type=ct, initial_state='0' * n, backend=CpuBackend(ft), overlap=True)
        sched5 = sh5.plan(gates)
        n_xo = sum(1 for op in sched5 if op[0] == 'XO')
        n_attached = sum(len(op[3]) for op in sched5 if op[0] == 'XO')
        sh5.run(sched5)
        psi5 = sh5.state_numpy()
        sh6 = ShardedEvolution(n, complex_type=ct, initial_state='0' * n, backend=CpuBackend(ft), overlap=True)
        sh6.simulate(gates, compress=4)
        sh6.restore_order()
        psi6 = sh6.state_numpy()
        if rank == 0:
            np.savez(os.path.join(out_dir, 'out.npz'), psi=psi, psi2=psi2, n_x=n_x, n_p=n_p, nrm=nrm, raw=raw,
                     moved=moved, psi4=psi4, nb=nb, psi5=psi5, psi6=psi6, n_xo=n_xo, n_attached=n_attached)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world,n,ct', [(2, 10, 'complex128'), (4, 11, 'complex128'), (2, 12, 'complex64'),
                                        (2, 15, 'complex64'), (8, 12, 'complex128')])  # 8 ranks = BASELINE configs 3 and 5
def test_sharded_matches_single_process(tmp_path, world, n, ct):
    import torch.multiprocessing as mp
    import oracle
    from hybridq_amd.circuits import random_dense, rqc_1q2q
    seed = 5
    mp.spawn(_worker, args=(world, _free_port(), n, seed, ct, str(tmp_path)), nprocs=world, join=True)
    out = np.load(os.path.join(str(tmp_path), 'out.npz'))
    gates = rqc_1q2q(n, depth=6, seed=seed) + random_dense(n, 30, kmax=4, seed=seed + 1)
    exp = oracle.evolve_tensordot(gates, n)
    tol = 1e-12 if ct == 'complex128' else 2e-6
    assert np.abs(out['psi'] - exp).max() / np.abs(exp).max() < tol
    assert int(out['n_x']) >= 1  # the circuit really needed exchanges
    assert abs(float(out['nrm']) - float(np.vdot(exp, exp).real)) < 1e-5 * float(np.vdot(exp, exp).real)
    assert np.abs(out['raw'] - exp).max() / np.abs(exp).max() < tol  # fused + restore_order
    assert np.abs(out['psi4'] - exp).max() / np.abs(exp).max() < tol  # blocked local passes
    assert (int(out['nb']) >= 1) == (n - int(np.log2(world)) >= 14)
    # overlapped exchanges: the same state (the attached gates act on pieces: the same arithmetic per amplitude)
    assert int(out['n_xo']) >= 1 and int(out['n_attached']) >= 1, (out['n_xo'], out['n_attached'])
    assert np.abs(out['psi5'] - exp).max() / np.abs(exp).max() < tol
    assert np.abs(out['psi5'] - out['psi']).max() <= (0 if ct == 'complex128' else 1e-7) + 1e-15
    assert np.abs(out['psi6'] - exp).max() / np.abs(exp).max() < tol
    g2 = random_dense(n, 25, kmax=3, seed=seed + 2)
    exp2 = oracle.evolve_tensordot(g2, n, initial_state=('-+10+' * n)[:n], qubits=list(range(n)))
    assert np.abs(out['psi2'] - exp2).max() / np.abs(exp2).max() < tol


def test_planner_properties():
    """Pure planner: every gate scheduled once, dependencies respected, all targets local,
    evictions never touch needed qubits, map bookkeeping consistent."""
    from hybridq_amd.dist import plan_schedule
    rng = np.random.default_rng(0)
    for n, g in ((12, 1), (12, 2), (14, 3), (9, 0)):
        gq = []
        for _ in range(200):
            k = int(rng.integers(1, 5))
            gq.append(tuple(int(q) for q in rng.permutation(n)[:k]))
        qubits = list(range(n))
        ops, final = plan_schedule(gq, qubits, g)
        m = n - g
        pos = {q: n - 1 - i for i, q in enumerate(qubits)}
        seen = []
        last = {q: -1 for q in qubits}
        for op in ops:
            if op[0] == 'G':
                gi, lp = op[1], op[2]
                qs = gq[gi]
                assert lp == [pos[q] for q in reversed(qs)] and all(p < m for p in lp)
                for q in qs:  # per-qubit program order
                    assert last[q] < gi
                    last[q] = gi
                seen.append(gi)
            elif op[0] == 'P':
                perm = op[1]
                assert sorted(perm) == list(range(m))
                at = {p: q for q, p in pos.items()}
                for i in range(m):  # dst bit i <- src bit perm[i]
                    pos[at[perm[i]]] = i
            else:
                at = {p: q for q, p in pos.items()}
                for i in range(g):
                    a, b = m - g + i, m + i
                    pos[at[a]], pos[at[b]] = b, a
        assert sorted(seen) == list(range(len(gq)))
        assert pos == final
        if g == 0:
            assert all(op[0] == 'G' for op in ops)


def test_plan_restore_properties():
    from hybridq_amd.dist import plan_restore
    rng = np.random.default_rng(1)
    for n, g in ((10, 0), (10, 1), (12, 2), (14, 3)):
        m = n - g
        qubits = list(range(n))
        for trial in range(50):
            perm = rng.permutation(n)
            pos = {q: int(perm[q]) for q in qubits}
            ops, final = plan_restore(pos, qubits, g)
            cur = dict(pos)
            for op in ops:
                at = {p: q for q, p in cur.items()}
                if op[0] == 'P':
                    assert sorted(op[1]) == list(range(m))
                    for i in range(m):
                        cur[at[op[1][i]]] = i
                else:
                    for i in range(g):
                        a, b = m - g + i, m + i
                        cur[at[a]], cur[at[b]] = b, a
            assert all(cur[q] == n - 1 - q for q in qubits) and final == cur
            assert sum(op[0] == 'X' for op in ops) <= 3


def _dm_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        sys.path.insert(0, os.path.join(ROOT, 'tests'))
        import golden_util as gu
        from hybridq_amd.dist import ShardedEvolution
        from hybridq_amd.dm import Kraus, to_statevector_circuit
        z = gu.load('e2e_dm_circuit.npz')
        circuit = []
        for i, kind in enumerate(bytes(z['kinds']).decode()):
            qs = tuple(int(q) for q in z[f'q{i}'])
            circuit.append(Kraus(list(z[f'L{i}']), qs, s=z[f's{i}']) if kind == 'K' else (z[f'U{i}'], qs))
        n = int(z['n_qubits'])
        sv = to_statevector_circuit(circuit)
        labels = [(0, q) for q in range(n)] + [(1, q) for q in range(n)]
        sh = ShardedEvolution(2 * n, complex_type='complex128', initial_state='0' * (2 * n), qubits=labels,
                              backend=CpuBackend(np.float64))
        sh.simulate(sv, compress=4)
        rho = sh.state_numpy()
        if rank == 0:
            np.save(os.path.join(out_dir, 'rho.npy'), rho)
    finally:
        dist.destroy_process_group()


def test_sharded_density_matrix_config5_shape(tmp_path):
    """BASELINE config 5 in miniature: a noisy 6-qubit circuit = 12-qubit state vector with
    tuple qubit labels (0,q)/(1,q) and non-unitary fused gates, sharded over 4 ranks."""
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import golden_util as gu
    mp.spawn(_dm_worker, args=(4, _free_port(), str(tmp_path)), nprocs=4, join=True)
    rho = np.load(os.path.join(str(tmp_path), 'rho.npy'))
    exp = gu.load('e2e_dm_circuit.npz')['rho']
    assert np.abs(rho - exp).max() / np.abs(exp).max() < 5e-6


def _api_worker(rank, world, port, out_dir):
    """simulate(..., devices=N) / dm.simulate(..., devices=N): the reference-shaped calls, on gloo with the host backend."""
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import hybridq_amd.dist as hdist
        import hybridq_amd.simulation as sim
        from hybridq_amd import dm
        from hybridq_amd.circuits import random_dense, rqc_1q2q
        hdist.HipBackend = lambda float_type, placement=None: CpuBackend(float_type)  # the only stand-in: per-shard numerics
        sim._torch = lambda: None
        n = 12
        gates = rqc_1q2q(n, depth=8, seed=3) + random_dense(n, 20, kmax=3, seed=4)
        init = ('+-01' * n)[:n]
        results = {}
        for name, kw in (('auto', {}), ('hybridq', dict(optimize='evolution-hybridq')), ('c0', dict(compress=0)),
                         ('bits', dict(shard_bits=int(np.log2(world))))):
            kw = dict(kw)
            if 'shard_bits' not in kw:
                kw['devices'] = world
            psi, info = sim.simulate(gates, initial_state=init, complex_type='complex128', qubits=list(range(n)), return_info=True, **kw)
            assert info['n_ranks'] == world and info['n_qubits'] == n and info['n_exchanges'] >= 1
            assert ('schedule' in info) == (name in ('auto', 'bits'))
            results[name] = psi.reshape(-1)
        sh = sim.simulate(gates, initial_state=init, complex_type='complex128', qubits=list(range(n)), devices=world,
                          return_numpy_array=False)
        results['sharded_object'] = sh.state_numpy().reshape(-1)
        for bad in (dict(devices=2 * world), dict(devices=world, shard_bits=5)):
            try:
                sim.simulate(gates, initial_state=init, qubits=list(range(n)), **bad)
                raise AssertionError('accepted ' + repr(bad))
            except (RuntimeError, ValueError):
                pass
        try:
            sim.simulate(gates, initial_state=np.zeros((2,) * n), qubits=list(range(n)), devices=world)
            raise AssertionError('accepted an array initial state')
        except NotImplementedError:
            pass
        noisy = [(U, qs) for U, qs in random_dense(6, 12, kmax=2, seed=8, unitary=True)]
        noisy.insert(5, dm.depolarizing((1, 4), 0.1))
        noisy.append(dm.depolarizing((0,), 0.05))
        results['rho'] = dm.simulate(noisy, initial_state='0+1-00', complex_type='complex128', devices=world).reshape(-1)
        if rank == 0:
            np.savez(os.path.join(out_dir, 'api.npz'), **results)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 4])
def test_simulate_devices_api_over_gloo(tmp_path, world):
    """The reference-shaped sharded calls end to end on CPU: simulate(devices=N / shard_bits=g) under every schedule keyword,
    the returned ShardedEvolution, the argument errors, dm.simulate(devices=N) -- against single-process evolutions."""
    import torch.multiprocessing as mp
    import oracle
    from hybridq_amd import dm
    from hybridq_amd.circuits import random_dense, rqc_1q2q
    mp.spawn(_api_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    out = np.load(os.path.join(str(tmp_path), 'api.npz'))
    n = 12
    gates = rqc_1q2q(n, depth=8, seed=3) + random_dense(n, 20, kmax=3, seed=4)
    exp = oracle.evolve_tensordot(gates, n, initial_state=('+-01' * n)[:n], qubits=list(range(n)))
    for name in ('auto', 'hybridq', 'c0', 'bits', 'sharded_object'):
        assert np.abs(out[name] - exp).max() / np.abs(exp).max() < 1e-12, name
    noisy = [(U, qs) for U, qs in random_dense(6, 12, kmax=2, seed=8, unitary=True)]
    noisy.insert(5, dm.depolarizing((1, 4), 0.1))
    noisy.append(dm.depolarizing((0,), 0.05))
    sv = dm.to_statevector_circuit(noisy)
    qubits = [(0, q) for q in range(6)] + [(1, q) for q in range(6)]
    rho = oracle.evolve_tensordot(sv, 12, initial_state='0+1-00' * 2, qubits=qubits)
    assert np.abs(out['rho'] - rho).max() / np.abs(rho).max() < 1e-12
    assert abs(np.trace(out['rho'].reshape(64, 64)) - 1) < 1e-12
