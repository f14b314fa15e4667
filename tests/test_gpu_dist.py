"""Two and four ranks sharing ONE GPU: the sharded evolution with the real HIP backend and the
library's own exchange (hq_exchange_*, include/hq_hip.h).  RCCL refuses two ranks on one device,
so the transport here is the peer-to-peer one (the other ranks' planes mapped through HIP IPC, one
pack pass storing straight into them); the RCCL transport shares the pack / layout / self-chunk
code and is checked as far as one GPU allows by test_rccl_transport_plumbing.  The round-1 path
(permute_bits + all_to_all staged through the host over gloo) runs as the cross-check."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


EMU = os.environ.get('HQ_EMU_GPU_SUITE') == '1'  # host emulation: no HIP IPC, the exchange falls back to torch.distributed
# 'p2p' on the device and under the emulation alike (HIP IPC is emulated over POSIX shared memory); the emulated run can also
# force the RCCL transport (HQ_SHARD_TRANSPORT=rccl: tests/emu/rccl_emu.cpp stands in for librccl between the processes)
WANT_TRANSPORT = os.environ.get('HQ_SHARD_TRANSPORT', 'p2p') if EMU else 'p2p'
if EMU and os.environ.get('HQ_EMU_ASAN') == '1' and WANT_TRANSPORT == 'p2p':
    WANT_TRANSPORT = 'torch'  # the sanitizer build keeps device memory on the (checked) heap: nothing to export through IPC


def _worker(rank, world, port, n, ct, out_dir):
    import emu_boot
    emu_boot.maybe_install()
    import torch
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from hybridq_amd.circuits import random_dense, rqc_1q2q
        from hybridq_amd.dist import HipBackend, ShardedEvolution

        class HostStagedExchange(HipBackend):
            def all_to_all(self, dst, src, group):
                self.sync()
                for p in (0, 1):
                    h_src = src[p].cpu()
                    h_dst = torch.empty_like(h_src)
                    dist.all_to_all_single(h_dst, h_src, group=group)
                    dst[p].copy_(h_dst)

        ft = np.float32 if ct == 'complex64' else np.float64
        gates = rqc_1q2q(n, depth=8, seed=11) + random_dense(n, 40, kmax=5, seed=12)
        # (a) the library's own exchange: hq_exchange_* with peer-to-peer stores into the other ranks'
        #     planes (HIP IPC), eviction permutation folded into the pack pass -- the transport 'auto'
        #     selects for ranks that share a GPU (gloo group)
        sh = ShardedEvolution(n, complex_type=ct, initial_state='0' * n)
        transport = sh.backend.transport + (' | ' + sh.backend.transport_note if sh.backend.transport_note else '')
        sched = sh.plan(gates)
        sh.run(sched)
        psi = sh.state_numpy()
        sh.restore_order()  # exchanges + permutation passes back to the canonical placement
        assert all(sh.pos[q] == n - 1 - q for q in range(n))
        loc = torch.from_numpy(np.ascontiguousarray(sh.backend.to_numpy(sh.planes)))
        parts = [torch.empty_like(loc) for _ in range(world)]
        dist.all_gather(parts, loc)
        raw = np.concatenate([p[0].numpy() + 1j * p[1].numpy() for p in parts])
        # (b) the round-1 path as the cross-check: permute_bits + all_to_all staged through the host
        shh = ShardedEvolution(n, complex_type=ct, initial_state='0' * n, backend=HostStagedExchange(ft, transport='torch'))
        shh.run(shh.plan(gates))
        psi_h = shh.state_numpy()
        # same circuit, cache-blocked local passes between the exchanges, then canonical order
        shb = ShardedEvolution(n + 2, complex_type=ct, initial_state='0' * (n + 2))
        gb = rqc_1q2q(n + 2, depth=8, seed=13)
        schedb = shb.plan(gb, blocked=True)
        shb.run(schedb)
        psib = shb.state_numpy()
        nb = sum(1 for op in schedb if op[0] == 'B')
        # exchange / compute overlap: exchanges in rounds behind the C ABI (hq_exchange_rounds_*: folded pack, the
        # library's transports, one completion event per round), the attached gates applied to the pieces as they land
        # (RCCL transport: 2^sub_bits rounds on the communication stream; peer-to-peer stores between ranks that share a
        # GPU: one round)
        import hybridq_amd.dist as dist_mod
        dist_mod.OVERLAP_MIN_SUB_QUBITS = 8
        sho = ShardedEvolution(n, complex_type=ct, initial_state='0' * n, overlap='force')  # (one-round transports: the executor path all the same)
        scho = sho.plan(gates)
        n_xo = sum(1 for op in scho if op[0] == 'XO')
        sho.run(scho)
        psi_o = sho.state_numpy()
        if rank == 0:
            np.savez(os.path.join(out_dir, 'out.npz'), psi=psi, psib=psib, nb=nb, raw=raw, psi_h=psi_h, transport=transport,
                     n_xo=n_xo, psi_o=psi_o,
                     n_x=sum(1 for op in sched if op[0] in ('X', 'XP')), n_p=sum(1 for op in sched if op[0] in ('P', 'XP')))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world,n,ct', [(2, 15, 'complex64'), (4, 16, 'complex128')])
def test_sharded_hip_backend_two_ranks_one_gpu(torch_cuda, tmp_path, world, n, ct):
    import torch.multiprocessing as mp
    import oracle
    from hybridq_amd.circuits import random_dense, rqc_1q2q
    mp.spawn(_worker, args=(world, _free_port(), n, ct, str(tmp_path)), nprocs=world, join=True)
    out = np.load(os.path.join(str(tmp_path), 'out.npz'))
    gates = rqc_1q2q(n, depth=8, seed=11) + random_dense(n, 40, kmax=5, seed=12)
    exp = oracle.evolve_tensordot(gates, n)
    tol = 1e-6 if ct == 'complex64' else 1e-12
    assert str(out['transport']).split(' | ')[0] == WANT_TRANSPORT, str(out['transport'])  # the C-ABI exchange really ran
    assert np.abs(out['psi'] - exp).max() / np.abs(exp).max() < tol
    assert np.abs(out['raw'] - exp).max() / np.abs(exp).max() < tol  # restore_order: raw shards ARE the canonical state
    assert np.array_equal(out['psi'], out['psi_h'])  # same kernels, different transport: bit-identical
    assert int(out['n_x']) >= 1 and int(out['n_p']) >= 1  # exchanges, some with a folded permutation
    expb = oracle.evolve_tensordot(rqc_1q2q(n + 2, depth=8, seed=13), n + 2)
    assert np.abs(out['psib'] - expb).max() / np.abs(expb).max() < 5 * tol
    assert int(out['nb']) >= 1  # blocked passes were really used
    # overlapped exchanges (the pieces may dispatch to other kernels than the whole shard: equal to rounding)
    assert int(out['n_xo']) >= 1
    assert np.abs(out['psi_o'] - exp).max() / np.abs(exp).max() < tol, np.abs(out['psi_o'] - exp).max() / np.abs(exp).max()
    assert np.abs(out['psi_o'] - out['psi']).max() / np.abs(exp).max() < tol


def _rounds_worker(rank, world, port, out_dir):
    import emu_boot
    emu_boot.maybe_install()
    import torch
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from hybridq_amd import core
        from hybridq_amd.dist import HipBackend
        g = int(np.log2(world))
        report = {}
        for ft, tdt, m in ((np.float32, torch.float32, 13), (np.float64, torch.float64, 12)):
            be = HipBackend(ft, placement='plain')
            bufs = [be.empty_planes(m), be.empty_planes(m)]
            be.setup_exchange(None, bufs)
            rng = np.random.default_rng(100 * rank + m)
            data = torch.from_numpy(rng.standard_normal((2, 1 << m))).to(tdt).cuda()
            prng = np.random.default_rng(m)  # the same permutations on every rank
            perms = [None, prng.permutation(m), np.concatenate([np.arange(4), 4 + prng.permutation(m - 4)])]
            for pi, perm in enumerate(perms):
                for sub_bits in (1, 2):
                    bufs[0].copy_(data)
                    bufs[1].zero_()
                    where = be.exchange(bufs[0], bufs[1], perm, m, None)
                    plain = (bufs[0] if where else bufs[1]).clone()
                    bufs[0].copy_(data)
                    bufs[1].zero_()
                    where_r, n_rounds = be.exchange_rounds(bufs[0], bufs[1], perm, m, sub_bits, None)
                    res = bufs[0] if where_r else bufs[1]
                    G, S = world, 1 << sub_bits
                    seen = torch.zeros_like(res)
                    for r in range(n_rounds):  # copy out the pieces of round r as soon as that round has landed
                        be.exchange_round_wait(r)
                        for s_ in range(r * S // n_rounds, (r + 1) * S // n_rounds):
                            for pl in (0, 1):
                                seen[pl].view(G, S, -1)[:, s_].copy_(res[pl].view(G, S, -1)[:, s_])
                    be.sync()
                    key = f'{ft.__name__} perm{pi} sub_bits={sub_bits}'
                    report[key] = dict(rounds=n_rounds, where=(where, where_r), equal=bool(torch.equal(seen, plain)), transport=be.transport)
            del bufs
        everyone = [None] * world
        dist.all_gather_object(everyone, report)
        if rank == 0:
            import json
            with open(os.path.join(out_dir, 'rounds.json'), 'w') as f:
                json.dump(everyone, f)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 4, 8])
def test_exchange_rounds_equal_the_plain_exchange(torch_cuda, tmp_path, world):
    """hq_exchange_rounds_* against hq_exchange_* between 2 / 4 / 8 processes on the library's own transport, without and
    with a folded eviction permutation, 2 and 4 rounds: every piece, copied out right after ITS round's completion event,
    is bit-identical to the plain exchange's result; same result planes.  (Ranks sharing one GPU: peer-to-peer stores,
    one round; under the host emulation also the RCCL transport with real rounds on the communication stream.)"""
    import json
    import torch.multiprocessing as mp
    mp.spawn(_rounds_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    everyone = json.load(open(os.path.join(str(tmp_path), 'rounds.json')))
    assert len(everyone) == world
    for rank, report in enumerate(everyone):
        assert len(report) == 12
        for key, r in report.items():
            assert r['transport'] == WANT_TRANSPORT, (rank, key, r)
            assert r['equal'] and r['where'][0] == r['where'][1], (rank, key, r)
            assert r['rounds'] == (1 << int(key[-1]) if WANT_TRANSPORT == 'rccl' else 1), (rank, key, r)


def _per_gpu_worker(rank, world, port, n, n_dm, out_dir):
    """One rank PER GPU on the `nccl` (= RCCL) process group: what `bench.py --gpus N` and `simulate(devices=N)` are on a
    multi-GPU node.  Under the host emulation the ranks are processes on the emulated device, the process group is gloo
    and HQ_SHARD_TRANSPORT=rccl selects the same library transport (tests/emu/rccl_emu.cpp between the processes)."""
    import emu_boot
    emu = emu_boot.maybe_install()
    import torch
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dev = 0 if emu else rank
    torch.cuda.set_device(dev)
    if emu:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    else:
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', dev))
    try:
        from bench import dm_workload
        from hybridq_amd import core
        from hybridq_amd.circuits import rqc_1q2q
        from hybridq_amd.dist import ShardedEvolution
        import hybridq_amd.dist as dist_mod
        out = {}
        cfg3 = rqc_1q2q(n, depth=8, seed=33)                  # BASELINE configs[2]: random circuit, high-qubit shard
        cfg5 = dm_workload(n_dm // 2, 4)                      # BASELINE configs[4]: noisy circuit as a 2 nq-qubit state vector
        sh = ShardedEvolution(n, complex_type='complex64', initial_state='0' * n)
        info = core.shard_info()
        sched = sh.plan(cfg3)
        sh.run(sched)
        out['cfg3'] = sh.state_numpy()
        out['cfg3_exchanges'] = sum(1 for op in sched if op[0] in ('X', 'XP'))
        out['cfg3_folded'] = sum(1 for op in sched if op[0] == 'XP')
        del sh
        if emu:
            dist_mod.OVERLAP_MIN_SUB_QUBITS = 8  # (the toy shards of the emulated run)
        sho = ShardedEvolution(n, complex_type='complex64', initial_state='0' * n, overlap=True)  # exchanges in rounds, gates on the pieces
        scho = sho.plan(cfg3)
        sho.run(scho)
        out['cfg3_overlap'] = sho.state_numpy()
        out['cfg3_rounds_exchanges'] = sum(1 for op in scho if op[0] == 'XO')
        del sho
        shd = ShardedEvolution(n_dm, complex_type='complex64', initial_state='0' * n_dm)
        shd.run(shd.plan(cfg5))
        out['cfg5'] = shd.state_numpy()
        del shd
        infos = [None] * world
        dist.all_gather_object(infos, dict(info, device=torch.cuda.current_device(), backend=dist.get_backend()))
        if rank == 0:
            import json
            np.savez(os.path.join(out_dir, 'per_gpu.npz'), **out)
            with open(os.path.join(out_dir, 'per_gpu.json'), 'w') as f:
                json.dump(infos, f)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 4, 8])
def test_sharded_one_rank_per_gpu_over_rccl(torch_cuda, tmp_path, world, capsys):
    """Lights up by itself on a box with >= `world` GPUs (skipped on a one-GPU box; runs against tests/emu/rccl_emu.cpp in the
    CPU suite): one process per GPU on the nccl backend, so HipBackend takes the RCCL transport -- a communicator of `world`
    ranks made by the LIBRARY (hq_shard_init_rccl), grouped ncclSend / ncclRecv over xGMI -- and runs the BASELINE config-3
    generator at n = 22 + g (plain exchanges, exchanges with the folded eviction permutation, exchanges in rounds with the
    local gates applied to the pieces as they land) and the config-5 noisy-dm generator, each against the reference core
    driven by the reference protocol on ONE process.  No reference counterpart (simulation.py:379-380: no multi-process
    path); oracle: /root/reference/include/U.h:28-202 via oracle.evolve_reference_protocol."""
    torch = torch_cuda
    if EMU:
        if WANT_TRANSPORT != 'rccl':
            pytest.skip('one rank per GPU needs the RCCL transport (emulated: HQ_SHARD_TRANSPORT=rccl)')
    elif torch.cuda.device_count() < world:
        pytest.skip(f'needs {world} GPUs, this box has {torch.cuda.device_count()}')
    import json
    import torch.multiprocessing as mp
    import oracle
    from bench import dm_workload
    from hybridq_amd.circuits import rqc_1q2q
    from tolerances import circuit_tol
    g = int(np.log2(world))
    n = (14 if EMU else 22) + g  # (every rank gathers the full state for the comparison: 2^25 amplitudes at world 8)
    n_dm = 2 * ((n + 1) // 2)
    mp.spawn(_per_gpu_worker, args=(world, _free_port(), n, n_dm, str(tmp_path)), nprocs=world, join=True)
    out = np.load(os.path.join(str(tmp_path), 'per_gpu.npz'))
    infos = json.load(open(os.path.join(str(tmp_path), 'per_gpu.json')))
    # the exchange really ran over RCCL between `world` processes, one per device
    assert len(infos) == world
    for r, info in enumerate(infos):
        assert info['transport'] == 'rccl' and info['world'] == world and info['rank'] == r and info['rccl_ranks_seen'] == world, info
        assert EMU or (info['device'] == r and info['backend'] == 'nccl'), info
    lib = oracle.load_ref() if oracle.have_ref() else oracle.load_port()
    cfg3 = rqc_1q2q(n, depth=8, seed=33)
    exp3, _ = oracle.evolve_reference_protocol(lib, cfg3, n, complex_type='complex64', qubits=list(range(n)))
    tol3 = circuit_tol(cfg3, cfg3, complex_type='complex64')
    scale = np.abs(exp3).max()
    err = np.abs(out['cfg3'].reshape(-1) - exp3).max() / scale
    err_o = np.abs(out['cfg3_overlap'].reshape(-1) - exp3).max() / scale
    assert err <= tol3 and err_o <= tol3, (err, err_o, tol3)
    assert int(out['cfg3_exchanges']) >= 1 and int(out['cfg3_folded']) >= 1 and int(out['cfg3_rounds_exchanges']) >= 1
    cfg5 = dm_workload(n_dm // 2, 4)
    exp5, _ = oracle.evolve_reference_protocol(lib, cfg5, n_dm, complex_type='complex64', qubits=list(range(n_dm)))
    tol5 = circuit_tol(cfg5, cfg5, complex_type='complex64')
    err5 = np.abs(out['cfg5'].reshape(-1) - exp5).max() / np.abs(exp5).max()
    assert err5 <= tol5, (err5, tol5)
    with capsys.disabled():
        print(f'\n  {world} ranks, one per {"emulated process" if EMU else "GPU"}, RCCL transport (ncclCommCount = {infos[0]["rccl_ranks_seen"]}): config 3 n={n} '
              f'err {err:.2e} (in rounds with overlapped gates {err_o:.2e}; allowed {tol3:.2e}; literal bar met: {bool(max(err, err_o) <= 1e-6)}), '
              f'config 5 n={n_dm} err {err5:.2e} (allowed {tol5:.2e}); {int(out["cfg3_exchanges"])} exchanges, {int(out["cfg3_folded"])} with a folded permutation')


def test_exchange_one_rank_is_the_permutation(torch_cuda):
    """hq_exchange_* with one rank: no permutation = nothing to do (result stays in src), with a
    permutation = exactly hq_permute_bits on both planes (the pack pass alone)."""
    from hybridq_amd import core
    torch = torch_cuda
    rng = np.random.default_rng(4)
    core.shard_free()
    for dt, m in ((torch.float32, 20), (torch.float64, 18)):
        src = torch.from_numpy(rng.standard_normal((2, 1 << m))).to(dt).cuda()
        dst = torch.zeros_like(src)
        assert core.exchange(src[0], src[1], dst[0], dst[1], None, m) is True
        for perm in (rng.permutation(m), np.roll(np.arange(m), 3), np.concatenate([[1, 0], np.arange(2, m)])):
            where = core.exchange(src[0], src[1], dst[0], dst[1], perm, m)
            assert where is False
            ref = torch.empty_like(src)
            core.permute_bits(src[0], ref[0], perm, m)
            core.permute_bits(src[1], ref[1], perm, m)
            core.sync()
            assert torch.equal(dst, ref), list(perm)


def test_rccl_transport_plumbing(torch_cuda):
    """What of the RCCL transport can run on one GPU: librccl is found and bound at run time
    (dlopen), a communicator is created from a unique id, and a grouped ncclSend/ncclRecv (this
    rank as its own peer) moves a plane on the communication stream, ordered against the library
    stream by the same events hq_exchange_* uses."""
    from hybridq_amd import core
    torch = torch_cuda
    core.use_torch_stream()
    uid = core.shard_unique_id()
    assert len(uid) == 128 and any(uid)
    core.shard_init_rccl(1, 0, uid)
    try:
        a = torch.arange(1 << 22, dtype=torch.float32, device='cuda')
        b = torch.zeros_like(a)
        a.mul_(2.0)  # pending work on the library stream that the transfer must wait for
        core.shard_rccl_selftest(a, b)
        b.add_(1.0)  # and work after it that must wait for the transfer
        core.sync()
        exp = torch.arange(1 << 22, dtype=torch.float32, device='cuda') * 2.0 + 1.0
        assert torch.equal(b, exp)
        # the self-test HipBackend runs after creating its communicator on a real multi-GPU job (world = 1 here:
        # checks its expectation for the permuted case and that it runs through hq_exchange_*)
        from hybridq_amd.dist import HipBackend
        HipBackend(np.float32)._rccl_selftest(1, 0)
        HipBackend(np.float64)._rccl_selftest(1, 0)
    finally:
        core.shard_free()


def _api_worker(rank, world, port, out_dir):
    import emu_boot
    emu_boot.maybe_install()
    import torch
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        sys.path.insert(0, os.path.join(ROOT, 'tests'))
        import golden_util as gu
        from hybridq_amd.circuits import random_dense
        from hybridq_amd.dm import Kraus
        from hybridq_amd.dm import simulate as dm_simulate
        from hybridq_amd.simulation import simulate
        z = gu.load('e2e_dm_circuit.npz')
        circuit = []
        for i, kind in enumerate(bytes(z['kinds']).decode()):
            qs = tuple(int(q) for q in z[f'q{i}'])
            circuit.append(Kraus(list(z[f'L{i}']), qs, s=z[f's{i}']) if kind == 'K' else (z[f'U{i}'], qs))
        rho, info = dm_simulate(circuit, initial_state='0', complex_type='complex64', devices=world, return_info=True)
        # a plain circuit from a mixed '01+-' string, fused to 4 qubits like the reference's default
        n = 14
        init = ('+-01-' * n)[:n]
        gates = random_dense(n, 60, kmax=3, seed=31, unitary=True)
        psi = simulate(gates, initial_state=init, complex_type='complex128', devices=world, qubits=list(range(n)))
        # device-resident result: every rank keeps its slice of the canonical state
        sh = simulate(gates, initial_state=init, complex_type='complex128', shard_bits=int(np.log2(world)),
                      qubits=list(range(n)), return_numpy_array=False)
        mine = sh.to_complex().cpu().numpy()
        parts = [None] * world
        dist.all_gather_object(parts, mine)
        if rank == 0:
            np.savez(os.path.join(out_dir, 'api.npz'), rho=rho.reshape(-1), psi=psi.reshape(-1), cat=np.concatenate(parts),
                     n_x=info['n_exchanges'], transport=str(info['exchange_transport']))
        try:
            simulate(gates, initial_state=init, devices=2 * world, qubits=list(range(n)))
            raise SystemExit('devices != world size must be refused')
        except RuntimeError:
            pass
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 4])
def test_simulate_and_dm_simulate_sharded_api(torch_cuda, tmp_path, world):
    """simulate(..., devices=N) / dm.simulate(..., devices=N): BASELINE config 5 in miniature (noisy
    6-qubit circuit = 12-qubit state vector, the reference's rho from e2e_dm_circuit.npz) and a
    mixed-string circuit, sharded over N ranks on one GPU through the library's exchange."""
    import torch.multiprocessing as mp
    import golden_util as gu
    import oracle
    from hybridq_amd.circuits import random_dense
    from hybridq_amd.dm import Kraus, to_statevector_circuit
    from tolerances import circuit_tol
    mp.spawn(_api_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    out = np.load(os.path.join(str(tmp_path), 'api.npz'))
    z = gu.load('e2e_dm_circuit.npz')
    circuit = []
    for i, kind in enumerate(bytes(z['kinds']).decode()):
        qs = tuple(int(q) for q in z[f'q{i}'])
        circuit.append(Kraus(list(z[f'L{i}']), qs, s=z[f's{i}']) if kind == 'K' else (z[f'U{i}'], qs))
    sv = to_statevector_circuit(circuit)
    assert str(out['transport']).split(' | ')[0] == WANT_TRANSPORT and int(out['n_x']) >= 1
    assert np.abs(out['rho'] - z['rho']).max() / np.abs(z['rho']).max() < circuit_tol(sv, sv)
    n = 14
    init = ('+-01-' * n)[:n]
    gates = random_dense(n, 60, kmax=3, seed=31, unitary=True)
    exp = oracle.evolve_tensordot(gates, n, initial_state=init, qubits=list(range(n)))
    assert np.abs(out['psi'] - exp).max() / np.abs(exp).max() < 1e-12
    assert np.abs(out['cat'] - exp).max() / np.abs(exp).max() < 1e-12  # rank-ordered shards = canonical state


@pytest.mark.parametrize('workload,qubits', [('rqc_1q2q', 21), ('dm', 20)])
def test_bench_eight_ranks_sharing_the_gpu(torch_cuda, workload, qubits):
    """`bench.py --gpus 8` exactly as the driver launches it (torch.distributed.run, one process per rank), both sharded
    workloads (BASELINE configs 3 and 5 at toy size), with the eight ranks sharing the one GPU of this box
    (HQ_BENCH_SHARE_GPU=1: gloo process group, peer-to-peer exchange through HIP IPC): the whole N > 1 code path of the
    bench -- planner, exchanges with folded permutations, per-op events, exchange timing block with its analytic
    expectation -- runs and prints its two well-formed JSON lines (headline first, complete line last).  The numbers mean nothing on a shared GPU."""
    import json
    import subprocess
    env = dict(os.environ, HQ_BENCH_SHARE_GPU='1', OMP_NUM_THREADS='1')
    script = [os.path.join(ROOT, 'tests', 'emu', 'run_emulated.py'), 'bench.py'] if EMU else [os.path.join(ROOT, 'bench.py')]
    if EMU:
        qubits -= 4  # 8 x 2^14-amplitude shards are plenty for the emulation
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '8', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port())] + script + ['--gpus', '8', '--steps', '2', '--warmup', '1',
           '--qubits', str(qubits), '--depth', '6', '--workload', workload, '--no-cpu-baseline']
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 2, out.stdout[-2000:]  # the headline right after the timed region, the complete line last (rank 0 only)
    h, d = json.loads(lines[0]), json.loads(lines[1])
    assert h['line'].startswith('headline') and d['line'] == 'complete' and 'exchange' not in h
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'scaling', 'dtype', 'config', 'roofline'):
        assert h[key] == d[key], key
    # which transport the exchange really took, and between how many processes (RCCL: as ncclCommCount reports it)
    assert h['exchange_transport']['transport'] == WANT_TRANSPORT and h['exchange_transport']['world'] == 8
    assert h['exchange_transport']['rccl_ranks_seen'] == (8 if WANT_TRANSPORT == 'rccl' else 0), h['exchange_transport']
    assert d['n_gpus'] == 8 and d['scaling'] == 'weak' and d['steps'] == 2 and d['value'] > 0
    assert d['config']['n_qubits'] == qubits and d['config']['exchanges_per_step'] >= 1
    ex = d['exchange']
    assert ex['transport'] == WANT_TRANSPORT and (EMU or not ex['transport_note']), ex
    assert ex['expected']['bytes_per_link_per_exchange'] == ex['bytes_per_link'] == d['config']['state_bytes_per_gpu'] // 8
    assert ex['expected']['ms_at_153GBps'] == pytest.approx(ex['bytes_per_link'] / 153e9 * 1e3)
    assert 'extras_error' not in d, d['extras_error']
