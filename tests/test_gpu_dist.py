"""Two ranks sharing ONE GPU: the sharded evolution with the real HIP backend (kernels,
bit-permutation, device buffers).  RCCL refuses two ranks on one device, so the exchange
is staged through the host over gloo here; the 8-GPU run uses RCCL all_to_all_single on
the same buffers (hybridq_amd.dist.HipBackend.all_to_all)."""
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


def _worker(rank, world, port, n, ct, out_dir):
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
        sh = ShardedEvolution(n, complex_type=ct, initial_state='0' * n, backend=HostStagedExchange(ft))
        sched = sh.plan(gates)
        sh.run(sched)
        psi = sh.state_numpy()
        # same circuit, cache-blocked local passes between the exchanges, then canonical order
        shb = ShardedEvolution(n + 2, complex_type=ct, initial_state='0' * (n + 2), backend=HostStagedExchange(ft))
        gb = rqc_1q2q(n + 2, depth=8, seed=13)
        schedb = shb.plan(gb, blocked=True)
        shb.run(schedb)
        psib = shb.state_numpy()
        nb = sum(1 for op in schedb if op[0] == 'B')
        if rank == 0:
            np.savez(os.path.join(out_dir, 'out.npz'), psi=psi, psib=psib, nb=nb,
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
    assert np.abs(out['psi'] - exp).max() / np.abs(exp).max() < tol
    assert int(out['n_x']) >= 1
    expb = oracle.evolve_tensordot(rqc_1q2q(n + 2, depth=8, seed=13), n + 2)
    assert np.abs(out['psib'] - expb).max() / np.abs(expb).max() < 5 * tol
    assert int(out['nb']) >= 1  # blocked passes were really used
