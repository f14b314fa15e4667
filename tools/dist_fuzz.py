"""Randomised campaign over the sharded evolution (hybridq_amd.dist: exchange planner, map bookkeeping, overlapped exchanges,
cache-blocked local passes, restore_order) between REAL processes over gloo, the per-shard arithmetic done by the CPU
oracle (the host stand-in of tests/test_dist_cpu.py) -- no GPU needed.  One spawn per world size, many random circuits,
initial states and options per spawn; rank 0 compares every final state with the oracle's single-process tensordot
evolution.  Test infrastructure (imports oracle/):

    python tools/dist_fuzz.py [trials per world size] [seed] [world sizes, e.g. 2,4,8]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]


def worker(rank, world, port, trials, seed):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    fails = 0
    try:
        import oracle
        import hybridq_amd.dist as dist_mod
        from hybridq_amd.circuits import random_dense, rqc_1q2q
        from hybridq_amd.dist import ShardedEvolution
        from test_dist_cpu import CpuBackend
        dist_mod.OVERLAP_MIN_SUB_QUBITS = 3  # (the product wants >= 12-qubit pieces; the shards here are tiny)
        g = world.bit_length() - 1
        rng = np.random.default_rng(seed)  # the same stream on every rank
        for trial in range(trials):
            ct = 'complex128' if rng.random() < 0.7 else 'complex64'
            ft = np.float64 if ct == 'complex128' else np.float32
            n = int(rng.integers(2 * g + 4, 2 * g + 4 + 6)) if rng.random() < 0.85 else g + 14  # sometimes big enough for blocked passes
            kind = int(rng.integers(0, 3))
            s1, s2 = int(rng.integers(1 << 30)), int(rng.integers(1 << 30))
            if kind == 0:
                gates = rqc_1q2q(n, depth=int(rng.integers(2, 10)), seed=s1)
            elif kind == 1:
                gates = random_dense(n, int(rng.integers(5, 50)), kmax=int(rng.integers(1, min(5, n - g - 1))), seed=s1)
            else:
                gates = rqc_1q2q(n, depth=int(rng.integers(2, 6)), seed=s1) + random_dense(n, int(rng.integers(3, 25)), kmax=3, seed=s2)
            init = ''.join(rng.choice(list('01+-'), size=n)) if rng.random() < 0.5 else '0' * n
            overlap = bool(rng.random() < 0.5)
            mode = str(rng.choice(['plan', 'plan_blocked', 'simulate', 'simulate_fused', 'two_parts']))
            restore = bool(rng.random() < 0.5)
            sh = ShardedEvolution(n, complex_type=ct, initial_state=init, backend=CpuBackend(ft), overlap=overlap)
            if mode == 'plan':
                sh.run(sh.plan(gates))
            elif mode == 'plan_blocked':
                sh.run(sh.plan(gates, blocked=True))
            elif mode == 'simulate':
                sh.simulate(gates)
            elif mode == 'simulate_fused':
                sh.simulate(gates, compress=int(rng.integers(2, 6)))
            else:
                cut = int(rng.integers(1, len(gates)))
                sh.simulate(gates[:cut])
                sh.simulate(gates[cut:], blocked=bool(rng.random() < 0.5))
            if restore:
                sh.restore_order()
            psi = sh.state_numpy()
            if rank == 0:
                exp = oracle.evolve_tensordot(gates, n, initial_state=init, qubits=list(range(n)))
                err = float(np.abs(psi - exp).max() / np.abs(exp).max())
                tol = 1e-12 if ct == 'complex128' else 1e-5
                if not err < tol:
                    fails += 1
                    print(f'FAIL world={world} trial={trial} n={n} {ct} kind={kind} seeds=({s1},{s2}) init={init} overlap={overlap} '
                          f'mode={mode} restore={restore} err={err:.3e}', flush=True)
        if rank == 0:
            print(f'dist_fuzz world {world}, seed {seed}: {trials} circuits, failures: {fails}', flush=True)
    finally:
        dist.destroy_process_group()
    if fails:
        sys.exit(1)


if __name__ == '__main__':
    import socket

    import torch.multiprocessing as mp
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    worlds = [int(w) for w in sys.argv[3].split(',')] if len(sys.argv) > 3 else [2, 4, 8]
    for world in worlds:
        s = socket.socket()
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
        s.close()
        mp.spawn(worker, args=(world, port, trials, seed + world), nprocs=world, join=True)
