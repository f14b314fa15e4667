"""Rates of the index-bit permutations on one 4 GiB plane (n = 30, float32; n = 29 float64): hq_permute_bits for a set of
permutations (what moves decides what the old gather kernel pays), the one-rank exchange pack (both planes), and the
in-place low-bit swaps s = 13..16.  Algorithmic bytes = read + write of the plane(s) once.
    HQ_PERM_TILE=0  the round-2 paths (gather kernel / two-pass swap)        HQ_PERM_TB, HQ_PERM_GRID: tile bits, grid
python tools/perm_rate.py [n]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hybridq_amd import core  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
tag = f'TILE={os.environ.get("HQ_PERM_TILE", "1")} TB={os.environ.get("HQ_PERM_TB", "-")} GRID={os.environ.get("HQ_PERM_GRID", "-")}'
core.use_torch_stream()
rng = np.random.default_rng(0)


def timeit(name, fn, nbytes, reps=4):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f'{tag} {name:<64} {ms:8.3f} ms {nbytes / ms / 1e6:8.0f} GB/s', flush=True)


def perms(m):
    out = {}
    p = np.arange(m); p[[m - 1, 20]] = p[[20, m - 1]]; out['swap(top,20)'] = p
    p = np.arange(m); p[[m - 1, 5]] = p[[5, m - 1]]; out['swap(top,5)'] = p
    p = np.arange(m); p[[m - 1, 3]] = p[[3, m - 1]]; out['swap(top,3)'] = p
    p = np.arange(m); p[[m - 1, 1]] = p[[1, m - 1]]; out['swap(top,1)'] = p
    p = np.arange(m); p[[m - 1, m - 2, m - 3]] = [2, 9, 17]; p[[2, 9, 17]] = [m - 1, m - 2, m - 3]; out['evict 3 qubits (2,9,17) to the top'] = p
    ev = [4, 12, 21]  # eviction as the planner writes it: evictees to the top, everything above them shifts down
    rest = [b for b in range(m) if b not in ev]
    out['evict (4,12,21) with shifts'] = np.array(rest + ev)
    out['random above bit 4'] = np.concatenate([np.arange(4), 4 + rng.permutation(m - 4)])
    out['random above bit 2'] = np.concatenate([np.arange(2), 2 + rng.permutation(m - 2)])
    out['random, every bit'] = rng.permutation(m)
    out['bit reversal'] = np.arange(m)[::-1].copy()
    out['rotate by 7'] = np.roll(np.arange(m), 7)
    return out


for dt, m in ((torch.float32, n), (torch.float64, n - 1)):
    src = torch.arange(1 << m, device='cuda').to(dt)
    dst = torch.empty_like(src)
    nbytes = src.numel() * src.element_size()
    for name, p in perms(m).items():
        timeit(f'{str(dt)[6:]} permute_bits {name}', lambda: core.permute_bits(src, dst, p, m), 2 * nbytes)
    for s in (8, 11, 13, 14, 15, 16):
        pos = rng.permutation(s)
        timeit(f'{str(dt)[6:]} swap in place s={s} ({sum(int(p) != i for i, p in enumerate(pos))} moved)', lambda: core.swap(src, pos, m), 2 * nbytes)
    pos = np.roll(np.arange(16), 3)  # no fixed point
    timeit(f'{str(dt)[6:]} swap in place s=16 (all 16 moved)', lambda: core.swap(src, pos, m), 2 * nbytes)
    del src, dst
    # the pack pass of the exchange on both planes (one rank: the permutation alone)
    core.shard_free()
    a = torch.zeros((2, 1 << (m - 1)), dtype=dt, device='cuda')
    b = torch.empty_like(a)
    p = np.array([x for x in range(m - 1) if x not in (4, 12, 21)] + [4, 12, 21])
    timeit(f'{str(dt)[6:]} exchange pack (both planes), evict (4,12,21)', lambda: core.exchange(a[0], a[1], b[0], b[1], p, m - 1), 2 * 2 * a[0].numel() * a.element_size())
    del a, b
