"""Developer sweep of the auxiliary kernels at n qubits: swap, to_complex, permute_bits,
norm2, probabilities, project.  python tools/sweep_aux.py [n]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hybridq_amd import core  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
core.set_stream(torch.cuda.current_stream().cuda_stream)
re = torch.empty(1 << n, dtype=torch.float32, device='cuda')
im = torch.empty(1 << n, dtype=torch.float32, device='cuda')
core.init_state(re, im, 'plus')
out = torch.empty(1 << n, dtype=torch.complex64, device='cuda')
tmp = torch.empty(1 << n, dtype=torch.float32, device='cuda')
rng = np.random.default_rng(0)


def timeit(name, fn, bytes_moved, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f'{name:<46} {ms:8.3f} ms {bytes_moved / ms / 1e6:8.1f} GB/s', flush=True)


P = 4 * (1 << n)  # bytes of one plane
for s in (3, 6, 8, 10, 12, 13):
    pos = rng.permutation(s)
    timeit(f'swap_float32 one plane s={s} (lds)', lambda: core.swap(re, pos, n), 2 * P)
pos = rng.permutation(14)
timeit('swap_float32 one plane s=14 (gather+copy)', lambda: core.swap(re, pos, n), 2 * P)
order8 = np.array([0, 1, 5, 6, 7, 2, 3, 4])
timeit('swap pair re+im, reference order s=8', lambda: (core.swap(re, order8, n), core.swap(im, order8, n)), 4 * P)
timeit('to_complex', lambda: core.to_complex(re, im, out), 4 * P)
perm = np.arange(n); perm[n - 1], perm[20] = 20, n - 1
timeit('permute_bits one plane swap(29,20)', lambda: core.permute_bits(re, tmp, perm, n), 2 * P)
perm = np.arange(n); perm[n - 1], perm[5] = 5, n - 1
timeit('permute_bits one plane swap(29,5)', lambda: core.permute_bits(re, tmp, perm, n), 2 * P)
perm = np.arange(n); perm[n - 1], perm[1] = 1, n - 1
timeit('permute_bits one plane swap(29,1) scalar', lambda: core.permute_bits(re, tmp, perm, n), 2 * P)
perm = np.arange(n); perm[n - 3:] = [10, 15, 20]; perm[10], perm[15], perm[20] = n - 3, n - 2, n - 1
timeit('permute_bits one plane 3 swaps', lambda: core.permute_bits(re, tmp, perm, n), 2 * P)
timeit('norm2', lambda: core.norm2(re, im), 2 * P)
timeit('probabilities k=1 pos 0', lambda: core.probabilities(re, im, [0], n), 2 * P)
timeit('probabilities k=3', lambda: core.probabilities(re, im, [3, 17, 25], n), 2 * P)
timeit('probabilities k=10', lambda: core.probabilities(re, im, list(range(5, 15)), n), 2 * P)
timeit('project k=2', lambda: core.project(re, im, [3, 17], 2, 1.0, n), 4 * P)
timeit('init_state', lambda: core.init_state(re, im, 'plus'), 2 * P)
