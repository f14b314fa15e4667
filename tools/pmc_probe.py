"""HBM-traffic probe for rocprofv3 --pmc passes: a read-only kernel (norm2: 8*2^n B), a
write-only kernel (init_state: 8*2^n B) for calibration, then one launch of each gate
kernel class (16*2^n algorithmic bytes each).  Usage: python tools/pmc_probe.py [n]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hybridq_amd import core  # noqa: E402
from hybridq_amd.circuits import haar_unitary  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
core.set_stream(torch.cuda.current_stream().cuda_stream)
from hybridq_amd.simulation import alloc_planes  # noqa: E402
planes = alloc_planes(n, torch.float32, 'cuda')  # the product's own (tuned) placement
rng = np.random.default_rng(0)
core.init_state(planes[0], planes[1], 'plus')
print('norm2', core.norm2(planes[0], planes[1]))
for pos in ([12], [3], [0], [12, 20], [2, 3], [0, 15], [10, 15, 20], [10, 14, 18, 22], [0, 9, 15, n - 1],
            [8, 9, 10, 11, 12], [0, 1, 9, 14, 20], [3, 9, 14, 20, 25, 28], [2, 5, 9, 12, 17, 21, 26],
            [0, 4, 8, 11, 13, 19, 22, 27], [1, 3, 6, 10, 12, 16, 20, 24, 28]):
    core.apply_U(planes[0], planes[1], haar_unitary(1 << len(pos), rng), pos)
    print(pos, core.last_kernel_desc())
# round 3: Auto sends k <= 3 gates with every target at bit >= 8 to the VALU kernel; the role kernel on the same positions
core.set_apply_mode('mfma')
for pos in ([12], [12, 20], [10, 15, 20]):
    core.apply_U(planes[0], planes[1], haar_unitary(1 << len(pos), rng), pos)
    print(pos, core.last_kernel_desc())
core.set_apply_mode('auto')
# round 3: index-bit permutations through bitperm_tile_kernel (algorithmic bytes: one plane read + written = 8 * 2^n B for
# permute_bits / swap, both planes = 16 * 2^n B for the exchange pack)
tmp = torch.empty(1 << n, dtype=torch.float32, device='cuda')
rng2 = np.random.default_rng(1)
for name, perm in (('random above bit 4', np.concatenate([np.arange(4), 4 + rng2.permutation(n - 4)])), ('random every bit', rng2.permutation(n)),
                   ('bit reversal', np.arange(n)[::-1].copy())):
    core.permute_bits(planes[0], tmp, perm, n)
    print('permute_bits', name)
for s in (8, 13, 15, 16):
    pos = np.roll(np.arange(s), 3)
    core.swap(planes[0], pos, n)
    print('swap in place s =', s)
del tmp
core.shard_free()
half = planes[:, :1 << (n - 1)]
dst = torch.empty((2, 1 << (n - 1)), dtype=torch.float32, device='cuda')
ev = [4, 12, 21]
core.exchange(half[0], half[1], dst[0], dst[1], np.array([b for b in range(n - 1) if b not in ev] + ev), n - 1)
print('exchange pack, both planes of an n-1 qubit shard')
core.sync()
print('norm2', core.norm2(planes[0], planes[1]))
