"""In-place low-bit swap (swap_float32/64, reference: include/swap.h) on one 4 GiB plane: GB/s per width s.
HQ_SWAP_PREF=0 switches the register prefetch of the LDS-tile kernel off."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hybridq_amd import core  # noqa: E402

core.use_torch_stream()
rng = np.random.default_rng(0)
for dt, n in ((torch.float32, 30), (torch.float64, 29)):
    a = torch.arange(1 << n, device='cuda').to(dt)
    nbytes = a.numel() * a.element_size()
    for s in (2, 5, 8, 10, 11, 12, 13, 14, 16):
        pos = rng.permutation(s).astype(np.uint32)
        if s == 2:
            pos = np.array([1, 0], dtype=np.uint32)
        core.swap(a, pos, n)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            core.swap(a, pos, n)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 4
        print(f'PREF={os.environ.get("HQ_SWAP_PREF", "1")} {str(dt):14s} n={n} s={s:2d} {ms:7.3f} ms {2 * nbytes / ms / 1e6:7.0f} GB/s', flush=True)
