"""PMC probe for the blocked kernel: one pass with 16 inner k=2 gates and one with 16 k=4 gates."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hybridq_amd import core  # noqa: E402
from hybridq_amd.circuits import haar_unitary  # noqa: E402

n = 30
core.set_stream(torch.cuda.current_stream().cuda_stream)
re = torch.empty(1 << n, dtype=torch.float32, device='cuda')
im = torch.empty((1 << n) + 3072, dtype=torch.float32, device='cuda')[3072:]
core.init_state(re, im, 'plus')
rng = np.random.default_rng(0)
tile = np.array(list(range(5)) + [7, 9, 12, 15, 18, 21, 25, 28], dtype=np.uint32)
for k in (2, 4):
    gates = [(haar_unitary(1 << k, rng), rng.permutation(tile)[:k]) for _ in range(16)]
    core.apply_blocked(re, im, tile, gates, n)
core.sync()
