"""One cache-blocked pass with 12 three-qubit inner gates at n = 30, a few launches (for rocprofv3 --pmc passes)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hybridq_amd import core  # noqa: E402
from hybridq_amd.simulation import alloc_planes  # noqa: E402

n = 30
rng = np.random.default_rng(0)
planes = alloc_planes(n, torch.float32, 'cuda', vmm=False) if 'vmm' in alloc_planes.__code__.co_varnames else alloc_planes(n, torch.float32, 'cuda')
core.init_state(planes[0], planes[1], 'plus')
tile = list(range(8)) + [12, 15, 19, 22, 27]
for G in (4, 8):
    gates = []
    for _ in range(G):
        q, _r = np.linalg.qr(rng.standard_normal((8, 8)) + 1j * rng.standard_normal((8, 8)))
        gates.append((q.astype(np.complex64), [5, 12, 19]))
    packed = core.pack_blocked(gates, 'complex64')
    for _ in range(3):
        core.apply_blocked(planes[0], planes[1], tile, packed=packed, n_qubits=n)
    core.sync()
    print(G, core.last_kernel_desc())
