"""One cache-blocked pass at n = 30 with G copies of a k-qubit inner gate: ms(G).  Serial phases give
base + c * G, overlapped phases max(base, c * G).  Usage: blocked_scaling.py [k] [n]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hybridq_amd import core  # noqa: E402
from hybridq_amd.simulation import alloc_planes  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rng = np.random.default_rng(0)
planes = alloc_planes(n, torch.float32, 'cuda')
core.init_state(planes[0], planes[1], 'plus')
tile = list(range(8)) + [12, 15, 19, 22, 27]
out = []
sets = [[int(x) for x in a.split(',')] for a in sys.argv[3:]] or [[5, 12, 19, 27][:k], [2, 3, 5, 12][:k], [15, 19, 22, 27][:k]]
for targets in sets:
    k = len(targets)
    row = []
    for G in (0, 1, 2, 4, 6, 8, 12, 16):
        gates = []
        for _ in range(max(G, 0)):
            q, _r = np.linalg.qr(rng.standard_normal((1 << k, 1 << k)) + 1j * rng.standard_normal((1 << k, 1 << k)))
            gates.append((q.astype(np.complex64), targets))
        if G == 0:  # a pass cannot be empty: time the streaming alone with a 1-qubit identity-free stand-in
            gates = [(np.eye(2, dtype=np.complex64), [5])]
        packed = core.pack_blocked(gates, 'complex64')
        core.apply_blocked(planes[0], planes[1], tile, packed=packed, n_qubits=n)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        core.use_torch_stream()
        e0.record()
        for _ in range(5):
            core.apply_blocked(planes[0], planes[1], tile, packed=packed, n_qubits=n)
        e1.record()
        torch.cuda.synchronize()
        row.append((G, e0.elapsed_time(e1) / 5))
    print(f'k={k} targets={targets} PREF={os.environ.get("HQ_BLOCKED_PREF", "1")}:', ' '.join(f'G{g}={t:.2f}' for g, t in row), flush=True)
