"""Developer sweep: time of one blocked pass vs the number / size of inner gates (n=30)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hybridq_amd import core  # noqa: E402
from hybridq_amd.circuits import haar_unitary  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
core.set_stream(torch.cuda.current_stream().cuda_stream)
re = torch.empty(1 << n, dtype=torch.float32, device='cuda')
im = torch.empty((1 << n) + 3072, dtype=torch.float32, device='cuda')[3072:]
core.init_state(re, im, 'plus')
rng = np.random.default_rng(0)
for p in range(0, n, 2):
    core.apply_U(re, im, haar_unitary(2, rng), [p])


def run(tile, gates, reps=5):
    packed = core.pack_blocked(gates)
    core.apply_blocked(re, im, tile, packed=packed, n_qubits=n)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        core.apply_blocked(re, im, tile, packed=packed, n_qubits=n)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


import itertools
for (tb, hi), dm in itertools.product(((13, [7, 9, 12, 15, 18, 21, 25, 28]),), ('dummy=auto', 'dummy=comp', 'dummy=low')):
    core.set_apply_mode(dm)
    print(dm)
    tile = np.array(list(range(tb - len(hi))) + hi, dtype=np.uint32)
    for k in (1, 2, 3):
        for ng in (8, 16):
            gates = [(haar_unitary(1 << k, rng), rng.permutation(tile)[:k]) for _ in range(ng)]
            ms = run(tile, gates)
            print(f'tb={tb} tile_hi={hi[:3]}.. k={k} gates={ng:2d}  {ms:8.3f} ms  ({ms / ng:6.3f} ms/gate)', flush=True)
