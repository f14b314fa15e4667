"""Developer sweep: does the byte distance between the re and im planes matter (DRAM
channel/bank mapping)?  python tools/sweep_pad.py [n]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hybridq_amd import core  # noqa: E402
from hybridq_amd.circuits import haar_unitary  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
core.set_stream(torch.cuda.current_stream().cuda_stream)
rng = np.random.default_rng(0)
buf = torch.empty((1 << (n + 1)) + (64 << 20), dtype=torch.float32, device='cuda')
cases = [[13], [12], [15], [20], [22], [24], [27], [29], [22, 27], [23, 25], [13, 14], [3, 22], [10, 15, 20], [22, 24, 27]]
PADS = [int(x) for x in os.environ.get('PADS', '0,4096').split(',')]
for pad_bytes in PADS:
    pe = pad_bytes // 4
    re = buf[:1 << n]
    im = buf[(1 << n) + pe:(1 << (n + 1)) + pe]
    core.init_state(re, im, 'plus')
    for p in range(0, n, 2):
        core.apply_U(re, im, haar_unitary(2, rng), [p])
    out = []
    for pos in cases:
        U = haar_unitary(1 << len(pos), rng)
        core.apply_U(re, im, U, pos)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(6):
            core.apply_U(re, im, U, pos)
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / 6)
    print(f'pad={pad_bytes:>9} ' + ' '.join(f'{str(c)}:{t:.3f}' for c, t in zip(cases, out)), flush=True)
