"""Developer sweep: per-call time of apply_U through the C ABI at a given n for a list of
(k, positions, mode) cases.  Usage on the GPU box: python tools/sweep_apply.py [n] [dtype]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hybridq_amd import core  # noqa: E402
from hybridq_amd.circuits import haar_unitary  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dt = sys.argv[2] if len(sys.argv) > 2 else 'float32'
tdt = getattr(torch, dt)
esz = 4 if dt == 'float32' else 8
core.set_stream(torch.cuda.current_stream().cuda_stream)
planes = torch.empty((2, 1 << n), dtype=tdt, device='cuda')
core.init_state(planes[0], planes[1], 'plus')
rng = np.random.default_rng(0)
for p in range(n):
    core.apply_U(planes[0], planes[1], haar_unitary(2, rng), [p])
core.sync()


def timeit(pos, mode, reps=8):
    U = haar_unitary(1 << len(pos), rng)
    for m in mode.split('+'):
        core.set_apply_mode(m)
    core.apply_U(planes[0], planes[1], U, pos)
    kern = core.last_kernel()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        core.apply_U(planes[0], planes[1], U, pos)
    e1.record()
    torch.cuda.synchronize()
    core.set_apply_mode('auto')
    core.set_apply_mode('nt=auto')
    core.set_apply_mode('dummy=auto')
    ms = e0.elapsed_time(e1) / reps
    gbs = 4 * esz * (1 << n) / ms / 1e6
    print(f'k={len(pos)} pos={str(pos):<22} mode={mode:<14} kern={kern:<8} {ms:8.3f} ms {gbs:8.1f} GB/s {gbs/80:5.1f}%',
          flush=True)


H = n - 1
cases = []
for pos in ([8, 9, 10, 11, 12], [3, 9, 14, 20, 25], [0, 1, 2, 3, 4], [2, 3, 4, 5, 6], [0, 7, 13, 21, H], [H - 4, H - 3, H - 2, H - 1, H],
            [2, 9, 14, 20, H],
            [8, 9, 10, 11, 12, 13], [1, 5, 9, 14, 20, 25], [2, 3, 4, 5, 6, 7], [3, 9, 14, 18, 22, 26], [0, 1, 2, 3, 4, 5], [H - 5, H - 4, H - 3, H - 2, H - 1, H]):
    cases += [(pos, 'auto'), (pos, 'auto+nt=0'), (pos, 'auto+nt=1'), (pos, 'tile')]
for pos, mode in cases:
    timeit(pos, mode)
