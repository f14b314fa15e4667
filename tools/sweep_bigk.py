"""Developer sweep: per-call time of apply_U for k = 5..10 (auto / tile / generic).
Usage on the GPU box: python tools/sweep_bigk.py [n] [dtype] [kmax]"""
import os
import sys

os.environ.setdefault('OPENBLAS_NUM_THREADS', '8')  # numpy's QR would spin 256 threads into the CFS quota

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hybridq_amd import core  # noqa: E402
from hybridq_amd.circuits import haar_unitary  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dt = sys.argv[2] if len(sys.argv) > 2 else 'float32'
kmax = int(sys.argv[3]) if len(sys.argv) > 3 else 10
esz = 4 if dt == 'float32' else 8
core.set_stream(torch.cuda.current_stream().cuda_stream)
planes = torch.empty((2, 1 << n), dtype=getattr(torch, dt), device='cuda')
core.init_state(planes[0], planes[1], 'plus')
rng = np.random.default_rng(0)
for p in range(0, n, 2):
    core.apply_U(planes[0], planes[1], haar_unitary(2, rng), [p])
core.sync()
for k in range(5, kmax + 1):
    for pos in (list(range(8, 8 + k)), list(range(k)), sorted(int(p) for p in rng.permutation(n)[:k])):
        U = haar_unitary(1 << k, rng)
        for mode in ('auto', 'tile', 'generic'):
            if mode == 'tile' and k > 6:
                continue
            core.set_apply_mode(mode)
            core.apply_U(planes[0], planes[1], U, pos)
            kern = core.last_kernel()
            reps = 4 if k <= 7 else 2
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(reps):
                core.apply_U(planes[0], planes[1], U, pos)
            e1.record()
            torch.cuda.synchronize()
            core.set_apply_mode('auto')
            ms = e0.elapsed_time(e1) / reps
            tf = 8.0 * (1 << k) * (1 << n) / ms / 1e9
            print(f'{dt} n={n} k={k} pos={str(pos):<40} mode={mode:<8} kern={kern:<10} {ms:9.3f} ms  {4 * esz * (1 << n) / ms / 1e6:8.1f} GB/s  {tf:6.1f} TFLOP/s',
                  flush=True)
