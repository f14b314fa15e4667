"""k = 7..10 matrix-core kernel (apply_gemm_kernel): ms per call and TFLOP/s at n (default 30) complex64 and n-1
complex128, two position patterns each.  HQ_GEMM_PREF=0 switches the register prefetch off."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hybridq_amd import core  # noqa: E402
from hybridq_amd.circuits import haar_unitary  # noqa: E402

n0 = int(sys.argv[1]) if len(sys.argv) > 1 else 30
core.use_torch_stream()
rng = np.random.default_rng(0)
for dt, n, ks in (('float32', n0, (7, 8, 9, 10)), ('float64', n0 - 1, (7, 8, 9))):
    if os.environ.get('SWEEP_ALLOC', 'torch') == 'tuned':
        from hybridq_amd.simulation import alloc_planes
        planes = alloc_planes(n, getattr(torch, dt), 'cuda')
    else:
        planes = torch.empty((2, 1 << n), dtype=getattr(torch, dt), device='cuda')
    core.init_state(planes[0], planes[1], 'plus')
    for k in ks:
        for pos in (sorted(int(p) for p in rng.permutation(n)[:k]), list(range(3, 3 + k)), [6, 15, 16, 19, 20, 21, 23, 27, 28, 29][:k]):
            U = haar_unitary(1 << k, rng).astype('complex64' if dt == 'float32' else 'complex128')
            core.apply_U(planes[0], planes[1], U, pos, n)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                core.apply_U(planes[0], planes[1], U, pos, n)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 3
            tf = 8.0 * (1 << k) * (1 << n) / ms / 1e9
            print(f'PREF={os.environ.get("HQ_GEMM_PREF", "1")} {dt} n={n} k={k} {core.last_kernel_desc():42s} {ms:8.3f} ms {tf:7.1f} TFLOP/s  pos={pos}', flush=True)
