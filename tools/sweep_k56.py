"""Developer sweep of the k = 5, 6 matrix-core kernel (apply_mfma_big_kernel): per-call time for a set of
position patterns, float32 at n (default 30) and float64 at n-1.  A/B between builds of the library:
    HQ_HIP_LIBRARY=tools/_ab/libhq_hip_vX.so python tools/sweep_k56.py [n]"""
import os
import sys

os.environ.setdefault('OPENBLAS_NUM_THREADS', '8')
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hybridq_amd import core  # noqa: E402
from hybridq_amd.circuits import haar_unitary  # noqa: E402

n0 = int(sys.argv[1]) if len(sys.argv) > 1 else 30
tag = os.path.basename(os.environ.get('HQ_HIP_LIBRARY', 'in-tree'))
core.set_stream(torch.cuda.current_stream().cuda_stream)
rng = np.random.default_rng(0)
for dt, n in (('float32', n0), ('float64', n0 - 1)):
    esz = 4 if dt == 'float32' else 8
    planes = torch.empty((2, 1 << n), dtype=getattr(torch, dt), device='cuda')
    core.init_state(planes[0], planes[1], 'plus')
    for p in range(0, n, 2):
        core.apply_U(planes[0], planes[1], haar_unitary(2, rng), [p])
    core.sync()
    for k in (5, 6):
        pats = [list(range(8, 8 + k)), [1, 5, 9, 14, 20, 25][:k], list(range(2, 2 + k)), list(range(n - k, n)),
                [0, 7, 13, 21, n - 2, n - 1][:k], list(range(k))]
        tot = 0.0
        for pos in pats:
            U = haar_unitary(1 << k, rng)
            core.apply_U(planes[0], planes[1], U, pos)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(5):
                core.apply_U(planes[0], planes[1], U, pos)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            tot += ms
            print(f'{tag:<22} {dt} n={n} k={k} pos={str(pos):<28} {core.last_kernel_desc():<52} {ms:7.3f} ms {4 * esz * (1 << n) / ms / 1e6:7.0f} GB/s '
                  f'{8.0 * (1 << k) * (1 << n) / ms / 1e9:6.1f} TF', flush=True)
        print(f'{tag:<22} {dt} k={k} mean {tot / len(pats):7.3f} ms', flush=True)
    del planes
