"""Developer check: wall time per call (host table building + upload + kernel) for k = 7..10.
Run with OPENBLAS_NUM_THREADS=4: numpy's QR of a 512x512 matrix spins one BLAS thread per hardware
thread (256) and the container's CFS quota (16 CPUs) then freezes the whole process for ~75 ms."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hybridq_amd import core  # noqa: E402
from hybridq_amd.circuits import haar_unitary  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 26
dt = sys.argv[2] if len(sys.argv) > 2 else 'float32'
core.set_stream(torch.cuda.current_stream().cuda_stream)
planes = torch.empty((2, 1 << n), dtype=getattr(torch, dt), device='cuda')
core.init_state(planes[0], planes[1], 'plus')
rng = np.random.default_rng(0)
ks = [int(x) for x in sys.argv[3].split(',')] if len(sys.argv) > 3 else range(5, (10 if dt == 'float32' else 9) + 1)
for k in ks:
    U = np.ascontiguousarray(haar_unitary(1 << k, rng), dtype='complex64' if dt == 'float32' else 'complex128')
    pos = np.ascontiguousarray(sorted(int(p) for p in rng.permutation(n)[:k]), dtype=np.uint32)
    core.apply_U(planes[0], planes[1], U, pos, n)
    core.sync()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        core.apply_U(planes[0], planes[1], U, pos, n)
    t_issue = (time.perf_counter() - t0) / reps
    core.sync()
    t_all = (time.perf_counter() - t0) / reps
    print(f'{dt} n={n} k={k}: issue (host) {1e3 * t_issue:8.3f} ms/call, wall {1e3 * t_all:8.3f} ms/call  kernel={core.last_kernel()} pos={pos.tolist()}', flush=True)
