"""Developer sweep: kernel time of apply_gemm_kernel vs target positions (random sets)."""
import os
import sys

os.environ.setdefault('OPENBLAS_NUM_THREADS', '8')  # numpy's QR would spin 256 threads into the CFS quota

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hybridq_amd import core  # noqa: E402
from hybridq_amd.circuits import haar_unitary  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 26
k = int(sys.argv[2]) if len(sys.argv) > 2 else 9
ntry = int(sys.argv[3]) if len(sys.argv) > 3 else 24
core.set_stream(torch.cuda.current_stream().cuda_stream)
planes = torch.empty((2, 1 << n), dtype=torch.float32, device='cuda')
core.init_state(planes[0], planes[1], 'plus')
rng = np.random.default_rng(1)
U = np.ascontiguousarray(haar_unitary(1 << k, rng), dtype='complex64')
extra = [[4, 9, 11, 12, 17, 18, 19, 20, 25], [0, 1, 2, 3, 9, 12, 18, 21, 24], [7, 10, 11, 13, 14, 15, 16, 18, 23]] if k == 9 else [[0, 5, 11, 17, 19, 20, 23, 24]]
for it in range(ntry):
    pos = np.ascontiguousarray(extra[it] if it < len(extra) else sorted(int(p) for p in rng.permutation(n)[:k]), dtype=np.uint32)
    if it % 2:
        U = np.ascontiguousarray(haar_unitary(1 << k, rng), dtype='complex64')
    core.apply_U(planes[0], planes[1], U, pos, n)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(3):
        core.apply_U(planes[0], planes[1], U, pos, n)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    cols = [p for p in range(n) if p not in pos][:14 - k]
    print(f'n={n} k={k} {ms:8.3f} ms  pos={pos.tolist()} cols={cols}', flush=True)
