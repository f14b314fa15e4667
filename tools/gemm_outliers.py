"""Developer check: per-call GPU time (events) and wall time of consecutive k=9 calls."""
import os
import sys

os.environ.setdefault('OPENBLAS_NUM_THREADS', '8')  # numpy's QR would spin 256 threads into the CFS quota
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hybridq_amd import core  # noqa: E402
from hybridq_amd.circuits import haar_unitary  # noqa: E402

n, k = 26, int(sys.argv[1]) if len(sys.argv) > 1 else 9
if len(sys.argv) > 2:
    core.set_apply_mode(sys.argv[2])
core.set_stream(torch.cuda.current_stream().cuda_stream)
planes = torch.empty((2, 1 << n), dtype=torch.float32, device='cuda')
core.init_state(planes[0], planes[1], 'plus')
rng = np.random.default_rng(0)
rows = []
for call in range(80):
    if call % 6 == 0:
        U = np.ascontiguousarray(haar_unitary(1 << k, rng), dtype='complex64')
        pos = np.ascontiguousarray(sorted(int(p) for p in rng.permutation(n)[:k]), dtype=np.uint32)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    core.apply_U(planes[0], planes[1], U, pos, n)
    e1.record()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    rows.append((call, e0.elapsed_time(e1), 1e3 * (t1 - t0), 1e3 * (t2 - t0)))
print('kernel', core.last_kernel(), 'k', k, 'slow calls:', [(r[0], round(r[1], 1), round(r[2], 1), round(r[3], 1)) for r in rows if r[3] > 20])
for r in rows[:0]:
    flag = ' <-- slow' if r[1] > 4 else ''
    print('call %2d  gpu %7.3f ms  issue %6.3f ms  wall %7.3f ms%s' % (r + (flag,)))
