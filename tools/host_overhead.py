"""Developer check: per-call host overhead of apply_U through ctypes at small n."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hybridq_amd import core  # noqa: E402
from hybridq_amd.circuits import haar_unitary  # noqa: E402

core.set_stream(torch.cuda.current_stream().cuda_stream)
rng = np.random.default_rng(0)
for n in (14, 18, 22, 24, 26):
    re = torch.zeros(1 << n, dtype=torch.float32, device='cuda')
    im = torch.zeros(1 << n, dtype=torch.float32, device='cuda')
    core.init_state(re, im, 'plus')
    for k in (1, 2, 4):
        gates = [(np.ascontiguousarray(haar_unitary(1 << k, rng), dtype=np.complex64),
                  np.ascontiguousarray(rng.permutation(n)[:k], dtype=np.uint32)) for _ in range(64)]
        for U, pos in gates[:8]:
            core.apply_U(re, im, U, pos, n)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            for U, pos in gates:
                core.apply_U(re, im, U, pos, n)
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        print(f'n={n} k={k}: issue {1e6 * t_issue / (reps * 64):6.1f} us/call, end-to-end {1e6 * t_all / (reps * 64):6.1f} us/call',
              flush=True)
