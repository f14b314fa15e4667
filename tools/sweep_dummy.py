"""Developer sweep: placement of the identity dummy digits of k < 3 gates in the role kernel."""
import os
import sys

os.environ.setdefault('OPENBLAS_NUM_THREADS', '8')
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hybridq_amd import core  # noqa: E402
from hybridq_amd.circuits import haar_unitary  # noqa: E402
from hybridq_amd.simulation import alloc_planes  # noqa: E402

n = 30
core.set_stream(torch.cuda.current_stream().cuda_stream)
planes = alloc_planes(n, torch.float32, torch.device('cuda'))
core.init_state(planes[0], planes[1], 'plus')
rng = np.random.default_rng(0)
for pos in ([0], [1], [2], [3], [4], [5], [6], [7], [9], [12], [20], [0, 1], [0, 2], [1, 3], [2, 3], [2, 4], [3, 5], [4, 5], [2, 12], [5, 12], [6, 12], [12, 20]):
    U = haar_unitary(1 << len(pos), rng)
    row = []
    for mode in ('dummy=comp', 'dummy=low', 'dummy=high'):
        core.set_apply_mode('mfma')
        core.set_apply_mode(mode)
        core.apply_U(planes[0], planes[1], U, pos)
        desc = core.last_kernel_desc()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(8):
            core.apply_U(planes[0], planes[1], U, pos)
        e1.record()
        torch.cuda.synchronize()
        row.append((e0.elapsed_time(e1) / 8, desc))
    core.set_apply_mode('dummy=auto')
    core.apply_U(planes[0], planes[1], U, pos)
    auto_desc = core.last_kernel_desc()
    core.set_apply_mode('auto')
    print(f'pos={str(pos):<10} comp {row[0][0]:6.3f} ms {row[0][1][18:36]}   low {row[1][0]:6.3f} ms {row[1][1][18:36]}   high {row[2][0]:6.3f} ms {row[2][1][18:36]}   auto -> {auto_desc[18:36]}', flush=True)
