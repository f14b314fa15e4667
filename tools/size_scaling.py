"""Does the per-byte rate of a gate depend on the state size?  One gate per position class at
n = 30..35 (single GPU), ms and TB/s; plus a correctness property at every size (U then U^dagger
returns the uniform state: norm and marginals)."""
import os
import sys

os.environ.setdefault('OPENBLAS_NUM_THREADS', '8')
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hybridq_amd import core  # noqa: E402
from hybridq_amd.circuits import haar_unitary  # noqa: E402
from hybridq_amd.simulation import alloc_planes  # noqa: E402

sizes = [int(x) for x in sys.argv[1:]] or [30, 32, 34]
core.use_torch_stream()
rng = np.random.default_rng(0)
for n in sizes:
    planes = alloc_planes(n, torch.float32, torch.device('cuda'))
    core.init_state(planes[0], planes[1], 'plus')
    for pos in ([3], [12], [n - 8], [n - 1], [4, n - 2], [9, 17], [n - 3, n - 1]):
        U = haar_unitary(1 << len(pos), rng)
        core.apply_U(planes[0], planes[1], U, pos, n)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        reps = 3
        for _ in range(reps):
            core.apply_U(planes[0], planes[1], U, pos, n)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        for _ in range(reps + 1):  # undo: the state is |+...+> again
            core.apply_U(planes[0], planes[1], U.conj().T, pos, n)
        print(f'n={n} pos={str(pos):<12} {core.last_kernel_desc():<44} {ms:9.3f} ms  {16.0 * (1 << n) / ms / 1e9:6.3f} TB/s', flush=True)
    nrm = core.norm2(planes[0], planes[1])
    pr = core.probabilities(planes[0], planes[1], [3, 12, n - 8, n - 1], n)
    samp = planes[0][:: max(1, (1 << n) // 4096)][:4096].cpu().numpy()
    print(f'n={n} after U/U^dagger pairs: norm {nrm:.6f}, marginal spread {np.abs(pr - 1 / 16).max():.2e}, '
          f'sample max dev {np.abs(samp * 2 ** (n / 2) - 1).max():.2e}', flush=True)
    del planes
    torch.cuda.empty_cache()
