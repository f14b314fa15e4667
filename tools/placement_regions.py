"""Does the rate of a placement come from particular physical granules?  One n = 30 float32 state on library-mapped granules;
the same k = 1 gate is timed on SUB-STATES: region r of 2^m amplitudes (both planes, same offsets) for m = 27, 25, 23.  If some
regions are reproducibly slower than others the slow draws are made of slow granules (and a selection could replace the lottery);
if all regions run alike the loss only exists when the whole state streams (interaction between far-apart pages)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from hybridq_amd import core  # noqa: E402
from hybridq_amd.circuits import haar_unitary  # noqa: E402
import placement_util as pu  # noqa: E402

n = 30
N = 1 << n
stride = N + 3072
gran = (int(sys.argv[1]) if len(sys.argv) > 1 else 2) << 20
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
core.use_torch_stream()
owner = pu._VmmPlanes(8 * stride, (2, stride), '<f4', gran, seed)
pl = torch.as_tensor(owner, device='cuda')[:, :N]
ms_full = pu._probe_ms(pl, n, np.float32)
print(f'granule {gran >> 20} MiB seed {seed}: full-state probe {ms_full:.3f} ms = {4 * N * 4 / ms_full / 1e9:.2f} TB/s', flush=True)
rng = np.random.default_rng(0)
U = haar_unitary(2, rng)
core.init_state(pl[0], pl[1], 'plus')
for m in (27, 25, 23):
    nreg = 1 << (n - m)
    reps = {27: 12, 25: 24, 23: 48}[m]
    rates = np.zeros((2, nreg))
    for trial in range(2):
        for r in range(nreg):
            a, b = pl[0][r << m:(r + 1) << m], pl[1][r << m:(r + 1) << m]
            core.apply_U(a, b, U, [m - 3], m)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(reps):
                core.apply_U(a, b, U, [m - 3], m)
            e1.record()
            torch.cuda.synchronize()
            rates[trial, r] = 16 * (1 << m) / (e0.elapsed_time(e1) / reps) / 1e9
    mean = rates.mean(0)
    corr = np.corrcoef(rates[0], rates[1])[0, 1] if nreg > 2 else float('nan')
    print(f'm={m}: {nreg} regions of {8 << (m - 20)} MiB (both planes): TB/s min {mean.min():.2f} median {np.median(mean):.2f} max {mean.max():.2f}; '
          f'trial-to-trial correlation {corr:.2f}; spread between the two trials (median |d|) {np.median(np.abs(rates[0] - rates[1])):.3f}', flush=True)
    if nreg <= 32:
        print('   per region:', ' '.join(f'{v:.2f}' for v in mean), flush=True)
    else:
        order = np.argsort(mean)
        print('   slowest 8:', ' '.join(f'{int(i)}:{mean[i]:.2f}' for i in order[:8]), '| fastest 8:', ' '.join(f'{int(i)}:{mean[i]:.2f}' for i in order[-8:]), flush=True)
