"""Placement effect, fifth experiment: physical granules mapped in a SHUFFLED order (VMM)."""
import os
import sys

os.environ.setdefault('OPENBLAS_NUM_THREADS', '8')
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hybridq_amd import core  # noqa: E402
from hybridq_amd.circuits import haar_unitary  # noqa: E402

n = 30
N = 1 << n
core.use_torch_stream()
torch.zeros(1, device='cuda')
rng = np.random.default_rng(0)
GATES = [([3], haar_unitary(2, rng)), ([12], haar_unitary(2, rng)), ([22], haar_unitary(2, rng)), ([n - 1], haar_unitary(2, rng)),
         ([4, n - 2], haar_unitary(4, rng)), ([9, 17], haar_unitary(4, rng))]


def measure(tag, re, im):
    core.init_state(re, im, 'plus')
    out = []
    for pos, U in GATES:
        core.apply_U(re, im, U, pos, n)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(4):
            core.apply_U(re, im, U, pos, n)
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / 4)
    nrm = core.norm2(re, im)
    m = sum(out) / len(out)
    print(f'{tag:<48} mean {m:.3f} ms = {16.0 * (1 << n) / m / 1e9:.3f} TB/s  [{" ".join("%.2f" % x for x in out)}] norm {nrm:.6f}', flush=True)


import time
for gran, seed in ((2 << 20, 1), (2 << 20, 0), (16 << 20, 1), (64 << 20, 1), (256 << 20, 1), (1 << 30, 1), (2 << 20, 7)):
    t0 = time.time()
    try:
        buf = core.DeviceBuffer(8 * N + (64 << 20), contiguous=False, scattered=gran, seed=seed)
    except Exception as e:  # noqa: BLE001
        print('granule', gran, 'failed:', repr(e)[:200])
        continue
    t1 = time.time()
    re = torch.as_tensor(buf.view(0, (N,), '<f4'), device='cuda')
    im = torch.as_tensor(buf.view(4 * N + 12288, (N,), '<f4'), device='cuda')
    measure(f'VMM granule {gran >> 20} MiB, seed {seed} (alloc {t1 - t0:.2f} s)', re, im)
    del re, im
    buf.free()
