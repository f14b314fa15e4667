"""Why does the same gate kernel stream at 6.3-6.4 TB/s on a 64+ GiB state and at 5.3-5.6 TB/s on the
8 GiB state of n = 30?  Same n = 30 planes placed in allocations of different size / alignment."""
import os
import sys

os.environ.setdefault('OPENBLAS_NUM_THREADS', '8')
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hybridq_amd import core  # noqa: E402
from hybridq_amd.circuits import haar_unitary  # noqa: E402

n = 30
core.use_torch_stream()
rng = np.random.default_rng(0)
GATES = [([3], haar_unitary(2, rng)), ([12], haar_unitary(2, rng)), ([22], haar_unitary(2, rng)), ([29], haar_unitary(2, rng)),
         ([4, 28], haar_unitary(4, rng)), ([9, 17], haar_unitary(4, rng)), ([27, 29], haar_unitary(4, rng))]


def measure(tag, re, im):
    core.init_state(re, im, 'plus')
    tot = 0.0
    out = []
    for pos, U in GATES:
        core.apply_U(re, im, U, pos, n)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(5):
            core.apply_U(re, im, U, pos, n)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        tot += ms
        out.append(f'{ms:.3f}')
    print(f'{tag:<58} re=0x{re.data_ptr():x} im-re=0x{im.data_ptr() - re.data_ptr():x}  mean {tot / len(GATES):.3f} ms = '
          f'{16.0 * (1 << n) / (tot / len(GATES)) / 1e9:.3f} TB/s   [{" ".join(out)}]', flush=True)


N = 1 << n
pad = 12288 // 4
# A: what alloc_planes does today
raw = torch.empty((2, N + pad), dtype=torch.float32, device='cuda')
measure('A own 8 GiB allocation (alloc_planes)', raw[0, :N], raw[1, :N])
del raw
torch.cuda.empty_cache()
# B..: planes inside a slab of S GiB, at offset off GiB
for S, off in ((16, 0), (32, 0), (64, 0), (128, 0), (128, 64), (200, 0), (200, 150)):
    try:
        slab = torch.empty(S << 28, dtype=torch.float32, device='cuda')  # S GiB
    except Exception as e:  # noqa: BLE001
        print('slab', S, 'failed', repr(e)[:80])
        continue
    base = (off << 28)
    re = slab[base:base + N]
    im = slab[base + N + pad:base + 2 * N + pad]
    measure(f'B planes inside a {S} GiB slab at +{off} GiB', re, im)
    del slab, re, im
    torch.cuda.empty_cache()
# C: two separate 4 GiB allocations
re = torch.empty(N, dtype=torch.float32, device='cuda')
im = torch.empty(N + pad, dtype=torch.float32, device='cuda')[pad:]
measure('C two separate allocations', re, im)
del re, im
torch.cuda.empty_cache()
# D: A again (is it the order of allocation / what was freed before?)
raw = torch.empty((2, N + pad), dtype=torch.float32, device='cuda')
measure('D own 8 GiB allocation again, after the slabs were freed', raw[0, :N], raw[1, :N])
