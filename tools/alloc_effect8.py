"""Placement effect, eighth experiment: which VMM layout is reproducibly fastest?  Granule sizes x layouts,
each twice, n = 30 (and a check at n = 28 / 31 for the best ones)."""
import os
import sys

os.environ.setdefault('OPENBLAS_NUM_THREADS', '8')
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hybridq_amd import core  # noqa: E402
from hybridq_amd.circuits import haar_unitary  # noqa: E402

core.use_torch_stream()
torch.zeros(1, device='cuda')


def gates_for(n):
    rng = np.random.default_rng(0)
    return [([3], haar_unitary(2, rng)), ([12], haar_unitary(2, rng)), ([22], haar_unitary(2, rng)), ([n - 1], haar_unitary(2, rng)),
            ([4, n - 2], haar_unitary(4, rng)), ([9, 17], haar_unitary(4, rng)), ([n - 3, n - 1], haar_unitary(4, rng)),
            ([n - 9, n - 5, n - 2], haar_unitary(8, rng))]


def measure(tag, re, im, n):
    core.init_state(re, im, 'plus')
    out = []
    for pos, U in gates_for(n):
        core.apply_U(re, im, U, pos, n)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(3):
            core.apply_U(re, im, U, pos, n)
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / 3)
    nrm = core.norm2(re, im)
    m = sum(out) / len(out)
    print(f'{tag:<50} mean {m:7.3f} ms = {16.0 * (1 << n) / m / 1e9:.3f} TB/s  worst {max(out):.2f} norm {nrm:.6f}', flush=True)
    return m


def run(tag, n, gran, layout):
    N = 1 << n
    pp = (4 * N) // gran
    tot = 2 * pp
    if layout == 'identity':
        slots = list(range(tot))
    elif layout.startswith('rot'):
        num, den = (int(x) for x in layout[3:].split('/'))
        sh = pp * num // den
        slots = list(range(pp)) + [pp + (i + sh) % pp for i in range(pp)]
    elif layout.startswith('rotg'):
        pass
    elif layout.startswith('shuffle'):
        slots = [int(x) for x in np.random.default_rng(int(layout[7:])).permutation(tot)]
    buf = core.DeviceBuffer(tot * gran, scattered=gran, va_slots=slots)
    re = torch.as_tensor(buf.view(0, (N,), '<f4'), device='cuda')
    im = torch.as_tensor(buf.view(4 * N, (N,), '<f4'), device='cuda')
    m = measure(f'n={n} {gran >> 10:6d} KiB {tag}{layout}', re, im, n)
    del re, im
    buf.free()
    return m


for rep in range(2):
    for gran in (512 << 10, 2 << 20, 8 << 20):
        for layout in ('identity', 'rot1/2', 'rot1/4', 'rot1/8', 'rot3/8', 'rot1/3', 'shuffle1', 'shuffle2'):
            run(f'rep{rep} ', 30, gran, layout)
for n in (28, 31, 32):
    for layout in ('identity', 'rot1/2', 'shuffle1'):
        run('', n, 2 << 20, layout)
    N = 1 << n
    raw = torch.empty((2, N + 3072), dtype=torch.float32, device='cuda')
    measure(f'n={n} torch.empty (alloc_planes layout)', raw[0, :N], raw[1, :N], n)
    del raw
    torch.cuda.empty_cache()
