"""Placement effect, sixth experiment: hipMalloc and VMM buffers alternating inside ONE process, several
rounds, same six gates: is the VMM advantage a property of the mapping or of the moment?"""
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
    m = sum(out) / len(out)
    print(f'{tag:<44} re=0x{re.data_ptr():x} mean {m:.3f} ms = {16.0 * (1 << n) / m / 1e9:.3f} TB/s  [{" ".join("%.2f" % x for x in out)}]', flush=True)


for rnd in range(3):
    p = alloc_planes(n, torch.float32, 'cuda', vmm=True)
    measure(f'round {rnd}: alloc_planes VMM', p[0], p[1])
    del p
    p = alloc_planes(n, torch.float32, 'cuda', vmm=False)
    measure(f'round {rnd}: alloc_planes torch', p[0], p[1])
    del p
    torch.cuda.empty_cache()
    buf = core.DeviceBuffer(8 * N + (64 << 20), contiguous=False, scattered=2 << 20, seed=1)
    re = torch.as_tensor(buf.view(0, (N,), '<f4'), device='cuda')
    im = torch.as_tensor(buf.view(4 * N + 12288, (N,), '<f4'), device='cuda')
    measure(f'round {rnd}: VMM 2 MiB granules shuffled', re, im)
    del re, im
    buf.free()
# both kinds alive at the same time
a = alloc_planes(n, torch.float32, 'cuda', vmm=True)
b = alloc_planes(n, torch.float32, 'cuda', vmm=False)
for rnd in range(2):
    measure('coexisting: VMM', a[0], a[1])
    measure('coexisting: torch', b[0], b[1])
