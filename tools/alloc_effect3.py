"""Placement effect, third experiment: physically CONTIGUOUS VRAM (hipDeviceMallocContiguous through
hq_alloc) against the default allocation, fresh process per recipe."""
import os
import sys

os.environ.setdefault('OPENBLAS_NUM_THREADS', '8')
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hybridq_amd import core  # noqa: E402
from hybridq_amd.circuits import haar_unitary  # noqa: E402

recipe = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
N = 1 << n
pad = 12288
core.use_torch_stream()
torch.zeros(1, device='cuda')
rng = np.random.default_rng(0)
GATES = [([3], haar_unitary(2, rng)), ([12], haar_unitary(2, rng)), ([22], haar_unitary(2, rng)), ([n - 1], haar_unitary(2, rng)),
         ([4, n - 2], haar_unitary(4, rng)), ([9, 17], haar_unitary(4, rng)), ([n - 3, n - 1], haar_unitary(4, rng))]


def measure(tag, re, im):
    core.init_state(re, im, 'plus')
    tot = 0.0
    for pos, U in GATES:
        core.apply_U(re, im, U, pos, n)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(5):
            core.apply_U(re, im, U, pos, n)
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1) / 5
    print(f'{recipe:<12} n={n} {tag:<46} mean {tot / len(GATES):.3f} ms = {16.0 * (1 << n) / (tot / len(GATES)) / 1e9:.3f} TB/s', flush=True)


def planes_from(buf, off_re, off_im):
    re = torch.as_tensor(buf.view(off_re, (N,), '<f4'), device='cuda')
    im = torch.as_tensor(buf.view(off_im, (N,), '<f4'), device='cuda')
    return re, im


if recipe == 'default':
    raw = torch.empty((2, N + pad // 4), dtype=torch.float32, device='cuda')
    measure('torch.empty (alloc_planes layout)', raw[0, :N], raw[1, :N])
elif recipe == 'hipmalloc':
    buf = core.DeviceBuffer(8 * N + pad, contiguous=False)
    re, im = planes_from(buf, 0, 4 * N + pad)
    measure('hq_alloc default flags, one buffer', re, im)
elif recipe == 'contig1':
    buf = core.DeviceBuffer(8 * N + pad, contiguous=True)
    re, im = planes_from(buf, 0, 4 * N + pad)
    measure('hq_alloc CONTIGUOUS, one buffer', re, im)
elif recipe == 'contig2':
    b0, b1 = core.DeviceBuffer(4 * N, contiguous=True), core.DeviceBuffer(4 * N + pad, contiguous=True)
    re = torch.as_tensor(b0.view(0, (N,), '<f4'), device='cuda')
    im = torch.as_tensor(b1.view(pad, (N,), '<f4'), device='cuda')
    measure('hq_alloc CONTIGUOUS, one buffer per plane', re, im)
    print('   re 0x%x im 0x%x' % (re.data_ptr(), im.data_ptr()))
