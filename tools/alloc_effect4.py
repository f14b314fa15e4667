"""Placement effect, fourth experiment: with physically contiguous planes the layout is deterministic,
so sweep (a) the distance between the re and im planes and (b) the absolute position (dummy
allocations in front)."""
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
    m = sum(out) / len(out)
    print(f'{tag:<44} mean {m:.3f} ms = {16.0 * (1 << n) / m / 1e9:.3f} TB/s  [{" ".join("%.2f" % x for x in out)}]', flush=True)


mode = sys.argv[1]
if mode == 'pad':
    buf = core.DeviceBuffer(10 * N, contiguous=True)  # 10 GiB: re at 0, im anywhere in [4, 6) GiB
    for pad in (0, 256, 1024, 4096, 8192, 12288, 16384, 20480, 32768, 65536, 98304, 1 << 17, 3 << 16, 1 << 18, 1 << 19, 1 << 20,
                3 << 19, 1 << 21, 3 << 20, 1 << 22, 1 << 23, 1 << 24, 3 << 23, 1 << 25, 1 << 26, 1 << 27, 1 << 28, 3 << 27, 1 << 29,
                1 << 30, (1 << 30) + 12288, (1 << 30) + (1 << 20)):
        re = torch.as_tensor(buf.view(0, (N,), '<f4'), device='cuda')
        im = torch.as_tensor(buf.view(4 * N + pad, (N,), '<f4'), device='cuda')
        measure(f'contiguous, im = re + 4 GiB + {pad}', re, im)
elif mode == 'abs':
    keep = []
    for front in (0, 8, 16, 32, 64, 96, 128, 160, 192, 224):
        while sum(b.nbytes for b in keep) < front << 30:
            keep.append(core.DeviceBuffer(8 << 30, contiguous=True))
        buf = core.DeviceBuffer(8 * N + 12288, contiguous=True)
        re = torch.as_tensor(buf.view(0, (N,), '<f4'), device='cuda')
        im = torch.as_tensor(buf.view(4 * N + 12288, (N,), '<f4'), device='cuda')
        measure(f'contiguous 8 GiB behind {front} GiB of other buffers', re, im)
        del re, im
        buf.free()
