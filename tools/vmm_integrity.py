"""Integrity of VMM-backed buffers (hq_alloc_scattered): distinct pattern written and read back over the
whole range, allocate / free / allocate again, granule sizes and shuffled mapping."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hybridq_amd import core  # noqa: E402

core.use_torch_stream()
torch.zeros(1, device='cuda')
NB = 8 << 30


def check(tag, buf):
    t = torch.as_tensor(buf.view(0, (NB // 8,), '<i8'), device='cuda')
    torch.arange(NB // 8, out=t)
    torch.cuda.synchronize()
    ok1 = bool((t[::1048573] == torch.arange(0, NB // 8, 1048573, device='cuda')).all())
    s = int(t.sum())  # wraps mod 2^64 like the exact value does
    exp = (NB // 8) * (NB // 8 - 1) // 2
    t.mul_(3)
    torch.cuda.synchronize()
    ok2 = bool((t[7::2097143] == 3 * torch.arange(7, NB // 8, 2097143, device='cuda')).all())
    print(f'{tag:<52} ptr 0x{buf.ptr:x} sampled {ok1} sum {"ok" if s == exp else "WRONG"} after-update {ok2}', flush=True)
    del t


for rnd in range(3):
    for gran, seed in ((2 << 20, 1), (2 << 20, 0), (64 << 20, 3), (1 << 30, 1)):
        buf = core.DeviceBuffer(NB, contiguous=False, scattered=gran, seed=seed)
        check(f'round {rnd}: VMM granule {gran >> 20} MiB seed {seed}', buf)
        torch.cuda.synchronize()
        buf.free()
    b = core.DeviceBuffer(NB, contiguous=False)
    check(f'round {rnd}: hipMalloc', b)
    b.free()
