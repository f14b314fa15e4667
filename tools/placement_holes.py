"""Forced physical scatter: does a set of granules taken as EVERY OTHER one of a twice-as-large sequence of allocations
(the rest released) stream faster than a set allocated in one go?  (Hypothesis: fast draws are the physically scattered
ones.)  n = 30 float32; hq_alloc_mapped creates the granules in sequence, so "dense" = N granules, "holes" = the buffer is
built while a dummy buffer of the same granule size is allocated alternately (dummy granules freed afterwards)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from hybridq_amd import core  # noqa: E402
import placement_util as pu  # noqa: E402

n = 30
N = 1 << n
stride = N + 3072
nbytes = 8 * stride
gran = (int(sys.argv[1]) if len(sys.argv) > 1 else 2) << 20
core.use_torch_stream()


def view(ptr):
    class V:
        __cuda_array_interface__ = {'shape': (2, stride), 'typestr': '<f4', 'data': (ptr, False), 'version': 2, 'strides': None}
    return torch.as_tensor(V(), device='cuda')[:, :N]


def probe(ptr):
    ms = pu._probe_ms(view(ptr), n, np.float32)
    return 4 * N * 4 / ms / 1e9


ng = -(-nbytes // gran)
for rnd in range(3):
    dense = core.DeviceBuffer(ng * gran, scattered=gran, va_slots=np.arange(ng))
    r_dense = probe(dense.ptr)
    # interleave: chunks of `c` granules for the state, `c` granules for a dummy, alternately
    for c in (1, 8):
        parts, dummies = [], []
        for i in range(0, ng, c):
            k = min(c, ng - i)
            parts.append(core.DeviceBuffer(k * gran, scattered=gran, va_slots=np.arange(k)))
            dummies.append(core.DeviceBuffer(k * gran, scattered=gran, va_slots=np.arange(k)))
        for d in dummies:
            d.free()
        # the parts are separate virtual ranges: the state needs ONE range, so time a gate per part instead?  No: map them
        # into one range is not possible without the handles; instead probe each placement through hq_alloc_state-like
        # draws is what the product does.  Here: free the parts and allocate the state NOW, into the holes just made.
        for p in parts:
            p.free()
        holes = core.DeviceBuffer(ng * gran, scattered=gran, va_slots=np.arange(ng))
        r_holes = probe(holes.ptr)
        print(f'round {rnd} granule {gran >> 20} MiB: dense {r_dense:.2f} TB/s | after punching holes in chunks of {c}: {r_holes:.2f} TB/s', flush=True)
        holes.free()
    dense.free()
