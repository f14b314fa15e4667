"""Is the speed of a placement a property of WHICH physical granules back the state, or of the ORDER they are mapped in?
Draw K placements of an n = 30 float32 state (8 MiB granules, shuffled), probe each, then map the SAME physical granules
of the fastest and of the slowest draw in other orders (hq_vmm_remap: fresh virtual range, same handles) and probe again.
If the rate stays with the granule set the cause is physical (which pages / channels / rows), if it follows the order it
is the arrangement in the virtual index space."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from hybridq_amd import core  # noqa: E402
import placement_util as pu  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
K = int(sys.argv[2]) if len(sys.argv) > 2 else 6
gran = (int(sys.argv[3]) if len(sys.argv) > 3 else 8) << 20
N = 1 << n
stride = N + 3072
nbytes = 8 * stride
core.use_torch_stream()
remap = core._lib.hq_vmm_remap
remap.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_void_p)]
remap.restype = ctypes.c_int


def view(ptr):
    class V:
        __cuda_array_interface__ = {'shape': (2, stride), 'typestr': '<f4', 'data': (ptr, False), 'version': 2, 'strides': None}
    return torch.as_tensor(V(), device='cuda')[:, :N]


def tbps(ms):
    return 4 * N * 4 / ms / 1e9


draws = []
for k in range(K):
    owner = pu._VmmPlanes(nbytes, (2, stride), '<f4', gran, 100 + k)
    ms = pu._probe_ms(view(owner.buf.ptr), n, np.float32)
    draws.append([ms, owner, np.random.default_rng(100 + k).permutation(-(-nbytes // gran))])
    print(f'draw {k}: {ms:.3f} ms = {tbps(ms):.2f} TB/s', flush=True)
draws.sort(key=lambda d: d[0])
ng = len(draws[0][2])
orders = {'creation order': np.arange(ng), 'reversed': np.arange(ng)[::-1].copy(), 'shuffle A': np.random.default_rng(1).permutation(ng),
          'shuffle B': np.random.default_rng(2).permutation(ng), 'order of the other draw': None, 'its own first order again': None}
for name, (ms0, owner, order0), other in (('FASTEST', draws[0], draws[-1]), ('SLOWEST', draws[-1], draws[0])):
    print(f'{name} draw: {ms0:.3f} ms = {tbps(ms0):.2f} TB/s in its own order', flush=True)
    for oname, order in orders.items():
        if oname == 'order of the other draw':
            order = other[2]
        if oname == 'its own first order again':
            order = order0
        slots = np.ascontiguousarray(order, dtype=np.uint32)
        new = ctypes.c_void_p(None)
        rc = remap(ctypes.c_void_p(owner.buf.ptr), slots.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), ctypes.byref(new))
        if rc:
            print('   remap failed:', core.last_error())
            break
        owner.buf.ptr = int(new.value)
        ms = pu._probe_ms(view(owner.buf.ptr), n, np.float32)
        print(f'   same granules, {oname:<26}: {ms:.3f} ms = {tbps(ms):.2f} TB/s', flush=True)
