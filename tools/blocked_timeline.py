"""Read the s_memtime stamps of a -DHQ_EXP_TIMELINE build (HQ_HIP_LIBRARY=...): one workgroup's 8 waves through one
three-qubit inner gate of a 12-gate cache-blocked pass at n = 30.  Cycles relative to the earliest stamp 0."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hybridq_amd import core  # noqa: E402
from hybridq_amd.simulation import alloc_planes  # noqa: E402

n = 30
rng = np.random.default_rng(0)
planes = alloc_planes(n, torch.float32, 'cuda')
core.init_state(planes[0], planes[1], 'plus')
tile = list(range(8)) + [12, 15, 19, 22, 27]
gates = []
for _ in range(8):
    q, _r = np.linalg.qr(rng.standard_normal((8, 8)) + 1j * rng.standard_normal((8, 8)))
    gates.append((q.astype(np.complex64), [5, 12, 19]))
packed = core.pack_blocked(gates, 'complex64')
for _ in range(3):
    core.apply_blocked(planes[0], planes[1], tile, packed=packed, n_qubits=n)
core.sync()
buf = (ctypes.c_ulonglong * 512)()
assert core._lib.hq_debug_timeline(buf) == 0
allt = np.array(buf[:], dtype=np.int64).reshape(32, 16)
t = allt[:8]
t0 = t[:, 0].min()
names = ['gate start', 'in gate fn', 'prologue done', 'it1 reads issued', 'it1 reads back', 'it1 MFMAs done', 'it1 writes issued',
         'it2 reads issued', 'it2 reads back', 'it2 MFMAs done', 'it2 writes issued', 'before barrier', 'after barrier']
idx = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12]
print('%-20s' % 'wave', ' '.join('%7d' % w for w in range(8)))
for nm, i in zip(names, idx):
    print('%-20s' % nm, ' '.join('%7d' % (t[w, i] - t0) for w in range(8)))

tt = allt[8:16]
t0 = tt[:, 0].min()
print()
print('tile level (same workgroup, 4th tile):')
for nm, i in zip(['tile start', 'tile base done', 'after barrier (gates start)', 'gates done', 'stores issued', 'after barrier',
                  'LDS filled (next tile)', 'next prefetch issued'], range(8)):
    print('%-28s' % nm, ' '.join('%7d' % (tt[w, i] - t0) for w in range(8)))
