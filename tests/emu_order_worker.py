"""Worker of tests/test_emu_kernels.py (fresh process: the emulator reads HQ_EMU_ORDER once).  Runs one small case of every
kernel family that synchronises waves through LDS / barriers against the EMULATED library and prints `name sha256` of the
result; the test compares the digests of a forward, a reverse and a random wave schedule.  TEST INFRASTRUCTURE."""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import emu_util  # noqa: E402

core = emu_util.emu_core()
rng = np.random.default_rng(7)


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()[:24]


def rand_u(k, dt):
    d = 1 << k
    return ((rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d))) / np.sqrt(2.0 * d)).astype(dt)


for ft, ct in ((np.float32, np.complex64), (np.float64, np.complex128)):
    n = 14
    re, im, free = emu_util.device_planes(core, n, ft)
    psi = rng.standard_normal((2, 1 << n)).astype(ft)
    cases = [('auto', [3]), ('auto', [0, 9]), ('auto', [2, 5, 13]), ('auto', [9, 10, 12]), ('auto', [1, 4, 8, 11]),
             ('auto', [0, 3, 6, 9, 12]), ('auto', [1, 2, 5, 7, 9, 10]), ('auto', [0, 1, 2, 3, 4, 5, 6]),
             ('auto', [1, 2, 3, 4, 6, 7, 9, 11]), ('direct', [2, 7, 11]), ('mfma', [8, 10]), ('generic', [0, 5, 6, 12]),
             ('tile', [0, 3, 6, 9, 11]), ('gemm', [2, 3, 5, 7, 8, 11, 13]), ('naive', [4, 12])]
    for mode, pos in cases:
        re[:], im[:] = psi[0], psi[1]
        core.set_apply_mode(mode)
        try:
            core.apply_U(re, im, rand_u(len(pos), ct), pos, n)
        finally:
            core.set_apply_mode('auto')
        print(f'apply_{ft.__name__}_{mode}_{len(pos)}_{core.last_kernel()}', digest(re, im))
    # cache-blocked pass: k <= 4 gates inside one tile, every LDS round trip separated by a workgroup barrier
    tb = 13 if ft == np.float32 else 12
    tile = np.concatenate([np.arange(5), np.sort(rng.permutation(np.arange(5, n))[:tb - 5])]).astype(np.uint32)
    gates = [(rand_u(k, ct), rng.permutation(tile)[:k]) for k in (2, 3, 3, 4, 3, 1, 3)]  # few enough for the in-LDS tables
    re[:], im[:] = psi[0], psi[1]
    core.apply_blocked(re, im, tile, gates, n)
    print(f'blocked_{ft.__name__}_' + core.last_kernel_desc().split()[-1].replace('=', ''), digest(re, im))
    # low-bit swaps and general bit permutations through LDS tiles
    for s in (3, 8, 11, 13):
        re[:] = psi[0]
        core.swap(re, rng.permutation(s), n)
        print(f'swap_{ft.__name__}_{s}', digest(re))
    perm = rng.permutation(n)
    re[:] = psi[0]
    core.permute_bits(re, im, perm, n)
    print(f'permute_{ft.__name__}', digest(im))
    re[:], im[:] = psi[0], psi[1]
    print(f'norm2_{ft.__name__}', digest(np.float64(core.norm2(re, im))))
    # (bins are accumulated with atomic adds: the order of the floating-point additions follows the wave schedule, on the
    # device as here -- compared to 1e-9 instead of bit by bit)
    print(f'prob_{ft.__name__}', digest(np.round(np.asarray(core.probabilities(re, im, [1, 6, 12], n)) / (1 << n), 9)))
    v = core.vdot(re, im, im, re)
    print(f'vdot_{ft.__name__}', digest(np.complex128(v)))
    free()
# the emulator's own check: a kernel with a missing barrier must come out differently under the greedy schedules
import ctypes  # noqa: E402
out = (ctypes.c_uint32 * 8)()
for wb in (1, 0):
    core._lib.hq_emu_selftest_race(wb, out)
    print(f'selftest_race_barrier{wb}', '-'.join(str(x) for x in out))
