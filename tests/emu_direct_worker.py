"""Worker of tests/test_emu_kernels.py::test_blocked_direct_pass (a fresh process: the library reads HQ_BLOCKED_* once).
Cache-blocked passes on EMULATED device memory with few resident workgroups (HQ_BLOCKED_GRID), so that every workgroup
walks several tiles -- the loop in which apply_blocked_direct_kernel streams the previous tile out while its first gate
multiplies the prefetched one.  Prints `case kernel-description max-rel-error sha1` per pass."""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import emu_util  # noqa: E402

core = emu_util.emu_core()


def ref_apply(psi, U, pos, n):
    k = len(pos)
    x = psi.astype(np.complex128).reshape((2,) * n)
    Ut = np.asarray(U, dtype=np.complex128).reshape((2,) * (2 * k))
    in_axes = [n - 1 - pos[j] for j in reversed(range(k))]
    y = np.tensordot(Ut, x, axes=(list(range(k, 2 * k)), in_axes))
    return np.moveaxis(y, list(range(k)), in_axes).reshape(-1)


def rand_u(k, ct, rng):
    d = 1 << k
    return ((rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d))) / np.sqrt(2.0 * d)).astype(ct)


CASES = [(3, 3, 2, 4, 3), (4, 2, 3, 3), (1, 3, 3), (3, 3), (3, 3, 2), (4, 2), (4, 3), (4, 2, 2)]  # 5-7: a k = 4 FIRST gate (round 5)
for ft in (np.float32, np.float64):
    ct = np.complex64 if ft == np.float32 else np.complex128
    tb = (13 if ft == np.float32 else 12) + (os.environ.get('HQ_TEST_BIG_TILES', os.environ.get('HQ_BLOCKED_BIG')) == '1')  # 128 KiB tiles
    n = tb + 4  # 16 tiles
    re, im, free = emu_util.device_planes(core, n, ft)
    for case, ks in enumerate(CASES):
        rng = np.random.default_rng(100 + case)
        tile = np.concatenate([np.arange(5), np.sort(rng.permutation(np.arange(5, n))[:tb - 5])]).astype(np.uint32)
        gates = [(rand_u(k, ct, rng), rng.permutation(tile)[:k]) for k in ks]
        if case == 3:  # a first gate with a target among the vector-component bits and one on a low vector bit
            gates[0] = (gates[0][0], np.array([1, 3, int(tile[-1])], dtype=np.uint32))
        if case == 4:  # both vector-component bits among the targets: the most wave-iterations a gate can have (128 on 128 KiB tiles)
            gates[1] = (gates[1][0], np.array([0, int(tile[-2]), 1], dtype=np.uint32))
            gates[2] = (gates[2][0], np.array([1, 0], dtype=np.uint32))
        if case == 6:  # k = 4 first gate with one target among the vector-component bits
            gates[0] = (gates[0][0], np.array([int(tile[-1]), 0, int(tile[7]), int(tile[9])], dtype=np.uint32))
        if case == 7:  # ... and with as many as the precision has (complex64: both, complex128: the one)
            gates[0] = (gates[0][0], np.array([0, int(tile[8]), 1 if ft == np.float32 else int(tile[6]), int(tile[-2])], dtype=np.uint32))
        psi = (rng.standard_normal(1 << n) + 1j * rng.standard_normal(1 << n)).astype(ct)
        re[:], im[:] = psi.real, psi.imag
        core.apply_blocked(re, im, tile, gates, n)
        want = psi
        for U, pos in gates:
            want = ref_apply(want, U, [int(p) for p in pos], n)
        err = np.abs((re + 1j * im) - want).max() / np.abs(want).max()
        desc = core.last_kernel_desc().replace(' ', '_')
        print(f'{ft.__name__}_{case} {desc} {err:.3e} {hashlib.sha1(re.tobytes() + im.tobytes()).hexdigest()[:16]}', flush=True)
    free()
