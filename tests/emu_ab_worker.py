"""Worker of tests/test_emu_kernels.py::test_pipelined_loops_are_bit_identical: digests of cache-blocked passes (every shape of
inner gate) and of k = 4..10 gates through the tile GEMM kernel on the emulated device; the caller runs it against the
default switches (the loops hardware has run) and under HQ_BLOCKED_PIPE=1 HQ_GEMM_PIPE=1 HQ_BIG_TWOBASE=1 (the operand-ahead
loops of rounds 4-5) and compares line by line; k = 5, 6 through the role kernel as well (complex128 k = 6: two LDS bases)."""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import emu_util  # noqa: E402

core = emu_util.emu_core()
rng = np.random.default_rng(5)


def rand_u(k, ct):
    d = 1 << k
    return ((rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d))) / np.sqrt(2.0 * d)).astype(ct)


def digest(*a):
    return hashlib.sha1(b''.join(np.ascontiguousarray(x).tobytes() for x in a)).hexdigest()[:16]


for ft in (np.float32, np.float64):
    ct = np.complex64 if ft == np.float32 else np.complex128
    tb = 13 if ft == np.float32 else 12
    n = tb + 2
    re, im, free = emu_util.device_planes(core, n, ft)
    psi = (rng.standard_normal(1 << n) + 1j * rng.standard_normal(1 << n)).astype(ct)
    tile = np.concatenate([np.arange(5), np.sort(rng.permutation(np.arange(5, n))[:tb - 5])]).astype(np.uint32)
    # inner gates of every shape: k = 2, 3, 4 with 0, 1, 2 targets among the vector-component bits
    shapes = [[int(tile[6]), int(tile[9])], [0, int(tile[7])], [1, 0], [int(tile[5]), int(tile[8]), int(tile[11])], [0, int(tile[6]), int(tile[10])],
              [1, int(tile[7]), 0], [int(tile[5]), int(tile[7]), int(tile[9]), int(tile[12 if tb > 12 else 11])], [0, int(tile[6]), int(tile[8]), int(tile[10])],
              [0, 1, int(tile[9]), int(tile[11])]]
    for i in range(0, len(shapes), 3):
        gates = [(rand_u(len(p), ct), np.array(p, dtype=np.uint32)) for p in shapes[i:i + 3]]
        re[:], im[:] = psi.real, psi.imag
        core.apply_blocked(re, im, tile, gates, n)
        print(f'blocked_{ft.__name__}_{i}', core.last_kernel_desc().replace(' ', '_'), digest(re, im), flush=True)
    for k in range(4, 11):
        pos = [int(p) for p in rng.permutation(n)[:k]]
        U = rand_u(k, ct)
        re[:], im[:] = psi.real, psi.imag
        core.set_apply_mode('gemm')
        try:
            core.apply_U(re, im, U, pos, n)
        finally:
            core.set_apply_mode('auto')
        print(f'gemm_{ft.__name__}_{k}', core.last_kernel_desc().replace(' ', '_'), digest(re, im), flush=True)
    for k in (5, 6):  # role kernel (apply_mfma_big_kernel); no target on index bit 0: the widest instantiation
        pos = [int(p) for p in 1 + rng.permutation(n - 1)[:k]]
        U = rand_u(k, ct)
        re[:], im[:] = psi.real, psi.imag
        core.apply_U(re, im, U, pos, n)
        print(f'role_{ft.__name__}_{k}', core.last_kernel_desc().replace(' ', '_'), digest(re, im), flush=True)
    free()
