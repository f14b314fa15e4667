"""Randomised campaign over the C ABI on the EMULATED device (tests/emu; no GPU needed): apply_U in every mode with random k
and positions, cache-blocked passes with random tile shapes and gate lists, low-bit swaps, general bit permutations,
to_complex, probabilities / project / norm2 / vdot, initial states -- random state sizes, both precisions, each result
against numpy.  The library reads its switches once per process, so a campaign is run once per switch setting:

    python tools/emu_fuzz.py [seconds] [seed]                      # defaults
    HQ_BLOCKED_DIRECT=1 HQ_BLOCKED_BIG=1 python tools/emu_fuzz.py 600 3
    HQ_EMU_ORDER=random python tools/emu_fuzz.py 600 4             # adversarial wave schedules

Prints one line per failure (with everything needed to replay it) and a summary; exit code 1 if anything failed.
Test infrastructure: the product never loads the emulation."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import emu_util  # noqa: E402

core = emu_util.emu_core()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)


def ref_apply(psi, U, pos, n):
    k = len(pos)
    x = psi.astype(np.complex128).reshape((2,) * n)
    Ut = np.asarray(U, dtype=np.complex128).reshape((2,) * (2 * k))
    in_axes = [n - 1 - pos[j] for j in reversed(range(k))]
    y = np.tensordot(Ut, x, axes=(list(range(k, 2 * k)), in_axes))
    return np.moveaxis(y, list(range(k)), in_axes).reshape(-1)


def rand_u(k, ct):
    d = 1 << k
    return ((rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d))) / np.sqrt(2.0 * d)).astype(ct)


def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


fails, counts, kernels = 0, {}, {}
t_end = time.time() + budget
while time.time() < t_end:
    ft = np.float32 if rng.random() < 0.6 else np.float64
    ct = np.complex64 if ft == np.float32 else np.complex128
    tol = 3e-6 if ft == np.float32 else 2e-13
    n = int(rng.integers(10, 18))
    re, im, free = emu_util.device_planes(core, n, ft)
    try:
        for _ in range(6):
            psi = (rng.standard_normal(1 << n) + 1j * rng.standard_normal(1 << n)).astype(ct)
            re[:], im[:] = psi.real, psi.imag
            what = rng.choice(['apply', 'apply', 'blocked', 'blocked', 'blocked', 'swap', 'permute', 'aux'])
            desc, ok = '', True
            if what == 'apply':
                mode = str(rng.choice(['auto', 'auto', 'direct', 'mfma', 'generic', 'tile', 'gemm', 'naive']))
                kmax = 10 if mode in ('auto', 'gemm', 'generic') else 6
                k = int(rng.integers(1, min(kmax, n - 4) + 1))
                pos = [int(p) for p in rng.permutation(n)[:k]]
                if rng.random() < 0.3:
                    pos = sorted(pos)
                if rng.random() < 0.25:  # low targets: the vector-component and slot bits
                    pos = [int(p) for p in rng.permutation(min(n, k + 3))[:k]]
                U = rand_u(k, ct)
                core.set_apply_mode(mode)
                try:
                    core.apply_U(re, im, U, pos, n)
                finally:
                    core.set_apply_mode('auto')
                err = rel(re + 1j * im, ref_apply(psi, U, pos, n))
                ok = err < tol * (1 if k < 7 else 4)
                desc = f'apply mode={mode} k={k} pos={pos} [{core.last_kernel_desc()}] err={err:.2e}'
            elif what == 'blocked':
                cb = 2 if ft == np.float32 else 1
                tb = int(rng.integers(10, min(n, 14 if ft == np.float32 else 13) + 1))
                focus = rng.random() < 0.6  # the tile sizes of the prefetching kernels (64 / 128 KiB), few gates: direct first gate, 1024 threads
                if focus:
                    tb = min(n, (13 if ft == np.float32 else 12) + int(rng.integers(0, 2)))
                low = int(rng.integers(cb, 6))
                rest = np.arange(low, n)
                tile = np.concatenate([np.arange(low), np.sort(rng.permutation(rest)[:tb - low])]).astype(np.uint32)
                ng = int(rng.integers(2, 6)) if focus else int(rng.integers(1, 11))
                gates = []
                for _g in range(ng):
                    k = int(rng.integers(1, 5))
                    src = tile[:max(k, 4)] if rng.random() < 0.3 else tile  # a third of the gates sit on the lowest tile bits
                    gates.append((rand_u(k, ct), rng.permutation(src)[:k].astype(np.uint32)))
                core.apply_blocked(re, im, tile, gates, n)
                want = psi
                for U, pos in gates:
                    want = ref_apply(want, U, [int(p) for p in pos], n)
                err = rel(re + 1j * im, want)
                ok = err < tol * 2
                desc = (f'blocked n={n} tile={tile.tolist()} gates={[(len(p), p.tolist()) for _, p in gates]} '
                        f'[{core.last_kernel_desc()}] err={err:.2e}')
            elif what == 'swap':
                s = int(rng.integers(1, min(n, 16) + 1))
                perm = rng.permutation(s)
                arr = re if rng.random() < 0.5 else None
                if arr is None:  # the integer element types of the reference's swap exports
                    dt = rng.choice([np.int32, np.int64, np.uint32, np.uint64]) if ft == np.float32 else np.int64
                    arr = np.arange(1 << n).astype(dt) * 3 + 1
                old = arr.copy()
                core.swap(arr, perm, n)
                x = np.arange(1 << n)
                src = x & ~((1 << s) - 1)
                for i in range(s):
                    src |= ((x >> i) & 1) << int(perm[i])
                ok = bool(np.array_equal(arr, old[src]))
                desc = f'swap n={n} s={s} perm={perm.tolist()} dtype={arr.dtype}'
            elif what == 'permute':
                perm = rng.permutation(n)
                if rng.random() < 0.5:  # only a few bits move
                    perm = np.arange(n)
                    idx = rng.permutation(n)[:int(rng.integers(2, 5))]
                    perm[idx] = np.roll(perm[idx], 1)
                src = re.copy()
                core.permute_bits(re, im, perm, n)
                x = np.arange(1 << n)
                s_idx = np.zeros_like(x)
                for i in range(n):
                    s_idx |= ((x >> i) & 1) << int(perm[i])
                ok = bool(np.array_equal(im, src[s_idx]))
                desc = f'permute n={n} perm={perm.tolist()}'
            else:
                k = int(rng.integers(1, min(n, 6) + 1))
                pos = [int(p) for p in rng.permutation(n)[:k]]
                p = np.asarray(core.probabilities(re, im, pos, n))
                a2 = np.abs(psi.astype(np.complex128)) ** 2
                x = np.arange(1 << n)
                outcome = np.zeros(1 << n, dtype=np.int64)
                for j, q in enumerate(pos):  # bit j of the outcome <-> index bit pos[j]
                    outcome |= ((x >> q) & 1) << j
                want = np.bincount(outcome, weights=a2, minlength=1 << k)
                nrm = core.norm2(re, im)
                v = core.vdot(re, im, im, re)
                vw = np.vdot(psi.astype(np.complex128), (psi.imag + 1j * psi.real).astype(np.complex128))
                out = np.empty(2 << n, dtype=ft)
                core.to_complex(re, im, out)
                ok = (rel(p, want) < 1e-5 and abs(nrm - a2.sum()) < 1e-5 * a2.sum() and abs(v - vw) < 1e-5 * max(1.0, abs(vw))
                      and np.array_equal(out.view(ct), psi))
                desc = f'aux n={n} pos={pos} prob_err={rel(p, want):.2e} norm {nrm} vs {a2.sum()} vdot {v} vs {vw}'
                # projection onto an outcome of `pos` (with a scale), then a random product state of '0', '1', '+', '-'
                st, sc = int(rng.integers(0, 1 << k)), float(rng.uniform(0.5, 2.0))
                core.project(re, im, pos, st, sc, n)
                wantp = np.where(outcome == st, psi.astype(np.complex128) * sc, 0)
                chars = ''.join(rng.choice(list('01+-'), size=n))
                ok = ok and rel(re + 1j * im, wantp) < 1e-6
                core.init_product_state(re, im, chars)
                fac = {'0': np.array([1.0, 0.0]), '1': np.array([0.0, 1.0]), '+': np.array([1.0, 1.0]) / np.sqrt(2), '-': np.array([1.0, -1.0]) / np.sqrt(2)}
                wants = np.ones(1)
                for pbit in reversed(range(n)):  # index bit p carries the factor chars[p]: most significant bit first
                    wants = np.kron(wants, fac[chars[pbit]])
                ok = ok and rel(re + 1j * im, wants) < 1e-6 and float(np.abs(im).max()) == 0.0
                desc += f' project state={st} scale={sc:.3f} product={chars}'
            counts[what] = counts.get(what, 0) + 1
            if what in ('apply', 'blocked'):  # which kernel the dispatch took
                d = core.last_kernel_desc()
                key = d.split('(')[0].split(' tb=')[0].strip() + (' direct' if d.endswith('direct') else '')
                kernels[key] = kernels.get(key, 0) + 1
            if not ok:
                fails += 1
                print('FAIL', ft.__name__, f'n={n}', desc, flush=True)
    finally:
        free()
env = {k: v for k, v in os.environ.items() if k.startswith('HQ_') and k not in ('HQ_HIP_LIBRARY',)}
print(f'emu_fuzz seed {seed}, {budget:.0f} s, switches {env}: {sum(counts.values())} cases { {str(k): v for k, v in counts.items()} }, failures: {fails}')
for k in sorted(kernels):
    print(f'    {kernels[k]:6d}  {k}')
sys.exit(1 if fails else 0)
