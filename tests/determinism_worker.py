"""Worker of tests/test_gpu_determinism.py (run as a script in a fresh process: the library reads its HQ_* switches
once).  Runs every kernel that synchronises through LDS / barriers / hand-placed wait counts REPS times on the same
input and checks that all repetitions are bit-identical; prints one line `name sha256` per kernel (first repetition)."""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import emu_boot  # noqa: E402

EMU = emu_boot.maybe_install()  # host emulation (HQ_EMU_GPU_SUITE=1): small states, few repetitions, RANDOM wave schedules --
#                                 there a repetition that differs means a race, not a hardware hiccup
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hybridq_amd import core  # noqa: E402
from hybridq_amd.blocking import plan_blocked  # noqa: E402
from hybridq_amd.circuits import haar_unitary, rqc_1q2q  # noqa: E402

REPS = int(os.environ.get('HQ_DET_REPS', '3' if EMU else '30'))
N32, N64 = (15, 14) if EMU else (22, 21)
core.use_torch_stream()
rng = np.random.default_rng(17)


def digest(t):
    return hashlib.sha256(t.detach().cpu().numpy().tobytes()).hexdigest()[:24]


def check(name, make_input, run):
    """run(planes) mutates / returns a tensor; every repetition starts from an identical copy of the input."""
    first = None
    for r in range(REPS):
        out = run(make_input())
        core.sync()
        if first is None:
            first = out.clone()
        elif not torch.equal(out, first):
            bad = int((out != first).sum())
            print(f'NONDETERMINISTIC {name}: repetition {r} differs from repetition 0 in {bad} elements', flush=True)
            sys.exit(1)
    print(f'{name} {digest(first)}', flush=True)


for ct, ft, n in (('complex64', torch.float32, N32), ('complex128', torch.float64, N64)):
    base = torch.from_numpy(rng.standard_normal((2, 1 << n))).to(ft).cuda()
    base /= base.norm()

    def fresh():
        return base.clone()

    # cache-blocked passes (apply_blocked_kernel: tile in LDS, one barrier per inner gate, table-driven gates, prefetch)
    gates = rqc_1q2q(n, depth=12, seed=5)
    ident = {q: n - 1 - q for q in range(n)}
    tb = 13 if ct == 'complex64' else 12
    bops = plan_blocked(gates, ident, n, tile_bits=tb, low_bits=tb - 8, complex_type=ct)
    packed = [('B', op[1], core.pack_blocked(op[2], ct)) if op[0] == 'B' else op for op in bops]

    def run_blocked(pl):
        for op in packed:
            if op[0] == 'G':
                core.apply_U(pl[0], pl[1], op[1], op[2], n)
            else:
                core.apply_blocked(pl[0], pl[1], op[1], packed=op[2], n_qubits=n)
        return pl
    check(f'{ct} blocked ({sum(1 for o in bops if o[0] == "B")} passes)', fresh, run_blocked)
    # k = 5, 6 (apply_mfma_big_kernel: operand table in LDS, barrier-phased halves) and k = 7, 8 (apply_gemm_kernel)
    for k in (5, 6, 7, 8):
        for pos in (sorted(int(p) for p in rng.permutation(n)[:k]), list(range(k)), list(range(n - k, n))):
            U = np.ascontiguousarray(haar_unitary(1 << k, rng), dtype=ct)

            def run_k(pl, U=U, pos=pos):
                core.apply_U(pl[0], pl[1], U, pos, n)
                return pl
            check(f'{ct} apply_U k={k} pos={pos} [{core.last_kernel()}]', fresh, run_k)

for dt, n in ((torch.float32, 22), (torch.float64, 21)):
    data = torch.arange(1 << n, device='cuda').to(dt)
    for s in (6, 8, 12, 13, 14, 15, 16):
        pos = np.roll(np.arange(s), 3) if s == 16 else rng.permutation(s)

        def run_swap(a, pos=pos):
            core.swap(a, pos, n)
            return a
        check(f'{str(dt)[6:]} swap s={s}', data.clone, run_swap)
    for name, perm in (('random', rng.permutation(n)), ('reversal', np.arange(n)[::-1].copy()),
                       ('evict', np.array([b for b in range(n) if b not in (3, 9, 15)] + [3, 9, 15]))):
        dst = torch.empty_like(data)

        def run_perm(a, perm=perm, dst=dst):
            dst.zero_()
            core.permute_bits(a, dst, perm, n)
            return dst
        check(f'{str(dt)[6:]} permute_bits {name}', lambda: data, run_perm)
    core.shard_free()
    two = torch.stack([data[:1 << (n - 1)], -data[:1 << (n - 1)]]).contiguous()
    out2 = torch.empty_like(two)
    perm = rng.permutation(n - 1)

    def run_pack(a):
        out2.zero_()
        core.exchange(a[0], a[1], out2[0], out2[1], perm, n - 1)
        return out2
    check(f'{str(dt)[6:]} exchange pack', lambda: two, run_pack)
print('DETERMINISTIC', flush=True)
