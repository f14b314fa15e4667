"""Prototype of placement autotuning: draw K VMM mappings for the n-qubit planes, probe each with a few gates,
keep the fastest, then run the real 900-gate circuit on the winner and on a plain torch allocation."""
import os
import sys
import time

os.environ.setdefault('OPENBLAS_NUM_THREADS', '8')
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hybridq_amd import core  # noqa: E402
from hybridq_amd.circuits import haar_unitary, rqc_1q2q  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
K = int(sys.argv[2]) if len(sys.argv) > 2 else 6
hold = (sys.argv[3] == 'hold') if len(sys.argv) > 3 else True
N = 1 << n
core.use_torch_stream()
torch.zeros(1, device='cuda')
rng = np.random.default_rng(0)
PROBE = [([3], haar_unitary(2, rng)), ([n // 2], haar_unitary(2, rng)), ([n - 1], haar_unitary(2, rng)), ([5, n - 3], haar_unitary(4, rng))]


def probe(re, im):
    core.init_state(re, im, 'plus')
    for pos, U in PROBE:
        core.apply_U(re, im, U, pos, n)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(2):
        for pos, U in PROBE:
            core.apply_U(re, im, U, pos, n)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (2 * len(PROBE))


def circuit_ms(re, im):
    gates = rqc_1q2q(n, depth=40, seed=n)
    plan = [(U, [n - 1 - q for q in reversed(qs)]) for U, qs in gates]
    core.init_state(re, im, 'basis', 0)
    for U, pos in plan[:50]:
        core.apply_U(re, im, U, pos, n)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for U, pos in plan:
        core.apply_U(re, im, U, pos, n)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / len(plan)


t0 = time.time()
cands = []
gran = 8 << 20
pp = (4 * N) // gran
for k in range(K):
    slots = [int(x) for x in np.random.default_rng(100 + k).permutation(2 * pp)] if k % 2 == 0 else list(range(2 * pp))
    buf = core.DeviceBuffer(2 * pp * gran, scattered=gran if k % 2 == 0 else (2 << 20), va_slots=slots if k % 2 == 0 else list(range(2 * (4 * N) // (2 << 20))))
    re = torch.as_tensor(buf.view(0, (N,), '<f4'), device='cuda')
    im = torch.as_tensor(buf.view(4 * N, (N,), '<f4'), device='cuda')
    ms = probe(re, im)
    print(f'draw {k} ({"8 MiB shuffled" if k % 2 == 0 else "2 MiB sequential"}): probe {ms:.3f} ms/gate', flush=True)
    cands.append((ms, buf, re, im))
    if not hold and len(cands) > 1:
        cands.sort(key=lambda c: c[0])
        _, b, r, i = cands.pop()
        del r, i
        b.free()
cands.sort(key=lambda c: c[0])
best = cands[0]
for _, b, r, i in cands[1:]:
    del r, i
    b.free()
print(f'search took {time.time() - t0:.2f} s; best probe {best[0]:.3f}', flush=True)
print(f'winner: probe again {probe(best[2], best[3]):.3f}; 900-gate circuit {circuit_ms(best[2], best[3]):.3f} ms/gate', flush=True)
raw = torch.empty((2, N + 3072), dtype=torch.float32, device='cuda')
print(f'torch.empty: probe {probe(raw[0, :N], raw[1, :N]):.3f}; 900-gate circuit {circuit_ms(raw[0, :N], raw[1, :N]):.3f} ms/gate', flush=True)
print(f'winner again: 900-gate circuit {circuit_ms(best[2], best[3]):.3f} ms/gate', flush=True)
