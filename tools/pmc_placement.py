"""Counter comparison of the placement effect: the SAME gate kernel on torch memory, on a slow VMM draw and on a
fast VMM draw, in one process (run under rocprofv3 --pmc ...; the last 6 launches of the gate kernel are
torch, torch, slow, slow, fast, fast)."""
import os
import sys

os.environ.setdefault('OPENBLAS_NUM_THREADS', '8')
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hybridq_amd import core  # noqa: E402
from hybridq_amd.circuits import haar_unitary  # noqa: E402
import placement_util as sim  # noqa: E402

n = 30
N = 1 << n
core.use_torch_stream()
raw = torch.empty((2, N + 3072), dtype=torch.float32, device='cuda')
tplanes = raw[:, :N]
cands = []
for k in range(8):
    owner = sim._VmmPlanes(8 * N + 24576, (2, N + 3072), '<f4', (2 << 20) if k % 2 else (8 << 20), 0 if k % 2 else 100 + k)
    r = torch.as_tensor(owner, device='cuda')[:, :N]
    ms = sim._probe_ms(r, n, np.float32)
    cands.append((ms, r, owner.layout))
    print(f'draw {k}: {ms:.3f} ms  {owner.layout}', flush=True)
cands.sort(key=lambda c: c[0])
fast, slow = cands[0], cands[-1]
print(f'torch probe {sim._probe_ms(tplanes, n, np.float32):.3f}; slow {slow[0]:.3f} ({slow[2]}); fast {fast[0]:.3f} ({fast[2]})', flush=True)
U = haar_unitary(2, np.random.default_rng(1))
for name, pl in (('torch', tplanes), ('slow', slow[1]), ('fast', fast[1])):
    core.init_state(pl[0], pl[1], 'plus')
core.sync()
print('MARK', flush=True)
for name, pl in (('torch', tplanes), ('slow', slow[1]), ('fast', fast[1])):
    for _ in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        core.apply_U(pl[0], pl[1], U, [12], n)
        e1.record()
        torch.cuda.synchronize()
        print(f'{name}: {e0.elapsed_time(e1):.3f} ms', flush=True)
