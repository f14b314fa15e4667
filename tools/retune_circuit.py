"""The 900-gate n = 30 benchmark circuit under the library's policy switches, on the tuned placement:
non-temporal policy, dummy placement, VALU-only (which policies still hold on VMM-mapped memory?)."""
import os
import sys
import time

os.environ.setdefault('OPENBLAS_NUM_THREADS', '8')
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hybridq_amd import core  # noqa: E402
from hybridq_amd.circuits import rqc_1q2q  # noqa: E402
from hybridq_amd.simulation import EvolutionState  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
gates = rqc_1q2q(n, depth=40, seed=n)
state = EvolutionState(list(range(n)), complex_type='complex64', initial_state='0' * n)
plan = [(U, [state.map[q] for q in reversed(qs)]) for U, qs in gates]


def run():
    for U, pos in plan:
        core.apply_U(state.planes[0], state.planes[1], U, pos, n)


for modes in (['auto'], ['nt=0'], ['nt=1'], ['dummy=comp'], ['dummy=low'], ['dummy=high'], ['direct'], ['direct', 'nt=1'], ['auto']):
    for m in ('auto', 'nt=auto', 'dummy=auto'):
        core.set_apply_mode(m)
    for m in modes:
        core.set_apply_mode(m)
    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    print(f'{"+".join(modes):<16} {ms:8.1f} ms per circuit  {ms / len(plan):.4f} ms/gate  {16.0 * (1 << n) * len(plan) / ms / 1e9:.3f} TB/s', flush=True)
